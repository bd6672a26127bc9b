#!/usr/bin/env python3
"""bench.py -- SQP(=LM outer)-iterations/s of the MI355X-native NLP inner loop on the BASELINE configurations.

Default workload (BASELINE.json configs[2], SURVEY.md 8d, the one the metric is quoted on): unicycle point-to-point OCP, nx=3 nu=2,
FiniteDifferencesGrid N=100, Crank-Nicolson, fp64, batch=1024 independent seeded instances PER GPU, 10 LM iterations per solve
(reference default, no early exit).  One "step" = device-to-device re-arm of the initial trajectories (inputs are resident in HBM
before the timed region starts) + one corbo_hip_solve of the whole resident batch (= batch x 10 SQP iterations) + the results
(trajectories, chi2, status) copied into host-visible pinned memory ("to last result resident on host-visible memory").

    python bench.py [--config 1|2|3|5] [--gpus N] [--steps K] [--warmup W] [--batch B] [--iterations I]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

--config selects another BASELINE.json configuration (parity-test cases; their lines are diagnostics, the driver's line is config 3):
  1  Van-der-Pol, FiniteDifferencesGrid N=20, batch 1           (configs[0]: plumbing; single-OCP latency through the boundary)
  2  double integrator, time-optimal, variable grid N=50, batch 1, 5 solves per step (configs[1]: single-OCP latency)
  5  quadrotor nx=12 nu=4, multiple shooting N=200 + RK4, batch 512 (configs[4]: big-block family, fp64 MFMA factorisation)

Rank 0 prints ONE JSON line.  Multi-GPU: the batch is the sharding unit (independent OCPs, no data-path collective);
every rank solves its own `batch` instances (weak scaling), RCCL is used only for the barrier, the max-over-ranks time, the
per-rank times and the reduction of the solution statistics.
"""
from __future__ import annotations

import argparse
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0      # MI355X HBM3E (MI355X_MICROARCH.md)
PEAK_F64_MFMA_TFLOPS = 78.6  # gfx950 dense fp64 matrix peak
PROFILE_ROUND = "r06"         # prefix of the committed rocprofv3 summaries under profiles/ the line cross-checks itself against


def workload(cfg: int, batch: int, first: int = 0):
    """(name, descriptor, weights, x0, xf, solves per step, ref_driver scenario, default batch) of a BASELINE configuration."""
    from control_box_rst_amd import problems
    if cfg == 3:
        d = problems.unicycle_desc()
        x0, xf = problems.unicycle_instances(batch, first=first)
        return dict(name="configs[2]: unicycle point-to-point nx=3 nu=2, FiniteDifferencesGrid N=100 Crank-Nicolson", desc=d,
                    weights=problems.UNICYCLE_WEIGHTS, x0=x0, xf=xf, solves=1, scenario="unicycle")
    if cfg == 5:
        d = problems.quad_desc()
        x0, xf = problems.quad_instances(batch, first=first)
        return dict(name="configs[4]: quadrotor nx=12 nu=4, MultipleShootingGrid N=200 RK4, u box + keep-out ball per stage", desc=d,
                    weights=problems.QUAD_WEIGHTS, x0=x0, xf=xf, solves=1, scenario="quad")
    if cfg == 2:
        d = problems.dint_desc()
        rng = np.random.default_rng(20260928 + first)
        x0 = np.zeros((batch, 2))
        xf = np.tile([1.0, 0.0], (batch, 1))
        if batch > 1:
            x0[1:] = rng.uniform(-0.1, 0.1, (batch - 1, 2))
        return dict(name="configs[1]: double integrator time-optimal, FiniteDifferencesVariableGrid N=50, x_f fixed, MinimumTime", desc=d,
                    weights=problems.DINT_WEIGHTS, x0=x0, xf=xf, solves=5, scenario="dint")
    if cfg == 1:
        d = problems.vdp_desc()
        rng = np.random.default_rng(20260928 + first)
        x0 = np.tile([1.0, 0.0], (batch, 1))
        if batch > 1:
            x0[1:] += rng.uniform(-0.5, 0.5, (batch - 1, 2))
        return dict(name="configs[0]: Van-der-Pol regulator, FiniteDifferencesGrid N=20", desc=d, weights=problems.VDP_WEIGHTS, x0=x0,
                    xf=np.zeros((batch, 2)), solves=1, scenario="vdp")
    raise SystemExit(f"unknown --config {cfg}")


DEFAULT_BATCH = {1: 1, 2: 1, 3: 1024, 5: 512}
REF_SAMPLE = {1: 2000, 2: 400, 3: 512, 5: 1}       # instances of the genuine reference's bounded sample (10-30 s of CPU work)
PORT_SAMPLE = {1: 4000, 2: 1000, 3: 256, 5: 64}


def cpu_baseline(cfg, w, opts, seconds_budget=20.0):
    """Reference CPU path timed on this box's host cores (rank 0, bounded sample of the same seeded workload).

    kind="reference": the genuine control_box_rst LevenbergMarquardtSparse path (oracle/_ref/ref_driver, compiled from the
    reference's own sources by oracle/Makefile in the build container; the binary travels, the sources do not).
    The C restatement (oracle/, kind="port", static sparsity pattern) is timed next to it and reported as `port_value`.
    """
    from oracle import oracle as O
    desc, solves = w["desc"], w["solves"]
    x0, xf = w["x0"], w["xf"]
    if len(x0) < max(REF_SAMPLE[cfg], PORT_SAMPLE[cfg]):   # batch-1 configurations: the sample is more instances of the same family
        big = workload(cfg, max(REF_SAMPLE[cfg], PORT_SAMPLE[cfg]))
        x0, xf = big["x0"], big["xf"]
    out = {}
    # --- port: oracle/liboracle.so, 1 thread
    p = O.OracleProblem(desc)
    n_port = min(len(x0), PORT_SAMPLE[cfg])
    X = np.stack([p.init_trajectory(x0[b], xf[b]) for b in range(n_port)])
    t0 = time.perf_counter()
    done = 0
    chunk = max(1, n_port // 8)
    if solves == 1:
        for lo in range(0, n_port, chunk):
            hi = min(n_port, lo + chunk)
            O.solve_batch(desc, X[lo:hi], xf[lo:hi], opts)
            done = hi
            if time.perf_counter() - t0 > seconds_budget / 2:
                break
    else:
        for b in range(n_port):
            q = O.OracleProblem(desc)
            q.set_data(X[b], xref=xf[b])
            for i in range(solves):
                q.solve(opts, new_run=(i == 0))
            done = b + 1
            if time.perf_counter() - t0 > seconds_budget / 2:
                break
    t_port = time.perf_counter() - t0
    port_value = done * opts.iterations * solves / t_port
    out["port_value"] = port_value
    out["port_sample"] = f"{done} seeded instances x {solves} solve(s) x {opts.iterations} LM iterations, oracle/liboracle.so, 1 thread"
    # --- port on all host cores: one worker process per core, started together (oracle/port_worker.py; headline config only)
    cores = min(os.cpu_count() or 1, 128)
    if cfg == 3 and cores > 1:
        per = 1024
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "port_worker.py"), str(i * per), str(per), str(opts.iterations)],
                                  stdout=subprocess.PIPE) for i in range(cores)]
        res = []
        for pr in procs:
            so, _ = pr.communicate(timeout=300)
            if pr.returncode == 0:
                res.append(json.loads(so.decode().strip().splitlines()[-1]))
        if len(res) == cores:
            out["port_allcores_value"] = sum(r["n"] * r["iterations"] for r in res) / max(r["seconds"] for r in res)
            out["port_allcores_sample"] = (f"{cores} worker processes x {per} seeded instances x {opts.iterations} LM iterations, "
                                           f"oracle/liboracle.so, slowest worker {max(r['seconds'] for r in res):.2f} s")
            out["port_allcores_cores"] = cores
    # --- genuine reference, 1 thread
    drv = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    if os.path.exists(drv) and os.access(drv, os.X_OK):
        n_ref = min(len(x0), REF_SAMPLE[cfg])
        with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
            np.savetxt(f, np.hstack([x0[:n_ref], xf[:n_ref]]), fmt="%.17g")
            path = f.name
        try:
            r = json.loads(subprocess.check_output([drv, "bench", f"scenario={w['scenario']}", f"instances={path}", f"iters={opts.iterations}",
                                                    f"solves={solves}", f"N={desc.N}"], timeout=600))
            out.update({"value": r["iter_per_s"], "unit": "SQP-iterations/s", "cores": 1, "kind": "reference",
                        "sample": f"{r['batch']} seeded instance(s) x {solves} solve(s) x {opts.iterations} LM iterations = {r['solve_seconds']:.2f} s of "
                                  "LevenbergMarquardtSparse::solve (genuine reference, oracle/_ref/ref_driver), 1 thread",
                        "ms_per_ocp": 1e3 * r["solve_seconds"] / r["batch"]})
        finally:
            os.unlink(path)
    if "value" not in out:
        out.update({"value": port_value, "unit": "SQP-iterations/s", "cores": 1, "kind": "port", "sample": out["port_sample"]})
    return out


def profile_kernel_avg_ns(pattern, kernel_substr):
    """Average duration [ns] of a kernel in the committed rocprofv3 --kernel-trace --stats summary (profiles/), or None."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        try:
            for r in csv.DictReader(open(path)):
                if kernel_substr in r.get("Name", ""):
                    return float(r["AverageNs"]), int(r["Calls"]), os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None


def profile_kernel_max_ns(csv_name, kernels):
    """Sum over `kernels` of the longest launch (MaxNs = a launch with every instance active) in a committed rocprofv3 kernel-stats CSV."""
    path = os.path.join(ROOT, "profiles", csv_name)
    if not os.path.exists(path):
        return None
    tot = 0.0
    for k in kernels:
        rows = [r for r in csv.DictReader(open(path)) if k in r["Name"]]
        if not rows:
            return None
        tot += max(float(r["MaxNs"]) for r in rows)
    return tot, "profiles/" + csv_name


def load_profile_json(name, batch, N):
    path = os.path.join(ROOT, "profiles", name)
    if os.path.exists(path):
        try:
            j = json.load(open(path))
            if j.get("batch") == batch and j.get("N") == N:
                return j
        except Exception:
            pass
    return None


def counted_per_step(solver, solves):
    """Outer iterations of ONE step (re-arm + `solves` solves) that the device counted instead of executing (corbo_hip_stats.counted_iterations,
    option ff_converged): an untimed extra step with the statistics read after every solve."""
    solver.restore_instance_data()
    tot = 0
    for i in range(solves):
        solver.solve(new_run=(i == 0))
        tot += int(solver.get_stats()["counted_iterations"])
    return tot


def secondary_leg(cfg, iterations, device):
    """One BASELINE configuration other than the headline, measured in the same process (rank 0 of a 1-GPU run): ms per step,
    SQP-iterations/s, the chi2 sum of the timed workload next to the genuine reference's (tests/golden/bench_secondary.json,
    oracle/gen_golden.py secondary) and the roofline fraction of its dominant kernel(s).  No CPU baseline in this leg."""
    import torch
    from control_box_rst_amd.solver import BatchedLevenbergMarquardt
    B = DEFAULT_BATCH[cfg]
    w = workload(cfg, B)
    desc, solves = w["desc"], w["solves"]
    solver = BatchedLevenbergMarquardt(desc, B, device=device)
    solver.setIterations(iterations)
    solver.setPenaltyWeights(*w["weights"])
    X0 = solver.init_trajectory(w["x0"], w["xf"])
    solver.set_instance_data(X0, xref=w["xf"])
    solver.set_result_sink(True)
    steps, warmup = {1: 1500, 2: 300, 5: 10}[cfg], {1: 1000, 2: 200, 5: 2}[cfg]   # (batch 1: steady-state clocks need ~0.15 s of warm-up, see main())

    def step():
        solver.restore_instance_data()
        for i in range(solves):
            solver.solve(new_run=(i == 0))
        return solver.fetch_solution()

    def step_enqueued():   # the headline's timed region: re-arm inside the solve, no wait per step, every step's results delivered to pinned host memory
        for i in range(solves):
            solver.solve_async(new_run=(i == 0), rearm=(i == 0))

    for _ in range(warmup):
        step()
    solver.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    solver.synchronize(); torch.cuda.synchronize()
    dt_sync = time.perf_counter() - t0
    solver.get_timing(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step_enqueued()
    solver.synchronize(); torch.cuda.synchronize()
    solver.fetch_solution()
    dt = time.perf_counter() - t0
    solve_ms_sum, n_solves = solver.get_timing(reset=True)
    stats = solver.get_stats()
    _, chi2, status = solver.get_solution()
    iters_per_step = B * iterations * solves
    counted = counted_per_step(solver, solves)
    out = {"workload": f"{w['name']}, batch={B}, {solves} solve(s) x {iterations} LM iterations per step", "steps": steps, "warmup": warmup,
           "ms_per_step": 1e3 * dt / steps, "value": (iters_per_step - counted) * steps / dt, "unit": "SQP-iterations/s",
           "counted_iterations": counted, "value_computed": (iters_per_step - counted) * steps / dt, "value_incl_counted": iters_per_step * steps / dt,
           "counted_note": "`value` = EXECUTED outer iterations / s; outer iterations after a converged step (|delta| <= eps2/2) are counted, not executed (corbo_hip_stats.counted_iterations) -- value_incl_counted adds them (what the reference's loop count would credit)",
           "timed_region": "as the headline's: the steps enqueued back to back (corbo_hip_solve_async, re-arm inside the solve), every step's results delivered to pinned host memory, one wait at the end",
           "synchronous_steps": {"ms_per_step": 1e3 * dt_sync / steps, "value": (iters_per_step - counted) * steps / dt_sync,
                                 "what": "the same steps with the host waiting for every step (re-arm copy + solve + fetch): the timed region of this leg in rounds 3 - 4"},
           "chi2_sum": float(chi2.sum()), "ok_instances": int((status <= 1).sum()), "ms_per_solve_launch": solve_ms_sum / max(1, n_solves)}
    gpath = os.path.join(ROOT, "tests", "golden", "bench_secondary.json")
    if os.path.exists(gpath) and iterations == 10:
        g = json.load(open(gpath)).get(f"cfg{cfg}")
        if g and g["batch"] == B and g["N"] == desc.N:
            out["chi2_sum_reference"] = g["chi2_sum"]
            out["chi2_sum_rel_diff"] = abs(out["chi2_sum"] - g["chi2_sum"]) / abs(g["chi2_sum"])
    dims = solver.dims
    if cfg != 5:
        b_sweep = 8 * (dims.nv + 2 * dims.n + dims.m + dims.nnz)
        b_val = 8 * (dims.nv + 2 * dims.n + dims.m)
        alg = stats["jacobian_sweeps"] * b_sweep + max(0, stats["residual_sweeps"] - stats["jacobian_sweeps"]) * b_val
        ms = out["ms_per_solve_launch"]
        out["roofline"] = {"bound": "hbm", "kernel": "lm_pass_kernel", "bytes_per_launch": alg, "ms_per_launch": ms,
                           "achieved": alg / (ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": alg / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                           "note": "one OCP = one workgroup: a single-instance latency chain, not a bandwidth figure"}
    else:
        nxq, nuq = desc.nx, desc.nu
        rec = 3 * nxq * nxq + nuq * nuq + 2 * nuq * nxq + 2 * nxq + nuq + 2
        per_stage = 8 * ((nxq + nuq) + nxq + 2 * (nxq + nuq) + rec + rec + 2 * (nxq * nxq + nxq) + 2 * (nxq + nuq))
        f_ms = solver.time_factor(repeat=5)
        # SURVEY 8d's algorithmic bytes of one Jacobian sweep, 8 (n_vert + 2 n + m + nnz) per instance: what `frac` is priced on.  The kernels' OWN intermediate
        # traffic (the stage records big_stage_kernel writes and big_chain3_kernel re-reads) is the model figure `traffic_model`; the counters' figure is `traffic`.
        alg = 8 * (dims.nv + 2 * dims.n + dims.m + dims.nnz) * B
        pmc5 = load_profile_json(PROFILE_ROUND + "_cfg5_pmc.json", B, desc.N)
        out["roofline"] = {"bound": "hbm", "kernel": "big_stage_kernel + big_chain3_kernel (one factorisation of every instance)", "bytes_per_launch": alg,
                           "bytes_per_instance": alg // B, "bytes_definition": "SURVEY 8d: 8 (n_vert + 2 n + m + nnz) per instance per Jacobian sweep",
                           "ms_per_launch": f_ms, "achieved": alg / (f_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                           "frac": alg / (f_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                           "traffic_model": per_stage * desc.N * B, "traffic_model_note": "the pair's own reads + writes incl. the per-stage records handed from the stage kernel to the chain",
                           "traffic": pmc5["hbm_bytes_per_launch"] if pmc5 else None, "traffic_source": pmc5["source"] if pmc5 else None}
    del solver
    return out


def batch8192_leg(iterations, device, value_1024):
    """BASELINE cfg 4's whole batch (8192 unicycle OCPs) on ONE GPU through the instance queue of the run-to-completion kernel: CUs x 4 persistent workgroups
    pull instance after instance, so a finished instance's slot is refilled at once.  Throughput beyond the latency-bound headline point."""
    import torch
    from control_box_rst_amd.solver import BatchedLevenbergMarquardt
    B = 8192
    w = workload(3, B)
    solver = BatchedLevenbergMarquardt(w["desc"], B, device=device)
    solver.setIterations(iterations)
    solver.setPenaltyWeights(*w["weights"])
    solver.set_instance_data(solver.init_trajectory(w["x0"], w["xf"]), xref=w["xf"])
    solver.set_result_sink(True)
    steps, warmup = 20, 3
    for _ in range(warmup):
        solver.solve(rearm=True)
    solver.synchronize(); torch.cuda.synchronize()
    solver.get_timing(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        solver.solve_async(rearm=True)
    solver.synchronize(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms_sum, n = solver.get_timing(reset=True)
    st = solver.get_stats()
    _, chi2, status = solver.get_solution()
    passes = st["factorizations"] - st["counted_iterations"]
    # slot utilisation from the in-kernel phase totals (one more solve with the option on): busy cycles of all instances / (slots x kernel cycles)
    util = None
    try:
        solver.set_option("phase_cycles", 1)
        solver.solve(rearm=True)
        ms_pc = solver.get_stats()["solve_ms"]
        pc = solver.get_phase_cycles()
        solver.set_option("phase_cycles", 0)
        import torch as _t
        slots = 4 * _t.cuda.get_device_properties(device).multi_processor_count
        busy = float(pc[:, 0:3].sum())
        util = {"busy_cycles_sum": busy, "slots": slots, "kernel_ms": ms_pc,
                "mean_cycles_per_pass": busy / max(1, int(pc[:, 5].sum())),
                "what": "sum over instances of the cycles between an instance's first and last phase, per resident workgroup slot; divide by slots x kernel cycles for the slot utilisation (shader clock ~ 2.4 GHz): "
                        "busy / (slots x kernel_ms x 2.4e6)", "slot_utilisation_at_2p4GHz": busy / (slots * ms_pc * 2.4e6)}
    except Exception as e:   # (an older library without the option)
        util = {"error": repr(e)}
    value = B * iterations * steps / dt
    del solver
    return {"workload": "BASELINE cfg 4's batch on one GPU: unicycle nx=3 nu=2 N=100 CN, batch=8192, instance queue (persistent workgroups)", "steps": steps, "warmup": warmup,
            "ms_per_step": 1e3 * dt / steps, "ms_per_solve_launch": ms_sum / max(1, n), "value": value, "unit": "SQP-iterations/s",
            "vs_batch1024": value / value_1024 if value_1024 else None, "executed_passes_per_solve": int(passes),
            "chi2_sum": float(chi2.sum()), "ok_instances": int((status <= 1).sum()), "slot_occupancy": util}


def band_leg(device):
    """The band factorisation (free dt around a big-block model; integral-form constraint edges / control-deviation edges): the general, slower path
    (a measurement row).  Time-optimal 12-state quadrotor on the MultipleShootingVariableGrid, N = 100 (n = 1613 parameters, dt as border)."""
    import torch
    from control_box_rst_amd import problems
    from control_box_rst_amd.solver import BatchedLevenbergMarquardt, get_dims, get_structure
    out = {}
    # the same workload through the stage / partitioned-chain kernels with the dt column as a second right-hand side (round 5; the default route of
    # these descriptors for state blocks of 8 / 12 rows) -- the band route is kept for the A/B and for the structures only it covers
    for B in (1, 64, 512):
        d = problems.quad_desc(N=100, time_optimal=True)
        x0 = np.zeros((B, d.nx)); xf = np.zeros((B, d.nx)); xf[:, 0] = 2.0; xf[:, 1] = 1.0
        s = BatchedLevenbergMarquardt(d, B, device=device)
        s.setPenaltyWeights(100.0, 100.0, 100.0)
        s.set_instance_data(s.init_trajectory(x0, xf), xref=xf)
        s.solve(new_run=True); s.synchronize()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            s.restore_instance_data(); s.solve(new_run=True)
        s.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        st = s.get_stats()
        out[f"chain_route_batch{B}"] = {"ms_per_solve": ms, "passes": int(st["passes"]), "factorizations": int(st["factorizations"]),
                                        "chi2_sum": float(np.sum(s.get_solution()[1]))}
        del s
    from control_box_rst_amd.capi import ROUTE_FREE_DT_BAND
    for B in (1, 64):
        d = problems.quad_desc(N=100, time_optimal=True)
        x0 = np.zeros((B, d.nx)); xf = np.zeros((B, d.nx)); xf[:, 0] = 2.0; xf[:, 1] = 1.0
        s = BatchedLevenbergMarquardt(d, B, device=device, route=ROUTE_FREE_DT_BAND)   # (corbo_hip_create_routed)
        s.setPenaltyWeights(100.0, 100.0, 100.0)
        s.set_instance_data(s.init_trajectory(x0, xf), xref=xf)
        s.solve(new_run=True); s.synchronize()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            s.restore_instance_data(); s.solve(new_run=True)
        s.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        st = s.get_stats()
        chi2_band = float(np.sum(s.get_solution()[1]))
        s.restore_instance_data()
        f_ms = s.time_factor(repeat=3)
        rows, cols = get_structure(d)
        dims = get_dims(d)
        n = dims.n
        # half-bandwidth of H = J^T J in natural parameter order without the border (dt = the last parameter)
        import scipy.sparse as sp
        J = sp.coo_matrix((np.ones(len(rows)), (rows, cols)), shape=(dims.m, n)).tocsr()[:, : n - 1]
        H = (J.T @ J).tocoo()
        bw = int(np.abs(H.row - H.col).max())
        flops = float(n - 1) * (bw * bw + 3 * bw + 1) + 4.0 * (n - 1) * bw   # banded Cholesky + the two triangular solves (multiply-adds counted as 2)
        out[f"batch{B}"] = {"ms_per_solve": ms, "passes": int(st["passes"]), "factorizations": int(st["factorizations"]),
                            "band_assemble_plus_factor_ms_per_launch": f_ms, "parameters": n, "half_bandwidth": bw,
                            "us_per_pivot": 1e3 * f_ms / (n - 1), "cycles_per_pivot_at_2p4GHz": 2.4e6 * f_ms / (n - 1),
                            "flops_per_factorization_per_instance": flops,
                            "achieved_GFLOPs": B * flops / (f_ms * 1e-3) / 1e9, "frac_of_fp64_vector_peak": B * flops / (f_ms * 1e-3) / 1e12 / PEAK_F64_MFMA_TFLOPS}
        out[f"batch{B}"]["chi2_sum"] = chi2_band
        del s
    # the headline batch with the control-deviation term (a rate limit on the controls): the block-tridiagonal route (round 6: lm_bt_kernel, run to completion,
    # one launch per solve) and -- the A/B -- the band route it replaced (band_narrow_kernel per host-launched LM pass; corbo_hip_create_routed)
    try:
        from control_box_rst_amd import capi
        def rate_limit_solver(B, route):
            wl = workload(3, B)
            dx = wl["desc"]
            dx.ctrl_dev = capi.CTRL_DEV_RATE
            dx.ctrl_dev_params[0] = 1.0; dx.ctrl_dev_params[1] = 1.0
            s = BatchedLevenbergMarquardt(dx, B, device=device, route=route)
            s.setIterations(10); s.setPenaltyWeights(*wl["weights"])
            s.set_instance_data(s.init_trajectory(wl["x0"], wl["xf"]), xref=wl["xf"])
            s.solve(new_run=True); s.synchronize()
            return s
        def timed(s, reps=3):
            t0 = time.perf_counter()
            for _ in range(reps):
                s.restore_instance_data(); s.solve(new_run=True)
            s.synchronize()
            return (time.perf_counter() - t0) / reps * 1e3
        s = rate_limit_solver(1024, 0)
        ms = timed(s, 5)
        st = s.get_stats()
        row = {"ms_per_solve": ms, "factorizations": int(st["factorizations"]), "passes": int(st["passes"]), "chi2_sum": float(np.sum(s.get_solution()[1])),
               "kernel_ms_hip_events": float(st["solve_ms"]),
               "what": "configs[2]'s 1024 unicycle OCPs + the control-deviation term: ONE launch of lm_bt_kernel per solve (sweep phase with one lane per extra edge, H = J^T J in (x_k, u_k) "
                       "blocks, block cyclic reduction; DESIGN.md 3.5d).  1024 instances = 768 resident workgroups (3 per CU: 52.6 KB of LDS each) + a second round of 256: the time per "
                       "instance at 768 / 1536 / 2048 is what the route sustains (batch_sweep)"}
        # HBM roofline of the solve on SURVEY 8d's algorithmic bytes (the same convention as the headline's `roofline`)
        dims = s.dims
        b_sweep = 8 * (dims.nv + 2 * dims.n + dims.m + dims.nnz); b_val = 8 * (dims.nv + 2 * dims.n + dims.m)
        alg = st["jacobian_sweeps"] * b_sweep + max(0, st["residual_sweeps"] - st["jacobian_sweeps"]) * b_val
        row["roofline"] = {"bound": "hbm", "bytes_per_launch": alg, "ms_per_launch": float(st["solve_ms"]), "achieved": alg / (st["solve_ms"] * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                           "frac": alg / (st["solve_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, "note": "a latency chain per instance like the headline's kernel; 22 factorisations per instance, 54 % of them rejected steps"}
        del s
        sb = rate_limit_solver(1024, capi.ROUTE_XE_BAND)
        row["band_route_ms_per_solve"] = timed(sb, 2)
        row["band_route_chi2_sum"] = float(np.sum(sb.get_solution()[1]))
        del sb
        sweep = {}
        for B in (1, 64, 768, 2048):
            s = rate_limit_solver(B, 0)
            sweep[str(B)] = {"ms_per_solve": timed(s, 3), "us_per_instance": None}
            sweep[str(B)]["us_per_instance"] = 1e3 * sweep[str(B)]["ms_per_solve"] / B
            del s
        row["batch_sweep"] = sweep
        out["headline_batch_with_rate_limit"] = row
    except Exception as e:   # (a measurement row: never fatal for the line)
        out["headline_batch_with_rate_limit"] = {"error": str(e)[:200]}
    out["workload"] = "time-optimal quadrotor nx=12 nu=4, MultipleShootingVariableGrid N=100, RK4, MinimumTime, x_f fixed: band_assemble_kernel + band_factor_kernel per LM pass (batch1 / batch64: corbo_hip_create_routed with CORBO_HIP_ROUTE_FREE_DT_BAND); chain_route_*: the same solves through big_stage_kernel / big_chain3_kernel with the border column"
    out["bound"] = "latency: n sequential pivots per instance (one barrier each, eight waves on a sliding LDS window); the flop rate is quoted for completeness"
    return out


def sweep_phase_leg(solver, B, b_sweep, b_val, launch_ms):
    """The sweep PHASE of the kernel the timed region runs: per-instance phase totals accumulated by lm_pass_kernel itself
    (corbo_hip_get_phase_cycles; one extra solve with the option on, outside the timed region).  `achieved` = algorithmic bytes of every Jacobian sweep of the
    solve / the time those sweep phases would take if all resident workgroups ran nothing else: sum of the phases' cycles / resident workgroups / shader clock
    -- the in-kernel counterpart of `roofline_sweep` (the stand-alone kernel).  The shader clock is taken from the kernel itself: the slowest instance's phase
    cycles add up to (almost) the launch's duration."""
    solver.set_option("phase_cycles", 1)
    solver.solve(rearm=True)
    ms = solver.get_stats()["solve_ms"]
    pc = solver.get_phase_cycles().astype(np.float64)
    solver.set_option("phase_cycles", 0)
    tot = pc[:, 0:3].sum(axis=1)
    clk_ghz = tot.max() / (ms * 1e6)                      # cycles of the slowest instance / kernel time [ns] (lower bound of the clock: launch ramp excluded)
    n_j, n_r, n_f = pc[:, 3].sum(), pc[:, 4].sum(), pc[:, 5].sum()
    cyc_j, cyc_r, cyc_f = pc[:, 0].sum(), pc[:, 1].sum(), pc[:, 2].sum()
    t_j = cyc_j / B / (clk_ghz * 1e9)                     # seconds: the Jacobian sweep phases of one solve, all B workgroups side by side
    t_r = cyc_r / B / (clk_ghz * 1e9)
    ach_j = n_j * b_sweep / t_j / 1e9
    ach_all = (n_j * b_sweep + n_r * b_val) / (t_j + t_r) / 1e9
    return {"kind": "time share, NOT an HBM bandwidth measurement", "kernel": "lm_pass_kernel, sweep phases only (residual + Jacobian of accepted steps and of the prologue)",
            "time_share_equivalent_GBs": ach_j, "peak": PEAK_HBM_GBS, "unit": "GB/s", "time_share_equivalent": ach_j / PEAK_HBM_GBS,
            "time_share_equivalent_all_sweeps_GBs": ach_all, "time_share_equivalent_all_sweeps": ach_all / PEAK_HBM_GBS,
            "bytes_per_jacobian_sweep": b_sweep, "jacobian_sweeps": int(n_j), "residual_sweeps": int(n_r), "factor_phases": int(n_f),
            "mean_cycles": {"jacobian_sweep_phase": cyc_j / max(1, n_j), "residual_sweep_phase": cyc_r / max(1, n_r), "factor_phase": cyc_f / max(1, n_f)},
            "share_of_workgroup_time": {"jacobian_sweeps": cyc_j / tot.sum(), "residual_sweeps": cyc_r / tot.sum(), "factor": cyc_f / tot.sum()},
            "slowest_instance_cycles": float(tot.max()), "mean_instance_cycles": float(tot.mean()), "shader_clock_ghz_estimate": clk_ghz, "kernel_ms": ms,
            "note": "NOT HBM traffic and not a roofline fraction: algorithmic bytes of SURVEY 8d divided by (sum of the sweep phases' cycles / resident workgroups / clock), i.e. what the "
                    "rate WOULD be if all resident workgroups ran nothing but their sweep phases side by side.  It restates the sweep phases' share of workgroup time (share_of_workgroup_time) "
                    "in the roofline's unit; inside the fused kernel most of these bytes never leave the chip (the Jacobian goes from the sweep phase to the factor phase through LDS).  The "
                    "measured bandwidth figures are `roofline` (whole launch, `traffic` from the counters) and `roofline_sweep` (the stand-alone sweep_kernel)"}


def dense_weights_leg(device):
    """Non-diagonal Q / R / Qf (QuadraticFormCost::setWeightQ with a full matrix: dense cost blocks, quadratic_cost.cpp:36-55) on the headline structure: since
    round 6 such handles run to completion in one launch (lm_pass_kernel<.., DENSE>); `separate_launches_ms` = the same solve with the phases of every LM pass as
    two launches (corbo_hip_set_profiling: what these handles did until round 5)."""
    from control_box_rst_amd.solver import BatchedLevenbergMarquardt
    B = 1024
    w = workload(3, B)
    d = w["desc"]
    Q = np.array([[1.0, 0.2, 0.0], [0.2, 1.0, 0.05], [0.0, 0.05, 0.1]]); R = np.array([[0.1, 0.02], [0.02, 0.05]])
    for name, M in (("q_sqrt", Q), ("r_sqrt", R), ("qf_sqrt", 10.0 * Q)):
        U = np.linalg.cholesky(M).T
        n = U.shape[0]
        arr = getattr(d, name)
        for i in range(n):
            for j in range(n):
                arr[i * n + j] = U[i, j]
    d.weights_dense = 7
    out = {"workload": "configs[2]'s structure, batch 1024, non-diagonal Q, R, Qf (upper Cholesky factors), 10 LM iterations"}
    for tag, prof in (("ms_per_solve", False), ("separate_launches_ms", True)):
        s = BatchedLevenbergMarquardt(d, B, device=device)
        s.setIterations(10); s.setPenaltyWeights(*w["weights"])
        s.set_instance_data(s.init_trajectory(w["x0"], w["xf"]), xref=w["xf"])
        s.set_profiling(prof)
        s.solve(new_run=True); s.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            s.restore_instance_data(); s.solve(new_run=True)
        s.synchronize()
        out[tag] = (time.perf_counter() - t0) / 3 * 1e3
        st = s.get_stats()
        out["factorizations"] = int(st["factorizations"]); out["passes"] = int(st["passes"])
        out["chi2_sum" if not prof else "chi2_sum_separate_launches"] = float(np.sum(s.get_solution()[1]))
        del s
    return out


def long_horizon_leg(device):
    """Horizons beyond 256 grid points (FiniteDifferencesVariableGrid's default n_max is 1000): the factor workspace lives in HBM, one 1024-thread workgroup per instance,
    every LM pass is two launches (sweep_kernel<.., LONG>, factor_long_kernel) -- NOT fused into a run-to-completion launch yet (measured here so that
    the cost is on record).  Unicycle, N = 512, batch 1024, 10 LM iterations; per pass per stage next to the headline's fused kernel."""
    from control_box_rst_amd import problems
    from control_box_rst_amd.solver import BatchedLevenbergMarquardt
    B, N = 1024, 512
    d = problems.unicycle_desc(N=N, dt=0.02)
    x0, xf = problems.unicycle_instances(B)
    s = BatchedLevenbergMarquardt(d, B, device=device)
    s.setIterations(10); s.setPenaltyWeights(*problems.UNICYCLE_WEIGHTS)
    s.set_instance_data(s.init_trajectory(x0, xf), xref=xf)
    s.solve(new_run=True); s.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        s.restore_instance_data(); s.solve(new_run=True)
    s.synchronize()
    ms = (time.perf_counter() - t0) / 2 * 1e3
    st = s.get_stats()
    out = {"workload": f"unicycle N={N}, batch {B}, 10 LM iterations (per-pass launches, workspace in HBM)", "ms_per_solve": ms, "passes": int(st["passes"]),
           "factorizations": int(st["factorizations"]), "us_per_pass_per_1000_stages": 1e3 * ms / max(1, int(st["passes"])) / (N / 1000.0),
           "chi2_sum": float(np.sum(s.get_solution()[1]))}
    del s
    return out


def hessian_leg(desc, B, x0, xf, device):
    """Operators of the exact-Hessian path on the headline structure: corbo_hip_eval_hessians (lower part, values left in HBM) for the
    bench batch and for ONE OCP (what the drop-in adapter's Hessian-path entry points run).  Wall time per call around a device-resident call +
    synchronize; algorithmic bytes = read vertices + equality multipliers, write the three value lists."""
    from control_box_rst_amd.solver import BatchedLevenbergMarquardt
    out = {"bound": "hbm", "kernel": "hessian_kernel (mode 0: Hessian values of every stage's edges, forward differences of central-difference Jacobians)",
           "peak": PEAK_HBM_GBS, "unit": "GB/s"}
    prof = profile_kernel_avg_ns(PROFILE_ROUND + "_hessian_kernel_stats.csv", "hessian_kernel")
    for tag, b in (("batch", B), ("single", 1)):
        s = BatchedLevenbergMarquardt(desc, b, device=device)
        rng = np.random.default_rng(0)
        X = s.init_trajectory(x0[:b], xf[:b]) + 0.02 * rng.normal(size=(b, s.dims.nv))
        X[:, : desc.nx] = x0[:b]
        s.set_instance_data(X, xref=xf[:b])
        me = rng.uniform(0.2, 1.0, (b, s.dims.eq))
        st = s.hessian_structure(True)
        nnz = [len(st[c][0]) for c in range(3)]
        s.eval_hessians(True, 1.0, me, None)   # multipliers resident from here on
        n = 30 if b > 1 else 200
        s.eval_hessians_views(True, 1.0, None, None, device=True)
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            s.eval_hessians_views(True, 1.0, None, None, device=True)
        s.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        alg = 8 * b * (s.dims.nv + s.dims.eq + sum(nnz))
        out[tag] = {"instances": b, "nnz_obj_eq_ineq": nnz, "ms_per_call": ms, "bytes_per_call": alg, "achieved": alg / (ms * 1e-3) / 1e9,
                    "frac": alg / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS}
        del s
    out["achieved"], out["frac"] = out["batch"]["achieved"], out["batch"]["frac"]
    out["profile_avg_ms"] = prof[0] * 1e-6 if prof else None
    out["profile"] = prof[2] if prof else None
    out["timing"] = "wall clock around device-resident calls (values left in HBM), synchronize on both sides"
    out["note"] = ("fp64-compute-bound, not a bandwidth kernel: ~100 central-difference edge evaluations per stage (2 dynamics evaluations each); the fraction is quoted "
                   "for completeness.  One OCP: the stage's blocks are spread over ~15 waves per 64 stages (HessParams::split = 2), a batch that fills the chip walks them per lane")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=3, choices=(1, 2, 3, 5))
    ap.add_argument("--batch", type=int, default=None, help="OCP instances per GPU")
    ap.add_argument("--iterations", type=int, default=10, help="LM outer iterations per solve")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sink", action="store_true", help="results via D2H copies after the solve instead of the kernel-written pinned sink")
    ap.add_argument("--preheat-ms", type=float, default=300.0, help="untimed steps for this long before the warm-up steps (clock ramp, cold pages); 0 = none")
    ap.add_argument("--sync-steps", action="store_true", help="the host waits for every solve of the timed region (rounds 1-3) instead of enqueueing the steps back to back")
    ap.add_argument("--no-secondary", action="store_true", help="skip the legs over BASELINE configs 1, 2, 5 in the default (config 3, 1 GPU) line")
    ap.add_argument("--solve-only", action="store_true", help="profiling: only warm-up + timed steps (no roofline / host legs)")
    args = ap.parse_args()
    cfg = args.config
    batch = args.batch if args.batch is not None else DEFAULT_BATCH[cfg]
    # (batch-1 configurations: a launch of ~0.1 ms does not keep the GPU out of its low-power clocks by itself -- the first few hundred steps of a
    #  cold process run at 0.26 ms, the steady state at 0.15 ms; their defaults warm up for ~0.15 s)
    steps = args.steps if args.steps is not None else {1: 2000, 2: 400, 3: 100, 5: 20}[cfg]
    warmup = args.warmup if args.warmup is not None else {1: 1000, 2: 200, 3: 10, 5: 3}[cfg]

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU fallback)")
    # tests/test_gpu_bench_ranks.py runs the N > 1 path on a 1-GPU box: every rank on GPU 0, gloo instead of RCCL (RCCL refuses two
    # ranks on one device).  Never set by the driver: one rank per GPU over RCCL is the product configuration.
    shared_gpu = os.environ.get("CORBO_BENCH_TEST_SHARED_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    red_dev = "cpu" if shared_gpu else "cuda"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    if dist is not None:
        dist.barrier()
    from control_box_rst_amd import sharding
    from control_box_rst_amd.solver import BatchedLevenbergMarquardt

    first, B = sharding.shard_bounds(batch * world, world, rank)  # weak scaling: `batch` instances per GPU
    w = workload(cfg, B, first=first)                             # rank r owns global instances [first, first+B)
    desc, x0, xf, solves = w["desc"], w["x0"], w["xf"], w["solves"]
    solver = BatchedLevenbergMarquardt(desc, B, device=local_rank)
    solver.setIterations(args.iterations)
    solver.setPenaltyWeights(*w["weights"])
    X0 = solver.init_trajectory(x0, xf)
    solver.set_instance_data(X0, xref=xf)  # H2D once; everything below runs on HBM-resident data

    use_sink = not args.no_sink
    solver.set_result_sink(use_sink)   # the solve kernel writes each finished instance's results into pinned host memory itself

    def step(fetch=True):
        solver.restore_instance_data()
        for i in range(solves):
            solver.solve(new_run=(i == 0))
        if fetch:
            return solver.fetch_solution()   # trajectories, chi2, status -> pinned host memory (inside the timed region)

    # The timed steps are ENQUEUED back to back (corbo_hip_solve_async: the solve without the wait): the host side of step k + 1 -- re-arm, launch --
    # overlaps the kernel of step k, as a caller that streams batch after batch through a handle would run it.  Every step does the same work as the
    # synchronous one (re-arm + every LM pass + the results written into pinned host memory by the kernel); the closing fence waits for all of them.
    # Handles whose passes are driven from the host (cfg 5) solve synchronously inside solve_async and deliver each step's results with a copy on a
    # second stream (device-side snapshot first), which overlaps the next step's kernels.  --sync-steps: the per-step wait of rounds 1-3.
    pipelined = use_sink and not args.sync_steps

    def step_timed():
        if not pipelined:
            return step()
        for i in range(solves):
            solver.solve_async(new_run=(i == 0), rearm=(i == 0))   # (rearm: the re-arm of the step inside the solve -- new_run = 2 of the C-ABI)

    def fence():
        solver.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # Steady state before anything is counted: a fresh process on a fresh box runs its first tens of milliseconds of GPU work at ramping
    # clocks and with cold page tables / pinned pages (measured: the same 20 timed steps 0.74 - 0.77 ms right after start, 0.68 ms after
    # 0.3 s of work).  `preheat_ms` of UNTIMED steps come first -- then the W warm-up steps and the K timed steps the contract asks for.
    preheat_steps = 0
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.preheat_ms:
        step()
        preheat_steps += 1
    for _ in range(warmup):
        step()
    fence()
    solver.get_timing(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step_timed()
    fence()
    if pipelined:
        solver.fetch_solution()                   # (views of the pinned result buffers the last step's kernel filled)
    elapsed = time.perf_counter() - t0
    solve_ms_sum, n_solves = solver.get_timing(reset=True)
    # the same K steps with the host waiting for every solve (how rounds 1-3 timed the step): reported next to `value`, never in its place
    sync_ms_per_step = None
    if pipelined:
        fence()
        t1 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        sync_ms_per_step = 1e3 * sharding.reduce_max(time.perf_counter() - t1, dist, device=red_dev) / steps
        solver.get_timing(reset=True)

    stats = solver.get_stats()                    # statistics of the LAST solve of a step
    X, chi2, status = solver.get_solution()
    t_max = sharding.reduce_max(elapsed, dist, device=red_dev)                       # MAX over ranks (RCCL)
    t_ranks = sharding.gather_scalars(elapsed, dist, device=red_dev)                 # every rank's own time (straggler visibility)
    red = sharding.reduce_stats(stats, float(chi2.sum()), int((status <= 1).sum()), dist, device=red_dev)  # SUM over ranks
    total_iters_per_step = red["lm_iterations"] * solves  # = world * batch * iterations * solves
    value = total_iters_per_step * steps / t_max
    counted_step = sharding.reduce_stats({k: (counted_per_step(solver, solves) if k == "counted_iterations" else 0) for k in sharding.STAT_KEYS},
                                         0.0, 0, dist, device=red_dev)["counted_iterations"]   # per step, all ranks (untimed extra step)

    line = {
        "metric": "SQP-iterations/sec over batch=1024 OCPs (nx=3,nu=2,N=100,fp64)",
        "value": value,
        "unit": "SQP-iterations/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": 1e3 * t_max / steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"{w['name']}, batch={B} per GPU, {solves} solve(s) x {args.iterations} LM iterations per step, seeds 20260928+i",
                   "batch_per_gpu": B, "global_batch": B * world, "iterations": args.iterations, "solves_per_step": solves,
                   "parallelism": f"batch-sharded x{world}"},
        "timed_region": ("corbo_hip_solve_async(new_run = 2: re-arm from the uploaded start, done by the solve kernel itself; handles with host-launched passes copy first) per step, the K steps enqueued back to back (the host side of a step overlaps the previous step's kernel), "
                         "every step's trajectories/chi2/status written into pinned host memory (corbo_hip_set_result_sink): by the solve kernel as each instance finishes, or -- "
                         "handles whose passes are launched from the host, cfg 5 -- by a copy on a second stream behind a device-side snapshot, overlapping the next step; "
                         "wall clock, barrier + synchronize on both sides, MAX over ranks" if pipelined else
                         "re-arm (D2D) + corbo_hip_solve + trajectories/chi2/status resident in pinned host memory ("
                         + ("written by the solve kernel as each instance finishes, corbo_hip_set_result_sink" if use_sink else "two D2H copies behind the solve")
                         + ", views from corbo_hip_fetch_solution); wall clock, barrier + synchronize on both sides, MAX over ranks"),
        "synchronous_steps": ({"ms_per_step": sync_ms_per_step, "value": total_iters_per_step / (sync_ms_per_step * 1e-3),
                               "what": "the same K steps with the host waiting for every solve before it re-arms the next one (corbo_hip_solve; the timed region of rounds 1-3)"}
                              if sync_ms_per_step else None),
        "preheat": {"ms": args.preheat_ms, "untimed_steps": preheat_steps, "what": "untimed steps before the warm-up steps: steady-state clocks and warm pages (a cold process measures 0.74 - 0.77 ms per step over its first 25 steps, 0.68 ms afterwards)"},
        "counted_iterations": int(counted_step),
        "value_computed": (total_iters_per_step - counted_step) * steps / t_max,
        "counted_note": "outer iterations after a converged step (|delta| <= eps2/2) are counted, not executed (corbo_hip_stats.counted_iterations); value_computed = executed iterations / s",
        "batch_steps_per_s": value / (B * world),
        "ms_per_step_ranks": [1e3 * t / steps for t in t_ranks],
        "solve_stats": {"passes_rank0": stats["passes"], "lm_iterations": int(red["lm_iterations"]),
                        "accepted": int(red["accepted_steps"]), "rejected": int(red["rejected_steps"]),
                        "factorizations": int(red["factorizations"]), "counted_iterations": int(red["counted_iterations"]), "chi2_sum": red["chi2_sum"],
                        "ok_instances": int(red["ok_instances"])},
    }
    # ---- outside the timed region: the trajectories of ALL ranks collected straight from device memory (one RCCL all-gather on the
    #      handles' own HBM buffers, no host round trip) -- checked against the per-rank results that went through the host
    fence()
    t_g0 = time.perf_counter()
    allx = sharding.gather_trajectories_device(solver, B * world, dist)   # first call: creates the communicator / the torch view (one-off)
    torch.cuda.synchronize()
    t_g0 = time.perf_counter() - t_g0
    fence()
    t_g = time.perf_counter()
    allx = sharding.gather_trajectories_device(solver, B * world, dist)   # steady state: the collective itself
    torch.cuda.synchronize()
    t_g = time.perf_counter() - t_g
    mine = allx[first:first + B].cpu().numpy()
    line["gather"] = {"what": "all ranks' final trajectories, torch.distributed all_gather_into_tensor on views of the handles' HBM buffers",
                      "ms": 1e3 * t_g, "first_call_ms": 1e3 * t_g0, "first_call_what": "includes the one-off RCCL communicator set-up / lazy initialisation; `ms` is a second, steady-state call", "bytes": int(allx.numel() * 8), "matches_host_copy": bool(np.array_equal(mine, X))}
    if args.solve_only:
        if rank == 0:
            print(json.dumps(line), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- the same steps with the results left in HBM (what a device-resident caller -- the closed loop, an RCCL gather -- sees)
    solver.set_result_sink(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(fetch=False)
    fence()
    t_res = sharding.reduce_max(time.perf_counter() - t0, dist, device=red_dev)
    line["results_left_in_hbm"] = {"ms_per_step": 1e3 * t_res / steps, "value": total_iters_per_step * steps / t_res}

    # ---- roofline of the kernel that dominates the timed region
    dims = solver.dims
    b_sweep = 8 * (dims.nv + 2 * dims.n + dims.m + dims.nnz)  # SURVEY 8d algorithmic bytes per instance per residual + Jacobian sweep
    b_val = 8 * (dims.nv + 2 * dims.n + dims.m)               # residual-only (trial step) sweep
    sweeps_j, sweeps_r = stats["jacobian_sweeps"], stats["residual_sweeps"]   # this rank, one solve
    alg_solve = sweeps_j * b_sweep + max(0, sweeps_r - sweeps_j) * b_val
    launch_ms = solve_ms_sum / max(1, n_solves)               # HIP events on the handle's stream, every solve of the timed region
    if cfg != 5:
        # run-to-completion families: ONE launch per solve (lm_pass_kernel); its bytes = what the reference's algorithm moves for the
        # same sweeps: sweeps_j x (read vertices + bounds, write residual + Jacobian) + trial sweeps x (read vertices + bounds, write residual)
        achieved = alg_solve / (launch_ms * 1e-3) / 1e9
        pmc = load_profile_json(PROFILE_ROUND + "_solve_pmc.json", B, desc.N)
        prof = profile_kernel_avg_ns(PROFILE_ROUND + "_bench_kernel_stats.csv", "lm_pass_kernel")
        line["roofline"] = {"bound": "hbm", "kernel": "lm_pass_kernel (run-to-completion: prologue sweep + every LM pass of every instance, one launch per solve)",
                            "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS,
                            "traffic": pmc["hbm_bytes_per_launch"] if pmc else None, "traffic_source": pmc["source"] if pmc else None,
                            "bytes_per_launch": alg_solve, "ms_per_launch": launch_ms, "launches_timed": n_solves,
                            "jacobian_sweeps": int(sweeps_j), "residual_sweeps": int(sweeps_r),
                            "timing": "HIP events on the handle's stream around every launch of the timed region (corbo_hip_get_timing)",
                            "profile_avg_ms": prof[0] * 1e-6 if prof else None, "profile": prof[2] if prof else None,
                            "frac_from_profile": (alg_solve / (prof[0] * 1e-9) / 1e9 / PEAK_HBM_GBS) if (prof and B == 1024 and cfg == 3) else None,
                            "note": "a latency chain per instance (DESIGN.md 3.3), priced against HBM because its algorithmic work is the sweep traffic"}
        try:
            line["sweep_phase_time_share"] = sweep_phase_leg(solver, B, b_sweep, b_val, launch_ms)
        except Exception as e:
            line["sweep_phase_time_share"] = {"error": repr(e)}
    # ---- the stand-alone edge/Jacobian sweep (north star: ">= 40 % of the HBM roofline on the Jacobian sweep")
    each = solver.time_sweep_each(with_jacobian=True, repeat=50)        # one event pair per launch (what a kernel trace reports)
    b2b_ms = solver.time_sweep(with_jacobian=True, repeat=50)            # back-to-back launches, one event pair around all of them
    sweep_ms = float(np.mean(each))
    prof = profile_kernel_avg_ns(PROFILE_ROUND + "_sweep_kernel_stats.csv", "sweep_kernel")
    spmc = load_profile_json("sweep_pmc_latest.json", B, desc.N)
    rs = {"bound": "hbm", "kernel": "sweep_kernel (residual + Jacobian of every instance, stand-alone launch)",
          "achieved": B * b_sweep / (sweep_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
          "frac": B * b_sweep / (sweep_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
          "traffic": spmc["hbm_bytes_per_launch"] if spmc else None, "traffic_source": spmc["source"] if spmc else None,
          "bytes_per_launch": B * b_sweep, "ms_per_launch": sweep_ms, "ms_per_launch_min": float(each.min()), "ms_per_launch_max": float(each.max()),
          "timing": "50 launches, each bracketed by its own HIP event pair on the handle's stream (corbo_hip_time_sweep_each)",
          "back_to_back": {"ms_per_launch": b2b_ms, "frac": B * b_sweep / (b2b_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                           "what": "50 launches between ONE event pair: the next launch's ramp-up overlaps the previous one's store drain"},
          "profile_avg_ms": prof[0] * 1e-6 if prof else None, "profile": prof[2] if prof else None}
    line["roofline_sweep"] = rs
    if cfg == 5:
        # big-block family: an LM pass is sweep (residual) -> big_stage_kernel -> big_chain3_kernel (N >= 64; shorter horizons: big_chain2_kernel); the last two are 80 % of the GPU time
        # of a solve (profiles/r03_cfg5_kernel_stats.csv).  One factorisation of EVERY instance (stage + chain launch), HIP events
        # around 5 back-to-back pairs (corbo_hip_time_factor).  Algorithmic bytes per shooting interval: stage kernel
        # reads x_k u_k, the stored RK4 end state, the bounds (60 doubles) and writes the 572-double stage record; the chain kernel
        # reads the record, writes G_k and a_k (156), reads them back in the back-substitution and moves the iterate (32): 1548 doubles
        rs["note"] = "parity hook of this family (residual + Jacobian in HBM); its LM passes launch the residual-only sweep and the stage kernel instead"
        nxq, nuq = desc.nx, desc.nu
        rec = 3 * nxq * nxq + nuq * nuq + 2 * nuq * nxq + 2 * nxq + nuq + 2
        per_stage = 8 * ((nxq + nuq) + nxq + 2 * (nxq + nuq) + rec + rec + 2 * (nxq * nxq + nxq) + 2 * (nxq + nuq))
        f_ms = solver.time_factor(repeat=5)
        model_pair = per_stage * desc.N * B    # the kernels' OWN reads + writes (stage records handed from the stage kernel to the chain): `traffic_model`
        dq = solver.dims
        alg_pair = 8 * (dq.nv + 2 * dq.n + dq.m + dq.nnz) * B   # the algorithmic bytes of one Jacobian sweep -- what `frac` is priced on
        # (the longest launches of a solve WITH the reject-streak speculation carry 32 candidate instances more than the batch -- a third round of the chain
        # kernel; the cross-check of a full launch over exactly `batch` instances reads the trace taken without it: tools/profile_cfg5.py .. nospec)
        pc = (profile_kernel_max_ns(PROFILE_ROUND + "_cfg5_nospec_kernel_stats.csv", ("big_stage_kernel", "big_chain3_kernel" if desc.N >= 64 else "big_chain2_kernel"))
              or profile_kernel_max_ns(PROFILE_ROUND + "_cfg5_kernel_stats.csv", ("big_stage_kernel", "big_chain3_kernel" if desc.N >= 64 else "big_chain2_kernel")))
        qpmc = load_profile_json(PROFILE_ROUND + "_cfg5_pmc.json", B, desc.N)
        line["roofline"] = {"bound": "hbm", "kernel": "big_stage_kernel + big_chain3_kernel (one factorisation of every instance: FD Jacobian + assemble, then the partitioned block chain)",
                            "achieved": alg_pair / (f_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": alg_pair / (f_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                            "traffic": qpmc["hbm_bytes_per_launch"] if qpmc else None, "traffic_source": qpmc["source"] if qpmc else None,
                            "bytes_per_launch": alg_pair, "bytes_per_instance": alg_pair // B, "bytes_definition": "SURVEY 8d: 8 (n_vert + 2 n + m + nnz) per instance per Jacobian sweep",
                            "traffic_model": model_pair, "traffic_model_bytes_per_interval": per_stage, "ms_per_launch": f_ms,
                            "timing": "HIP events around 5 back-to-back (stage, chain) launch pairs over all instances (corbo_hip_time_factor)",
                            "profile_full_launch_ms": pc[0] * 1e-6 if pc else None, "profile": pc[1] if pc else None,
                            "note": "the stage kernel is fp64-VALU-bound (RK4 finite differences), the chain a latency chain of 12 x 12 pivots; matrix-core view in `factorization`"}
    # per-kernel split inside one solve (separate, profiled solve: event stamping is kept out of the timed region)
    solver.set_profiling(True)
    step(fetch=False)
    prof_stats = solver.get_stats()
    solver.set_profiling(False)
    line["kernel_split_ms"] = {"solve": prof_stats["solve_ms"], "sweep": prof_stats["sweep_ms"], "factor": prof_stats["factor_ms"],
                               "what": "one solve with the phases as separate launches (per-pass mode), HIP events between them"}
    if cfg == 5:
        # block factorisation on the fp64 matrix cores: SURVEY 8d algorithmic flops per instance per factorisation
        s_blk, nb = desc.nx + desc.nu, desc.N - 1
        flops_fact = (7.0 / 3.0) * s_blk ** 3 * nb + 8.0 * s_blk ** 2 * nb + 2.0 * desc.nx * (2 * desc.nx + desc.nu) ** 2 * nb
        n_fact = stats["factorizations"] - stats["counted_iterations"]   # executed factorisations only
        mf = load_profile_json(PROFILE_ROUND + "_cfg5_mfma.json", B, desc.N)
        line["factorization"] = {"bound": "mfma", "algorithmic_flops_per_instance": flops_fact, "factorizations_per_solve": int(n_fact),
                                 "factor_ms_per_solve": prof_stats["factor_ms"],
                                 "achieved_TFLOPs": (n_fact * flops_fact / (prof_stats["factor_ms"] * 1e-3) / 1e12) if prof_stats["factor_ms"] > 0 else None,
                                 "peak_TFLOPs": PEAK_F64_MFMA_TFLOPS, "mfma_counters": mf}
    # the boundary takes HOST buffers: one-shot rate including the upload of trajectories / references over PCIe, the solve(s)
    # and the download of trajectories, chi2 and status into the caller's arrays (never `value`)
    fence()
    n_h = 5 if B >= 64 else 50
    t_h = time.perf_counter()
    for _ in range(n_h):
        solver.set_instance_data(X0, xref=xf)
        for i in range(solves):
            solver.solve(new_run=(i == 0))
        solver.get_solution()
    host_ms = (time.perf_counter() - t_h) / n_h * 1e3
    line["host_inclusive"] = {"ms_per_step": host_ms, "value_rank0": B * args.iterations * solves / (host_ms * 1e-3),
                              "what": "set_instance_data (H2D of x, xref from pageable host memory) + solve(s) + get_solution (D2H into caller arrays), rank 0"}
    lm_opts = solver.opts
    if rank == 0 and world == 1 and cfg == 3 and not args.no_secondary:
        try:
            line["roofline_hessian"] = hessian_leg(desc, B, x0, xf, local_rank)
        except Exception as e:
            line["roofline_hessian"] = {"error": repr(e)}
        # every other BASELINE configuration, driver-visible in the same line (a few hundred ms of GPU time each)
        line["secondary"] = {}
        for c2 in (1, 2, 5):
            try:
                line["secondary"][f"config{c2}"] = secondary_leg(c2, args.iterations, local_rank)
            except Exception as e:   # a failing leg must not take the headline line with it
                line["secondary"][f"config{c2}"] = {"error": repr(e)}
        try:
            line["secondary"]["band_path"] = band_leg(local_rank)
        except Exception as e:
            line["secondary"]["band_path"] = {"error": repr(e)}
        try:
            line["secondary"]["long_horizon"] = long_horizon_leg(local_rank)
        except Exception as e:
            line["secondary"]["long_horizon"] = {"error": repr(e)}
        try:
            line["secondary"]["dense_weights"] = dense_weights_leg(local_rank)
        except Exception as e:
            line["secondary"]["dense_weights"] = {"error": repr(e)}
        try:
            lm_opts = solver.opts
            del solver
            line["secondary"]["batch8192"] = batch8192_leg(args.iterations, local_rank, value if B == 1024 else None)
        except Exception as e:
            line["secondary"]["batch8192"] = {"error": repr(e)}
    if rank == 0:
        if not args.no_cpu_baseline:
            cb = cpu_baseline(cfg, w, lm_opts)
            line["cpu_baseline"] = cb
            if "ms_per_ocp" in cb:
                line["cpu_baseline"]["gpu_ms_per_ocp_batch1_equivalent"] = host_ms if B == 1 else None
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
