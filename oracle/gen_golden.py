"""Generate tests/golden/*.json from the GENUINE reference -- TEST INFRASTRUCTURE ONLY.

Runs here (build container) only: needs oracle/_ref/ref_driver, which oracle/Makefile compiles from the reference's
own sources under /root/reference.  The fixtures are data (inputs + expected outputs); no reference source travels.

    make -C oracle ref && python oracle/gen_golden.py
"""
from __future__ import annotations

import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from control_box_rst_amd import problems  # noqa: E402

DRIVER = os.path.join(HERE, "_ref", "ref_driver")
OUT = os.path.join(ROOT, "tests", "golden")


def run(mode, **kv):
    args = [DRIVER, mode] + [f"{k}={v}" for k, v in kv.items()]
    return json.loads(subprocess.check_output(args))


def vec(v):
    return ",".join(repr(float(a)) for a in v)


def slim(d, keep_iters):
    d = dict(d)
    d["after_iter"] = [a for a in d["after_iter"] if a["k"] in keep_iters]
    return d


def final_of(kv):
    d = run("dump", **kv)
    last = d["after_iter"][-1]
    return np.array(last["vertex"]), last["chi2"]


def ulp_runs(scenario, x0, xf, n_perturb, **kv):
    """Reference result for (x0, xf) and for x0 with component k moved by ONE ulp, k < n_perturb: the reference's own reproducibility
    under a rounding-level change of its input (central differences with delta = 1e-9 amplify the last bits)."""
    base, chi2 = final_of(dict(scenario=scenario, x0=vec(x0), xf=vec(xf), **kv))
    pert = []
    for k in range(n_perturb):
        x1 = np.array(x0, dtype=float)
        x1[k] = np.nextafter(x1[k], 10.0)
        pert.append(final_of(dict(scenario=scenario, x0=vec(x1), xf=vec(xf), **kv))[0])
    return base, chi2, pert


def _uni_job(args):
    b, x0, xf = args
    base, chi2, pert = ulp_runs("unicycle", x0, xf, 3, iters=10)
    return b, base, chi2, max(float(np.abs(p - base)[3:].max()) for p in pert)


def fullsize():
    """Full-size parity fixtures: the reference's result for EVERY instance of the headline batch, and the
    reference's own one-ulp reproducibility per instance (binary .npz: 1024 x 498 doubles)."""
    from concurrent.futures import ProcessPoolExecutor
    B = 1024
    x0, xf = problems.unicycle_instances(B)
    vertex, chi2, spread = None, np.zeros(B), np.zeros(B)
    with ProcessPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        for b, base, c, s in ex.map(_uni_job, [(b, x0[b], xf[b]) for b in range(B)], chunksize=8):
            if vertex is None:
                vertex = np.zeros((B, len(base)))
            vertex[b], chi2[b], spread[b] = base, c, s
    np.savez_compressed(os.path.join(OUT, "unicycle_seeded1024.npz"), seed=20260928, iters=10, vertex=vertex, chi2=chi2, ulp_spread=spread)
    print("unicycle_seeded1024: spread median %.2e p99 %.2e max %.2e (instance %d), n(spread > 3e-6) = %d" % (
        np.median(spread), np.quantile(spread, 0.99), spread.max(), spread.argmax(), int((spread > 3e-6).sum())))
    # cfg 5 family: 6 seeded instances at N = 40 (base + 2 one-ulp perturbations each) and instance 0 at the full N = 200
    nq = 6
    x0, xf = problems.quad_instances(nq)
    inst = []
    for b in range(nq):
        base, c, pert = ulp_runs("quad", x0[b], xf[b], 2, N=40, iters=10)
        inst.append({"x0": list(x0[b]), "xf": list(xf[b]), "chi2": c, "vertex": list(base), "vertex_ulp": [list(p) for p in pert]})
        print("quad N=40 instance", b, "ulp spread", [float(np.abs(p - base)[12:].max()) for p in pert])
    with open(os.path.join(OUT, "quad_n40_seeded_ulp.json"), "w") as f:
        json.dump({"scenario": "quad", "N": 40, "dt": 0.05, "iters": 10, "seed": 20260928, "weights": [10, 10, 10], "instances": inst}, f, separators=(",", ":"))
    base, c, pert = ulp_runs("quad", x0[0], xf[0], 1, N=200, iters=10)
    print("quad N=200 instance 0 ulp spread", float(np.abs(pert[0] - base)[12:].max()))
    np.savez_compressed(os.path.join(OUT, "quad_n200_seeded_ulp.npz"), seed=20260928, iters=10, x0=x0[0], xf=xf[0], chi2=c, vertex=base, vertex_ulp=pert[0])


def adapt():
    """Time-optimal grid adaptation (SURVEY 8f rank 2 remainder): moving-horizon sequences on the FiniteDifferencesVariableGrid with
    TimeBasedSingleStep / TimeBasedAggressiveEstimate / SimpleShrinkingHorizon (finite_differences_variable_grid.cpp:101-163), K
    compute() calls per step like PredictiveController::step.  iters=0 after step 0: the dumps show the pure grid update (chains of
    resampleTrajectory, full_discretization_grid_base.cpp:397-474); iters=5: the controller as it runs."""
    base = dict(scenario="dint", steps=5, iters0=10, shift=0, ocp_iters=3)
    for name, kv in [
        ("mpc_dint_adapt_single_grow_init", dict(iters=0, adapt="single", nmax=80, hyst=0.02, dt=0.06)),
        ("mpc_dint_adapt_single_shrink_init", dict(iters=0, adapt="single", nmax=80, hyst=0.1)),
        ("mpc_dint_adapt_aggressive_init", dict(iters=0, adapt="aggressive", nmax=70, hyst=0.05, dt=0.09, adapt_first=1)),
        ("mpc_dint_adapt_shrink_init", dict(iters=0, adapt="shrink", nmin=44)),
        ("mpc_dint_adapt_single", dict(iters=5, adapt="single", nmax=80, hyst=0.1, steps=6)),
        ("mpc_dint_adapt_aggressive", dict(iters=5, adapt="aggressive", nmax=80, hyst=0.1, steps=4)),
        # grids that grow past 256 points (the LDS-resident kernels' limit: beyond it the long-horizon kernels take over): one jump, and point by point
        ("mpc_dint_adapt_cross256_aggressive", dict(iters0=6, iters=3, adapt="aggressive", N=250, dt=0.007, nmax=300, hyst=0.05, steps=3)),
        ("mpc_dint_adapt_cross256_single", dict(iters0=6, iters=3, adapt="single", N=255, dt=0.007, nmax=300, hyst=0.05, steps=2)),
        # the same on the MultipleShootingVariableGrid (multiple_shooting_variable_grid.cpp:58-152 -> ShootingGridBase::resampleTrajectory,
        # shooting_grid_base.cpp:473-547)
        ("mpc_dint_ms_adapt_single_init", dict(grid="ms", iters=0, adapt="single", nmax=80, hyst=0.02, dt=0.06)),
        ("mpc_dint_ms_adapt_shrink_init", dict(grid="ms", iters=0, adapt="shrink", nmin=44)),
        # (4 + 2 LM iterations per compute(): with 10 + 5 at penalty weight 100 the iterates sit next to the bang-bang solution, where the
        # restatement's own result moves by 1e-4 under a 1e-13 perturbation of its start -- nothing to pin a tolerance on)
        ("mpc_dint_ms_adapt_single", dict(grid="ms", iters0=4, iters=2, adapt="single", nmax=80, hyst=0.1, steps=6)),
        ("mpc_dint_ms_adapt_aggressive", dict(grid="ms", iters0=4, iters=2, adapt="aggressive", dt=0.03, N=30, nmax=70, nmin=20, hyst=0.1, steps=4)),
        # n * (int)round(dt / dt_ref) with dt < dt_ref / 2: the grid collapses to n_min (multiple_shooting_variable_grid.cpp:127-133)
        ("mpc_dint_ms_adapt_aggressive_collapse", dict(grid="ms", iters0=4, iters=2, adapt="aggressive", dt=0.2, N=60, nmax=90, nmin=12, hyst=0.1, steps=4)),
    ]:
        d = run("mpc", **{**base, **kv})
        with open(os.path.join(OUT, f"{name}.json"), "w") as f:
            json.dump(d, f, separators=(",", ":"))
        print(name, [st["n_seq"] for st in d["steps"]])


def hess():
    """Operators of the exact-Hessian path at a generic point: computeSparseHessians{Structure,Values} (full and
    lower part, seeded multipliers) and the two-side-bounded linear form with its bounds, from the genuine reference
    (ref_driver hess).  Small horizons: the operators are per edge, the structure repeats."""
    for name, kv in [
        ("hess_vdp", dict(scenario="vdp")),
        ("hess_vdp_forward", dict(scenario="vdp", collocation="forward", N=10)),
        ("hess_vdp_backward", dict(scenario="vdp", collocation="backward", N=10)),
        ("hess_vdp_midpoint", dict(scenario="vdp", collocation="midpoint", N=10)),
        ("hess_vdp_teq", dict(scenario="vdp", N=10, teq=1)),
        ("hess_dint", dict(scenario="dint", N=20)),
        ("hess_int3_time_optimal", dict(scenario="int3", vargrid=1, N=12)),
        ("hess_unicycle_n16", dict(scenario="unicycle", N=16)),
        ("hess_unicycle_xf_fixed", dict(scenario="unicycle", N=10, xf_fixed=3)),
        ("hess_unicycle_n24_ball", dict(scenario="unicycle", N=24, ball="2,1.2,0.3,0.6", tball=0.02, tball_s="1,1,0.1")),
        ("hess_pendulum_ms_rk4", dict(scenario="pendulum", grid="ms", N=8)),
        ("hess_cartpole", dict(scenario="cartpole", N=8)),
        ("hess_quad_n4", dict(scenario="quad", N=4)),
    ]:
        d = run("hess", **kv)
        with open(os.path.join(OUT, f"{name}.json"), "w") as f:
            json.dump(d, f, separators=(",", ":"))
        print(name, d["n"], len(d["heq_vals_full"]), len(d["hineq_vals_full"]), os.path.getsize(os.path.join(OUT, f"{name}.json")))


def tvref():
    """Time-varying state reference (DiscreteTimeReferenceTrajectory, one sample per grid point): what getReferenceCached(k) hands the cost
    / final-stage terms is recorded in the fixture ("ref_vertex", vertex layout).  (No control reference: the reference's least-squares
    control term with a non-zero uref writes one element of the nu-vector and leaves the rest uninitialised, quadratic_cost.cpp:160-163 --
    ref_driver's uref= option shows it.)"""
    for name, kv, keep in [
        ("unicycle_n12_tvref", dict(scenario="unicycle", N=12, iters=6, xref_traj=1), (1, 2, 3, 4, 5, 6)),
        ("vdp_tvref", dict(scenario="vdp", iters=6, xref_traj=1), (1, 2, 3, 6)),
        ("unicycle_n12_tball_tvref", dict(scenario="unicycle", N=12, iters=6, xref_traj=1, tball=1e-4, tball_s="1,1,0.1"), (1, 2, 3, 6)),
        ("vdp_teq_tvref", dict(scenario="vdp", N=12, iters=5, xref_traj=1, teq=1), (1, 2, 5)),
        ("pendulum_ms_rk4_tvref", dict(scenario="pendulum", grid="ms", N=12, iters=5, xref_traj=1), (1, 2, 5)),
        ("quad_n10_tvref", dict(scenario="quad", N=10, iters=4, xref_traj=1), (1, 2, 4)),
    ]:
        d = slim(run("dump", **kv), keep)
        with open(os.path.join(OUT, f"{name}.json"), "w") as f:
            json.dump(d, f, separators=(",", ":"))
        print(name, d["n"], d["m"], [a["chi2"] for a in d["after_iter"]][-1])
    # tracking closed loops: the reference trajectory is sampled at t + k dt, so every control step sees a shifted window of it
    for name, kv in [
        ("loop_unicycle_tvref", dict(scenario="unicycle", N=20, steps=6, iters=5, shift=1, integrator="rk4", xref_traj=1)),
        ("loop_vdp_tvref", dict(scenario="vdp", steps=5, iters=5, shift=1, integrator="euler", xref_traj=1)),
        # a plant that is NOT the controller's model: Van der Pol with damping 1.8 (the OCP keeps the default 1)
        ("loop_vdp_plant_mismatch", dict(scenario="vdp", steps=6, iters=5, shift=1, integrator="rk4", plant_a=1.8)),
    ]:
        d = run("loop", **kv)
        with open(os.path.join(OUT, f"{name}.json"), "w") as f:
            json.dump(d, f, separators=(",", ":"))
        print(name, [round(st["chi2"], 6) for st in d["steps"]])
    for name, kv in [("hess_unicycle_tvref", dict(scenario="unicycle", N=12, xref_traj=1)),
                     # the cost forms of the Hessian path against a time-varying reference: plain terms, integral edges (both ends of an interval use
                     # reference k), the shooting grid's mixed edges (the integrand along the step uses reference k)
                     ("hess_unicycle_tvref_nonlsq", dict(scenario="unicycle", N=8, xref_traj=1, lsq=0)),
                     ("hess_unicycle_tvref_integral_trap", dict(scenario="unicycle", N=8, xref_traj=1, lsq=0, integral="trap")),
                     ("hess_unicycle_tvref_ms_integral", dict(scenario="unicycle", grid="ms", N=7, xref_traj=1, lsq=0, integral="trap"))]:
        d = run("hess", **kv)
        with open(os.path.join(OUT, f"{name}.json"), "w") as f:
            json.dump(d, f, separators=(",", ":"))
        print(name, d["n"])


def msvar():
    """Time-optimal OCPs on the shooting grid: MultipleShootingVariableGrid (free dt, RK4, x_f fixed, MinimumTime) -- single solves, a
    moving-horizon sequence and the exact-Hessian operators."""
    for name, kv, keep in [
        ("int3_ms_time_optimal", dict(scenario="int3", vargrid=1, grid="ms", N=16, iters=8), (1, 2, 3, 5, 8)),
        ("int3_ms_time_optimal_n40", dict(scenario="int3", vargrid=1, grid="ms", N=40, iters=6), (1, 3, 6)),
    ]:
        d = slim(run("dump", **kv), keep)
        with open(os.path.join(OUT, f"{name}.json"), "w") as f:
            json.dump(d, f, separators=(",", ":"))
        print(name, d["n"], d["m"], [a["chi2"] for a in d["after_iter"]])
    d = run("hess", scenario="int3", vargrid=1, grid="ms", N=10)
    with open(os.path.join(OUT, "hess_int3_ms_time_optimal.json"), "w") as f:
        json.dump(d, f, separators=(",", ":"))
    print("hess_int3_ms_time_optimal", d["n"])


def mtq():
    """MinTimeQuadratic (hybrid_cost.h:189-303: minimum time + quadratic form, least-squares form) on the free-dt grids: single solves and
    the exact-Hessian operators."""
    for name, kv, keep in [
        ("dint_mtq", dict(scenario="dint", cost="mtq", iters=6, solves=2), (1, 2, 3, 6)),   # (cfg 2 runs 5 solves: 40 iterations at weight 100 amplify the FD noise to 3e-5)
        ("int3_mtq_n20", dict(scenario="int3", vargrid=1, cost="mtq", N=20, iters=6), (1, 2, 3, 6)),
        ("int3_ms_mtq", dict(scenario="int3", vargrid=1, grid="ms", cost="mtq", N=16, iters=6), (1, 2, 3, 6)),
        ("int3_mtq_last6", dict(scenario="int3", vargrid=1, cost="mtq", last_n=6, N=20, iters=6), (1, 2, 3, 6)),   # only_last_n: quadratic terms near the goal only
    ]:
        d = slim(run("dump", **kv), keep)
        with open(os.path.join(OUT, f"{name}.json"), "w") as f:
            json.dump(d, f, separators=(",", ":"))
        print(name, d["n"], d["m"], [a["chi2"] for a in d["after_iter"]])
    for name, kv in [("hess_dint_mtq", dict(scenario="dint", cost="mtq", N=12)), ("hess_int3_ms_mtq", dict(scenario="int3", vargrid=1, grid="ms", cost="mtq", N=8)),
                     ("hess_dint_mtq_last5", dict(scenario="dint", cost="mtq", last_n=5, N=12))]:
        d = run("hess", **kv)
        with open(os.path.join(OUT, f"{name}.json"), "w") as f:
            json.dump(d, f, separators=(",", ":"))
        print(name, d["n"])


def nonlsq():
    """The exact-Hessian path on costs that are NOT in least-squares form (QuadraticFormCost / QuadraticFinalStateCost with lsq_form = false:
    plain objective edges, what the reference's IPOPT / QP callers use) -- SURVEY 8f rank 4's own case."""
    for name, kv in [
        ("hess_vdp_nonlsq", dict(scenario="vdp", N=10, lsq=0)),
        ("hess_unicycle_nonlsq", dict(scenario="unicycle", N=12, lsq=0)),
        ("hess_unicycle_nonlsq_tball", dict(scenario="unicycle", N=10, lsq=0, xf_fixed=4, tball=0.02, tball_s="1,1,0.1")),
        # MinimumTime(lsq_form = false): (n - 1) dt, created twice, NOT flagged linear (stage_functions.h:73) -- finite-difference Hessians of a
        # linear term; MinTimeQuadratic(.., lsq_form = false): all three terms plain
        ("hess_dint_nonlsq", dict(scenario="dint", N=12, lsq=0)),
        ("hess_dint_mtq_nonlsq", dict(scenario="dint", N=10, cost="mtq", lsq=0)),
        ("hess_int3_ms_nonlsq", dict(scenario="int3", vargrid=1, grid="ms", N=8, lsq=0)),
        # QuadraticFormCost in integral form: TrapezoidalIntegralCostEdge / LeftSumCostEdge (finite_differences_collocation_edges.h:98-152, 323-368)
        ("hess_vdp_integral_trap", dict(scenario="vdp", N=10, lsq=0, integral="trap")),
        ("hess_unicycle_integral_trap", dict(scenario="unicycle", N=10, lsq=0, integral="trap", xf_fixed=2)),
        ("hess_unicycle_integral_left", dict(scenario="unicycle", N=10, lsq=0, integral="left")),
    ]:
        d = run("hess", **kv)
        with open(os.path.join(OUT, f"{name}.json"), "w") as f:
            json.dump(d, f, separators=(",", ":"))
        print(name, d["n"], len(d["hobj_vals_full"]))


def msmixed():
    """MultipleShootingEdgeSingleControl (multiple_shooting_edges.h:151-303): what a MultipleShootingGrid creates instead of the dynamics-only edge
    when the stage cost has integral terms (multiple_shooting_grid.cpp:70-77) -- QuadraticFormCost(Q, R, integral_form = true): ONE mixed edge
    per interval (objective part: the cost integrated along the shooting step by the grid's integrator; equality part: the defect).
    Hessian-path operators only (a mixed edge with a plain objective part is refused by LevenbergMarquardtSparse)."""
    for name, kv in [
        ("hess_unicycle_ms_integral", dict(scenario="unicycle", grid="ms", N=8, lsq=0, integral="trap")),
        ("hess_unicycle_ms_integral_xf_fixed", dict(scenario="unicycle", grid="ms", N=6, lsq=0, integral="left", xf_fixed=5)),
        ("hess_vdp_ms_integral_euler", dict(scenario="vdp", grid="ms", N=10, lsq=0, integral="trap", ms_integrator="euler")),
        ("hess_vdp_ms_integral_rk3", dict(scenario="vdp", grid="ms", N=8, lsq=0, integral="trap", ms_integrator="rk3")),
        ("hess_unicycle_ms_integral_rk5", dict(scenario="unicycle", grid="ms", N=5, lsq=0, integral="trap", ms_integrator="rk5")),
        # MinTimeQuadratic(integral_form = true, lsq_form = false) on the FiniteDifferencesVariableGrid: the dt terms (plain, twice) are filed before
        # interval 0's integral edge; with only_last_n the integral edges exist on the last intervals only (hybrid_cost.h:209)
        ("hess_dint_mtq_integral_trap", dict(scenario="dint", cost="mtq", N=8, lsq=0, integral="trap")),
        ("hess_dint_mtq_integral_left_last4", dict(scenario="dint", cost="mtq", N=10, lsq=0, integral="left", last_n=4)),
        # with a terminal equality / a TerminalBall: regular equality and inequality edges come BEFORE the mixed edges in every list and in the row order
        ("hess_unicycle_ms_integral_teq", dict(scenario="unicycle", grid="ms", N=6, lsq=0, integral="trap", teq=1, ms_integrator="rk2")),
        ("hess_unicycle_ms_integral_tball", dict(scenario="unicycle", grid="ms", N=6, lsq=0, integral="trap", tball=0.02, tball_s="1,1,0.1")),
        # the mixed edge around the big-block models (their scenarios' default grid is the shooting grid; noball: the mixed edge's device form has no stage inequality)
        ("hess_pquad_ms_integral", dict(scenario="pquad", N=5, lsq=0, integral="trap", noball=1)),
        ("hess_pquad_ms_integral_rk3", dict(scenario="pquad", N=4, lsq=0, integral="left", noball=1, ms_integrator="rk3")),
        ("hess_quad_ms_integral", dict(scenario="quad", N=4, lsq=0, integral="trap", noball=1)),
    ]:
        if len(sys.argv) > 2 and sys.argv[2] not in name:
            continue
        d = run("hess", **kv)
        with open(os.path.join(OUT, f"{name}.json"), "w") as f:
            json.dump(d, f, separators=(",", ":"))
        print(name, d["n"], len(d["hobj_vals_full"]), len(d["heq_vals_full"]))


def bigterm():
    """Final-stage constraints on the 12-state quadrotor (big-block family): TerminalBall (violated: active row) and the terminal equality."""
    for name, kv, keep in [
        ("quad_n10_tball", dict(scenario="quad", N=10, iters=5, tball=0.05, tball_s="1,1,1,0.2,0.2,0.2,0.5,0.5,0.5,0.1,0.1,0.1"), (1, 2, 3, 5)),
        ("quad_n10_tball_loose", dict(scenario="quad", N=10, iters=5, tball=4.0, tball_s="1,1,1,0.2,0.2,0.2,0.5,0.5,0.5,0.1,0.1,0.1"), (1, 2, 5)),
        ("quad_n10_teq", dict(scenario="quad", N=10, iters=5, teq=1), (1, 2, 3, 5)),
    ]:
        d = slim(run("dump", **kv), keep)
        with open(os.path.join(OUT, f"{name}.json"), "w") as f:
            json.dump(d, f, separators=(",", ":"))
        print(name, d["n"], d["m"], d["ineq"], d["eq"], [a["chi2"] for a in d["after_iter"]])



def _bench_chunk(args):
    scenario, x0, xf, kv = args
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        np.savetxt(f, np.hstack([x0, xf]), fmt="%.17g")
        path = f.name
    try:
        r = json.loads(subprocess.check_output([DRIVER, "bench", f"scenario={scenario}", f"instances={path}"] + [f"{k}={v}" for k, v in kv.items()]))
    finally:
        os.unlink(path)
    return r["chi2_sum"]


def secondary():
    """chi2 sums of the genuine reference over the workloads of bench.py's `secondary` legs (BASELINE configs 1, 2, 5 at their full
    sizes): cfg 1 Van-der-Pol batch 1, cfg 2 double integrator batch 1 x 5 warm-started solves, cfg 5 the 512 seeded
    quadrotor instances at N = 200 (one reference process per instance, 8.8 s each)."""
    from concurrent.futures import ProcessPoolExecutor
    import bench
    out = {}
    for cfg, scen in ((1, "vdp"), (2, "dint")):
        w = bench.workload(cfg, 1)
        out[f"cfg{cfg}"] = {"batch": 1, "iters": 10, "solves": w["solves"], "N": int(w["desc"].N),
                            "chi2_sum": _bench_chunk((scen, w["x0"], w["xf"], dict(iters=10, solves=w["solves"], N=w["desc"].N)))}
        print(cfg, out[f"cfg{cfg}"])
    B = 512
    w = bench.workload(5, B)
    jobs = [("quad", w["x0"][b:b + 1], w["xf"][b:b + 1], dict(iters=10, solves=1, N=w["desc"].N)) for b in range(B)]
    with ProcessPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        chi2 = list(ex.map(_bench_chunk, jobs, chunksize=4))
    out["cfg5"] = {"batch": B, "iters": 10, "solves": 1, "N": int(w["desc"].N), "chi2_sum": float(np.sum(chi2)), "chi2": [float(c) for c in chi2]}
    print(5, out["cfg5"]["chi2_sum"])
    with open(os.path.join(OUT, "bench_secondary.json"), "w") as f:
        json.dump({"source": "oracle/_ref/ref_driver bench (genuine reference), seeds 20260928+i, weights as bench.workload", **out}, f, separators=(",", ":"))



FULLQ = [
    # NON-DIAGONAL Q / R / Qf (fullq = 1 of ref_driver: off-diagonal entries 0.25 sqrt(w_i w_j)): the dense branch of QuadraticFormCost::setWeightQ /
    # setWeightR and QuadraticFinalStateCost::setWeightQf -- upper Cholesky factors, dense cost blocks
    ("unicycle_n12_fullq", dict(scenario="unicycle", N=12, iters=6, fullq=1), (1, 2, 3, 4, 5, 6)),
    # (a ZERO state reference takes another branch of the reference, quadratic_cost.cpp:108-111: `cost.noalias() = x_k.transpose() * _Q_sqrt * x_k`, a
    #  scalar assigned to the nx-vector -- rows beyond the first are uninitialised memory; nothing to pin, like the non-zero control reference)
    ("vdp_fullq", dict(scenario="vdp", iters=6, fullq=1, xf="0.4,0.1"), (1, 2, 3, 4, 5, 6)),
    ("unicycle_n12_fullq_patterns", dict(scenario="unicycle", N=12, iters=5, fullq=1, xlb="-inf,-1.5,-inf", xub="1.2,inf,inf", ulb="-0.8,-inf", uub="inf,0.6", xf_fixed=5), (1, 2, 3, 4, 5)),
    ("unicycle_n12_fullq_ms", dict(scenario="unicycle", grid="ms", N=12, iters=6, fullq=1), (1, 2, 3, 4, 5, 6)),
    ("cartpole_fullq", dict(scenario="cartpole", N=14, iters=5, fullq=1), (1, 2, 3, 4, 5)),            # nx = 4: Eigen's four-column gemv block
    ("par3_fullq", dict(scenario="par3", iters=5, fullq=1), (1, 2, 3, 4, 5)),                          # nu = 3: dense R
    ("lin33_fullq", dict(scenario='lin', nx=3, nu=3, lin_a='-0.16999999999999998,-0.338,0.807,-0.486,-0.8200000000000001,-0.482,-0.289,-0.99,-0.243', lin_b='-0.435,-0.864,0.234,-0.647,-0.391,-0.118,-0.7,-0.564,-0.051', iters=4, collocation='forward', N=14, fullq=1), (1, 2, 3, 4)),
    # ... on horizons beyond 256 grid points (the long-horizon kernels' DENSE instantiation: Jacobian and factor workspace in HBM)
    ("unicycle_n300_fullq", dict(scenario="unicycle", N=300, iters=3, fullq=1), (1, 3)),
    ("vdp_n400_fullq_ms", dict(scenario="vdp", grid="ms", N=400, iters=3, fullq=1, xf="0.4,0.1"), (1, 3)),
    ("unicycle_n12_fullq_tvref", dict(scenario="unicycle", N=12, iters=5, fullq=1, xref_traj=1), (1, 2, 3, 4, 5)),
]


def fullq():
    for name, kv, keep in FULLQ:
        d = slim(run("dump", **kv), keep)
        with open(os.path.join(OUT, f"{name}.json"), "w") as f:
            json.dump(d, f, separators=(",", ":"))
        print(name, {k: d[k] for k in ("n", "m", "nnz")}, "chi2", d["after_iter"][-1]["chi2"])
    d = run("hess", scenario="unicycle", N=10, fullq=1)
    with open(os.path.join(OUT, "hess_unicycle_fullq.json"), "w") as f:
        json.dump(d, f, separators=(",", ":"))
    print("hess_unicycle_fullq")



MSINT = [
    # the shooting grids' other explicit integrators (explicit_integrators.h:47-213): Euler, Runge-Kutta 2 / 3 instead of Runge-Kutta 4
    ("vdp_ms_euler", dict(scenario="vdp", grid="ms", iters=6, ms_integrator="euler"), (1, 2, 3, 4, 5, 6)),
    ("unicycle_n12_ms_rk2", dict(scenario="unicycle", grid="ms", N=12, iters=6, ms_integrator="rk2"), (1, 2, 3, 4, 5, 6)),
    ("pendulum_ms_rk3", dict(scenario="pendulum", grid="ms", N=16, iters=5, ms_integrator="rk3"), (1, 2, 3, 4, 5)),
    ("cartpole_ms_rk2", dict(scenario="cartpole", grid="ms", N=16, iters=5, ms_integrator="rk2"), (1, 2, 3, 4, 5)),
    ("int3_ms_time_optimal_rk2", dict(scenario="int3", grid="ms", vargrid=1, N=25, iters=8, w="100,100,100", ms_integrator="rk2"), (1, 4, 8)),
    ("quad_n10_rk3", dict(scenario="quad", N=10, iters=6, ms_integrator="rk3"), (1, 2, 4, 6)),
    ("quad_n10_euler", dict(scenario="quad", N=10, iters=6, ms_integrator="euler"), (1, 2, 4, 6)),
    # Runge-Kutta 5 / 6 / 7 (explicit_integrators.h:327-628: six / eight / eleven stages), families with nx <= 4
    ("vdp_ms_rk5", dict(scenario="vdp", grid="ms", iters=6, ms_integrator="rk5"), (1, 2, 3, 4, 5, 6)),
    ("unicycle_n12_ms_rk6", dict(scenario="unicycle", grid="ms", N=12, iters=6, ms_integrator="rk6"), (1, 2, 3, 4, 5, 6)),
    ("cartpole_ms_rk7", dict(scenario="cartpole", grid="ms", N=16, iters=5, ms_integrator="rk7"), (1, 2, 3, 4, 5)),
    ("pendulum_ms_rk5", dict(scenario="pendulum", grid="ms", N=16, iters=5, ms_integrator="rk5"), (1, 2, 3, 4, 5)),
    ("int3_ms_time_optimal_rk7", dict(scenario="int3", grid="ms", vargrid=1, N=25, iters=8, w="100,100,100", ms_integrator="rk7"), (1, 4, 8)),
    ("par3_ms_rk6", dict(scenario="par3", grid="ms", iters=5, ms_integrator="rk6"), (1, 2, 3, 4, 5)),
]


def msint():
    only = sys.argv[2] if len(sys.argv) > 2 else ""   # e.g. `gen_golden.py msint rk` regenerates the Runge-Kutta 5 - 7 fixtures only
    for name, kv, keep in MSINT:
        if only and only not in name:
            continue
        d = slim(run("dump", **kv), keep)
        with open(os.path.join(OUT, f"{name}.json"), "w") as f:
            json.dump(d, f, separators=(",", ":"))
        print(name, {k: d[k] for k in ("n", "m", "nnz")}, "chi2", d["after_iter"][-1]["chi2"])
    d = run("hess", scenario="pendulum", grid="ms", N=8, ms_integrator="rk5")   # the Hessian-path operators with a six-stage integrator in the defect edges
    with open(os.path.join(OUT, "hess_pendulum_ms_rk5.json"), "w") as f:
        json.dump(d, f, separators=(",", ":"))



USERMODEL = [
    # a USER dynamics model (control_box_rst_amd/csrc/models/kinematic_car.hpp <-> class KinematicCarRef of ref_driver.cpp): fixed grid, another scheme, shooting
    ("kcar_n16", dict(scenario="kcar", N=16, iters=6), (1, 2, 3, 4, 5, 6)),
    ("kcar_midpoint", dict(scenario="kcar", N=12, iters=4, collocation="midpoint"), (1, 2, 3, 4)),
    ("kcar_ms_rk4", dict(scenario="kcar", grid="ms", N=12, iters=5), (1, 2, 3, 4, 5)),
]


PTEQ = [
    # TerminalPartialEqualityConstraint (final_state_constraints.h:198-300): equality rows on a subset of the components of x_f
    ("unicycle_n12_pteq", dict(scenario="unicycle", N=12, iters=6, teq=1, teq_mask=5), (1, 2, 3, 4, 5, 6)),
    ("vdp_pteq", dict(scenario="vdp", iters=6, teq=1, teq_mask=2), (1, 2, 3, 4, 5, 6)),
    ("cartpole_pteq", dict(scenario="cartpole", N=14, iters=5, teq=1, teq_mask=9), (1, 2, 3, 4, 5)),
    ("unicycle_n12_ms_pteq", dict(scenario="unicycle", grid="ms", N=12, iters=5, teq=1, teq_mask=3, xf_fixed=4), (1, 2, 3, 4, 5)),
]


def pteq():
    for name, kv, keep in PTEQ:
        d = slim(run("dump", **kv), keep)
        with open(os.path.join(OUT, f"{name}.json"), "w") as f:
            json.dump(d, f, separators=(",", ":"))
        print(name, {k: d[k] for k in ("n", "m", "nnz")}, "chi2", d["after_iter"][-1]["chi2"])


BIGMODEL = [
    # a USER dynamics model of the big-block family (csrc/models/planar_quadrotor.hpp <-> class PlanarQuadrotorRef of ref_driver.cpp; nx = 6, nu = 2):
    # multiple shooting + RK4, thrust bounds, keep-out ball; final-stage constraints; another integrator
    ("pquad_n10", dict(scenario="pquad", N=10, iters=6), (1, 2, 3, 4, 5, 6)),
    # round 6: a USER stage inequality (state term) around the 12-state quadrotor -- the tilt cone of csrc/stage_functions/tilt_cone.hpp instead of the keep-out ball
    ("sf_quad_tilt", dict(scenario="quad", N=10, iters=5, noball=1, tilt=0.15), (1, 2, 3, 5)),
    ("sf_quad_fd_tilt", dict(scenario="quad", grid="fd", N=10, iters=5, noball=1, tilt=0.15), (1, 2, 3, 5)),
    ("pquad_n24", dict(scenario="pquad", N=24, iters=8), (1, 2, 4, 8)),
    ("pquad_n10_teq", dict(scenario="pquad", N=10, iters=5, teq=1), (1, 2, 3, 5)),
    ("pquad_n10_tball", dict(scenario="pquad", N=10, iters=5, tball=0.05, tball_s="1,1,0.5,0.2,0.2,0.1"), (1, 2, 3, 5)),
    ("pquad_n10_rk3", dict(scenario="pquad", N=10, iters=6, ms_integrator="rk3"), (1, 2, 4, 6)),
    # the big-block family on the FiniteDifferencesGrid: the four collocation formulas (dense x_{k+1} block in the stage kernel)
    ("pquad_fd_n10", dict(scenario="pquad", grid="fd", N=10, iters=6), (1, 2, 3, 4, 5, 6)),
    ("pquad_fd_n24", dict(scenario="pquad", grid="fd", N=24, iters=8), (1, 2, 4, 8)),
    ("pquad_fd_n10_forward", dict(scenario="pquad", grid="fd", N=10, iters=5, collocation="forward"), (1, 2, 3, 5)),
    ("pquad_fd_n10_backward", dict(scenario="pquad", grid="fd", N=10, iters=5, collocation="backward"), (1, 2, 3, 5)),
    ("pquad_fd_n10_midpoint", dict(scenario="pquad", grid="fd", N=10, iters=5, collocation="midpoint"), (1, 2, 3, 5)),
    ("pquad_fd_n10_teq", dict(scenario="pquad", grid="fd", N=10, iters=5, teq=1), (1, 2, 3, 5)),
    ("quad_fd_n10", dict(scenario="quad", grid="fd", N=10, iters=6), (1, 2, 4, 6)),
    # a FREE dt around the big-block models (time-optimal transfer to a fixed x_f, MinimumTime): MultipleShootingVariableGrid / FiniteDifferencesVariableGrid
    ("pquad_topt_n10", dict(scenario="pquad", vargrid=1, N=10, iters=6, w="100,100,100"), (1, 2, 3, 6)),
    ("pquad_topt_n30", dict(scenario="pquad", vargrid=1, N=30, iters=8, w="100,100,100"), (1, 4, 8)),
    ("pquad_fd_topt_n12", dict(scenario="pquad", grid="fd", vargrid=1, N=12, iters=6, w="100,100,100"), (1, 2, 3, 6)),
    ("quad_topt_n8", dict(scenario="quad", vargrid=1, N=8, iters=5, w="100,100,100"), (1, 2, 5)),
    # TerminalPartialEqualityConstraint around the big-block models (VERDICT r4 "missing" 3: the cap nx <= 4 is gone): position + attitude of the planar
    # quadrotor pinned at the goal, velocities free; position only for the 12-state quadrotor, shooting and collocation grid
    ("pquad_n10_pteq", dict(scenario="pquad", N=10, iters=5, teq=1, teq_mask=7), (1, 2, 3, 5)),
    ("pquad_fd_n10_pteq", dict(scenario="pquad", grid="fd", N=10, iters=5, teq=1, teq_mask=0b101011), (1, 2, 3, 5)),
    ("quad_n10_pteq", dict(scenario="quad", N=10, iters=5, teq=1, teq_mask=7), (1, 2, 3, 5)),
    # Runge-Kutta 5 / 6 / 7 around the big-block models (VERDICT r4 "missing" 3: the cap nx <= 4 is gone), incl. a free dt around the 12-state quadrotor
    ("quad_n10_rk5", dict(scenario="quad", N=10, iters=5, ms_integrator="rk5"), (1, 2, 3, 5)),
    ("quad_n10_rk7", dict(scenario="quad", N=10, iters=5, ms_integrator="rk7"), (1, 3, 5)),
    ("pquad_n10_rk6", dict(scenario="pquad", N=10, iters=6, ms_integrator="rk6"), (1, 2, 4, 6)),
    ("quad_topt_n8_rk6", dict(scenario="quad", vargrid=1, N=8, iters=5, w="100,100,100", ms_integrator="rk6"), (1, 2, 5)),
]


def bigmodel():
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, kv, keep in BIGMODEL:
        if only and only not in name:
            continue
        d = slim(run("dump", **kv), keep)
        with open(os.path.join(OUT, f"{name}.json"), "w") as f:
            json.dump(d, f, separators=(",", ":"))
        print(name, {k: d[k] for k in ("n", "m", "nnz")}, "chi2", [a["chi2"] for a in d["after_iter"]])
    for name, kv in (("hess_pquad_n5", dict(scenario="pquad", N=5)), ("hess_pquad_fd_n5", dict(scenario="pquad", grid="fd", N=5))):
        if only and only not in name:
            continue
        d = run("hess", **kv)
        with open(os.path.join(OUT, f"{name}.json"), "w") as f:
            json.dump(d, f, separators=(",", ":"))
    if only:
        return
    for name, mode, kv in [("mpc_pquad_shift_init", "mpc", dict(scenario="pquad", N=10, steps=3, iters=0, iters0=4, shift=1)),
                           ("loop_pquad_rk4", "loop", dict(scenario="pquad", N=10, steps=3, iters=4, shift=1, integrator="rk4", disturbance=0.002))]:
        d = run(mode, **kv)
        with open(os.path.join(OUT, f"{name}.json"), "w") as f:
            json.dump(d, f, separators=(",", ":"))
        print(name, [round(st["chi2"], 6) for st in d["steps"]])


# Integral-form constraint edges and the control-deviation term: user stage functions of ref_driver
# (UserStageInequalities: the keep-out ball as INTEGRAL state-control term, an input-rate limit as control-deviation term; LinearIntegralEquality),
# both integration rules of the grid (crule), fixed- and free-dt finite-differences grids.
XE = [
    ("xe_unicycle_ballint_trap", dict(scenario="unicycle", N=12, iters=6, crule="trap", ball="1.0,0.5,0.2,0.3", ball_int=1), (1, 2, 3, 4, 6)),
    ("xe_unicycle_ballint_left", dict(scenario="unicycle", N=12, iters=6, crule="left", ball="1.0,0.5,0.2,0.3", ball_int=1), (1, 2, 3, 4, 6)),
    ("xe_unicycle_eqlin_trap", dict(scenario="unicycle", N=12, iters=6, crule="trap", eq_lin="0.3,-0.2,0.1,0.05,0.02,0.1"), (1, 2, 3, 4, 6)),
    ("xe_unicycle_eqlin_left", dict(scenario="unicycle", N=12, iters=6, crule="left", eq_lin="0.3,-0.2,0.1,0.05,0.02,0.1"), (1, 2, 3, 4, 6)),
    ("xe_unicycle_rate", dict(scenario="unicycle", N=12, iters=6, rate="0.8,0.5", u_prev="0.2,-0.1", u_prev_dt=0.07), (1, 2, 3, 4, 6)),
    ("xe_unicycle_rate_default_prev", dict(scenario="unicycle", N=10, iters=4, rate="0.6,0.4"), (1, 2, 4)),
    ("xe_unicycle_all_n40", dict(scenario="unicycle", N=40, iters=8, crule="trap", ball="1.0,0.5,0.2,0.3", ball_int=1, eq_lin="0.3,-0.2,0.1,0.05,0.02,0.1",
                                 rate="0.9,0.6", u_prev="0.1,0.1"), (1, 2, 4, 8)),
    ("xe_unicycle_all_left_n100", dict(scenario="unicycle", N=100, iters=5, crule="left", ball="1.0,0.5,0.2,0.3", ball_int=1, eq_lin="0.3,-0.2,0.1,0.05,0.02,0.1",
                                       rate="0.9,0.6"), (1, 2, 5)),
    ("xe_vdp_eqlin_rate", dict(scenario="vdp", N=20, iters=6, crule="trap", eq_lin="0.3,-0.2,0.05,0.1", rate="0.7"), (1, 2, 3, 6)),
    ("xe_int3_vargrid_left", dict(scenario="int3", N=20, iters=5, vargrid=1, xf="1.0,0.0,0.0", crule="left", eq_lin="0.01,0.02,0.0,0.05,0.0", rate="3.0"), (1, 2, 3, 5)),
    ("xe_int3_vargrid_trap", dict(scenario="int3", N=20, iters=5, vargrid=1, xf="1.0,0.0,0.0", crule="trap", eq_lin="0.01,0.02,0.0,0.05,0.0"), (1, 2, 3, 5)),
    ("xe_rocket_rate_eq", dict(scenario="rocket", N=16, iters=5, crule="trap", eq_lin="0.1,0.0,0.05,0.02,0.0", rate="0.5"), (1, 2, 3, 5)),
    ("xe_par3_rate_eq", dict(scenario="par3", N=14, iters=5, crule="left", eq_lin="0.1,0.0,0.05,0.02,0.0,0.01,0.0", rate="0.8,0.6,0.9"), (1, 2, 3, 5)),
    # round 5 (VERDICT r4 "missing" 1): the same edge kinds around the big-block models, on the shooting grids (there the control-deviation term only:
    # integral terms make a shooting interval a mixed edge, multiple_shooting_grid.cpp:70-77) and beyond 256 grid points
    ("xe_quad_fd_ballint_trap", dict(scenario="quad", grid="fd", N=10, iters=5, crule="trap", ball="1.0,0.5,0.6,0.4", ball_int=1), (1, 2, 3, 5)),
    ("xe_quad_ms_rate", dict(scenario="quad", N=10, iters=5, ball="1.0,0.5,0.6,0.4", rate="4.0,2.0,2.0,2.0"), (1, 2, 3, 5)),
    ("xe_pquad_ms_rate", dict(scenario="pquad", N=10, iters=5, ball="1.0,0.5,0.0,0.3", rate="40.0,40.0"), (1, 2, 3, 5)),
    ("xe_pquad_fd_all_left", dict(scenario="pquad", grid="fd", N=10, iters=5, crule="left", ball="1.0,0.5,0.0,0.3", ball_int=1,
                                  eq_lin="0.1,0.0,0.05,0.02,0.0,0.01,0.03,0.02,0.4", rate="3.0,3.0"), (1, 2, 3, 5)),
    ("xe_unicycle_ms_rate", dict(scenario="unicycle", grid="ms", N=12, iters=6, ball="1.0,0.5,0.2,0.3", rate="0.8,0.5", u_prev="0.2,-0.1", u_prev_dt=0.07), (1, 2, 3, 4, 6)),
    ("xe_int3_ms_vargrid_rate", dict(scenario="int3", grid="ms", N=20, iters=5, vargrid=1, xf="1.0,0.0,0.0", rate="3.0"), (1, 2, 3, 5)),
    # round 6: USER stage functions (csrc/stage_functions/): the input-magnitude bound as the stage inequalities' control term (an edge on u_k), alone, next to the
    # ball (state term) and the rate limit (control-deviation term: the three edges of an interval in the reference's order), around a big-block model
    ("xe_sf_unicycle_unorm", dict(scenario="unicycle", N=12, iters=6, unorm=0.6), (1, 2, 3, 4, 6)),
    ("xe_sf_unicycle_ball_unorm_rate", dict(scenario="unicycle", N=16, iters=6, ball="1.0,0.5,0.2,0.3", unorm=0.7, rate="0.8,0.5", u_prev="0.2,-0.1", u_prev_dt=0.07), (1, 2, 3, 4, 6)),
    ("xe_sf_int3_unorm_vargrid", dict(scenario="int3", N=20, iters=5, vargrid=1, xf="1.0,0.0,0.0", unorm=1.5), (1, 2, 3, 5)),
    ("xe_sf_quad_tilt_unorm", dict(scenario="quad", N=10, iters=5, noball=1, tilt=0.15, unorm=11.0), (1, 2, 3, 5)),
    # ... and a horizon of 200 grid points: the BIG instantiation of the block-tridiagonal route (129 .. 256 grid points)
    # non-diagonal weights TOGETHER with extra edges (band route: the DENSE x XE sweep instantiation)
    ("xe_unicycle_rate_fullq", dict(scenario="unicycle", N=12, iters=6, fullq=1, rate="0.8,0.5", u_prev="0.2,-0.1", u_prev_dt=0.07), (1, 2, 3, 4, 6)),
    ("xe_vdp_eqlin_rate_fullq", dict(scenario="vdp", N=20, iters=6, fullq=1, xf="0.4,0.1", crule="trap", eq_lin="0.3,-0.2,0.05,0.1", rate="0.7"), (1, 2, 3, 6)),   # (xf: a ZERO state reference is the reference's 1 x 1-into-vector branch, see vdp_fullq)
    ("xe_unicycle_rate_fullq_n300", dict(scenario="unicycle", N=300, dt=0.04, iters=3, fullq=1, rate="0.9,0.6", u_prev="0.1,0.1"), (1, 3)),
    ("xe_unicycle_rate_n200", dict(scenario="unicycle", N=200, dt=0.05, iters=4, rate="0.9,0.6", u_prev="0.1,0.1"), (1, 2, 4)),
    ("xe_unicycle_all_n300", dict(scenario="unicycle", N=300, dt=0.04, iters=4, crule="trap", ball="1.0,0.5,0.2,0.3", ball_int=1, eq_lin="0.3,-0.2,0.1,0.05,0.02,0.1",
                                  rate="0.9,0.6"), (1, 2, 4)),
]


def xe():
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, kv, keep in XE:
        if only and only not in name:
            continue
        d = slim(run("dump", **kv), keep)
        with open(os.path.join(OUT, f"{name}.json"), "w") as f:
            json.dump(d, f, separators=(",", ":"))
        print(name, {k: d[k] for k in ("n", "m", "nnz")}, "chi2", [round(a["chi2"], 6) for a in d["after_iter"]])


def usermodel():
    for name, kv, keep in USERMODEL:
        d = slim(run("dump", **kv), keep)
        with open(os.path.join(OUT, f"{name}.json"), "w") as f:
            json.dump(d, f, separators=(",", ":"))
        print(name, {k: d[k] for k in ("n", "m", "nnz")}, "chi2", d["after_iter"][-1]["chi2"])
    d = run("hess", scenario="kcar", N=10)
    with open(os.path.join(OUT, "hess_kcar.json"), "w") as f:
        json.dump(d, f, separators=(",", ":"))


def hesspteq():
    """Hessian-path operators with a TerminalPartialEqualityConstraint (final_state_constraints.h:236-252): equality rows, multipliers and
    linear-form rows for the active components of x_f only."""
    for name, kv in [
        ("hess_unicycle_pteq", dict(scenario="unicycle", N=10, teq=1, teq_mask=5)),
        ("hess_cartpole_pteq", dict(scenario="cartpole", N=8, teq=1, teq_mask=6)),
        ("hess_unicycle_ms_pteq", dict(scenario="unicycle", grid="ms", N=8, teq=1, teq_mask=3)),
        ("hess_unicycle_ms_integral_pteq", dict(scenario="unicycle", grid="ms", N=6, lsq=0, integral="trap", teq=1, teq_mask=6)),
        ("hess_pquad_pteq", dict(scenario="pquad", N=5, teq=1, teq_mask=0b100101)),
    ]:
        d = run("hess", **kv)
        with open(os.path.join(OUT, f"{name}.json"), "w") as f:
            json.dump(d, f, separators=(",", ":"))
        print(name, d["n"], d.get("eq"), len(d["heq_vals_full"]), os.path.getsize(os.path.join(OUT, f"{name}.json")))


FUZZ_SEEDS = (23091,)   # tests/test_gpu_fuzz.py::test_random_descriptor_vs_oracle seeds whose device result needed the 48-trial spread (stage 2)


def fuzzseed():
    """Randomized-start cases of the fuzz suite pinned to the reference: the seed's descriptor, weights and noisy start re-drawn exactly as
    tests/test_gpu_fuzz.py draws them, handed to the reference through ref_driver's start= (its parameter vector) -- every instance of the seed."""
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_fuzz as F
    from control_box_rst_amd import capi
    from control_box_rst_amd.solver import init_trajectory
    for seed in FUZZ_SEEDS:
        rng = np.random.default_rng(1000 + seed)
        fam, d = F.random_desc(rng, long_horizon=(seed >= 10000))
        assert fam == "unicycle" and d.grid == capi.GRID_MS and d.xf_fixed_mask == 0, "fuzzseed: only what ref_driver's start= covers"
        B = 3
        w = tuple(float(v) for v in rng.uniform(1.0, 50.0, 3))
        x0 = rng.uniform(-1, 1, (B, d.nx))
        xf = rng.uniform(-1, 1, (B, d.nx)) + np.array([1.5, 0.5, 0.2, 0.0])[: d.nx]
        X0 = init_trajectory(d, x0, xf)
        X0 = X0 + 0.05 * rng.normal(size=X0.shape)
        X0[:, : d.nx] = x0
        inf = lambda v: ("inf" if v > 0 else "-inf") if abs(v) >= 1e30 else repr(float(v))   # noqa: E731
        for b in range(B):
            with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
                f.write(" ".join(repr(float(v)) for v in X0[b, d.nx:]))   # active parameters: everything behind the fixed x_0
            kv = dict(scenario="unicycle", grid="ms", N=d.N, dt=repr(float(d.dt_ref)), iters=3, w=vec(w), x0=vec(x0[b]), xf=vec(xf[b]),
                      qdiag=vec(list(d.q_diag)[:3]), rdiag=vec(list(d.r_diag)[:2]), qfdiag=vec(list(d.qf_diag)[:3]), final_cost=int(d.final_cost),
                      xlb=",".join(inf(v) for v in list(d.x_lb)[:3]), xub=",".join(inf(v) for v in list(d.x_ub)[:3]),
                      ulb=",".join(inf(v) for v in list(d.u_lb)[:2]), uub=",".join(inf(v) for v in list(d.u_ub)[:2]), start=f.name)
            if d.stage_ineq == capi.INEQ_BALL:
                kv["ball"] = vec(list(d.ineq_params)[:4])
            else:
                kv["noball"] = 1
            if d.final_ineq == capi.FINAL_INEQ_TERMINAL_BALL:
                kv["tball"] = repr(float(d.final_ineq_params[3]))
                kv["tball_s"] = vec(list(d.final_ineq_params)[:3])
            if d.final_eq:
                kv["teq"] = 1
            g = slim(run("dump", **kv), (1, 2, 3))
            os.unlink(f.name)
            g["fuzz_seed"], g["fuzz_instance"] = seed, b
            # the reference's own reproducibility on this case: the same solve from starts ONE ulp away (one component each, spread over the
            # horizon) -- how far its third iterate moves, per perturbed start (heavy-tailed: most starts stay close, a few jump)
            base, base_chi2 = np.array(g["after_iter"][-1]["vertex"]), g["after_iter"][-1]["chi2"]
            n_act = X0.shape[1] - d.nx
            ulp_dx, ulp_dchi2 = [], []
            for j in range(64):
                xs = X0[b, d.nx:].copy()
                c = (j * 97) % n_act
                xs[c] = np.nextafter(xs[c], 10.0)
                with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f2:
                    f2.write(" ".join(repr(float(v)) for v in xs))
                last = run("dump", **dict(kv, start=f2.name))["after_iter"][-1]
                os.unlink(f2.name)
                ulp_dx.append(float(np.abs(np.array(last["vertex"]) - base).max()))
                ulp_dchi2.append(float(abs(last["chi2"] - base_chi2) / abs(base_chi2)))
            g["ulp_dx"], g["ulp_dchi2"] = ulp_dx, ulp_dchi2
            assert np.array_equal(np.array(g["vertex_init"])[: X0.shape[1]], X0[b]), "the reference did not start where the fuzz test starts"
            name = f"fuzz_{seed}_b{b}"
            with open(os.path.join(OUT, f"{name}.json"), "w") as fo:
                json.dump(g, fo, separators=(",", ":"))
            print(name, {k: g[k] for k in ("n", "m", "nnz")}, "chi2", [a["chi2"] for a in g["after_iter"]],
                  "own one-ulp spread: x median %.2e max %.2e, chi2 max %.2e" % (np.median(ulp_dx), max(ulp_dx), max(ulp_dchi2)))


# ---- tolerance ledger: tests/tolerances.json = fixture -> tolerance of the device iterate / chi2 against the fixture, next to the
# REFERENCE's own reproducibility on that very fixture: the genuine solver re-run from inputs ONE ulp away (one component of x0 / xf per trial; a
# component that is exactly 0 is moved by one ulp of 1.0).  tests/test_gpu_parity.py reads the ledger and FAILS when a tolerance beyond the hard gate
# of SURVEY 8d (1e-5 on the trajectory, 2e-6 relative on chi2) exceeds 4 x the recorded spread.
LEDGER_WIDENED = [   # the fixtures whose tolerance is wider than the default, with the ref_driver arguments that made them
    ("quad_n10", dict(scenario="quad", N=10, iters=6)),
    ("quad_n10_tball", dict(scenario="quad", N=10, iters=5, tball=0.05, tball_s="1,1,1,0.2,0.2,0.2,0.5,0.5,0.5,0.1,0.1,0.1")),
    ("quad_n10_tball_loose", dict(scenario="quad", N=10, iters=5, tball=4.0, tball_s="1,1,1,0.2,0.2,0.2,0.5,0.5,0.5,0.1,0.1,0.1")),
    ("quad_n10_teq", dict(scenario="quad", N=10, iters=5, teq=1)),
    ("quad_n10_rk3", dict(scenario="quad", N=10, iters=6, ms_integrator="rk3")),
    ("quad_n10_euler", dict(scenario="quad", N=10, iters=6, ms_integrator="euler")),
    ("pendulum_ms_rk4", dict(scenario="pendulum", grid="ms", N=16, iters=5)),
]


def _ledger_trials(g, kv, n_trials):
    """(max |x - base|_inf over the trials and over the iterates the fixture records, max relative chi2 difference, trials) for one fixture."""
    x0, xf = np.array(g["x0"], dtype=float), np.array(g["xf"], dtype=float)
    keep = [a["k"] for a in g["after_iter"]]
    nv = len(g["after_iter"][-1]["vertex"])

    def iterates(args):
        d = run("dump", **args)
        return {a["k"]: (np.array(a["vertex"]), a["chi2"]) for a in d["after_iter"] if a["k"] in keep}
    base = iterates(dict(kv, x0=vec(x0), xf=vec(xf)))
    for a in g["after_iter"]:
        assert np.array_equal(base[a["k"]][0], np.array(a["vertex"])), "the ledger's arguments do not reproduce the committed fixture"
    comps = [("x0", i) for i in range(len(x0))] + [("xf", i) for i in range(len(xf))]
    # the grid's dt first (it multiplies every dynamics evaluation: the input whose last bit reaches the defects like a last-bit difference of sin / cos
    # between two math libraries does), then the non-zero components (a true one-ulp move), then the zeros (one ulp of 1.0); both signs
    comps.sort(key=lambda c: (0 if (x0 if c[0] == "x0" else xf)[c[1]] != 0.0 else 1))
    comps = [("dt", 0)] + comps
    dx, dc, t = 0.0, 0.0, 0
    skip = len(x0)   # (x_0 itself is an input: not part of the comparison)
    for sign in (+1.0, -1.0):
        for which, i in comps:
            if t >= n_trials:
                break
            a, b, extra = x0.copy(), xf.copy(), {}
            if which == "dt":
                extra["dt"] = repr(float(np.nextafter(float(g["dt"]), sign * np.inf)))
            else:
                v = a if which == "x0" else b
                v[i] = np.nextafter(v[i], sign * np.inf) if v[i] != 0.0 else sign * np.finfo(float).eps
            for k, (x, c) in iterates(dict(kv, x0=vec(a), xf=vec(b), **extra)).items():
                dx = max(dx, float(np.abs(x - base[k][0])[skip:nv].max()))
                dc = max(dc, abs(c - base[k][1]) / max(1.0, abs(base[k][1])))
            t += 1
    return dx, dc, t


def _ledger_job(args):
    name, kv, n_trials = args
    g = json.load(open(os.path.join(OUT, f"{name}.json")))
    return (name,) + _ledger_trials(g, kv, n_trials)


def tolerances():
    from concurrent.futures import ProcessPoolExecutor
    HARD_X, HARD_CHI2, DEF_X, DEF_CHI2 = 1e-5, 2e-6, 5e-6, 2e-6
    jobs = [(n, kv, 16) for n, kv in LEDGER_WIDENED] + [(n, kv, 16) for n, kv, _ in BIGMODEL] + [(n, kv, 16) for n, kv, _ in XE if ("quad" in n or "_ms_" in n)]   # (every fixture of the big-block models; the shooting-grid fixtures with a control-deviation term)
    path = os.path.join(ROOT, "tests", "tolerances.json")
    old = json.load(open(path))["fixtures"] if os.path.exists(path) else {}
    fixtures = {}
    with ProcessPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        for name, dx, dc, t in ex.map(_ledger_job, jobs):
            def nice(v):   # tolerance = 4 x spread, rounded UP to two significant digits (4 x means <= 4.4 x)
                if v <= 0:
                    return 0.0
                e = 10.0 ** (np.floor(np.log10(v)) - 1)
                return float(np.ceil(v / e - 1e-9) * e)
            x_tol = max(DEF_X, nice(4 * dx)) if 4 * dx > HARD_X else DEF_X
            c_tol = max(DEF_CHI2, nice(4 * dc)) if 4 * dc > HARD_CHI2 else DEF_CHI2
            fixtures[name] = {"x_tol": x_tol, "chi2_rtol": c_tol, "ref_spread_x": dx, "ref_spread_chi2": dc, "trials": t}
            if name in old and "note" in old[name]:
                fixtures[name]["note"] = old[name]["note"]
            print("%-24s spread x %.2e chi2 %.2e (%d trials) -> x_tol %.1e chi2_rtol %.1e" % (name, dx, dc, t, x_tol, c_tol))
    ledger = {
        "_rule": "x_tol <= max(hard_x, 4 * ref_spread_x) and chi2_rtol <= max(hard_chi2, 4 * ref_spread_chi2); fixtures not listed use the defaults. "
                 "ref_spread_* = the genuine reference against itself from inputs one ulp away (oracle/gen_golden.py tolerances).",
        "hard_x": HARD_X, "hard_chi2": HARD_CHI2, "default_x_tol": DEF_X, "default_chi2_rtol": DEF_CHI2,
        "fixtures": dict(sorted(fixtures.items())),
    }
    with open(path, "w") as f:
        json.dump(ledger, f, indent=1)


def main():
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "tolerances":
        return tolerances()
    if len(sys.argv) > 1 and sys.argv[1] == "fuzzseed":
        return fuzzseed()
    if len(sys.argv) > 1 and sys.argv[1] == "hesspteq":
        return hesspteq()
    if len(sys.argv) > 1 and sys.argv[1] == "pteq":
        return pteq()
    if len(sys.argv) > 1 and sys.argv[1] == "xe":
        return xe()
    if len(sys.argv) > 1 and sys.argv[1] == "usermodel":
        return usermodel()
    if len(sys.argv) > 1 and sys.argv[1] == "bigmodel":
        return bigmodel()
    if len(sys.argv) > 1 and sys.argv[1] == "msint":
        return msint()
    if len(sys.argv) > 1 and sys.argv[1] == "fullq":
        return fullq()
    if len(sys.argv) > 1 and sys.argv[1] == "secondary":
        return secondary()
    if len(sys.argv) > 1 and sys.argv[1] == "msvar":
        return msvar()
    if len(sys.argv) > 1 and sys.argv[1] == "bigterm":
        return bigterm()
    if len(sys.argv) > 1 and sys.argv[1] == "hess":
        return hess()
    if len(sys.argv) > 1 and sys.argv[1] == "tvref":
        return tvref()
    if len(sys.argv) > 1 and sys.argv[1] == "fullsize":
        return fullsize()
    if len(sys.argv) > 1 and sys.argv[1] == "adapt":
        return adapt()
    if len(sys.argv) > 1 and sys.argv[1] == "mtq":
        return mtq()
    if len(sys.argv) > 1 and sys.argv[1] == "nonlsq":
        return nonlsq()
    if len(sys.argv) > 1 and sys.argv[1] == "msmixed":
        return msmixed()
    # cfg 3 (headline structure, single instance, the SURVEY 8c known-answer trace), cfg 1, cfg 2
    for name, kv, keep in [
        ("unicycle", dict(scenario="unicycle"), (1, 2, 5, 10)),
        ("vdp", dict(scenario="vdp"), (1, 2, 5, 10)),
        ("dint", dict(scenario="dint"), (1, 5, 10)),
        # other collocation schemes on the small problem (finite_differences_collocation.h:119-241)
        ("vdp_forward", dict(scenario="vdp", collocation="forward", iters=5), (1, 5)),
        ("vdp_backward", dict(scenario="vdp", collocation="backward", iters=5), (1, 5)),
        ("vdp_midpoint", dict(scenario="vdp", collocation="midpoint", iters=5), (1, 5)),
        # a short unicycle horizon: cheap full check incl. every iterate
        ("unicycle_n12", dict(scenario="unicycle", N=12, iters=6), (1, 2, 3, 4, 5, 6)),
        # reduced cfg 5: quadrotor, MultipleShootingGrid + RK4, u bounds, keep-out ball inequality
        ("quad_n10", dict(scenario="quad", N=10, iters=6), (1, 2, 4, 6)),
        # TerminalBall final-stage inequality (final_state_constraints.h:38-96): violated on the short horizon (active row), and on
        # the small problem with a radius that lets it switch between active and inactive
        ("unicycle_n12_tball", dict(scenario="unicycle", N=12, iters=6, tball=1e-4, tball_s="1,1,0.1"), (1, 2, 3, 4, 5, 6)),
        ("vdp_tball", dict(scenario="vdp", iters=6, tball=0.02, tball_s="1,2"), (1, 2, 3, 4, 5, 6)),
        # MultipleShootingGrid + explicit RK4 on the small models (multiple_shooting_grid.cpp:38-197, explicit_integrators.h:280-295)
        ("vdp_ms_rk4", dict(scenario="vdp", grid="ms", iters=6), (1, 2, 3, 4, 5, 6)),
        ("unicycle_n12_ms_rk4", dict(scenario="unicycle", grid="ms", N=12, iters=6), (1, 2, 3, 4, 5, 6)),
        # stage inequality (keep-out ball in the first three state components) on the small-block family, with a terminal ball
        # TerminalEqualityConstraint x_f = xref (final_state_constraints.h:130-160): nx more equality rows on x_f
        ("unicycle_n12_teq", dict(scenario="unicycle", N=12, iters=6, teq=1), (1, 2, 3, 4, 5, 6)),
        ("vdp_teq", dict(scenario="vdp", iters=6, teq=1), (1, 2, 3, 4, 5, 6)),
        # bound patterns with unbounded and one-sided components (finite <=> lb > -2e30 || ub < 2e30, vector_vertex.h:174-184) and a
        # partially fixed goal state (PartiallyFixedVectorVertex, setXfFixed)
        ("unicycle_n12_patterns", dict(scenario="unicycle", N=12, iters=5, xlb="-inf,-1.5,-inf", xub="1.2,inf,inf", ulb="-0.8,-inf", uub="inf,0.6",
                                       xf_fixed=5), (1, 2, 3, 4, 5)),
        ("vdp_patterns", dict(scenario="vdp", iters=5, xlb="-0.5,-inf", xub="inf,inf", ulb="-inf", uub="0.7", xf_fixed=2, final_cost=0), (1, 2, 3, 4, 5)),
        # SerialIntegratorSystem of order 3 (nx = 3, nu = 1): fixed grid, multiple shooting, and time-optimal with the free dt
        ("int3", dict(scenario="int3", iters=6), (1, 2, 3, 4, 5, 6)),
        ("int3_ms_rk4", dict(scenario="int3", grid="ms", N=16, iters=5), (1, 2, 3, 4, 5)),
        ("int3_time_optimal", dict(scenario="int3", vargrid=1, N=25, iters=8, w="100,100,100", solves=3), (1, 4, 8)),
        # the reference's other benchmark systems with nx <= 3 (nonlinear_benchmark_systems.h): Crank-Nicolson on the fixed grid, one
        # other scheme each, and multiple shooting with RK4 for the two with sin()
        ("duffing", dict(scenario="duffing", iters=6), (1, 2, 3, 4, 5, 6)),
        ("rocket", dict(scenario="rocket", iters=6), (1, 2, 3, 4, 5, 6)),
        ("pendulum", dict(scenario="pendulum", iters=6), (1, 2, 3, 4, 5, 6)),
        ("mpendulum", dict(scenario="mpendulum", iters=6), (1, 2, 3, 4, 5, 6)),
        ("toy", dict(scenario="toy", iters=6), (1, 2, 3, 4, 5, 6)),
        ("artstein", dict(scenario="artstein", iters=6), (1, 2, 3, 4, 5, 6)),
        ("duffing_midpoint", dict(scenario="duffing", collocation="midpoint", N=12, iters=4), (1, 2, 3, 4)),
        ("rocket_forward", dict(scenario="rocket", collocation="forward", N=12, iters=4), (1, 2, 3, 4)),
        ("toy_backward", dict(scenario="toy", collocation="backward", N=12, iters=4), (1, 2, 3, 4)),
        ("pendulum_ms_rk4", dict(scenario="pendulum", grid="ms", N=16, iters=5), (1, 2, 3, 4, 5)),
        ("mpendulum_ms_rk4", dict(scenario="mpendulum", grid="ms", N=16, iters=5), (1, 2, 3, 4, 5)),
        ("rocket_ms_rk4", dict(scenario="rocket", grid="ms", N=16, iters=5), (1, 2, 3, 4, 5)),
        ("artstein_ms_rk4", dict(scenario="artstein", grid="ms", N=16, iters=5), (1, 2, 3, 4, 5)),
        # CartPole (nx = 4): fixed grid, another scheme, multiple shooting, bound patterns with a partially fixed goal
        ("cartpole", dict(scenario="cartpole", iters=6), (1, 2, 3, 4, 5, 6)),
        ("cartpole_midpoint", dict(scenario="cartpole", collocation="midpoint", N=12, iters=4), (1, 2, 3, 4)),
        ("cartpole_ms_rk4", dict(scenario="cartpole", grid="ms", N=16, iters=5), (1, 2, 3, 4, 5)),
        ("cartpole_patterns", dict(scenario="cartpole", N=14, iters=4, xlb="-0.3,-inf,-inf,-1.0", xub="0.4,inf,0.8,inf", ulb="-2.0", uub="inf",
                                   xf_fixed=5), (1, 2, 3, 4)),
        ("cartpole_tball", dict(scenario="cartpole", N=14, iters=5, tball=0.01, tball_s="1,2,0.5,0.5"), (1, 2, 3, 4, 5)),
        ("cartpole_teq", dict(scenario="cartpole", N=14, iters=5, teq=1), (1, 2, 3, 4, 5)),
        # ParallelIntegratorSystem of dimension 2 / 3 (nx = nu): the families with as many controls as states
        ("par2", dict(scenario="par2", iters=5), (1, 2, 3, 4, 5)),
        ("par3", dict(scenario="par3", iters=5), (1, 2, 3, 4, 5)),
        ("par2_ms_rk4", dict(scenario="par2", grid="ms", N=12, iters=4), (1, 2, 3, 4)),
        ("par3_forward", dict(scenario="par3", collocation="forward", N=12, iters=4, xf_fixed=5), (1, 2, 3, 4)),
        # LinearStateSpaceModel f = A x + B u, one fixture per (nx, nu) block family (matrices drawn once, rounded to 3 digits)
        ("lin21", dict(scenario='lin', nx=2, nu=1, lin_a='-1.045,-0.366,0.595,-0.14700000000000002', lin_b='-0.218,-0.334', iters=4), (1, 2, 3, 4)),
        ("lin22", dict(scenario='lin', nx=2, nu=2, lin_a='-0.303,-0.627,0.346,0.384', lin_b='-0.504,0.898,0.334,-0.808', iters=4, collocation='midpoint'), (1, 2, 3, 4)),
        ("lin31", dict(scenario='lin', nx=3, nu=1, lin_a='-0.616,0.773,0.395,-0.347,-0.03199999999999997,-0.56,-0.837,-0.68,-0.8200000000000001', lin_b='-0.07,-0.467,0.632', iters=4, grid='ms', N=14), (1, 2, 3, 4)),
        ("lin32", dict(scenario='lin', nx=3, nu=2, lin_a='-1.113,-0.741,-0.817,0.197,0.20899999999999996,0.203,0.864,0.45,0.22099999999999997', lin_b='0.859,0.092,0.875,-0.01,-0.452,-0.096', iters=4), (1, 2, 3, 4)),
        ("lin33", dict(scenario='lin', nx=3, nu=3, lin_a='-0.16999999999999998,-0.338,0.807,-0.486,-0.8200000000000001,-0.482,-0.289,-0.99,-0.243', lin_b='-0.435,-0.864,0.234,-0.647,-0.391,-0.118,-0.7,-0.564,-0.051', iters=4, collocation='forward', N=14), (1, 2, 3, 4)),
        ("lin41", dict(scenario='lin', nx=4, nu=1, lin_a='-0.547,-0.49,-0.405,-0.442,-0.479,-0.534,-0.576,-0.009,-0.507,0.677,-1.1400000000000001,0.724,-0.643,0.501,0.222,-1.0819999999999999', lin_b='0.52,-0.501,-0.829,0.236', iters=4, N=16), (1, 2, 3, 4)),
        ("unicycle_n24_ball", dict(scenario="unicycle", N=24, iters=6, ball="1,0.5,0.25,0.35", tball=0.02, tball_s="1,1,0.1"), (1, 2, 3, 4, 5, 6)),
    ]:
        d = slim(run("dump", **kv), keep)
        with open(os.path.join(OUT, f"{name}.json"), "w") as f:
            json.dump(d, f, separators=(",", ":"))
        print(name, {k: d[k] for k in ("n", "m", "nnz")}, "chi2", d["after_iter"][-1]["chi2"])

    # known-answer cases of the reference's own LM solver test (test_levenberg_marquardt_sparse.cpp:71-371), re-run against the compiled
    # reference: pins LevenbergMarquardtSparse::solve on problems that are not OCPs
    d = json.loads(subprocess.check_output([DRIVER, "kat"]))
    d["first_iteration"] = json.loads(subprocess.check_output([DRIVER, "kat", "cap=1"]))["cases"]   # every phase capped at ONE LM iteration
    with open(os.path.join(OUT, "lm_known_answers.json"), "w") as f:
        json.dump(d, f, separators=(",", ":"))
    print("lm_known_answers", [(c["name"], [round(v, 9) for v in c["phases"][-1]["x_final"]]) for c in d["cases"]])

    # 8 seeded cfg-3 instances (SURVEY 8d seeds): final trajectory + chi2 only
    x0, xf = problems.unicycle_instances(8)
    inst = []
    for b in range(8):
        d = run("dump", scenario="unicycle", x0=vec(x0[b]), xf=vec(xf[b]), iters=10)
        last = d["after_iter"][-1]
        assert last["k"] == 10
        inst.append({"x0": d["x0"], "xf": d["xf"], "chi2": last["chi2"], "vertex": last["vertex"]})
    with open(os.path.join(OUT, "unicycle_seeded8.json"), "w") as f:
        json.dump({"scenario": "unicycle", "seed": 20260928, "iters": 10, "instances": inst}, f, separators=(",", ":"))
    print("unicycle_seeded8", [round(i["chi2"], 6) for i in inst])

    # moving-horizon sequences (the grid update either side of the solve): one OCP object, new_run every step,
    # measured state = x_1 of the previous solution + disturbance; with / without the grid's shifting warm start; iters=0 pins the
    # warm-started initial guess itself
    for name, kv in [
        ("mpc_unicycle_shift_init", dict(scenario="unicycle", N=30, steps=3, iters=0, shift=1)),
        ("mpc_unicycle_shift", dict(scenario="unicycle", N=30, steps=4, iters=5, shift=1)),
        ("mpc_unicycle_noshift", dict(scenario="unicycle", N=30, steps=3, iters=5, shift=0)),
        ("mpc_vdp_shift", dict(scenario="vdp", steps=4, iters=5, shift=1)),
        ("mpc_dint", dict(scenario="dint", steps=3, iters=5, shift=1)),   # variable grid: never shifts, x_f fixed components re-set
        ("mpc_quad_shift_init", dict(scenario="quad", N=10, steps=3, iters=0, iters0=4, shift=1)),   # MultipleShootingGrid (ShootingGridBase)
    ]:
        d = run("mpc", **kv)
        with open(os.path.join(OUT, f"{name}.json"), "w") as f:
            json.dump(d, f, separators=(",", ":"))
        print(name, [round(st["chi2"], 6) for st in d["steps"]])
    adapt()

    # closed loops with the reference's own SimulatedPlant: plant.output -> compute(new_run) -> plant.control, the
    # plant integrating the OCP's dynamics with explicit Euler (its default) or RK4, plus a deterministic state disturbance
    for name, kv in [
        ("loop_unicycle_rk4", dict(scenario="unicycle", N=30, steps=5, iters=5, shift=1, integrator="rk4")),
        ("loop_unicycle_euler_noshift", dict(scenario="unicycle", N=20, steps=4, iters=5, shift=0, integrator="euler")),
        ("loop_vdp_euler", dict(scenario="vdp", steps=5, iters=5, shift=1, integrator="euler")),
        ("loop_int3_rk4", dict(scenario="int3", steps=4, iters=5, shift=1, integrator="rk4", disturbance=0.002)),
        ("loop_pendulum_rk4", dict(scenario="pendulum", steps=4, iters=5, shift=1, integrator="rk4", disturbance=0.002)),
        ("loop_duffing_euler", dict(scenario="duffing", steps=4, iters=5, shift=1, integrator="euler", disturbance=0.002)),
        ("loop_cartpole_rk4", dict(scenario="cartpole", steps=4, iters=5, shift=1, integrator="rk4", disturbance=0.002)),
        ("loop_lin32_rk4", dict(scenario='lin', nx=3, nu=2, lin_a='-1.113,-0.741,-0.817,0.197,0.20899999999999996,0.203,0.864,0.45,0.22099999999999997', lin_b='0.859,0.092,0.875,-0.01,-0.452,-0.096', iters=4, steps=4, shift=1, integrator='rk4', disturbance=0.002)),
        ("loop_quad_rk4", dict(scenario="quad", N=10, steps=3, iters=4, shift=1, integrator="rk4", disturbance=0.002)),
    ]:
        d = run("loop", **kv)
        with open(os.path.join(OUT, f"{name}.json"), "w") as f:
            json.dump(d, f, separators=(",", ":"))
        print(name, [round(st["chi2"], 6) for st in d["steps"]])


if __name__ == "__main__":
    main()
