"""GPU tests (-m gpu): integral-form constraint edges and the control-deviation term -- user stage
functions of the reference (oracle/ref_driver.cpp: UserStageInequalities, LinearIntegralEquality) as descriptor plug-ins: structure, residual
and Jacobian of the device against the reference goldens (tests/golden/xe_*.json) and the oracle, LM iterates at every iteration count."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import GOLDEN, desc_for, ledger_tolerances
from control_box_rst_amd import capi
from control_box_rst_amd.solver import BatchedLevenbergMarquardt, get_structure

pytestmark = pytest.mark.gpu
FIXTURES = sorted(f[:-5] for f in os.listdir(GOLDEN) if f.startswith("xe_") and f.endswith(".json"))


@pytest.fixture(scope="module", autouse=True)
def _built():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import __graft_entry__ as g
    g.build()


def _solver(g, d, iters, B=1, route=0):
    s = BatchedLevenbergMarquardt(d, B, route=route)
    s.setIterations(iters)
    s.setPenaltyWeights(*g["weights"])
    X0 = np.tile(np.array(g["vertex_init"])[: s.dims.nv], (B, 1))
    s.set_instance_data(X0, xref=np.tile(np.array(g["xf"]), (B, 1)))
    s.set_previous_control(g.get("u_prev"), g.get("u_prev_dt"))
    return s


@pytest.mark.parametrize("name", FIXTURES)
def test_structure_values_jacobian_vs_reference(name):
    g = json.load(open(os.path.join(GOLDEN, name + ".json")))
    d = desc_for(g)
    s = _solver(g, d, 1)
    for k in ("n", "lsq", "eq", "ineq", "bounds", "m", "nnz"):
        assert getattr(s.dims, k) == g[k], (name, k)
    rows, cols = get_structure(d)
    assert sorted(zip(rows.tolist(), cols.tolist())) == sorted(zip(g["jac_rows"], g["jac_cols"]))
    values, jac = s.eval()
    vr = np.array(g["values_init"])
    assert np.abs(values[0] - vr).max() <= 1e-12 * max(1.0, np.abs(vr).max()), name
    Jd = sp.coo_matrix((jac[0], (rows, cols)), shape=(s.dims.m, s.dims.n)).tocsr()
    Jr = sp.coo_matrix((g["jac_vals"], (g["jac_rows"], g["jac_cols"])), shape=(s.dims.m, s.dims.n)).tocsr()
    assert abs(Jd - Jr).max() <= 1e-6 * max(1.0, abs(Jr).max()), name


@pytest.mark.parametrize("name", FIXTURES)
def test_lm_iterates_vs_reference(name):
    g = json.load(open(os.path.join(GOLDEN, name + ".json")))
    d = desc_for(g)
    for a in g["after_iter"]:
        s = _solver(g, d, a["k"])
        for i in range(g["solves"]):
            s.solve(new_run=(i == 0))
        X, chi2, status = s.get_solution()
        ref = np.array(a["vertex"])[: s.dims.nv]
        xt, ct = ledger_tolerances(name)   # (the defaults 5e-6 / 2e-6 unless tests/tolerances.json lists the fixture: the big-block models' soft directions)
        assert np.abs(X[0] - ref).max() <= xt * max(1.0, np.abs(ref).max()), (name, a["k"], np.abs(X[0] - ref).max())
        assert abs(chi2[0] - a["chi2"]) <= ct * max(1.0, abs(a["chi2"])), (name, a["k"])


@pytest.mark.parametrize("seed", range(12))
def test_random_batches_vs_oracle(oracle_mod, seed):
    """Random combinations of the new edge kinds on the unicycle / Van der Pol / time-optimal integrator families, batches of perturbed starts."""
    _random_batch(oracle_mod, seed, 4, 40)


@pytest.mark.parametrize("seed", range(100, 106))
def test_random_batches_129_to_256_grid_points_vs_oracle(oracle_mod, seed):
    """The same on horizons of 129 .. 256 grid points: the BIG instantiation of the block-tridiagonal route (bt_factor_body<.., BIG>: one lane per stage,
    thirty-two list rounds), with and without a free dt."""
    assert random_batch_129_to_256(oracle_mod, seed) == capi.FACTOR_BLOCK_TRI


def random_batch_129_to_256(oracle_mod, seed):
    """-> the route the handle took (a structure whose product lists need more than thirty-two rounds per lane keeps the band route: tools/fuzz_campaign.py counts them)"""
    return _random_batch(oracle_mod, seed, 129, 257)


@pytest.mark.parametrize("seed", range(200, 208))
def test_random_batches_with_dense_weights_vs_oracle(oracle_mod, seed):
    """Non-diagonal Q / R / Qf TOGETHER with extra edges (band route: sweep_kernel<.., DENSE, .., XE>), short horizons and -- every other seed -- beyond 256 grid points."""
    lo, hi = ((257, 300) if seed % 2 else (4, 40))
    assert _random_batch(oracle_mod, seed, lo, hi, dense=True) == capi.FACTOR_BAND


def _random_batch(oracle_mod, seed, n_lo, n_hi, dense=False):
    from control_box_rst_amd import problems
    rng = np.random.default_rng(8800 + seed)
    fam = ["unicycle", "vdp", "int3t"][seed % 3]
    if dense and fam == "int3t":
        fam = "vdp"   # (the time-optimal family has no quadratic cost to make dense)
    N = int(rng.integers(n_lo, n_hi))
    d = {"unicycle": problems.unicycle_desc, "vdp": problems.vdp_desc}[fam](N=N) if fam != "int3t" else problems.int3_desc(N=N, dt=0.1, time_optimal=True)
    d.constraint_integration = int(rng.integers(1, 3))
    if rng.random() < 0.7:
        d.stage_eq = capi.STAGE_EQ_LINEAR
        for i in range(d.nx + d.nu + 1):
            d.stage_eq_params[i] = float(rng.uniform(-0.3, 0.3))
    if rng.random() < 0.7:
        d.ctrl_dev = capi.CTRL_DEV_RATE
        for i in range(d.nu):
            d.ctrl_dev_params[i] = float(rng.uniform(0.3, 3.0))
    if fam == "unicycle" and rng.random() < 0.7:
        d.stage_ineq, d.stage_ineq_integral = capi.INEQ_BALL, 1
        for i, v in enumerate((1.0, 0.5, 0.2, float(rng.uniform(0.2, 0.6)))):
            d.ineq_params[i] = v
    if not (d.stage_eq or d.ctrl_dev or d.stage_ineq_integral):
        d.ctrl_dev = capi.CTRL_DEV_RATE
        d.ctrl_dev_params[0] = 1.0
        d.ctrl_dev_params[1] = 1.0
    if seed >= 12 and rng.random() < 0.5:   # (campaign seeds -- tools/fuzz_campaign.py: a user control function, csrc/stage_functions/control_norm.hpp, on top)
        d.stage_ineq_control = capi.STAGE_FN_USER + 1
        d.ineq_control_params[0] = float(rng.uniform(0.3, 1.5))
    if dense:   # random symmetric positive definite weights, handed over as upper Cholesky factors (U^T U = the weight)
        rd = np.random.default_rng(99000 + seed)
        d.weights_dense = 1 | (2 if d.nu > 1 else 0) | (4 if d.final_cost else 0)
        for dst, n in ((d.q_sqrt, d.nx), (d.r_sqrt, d.nu), (d.qf_sqrt, d.nx)):
            a = rd.uniform(-1, 1, (n, n))
            for i, v in enumerate(np.linalg.cholesky(a.T @ a + 0.5 * np.eye(n)).T.ravel()):
                dst[i] = float(v)
    B = 3
    x0 = rng.uniform(-1, 1, (B, d.nx))
    xf = rng.uniform(-0.5, 0.5, (B, d.nx)) + (np.array([1.5, 0.5, 0.2])[: d.nx] if fam != "int3t" else 0.0)
    if fam == "int3t":
        xf = np.tile([1.0, 0.0, 0.0], (B, 1))
    s = BatchedLevenbergMarquardt(d, B)
    s.setIterations(3)
    w = tuple(float(v) for v in rng.uniform(2.0, 30.0, 3))
    s.setPenaltyWeights(*w)
    X0 = s.init_trajectory(x0, xf) + 0.02 * rng.normal(size=(B, s.dims.nv))
    X0[:, : d.nx] = x0
    if fam == "int3t":
        X0[:, -1] = d.dt_ref
        X0[:, (N - 1) * (d.nx + d.nu): (N - 1) * (d.nx + d.nu) + d.nx] = xf
    up = rng.uniform(-0.3, 0.3, (B, d.nu))
    dtp = rng.uniform(0.05, 0.2, B)
    s.set_instance_data(X0, xref=xf)
    s.set_previous_control(up, dtp)
    po = oracle_mod.OracleProblem(d)
    rows, cols = get_structure(d)
    ro, co = po.structure()
    assert np.array_equal(rows, ro) and np.array_equal(cols, co)
    assert s.dims.as_dict() == po.dims.as_dict()
    values, jac = s.eval()
    Xo = np.zeros_like(X0)
    chi2o = np.zeros(B)
    for b in range(B):
        p = oracle_mod.OracleProblem(d)
        p.set_data(X0[b], xref=xf[b])
        p.set_previous_control(up[b], dtp[b])
        vo, jo = p.eval(*w)
        assert np.abs(values[b] - vo).max() <= 1e-11 * max(1.0, np.abs(vo).max()), (seed, b)
        assert np.abs(jac[b] - jo).max() <= 1e-6 * max(1.0, np.abs(jo).max()), (seed, b)
        _, chi2o[b], _ = p.solve(s.opts, new_run=True)
        Xo[b] = p.x()
    s.solve()
    X, chi2, _ = s.get_solution()
    assert np.abs(X - Xo).max() <= 3e-5 * max(1.0, np.abs(Xo).max()), (seed, fam, np.abs(X - Xo).max())
    assert np.allclose(chi2, chi2o, rtol=5e-5, atol=1e-10), (seed, chi2, chi2o)
    return s.factor_route()


@pytest.mark.parametrize("name", ["xe_unicycle_all_left_n100", "xe_unicycle_rate", "xe_vdp_eqlin_rate", "xe_int3_vargrid_trap", "xe_rocket_rate_eq", "xe_unicycle_all_n300"])
def test_narrow_band_kernel_vs_eight_wave_kernel(name):
    """Half-bandwidths up to 7 take band_narrow_kernel (one wave per instance, the 8 x 8 window in registers; incl. a free dt as a border); option band_wide
    keeps band_factor_kernel.  Same products in the same order per entry of the factor: the LM decisions are the same, the iterates agree far inside the fixtures' tolerances."""
    g = json.load(open(os.path.join(GOLDEN, name + ".json")))
    d = desc_for(g)
    a = g["after_iter"][-1]
    out = []
    for wide in (0, 1):
        s = _solver(g, d, a["k"], B=3, route=capi.ROUTE_XE_BAND)   # (the band route: these handles take the block-tridiagonal route by default since round 6)
        s.set_option("band_wide", wide)
        for i in range(g["solves"]):
            s.solve(new_run=(i == 0))
        x, chi2, status = s.get_solution()
        out.append((x, chi2, status, s.get_stats()))
        assert s.factor_route() == capi.FACTOR_BAND, name
    (x0, c0, s0, t0), (x1, c1, s1, t1) = out
    assert np.array_equal(s0, s1)
    assert t0["factorizations"] == t1["factorizations"] and t0["accepted_steps"] == t1["accepted_steps"]
    # (|y|^2 is summed in another order: the damping after an accepted step differs in its last bits, the finite-difference Jacobians amplify that like any
    #  other rounding difference -- both kernels stay inside the reference's own one-ulp spread of these fixtures, tests/tolerances.json)
    assert np.allclose(c0, c1, rtol=2e-7, atol=1e-12), (name, c0, c1)
    assert np.abs(x0 - x1).max() <= 2e-6, (name, np.abs(x0 - x1).max())


SMALL_BLOCK = [n for n in FIXTURES if "quad" not in n and "n300" not in n and "fullq" not in n]   # (the big-block family and horizons beyond 256 grid points keep the band route; n200: the BIG instantiation)


@pytest.mark.parametrize("name", SMALL_BLOCK)
def test_block_tridiagonal_route_vs_band_route(name):
    """Round 6: the small-block families with extra edges run to completion in ONE launch (lm_bt_kernel: block cyclic reduction on (x_k, u_k) blocks, a free dt
    as a border); corbo_hip_create_routed(.., CORBO_HIP_ROUTE_XE_BAND) keeps the band route (host-launched passes).  Same LM decisions, iterates far inside the
    fixtures' tolerances of each other -- and both inside them against the reference (test_lm_iterates_vs_reference runs the default route)."""
    g = json.load(open(os.path.join(GOLDEN, name + ".json")))
    d = desc_for(g)
    a = g["after_iter"][-1]
    out = []
    for route in (0, capi.ROUTE_XE_BAND):
        s = _solver(g, d, a["k"], B=3, route=route)
        for i in range(g["solves"]):
            s.solve(new_run=(i == 0))
        x, chi2, status = s.get_solution()
        out.append((x, chi2, status, s.get_stats()))
        assert s.factor_route() == (capi.FACTOR_BAND if route else capi.FACTOR_BLOCK_TRI), (name, route)
    (x0, c0, s0, t0), (x1, c1, s1, t1) = out
    assert np.array_equal(s0, s1)
    for k in ("factorizations", "accepted_steps", "rejected_steps", "jacobian_sweeps"):
        assert t0[k] == t1[k], (name, k)
    assert np.allclose(c0, c1, rtol=5e-7, atol=1e-12), (name, c0, c1)
    assert np.abs(x0 - x1).max() <= 2e-6 * max(1.0, np.abs(x1).max()), (name, np.abs(x0 - x1).max())
    ref = np.array(a["vertex"])[: x1.shape[1]]
    xt, _ = ledger_tolerances(name)
    assert np.abs(x1[0] - ref).max() <= xt * max(1.0, np.abs(ref).max()), name   # (the band route against the reference as well)


def test_block_tridiagonal_route_batch_async_and_per_pass_mode():
    """The headline structure with a rate limit on the controls: a batch through the run-to-completion kernel (synchronous, enqueued, re-armed) and through the
    per-pass mode of the same handle (option run_to_completion = 0: band kernels) -- identical decisions, iterates within rounding-level amplification."""
    import bench
    w = bench.workload(3, 96)
    d = w["desc"]
    d.ctrl_dev = capi.CTRL_DEV_RATE
    d.ctrl_dev_params[0] = 1.0
    d.ctrl_dev_params[1] = 1.0
    s = BatchedLevenbergMarquardt(d, 96)
    s.setPenaltyWeights(*w["weights"])
    s.set_instance_data(s.init_trajectory(w["x0"], w["xf"]), xref=w["xf"])
    s.solve()
    X, chi2, status = [a.copy() for a in s.get_solution()]
    st = s.get_stats()
    assert (status <= 1).all() and st["lm_iterations"] == 96 * 10
    s.set_result_sink(True)
    for _ in range(2):
        s.solve_async(rearm=True)
    s.synchronize()
    Xa, ca, sa = s.fetch_solution()
    assert np.array_equal(np.asarray(Xa)[:, : X.shape[1]], X) and np.array_equal(ca, chi2) and np.array_equal(sa, status)
    s.set_result_sink(False)
    s.set_option("run_to_completion", 0)
    s.restore_instance_data()
    s.solve()
    Xp, cp, sp_ = s.get_solution()
    stp = s.get_stats()
    assert np.array_equal(sp_, status)
    assert stp["factorizations"] == st["factorizations"] and stp["accepted_steps"] == st["accepted_steps"]
    # (two elimination orders of the same H: rounding-level differences amplified by ten iterations of finite-difference Jacobians -- measured 1.2e-6 / 1.8e-6,
    #  inside the default tolerances of the reference fixtures, 2e-6 relative on chi2 and 5e-6 on the iterates)
    assert np.allclose(cp, chi2, rtol=2e-6) and np.abs(Xp - X).max() <= 5e-6


def test_block_tridiagonal_route_two_and_three_workgroups_per_cu():
    """lm_bt_kernel's two instantiations (option bt_waves: 256 VGPRs / two workgroups per CU, 168 VGPRs / three; launch_bt_t picks by the rounds a batch needs): the
    same source, the same arithmetic -- bit-identical results; and the automatic choice is one of them."""
    import bench
    B = 160
    w = bench.workload(3, B)
    d = w["desc"]
    d.ctrl_dev = capi.CTRL_DEV_RATE
    d.ctrl_dev_params[0] = 1.0
    d.ctrl_dev_params[1] = 0.8
    out = []
    for waves in (0, 2, 3):
        s = BatchedLevenbergMarquardt(d, B)
        s.set_option("bt_waves", waves)
        s.setPenaltyWeights(*w["weights"])
        s.set_instance_data(s.init_trajectory(w["x0"], w["xf"]), xref=w["xf"])
        s.solve()
        out.append([a.copy() for a in s.get_solution()])
    for X, chi2, status in out[1:]:
        assert np.array_equal(X, out[0][0]) and np.array_equal(chi2, out[0][1]) and np.array_equal(status, out[0][2])


def test_block_tridiagonal_route_instance_queue_bit_equal_to_slices():
    """More instances than the chip holds workgroups of lm_bt_kernel (1200 > 4 x 256: the instance queue -- a workgroup pulls instance after instance, its
    LDS and the instance's row of the re-used blocks in HBM change hands): every instance bit for bit what it is in a batch of 600 that is resident at once."""
    import bench
    B = 1200
    w = bench.workload(3, B)
    d = w["desc"]
    d.ctrl_dev = capi.CTRL_DEV_RATE
    d.ctrl_dev_params[0] = 0.9
    d.ctrl_dev_params[1] = 0.7

    def run(lo, hi):
        s = BatchedLevenbergMarquardt(d, hi - lo)
        assert s.factor_route() == capi.FACTOR_BLOCK_TRI
        s.setPenaltyWeights(*w["weights"])
        s.set_instance_data(s.init_trajectory(w["x0"][lo:hi], w["xf"][lo:hi]), xref=w["xf"][lo:hi])
        s.solve()
        X, chi2, status = [a.copy() for a in s.get_solution()]
        return X, chi2, status, s.get_stats()

    X, chi2, status, st = run(0, B)
    assert st["rejected_steps"] > B   # (the rate-limited problem rejects every other step: the re-used blocks are exercised)
    for lo, hi in ((0, 600), (600, 1200)):
        Xs, cs, ss, _ = run(lo, hi)
        assert np.array_equal(Xs, X[lo:hi]) and np.array_equal(cs, chi2[lo:hi]) and np.array_equal(ss, status[lo:hi]), (lo, hi)


def test_factor_route_of_the_families():
    """corbo_hip_factor_route: the headline structure -> stage-parallel cyclic reduction; the quadrotor -> stage + chain kernels; a rate limit on top -> the
    block-tridiagonal route resp. (big-block family, horizons beyond 256 grid points) the band route."""
    from control_box_rst_amd import problems
    d = problems.unicycle_desc(N=100)
    assert BatchedLevenbergMarquardt(d, 2).factor_route() == capi.FACTOR_STAGE_CR
    for N, want in ((100, capi.FACTOR_BLOCK_TRI), (128, capi.FACTOR_BLOCK_TRI), (129, capi.FACTOR_BLOCK_TRI), (256, capi.FACTOR_BLOCK_TRI), (257, capi.FACTOR_BAND)):
        d = problems.unicycle_desc(N=N)
        d.ctrl_dev = capi.CTRL_DEV_RATE
        d.ctrl_dev_params[0] = d.ctrl_dev_params[1] = 1.0
        assert BatchedLevenbergMarquardt(d, 2).factor_route() == want, N
        assert BatchedLevenbergMarquardt(d, 2, route=capi.ROUTE_XE_BAND).factor_route() == capi.FACTOR_BAND, N
    q = problems.quad_desc(N=20)
    assert BatchedLevenbergMarquardt(q, 2).factor_route() == capi.FACTOR_STAGE_CHAIN
    q.ctrl_dev = capi.CTRL_DEV_RATE
    for i in range(q.nu):
        q.ctrl_dev_params[i] = 5.0
    assert BatchedLevenbergMarquardt(q, 2).factor_route() == capi.FACTOR_BAND
