"""-m gpu: corbo_hip_set_instance_params -- a batch of controllers for plants of ONE class with DIFFERENT parameters (the `params` argument of
SURVEY 8b's set_instance_data sketch; in the reference: one SystemDynamicsInterface object per OCP, e.g. VanDerPolOscillator::setParameters).

The checker needs nothing new: instance b against an oracle problem built from the descriptor with instance b's parameters written into
dyn_params -- values / Jacobian (bit-level tolerances of the parity suite), LM solves, the operators of the exact-Hessian path, the
simulated plants."""
import copy

import numpy as np
import pytest

from control_box_rst_amd import capi, problems
from control_box_rst_amd.solver import BatchedLevenbergMarquardt, CorboHipError

pytestmark = pytest.mark.gpu


def _with_params(d, prm):
    e = copy.deepcopy(d)
    for i in range(8):
        e.dyn_params[i] = prm[i]
    return e


def _family(name):
    rng = np.random.default_rng({"vdp": 11, "pendulum_ms": 12, "duffing_midpoint": 13, "quad": 14}[name])
    if name == "vdp":
        d, w = problems.vdp_desc(N=20), problems.VDP_WEIGHTS
    elif name == "pendulum_ms":
        d, w = problems.benchmark_desc("pendulum", N=16), problems.BENCHMARK_WEIGHTS
        d.grid, d.defect = capi.GRID_MS, capi.DEFECT_RK4_SHOOTING
    elif name == "duffing_midpoint":
        d, w = problems.benchmark_desc("duffing", N=18, defect=capi.DEFECT_MIDPOINT), problems.BENCHMARK_WEIGHTS
    elif name == "quad":
        d, w = problems.quad_desc(N=10, dt=0.05), problems.QUAD_WEIGHTS
    else:
        raise KeyError(name)
    B = 4
    base = np.array([d.dyn_params[i] for i in range(8)])
    prm = np.tile(base, (B, 1))
    used = base != 0.0
    prm[:, used] *= rng.uniform(0.7, 1.4, (B, int(used.sum())))     # every parameter the family uses, scaled per instance
    if name == "pendulum_ms":
        prm[:, 3] = rng.uniform(0.0, 0.1, B)                         # friction (0 by default)
    x0 = rng.uniform(-0.5, 0.5, (B, d.nx))
    xf = rng.uniform(-0.3, 0.3, (B, d.nx))
    if name == "quad":
        x0[:, 3:] *= 0.2
        xf[:, 3:] = 0.0
    return d, w, prm, x0, xf


@pytest.mark.parametrize("name", ["vdp", "pendulum_ms", "duffing_midpoint", "quad"])
def test_values_jacobian_and_solve_per_instance(oracle_mod, name):
    d, w, prm, x0, xf = _family(name)
    B = len(prm)
    s = BatchedLevenbergMarquardt(d, B)
    s.setIterations(4)
    s.setPenaltyWeights(*w)
    rng = np.random.default_rng(3)
    X0 = s.init_trajectory(x0, xf)
    X0 = X0 + 0.02 * rng.normal(size=X0.shape)
    X0[:, : d.nx] = x0
    s.set_instance_data(X0, xref=xf)
    s.set_instance_params(prm)
    values, jac = s.eval()
    shared = BatchedLevenbergMarquardt(d, B)          # the same data with the descriptor's parameters: must differ
    shared.set_instance_data(X0, xref=xf)
    v_shared, _ = shared.eval(*w)
    assert np.abs(values - v_shared).max() > 1e-4
    for b in range(B):
        p = oracle_mod.OracleProblem(_with_params(d, prm[b]))
        p.set_data(X0[b], xref=xf[b])
        vo, jo = p.eval(*w)
        assert np.abs(values[b] - vo).max() <= 1e-11 * max(1.0, np.abs(vo).max()), (name, b)
        assert np.abs(jac[b] - jo).max() <= 1e-6 * max(1.0, np.abs(jo).max()), (name, b)
    s.solve()
    X, chi2, status = s.get_solution()
    for b in range(B):
        Xo, co, _ = oracle_mod.solve_batch(_with_params(d, prm[b]), X0[b : b + 1], xf[b : b + 1], s.opts)
        if name == "quad":   # flat directions of the quadrotor problem: chi2 carries the comparison (tests/test_gpu_parity.py)
            assert abs(chi2[b] - co[0]) <= 1e-6 * abs(co[0]), (name, b)
            assert np.abs(X[b] - Xo[0]).max() <= 5e-3, (name, b)
        else:
            assert np.abs(X[b] - Xo[0]).max() <= 1e-5 * max(1.0, np.abs(Xo[0]).max()), (name, b, np.abs(X[b] - Xo[0]).max())
            assert abs(chi2[b] - co[0]) <= 2e-6 * max(1e-12, abs(co[0])), (name, b)
    # back to the descriptor's parameters
    s.set_instance_data(X0, xref=xf)
    s.set_instance_params(None)
    v2, _ = s.eval(*w)
    assert np.array_equal(v2, v_shared)


@pytest.mark.parametrize("name", ["vdp", "pendulum_ms", "quad"])
def test_hessian_path_operators_per_instance(oracle_mod, name):
    d, w, prm, x0, xf = _family(name)
    B = len(prm)
    s = BatchedLevenbergMarquardt(d, B)
    rng = np.random.default_rng(5)
    X0 = s.init_trajectory(x0, xf) + 0.05 * rng.normal(size=(B, s.dims.nv))
    X0[:, : d.nx] = x0
    s.set_instance_data(X0, xref=xf)
    s.set_instance_params(prm)
    mult_eq = rng.uniform(-1, 1, (B, s.dims.eq))
    mi = None if s.dims.ineq == 0 else rng.uniform(0, 1, (B, s.dims.ineq))
    vals = s.eval_hessians(True, 1.3, mult_eq, mi)
    grad, obj = s.objective_gradient()
    lin = s.linear_form()
    for b in range(B):
        p = oracle_mod.OracleProblem(_with_params(d, prm[b]))
        p.set_data(X0[b], xref=xf[b])
        go, oo = p.objective_gradient()
        assert np.abs(grad[b] - go).max() <= 1e-6 * max(1.0, np.abs(go).max()) and abs(obj[b] - oo) <= 1e-11 * max(1.0, abs(oo))
        lo = p.linear_form()
        assert np.abs(lin[2][b] - lo[2]).max() <= 1e-6 * max(1.0, np.abs(lo[2]).max()), (name, b)      # values of A
        fin = np.isfinite(lo[3])
        assert np.array_equal(np.isfinite(lin[3][b]), fin)
        assert np.abs(lin[3][b][fin] - lo[3][fin]).max() <= 1e-11 * max(1.0, np.abs(lo[3][fin]).max())   # lbA: -c_eq, lb - x
        ho = p.hessians(1, 1.3, mult_eq[b], None if mi is None else mi[b])
        for c in range(3):
            if len(ho[c][2]):
                assert np.abs(vals[c][b] - ho[c][2]).max() <= 2e-4 * max(1.0, np.abs(ho[c][2]).max()), (name, b, c)   # tests/test_gpu_hessian.py's bound
    assert np.abs(vals[1][0] - vals[1][1]).max() > 0.0   # the equality list carries the dynamics' second derivatives: they differ


def test_plants_follow_the_controllers_parameters(oracle_mod):
    d, w, prm, x0, xf = _family("vdp")
    B = len(prm)
    s = BatchedLevenbergMarquardt(d, B)
    s.set_instance_data(s.init_trajectory(x0, xf), xref=xf)
    X0, _, _ = s.get_solution()
    X0[:, d.nx] = np.linspace(-0.8, 0.9, B)       # a first control for the plants to apply
    s.set_instance_data(X0, xref=xf)
    s.set_instance_params(prm)
    s.plant_set_state(x0)
    s.plant_step(dt=0.1, integrator=capi.INTEGRATOR_RK4)
    xp = s.plant_get_state()
    for b in range(B):
        p = oracle_mod.OracleProblem(_with_params(d, prm[b]))
        p.set_data(X0[b], xref=xf[b])
        assert np.array_equal(xp[b], p.plant_step(x0[b], capi.INTEGRATOR_RK4, 0.1, None)), b


def test_refusals():
    A = np.array([[0.0, 1.0], [-1.0, -0.5]])
    s = BatchedLevenbergMarquardt(problems.linear_desc(A, np.array([[0.0], [1.0]]), N=10), 2)
    with pytest.raises(CorboHipError):
        s.set_instance_params(np.ones((2, 8)))
    t = BatchedLevenbergMarquardt(problems.vdp_desc(N=10), 2)
    bad = np.ones((2, 8))
    bad[1, 0] = np.nan
    with pytest.raises(CorboHipError):
        t.set_instance_params(bad)


def test_large_batch_through_the_instance_queue(oracle_mod):
    """More instances than resident workgroups: persistent workgroups pull instance after instance -- each with its own parameters."""
    d = problems.vdp_desc(N=20)
    B = 4500
    rng = np.random.default_rng(21)
    prm = np.zeros((B, 8))
    prm[:, 0] = rng.uniform(0.5, 1.8, B)
    x0 = rng.uniform(-1, 1, (B, 2))
    xf = np.zeros((B, 2))
    s = BatchedLevenbergMarquardt(d, B)
    s.setIterations(5)
    s.setPenaltyWeights(*problems.VDP_WEIGHTS)
    X0 = s.init_trajectory(x0, xf)
    s.set_instance_data(X0, xref=xf)
    s.set_instance_params(prm)
    s.solve()
    X, chi2, _ = s.get_solution()
    for b in (0, 1, 1023, 1024, 2047, 3000, B - 1):
        Xo, co, _ = oracle_mod.solve_batch(_with_params(d, prm[b]), X0[b : b + 1], xf[b : b + 1], s.opts)
        assert np.abs(X[b] - Xo[0]).max() <= 1e-5 and abs(chi2[b] - co[0]) <= 2e-6 * abs(co[0]), b
