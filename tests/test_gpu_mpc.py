"""GPU tests (-m gpu) of the moving-horizon grid update on the device (corbo_hip_warm_start) against the
oracle (bit for bit: copies, one extrapolation formula, comparisons) and against sequences of the genuine reference."""
import numpy as np
import pytest

from conftest import desc_for, load_golden
from control_box_rst_amd import problems
from control_box_rst_amd.solver import BatchedLevenbergMarquardt

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _built():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import __graft_entry__ as g
    g.build()


@pytest.mark.parametrize("scenario,N", [("unicycle", 100), ("unicycle", 30), ("vdp", 20), ("dint", 50), ("quad", 24), ("unicycle", 4),
                                        ("int3", 17), ("vdp", 3), ("unicycle", 256), ("quad", 5)])
def test_warm_start_bit_exact_vs_oracle(oracle_mod, scenario, N):
    """Random resident trajectories; the measured state of instance b is stored state (b mod 5) plus a small offset, so shifts of
    0 .. 4 samples (and the `same start` early exit) all occur in one batch; with and without shifting."""
    mk, w = problems.SCENARIOS[scenario]
    d = mk(N=N)
    B = 10
    rng = np.random.default_rng(99 + N)
    s = BatchedLevenbergMarquardt(d, B)
    nv, nx, st = s.dims.nv, d.nx, d.nx + d.nu
    X = rng.normal(size=(B, nv))
    if scenario == "dint":
        X[:, -1] = 0.1
    xref = rng.normal(size=(B, nx))
    x0 = np.empty((B, nx))
    for b in range(B):
        j = min(b % 5, max(N - 3, 0))
        x0[b] = X[b, j * st: j * st + nx] + (0.0 if b == 5 else 1e-3 * rng.normal(size=nx))   # b == 5: exactly the old start
    for shift in (True, False):
        s.set_instance_data(X, xref=xref)
        s.warm_start(x0, shift=shift)
        Xd, _, _ = s.get_solution()
        for b in range(B):
            p = oracle_mod.OracleProblem(d)
            p.set_data(X[b], xref=xref[b])
            p.warm_start(x0[b], shift=shift)
            assert np.array_equal(Xd[b], p.x()), (scenario, N, shift, b)
        if shift and scenario != "dint" and N > 6:
            assert not np.array_equal(Xd[1, st:2 * st], X[1, st:2 * st])   # something did move


@pytest.mark.parametrize("name", ["mpc_unicycle_shift_init", "mpc_unicycle_shift", "mpc_unicycle_noshift", "mpc_vdp_shift", "mpc_dint", "mpc_quad_shift_init", "mpc_pquad_shift_init"])
def test_sequence_vs_reference(name):
    g = load_golden(name)
    d = desc_for(g)
    s = BatchedLevenbergMarquardt(d, 1)
    s.setPenaltyWeights(*g["weights"])
    xf = np.array(g["xf"])[None, :]
    for k, st in enumerate(g["steps"]):
        if k == 0:
            s.setIterations(g["iters0"])
            s.set_instance_data(s.init_trajectory(st["x0"], g["xf"]), xref=xf)
        else:
            s.setIterations(g["iters"])
            s.warm_start(np.array(st["x0"])[None, :], shift=bool(g["shift"]))
        s.solve(new_run=True)
        x, chi2, _ = s.get_solution()
        ref = np.array(st["vertex"])[: s.dims.nv]
        tol = 3e-4 if "quad" in name else 1e-5   # quadrotor: nearly flat directions
        assert np.abs(x[0] - ref).max() <= tol, (name, k, np.abs(x[0] - ref).max())
        if (g["iters0"] if k == 0 else g["iters"]) > 0:
            assert abs(chi2[0] - st["chi2"]) <= 2e-6 * abs(st["chi2"]), (name, k)


def test_closed_loop_batch_vs_oracle(oracle_mod):
    """Four MPC steps of 6 seeded unicycle instances on the device (trajectories never leave HBM).  Every step is checked against
    the oracle started from the device's previous trajectories (single-step comparison: rounding-level differences -- device
    sin/cos vs the host's libm variant -- are amplified by every LM solve and would compound over the steps otherwise; the
    disturbance is large enough that the accept / reject decisions of each solve are not noise-driven)."""
    d = problems.unicycle_desc(N=40)
    B = 6
    x0, xf = problems.unicycle_instances(B, seed=31)
    s = BatchedLevenbergMarquardt(d, B)
    s.setIterations(4)
    s.setPenaltyWeights(*problems.UNICYCLE_WEIGHTS)
    X = s.init_trajectory(x0, xf)
    s.set_instance_data(X, xref=xf)
    st = d.nx + d.nu
    for step in range(4):
        Xprev = X
        if step > 0:
            meas = X[:, st: st + d.nx] + 0.05 * np.sin(step + np.arange(d.nx))[None, :]
            s.warm_start(meas, shift=True)
        s.solve(new_run=True)
        X, chi2, _ = s.get_solution()
        assert np.array_equal(s.get_first_control(), X[:, d.nx: d.nx + d.nu])   # getFirstControlInput
        for b in range(B):
            p = oracle_mod.OracleProblem(d)
            p.set_data(Xprev[b], xref=xf[b])
            if step > 0:
                p.warm_start(meas[b], shift=True)
            p.solve(s.opts, new_run=True)
            assert np.abs(X[b] - p.x()).max() <= 1e-5, (step, b, np.abs(X[b] - p.x()).max())
