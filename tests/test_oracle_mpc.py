"""CPU tests of the moving-horizon grid update: the oracle's restatement of
FullDiscretizationGridBase::update / warmStartShifting against sequences produced by the genuine reference
(tests/golden/mpc_*.json, generator oracle/gen_golden.py -> oracle/_ref/ref_driver mpc)."""
import numpy as np
import pytest

from conftest import desc_for, load_golden
from control_box_rst_amd import capi

MPC = ["mpc_unicycle_shift_init", "mpc_unicycle_shift", "mpc_unicycle_noshift", "mpc_vdp_shift", "mpc_dint", "mpc_quad_shift_init", "mpc_pquad_shift_init"]


@pytest.mark.parametrize("name", MPC)
def test_warm_start_alone_is_exact(oracle_mod, name):
    """The grid update applied to the reference's own previous solution reproduces, bit for bit, the start point the reference
    solved from: with iters = 0 that is the dumped vertex vector itself; otherwise x_0 and every vertex the solver cannot move."""
    g = load_golden(name)
    d = desc_for(g)
    nv = None
    for s in range(1, len(g["steps"])):
        prev, cur = g["steps"][s - 1], g["steps"][s]
        p = oracle_mod.OracleProblem(d)
        nv = p.dims.nv
        p.set_data(np.array(prev["vertex"])[:nv], xref=np.array(g["xf"]))
        p.warm_start(cur["x0"], shift=bool(g["shift"]))
        x = p.x()
        assert np.array_equal(x[: d.nx], np.array(cur["x0"]))
        if g["iters"] == 0:
            # the reference's compute() still evaluates the Jacobian once (in-place central differences drift the vertices by
            # <= 1 ulp); the oracle's 0-iteration solve does the same sweep, after which the match is bit for bit
            assert np.abs(x - np.array(cur["vertex"])[:nv]).max() <= 1e-15, (name, s)
            p.solve(capi.default_lm_opts(0, *g["weights"]), new_run=True)
            assert np.array_equal(p.x(), np.array(cur["vertex"])[:nv]), (name, s)


@pytest.mark.parametrize("name", MPC)
def test_sequence_vs_reference(oracle_mod, name):
    g = load_golden(name)
    d = desc_for(g)
    w = g["weights"]
    p = oracle_mod.OracleProblem(d)
    nv = p.dims.nv
    for s, st in enumerate(g["steps"]):
        if s == 0:
            p.set_data(p.init_trajectory(st["x0"], g["xf"]), xref=np.array(g["xf"]))
            iters = g["iters0"]
        else:
            p.warm_start(st["x0"], shift=bool(g["shift"]))
            iters = g["iters"]
        status, chi2, _ = p.solve(capi.default_lm_opts(iters, *w), new_run=True)
        ref = np.array(st["vertex"])[:nv]
        tol = 3e-4 if "quad" in name else 5e-6   # quadrotor: nearly flat directions, see tests/test_oracle_golden.py
        assert np.abs(p.x() - ref).max() <= tol, (name, s, np.abs(p.x() - ref).max())
        if iters > 0:
            assert abs(chi2 - st["chi2"]) <= 2e-6 * abs(st["chi2"]), (name, s)
