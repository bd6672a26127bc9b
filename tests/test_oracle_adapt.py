"""CPU tests of the time-optimal grid adaptation (the remainder r1 left): the oracle's restatement of
FullDiscretizationGridBase::resampleTrajectory (full_discretization_grid_base.cpp:397-474) and of the adaptation rules of
FiniteDifferencesVariableGrid (finite_differences_variable_grid.cpp:101-163) against moving-horizon sequences of the genuine reference
(tests/golden/mpc_dint_adapt_*.json, generator oracle/gen_golden.py adapt -> oracle/_ref/ref_driver mpc adapt=...)."""
import numpy as np
import pytest

from conftest import load_golden
from control_box_rst_amd import adaptive_grid, capi, problems

STRATEGY = {"single": 1, "aggressive": 2, "shrink": 3}
# *_ms_*: the same on the MultipleShootingVariableGrid (multiple_shooting_variable_grid.cpp:58-152, shooting_grid_base.cpp:473-547)
INIT = ["mpc_dint_adapt_single_grow_init", "mpc_dint_adapt_single_shrink_init", "mpc_dint_adapt_aggressive_init", "mpc_dint_adapt_shrink_init",
        "mpc_dint_ms_adapt_single_init", "mpc_dint_ms_adapt_shrink_init"]
FULL = ["mpc_dint_adapt_cross256_aggressive", "mpc_dint_adapt_cross256_single", "mpc_dint_adapt_single", "mpc_dint_adapt_aggressive", "mpc_dint_ms_adapt_single", "mpc_dint_ms_adapt_aggressive",
        "mpc_dint_ms_adapt_aggressive_collapse"]


def _strategy(g):
    s = STRATEGY[g["adapt"]]
    return adaptive_grid.AGGRESSIVE_SHOOTING if (s == 2 and g.get("grid") == "ms") else s


def _desc(g, n):
    return problems.dint_desc(N=n, dt=g["dt"], shooting=(g.get("grid") == "ms"))


def _n_of(x, nx=2, nu=1):
    return (len(x) - nx - 1) // (nx + nu) + 1


def _solve(oracle_mod, g, x, iters, new_run, carry):
    """One compute() of the oracle on a vertex vector; carry = the OracleProblem of the previous call when N did not change (the penalty
    weights adapt across calls only through that object -- factor 1 in these fixtures, so a fresh object is equivalent)."""
    d = _desc(g, _n_of(x))
    p = oracle_mod.OracleProblem(d)
    p.set_data(x, xref=np.array(g["xf"]))
    # (a fresh object has no adapted weights to continue from: with adaptation factor 1, as in these fixtures, stating the weights anew
    # is what LevenbergMarquardtSparse's persistent _weight_* amount to)
    _, chi2, _ = p.solve(capi.default_lm_opts(iters, *g["weights"]), new_run=True)
    return p.x(), chi2


@pytest.mark.parametrize("name", INIT)
def test_resampling_chain_is_bit_exact(oracle_mod, name):
    """iters = 0 after the first step: every dumped vertex vector is the previous one after [x_0 overwrite, then per compute() the
    adaptation decision + resampleTrajectory + the vertex drift of one in-place Jacobian sweep] -- reproduced bit for bit, including
    the sequence of N."""
    g = load_golden(name)
    strat = _strategy(g)
    prev = None
    for st in g["steps"]:
        v = np.array(st["vertex"])
        if prev is not None:
            x = prev.copy()
            x[:2] = st["x0"]
            x[-3:-1] = g["xf"]
            n_seq = []
            for it in range(g["ocp_iters"]):
                if it > 0 or g["adapt_first"]:
                    n = _n_of(x)
                    n_new = oracle_mod.adapt_grid_n(strat, n, x[-1], g["dt"], g["hyst"], g["nmin"], g["nmax"])
                    assert n_new == adaptive_grid.adapt_grid_n(strat, n, x[-1], g["dt"], g["hyst"], g["nmin"], g["nmax"])   # the host-side mirror
                    x = oracle_mod.resample_trajectory(2, 1, x, n_new)
                x, _ = _solve(oracle_mod, g, x, 0, it == 0, None)
                n_seq.append(_n_of(x))
            assert n_seq == st["n_seq"], (name, n_seq, st["n_seq"])
            assert len(x) == len(v) and np.array_equal(x, v), (name, np.abs(x - v).max() if len(x) == len(v) else None)
        prev = v
    sizes = [st["n"] for st in g["steps"]]
    assert len(set(sizes)) > 1 or g["adapt"] == "aggressive"   # the fixture does exercise a change of resolution


@pytest.mark.parametrize("name", FULL)
def test_adaptive_controller_vs_reference(oracle_mod, name):
    """The controller as it runs (5 LM iterations per compute(), 3 compute() calls per step): same sequence of grid sizes, trajectories to
    the parity tolerance."""
    g = load_golden(name)
    strat = _strategy(g)
    x = None
    first = True
    for s, st in enumerate(g["steps"]):
        n_seq = []
        for it in range(g["ocp_iters"]):
            new_run = (it == 0)
            if x is None:
                d = _desc(g, g["N"])
                x = oracle_mod.OracleProblem(d).init_trajectory(st["x0"], g["xf"])
            if not first and (not new_run or g["adapt_first"]):
                n = _n_of(x)
                x = oracle_mod.resample_trajectory(2, 1, x, oracle_mod.adapt_grid_n(strat, n, x[-1], g["dt"], g["hyst"], g["nmin"], g["nmax"]))
            if new_run and not first:
                x[:2] = st["x0"]
                x[-3:-1] = g["xf"]
            x, chi2 = _solve(oracle_mod, g, x, g["iters0"] if s == 0 else g["iters"], new_run, None)
            first = False
            n_seq.append(_n_of(x))
        assert n_seq == st["n_seq"], (name, s, n_seq, st["n_seq"])
        ref = np.array(st["vertex"])
        assert np.abs(x - ref).max() <= 5e-6, (name, s, np.abs(x - ref).max())
        assert abs(chi2 - st["chi2"]) <= 2e-6 * max(1.0, abs(st["chi2"])), (name, s, chi2, st["chi2"])
