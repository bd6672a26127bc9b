"""corbo_hip_solve_async (-m gpu): the solve without the wait.  Enqueued back to back on one handle the solves give what the synchronous calls give;
timing and the pass-limit check of the enqueued solves arrive with the next synchronising call; handles whose passes are driven from the host solve
synchronously inside the same entry point."""
import numpy as np
import pytest

from control_box_rst_amd import problems
from control_box_rst_amd.solver import BatchedLevenbergMarquardt, CorboHipError

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _built():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import __graft_entry__ as g
    g.build()


def _solver(B=48, N=40):
    d = problems.unicycle_desc(N=N)
    x0, xf = problems.unicycle_instances(B)
    s = BatchedLevenbergMarquardt(d, B)
    s.setPenaltyWeights(*problems.UNICYCLE_WEIGHTS)
    s.set_instance_data(s.init_trajectory(x0, xf), xref=xf)
    return s


@pytest.mark.parametrize("sink", [False, True])
def test_enqueued_solves_match_synchronous_ones(sink):
    s = _solver()
    s.set_result_sink(sink)
    s.solve()
    X, chi2, status = s.get_solution()
    st = s.get_stats()
    s.get_timing(reset=True)
    for _ in range(5):                       # five steps enqueued back to back, one wait
        s.restore_instance_data()
        s.solve_async()
    s.synchronize()
    ms, n = s.get_timing(reset=True)
    assert n == 5 and ms > 0                 # every enqueued solve was timed by its own HIP event pair
    Xv, cv, sv = s.fetch_solution()
    X2, chi22, status2 = s.get_solution()
    assert np.array_equal(X, X2) and np.array_equal(chi2, chi22) and np.array_equal(status, status2)
    assert np.array_equal(np.asarray(cv), chi2)
    st2 = s.get_stats()
    for k in ("lm_iterations", "accepted_steps", "rejected_steps", "factorizations"):
        assert st[k] == st2[k], k
    s.restore_instance_data()
    s.solve_async()                          # a synchronous call behind an enqueued one waits for it first
    s.restore_instance_data()
    s.solve()
    X3, _, _ = s.get_solution()
    assert np.array_equal(X, X3)


def test_pass_limit_of_an_enqueued_solve_is_reported_by_the_next_wait():
    s = _solver()
    s.set_option("pass_limit", 3)
    s.solve_async()                          # returns at once: nothing has been checked yet
    with pytest.raises(CorboHipError, match="pass limit"):
        s.synchronize()
    s.set_option("pass_limit", 0)
    s.restore_instance_data()
    s.solve()                                # the handle is usable again
    assert (s.get_solution()[2] <= 1).all()


def test_mutator_behind_a_failed_enqueued_solve_still_mutates_and_the_error_stays_pending():
    """set_instance_data / restore between a failing solve_async and the next wait do their work and return OK; the enqueued solve's
    'pass limit' error is reported by the next result / solve call, once."""
    s = _solver()
    x0, xf = problems.unicycle_instances(48, seed=11)
    X_new = s.init_trajectory(x0, xf)
    s.set_option("pass_limit", 3)
    s.solve_async()
    s.set_instance_data(X_new, xref=xf)      # drains the failing solve; must upload and must not raise
    with pytest.raises(CorboHipError, match="pass limit"):
        s.synchronize()
    s.synchronize()                          # reported once
    X, _, _ = s.get_solution()
    assert np.array_equal(X, X_new[:, : s.dims.nv]), "the mutator's upload was skipped"
    s.set_option("pass_limit", 0)
    s.solve()
    assert (s.get_solution()[2] <= 1).all()


def test_host_driven_handles_solve_synchronously():
    d = problems.quad_desc(N=24)
    x0, xf = problems.quad_instances(4)
    s = BatchedLevenbergMarquardt(d, 4)
    s.setPenaltyWeights(*problems.QUAD_WEIGHTS)
    s.set_instance_data(s.init_trajectory(x0, xf), xref=xf)
    s.solve()
    X, chi2, _ = s.get_solution()
    s.restore_instance_data()
    s.solve_async()                          # big-block family: passes driven from the host -> a plain solve
    X2, chi22, _ = s.get_solution()
    assert np.array_equal(X, X2) and np.array_equal(chi2, chi22)


@pytest.mark.parametrize("cfg", [3, 5])
def test_rearmed_solve_equals_restore_then_solve(cfg):
    """new_run = 2 (solve(rearm=True)): restore_instance_data() + solve(new_run=True) in one call -- the run-to-completion kernel reads its start from the
    shadow copy itself (cfg 3), handles with host-launched passes copy first (cfg 5, reduced).  Bit-identical results and statistics, also back to back."""
    import bench
    B = 64 if cfg == 3 else 6
    w = bench.workload(cfg, B)
    d = w["desc"]
    if cfg == 5:
        from control_box_rst_amd import problems
        d = problems.quad_desc(N=24)
    s = BatchedLevenbergMarquardt(d, B)
    s.setIterations(6)
    s.setPenaltyWeights(*w["weights"])
    s.set_instance_data(s.init_trajectory(w["x0"], w["xf"]), xref=w["xf"])
    s.solve()
    ref = [a.copy() for a in s.get_solution()]
    ref_stats = {k: v for k, v in s.get_stats().items() if not k.endswith("_ms")}
    for rep in range(2):
        s.solve(rearm=True)            # (the iterate is the previous solution at this point: the start must come from the shadow copy)
        got = s.get_solution()
        assert all(np.array_equal(a, b) for a, b in zip(got, ref)), rep
        assert {k: v for k, v in s.get_stats().items() if not k.endswith("_ms")} == ref_stats
    s.set_result_sink(True)
    for _ in range(3):
        s.solve_async(rearm=True)
    s.synchronize()
    X, chi2, status = s.fetch_solution()
    assert np.array_equal(np.asarray(X), ref[0]) and np.array_equal(chi2, ref[1]) and np.array_equal(status, ref[2])


@pytest.mark.parametrize("mutator", ["warm_start", "set_instance_data", "restore"])
def test_mutator_behind_an_enqueued_solve_does_not_leave_stale_sink_views(mutator):
    """solve_async(); <mutator>; fetch_solution() must hand out what the synchronous sequence hands out (the device iterates AFTER the
    mutator), not the pinned rows the enqueued solve wrote before it."""
    def run(async_):
        s = _solver()
        s.set_result_sink(True)
        (s.solve_async if async_ else s.solve)()
        if mutator == "warm_start":
            x0, _ = problems.unicycle_instances(48, seed=7)
            s.warm_start(x0, shift=False)
        elif mutator == "set_instance_data":
            x0, xf = problems.unicycle_instances(48, seed=11)
            s.set_instance_data(s.init_trajectory(x0, xf), xref=xf)
        else:
            s.restore_instance_data()
        Xv, _, _ = s.fetch_solution()
        return np.array(Xv, copy=True)
    assert np.array_equal(run(True), run(False))
