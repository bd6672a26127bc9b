"""Result sink of the handles whose passes are launched from the host (big-block family): every solve's trajectories / chi2 / status are delivered
into pinned host memory by a copy on a second stream behind a device-side snapshot (corbo_hip.hip, deliver_results).  What corbo_hip_fetch_solution
returns must be what corbo_hip_get_solution returns -- also when the next step has already been re-armed and solved, and across repeated solves."""
import numpy as np
import pytest

from control_box_rst_amd import problems
from control_box_rst_amd.solver import BatchedLevenbergMarquardt

pytestmark = pytest.mark.gpu


def _solver(B, N):
    d = problems.quad_desc(N=N)
    x0, xf = problems.quad_instances(B)
    s = BatchedLevenbergMarquardt(d, B)
    s.setIterations(4)
    s.setPenaltyWeights(*problems.QUAD_WEIGHTS)
    X0 = s.init_trajectory(x0, xf)
    s.set_instance_data(X0, xref=xf)
    return s, X0, xf


@pytest.mark.parametrize("B,N", [(3, 12), (160, 64)])   # (3 OCPs: below 1 MB of results, fetched by the copy kernels as before; 160: delivered by the copy engine, reject-streak speculation on, partitioned chain)
def test_delivered_results_equal_get_solution(B, N):
    s, X0, xf = _solver(B, N)
    s.solve()
    Xr, chi2r, str_ = (a.copy() for a in s.get_solution())          # plain path: copies from the device
    s.set_result_sink(True)
    for rep in range(3):
        s.restore_instance_data()
        s.solve()
        X, chi2, status = s.fetch_solution()
        nv = s.dims.nv
        assert np.array_equal(np.asarray(X)[:, :nv], Xr) and np.array_equal(chi2, chi2r) and np.array_equal(status, str_), rep
    # the delivery of step k is still in flight when step k + 1 starts: both land, in order
    s.restore_instance_data(); s.solve_async()
    s.setIterations(2)
    s.restore_instance_data(); s.solve_async()
    s.synchronize()
    X2, chi22, _ = s.fetch_solution()
    s.set_result_sink(False)
    s.restore_instance_data(); s.solve()
    Xg, chi2g, _ = s.get_solution()
    assert np.array_equal(np.asarray(X2)[:, : s.dims.nv], Xg) and np.array_equal(chi22, chi2g)


def test_other_calls_between_solve_and_fetch_do_not_disturb_the_delivery():
    s, X0, xf = _solver(5, 16)
    s.set_result_sink(True)
    s.solve()
    u0 = s.get_first_control()          # (uses the handle's general staging buffer)
    st = s.get_stats()
    X, chi2, status = s.fetch_solution()
    Xg, chi2g, statusg = s.get_solution()
    assert np.array_equal(np.asarray(X)[:, : s.dims.nv], Xg) and np.array_equal(chi2, chi2g) and np.array_equal(status, statusg)
    assert np.array_equal(u0, Xg[:, 12:16]) and st["lm_iterations"] == 5 * 4
