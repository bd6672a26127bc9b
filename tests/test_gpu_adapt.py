"""GPU tests (-m gpu) of the time-optimal grid adaptation on the device: corbo_hip_resample_into against the oracle's restatement of
resampleTrajectory (bit for bit), and the bucketed batch of adaptive controllers (control_box_rst_amd/adaptive_grid.py) against
moving-horizon sequences of the genuine reference (tests/golden/mpc_dint_adapt_*.json; mpc_dint_ms_adapt_*: the MultipleShootingVariableGrid)."""
import numpy as np
import pytest

from conftest import load_golden
from control_box_rst_amd import adaptive_grid, problems
from control_box_rst_amd.solver import BatchedLevenbergMarquardt

pytestmark = pytest.mark.gpu
STRATEGY = {"single": adaptive_grid.SINGLE_STEP, "aggressive": adaptive_grid.AGGRESSIVE, "shrink": adaptive_grid.SHRINK}


@pytest.fixture(scope="module", autouse=True)
def _built():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import __graft_entry__ as g
    g.build()


@pytest.mark.parametrize("shooting", [False, True], ids=["fd", "ms"])   # ms: ShootingGridBase::resampleTrajectory (shooting_grid_base.cpp:473-547)
@pytest.mark.parametrize("n_dst", [19, 21, 33, 12, 20, 2, 150])
def test_resample_into_vs_oracle_bit_exact(oracle_mod, n_dst, shooting):
    n_src, B = 20, 5
    rng = np.random.default_rng(n_dst)
    src = BatchedLevenbergMarquardt(problems.dint_desc(N=n_src, shooting=shooting), B)
    dst = BatchedLevenbergMarquardt(problems.dint_desc(N=n_dst, shooting=shooting), B)
    X = rng.normal(size=(B, src.dims.nv))
    X[:, -1] = rng.uniform(0.03, 0.3, B)       # dt
    xref = rng.normal(size=(B, 2))
    src.set_instance_data(X, xref=xref)
    dst.prepare_slots(B)
    src.resample_into(dst, [4, 0, 2], [1, 3, 0])
    Y, _, _ = dst.get_solution()
    for s_i, d_i in ((4, 1), (0, 3), (2, 0)):
        assert np.array_equal(Y[d_i], oracle_mod.resample_trajectory(2, 1, X[s_i], n_dst)), (n_dst, s_i)
    assert np.array_equal(Y[2], np.zeros(dst.dims.nv)) and np.array_equal(Y[4], np.zeros(dst.dims.nv))   # untouched slots
    if n_dst == n_src:   # same N: a move inside one handle (compaction)
        src.resample_into(src, [4], [1])
        Z, _, _ = src.get_solution()
        assert np.array_equal(Z[1], X[4]) and np.array_equal(Z[0], X[0])


@pytest.mark.parametrize("name", ["mpc_dint_adapt_cross256_aggressive", "mpc_dint_adapt_cross256_single",   # the grid grows past 256 points: long-horizon kernels
                                  "mpc_dint_adapt_single", "mpc_dint_adapt_aggressive", "mpc_dint_ms_adapt_single", "mpc_dint_ms_adapt_aggressive",
                                  "mpc_dint_ms_adapt_aggressive_collapse"])
def test_adaptive_controller_on_the_device_vs_reference(name):
    g = load_golden(name)
    shooting = g.get("grid") == "ms"   # MultipleShootingVariableGrid: AdaptiveGridBatch picks its aggressive rule from the descriptor
    ctl = adaptive_grid.AdaptiveGridBatch(lambda n: problems.dint_desc(N=n, dt=g["dt"], shooting=shooting), 1, g["N"], strategy=STRATEGY[g["adapt"]], n_min=g["nmin"],
                                          n_max=g["nmax"], hyst=g["hyst"], adapt_first_iter=bool(g["adapt_first"]))
    ctl.setPenaltyWeights(*g["weights"])
    ctl.initialize([g["steps"][0]["x0"]], [g["xf"]])
    for s, st in enumerate(g["steps"]):
        ctl.setIterations(g["iters0"] if s == 0 else g["iters"])
        n_seq = []
        for it in range(g["ocp_iters"]):
            ctl.compute([st["x0"]], new_run=(it == 0))
            n_seq.append(int(ctl.grid_sizes()[0]))
        assert n_seq == st["n_seq"], (name, s, n_seq, st["n_seq"])
        x = ctl.trajectories()[0]
        ref = np.array(st["vertex"])
        assert len(x) == len(ref) and np.abs(x - ref).max() <= 5e-6, (name, s, np.abs(x - ref).max())
    ctl.close()


@pytest.mark.parametrize("shooting", [False, True], ids=["fd", "ms"])
def test_batch_of_adaptive_controllers_equals_the_instances_run_alone(shooting):
    """16 double-integrator controllers with different distances to the goal: their grids end up with different N (several buckets, instances
    moving between them, holes being closed).  Every instance's trajectory is bit-identical to the same instance run as a batch of one."""
    B, steps, K = 16, 5, 3
    rng = np.random.default_rng(11)
    x0 = np.zeros((B, 2))
    xf = np.tile([1.0, 0.0], (B, 1))
    xf[:, 0] = rng.uniform(0.3, 2.5, B)
    dist = rng.normal(scale=0.005, size=(steps, B, 2))

    def run(ids):
        ctl = adaptive_grid.AdaptiveGridBatch(lambda n: problems.dint_desc(N=n, shooting=shooting), len(ids), 30, strategy=adaptive_grid.SINGLE_STEP, n_min=5, n_max=60,
                                              hyst=0.05)
        ctl.setPenaltyWeights(*problems.DINT_WEIGHTS)
        ctl.setIterations(6)
        ctl.initialize(x0[ids], xf[ids])
        xm = x0[ids].copy()
        hist = []
        for s in range(steps):
            ctl.step(xm, ocp_iterations=K)
            tr = ctl.trajectories()
            hist.append((ctl.grid_sizes(), tr))
            xm = np.array([t[3:5] for t in tr]) + dist[s][ids]     # x_1 of the solution + disturbance = next measured state
        moves = ctl.moves
        ctl.close()
        return hist, moves

    full, moves = run(list(range(B)))
    assert moves > B and len(set(full[-1][0])) >= 3          # instances did move, several grid sizes are in use at the end
    for b in (0, 5, 11, 15):
        alone, _ = run([b])
        for s in range(steps):
            assert full[s][0][b] == alone[s][0][0], (b, s)
            assert np.array_equal(full[s][1][b], alone[s][1][0]), (b, s)
