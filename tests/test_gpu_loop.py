"""GPU tests (-m gpu) of the plant side of the closed loop on the device: corbo_hip_plant_* and
corbo_hip_warm_start_from_plant through the C-ABI, against the oracle's restatement of SimulatedPlant::control and against closed
loops of the genuine reference with its own SimulatedPlant (tests/golden/loop_*.json)."""
import numpy as np
import pytest

from conftest import desc_for, load_golden
from control_box_rst_amd import capi, problems
from control_box_rst_amd.solver import BatchedLevenbergMarquardt, CorboHipError

pytestmark = pytest.mark.gpu

LOOPS = ["loop_unicycle_rk4", "loop_unicycle_euler_noshift", "loop_vdp_euler", "loop_int3_rk4", "loop_quad_rk4", "loop_pquad_rk4", "loop_pendulum_rk4", "loop_duffing_euler", "loop_cartpole_rk4", "loop_lin32_rk4"]


def integrator_of(g):
    return capi.INTEGRATOR_RK4 if g["integrator"] == "rk4" else capi.INTEGRATOR_EULER


@pytest.mark.parametrize("scenario,N", [("vdp", 20), ("int3", 20), ("unicycle", 30), ("quad", 10), ("duffing", 12), ("rocket", 12), ("pendulum", 12),
                                        ("mpendulum", 12), ("toy", 12), ("artstein", 12), ("cartpole", 12), ("par2", 12), ("par3", 12), ("lin32", 12), ("lin41", 12)])
@pytest.mark.parametrize("integrator", [capi.INTEGRATOR_EULER, capi.INTEGRATOR_RK4])
def test_plant_step_vs_oracle(oracle_mod, scenario, N, integrator):
    """Per instance: x+ = integrator(x, u_0 of the resident trajectory, dt) + disturbance, same operations as the oracle -- bit for
    bit for the polynomial dynamics, to the last ulps of the device's sin / cos otherwise."""
    if scenario.startswith("lin"):
        nxl, nul = int(scenario[3]), int(scenario[4])
        r0 = np.random.default_rng(5)
        d = problems.linear_desc(r0.uniform(-1, 1, (nxl, nxl)), r0.uniform(-1, 1, (nxl, nul)), N=N)
    else:
        d = problems.SCENARIOS[scenario][0](N=N)
    B = 6
    rng = np.random.default_rng(7)
    p = oracle_mod.OracleProblem(d)
    nv, nx = p.dims.nv, d.nx
    X = rng.normal(scale=0.4, size=(B, nv))
    if d.grid in (capi.GRID_FD_VARIABLE, capi.GRID_MS_VARIABLE):
        X[:, -1] = 0.1
    xp = rng.normal(scale=0.5, size=(B, nx))
    dist = 1e-3 * rng.normal(size=(B, nx))
    s = BatchedLevenbergMarquardt(d, B)
    s.set_instance_data(X)
    s.plant_set_state(xp)
    assert np.array_equal(s.plant_get_state(), xp)
    dt = 0.07
    for rep, dd in enumerate((dist, None)):
        s.plant_step(dt=dt, integrator=integrator, disturbance=dd)
        got = s.plant_get_state()
        for b in range(B):
            p.set_data(X[b])
            xp[b] = p.plant_step(xp[b], integrator, dt, None if dd is None else dd[b])
        if scenario in ("vdp", "int3", "duffing", "rocket", "toy", "artstein", "par2", "par3", "lin32", "lin41"):
            assert np.array_equal(got, xp), (scenario, rep)
        else:
            assert np.abs(got - xp).max() <= 1e-14 * max(1.0, np.abs(xp).max()), (scenario, rep)
            xp = got.copy()   # continue from the device's states: one step at a time is compared


@pytest.mark.parametrize("name", LOOPS)
def test_closed_loop_vs_reference(name):
    """plant.output -> grid update -> solve -> plant.control entirely on the device, every step against the genuine reference."""
    g = load_golden(name)
    d = desc_for(g)
    B = 3
    xf = np.tile(np.array(g["xf"]), (B, 1))
    x0 = np.tile(np.array(g["steps"][0]["x0"]), (B, 1))
    s = BatchedLevenbergMarquardt(d, B)
    s.setIterations(g["iters"])
    s.setPenaltyWeights(*g["weights"])
    s.set_instance_data(s.init_trajectory(x0, xf), xref=xf)
    s.plant_set_state(x0)
    nv = s.dims.nv
    tol = 3e-4 if "quad" in name else 5e-6   # quadrotor: nearly flat directions, see tests/test_oracle_golden.py
    for k, st in enumerate(g["steps"]):
        if k > 0:
            s.warm_start_from_plant(shift=bool(g["shift"]))
        s.solve(new_run=True)
        X, chi2, status = s.get_solution()
        ref = np.array(st["vertex"])[:nv]
        for b in range(B):
            assert np.abs(X[b] - ref).max() <= tol, (name, k, b, np.abs(X[b] - ref).max())
            assert abs(chi2[b] - st["chi2"]) <= 2e-6 * abs(st["chi2"]), (name, k, b)
        s.plant_step(dt=st["plant_dt"], integrator=integrator_of(g), disturbance=np.tile(np.array(st["disturbance"]), (B, 1)))
        xp = s.plant_get_state()
        assert np.abs(xp - np.array(st["plant_after"])).max() <= tol, (name, k)
        assert np.array_equal(xp[0], xp[1]) and np.array_equal(xp[0], xp[2])   # identical instances stay identical


def test_device_plant_loop_equals_host_driven_loop():
    """warm_start_from_plant(device states) and warm_start(the same states read back to the host) give bit-identical loops."""
    d = problems.unicycle_desc(N=30)
    B = 16
    x0, xf = problems.unicycle_instances(B)
    a = BatchedLevenbergMarquardt(d, B)
    b = BatchedLevenbergMarquardt(d, B)
    rng = np.random.default_rng(3)
    for s in (a, b):
        s.setIterations(5)
        s.setPenaltyWeights(*problems.UNICYCLE_WEIGHTS)
        s.set_instance_data(s.init_trajectory(x0, xf), xref=xf)
        s.solve(new_run=True)
    a.plant_set_state(x0)
    for k in range(4):
        dist = 1e-2 * rng.normal(size=(B, d.nx))
        a.plant_step(integrator=capi.INTEGRATOR_RK4, disturbance=dist)
        x = a.plant_get_state()
        a.warm_start_from_plant(shift=True)
        b.warm_start(x, shift=True)
        a.solve(new_run=True)
        b.solve(new_run=True)
        Xa, ca, _ = a.get_solution()
        Xb, cb, _ = b.get_solution()
        assert np.array_equal(Xa, Xb) and np.array_equal(ca, cb), k
        # keep b's u_0 = a's: b has no plant of its own
    assert np.linalg.norm(x[:, :2] - xf[:, :2], axis=1).mean() < np.linalg.norm(x0[:, :2] - xf[:, :2], axis=1).mean()


def test_plant_calls_need_a_plant_state():
    d = problems.vdp_desc()
    s = BatchedLevenbergMarquardt(d, 2)
    x0 = np.array([[1.0, 0.0], [0.5, 0.1]])
    s.set_instance_data(s.init_trajectory(x0, np.zeros((2, 2))))
    for call in (lambda: s.plant_step(), lambda: s.plant_get_state(), lambda: s.warm_start_from_plant()):
        with pytest.raises(CorboHipError, match="plant_set_state"):
            call()
    s.plant_set_state(x0)
    with pytest.raises(CorboHipError, match="integrator"):
        s.plant_step(integrator=7)
    with pytest.raises(CorboHipError, match="dt"):
        s.plant_step(dt=-1.0)
    s.plant_step()
    assert np.isfinite(s.plant_get_state()).all()


@pytest.mark.parametrize("scenario,N,B,ocp_iterations", [("unicycle", 30, 16, 1), ("unicycle", 20, 5, 2), ("vdp", 20, 4, 1), ("dint", 20, 3, 1), ("quad", 10, 3, 1)])
def test_closed_loop_call_equals_stepwise_sequence(scenario, N, B, ocp_iterations):
    """corbo_hip_closed_loop (everything enqueued, one synchronisation; the quadrotor family runs it step by step inside) is
    bit-identical to the same sequence driven through the single-step entry points; its logs are the states / applied controls."""
    desc_fn, weights = problems.SCENARIOS[scenario]
    d = desc_fn(N=N)
    rng = np.random.default_rng(11)
    nx, nu = d.nx, d.nu
    x0 = rng.normal(scale=0.3, size=(B, nx))
    xf = np.zeros((B, nx))
    if scenario == "unicycle":
        x0, xf = problems.unicycle_instances(B)
    elif scenario == "dint":
        xf = np.tile(np.array([1.0, 0.0]), (B, 1))
    elif scenario == "quad":
        xf[:, :3] = [2.0, 1.0, 1.0]
    steps = 4
    dist = 1e-3 * rng.normal(size=(steps, B, nx))
    dt = d.dt_ref
    a = BatchedLevenbergMarquardt(d, B)
    b = BatchedLevenbergMarquardt(d, B)
    for s in (a, b):
        s.setIterations(4)
        s.setPenaltyWeights(*weights)
        if ocp_iterations > 1:
            s.setWeightAdapation(2.0, 2.0, 2.0, 500.0, 500.0, 500.0)
        s.set_instance_data(s.init_trajectory(x0, xf), xref=xf)
        s.solve(new_run=True)
        s.plant_set_state(x0)
    xs, us = a.closed_loop(steps, dt=dt, integrator=capi.INTEGRATOR_RK4, shift=True, disturbance=dist, ocp_iterations=ocp_iterations)
    for k in range(steps):
        u_applied = b.get_first_control()
        b.plant_step(dt=dt, integrator=capi.INTEGRATOR_RK4, disturbance=dist[k])
        assert np.array_equal(us[k], u_applied), k
        assert np.array_equal(xs[k], b.plant_get_state()), k
        b.warm_start_from_plant(shift=True)
        for it in range(ocp_iterations):
            b.solve(new_run=(it == 0))
    Xa, ca, sa = a.get_solution()
    Xb, cb, sb = b.get_solution()
    assert np.array_equal(Xa, Xb) and np.array_equal(ca, cb) and np.array_equal(sa, sb)
    assert np.array_equal(a.plant_get_state(), b.plant_get_state())
    assert a.closed_loop(2, dt=dt, log=False) is None          # without logs and without disturbance
    assert np.isfinite(a.plant_get_state()).all()


def test_long_closed_loop_and_handle_churn_leave_nothing_behind():
    """2000 control steps in one call stay finite and on target; creating and destroying handles (with plants, logs and per-instance
    bounds) in a loop returns all device memory."""
    import torch
    d = problems.unicycle_desc(N=30)
    B = 64
    x0, xf = problems.unicycle_instances(B)
    s = BatchedLevenbergMarquardt(d, B)
    s.setIterations(4)
    s.setPenaltyWeights(*problems.UNICYCLE_WEIGHTS)
    s.set_instance_data(s.init_trajectory(x0, xf), xref=xf)
    s.solve(new_run=True)
    s.plant_set_state(x0)
    s.closed_loop(2000, integrator=capi.INTEGRATOR_RK4, log=False)
    x = s.plant_get_state()
    assert np.isfinite(x).all() and np.linalg.norm(x[:, :2] - xf[:, :2], axis=1).max() < 0.15   # a unicycle parks near, not on, the goal
    _, _, status = s.get_solution()
    assert (status <= 1).all()
    del s
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for i in range(60):
        h = BatchedLevenbergMarquardt(d, B)
        X0 = h.init_trajectory(x0, xf)
        h.set_instance_data(X0, lb=np.full_like(X0, -5.0), ub=np.full_like(X0, 5.0), xref=xf)
        h.setIterations(2)
        h.solve(new_run=True)
        h.plant_set_state(x0)
        h.closed_loop(3, disturbance=np.zeros((3, B, 3)))
        del h
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    import os
    if "PYTEST_XDIST_WORKER" not in os.environ:   # (device-wide figure: meaningless while other test processes allocate on the same GPU)
        assert abs(free1 - free0) <= 8 << 20, (free0, free1)   # allocator granularity, not a per-handle leak


def test_plant_that_differs_from_the_controllers_model():
    """corbo_hip_plant_set_params: per-instance model parameters of the PLANT (SimulatedPlant takes its own dynamics object).  Instance 0
    runs the Van der Pol plant with damping 1.8 against the controller's 1.0 -- every step against the genuine reference's closed loop with
    that plant (tests/golden/loop_vdp_plant_mismatch.json); instance 1 keeps the nominal plant and follows the nominal loop's first step;
    one corbo_hip_closed_loop call gives the same as the steps one by one."""
    g = load_golden("loop_vdp_plant_mismatch")
    d = desc_for(g)
    B = 2
    xf = np.tile(np.array(g["xf"]), (B, 1))
    x0 = np.tile(np.array(g["steps"][0]["x0"]), (B, 1))
    prm = np.zeros((B, 8))
    prm[:] = [d.dyn_params[i] for i in range(8)]
    prm[0, 0] = g["plant_a"]
    dist = np.array([np.tile(np.array(st["disturbance"]), (B, 1)) for st in g["steps"]])

    def fresh():
        s = BatchedLevenbergMarquardt(d, B)
        s.setIterations(g["iters"])
        s.setPenaltyWeights(*g["weights"])
        s.set_instance_data(s.init_trajectory(x0, xf), xref=xf)
        s.plant_set_state(x0)
        s.plant_set_params(prm)
        return s

    s = fresh()
    nv = s.dims.nv
    for k, st in enumerate(g["steps"]):
        if k > 0:
            s.warm_start_from_plant(shift=bool(g["shift"]))
        s.solve(new_run=True)
        X, chi2, _ = s.get_solution()
        assert np.abs(X[0] - np.array(st["vertex"])[:nv]).max() <= 5e-6, k
        assert abs(chi2[0] - st["chi2"]) <= 2e-6 * abs(st["chi2"]), k
        s.plant_step(dt=st["plant_dt"], integrator=integrator_of(g), disturbance=dist[k])
        xp = s.plant_get_state()
        assert np.abs(xp[0] - np.array(st["plant_after"])).max() <= 5e-6, k
        if k == 0:
            assert np.abs(xp[1] - xp[0]).max() > 1e-5   # the nominal plant goes elsewhere (x2 starts at 0: the damping term is small)
    a = fresh()
    a.solve(new_run=True)
    a.closed_loop(len(g["steps"]) - 1, dt=g["dt"], integrator=integrator_of(g), shift=bool(g["shift"]), disturbance=dist[:-1], log=False)
    Xa, ca, _ = a.get_solution()
    assert np.abs(Xa[0] - np.array(g["steps"][-1]["vertex"])[:nv]).max() <= 5e-6
    a.plant_set_params(None)
    a.plant_step(dt=g["dt"], integrator=integrator_of(g))
    assert np.isfinite(a.plant_get_state()).all()
    lin = BatchedLevenbergMarquardt(problems.linear_desc(np.eye(2) * -0.1, np.ones((2, 1)), N=6), 2)
    with pytest.raises(CorboHipError, match="linear state-space"):
        lin.plant_set_params(np.zeros((2, 8)))
