"""Time-varying state reference (ReferenceTrajectoryInterface::getReferenceCached(k) in the cost and final-stage terms:
quadratic_cost.cpp:100-119, final_state_cost.cpp:72-92, final_state_constraints.cpp:60-80 / .h:149-154).
Fixtures from the genuine reference with a DiscreteTimeReferenceTrajectory, one sample per grid point (oracle/gen_golden.py tvref);
"ref_vertex" records what getReferenceCached handed out.  A non-zero CONTROL reference is refused: the reference's least-squares
control term is not defined for one (quadratic_cost.cpp:160-163 assigns a scalar to the nu-vector; with nu = 2 the second row of
the fixture this was tried on held uninitialised memory).
CPU: the oracle (oracle_set_references) bit for bit on values / Jacobian, iterates to the usual tolerance.
GPU (-m gpu): corbo_hip_set_references through the C-ABI against the fixtures, fused and big-block kernels, Hessian operators."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import desc_for, load_golden
from control_box_rst_amd import capi

TVREF = ["unicycle_n12_tvref", "vdp_tvref", "unicycle_n12_tball_tvref", "vdp_teq_tvref", "pendulum_ms_rk4_tvref", "quad_n10_tvref",
         "unicycle_n12_fullq_tvref"]   # non-diagonal weights against a time-varying reference
# unicycle_n12_tvref: the start (x_k = xref_k, u = 0) is far from feasible -- chi2 3080 -> 2158 -> 55 in the first steps -- and the k = 1
# iterate agrees to 1e-12; from there the finite-difference noise amplifies the few-ulp difference of the
# start (the fixture's vertex_init carries the drift of one in-place sweep) to 1e-5 at k = 6 with chi2 equal to 2e-8 relative
X_TOL = {"quad_n10_tvref": 3e-4, "pendulum_ms_rk4_tvref": 5e-5, "unicycle_n12_tvref": 3e-5, "unicycle_n12_tball_tvref": 3e-5, "unicycle_n12_fullq_tvref": 3e-5}
CHI2_RTOL = {"pendulum_ms_rk4_tvref": 2e-5}


def start_of(g, nv):
    """The start of the LM iterations: with a time-varying reference the grids initialise the trajectory from it (the finite-differences
    grid: x_k = xref_k); the fixture's vertex_init is that start a few ulps off (it was dumped after one in-place finite-difference sweep)."""
    return np.array(g["vertex_init"])[:nv].copy()


@pytest.mark.parametrize("name", TVREF)
def test_oracle_values_jacobian_and_iterates(oracle_mod, name):
    g = load_golden(name)
    d = desc_for(g)
    p = oracle_mod.OracleProblem(d)
    nv = p.dims.nv
    ref = np.array(g["ref_vertex"])[:nv]
    p.set_data(np.array(g["vertex_init"])[:nv], xref=np.array(g["xf"]))
    p.set_references(ref)
    values, jac = p.eval(*g["weights"])
    assert np.array_equal(values, np.array(g["values_init"]))
    rows, cols = p.structure()
    Jo = sp.coo_matrix((jac, (rows, cols)), shape=(p.dims.m, p.dims.n)).tocsr()
    Jr = sp.coo_matrix((g["jac_vals"], (g["jac_rows"], g["jac_cols"])), shape=(p.dims.m, p.dims.n)).tocsr()
    assert abs(Jo - Jr).max() == 0.0
    # without the references the residual differs (the fixture does exercise them)
    p.set_references(None)
    assert not np.array_equal(p.eval(*g["weights"])[0], np.array(g["values_init"]))
    for a in g["after_iter"]:
        q = oracle_mod.OracleProblem(d)
        q.set_data(start_of(g, nv), xref=np.array(g["xf"]))
        q.set_references(ref)
        status, chi2, _ = q.solve(capi.default_lm_opts(a["k"], *g["weights"]))
        assert np.abs(q.x() - np.array(a["vertex"])[:nv]).max() <= X_TOL.get(name, 5e-6), (name, a["k"])
        assert abs(chi2 - a["chi2"]) <= CHI2_RTOL.get(name, 2e-6) * max(1.0, abs(a["chi2"])), (name, a["k"])


HESS_TVREF = ["hess_unicycle_tvref", "hess_unicycle_tvref_nonlsq", "hess_unicycle_tvref_integral_trap", "hess_unicycle_tvref_ms_integral"]   # least-squares terms,
# plain terms, integral cost edges, the shooting grid's mixed edges -- each against one reference per grid point


@pytest.mark.parametrize("name", HESS_TVREF)
def test_oracle_hessians_with_references(oracle_mod, name):
    g = load_golden(name)
    d = desc_for(g)
    p = oracle_mod.OracleProblem(d)
    p.set_data(np.array(g["vertex_point"])[:p.dims.nv], xref=np.array(g["xf"]))
    p.set_references(np.array(g["ref_vertex"])[:p.dims.nv])
    for lower, tag in ((0, "full"), (1, "lower")):
        trip = p.hessians(lower, g["mult_obj"], g["mult_eq"], g["mult_ineq"])
        for (r, c, v), key in zip(trip, ("hobj", "heq", "hineq")):
            assert np.array_equal(v, np.array(g[f"{key}_vals_{tag}"])), (tag, key)
    r, c, v, lbA, ubA = p.linear_form()
    assert np.array_equal(v, np.array(g["lin_vals"])) and np.array_equal(lbA, np.array(g["lin_lbA"]))
    grad, obj = p.objective_gradient()
    assert np.array_equal(grad, np.array(g["grad_obj"])) and abs(obj - g["obj_value"]) <= 4e-16 * abs(g["obj_value"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", TVREF)
def test_device_values_jacobian_and_iterates(name):
    from control_box_rst_amd.solver import BatchedLevenbergMarquardt, get_structure
    g = load_golden(name)
    d = desc_for(g)
    B = 3
    s = BatchedLevenbergMarquardt(d, B)
    nv = s.dims.nv
    s.setPenaltyWeights(*g["weights"])
    ref = np.tile(np.array(g["ref_vertex"])[:nv], (B, 1))
    s.set_instance_data(np.tile(np.array(g["vertex_init"])[:nv], (B, 1)), xref=np.tile(np.array(g["xf"]), (B, 1)))
    assert s.lib.corbo_hip_set_references(s._h, ref.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_double))) == 0
    values, jac = s.eval()
    assert np.abs(values[1] - np.array(g["values_init"])).max() <= 1e-12 * max(1.0, max(g["weights"]))
    rows, cols = get_structure(d)
    Jg = sp.coo_matrix((jac[2], (rows, cols)), shape=(s.dims.m, s.dims.n)).tocsr()
    Jr = sp.coo_matrix((g["jac_vals"], (g["jac_rows"], g["jac_cols"])), shape=(s.dims.m, s.dims.n)).tocsr()
    assert abs(Jg - Jr).max() <= 1e-6 * max(1.0, abs(Jr).max())
    for a in g["after_iter"]:
        s.setIterations(a["k"])
        s.set_instance_data(np.tile(start_of(g, nv), (B, 1)), xref=np.tile(np.array(g["xf"]), (B, 1)))   # (keeps the references)
        for rtc in (1, 0):   # run-to-completion kernel and one launch per pass (the big-block family always runs per pass)
            s.set_option("run_to_completion", rtc)
            s.restore_instance_data()
            s.solve(new_run=True)
            x, chi2, status = s.get_solution()
            for b in range(B):
                assert np.abs(x[b] - np.array(a["vertex"])[:nv]).max() <= X_TOL.get(name, 5e-6), (name, a["k"], rtc, b)
                assert abs(chi2[b] - a["chi2"]) <= CHI2_RTOL.get(name, 2e-6) * max(1.0, abs(a["chi2"])), (name, a["k"], rtc, b)
    # back to the static reference: the residual changes
    s.set_references()
    s.set_instance_data(np.tile(np.array(g["vertex_init"])[:nv], (B, 1)), xref=np.tile(np.array(g["xf"]), (B, 1)))
    assert np.abs(s.eval()[0][0] - np.array(g["values_init"])).max() > 1e-6


@pytest.mark.gpu
def test_device_references_per_instance_and_hessians(oracle_mod):
    """Different references per instance (set_references from trajectories) against the oracle; Hessian operators with references
    against the reference fixture."""
    from control_box_rst_amd import problems
    from control_box_rst_amd.solver import BatchedLevenbergMarquardt
    d = problems.unicycle_desc(N=20)
    B = 6
    rng = np.random.default_rng(3)
    x0, xf = problems.unicycle_instances(B)
    s = BatchedLevenbergMarquardt(d, B)
    s.setIterations(5)
    s.setPenaltyWeights(*problems.UNICYCLE_WEIGHTS)
    X0 = s.init_trajectory(x0, xf)
    xr = xf[:, None, :] + 0.2 * rng.normal(size=(B, d.N, d.nx)) * np.linspace(1, 0, d.N)[None, :, None]
    s.set_instance_data(X0, xref=xf)
    s.set_references(xr)
    s.solve(new_run=True)
    X, chi2, status = s.get_solution()
    S = d.nx + d.nu
    for b in range(B):
        p = oracle_mod.OracleProblem(d)
        ref = np.zeros(p.dims.nv)
        for k in range(d.N):
            ref[k * S:k * S + d.nx] = xr[b, k]
        p.set_data(X0[b], xref=xf[b])
        p.set_references(ref)
        st, c, _ = p.solve(capi.default_lm_opts(5, *problems.UNICYCLE_WEIGHTS))
        assert np.abs(X[b] - p.x()).max() <= 5e-6, b
        assert abs(chi2[b] - c) <= 2e-6 * max(1.0, abs(c))
    bad = np.zeros((B, s.dims.nv))
    bad[0, d.nx] = 0.1   # a control reference
    import ctypes as C
    assert s.lib.corbo_hip_set_references(s._h, bad.ctypes.data_as(C.POINTER(C.c_double))) < 0
    for name in HESS_TVREF:
        g = load_golden(name)
        dg = desc_for(g)
        h = BatchedLevenbergMarquardt(dg, 2)
        nv = h.dims.nv
        h.set_instance_data(np.tile(np.array(g["vertex_point"])[:nv], (2, 1)), xref=np.tile(np.array(g["xf"]), (2, 1)))
        refv = np.ascontiguousarray(np.tile(np.array(g["ref_vertex"])[:nv], (2, 1)))
        assert h.lib.corbo_hip_set_references(h._h, refv.ctypes.data_as(C.POINTER(C.c_double))) == 0
        vals = h.eval_hessians(True, g["mult_obj"], np.array(g["mult_eq"]), None)
        for c, key in enumerate(("hobj", "heq")):
            gv = np.array(g[f"{key}_vals_lower"])
            assert vals[c].shape[1] == len(gv), (name, key)
            assert np.abs(vals[c][1] - gv).max() <= 2e-4 * max(1.0, np.abs(gv).max()), (name, key)
        grad, obj = h.objective_gradient()
        gg = np.array(g["grad_obj"])
        assert np.abs(grad[1] - gg).max() <= 1e-6 * max(1.0, np.abs(gg).max()), name
        assert abs(obj[1] - g["obj_value"]) <= 1e-13 * max(1.0, abs(g["obj_value"])), name


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["loop_unicycle_tvref", "loop_vdp_tvref"])
def test_tracking_closed_loop_vs_reference(name):
    """Tracking MPC: the reference's closed loop (SimulatedPlant, warm start with shifting) with a DiscreteTimeReferenceTrajectory sampled
    at t + k dt -- every control step sees a shifted window of it.  On the device: plant step, warm start from the plant, the step's
    references (corbo_hip_set_references), solve; every step against the genuine reference."""
    import ctypes as C
    from control_box_rst_amd import capi as cp
    from control_box_rst_amd.solver import BatchedLevenbergMarquardt
    g = load_golden(name)
    d = desc_for(g)
    B = 2
    s = BatchedLevenbergMarquardt(d, B)
    nv, nx = s.dims.nv, d.nx
    s.setIterations(g["iters"])
    s.setPenaltyWeights(*g["weights"])
    xf = np.tile(np.array(g["xf"]), (B, 1))
    x0 = np.tile(np.array(g["steps"][0]["x0"]), (B, 1))
    # cold start of the reference with a time-varying reference: x_k = xref_k, u = 0, x_0 = the measured state
    X = np.tile(np.array(g["steps"][0]["ref_vertex"])[:nv], (B, 1))
    X[:, :nx] = x0
    s.set_instance_data(X, xref=xf)
    s.plant_set_state(x0)
    integ = cp.INTEGRATOR_RK4 if g["integrator"] == "rk4" else cp.INTEGRATOR_EULER
    for k, st in enumerate(g["steps"]):
        if k > 0:
            s.warm_start_from_plant(shift=bool(g["shift"]))
        ref = np.ascontiguousarray(np.tile(np.array(st["ref_vertex"])[:nv], (B, 1)))
        assert s.lib.corbo_hip_set_references(s._h, ref.ctypes.data_as(C.POINTER(C.c_double))) == 0
        s.solve(new_run=True)
        Xs, chi2, status = s.get_solution()
        want = np.array(st["vertex"])[:nv]
        for b in range(B):
            assert np.abs(Xs[b] - want).max() <= 3e-5, (name, k, b, np.abs(Xs[b] - want).max())
            assert abs(chi2[b] - st["chi2"]) <= 2e-6 * abs(st["chi2"]), (name, k, b)
        s.plant_step(dt=st["plant_dt"], integrator=integ, disturbance=np.tile(np.array(st["disturbance"]), (B, 1)))
        assert np.abs(s.plant_get_state() - np.array(st["plant_after"])).max() <= 3e-5, (name, k)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["loop_unicycle_tvref", "loop_vdp_tvref"])
def test_resident_tracking_loop_vs_reference(name):
    """The same tracking loops with everything resident: corbo_hip_set_reference_trajectory + ONE corbo_hip_closed_loop call (plant step,
    warm start, reference window one sample on, solve -- per control step, no host in between) against the genuine reference, and
    bit-identical to the loop driven step by step with corbo_hip_set_references."""
    import ctypes as C
    from control_box_rst_amd import capi as cp
    from control_box_rst_amd.solver import BatchedLevenbergMarquardt
    g = load_golden(name)
    d = desc_for(g)
    B, steps = 2, len(g["steps"])
    nx, S = d.nx, d.nx + d.nu
    integ = cp.INTEGRATOR_RK4 if g["integrator"] == "rk4" else cp.INTEGRATOR_EULER
    r0 = np.array(g["steps"][0]["ref_vertex"])
    traj = np.array([r0[k * S:k * S + nx] for k in range(d.N)])   # one sample per grid point; the reference holds the last one beyond
    dist = np.array([np.tile(np.array(st["disturbance"]), (B, 1)) for st in g["steps"][:-1]])

    def fresh():
        s = BatchedLevenbergMarquardt(d, B)
        s.setIterations(g["iters"])
        s.setPenaltyWeights(*g["weights"])
        X = np.tile(r0[:s.dims.nv], (B, 1))
        X[:, :nx] = g["steps"][0]["x0"]
        s.set_instance_data(X, xref=np.tile(np.array(g["xf"]), (B, 1)))
        s.plant_set_state(np.tile(np.array(g["steps"][0]["x0"]), (B, 1)))
        return s

    a = fresh()
    a.set_reference_trajectory(traj, step=0)
    a.solve(new_run=True)
    xs, us = a.closed_loop(steps - 1, dt=g["dt"], integrator=integ, shift=bool(g["shift"]), disturbance=dist)
    Xa, ca, _ = a.get_solution()
    nv = a.dims.nv
    for k in range(steps - 1):
        assert np.abs(xs[k] - np.array(g["steps"][k]["plant_after"])).max() <= 3e-5, (name, k)
    assert np.abs(Xa[0] - np.array(g["steps"][-1]["vertex"])[:nv]).max() <= 3e-5
    assert abs(ca[0] - g["steps"][-1]["chi2"]) <= 2e-6 * abs(g["steps"][-1]["chi2"])
    # step by step with explicit references
    b = fresh()
    for k in range(steps):
        if k > 0:
            b.plant_step(dt=g["dt"], integrator=integ, disturbance=dist[k - 1])
            b.warm_start_from_plant(shift=bool(g["shift"]))
        win = traj[[min(k + j, d.N - 1) for j in range(d.N)]]
        b.set_references(np.tile(win, (B, 1, 1)))
        b.solve(new_run=True)
    Xb, cb, _ = b.get_solution()
    assert np.array_equal(Xa, Xb) and np.array_equal(ca, cb)
    a.set_reference_trajectory(None)
    a.set_instance_data(Xa, xref=np.tile(np.array(g["xf"]), (B, 1)))
    a.solve(new_run=True)   # (static reference again: still solves)
    assert np.isfinite(a.get_solution()[1]).all()
