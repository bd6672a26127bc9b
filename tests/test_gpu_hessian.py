"""Exact-Hessian path, -m gpu: corbo_hip_hessian_structure / corbo_hip_eval_hessians /
corbo_hip_eval_linear_form through the C-ABI against the genuine reference's outputs (tests/golden/hess_*.json) and against the oracle.

Structure: identical entry by entry.  Values: a lane reproduces the reference's finite-difference sequence on private copies of its
stage's vertices, so it cannot see the few-ulp drift the reference's IN-PLACE perturbations leave in vertices shared with edges
evaluated earlier.  A few ulps of the point are 1e-7 in a central-difference Jacobian entry (delta = 1e-9) and, through the forward
step of 1e-2, 1e-5 in a Hessian entry.  The tolerance is therefore the reference's own reproducibility: the difference between two
CONSECUTIVE evaluations of the same operator by the reference (second one from the drifted point), measured here with the
bit-exact oracle and asserted to be of the same size as the device's deviation."""
import numpy as np
import pytest

from conftest import desc_for, load_golden
from control_box_rst_amd.solver import BatchedLevenbergMarquardt
from control_box_rst_amd import problems

pytestmark = pytest.mark.gpu

HESS = ["hess_kcar", "hess_pquad_n5", "hess_pquad_fd_n5", "hess_unicycle_fullq", "hess_vdp", "hess_vdp_forward", "hess_vdp_backward", "hess_vdp_midpoint", "hess_vdp_teq", "hess_dint", "hess_int3_time_optimal",
        "hess_unicycle_n16", "hess_unicycle_xf_fixed", "hess_unicycle_n24_ball", "hess_pendulum_ms_rk4", "hess_pendulum_ms_rk5", "hess_cartpole", "hess_quad_n4", "hess_int3_ms_time_optimal",
        "hess_dint_mtq", "hess_int3_ms_mtq", "hess_dint_mtq_last5",
        "hess_vdp_nonlsq", "hess_unicycle_nonlsq", "hess_unicycle_nonlsq_tball", "hess_dint_nonlsq", "hess_dint_mtq_nonlsq", "hess_int3_ms_nonlsq",
        "hess_vdp_integral_trap", "hess_unicycle_integral_trap", "hess_unicycle_integral_left",
        # MultipleShootingEdgeSingleControl: shooting grid + integral-form cost -> one mixed edge (integrated cost + defect) per interval
        "hess_unicycle_ms_integral", "hess_unicycle_ms_integral_xf_fixed", "hess_vdp_ms_integral_euler", "hess_vdp_ms_integral_rk3", "hess_unicycle_ms_integral_rk5",
        "hess_unicycle_ms_integral_teq", "hess_unicycle_ms_integral_tball",
        "hess_pquad_ms_integral", "hess_pquad_ms_integral_rk3",   # the mixed edge around a big-block model (nx = 6; the 12-state fixture hess_quad_ms_integral pins the oracle only: the device refuses nx > 8 here)
        # TerminalPartialEqualityConstraint on the Hessian path: rows / multipliers for the active components of x_f only
        "hess_unicycle_pteq", "hess_cartpole_pteq", "hess_unicycle_ms_pteq", "hess_unicycle_ms_integral_pteq", "hess_pquad_pteq",
        "hess_dint_mtq_integral_trap", "hess_dint_mtq_integral_left_last4"]   # MinTimeQuadratic in integral form (free dt)   # integral-form cost: one objective edge per interval   # *_nonlsq: plain (non-least-squares) objective edges
KEYS = ("hobj", "heq", "hineq")
REL = 2e-4   # of max(1, max |value| of the list): see the module docstring; checked against the reference's own spread below


def device_at_point(g, B=3):
    d = desc_for(g)
    s = BatchedLevenbergMarquardt(d, B)
    nv = s.dims.nv
    X = np.tile(np.array(g["vertex_point"])[:nv], (B, 1))
    s.set_instance_data(X, xref=np.tile(np.array(g["xf"]), (B, 1)))
    return d, s


@pytest.mark.parametrize("name", HESS)
def test_hessians_vs_reference(oracle_mod, name):
    g = load_golden(name)
    d, s = device_at_point(g)
    B = s.batch
    # the reference's own reproducibility: the oracle (bit-exact with the fixture) evaluated twice in a row
    p = oracle_mod.OracleProblem(d)
    p.set_data(np.array(g["vertex_point"])[:p.dims.nv], xref=np.array(g["xf"]))
    first = p.hessians(1, g["mult_obj"], g["mult_eq"], g["mult_ineq"])
    second = p.hessians(1, g["mult_obj"], g["mult_eq"], g["mult_ineq"])
    own_spread = max([np.abs(a[2] - b[2]).max() / max(1.0, np.abs(a[2]).max()) for a, b in zip(first, second) if len(a[2])] + [0.0])
    assert own_spread <= REL
    for lower, tag in ((0, "full"), (1, "lower")):
        st = s.hessian_structure(bool(lower))
        vals = s.eval_hessians(bool(lower), g["mult_obj"], np.array(g["mult_eq"]), np.array(g["mult_ineq"]) if g["mult_ineq"] else None)
        for c, key in enumerate(KEYS):
            gr, gc, gv = np.array(g[f"{key}_rows_{tag}"], np.int32), np.array(g[f"{key}_cols_{tag}"], np.int32), np.array(g[f"{key}_vals_{tag}"])
            assert np.array_equal(st[c][0], gr) and np.array_equal(st[c][1], gc), (name, tag, key)
            assert vals[c].shape == (B, len(gv))
            if len(gv) == 0:
                continue
            for b in range(B):
                err = np.abs(vals[c][b] - gv).max() / max(1.0, np.abs(gv).max())
                assert err <= REL, (name, tag, key, b, err, own_spread)
            assert np.array_equal(vals[c][0], vals[c][1]) and np.array_equal(vals[c][0], vals[c][2])   # identical instances, identical bits
    if name in ("hess_vdp", "hess_dint"):   # polynomial dynamics, objective list: the Gauss-Newton blocks do not feel the drift -- bit-exact
        assert np.array_equal(s.eval_hessians(False, g["mult_obj"], np.array(g["mult_eq"]))[0][0], np.array(g["hobj_vals_full"]))


@pytest.mark.parametrize("name", HESS)
def test_linear_form_vs_reference(name):
    g = load_golden(name)
    d, s = device_at_point(g)
    rows, cols, vals, lbA, ubA = s.linear_form()
    assert np.array_equal(rows, np.array(g["lin_rows"], np.int32)) and np.array_equal(cols, np.array(g["lin_cols"], np.int32))
    gv, gl, gu = np.array(g["lin_vals"]), np.array(g["lin_lbA"]), np.array(g["lin_ubA"])
    # eval_grad_f / eval_f of the interior-point interface (computeGradientObjective, computeValueObjective)
    grad, obj = s.objective_gradient()
    gg = np.array(g["grad_obj"])
    for b in range(s.batch):
        assert np.abs(grad[b] - gg).max() <= 1e-6 * max(1.0, np.abs(gg).max()), (name, b)   # 2 v J: J's finite-difference noise (1e-7) times v
        assert abs(obj[b] - g["obj_value"]) <= 1e-13 * max(1.0, abs(g["obj_value"])), (name, b)
    # eval_g: the constraint values are -lbA / -ubA; eval_jac_g: the value list without the bound rows
    gcon, eq = np.array(g["g_values"]), s.dims.eq
    assert np.abs(-lbA[0][:eq] - gcon[:eq]).max() <= 1e-12 * max(1.0, np.abs(gcon).max())
    for b in range(s.batch):
        # Jacobian blocks: central differences at a point a few ulps away from the reference's drifted one
        assert np.abs(vals[b] - gv).max() <= 2e-6 * max(1.0, np.abs(gv).max()), (name, b)
        fin = np.isfinite(gl)
        assert np.array_equal(np.isfinite(lbA[b]), fin) and np.all(lbA[b][~fin] < -1e29)
        assert np.abs(lbA[b][fin] - gl[fin]).max() <= 1e-12 * max(1.0, np.abs(gl[fin]).max()), (name, b)
        fu = np.abs(gu) < 1e29
        assert np.abs(ubA[b][fu] - gu[fu]).max() <= 1e-12 * max(1.0, np.abs(gu[fu]).max()), (name, b)
        assert np.all(np.abs(ubA[b][~fu]) > 1e29)


def test_hessians_batch_vs_oracle(oracle_mod):
    """Different instances, per-instance multipliers: every instance against the oracle at its own point."""
    d = problems.unicycle_desc(N=20)
    B = 8
    rng = np.random.default_rng(5)
    x0, xf = problems.unicycle_instances(B)
    s = BatchedLevenbergMarquardt(d, B)
    X = s.init_trajectory(x0, xf) + 0.05 * rng.normal(size=(B, s.dims.nv))
    X[:, :d.nx] = x0
    s.set_instance_data(X, xref=xf)
    me = rng.uniform(0.2, 1.0, (B, s.dims.eq))
    vals = s.eval_hessians(True, 0.8, me, None)
    p = oracle_mod.OracleProblem(d)
    for b in range(B):
        p.set_data(X[b], xref=xf[b])
        ref = p.hessians(1, 0.8, me[b], None)
        for c in range(3):
            if len(ref[c][2]):
                assert np.abs(vals[c][b] - ref[c][2]).max() <= REL * max(1.0, np.abs(ref[c][2]).max()), (b, c)


@pytest.mark.parametrize("variant", ["xf_fixed", "teq", "tball", "vdp"])
def test_mixed_edges_structure_variants_vs_oracle(oracle_mod, variant):
    """The mixed edges next to what else a graph can carry: fixed x_f components (fewer columns in the last edge's blocks), a terminal equality
    (its rows and blocks come BEFORE the mixed edges'), a TerminalBall (the reference then lists every mixed edge's blocks in the inequality list
    too, as zeros) -- structure entry for entry, values within the operators' tolerance, per instance against the oracle."""
    from control_box_rst_amd import capi
    d = problems.hessian_path_cost_form(problems.vdp_desc(N=9) if variant == "vdp" else problems.unicycle_desc(N=9, terminal_ball=(((1.0, 1.0, 0.1), 0.02) if variant == "tball" else None)), integral="trapezoidal")
    d.grid, d.defect, d.shooting_integrator = capi.GRID_MS, capi.DEFECT_RK4_SHOOTING, 2
    if variant == "xf_fixed": d.xf_fixed_mask = 6
    if variant == "teq": d.final_eq = 1
    B = 4
    rng = np.random.default_rng(77)
    s = BatchedLevenbergMarquardt(d, B)
    x0 = rng.uniform(-0.5, 0.5, (B, d.nx))
    xf = rng.uniform(-0.5, 0.5, (B, d.nx))
    X = s.init_trajectory(x0, xf) + 0.05 * rng.normal(size=(B, s.dims.nv))
    X[:, :d.nx] = x0
    s.set_instance_data(X, xref=xf)
    me = rng.uniform(0.2, 1.0, (B, s.dims.eq))
    mi = rng.uniform(0.2, 1.0, (B, s.dims.ineq)) if s.dims.ineq else None
    p = oracle_mod.OracleProblem(d)
    assert (p.dims.n, p.dims.eq, p.dims.ineq, p.dims.bounds) == (s.dims.n, s.dims.eq, s.dims.ineq, s.dims.bounds)
    for lower in (True, False):
        st = s.hessian_structure(lower)
        vals = s.eval_hessians(lower, 1.3, me, mi)
        for b in range(B):
            p.set_data(X[b], xref=xf[b])
            ref = p.hessians(1 if lower else 0, 1.3, me[b], None if mi is None else mi[b])
            for c in range(3):
                assert np.array_equal(st[c][0], ref[c][0]) and np.array_equal(st[c][1], ref[c][1]), (variant, lower, c)
                if len(ref[c][2]):
                    assert np.abs(vals[c][b] - ref[c][2]).max() <= REL * max(1.0, np.abs(ref[c][2]).max()), (variant, b, c)
    rows, cols, vals, lbA, ubA = s.linear_form()
    for b in range(B):
        p.set_data(X[b], xref=xf[b])
        r, c, v, l, u = p.linear_form()
        assert np.array_equal(rows, r) and np.array_equal(cols, c)
        assert np.abs(vals[b] - v).max() <= 2e-6 * max(1.0, np.abs(v).max())


@pytest.mark.parametrize("integrator", [0, 1, 2, 3, 5, 6, 7])
def test_mixed_edges_batch_vs_oracle(oracle_mod, integrator):
    """MultipleShootingEdgeSingleControl (multiple_shooting_edges.h:151-303): different instances, per-instance multipliers, every shooting
    integrator -- Hessian lists, linear form, gradient and objective value of every instance against the oracle at its own point."""
    from control_box_rst_amd import capi
    d = problems.hessian_path_cost_form(problems.unicycle_desc(N=14), integral="trapezoidal")
    d.grid, d.defect, d.shooting_integrator = capi.GRID_MS, capi.DEFECT_RK4_SHOOTING, integrator
    B = 6
    rng = np.random.default_rng(40 + integrator)
    x0, xf = problems.unicycle_instances(B)
    s = BatchedLevenbergMarquardt(d, B)
    X = s.init_trajectory(x0, xf) + 0.05 * rng.normal(size=(B, s.dims.nv))
    X[:, :d.nx] = x0
    s.set_instance_data(X, xref=xf)
    me = rng.uniform(0.2, 1.0, (B, s.dims.eq))
    p = oracle_mod.OracleProblem(d)
    for lower in (True, False):
        st = s.hessian_structure(lower)
        vals = s.eval_hessians(lower, 0.8, me, None)
        for b in range(B):
            p.set_data(X[b], xref=xf[b])
            ref = p.hessians(1 if lower else 0, 0.8, me[b], None)
            for c in range(3):
                assert np.array_equal(st[c][0], ref[c][0]) and np.array_equal(st[c][1], ref[c][1])
                if len(ref[c][2]):
                    assert np.abs(vals[c][b] - ref[c][2]).max() <= REL * max(1.0, np.abs(ref[c][2]).max()), (b, c)
    rows, cols, vals, lbA, ubA = s.linear_form()
    grad, obj = s.objective_gradient()
    for b in range(B):
        p.set_data(X[b], xref=xf[b])
        r, c, v, l, u = p.linear_form()
        assert np.array_equal(rows, r) and np.array_equal(cols, c)
        assert np.abs(vals[b] - v).max() <= 2e-6 * max(1.0, np.abs(v).max())
        fin = np.abs(l) < 1e29
        assert np.abs(lbA[b][fin] - l[fin]).max() <= 1e-12 * max(1.0, np.abs(l[fin]).max())
        p.set_data(X[b], xref=xf[b])
        go, oo = p.objective_gradient()
        assert np.abs(grad[b] - go).max() <= 1e-6 * max(1.0, np.abs(go).max())
        assert abs(obj[b] - oo) <= 1e-13 * max(1.0, abs(oo))
    with pytest.raises(Exception):   # a mixed edge with a plain objective part is no least-squares problem: refused like LevenbergMarquardtSparse::solve does
        s.solve(new_run=True)


def test_hessians_big_block_family_vs_oracle(oracle_mod):
    """Quadrotor (nx = 12, multiple shooting + RK4, control bounds, keep-out ball): the same kernel, 12 x 12 blocks."""
    d = problems.quad_desc(N=8)
    B = 3
    rng = np.random.default_rng(9)
    x0, xf = problems.quad_instances(B)
    s = BatchedLevenbergMarquardt(d, B)
    X = s.init_trajectory(x0, xf) + 0.02 * rng.normal(size=(B, s.dims.nv))
    X[:, :d.nx] = x0
    s.set_instance_data(X, xref=xf)
    me = rng.uniform(0.2, 1.0, (B, s.dims.eq))
    mi = rng.uniform(0.1, 0.5, (B, s.dims.ineq))
    vals = s.eval_hessians(True, 1.3, me, mi)
    st = s.hessian_structure(True)
    rows, cols, lv, lbA, ubA = s.linear_form()
    p = oracle_mod.OracleProblem(d)
    for b in range(B):
        p.set_data(X[b], xref=xf[b])
        ref = p.hessians(1, 1.3, me[b], mi[b])
        for c in range(3):
            assert np.array_equal(st[c][0], ref[c][0]) and np.array_equal(st[c][1], ref[c][1])
            if len(ref[c][2]):
                assert np.abs(vals[c][b] - ref[c][2]).max() <= REL * max(1.0, np.abs(ref[c][2]).max()), (b, c)
        p.set_data(X[b], xref=xf[b])
        r2, c2, v2, l2, u2 = p.linear_form()
        assert np.array_equal(rows, r2) and np.array_equal(cols, c2)
        assert np.abs(lv[b] - v2).max() <= 2e-6 * max(1.0, np.abs(v2).max())
        fin = np.isfinite(l2) & (np.abs(l2) < 1e29)
        assert np.abs(lbA[b][fin] - l2[fin]).max() <= 1e-12 * max(1.0, np.abs(l2[fin]).max())


@pytest.mark.parametrize("seed", range(48))
def test_random_descriptor_hessians_vs_oracle(oracle_mod, seed):
    """The random descriptors of tests/test_gpu_fuzz.py (every small-block family, grid kind, defect formula, bound pattern, partially
    fixed x_f, cost terms on / off, stage inequality, final-stage constraints): structure of the three lists and of the linear form
    identical to the oracle's, values within the reference's own consecutive-call spread, per-instance multipliers."""
    from test_gpu_fuzz import random_desc
    from control_box_rst_amd import capi
    rng = np.random.default_rng(31000 + seed)
    fam, d = random_desc(rng)
    if seed % 3 == 0:
        d.cost_nonlsq = 1
        if d.grid == capi.GRID_FD and d.stage_cost == capi.COST_QUADRATIC_LSQ and seed % 2 == 0:
            d.cost_integral = 1 + (seed // 6) % 2   # QuadraticFormCost in integral form: trapezoidal / left-sum cost edge per interval   # the same terms as plain objective edges (lsq_form = false): what the Hessian path is for
        elif d.grid == capi.GRID_MS and d.stage_cost == capi.COST_QUADRATIC_LSQ and seed % 2 == 0 and not d.stage_ineq and not (d.weights_dense & 3):
            d.cost_integral = 1                     # the same on the shooting grid: one MultipleShootingEdgeSingleControl (mixed edge) per interval
        elif d.grid == capi.GRID_FD_VARIABLE and d.stage_cost == capi.COST_MIN_TIME_QUADRATIC_LSQ and seed % 2 == 0:
            d.cost_integral = 1 + (seed // 6) % 2   # MinTimeQuadratic in integral form: plain dt terms + integral edges (only_last_n as drawn)
    _check_hessian_operators(oracle_mod, d, fam, rng, seed)


@pytest.mark.parametrize("seed", range(24))
def test_random_descriptor_hessians_partial_terminal_equality(oracle_mod, seed):
    """The same operators with a TerminalPartialEqualityConstraint on a random subset of the components of x_f (equality rows, multipliers and
    linear-form rows for the active components only): random descriptors of the fixed-dt families, own generator for the mask."""
    from test_gpu_fuzz import random_desc
    from control_box_rst_amd import capi
    rng = np.random.default_rng(47000 + seed)
    while True:
        fam, d = random_desc(rng)
        if fam not in ("dint", "int3t") and d.xf_fixed_mask != 2 ** d.nx - 1:
            break
    d.final_ineq = capi.FINAL_INEQ_NONE
    d.final_eq = 1
    d.final_eq_mask = int(rng.integers(1, 2 ** d.nx))
    if seed % 3 == 0:
        d.cost_nonlsq = 1
    _check_hessian_operators(oracle_mod, d, fam, rng, seed)


def _check_hessian_operators(oracle_mod, d, fam, rng, seed):
    from control_box_rst_amd import capi
    B = 2
    x0 = rng.uniform(-1, 1, (B, d.nx))
    xf = rng.uniform(-1, 1, (B, d.nx)) + (np.array([1.5, 0.5, 0.2, 0.0])[: d.nx] if fam not in ("dint", "int3t") else np.array([1.0, 0.0, 0.0])[: d.nx])
    if fam == "rocket":
        x0[:, 2] = rng.uniform(0.9, 1.1, B)
        xf[:, 2] = rng.uniform(0.8, 1.0, B)
    s = BatchedLevenbergMarquardt(d, B)
    X0 = s.init_trajectory(x0, xf) + 0.05 * rng.normal(size=(B, s.dims.nv))
    X0[:, : d.nx] = x0
    if d.grid in (capi.GRID_FD_VARIABLE, capi.GRID_MS_VARIABLE):
        X0[:, -1] = d.dt_ref
    s.set_instance_data(X0, xref=xf)
    lower = bool(seed % 2)
    me = rng.uniform(0.2, 1.0, (B, s.dims.eq))
    mi = rng.uniform(0.1, 0.6, (B, max(1, s.dims.ineq)))[:, : s.dims.ineq]
    mobj = float(rng.uniform(0.5, 2.0))
    st = s.hessian_structure(lower)
    vals = s.eval_hessians(lower, mobj, me, mi if s.dims.ineq else None)
    rows, cols, lv, lbA, ubA = s.linear_form()
    grad, obj = s.objective_gradient()
    for b in range(B):
        p = oracle_mod.OracleProblem(d)
        p.set_data(X0[b], xref=xf[b])
        ref = p.hessians(int(lower), mobj, me[b], mi[b] if s.dims.ineq else None)
        again = p.hessians(int(lower), mobj, me[b], mi[b] if s.dims.ineq else None)   # the reference's own reproducibility: a second call, from the drifted point
        for c in range(3):
            assert np.array_equal(st[c][0], ref[c][0]) and np.array_equal(st[c][1], ref[c][1]), (seed, fam, c)
            if len(ref[c][2]):
                scale = max(1.0, np.abs(ref[c][2]).max())
                own = np.abs(again[c][2] - ref[c][2]).max() / scale
                # (tools/fuzz_campaign.py, 1200 more seeds: one case beyond REL -- a cart-pole entry the oracle itself reproduces to 3.4e-4 only)
                assert np.abs(vals[c][b] - ref[c][2]).max() <= max(REL, 4.0 * own) * scale, (seed, fam, b, c, own)
        p.set_data(X0[b], xref=xf[b])
        go, oo = p.objective_gradient()
        assert np.abs(grad[b] - go).max() <= 1e-6 * max(1.0, np.abs(go).max()), (seed, fam, b)
        assert abs(obj[b] - oo) <= 1e-12 * max(1.0, abs(oo)), (seed, fam, b)
        p.set_data(X0[b], xref=xf[b])
        r2, c2, v2, l2, u2 = p.linear_form()
        assert np.array_equal(rows, r2) and np.array_equal(cols, c2), (seed, fam)
        assert np.abs(lv[b] - v2).max() <= 2e-6 * max(1.0, np.abs(v2).max()), (seed, fam, b)
        fin = np.abs(l2) < 1e29
        assert np.array_equal(np.abs(lbA[b]) < 1e29, fin)
        if fin.any():
            assert np.abs(lbA[b][fin] - l2[fin]).max() <= 1e-11 * max(1.0, np.abs(l2[fin]).max()), (seed, fam, b)
        fu = np.abs(u2) < 1e29
        if fu.any():
            assert np.abs(ubA[b][fu] - u2[fu]).max() <= 1e-11 * max(1.0, np.abs(u2[fu]).max()), (seed, fam, b)


def test_a_problem_with_plain_objective_edges_is_refused_by_the_lm_entries():
    """cost_nonlsq: not a least-squares problem -- LevenbergMarquardtSparse::solve returns Error for it (levenberg_marquardt_sparse.cpp:48-55),
    corbo_hip_solve / corbo_hip_eval say so; the Hessian-path operators are what such a handle is for."""
    from control_box_rst_amd.solver import CorboHipError
    g = load_golden("hess_vdp_nonlsq")
    d, s = device_at_point(g, B=2)
    assert d.cost_nonlsq == 1 and s.dims.lsq == 0
    with pytest.raises(CorboHipError, match="least-squares"):
        s.solve()
    with pytest.raises(CorboHipError, match="least-squares"):
        s.eval()
    grad, obj = s.objective_gradient()
    assert np.isfinite(grad).all() and abs(obj[0] - g["obj_value"]) <= 1e-12 * abs(g["obj_value"])


def test_hessian_views_equal_the_copied_lists():
    """corbo_hip_eval_hessians_views (pinned-host views, no copy into caller arrays) returns the same numbers as corbo_hip_eval_hessians."""
    from control_box_rst_amd import problems
    from control_box_rst_amd.solver import BatchedLevenbergMarquardt
    d = problems.unicycle_desc(N=24)
    B = 7
    x0, xf = problems.unicycle_instances(B)
    s = BatchedLevenbergMarquardt(d, B)
    rng = np.random.default_rng(5)
    X = s.init_trajectory(x0, xf) + 0.02 * rng.normal(size=(B, s.dims.nv))
    X[:, : d.nx] = x0
    s.set_instance_data(X, xref=xf)
    me = rng.uniform(0.2, 1.0, (B, s.dims.eq))
    for lower in (True, False):
        a = s.eval_hessians(lower, 1.3, me, None)
        v = [np.array(q) for q in s.eval_hessians_views(lower, 1.3, me, None)]
        for c in range(3):
            assert a[c].shape[1] == 0 or np.array_equal(a[c], v[c]), (lower, c)
        # twice in a row: the cached structure and the grown buffers are reused
        a2 = s.eval_hessians(lower, 1.3, me, None)
        for c in range(3):
            assert np.array_equal(a[c], a2[c])


@pytest.mark.parametrize("name", ["hess_unicycle_n24_ball", "hess_pquad_n5", "hess_unicycle_ms_integral", "hess_dint_mtq"])
def test_hessian_work_splits_agree(name):
    """hessian_kernel spreads a stage's blocks over waves by batch size (HessParams::split: per stage / per (edge, row vertex) / per block; option
    hess_split): the three splits evaluate the same finite-difference nest from the pristine point of their part of the walk instead of the one the
    earlier parts left a few ulps away (the in-place perturbations' drift): the lists agree within the operators' tolerance, 1e-6 typically."""
    g = load_golden(name)
    d, s = device_at_point(g, B=5)
    me = np.array(g["mult_eq"]); mi = np.array(g["mult_ineq"]) if g["mult_ineq"] else None
    ref = None
    for split in (0, 1, 2):
        s.set_option("hess_split", split)
        for lower in (False, True):
            vals = s.eval_hessians(lower, g["mult_obj"], me, mi)
            if split == 0:
                ref = ref or {}
                ref[lower] = vals
            else:
                for a, b in zip(vals, ref[lower]):
                    if a.size:
                        assert np.abs(a - b).max() <= REL * max(1.0, np.abs(b).max()), (name, split, lower)
