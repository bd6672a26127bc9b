"""Exact-Hessian path, CPU: the oracle's restatement of computeSparseHessians{NNZ,Structure,Values}
(hyper_graph_optimization_problem_edge_based.cpp:2087-3760, BaseEdge::computeHessian edge_interface.cpp:151-255) and of the
two-side-bounded linear form (:4762-4968, optimization_problem_interface.cpp:1141-1183) against outputs of the genuine reference
(tests/golden/hess_*.json, oracle/gen_golden.py hess) -- structure and VALUES bit for bit, in the reference's call order: the finite
differences perturb the vertices in place, so the full pass, the lower-part pass and the linear form each start from the point the
previous one left (a few ulps off), exactly as in the fixture."""
import numpy as np
import pytest

from conftest import desc_for, load_golden

HESS = ["hess_kcar", "hess_pquad_n5", "hess_pquad_fd_n5", "hess_unicycle_fullq", "hess_vdp", "hess_vdp_forward", "hess_vdp_backward", "hess_vdp_midpoint", "hess_vdp_teq", "hess_dint", "hess_int3_time_optimal",
        "hess_unicycle_n16", "hess_unicycle_xf_fixed", "hess_unicycle_n24_ball", "hess_pendulum_ms_rk4", "hess_pendulum_ms_rk5", "hess_cartpole", "hess_quad_n4", "hess_int3_ms_time_optimal",
        "hess_dint_mtq", "hess_int3_ms_mtq", "hess_dint_mtq_last5",
        "hess_vdp_nonlsq", "hess_unicycle_nonlsq", "hess_unicycle_nonlsq_tball", "hess_dint_nonlsq", "hess_dint_mtq_nonlsq", "hess_int3_ms_nonlsq",
        "hess_vdp_integral_trap", "hess_unicycle_integral_trap", "hess_unicycle_integral_left",
        # MultipleShootingEdgeSingleControl (a mixed edge: integrated cost + defect, multiple_shooting_edges.h:151-303) -- shooting grid + integral-form cost
        "hess_unicycle_ms_integral", "hess_unicycle_ms_integral_xf_fixed", "hess_vdp_ms_integral_euler", "hess_vdp_ms_integral_rk3", "hess_unicycle_ms_integral_rk5",
        "hess_unicycle_ms_integral_teq", "hess_unicycle_ms_integral_tball",
        "hess_pquad_ms_integral", "hess_pquad_ms_integral_rk3", "hess_quad_ms_integral",   # the mixed edge around the big-block models (nx = 6, 12)
        # TerminalPartialEqualityConstraint on the Hessian path: rows / multipliers for the active components of x_f only
        "hess_unicycle_pteq", "hess_cartpole_pteq", "hess_unicycle_ms_pteq", "hess_unicycle_ms_integral_pteq", "hess_pquad_pteq",
        "hess_dint_mtq_integral_trap", "hess_dint_mtq_integral_left_last4"]   # MinTimeQuadratic in integral form (free dt)   # integral-form cost: one objective edge per interval   # *_nonlsq: plain (non-least-squares) objective edges
KEYS = ("hobj", "heq", "hineq")


def problem_at_point(oracle_mod, g):
    d = desc_for(g)
    p = oracle_mod.OracleProblem(d)
    p.set_data(np.array(g["vertex_point"])[:p.dims.nv], xref=np.array(g["xf"]))
    assert p.dims.n == g["n"] and p.dims.eq == g["eq"] and p.dims.ineq == g["ineq"] and p.dims.bounds == g["bounds"]
    return d, p


@pytest.mark.parametrize("name", HESS)
def test_hessian_triplets_and_linear_form_bit_exact(oracle_mod, name):
    g = load_golden(name)
    d, p = problem_at_point(oracle_mod, g)
    for lower, tag in ((0, "full"), (1, "lower")):
        trip = p.hessians(lower, g["mult_obj"], g["mult_eq"], g["mult_ineq"])
        for (r, c, v), key in zip(trip, KEYS):
            assert np.array_equal(r, np.array(g[f"{key}_rows_{tag}"], np.int32)), (name, tag, key)
            assert np.array_equal(c, np.array(g[f"{key}_cols_{tag}"], np.int32)), (name, tag, key)
            assert np.array_equal(v, np.array(g[f"{key}_vals_{tag}"])), (name, tag, key, np.abs(v - np.array(g[f"{key}_vals_{tag}"])).max())
    r, c, v, lbA, ubA = p.linear_form()
    assert np.array_equal(r, g["lin_rows"]) and np.array_equal(c, g["lin_cols"])
    assert np.array_equal(v, np.array(g["lin_vals"]))
    assert np.array_equal(lbA, np.array(g["lin_lbA"])) and np.array_equal(ubA, np.array(g["lin_ubA"]))
    # the first-order callbacks of IpoptWrapper: eval_grad_f / eval_f (computeGradientObjective, computeValueObjective) ...
    grad, obj = p.objective_gradient()
    assert np.array_equal(grad, np.array(g["grad_obj"])), np.abs(grad - np.array(g["grad_obj"])).max()
    assert abs(obj - g["obj_value"]) <= 4e-16 * abs(g["obj_value"])   # (Eigen's squaredNorm sums in packets)
    # ... eval_g / eval_jac_g = the linear form: constraint values are -lbA (equalities) / -ubA (inequalities), the Jacobian list is the
    # linear form's without the bound rows, same entry order
    r2, c2, v2, l2, u2 = p.linear_form()
    eq, ineq = p.dims.eq, p.dims.ineq
    gv = np.array(g["g_values"])
    # (values: the oracle evaluates them behind the Jacobian blocks of the same call, the fixture before -- a few ulps of drift apart)
    assert np.abs(-l2[:eq] - gv[:eq]).max() <= 1e-14 * max(1.0, np.abs(gv).max())
    if ineq:
        assert np.abs(-u2[eq:eq + ineq] - gv[eq:]).max() <= 1e-14 * max(1.0, np.abs(gv).max())
    nj = len(g["jacg_vals"])
    assert np.array_equal(r2[:nj], g["jacg_rows"]) and np.array_equal(c2[:nj], g["jacg_cols"]) and np.array_equal(v2[:nj], np.array(g["jacg_vals"]))
    assert np.array_equal(p.x(), np.array(g["vertex_after"])[:p.dims.nv])   # the drift of the in-place perturbations, reproduced


@pytest.mark.parametrize("name", ["hess_unicycle_n16", "hess_dint", "hess_unicycle_n24_ball"])
def test_hessian_is_what_it_says(oracle_mod, name):
    """Assembled from the triplets (the reference lists off-diagonal vertex pairs row-major but fills them column-major: assemble the
    way the VALUES are laid out), the equality Hessian is sum_i lambda_i grad^2 c_i: symmetric up to the forward-difference error, and
    equal to a second difference of the multiplier-weighted constraint values computed independently here."""
    g = load_golden(name)
    d, p = problem_at_point(oracle_mod, g)
    n = p.dims.n
    x0 = p.x()
    trip = p.hessians(0, g["mult_obj"], g["mult_eq"], g["mult_ineq"])
    p.set_data(x0, xref=np.array(g["xf"]))
    lam = np.array(g["mult_eq"])
    H = np.zeros((n, n))
    r, c, v = trip[1]
    # blocks = runs of one (vertex, vertex) pair; listed row-major, filled column-major
    offs = p.param_offsets()
    S = d.nx + d.nu
    vert = np.array([(-1 if o == (d.N - 1) * S + d.nx else 2 * (o // S) + (1 if o % S >= d.nx else 0)) for o in offs])
    k = 0
    while k < len(v):
        e = k
        while e < len(v) and vert[r[e]] == vert[r[k]] and vert[c[e]] == vert[c[k]] and (e == k or (r[e], c[e]) > (r[e - 1], c[e - 1])):
            e += 1
        ni, nj = len(set(r[k:e])), len(set(c[k:e]))
        assert ni * nj == e - k
        H[r[k]:r[k] + ni, c[k]:c[k] + nj] += v[k:e].reshape(nj, ni).T
        k = e
    assert np.abs(H - H.T).max() <= 2e-2 * max(1.0, np.abs(H).max())   # forward differences with h = 1e-2
    # independent check of a few entries: d/dp_j of (lambda^T J)_i by a forward difference of the oracle's own Jacobian
    rows, cols = p.structure()
    import scipy.sparse as sp
    lsq, eq = p.dims.lsq, p.dims.eq
    def lamJ(xv):
        p.set_data(xv, xref=np.array(g["xf"]))
        _, jac = p.eval(1.0, 1.0, 1.0)
        J = sp.csr_matrix((jac, (rows, cols)), shape=(p.dims.m, n)).toarray()
        return lam @ J[lsq:lsq + eq]
    base = lamJ(x0)
    for j in (0, n // 3, n // 2, n - 1):
        xv = x0.copy()
        xv[offs[j]] += 1e-2
        col = (lamJ(xv) - base) / 1e-2
        assert np.abs(col - H[:, j]).max() <= 1e-4 * max(1.0, np.abs(H).max()), (name, j, np.abs(col - H[:, j]).max())
