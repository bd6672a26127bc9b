import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a box without a HIP device skips the gpu-marked tests instead of failing them."""
    def have_gpu():
        try:
            import torch
            return torch.cuda.is_available()
        except Exception:
            return False
    if config.getoption("-m") and "gpu" in config.getoption("-m") and "not gpu" not in config.getoption("-m"):
        return   # -m gpu was asked for explicitly: let the tests speak (they fail loudly without a device)
    if have_gpu():
        return
    skip = pytest.mark.skip(reason="needs a HIP device (run with -m gpu on the GPU box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as O
    O.build()
    O.load()
    return O


def desc_for(g):
    """Descriptor matching a golden fixture's header."""
    from control_box_rst_amd import capi, problems
    defect = {"forward": capi.DEFECT_FORWARD, "backward": capi.DEFECT_BACKWARD, "midpoint": capi.DEFECT_MIDPOINT,
              "crank_nicolson": capi.DEFECT_CRANK_NICOLSON}[g.get("collocation", "crank_nicolson")]
    sc = g["scenario"]
    def cost_option(d):   # cost=mtq of oracle/ref_driver.cpp: MinTimeQuadratic, Q = diag(1, 0.5, 0.2, 0.1)[:nx], R = diag(0.1, 0.2, 0.05)[:nu]
        if g.get("cost") == "mtq":
            problems.min_time_quadratic(d, (1.0, 0.5, 0.2, 0.1)[: d.nx], (0.1, 0.2, 0.05)[: d.nu], only_last_n=g.get("last_n", 0))
        else:
            assert "cost" not in g, g["cost"]
        if g.get("lsq") == 0:   # lsq_form = false for every cost term (Hessian-path fixtures): plain objective edges
            d.cost_nonlsq = 1
        if "integral" in g:     # QuadraticFormCost in integral form + the grid's cost integration rule
            d.cost_integral = {"trap": 1, "left": 2}[g["integral"]]
        return d
    if sc == "dint":
        return cost_option(problems.dint_desc(N=g["N"], dt=g["dt"], shooting=(g.get("grid") == "ms")))
    if sc == "quad":
        d = problems.quad_desc(N=g["N"], dt=g["dt"], time_optimal=bool(g.get("vargrid")))
    elif sc == "pquad":   # the big-block user-model example (csrc/models/planar_quadrotor.hpp)
        d = problems.planar_quadrotor_desc(N=g["N"], dt=g["dt"], time_optimal=bool(g.get("vargrid")), shooting=(g.get("grid") != "fd"))
    elif sc == "int3":
        d = problems.int3_desc(N=g["N"], dt=g["dt"], defect=defect, time_optimal=bool(g.get("vargrid")))
    elif sc == "lin":
        import numpy as np
        d = problems.linear_desc(np.array(g["lin_a"]).reshape(g["nx"], g["nx"]), np.array(g["lin_b"]).reshape(g["nx"], g["nu"]),
                                 N=g["N"], dt=g["dt"], defect=defect)
    elif sc in ("par2", "par3"):
        d = problems.parallel_integrator_desc(int(sc[-1]), N=g["N"], dt=g["dt"], defect=defect)
    elif sc in problems.BENCHMARK_SYSTEMS:
        d = problems.benchmark_desc(sc, N=g["N"], dt=g["dt"], defect=defect)
    elif sc == "kcar":   # the user-model example (csrc/models/kinematic_car.hpp) in the unicycle's OCP
        d = problems.kinematic_car_desc(N=g["N"], dt=g["dt"], defect=defect)
    elif sc in ("unicycle", "vdp"):
        d = (problems.unicycle_desc if sc == "unicycle" else problems.vdp_desc)(N=g["N"], dt=g["dt"], defect=defect)
    else:
        raise KeyError(sc)
    # options of oracle/ref_driver.cpp recorded in the fixture header
    if g.get("grid") == "fd":   # scenarios whose default is the shooting grid (quad, pquad) on the FiniteDifferencesGrid (vargrid: ...VariableGrid)
        d.grid, d.defect = (capi.GRID_FD_VARIABLE if g.get("vargrid") else capi.GRID_FD), defect
    if g.get("grid") == "ms":   # MultipleShootingGrid (vargrid: MultipleShootingVariableGrid, free dt) + RK4
        d.grid, d.defect = (capi.GRID_MS_VARIABLE if g.get("vargrid") else capi.GRID_MS), capi.DEFECT_RK4_SHOOTING
    if "ms_integrator" in g:    # IntegratorExplicitEuler / RungeKutta2 / RungeKutta3 on the shooting grid
        d.shooting_integrator = {"euler": 1, "rk2": 2, "rk3": 3, "rk5": 5, "rk6": 6, "rk7": 7}[g["ms_integrator"]]
    if "xlb" in g or "ulb" in g:   # setBounds replaces all four vectors: the ones not given are unbounded
        for key, arr, n in (("xlb", d.x_lb, d.nx), ("xub", d.x_ub, d.nx), ("ulb", d.u_lb, d.nu), ("uub", d.u_ub, d.nu)):
            vals = g.get(key)
            for i in range(n):
                v = vals[i] if vals is not None else (-2e30 if key.endswith("lb") else 2e30)
                arr[i] = -capi.INF if v <= -2e30 else (capi.INF if v >= 2e30 else v)
    if "xf_fixed" in g:
        d.xf_fixed_mask = g["xf_fixed"]
    if "final_cost" in g:
        d.final_cost = g["final_cost"]
    if "tball_s" in g:          # TerminalBall(S, gamma)
        d.final_ineq = capi.FINAL_INEQ_TERMINAL_BALL
        for i, v in enumerate(list(g["tball_s"]) + [g["tball_gamma"]]):
            d.final_ineq_params[i] = v
    if g.get("teq"):            # TerminalEqualityConstraint(xf); teq_mask: TerminalPartialEqualityConstraint (active components)
        d.final_eq = 1
        d.final_eq_mask = g.get("teq_mask", 0)
    cost_option(d)
    if g.get("fullq"):          # non-diagonal weights: the reference's upper Cholesky factors as the fixture records them
        nx, nu = d.nx, d.nu
        d.weights_dense = 5 | (2 if "r_sqrt" in g else 0)
        for i, v in enumerate(g["q_sqrt"]): d.q_sqrt[i] = v
        for i, v in enumerate(g["qf_sqrt"]): d.qf_sqrt[i] = v
        for i, v in enumerate(g.get("r_sqrt", [])): d.r_sqrt[i] = v
    for key, arr in (("qdiag", "q_diag"), ("rdiag", "r_diag"), ("qfdiag", "qf_diag")):   # diagonal weights other than the scenario's (fuzz_* fixtures)
        for i, v in enumerate(g.get(key, [])):
            getattr(d, arr)[i] = v
    if g.get("noball"):         # quad / pquad without the keep-out ball their scenarios carry by default
        d.stage_ineq = capi.INEQ_NONE
    if "ball" in g:             # BallKeepOut stage inequality
        d.stage_ineq = capi.INEQ_BALL
        for i, v in enumerate(g["ball"]):
            d.ineq_params[i] = v
    if "crule" in g:            # the grid's integration rule for integral-form constraint edges (user stage functions of oracle/ref_driver.cpp)
        d.constraint_integration = {"trap": capi.RULE_TRAPEZOIDAL, "left": capi.RULE_LEFT_SUM}[g["crule"]]
    if g.get("ball_int"):       # the ball as the stage inequalities' INTEGRAL state-control term
        d.stage_ineq_integral = 1
    if "eq_lin" in g:           # LinearIntegralEquality a^T x + b^T u - c
        d.stage_eq = capi.STAGE_EQ_LINEAR
        for i, v in enumerate(g["eq_lin"]):
            d.stage_eq_params[i] = v
    if "tilt" in g:             # user stage function, slot 0 (csrc/stage_functions/tilt_cone.hpp): the stage inequalities' state term
        d.stage_ineq = capi.STAGE_FN_USER + 0
        for i in range(8):
            d.ineq_params[i] = 0.0
        d.ineq_params[0] = g["tilt"]
    if "unorm" in g:            # user stage function, slot 1 (control_norm.hpp): their control term
        d.stage_ineq_control = capi.STAGE_FN_USER + 1
        d.ineq_control_params[0] = g["unorm"]
    if "rate" in g:             # input-rate limit as the control-deviation term of the stage inequalities
        d.ctrl_dev = capi.CTRL_DEV_RATE
        for i, v in enumerate(g["rate"]):
            d.ctrl_dev_params[i] = v
    return d


def load_npz(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN, name + ".npz"))


SOFT_EIGENVALUE = 0.2   # eigen-directions of J^T J with lambda <= this are "soft" (cfg 5: thrust / rate / torque cost weights 0.01 .. 0.1)


def stiff_part(J, dp, lam=SOFT_EIGENVALUE):
    """Split a parameter-space difference dp by the eigen-directions of H = J^T J: returns (component of dp in the directions with
    eigenvalue > lam, |J dp|^2).  Differences confined to the soft directions change chi2 by ~lam |dp|^2: they are what the
    reference's own finite-difference noise (delta = 1e-9) leaves undetermined -- see tests/golden/quad_n40_seeded_ulp.json."""
    import numpy as np
    H = (J.T @ J).toarray()
    w, V = np.linalg.eigh(H)
    c = V.T @ dp
    stiff = V[:, w > lam] @ c[w > lam]
    return stiff, float(np.linalg.norm(J @ dp) ** 2)


# ---- tolerance ledger (tests/tolerances.json, generated by `python oracle/gen_golden.py tolerances` from the genuine reference):
#      fixture -> tolerance of the device result against the fixture + the reference's own one-ulp reproducibility on that fixture
with open(os.path.join(ROOT, "tests", "tolerances.json")) as _f:
    LEDGER = json.load(_f)


def ledger_rule_violations(ledger=None):
    """Entries whose tolerance is wider than the rule allows: beyond the hard gate a tolerance must be covered by 4 x the recorded spread."""
    L = ledger or LEDGER
    bad = []
    for name, e in L["fixtures"].items():
        if e["x_tol"] > max(L["hard_x"], _round_up(4.0 * e["ref_spread_x"])):
            bad.append((name, "x_tol", e["x_tol"], e["ref_spread_x"]))
        if e["chi2_rtol"] > max(L["hard_chi2"], _round_up(4.0 * e["ref_spread_chi2"])):
            bad.append((name, "chi2_rtol", e["chi2_rtol"], e["ref_spread_chi2"]))
    return bad


def _round_up(v):
    """two significant digits, upwards (how the generator rounds 4 x spread: a tolerance is at most 4.4 x the spread)"""
    import math
    if v <= 0:
        return 0.0
    e = 10.0 ** (math.floor(math.log10(v)) - 1)
    return math.ceil(v / e - 1e-9) * e


def ledger_tolerances(name):
    """(x_tol, chi2_rtol) of a fixture; the defaults for fixtures the ledger does not list.  Fails when the entry breaks the ledger's rule."""
    e = LEDGER["fixtures"].get(name)
    if e is None:
        return LEDGER["default_x_tol"], LEDGER["default_chi2_rtol"]
    bad = [b for b in ledger_rule_violations() if b[0] == name]
    assert not bad, f"tests/tolerances.json: tolerance wider than max(hard gate, 4 x the reference's own spread): {bad}"
    return e["x_tol"], e["chi2_rtol"]
