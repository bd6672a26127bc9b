"""The adapter's graph -> descriptor recogniser (control_box_rst_amd/adapter/graph_recogniser.cpp) on hypergraphs built by the GENUINE
reference (oracle/_ref/dropin_demo describe: StructuredOptimalControlProblem + the reference's grids, dynamics and cost classes, a
solver stub that only runs the recogniser).  CPU only: the reference's own classes need no device.  Skipped when the binary has not been
built (needs /root/reference at build time; it travels to the GPU box)."""
import json
import math
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from control_box_rst_amd import capi

DEMO = os.path.join(ROOT, "oracle", "_ref", "dropin_demo")
pytestmark = pytest.mark.skipif(not os.path.exists(DEMO), reason="oracle/_ref/dropin_demo not built (needs /root/reference at build time)")


@pytest.fixture(scope="module")
def described():
    p = subprocess.run([DEMO, "describe"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    return {r["scenario"]: r for r in (json.loads(l) for l in p.stdout.splitlines() if l.startswith("{"))}


def _sqrt_equal(q, expected):
    """The recogniser recovers sqrt(Q_ii) exactly; the descriptor stores its square: compare what the device will use."""
    return all(math.sqrt(a) == math.sqrt(b) for a, b in zip(q, expected))


def test_reference_benchmark_classes_with_private_parameters(described):
    nx2 = list(np.linspace(1.0, 0.3, 2))
    v = described["vdp"]
    assert v["recognised"] == 1 and (v["grid"], v["defect"], v["dynamics"]) == (capi.GRID_FD, capi.DEFECT_CRANK_NICOLSON, capi.DYN_VAN_DER_POL)
    assert v["dyn_params"][0] == 1.3 and v["N"] == 20 and (v["stage_cost"], v["final_cost"]) == (capi.COST_QUADRATIC_LSQ, 1)
    assert _sqrt_equal(v["q_diag"], nx2) and _sqrt_equal(v["r_diag"], [0.2])   # (Eigen::LinSpaced of size 1 is its upper end) and _sqrt_equal(v["qf_diag"], [7.0 * a for a in nx2])
    assert v["xref"] == [0.2, -0.1]                      # the StaticReference the OCP was given, read out of the cost edges bit for bit
    d = described["duffing"]                             # private members _damping / _spring_alpha / _spring_beta; midpoint collocation
    assert d["recognised"] == 1 and d["dynamics"] == capi.DYN_DUFFING and d["defect"] == capi.DEFECT_MIDPOINT
    assert d["dyn_params"][:3] == [0.7, 1.1, 0.9]
    pd = described["pendulum"]                           # + TerminalEqualityConstraint
    assert pd["recognised"] == 1 and pd["dynamics"] == capi.DYN_SIMPLE_PENDULUM and pd["final_eq"] == 1
    assert pd["dyn_params"][:4] == [0.3, 0.5, 9.81, 0.02]


def test_time_optimal_variable_grid_and_shooting_grid(described):
    t = described["dint"]                                # cfg 2: FiniteDifferencesVariableGrid, MinimumTime(lsq) with its duplicated dt edge
    assert t["recognised"] == 1 and t["grid"] == capi.GRID_FD_VARIABLE and t["stage_cost"] == capi.COST_MIN_TIME_LSQ and t["final_cost"] == 0
    assert t["dynamics"] == capi.DYN_SERIAL_INTEGRATOR and t["dyn_params"][0] == 1.0 and t["N"] == 50
    m = described["dint_ms"]                             # the same on the MultipleShootingVariableGrid (told apart from the fixed shooting grid by RTTI)
    assert m["recognised"] == 1 and (m["grid"], m["defect"], m["stage_cost"]) == (capi.GRID_MS_VARIABLE, capi.DEFECT_RK4_SHOOTING, capi.COST_MIN_TIME_LSQ)
    h = described["dint_mtq"]                            # MinTimeQuadratic: state, control and minimum-time terms (hybrid_cost.h:189-303)
    assert h["recognised"] == 1 and h["grid"] == capi.GRID_FD_VARIABLE and h["stage_cost"] == capi.COST_MIN_TIME_QUADRATIC_LSQ
    assert _sqrt_equal(h["q_diag"], [1.0, 0.5]) and _sqrt_equal(h["r_diag"], [0.1]) and h["final_cost"] == 0
    h8 = described["dint_mtq8"]                          # only_last_n = 8: the quadratic terms on the last intervals only
    assert h8["recognised"] == 1 and h8["stage_cost"] == capi.COST_MIN_TIME_QUADRATIC_LSQ and h8["quad_first_interval"] == h8["N"] - 8
    assert h["quad_first_interval"] == 0
    hs = described["dint_mtqs"]                          # MinTimeQuadraticStates with a diagonal Q: the reference creates no state term
    assert hs["recognised"] == 1 and hs["stage_cost"] == capi.COST_MIN_TIME_LSQ   # (quadratic_state_cost.cpp:33-62) -- the graph is MinimumTime's
    l = described["lin32"]                               # LinearStateSpaceModel on the MultipleShootingGrid with RK4
    assert l["recognised"] == 1 and (l["grid"], l["defect"], l["dynamics"]) == (capi.GRID_MS, capi.DEFECT_RK4_SHOOTING, capi.DYN_LINEAR_STATE_SPACE)
    assert l["lin_a"] == [-1.113, -0.741, -0.817, 0.197, 0.209, 0.203, 0.864, 0.45, 0.221]
    assert l["lin_b"] == [0.859, 0.092, 0.875, -0.01, -0.452, -0.096]


def test_the_rest_of_the_reference_benchmark_classes(described):
    """Every system class of nonlinear_benchmark_systems.h / linear_benchmark_systems.h is recognised by RTTI, with the parameters its
    setters were given (private members where the class has no getter) and the collocation scheme of its grid."""
    want = {"rocket": (capi.DYN_FREE_SPACE_ROCKET, capi.DEFECT_CRANK_NICOLSON, None), "mpendulum": (capi.DYN_MASSLESS_PENDULUM, capi.DEFECT_CRANK_NICOLSON, 1.4),
            "toy": (capi.DYN_TOY_EXAMPLE, capi.DEFECT_FORWARD, 0.35), "artstein": (capi.DYN_ARTSTEINS_CIRCLE, capi.DEFECT_BACKWARD, None),
            "cartpole": (capi.DYN_CART_POLE, capi.DEFECT_CRANK_NICOLSON, None), "par2": (capi.DYN_PARALLEL_INTEGRATOR, capi.DEFECT_CRANK_NICOLSON, 0.8)}
    for name, (dyn, defect, p0) in want.items():
        r = described[name]
        assert r["recognised"] == 1 and r["dynamics"] == dyn and r["defect"] == defect and r["grid"] == capi.GRID_FD, name
        if p0 is not None:
            assert r["dyn_params"][0] == p0, name
    assert (described["cartpole"]["nx"], described["par2"]["nu"]) == (4, 2)
    # the massless pendulum carries a TerminalPartialEqualityConstraint on its angle only
    assert (described["mpendulum"]["final_eq"], described["mpendulum"]["final_eq_mask"]) == (1, 1)


def test_plain_and_integral_cost_forms_are_recognised(described):
    """The IPOPT-style configuration: QuadraticFormCost / QuadraticFinalStateCost with lsq_form = false are plain objective edges, the
    integral form one TrapezoidalIntegralCostEdge per interval.  Weights and reference are identified through the edges themselves: the
    reference as the point where the term is exactly zero, plain weights exactly, integrand weights to an ulp (they come divided by dt)."""
    p = described["vdp_plain"]
    assert p["recognised"] == 1 and (p["cost_nonlsq"], p["cost_integral"], p["stage_cost"], p["final_cost"]) == (1, 0, capi.COST_QUADRATIC_LSQ, 1)
    assert p["q_diag"] == [1.0, 0.3] and p["r_diag"] == [0.2] and p["qf_diag"] == [7.0, 7.0 * 0.3] and p["xref"] == [0.2, -0.1]
    m = described["dint_plain"]                          # MinimumTime(lsq_form = false): (N - 1) dt, created twice
    assert m["recognised"] == 1 and (m["cost_nonlsq"], m["stage_cost"], m["grid"]) == (1, capi.COST_MIN_TIME_LSQ, capi.GRID_FD_VARIABLE)
    t = described["vdp_itrap"]
    assert t["recognised"] == 1 and (t["cost_nonlsq"], t["cost_integral"]) == (1, 1) and t["xref"] == [0.2, -0.1]
    assert np.allclose(t["q_diag"], [1.0, 0.3], rtol=4e-16, atol=0) and np.allclose(t["r_diag"], [0.2], rtol=4e-16, atol=0) and t["qf_diag"] == [7.0, 7.0 * 0.3]
    # the same cost on a MultipleShootingGrid (Runge-Kutta 3): one MultipleShootingEdgeSingleControl -- a MIXED edge -- per interval; dynamics, integrator
    # and stage cost come from the edge's private members, the integrand's weights (exactly: no dt in between) from the stage cost's own integrand
    ms = described["vdp_msint"]
    assert ms["recognised"] == 1 and (ms["grid"], ms["defect"], ms["cost_nonlsq"], ms["cost_integral"], ms["final_cost"]) == (capi.GRID_MS, capi.DEFECT_RK4_SHOOTING, 1, 1, 1)
    assert ms["q_diag"] == [1.0, 0.3] and ms["r_diag"] == [0.2] and ms["qf_diag"] == [7.0, 7.0 * 0.3] and ms["xref"] == [0.2, -0.1] and ms["shooting_integrator"] == 3
    # MinTimeQuadratic in integral form on the FiniteDifferencesVariableGrid: plain dt terms (twice) + one integral edge per interval -- with
    # only_last_n = 8 on the intervals k >= N - 8 only (hybrid_cost.h:209, 224-237)
    mi = described["dint_mtq_itrap"]
    assert mi["recognised"] == 1 and (mi["grid"], mi["stage_cost"], mi["cost_nonlsq"], mi["cost_integral"], mi["quad_first_interval"]) == (capi.GRID_FD_VARIABLE, capi.COST_MIN_TIME_QUADRATIC_LSQ, 1, 1, 0)
    ml = described["dint_mtq8_ileft"]
    assert ml["recognised"] == 1 and (ml["stage_cost"], ml["cost_integral"], ml["quad_first_interval"], ml["N"]) == (capi.COST_MIN_TIME_QUADRATIC_LSQ, 2, 22, 30)
    assert np.allclose(mi["q_diag"], [1.0, 0.5], rtol=4e-16, atol=0) and np.allclose(ml["r_diag"], [0.1], rtol=4e-16, atol=0)


def test_what_the_device_cannot_describe_is_refused_with_a_reason(described):
    # (unicycle_fullq -- non-diagonal Q, R, Qf -- is described since round 3: its weights are identified as upper Cholesky factors and the
    #  describe mode, which runs without a device, stops at the user dynamics class that is fingerprinted ON the device)
    f = described["unicycle_fullq"]
    assert f["recognised"] == 0 and "system dynamics class" in f["reason"]
    u = described["unicycle_uref"]
    assert u["recognised"] == 0 and "least-squares term" in u["reason"]


@pytest.mark.parametrize("seed", range(16))
def test_arbitrary_weights_references_and_parameters_are_identified(seed):
    """DROPIN_FUZZ=<seed>: the Van-der-Pol scenarios with weights over three decades, a goal and a damping coefficient drawn at random -- arbitrary
    doubles instead of the scenarios' round numbers (which hid two weaknesses of the plain-term identification until a time-varying reference
    met them).  Least-squares form, plain terms, integral cost edges, the shooting grid's mixed edges: the reference and the damping come out
    bit for bit, plain weights exactly, least-squares weights as the device will take their root, integrand weights to an ulp (they arrive
    multiplied by dt).  (2 400 further cases run clean: DESIGN.md 3.6.)"""
    p = subprocess.run([DEMO, "describe", "vdp", "vdp_plain", "vdp_itrap", "vdp_msint"], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, DROPIN_FUZZ=str(seed)))
    assert p.returncode == 0, p.stderr
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 8
    for drawn, r in zip(lines[0::2], lines[1::2]):
        xf, damping, q, rr, qf = drawn["drawn"][0:2], drawn["drawn"][2], drawn["drawn"][3:5], drawn["drawn"][5:6], drawn["drawn"][6:8]
        assert r["recognised"] == 1, r
        assert r["xref"] == xf and r["dyn_params"][0] == damping, (r["scenario"], seed)
        if r["cost_nonlsq"] == 0:
            assert _sqrt_equal(r["q_diag"], q) and _sqrt_equal(r["r_diag"], rr) and _sqrt_equal(r["qf_diag"], qf), (r["scenario"], seed)
        else:
            rtol = 4e-16 if (r["cost_integral"] and r["grid"] == capi.GRID_FD) else 0.0   # (mixed edges: probed through the stage cost itself, no dt in between)
            assert np.allclose(r["q_diag"], q, rtol=rtol, atol=0) and np.allclose(r["r_diag"], rr, rtol=rtol, atol=0), (r["scenario"], seed)
            assert r["qf_diag"] == qf, (r["scenario"], seed)
