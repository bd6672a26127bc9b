"""Drop-in test (-m gpu): the genuine reference's StructuredOptimalControlProblem, driven with the HIP solver injected
through corbo::NlpSolverInterface (control_box_rst_amd/adapter + libcorbo_hip.so), against the same OCP solved by the
reference's own LevenbergMarquardtSparse.  The HIP solver is configured with the reference solver's own setters ONLY
(setIterations, setPenaltyWeights): the device model is derived from the hypergraph (adapter/graph_recogniser.cpp).
The binary oracle/_ref/dropin_demo is built in the build container by `make -C oracle ref` (it needs the reference's
headers); it is skipped when it has not travelled."""
import json
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

DEMO = os.path.join(ROOT, "oracle", "_ref", "dropin_demo")


@pytest.mark.skipif(not os.path.exists(DEMO), reason="oracle/_ref/dropin_demo not built (needs /root/reference at build time)")
def test_reference_ocp_with_hip_solver_matches_reference_solver():
    import __graft_entry__ as g
    g.build()
    p = subprocess.run([DEMO], capture_output=True, text=True, timeout=600)
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    solved = [r for r in lines if "max_abs_diff" in r]
    hessian = [r for r in lines if r.get("mode") == "hessian"]
    refused = {r["scenario"]: r for r in lines if "max_abs_diff" not in r and r.get("mode") != "hessian"}
    # cfg 3, cfg 2, reduced cfg 5, TerminalBall, cfg 1 (a = 1.3), Duffing (midpoint, private parameters), pendulum (+ terminal equality),
    # linear state-space model on the shooting grid -- all recognised from the graph; then the stated-model override
    # ... and a time-varying state reference (DiscreteTimeReferenceTrajectory): one reference per cost edge, corbo_hip_set_references
    assert [r["scenario"] for r in solved] == ["unicycle", "dint", "quad", "unicycle_tball", "vdp", "duffing", "pendulum", "lin32", "unicycle_tvref", "dint_ms", "dint_mtq", "unicycle_tballc", "dint_mtq8", "rocket", "mpendulum",
                                              "toy", "artstein", "cartpole", "par2", "unicycle_fullq", "lin32_rk3", "kcar", "pquad", "lin32_rk7", "pquad_fd", "unicycle_moved",
                                              # integral-form constraints / control-deviation term: user stage functions recognised through the edges' own evaluation
                                              "unicycle_xe_ball", "unicycle_xe_eq", "unicycle_xe_rate", "unicycle_xe_all",
                                              # time-optimal transfer of the six-state user model on the MultipleShootingVariableGrid: a free dt around a big-block model
                                              "pquad_topt", "pquad_pteq", "pquad_fd_xe_ball", "pquad_xe_rate",
                                              # ... and of the 12-state quadrotor: the dt column rides through the stage / partitioned-chain kernels (round 5)
                                              "quad_topt", "quad_rk5",
                                              # round 6: USER stage functions of csrc/stage_functions/ matched against the graph's inequality edges by evaluation -- an input-magnitude
                                              # bound on every u_k (control term), a tilt cone on the quadrotor's roll / pitch (state term)
                                              "unicycle_sf_unorm", "quad_sf_tilt",
                                              # ... non-diagonal weights TOGETHER with a rate limit on the controls (band route: the DENSE x XE sweep instantiation)
                                              "unicycle_fullq_xe_rate", "unicycle"], (p.stdout, p.stderr)   # (quad_rk5: Runge-Kutta 5 around the 12-state model)
    assert [r["mode"] for r in solved] == ["recognised"] * 39 + ["stated"]   # (unicycle_moved: the setpoint moves between two runs without a structure change -- model tracking)   # (kcar, pquad: user dynamics classes matched against csrc/models/kinematic_car.hpp / planar_quadrotor.hpp -- the latter one of the big-block family)   # (dint_ms: cfg 2 on the MultipleShootingVariableGrid; dint_mtq: MinTimeQuadratic;
    # unicycle_tballc: TerminalBallInheritFromCost; dint_mtq8: MinTimeQuadratic with only_last_n)
    for r in solved:
        assert r["ok_reference"] == 1 and r["ok_hip"] == 1, (r, p.stderr[-2000:])
        # cfg 3: 10 LM iterations; cfg 2: 5 x 10 iterations with warm start -- same tolerance as the golden parity tests;
        # cfg 5 family (quadrotor, multiple shooting, N=30): soft directions, chi2 carries the comparison (tests/test_oracle_fullsize.py)
        assert r["max_abs_diff"] <= (5e-4 if r["scenario"] in ("pquad_pteq", "pquad_fd_xe_ball", "pquad_xe_rate") else 3e-4 if r["scenario"] in ("quad", "pquad", "pquad_fd", "pquad_topt", "quad_topt", "quad_rk5", "quad_sf_tilt") else 3e-5 if r["scenario"] == "unicycle_tvref" else 5e-6), r   # tvref: tests/test_references.py
        assert abs(r["chi2_hip"] - r["chi2_reference"]) <= 2e-6 * max(1.0, abs(r["chi2_reference"])), r
    # the operators of the exact-Hessian path through the adapter (LevenbergMarquardtSparseHip::computeSparseHessians*), at a generic point
    # of the same graphs, against the graph's own computeSparseHessians{NNZ,Structure,Values}: identical lists, values within the
    # reference's own consecutive-call spread (tests/test_gpu_hessian.py)
    assert [r["scenario"] for r in hessian] == ["unicycle", "dint", "unicycle_tball", "vdp", "duffing", "pendulum", "lin32", "unicycle_tvref", "dint_ms", "dint_mtq", "dint_mtq8", "rocket", "toy",
                                               "cartpole", "par2", "unicycle_fullq", "unicycle_plain", "unicycle_itrap", "unicycle_ileft",
                                               "unicycle_plain_stated", "vdp_plain", "vdp_itrap", "dint_plain", "unicycle_msint", "vdp_msint", "dint_mtq_itrap", "dint_mtq8_ileft", "unicycle_plain_tvref", "unicycle_msint_tvref"], p.stdout   # (the last nine: graphs with plain /
    # integral objective edges, the IPOPT-style cost forms -- recognised, one of them with a stated model; ..._msint: the shooting grid's MIXED edges,
    # MultipleShootingEdgeSingleControl)
    for r in hessian:
        assert r["ok_hip"] == 1 and r["structure_equal"] == 1 and r["nnz"][1] > 0 and r["max_rel_diff"] <= 2e-4, r
    # a stated model with a wrong CONTROL weight -- invisible in the residual at the reference's initial guess u = 0 -- is refused by the
    # perturbed probe (residual + Jacobian); a graph with a non-zero control reference is refused by the recogniser.  Never silently solved.
    # (unicycle_fullq -- non-diagonal Q, R, Qf -- was refused until round 3: it is solved above now)
    assert refused["unicycle_mismatch"]["ok_hip"] == 0 and refused["unicycle_uref"]["ok_hip"] == 0, refused
    assert "does not describe this hypergraph" in p.stdout + p.stderr
    assert "has no device description" in p.stdout + p.stderr
    assert p.returncode == 0
