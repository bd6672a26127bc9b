"""Drop-in test (-m gpu): the genuine reference's StructuredOptimalControlProblem, driven with the HIP solver injected
through corbo::NlpSolverInterface (control_box_rst_amd/adapter + libcorbo_hip.so), against the same OCP solved by the
reference's own LevenbergMarquardtSparse.  The binary oracle/_ref/dropin_demo is built in the build container by
`make -C oracle ref` (it needs the reference's headers); it is skipped when it has not travelled."""
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
    p = subprocess.run([DEMO], capture_output=True, text=True, timeout=300)
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 5, (p.stdout, p.stderr)   # cfg 3, cfg 2, reduced cfg 5, cfg 3 structure + TerminalBall, mismatching model
    refused = lines.pop()
    # the adapter compares the graph's own residual with the device's on every new structure: a device model with another
    # state weight than the OCP's QuadraticFormCost is refused (SolverStatus::Error), never silently solved
    assert refused["scenario"] == "unicycle_mismatch" and refused["ok_hip"] == 0, refused
    assert "does not describe this hypergraph" in p.stdout + p.stderr
    for r in lines:
        assert r["ok_reference"] == 1 and r["ok_hip"] == 1, r
        # cfg 3: 10 LM iterations; cfg 2: 5 x 10 iterations with warm start -- same tolerance as the golden parity tests;
        # cfg 5 family (quadrotor, multiple shooting, N=30): flat directions, chi2 carries the comparison
        assert r["max_abs_diff"] <= (3e-4 if r["scenario"] == "quad" else 5e-6), r
        assert abs(r["chi2_hip"] - r["chi2_reference"]) <= 2e-6 * max(1.0, abs(r["chi2_reference"])), r
    assert p.returncode == 0
