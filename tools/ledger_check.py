"""Device deviation from the reference fixture, per fixture of the tolerance ledger (development aid; the asserts are tests/test_gpu_parity.py).
    python tools/ledger_check.py        # on the GPU box: fixture, max |x - ref| / chi2 rel over the fixture's iterates, ledger tolerance, reference spread"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import desc_for, load_golden, LEDGER   # noqa: E402
from control_box_rst_amd.solver import BatchedLevenbergMarquardt   # noqa: E402

out = {}
names = list(LEDGER["fixtures"]) + [n for n in sys.argv[1:]]
for name in names:
    g = load_golden(name); d = desc_for(g)
    dx = dc = 0.0
    per = []
    for a in g["after_iter"]:
        s = BatchedLevenbergMarquardt(d, 1); s.setIterations(a["k"]); s.setPenaltyWeights(*g["weights"])
        s.set_instance_data(s.init_trajectory(g["x0"], g["xf"]), xref=np.array(g["xf"])[None, :])
        for i in range(g["solves"]): s.solve(new_run=(i == 0))
        x, chi2, _ = s.get_solution()
        ex = float(np.abs(x[0] - np.array(a["vertex"])[: s.dims.nv]).max()); ec = float(abs(chi2[0] - a["chi2"]) / max(1.0, abs(a["chi2"])))
        per.append((a["k"], ex, ec)); dx = max(dx, ex); dc = max(dc, ec)
    e = LEDGER["fixtures"].get(name, {})
    out[name] = {"device_dx": dx, "device_dchi2": dc, "per_iter": per}
    print("%-24s device dx %.2e dchi2 %.2e | ledger x_tol %.0e chi2 %.0e | ref spread x %.2e chi2 %.2e | %s" % (
        name, dx, dc, e.get("x_tol", LEDGER["default_x_tol"]), e.get("chi2_rtol", LEDGER["default_chi2_rtol"]), e.get("ref_spread_x", 0), e.get("ref_spread_chi2", 0),
        " ".join("k%d:%.1e" % (k, x) for k, x, _ in per)))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "ledger_check.json"), "w"), indent=1)
