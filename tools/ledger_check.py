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
        # the start the parity tests use: xe_* (and `start`) fixtures carry the reference's own initial vertex values and its previous control
        # (tests/test_gpu_extra_edges.py::_solver); the others the straight line init_trajectory builds (tests/test_gpu_parity.py)
        given = name.startswith("xe_") or g.get("start")
        start = np.array(g["vertex_init"])[None, : s.dims.nv] if given else s.init_trajectory(g["x0"], g["xf"])
        s.set_instance_data(start, xref=np.array(g["xf"])[None, :])
        if "u_prev" in g:
            s.set_previous_control(g.get("u_prev"), g.get("u_prev_dt"))
        for i in range(g["solves"]): s.solve(new_run=(i == 0))
        x, chi2, _ = s.get_solution()
        ex = float(np.abs(x[0] - np.array(a["vertex"])[: s.dims.nv]).max()); ec = float(abs(chi2[0] - a["chi2"]) / max(1.0, abs(a["chi2"])))
        per.append((a["k"], ex, ec)); dx = max(dx, ex); dc = max(dc, ec)
    e = LEDGER["fixtures"].get(name, {})
    out[name] = {"device_dx": dx, "device_dchi2": dc, "per_iter": per}
    print("%-24s device dx %.2e dchi2 %.2e | ledger x_tol %.1e chi2 %.1e | ref spread x %.2e chi2 %.2e | %s" % (
        name, dx, dc, e.get("x_tol", LEDGER["default_x_tol"]), e.get("chi2_rtol", LEDGER["default_chi2_rtol"]), e.get("ref_spread_x", 0), e.get("ref_spread_chi2", 0),
        " ".join("k%d:%.1e" % (k, x) for k, x, _ in per)))
bad = [n for n, o in out.items() if o["device_dx"] > LEDGER["fixtures"].get(n, {}).get("x_tol", LEDGER["default_x_tol"])]
print("fixtures whose device deviation exceeds their ledger tolerance:", bad or "none")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "ledger_check.json"), "w"), indent=1)
