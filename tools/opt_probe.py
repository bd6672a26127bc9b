"""diagnostics: headline solve time under handle options.  python tools/opt_probe.py name=v[,name=v] ...   (each argument = one configuration)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from control_box_rst_amd.solver import BatchedLevenbergMarquardt
batch = int(os.environ.get("BATCH", "1024"))
w = bench.workload(3, batch)
s = BatchedLevenbergMarquardt(w["desc"], batch)
s.setIterations(10)
s.setPenaltyWeights(*w["weights"])
X0 = s.init_trajectory(w["x0"], w["xf"])
s.set_instance_data(X0, xref=w["xf"])
ref = None
for cfg in sys.argv[1:] or [""]:
    for kv in [c for c in cfg.split(",") if c]:
        k, v = kv.split("=")
        s.set_option(k, int(v))
    ts = []
    for rep in range(14):
        s.restore_instance_data()
        s.solve(new_run=True)
        ts.append(s.get_stats()["solve_ms"])
    X, chi2, st = s.get_solution()
    if ref is None:
        ref = X.copy()
        if os.environ.get("SAVE_X"): np.save(os.environ["SAVE_X"], X)
        if os.environ.get("CMP_X") and os.path.exists(os.environ["CMP_X"]):
            R = np.load(os.environ["CMP_X"]); print("   vs", os.environ["CMP_X"], ": identical", bool(np.array_equal(X, R)), "max |dx|", float(np.abs(X - R).max()), "stats", {k: s.get_stats()[k] for k in ("accepted_steps", "rejected_steps", "factorizations")})
    print(f"{cfg:30s} solve_ms min {min(ts[2:]):.4f} median {sorted(ts[2:])[len(ts[2:]) // 2]:.4f}  chi2_sum {chi2.sum():.6f}  identical_to_first {bool(np.array_equal(X, ref))}")
