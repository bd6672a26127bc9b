"""diagnostics: headline solve time against the start stagger of the workgroups that share a CU.  python tools/stagger_probe.py s0 s1 ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from control_box_rst_amd.solver import BatchedLevenbergMarquardt
batch = 1024
w = bench.workload(3, batch)
s = BatchedLevenbergMarquardt(w["desc"], batch)
s.setIterations(10)
s.setPenaltyWeights(*w["weights"])
X0 = s.init_trajectory(w["x0"], w["xf"])
s.set_instance_data(X0, xref=w["xf"])
for stg in [int(a) for a in sys.argv[1:]] or [0]:
    s.set_option("stagger", stg)
    ts = []
    for rep in range(12):
        s.restore_instance_data()
        s.solve(new_run=True)
        ts.append(s.get_stats()["solve_ms"])
    print("stagger", stg, "solve_ms min/median", min(ts[2:]), sorted(ts[2:])[len(ts[2:]) // 2])
