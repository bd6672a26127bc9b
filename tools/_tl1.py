import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from control_box_rst_amd.solver import BatchedLevenbergMarquardt
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
w = bench.workload(cfg, 1)
s = BatchedLevenbergMarquardt(w["desc"], 1)
s.setIterations(10)
s.setPenaltyWeights(*w["weights"])
X0 = s.init_trajectory(w["x0"], w["xf"])
for rep in range(4):
    s.set_instance_data(X0, xref=w["xf"])
    if rep == 3:
        s.set_option("pass_timeline", 0)
    s.solve(new_run=True)
    s.synchronize()
print(s.get_stats())
