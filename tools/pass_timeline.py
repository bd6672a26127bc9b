"""Shader-clock timeline of one instance of the headline workload inside the run-to-completion kernel (diagnostics):
per pass sweep / factor cycles, and the factor phases of the last pass.   python tools/pass_timeline.py [instance] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from control_box_rst_amd.solver import BatchedLevenbergMarquardt

inst = int(sys.argv[1]) if len(sys.argv) > 1 else 0
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
w = bench.workload(3, batch)
s = BatchedLevenbergMarquardt(w["desc"], batch)
s.setIterations(10)
s.setPenaltyWeights(*w["weights"])
X0 = s.init_trajectory(w["x0"], w["xf"])
for rep in range(3):
    if os.environ.get("RAW"): s.set_option("raw_stamps", 1)
    for kv in [c for c in os.environ.get("OPTS","").split(",") if c]: s.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    s.set_instance_data(X0, xref=w["xf"])
    if rep == 2:
        s.set_option("pass_timeline", inst)
    s.solve(new_run=True)
    s.synchronize()
