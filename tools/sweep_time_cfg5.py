"""Residual sweep of cfg 5 (quadrotor, N = 200) on its own: ms per launch and the shader-clock phases of instance 0 (diagnostics).
    python tools/sweep_time_cfg5.py [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench   # noqa: E402
from control_box_rst_amd.solver import BatchedLevenbergMarquardt   # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
w = bench.workload(5, batch)
s = BatchedLevenbergMarquardt(w["desc"], batch)
s.setIterations(10)
s.setPenaltyWeights(*w["weights"])
s.set_instance_data(s.init_trajectory(w["x0"], w["xf"]), xref=w["xf"])
s.set_option("sweep_timeline", 1)
for wj in (False, True):
    print("with_jacobian", wj, "ms per launch %.4f" % s.time_sweep(with_jacobian=wj, repeat=20), flush=True)
