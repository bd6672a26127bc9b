"""Time the host-buffer boundary of one solve: set_instance_data (H2D), solve, get_solution (D2H), per call (diagnostics)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from control_box_rst_amd import problems  # noqa: E402
from control_box_rst_amd.solver import BatchedLevenbergMarquardt  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
desc = problems.unicycle_desc()
x0, xf = problems.unicycle_instances(B)
s = BatchedLevenbergMarquardt(desc, B, device=0)
s.setIterations(10)
s.setPenaltyWeights(*problems.UNICYCLE_WEIGHTS)
X0 = s.init_trajectory(x0, xf)
for rep in range(6):
    t0 = time.perf_counter(); s.set_instance_data(X0, xref=xf)
    t1 = time.perf_counter(); s.solve(new_run=True)
    t2 = time.perf_counter(); X, chi2, st = s.get_solution()
    t3 = time.perf_counter()
    print(f"rep {rep}: set_instance_data {1e3*(t1-t0):7.3f} ms | solve {1e3*(t2-t1):7.3f} ms (device {s.get_stats()['solve_ms']:.3f}) | get_solution {1e3*(t3-t2):7.3f} ms")
