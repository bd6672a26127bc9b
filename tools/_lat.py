import time, ctypes as C, numpy as np, sys, os
if os.environ.get("WITH_TORCH"):
    import torch; torch.cuda.set_device(0); torch.cuda.synchronize()
sys.path.insert(0,'.')
import bench
from control_box_rst_amd.solver import BatchedLevenbergMarquardt
w = bench.workload(1, 1)
s = BatchedLevenbergMarquardt(w["desc"], 1); s.setIterations(10); s.setPenaltyWeights(*w["weights"])
s.set_instance_data(s.init_trajectory(w["x0"], w["xf"]), xref=w["xf"]); s.set_result_sink(True)
s.solve(True)
def tm(f,n=500):
    t0=time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter()-t0)/n*1e6
print("wrapper fetch us", tm(s.fetch_solution))
print("restore us", tm(s.restore_instance_data))
def step():
    s.restore_instance_data(); s.solve(True); return s.fetch_solution()
print("step us", tm(step,300))
print("solve only us", tm(lambda: s.solve(True),300))
