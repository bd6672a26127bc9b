"""diagnostics: where the host time of a batch-1 solve goes (cfg 1 / cfg 2).  python tools/latency_probe.py [cfg]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from control_box_rst_amd.solver import BatchedLevenbergMarquardt
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
w = bench.workload(cfg, 1)
s = BatchedLevenbergMarquardt(w["desc"], 1)
s.setIterations(10); s.setPenaltyWeights(*w["weights"])
X0 = s.init_trajectory(w["x0"], w["xf"])
s.set_instance_data(X0, xref=w["xf"])
def t(f, n=300):
    for _ in range(20): f()
    s.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    s.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
print("restore_instance_data           us", round(t(s.restore_instance_data), 1))
print("solve (new_run)                 us", round(t(lambda: s.solve(True)), 1), " device ms", s.get_stats()["solve_ms"], "passes", s.get_stats()["factorizations"])
s.set_result_sink(True)
print("solve + sink                    us", round(t(lambda: s.solve(True)), 1))
print("fetch_solution (sink)           us", round(t(s.fetch_solution), 1))
s.set_result_sink(False)
print("get_solution                    us", round(t(s.get_solution), 1))
print("set_instance_data               us", round(t(lambda: s.set_instance_data(X0, xref=w["xf"])), 1))
def full():
    s.set_instance_data(X0, xref=w["xf"]); s.solve(True); s.get_solution()
print("set + solve + get (drop-in)     us", round(t(full), 1))
s.setIterations(0)
print("solve with 0 iterations         us", round(t(lambda: s.solve(True)), 1))
