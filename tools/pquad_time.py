"""Six-state planar quadrotor (user model of the big-block family): partitioned chain vs the first formulation (fixed dt), chain route vs band route (free dt);
ms per 10-iteration solve (diagnostics).  python tools/pquad_time.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from control_box_rst_amd import problems
from control_box_rst_amd.solver import BatchedLevenbergMarquardt
def run(d, B, label, variant=0, band=False, w=(10.0, 10.0, 10.0)):
    s = BatchedLevenbergMarquardt(d, B, route=1 if band else 0)   # (1 = CORBO_HIP_ROUTE_FREE_DT_BAND)
    s.setPenaltyWeights(*w)
    if variant: s.set_option("chain_variant", variant)
    x0 = np.zeros((B, 6)); xf = np.zeros((B, 6)); xf[:, 0] = 2.0; xf[:, 1] = np.linspace(0.8, 1.2, B)
    s.set_instance_data(s.init_trajectory(x0, xf), xref=xf); s.solve(new_run=True); s.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): s.restore_instance_data(); s.solve(new_run=True)
    s.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    st = s.get_stats(); chi2 = s.get_solution()[1]
    print(f"{label}: batch {B}: {ms:.2f} ms per solve, chi2 sum {chi2.sum():.10g}, factorizations {st['factorizations']}", flush=True)
for B in (1, 256):
    for N in (50, 100):
        d = problems.planar_quadrotor_desc(N=N)
        run(d, B, f"pquad fixed dt N={N} partitioned chain")
        run(d, B, f"pquad fixed dt N={N} first formulation", variant=1)
        d = problems.planar_quadrotor_desc(N=N, time_optimal=True)
        run(d, B, f"pquad free dt N={N} chain", w=(100.0,)*3)
        run(d, B, f"pquad free dt N={N} band ", band=True, w=(100.0,)*3)
