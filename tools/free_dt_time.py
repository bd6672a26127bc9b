"""Free dt around the 12-state quadrotor (time-optimal, MultipleShootingVariableGrid): the stage / partitioned-chain route against the band route
(corbo_hip_create_routed, CORBO_HIP_ROUTE_FREE_DT_BAND), same solves (diagnostics).  python tools/free_dt_time.py [N]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from control_box_rst_amd import problems
from control_box_rst_amd.solver import BatchedLevenbergMarquardt
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for B in (1, 64, 256, 512):
    for band in (False, True):
        if band and B > 256: continue
        d = problems.quad_desc(N=N, time_optimal=True)
        x0 = np.zeros((B, 12)); xf = np.zeros((B, 12)); xf[:, 0] = 2.0; xf[:, 1] = 1.0
        xf[:, 2] = np.linspace(-0.2, 0.4, B)
        s = BatchedLevenbergMarquardt(d, B, route=1 if band else 0); s.setPenaltyWeights(100.0, 100.0, 100.0)   # (1 = CORBO_HIP_ROUTE_FREE_DT_BAND)
        s.set_instance_data(s.init_trajectory(x0, xf), xref=xf); s.solve(new_run=True); s.synchronize()
        x, chi2, status = s.get_solution()
        t0 = time.perf_counter()
        for _ in range(3): s.restore_instance_data(); s.solve(new_run=True)
        s.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
        st = s.get_stats()
        print(f"N={N} batch {B:4d} {'band ' if band else 'chain'}: {ms:8.2f} ms per 10-iteration solve, chi2 sum {chi2.sum():.10g}, passes {st['passes']}, factorizations {st['factorizations']}", flush=True)
        del s
