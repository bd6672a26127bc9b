"""Headline workload (1024 unicycle OCPs, N = 100) with the control-deviation term (rate limit on the controls) / an integral-form inequality: the
block-tridiagonal route (run to completion) and the band route (corbo_hip_create_routed, CORBO_HIP_ROUTE_XE_BAND) per solve next to the
plain problem's (diagnostics).  python tools/xe_time.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from control_box_rst_amd import capi, problems
from control_box_rst_amd.solver import BatchedLevenbergMarquardt, get_structure, get_dims
import scipy.sparse as sp
for B in (1024, 64, 1):
    for kind in ("plain", "rate", "rate+ballint"):
        w = bench.workload(3, B)
        d = w["desc"]
        if kind != "plain":
            d.ctrl_dev = capi.CTRL_DEV_RATE
            d.ctrl_dev_params[0] = 1.0; d.ctrl_dev_params[1] = 1.0
        if kind == "rate+ballint":
            d.constraint_integration = 1
            d.stage_ineq, d.stage_ineq_integral = capi.INEQ_BALL, 1
            for i, v in enumerate((1.0, 0.5, 0.2, 0.3)): d.ineq_params[i] = v
        rows, cols = get_structure(d); dims = get_dims(d)
        J = sp.coo_matrix((np.ones(len(rows)), (rows, cols)), shape=(dims.m, dims.n)).tocsr(); H = (J.T @ J).tocoo()
        for route in ((0,) if kind == "plain" else (0, capi.ROUTE_XE_BAND)):
            s = BatchedLevenbergMarquardt(d, B, route=route); s.setIterations(10); s.setPenaltyWeights(*w["weights"])
            s.set_instance_data(s.init_trajectory(w["x0"], w["xf"]), xref=w["xf"]); s.solve(new_run=True); s.synchronize()
            t0 = time.perf_counter()
            for _ in range(3): s.restore_instance_data(); s.solve(new_run=True)
            s.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
            st = s.get_stats(); chi2 = s.get_solution()[1]
            tag = "" if kind == "plain" else (" band route " if route else " block route")
            print(f"batch {B:5d} {kind:13s}{tag}: {ms:8.2f} ms per solve, passes {st['passes']}, factorizations {st['factorizations']}, accepted {st['accepted_steps']}, chi2 sum {chi2.sum():.10g}, n {dims.n}, half-bandwidth {int(np.abs(H.row - H.col).max())}", flush=True)
            del s
