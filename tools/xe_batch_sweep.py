"""Block-tridiagonal route: time per solve of the headline structure with a rate limit on the controls over the batch size (three workgroups per CU: 768 resident
instances; diagnostics).   python tools/xe_batch_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from control_box_rst_amd import capi
from control_box_rst_amd.solver import BatchedLevenbergMarquardt
for B in (1, 64, 256, 512, 768, 1024, 1536, 2048, 4096):
    w = bench.workload(3, B); d = w["desc"]
    d.ctrl_dev = capi.CTRL_DEV_RATE; d.ctrl_dev_params[0] = 1.0; d.ctrl_dev_params[1] = 1.0
    s = BatchedLevenbergMarquardt(d, B); s.setIterations(10); s.setPenaltyWeights(*w["weights"])
    s.set_instance_data(s.init_trajectory(w["x0"], w["xf"]), xref=w["xf"]); s.solve(new_run=True); s.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): s.restore_instance_data(); s.solve(new_run=True)
    s.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
    st = s.get_stats()
    print(f"batch {B:5d}: {ms:7.3f} ms per solve, {ms / B * 1e3:.3f} us per instance, factorizations {st['factorizations']}", flush=True)
    del s
