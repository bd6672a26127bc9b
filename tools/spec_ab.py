"""Reject-streak speculation of the big-block family on / off (diagnostics): identical results, fewer passes.  python tools/spec_ab.py [batch] [N]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from control_box_rst_amd import problems
from control_box_rst_amd.solver import BatchedLevenbergMarquardt
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
d = problems.quad_desc(N=N); x0, xf = problems.quad_instances(B)
ref = None
for spec in (0, 1):
    s = BatchedLevenbergMarquardt(d, B); s.setPenaltyWeights(*problems.QUAD_WEIGHTS); s.set_option("reject_speculation", spec)
    s.set_instance_data(s.init_trajectory(x0, xf), xref=xf); s.solve(new_run=True); s.synchronize()
    x, chi2, status = s.get_solution(); st = s.get_stats()
    t0 = time.perf_counter()
    for _ in range(5): s.restore_instance_data(); s.solve(new_run=True)
    s.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    x2, chi22, _ = s.get_solution()
    if ref is None: ref = (x, chi2, st)
    same = np.array_equal(x, ref[0]) and np.array_equal(chi2, ref[1]) and np.array_equal(x2, x)
    keys = ("lm_iterations", "accepted_steps", "rejected_steps", "factorizations", "residual_sweeps", "jacobian_sweeps", "passes")
    print(f"speculation {spec}: B={B} N={N} solve {ms:.3f} ms, identical to off: {same}, stats {[st[k] for k in keys]} (off: {[ref[2][k] for k in keys]}), status {np.unique(status)}", flush=True)
