"""Hessian values kernel: the work splits side by side (diagnostics).  python tools/hessian_split_ab.py [batch] -- run under rocprofv3 --kernel-trace --stats
for the kernel's own duration; prints the wall time per call and the deviation of split 1 / 2 from split 0."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from control_box_rst_amd import problems
from control_box_rst_amd.solver import BatchedLevenbergMarquardt
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
splits = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2]
d = problems.unicycle_desc()
x0, xf = problems.unicycle_instances(B)
s = BatchedLevenbergMarquardt(d, B)
rng = np.random.default_rng(0)
X = s.init_trajectory(x0, xf) + 0.02 * rng.normal(size=(B, s.dims.nv)); X[:, : d.nx] = x0
s.set_instance_data(X, xref=xf)
me = rng.uniform(0.2, 1.0, (B, s.dims.eq))
ref = None
for sp in splits:
    s.set_option("hess_split", sp)
    v = s.eval_hessians(True, 1.0, me, None)
    t0 = time.perf_counter()
    for _ in range(10): s.eval_hessians_views(True, 1.0, None, None, device=True)
    ms = (time.perf_counter() - t0) / 10 * 1e3
    if ref is None: ref = v
    dev = max(np.abs(a - b).max() for a, b in zip(v, ref) if a.size)
    print(f"split {sp}: batch {B}: {ms:.3f} ms per call (wall), max |dev| vs first {dev:.3e}")
