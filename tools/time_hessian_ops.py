"""Wall time of the exact-Hessian-path operators over the headline batch (cfg 3 structure, 1024 instances): one call each of
corbo_hip_eval_hessians (lower part), corbo_hip_eval_objective_gradient, corbo_hip_eval_linear_form, results copied to host arrays.
    python tools/time_hessian_ops.py [batch] [repeat]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from control_box_rst_amd import problems  # noqa: E402
from control_box_rst_amd.solver import BatchedLevenbergMarquardt  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
R = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for label, nonlsq, integral in (("least-squares cost", 0, 0), ("plain quadratic cost (lsq_form = false)", 1, 0), ("integral cost, trapezoidal rule", 1, 1)):
    d = problems.unicycle_desc()
    if nonlsq:
        problems.hessian_path_cost_form(d, "trapezoidal" if integral else "")
    x0, xf = problems.unicycle_instances(B)
    s = BatchedLevenbergMarquardt(d, B)
    rng = np.random.default_rng(0)
    X = s.init_trajectory(x0, xf) + 0.02 * rng.normal(size=(B, s.dims.nv))
    X[:, : d.nx] = x0
    s.set_instance_data(X, xref=xf)
    me = rng.uniform(0.2, 1.0, (B, s.dims.eq))
    st = s.hessian_structure(True)
    out = {}
    for name, fn in (("eval_hessians", lambda: s.eval_hessians(True, 1.0, me, None)), ("eval_hessians_views(pinned)", lambda: s.eval_hessians_views(True, 1.0, me, None)),
                     ("eval_hessians_views(device, resident multipliers)", lambda: s.eval_hessians_views(True, 1.0, None, None, device=True)),
                     ("objective_gradient", s.objective_gradient), ("linear_form", s.linear_form)):
        fn()
        t0 = time.perf_counter()
        for _ in range(R):
            fn()
        out[name] = (time.perf_counter() - t0) / R * 1e3
    nnz = [len(st[c][0]) for c in range(3)]
    print(f"{label}: batch {B}, N {d.N}, Hessian nnz (obj, eq, ineq) = {nnz}: " + ", ".join(f"{k} {v:.2f} ms" for k, v in out.items()))
