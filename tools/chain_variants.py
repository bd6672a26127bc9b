"""Big-block family: the chain formulations side by side (diagnostics).  python tools/chain_variants.py [batch] [N]
For every variant (corbo_hip_set_option chain_variant): one 10-iteration solve of the seeded quadrotor batch -- deviation of the final iterate / chi2 from
the twisted chain (variant 2), LM counters, time per solve and per factor launch group."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from control_box_rst_amd import problems
from control_box_rst_amd.solver import BatchedLevenbergMarquardt

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
variants = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [2, 6, 4, 3, 5]
d = problems.quad_desc(N=N)
x0, xf = problems.quad_instances(B)
ref = None
for v in variants:
    s = BatchedLevenbergMarquardt(d, B)
    s.setPenaltyWeights(*problems.QUAD_WEIGHTS)
    s.set_option("chain_variant", v)
    s.set_instance_data(s.init_trajectory(x0, xf), xref=xf)
    s.solve(new_run=True); s.synchronize()
    x, chi2, status = s.get_solution()
    st = s.get_stats()
    R = 5
    t0 = time.perf_counter()
    for _ in range(R):
        s.restore_instance_data(); s.solve(new_run=True)
    s.synchronize()
    ms = (time.perf_counter() - t0) / R * 1e3
    s.restore_instance_data()
    fms = s.time_factor(repeat=5)
    if ref is None: ref = (x, chi2)
    dx = np.abs(x - ref[0]).max(); dc = (np.abs(chi2 - ref[1]) / np.maximum(1e-300, np.abs(ref[1]))).max()
    print(f"variant {v}: B={B} N={N} solve {ms:.3f} ms, factor group {fms:.3f} ms, stats { {k: st[k] for k in st if k in ('passes','accepted_steps','rejected_steps','factorizations','lm_iterations')} }, "
          f"max|dx| vs first {dx:.3e}, max rel dchi2 {dc:.3e}, chi2 sum {chi2.sum():.10g}, nan {int(np.isnan(x).sum())}", flush=True)
    del s
