"""Diagnostics (the idle tail of the headline launch): how much of it is PACKING?  The launch ends with its slowest CU; an instance's
number of passes (10 .. 23+) is what decides.  Solve the headline batch, read every instance's factorisation count (corbo_hip_get_phase_cycles), upload the SAME
instances again in another order and time the solve:
    as seeded | heaviest first (the first 256 workgroups of the dispatch = the 256 heaviest, one per CU) | lightest first | heavy and light interleaved
Same instances, same arithmetic per instance: the chi2 sum is the same in every order.
    python tools/batch_order_probe.py [batch]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench   # noqa: E402
from control_box_rst_amd.solver import BatchedLevenbergMarquardt   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
w = bench.workload(3, B)


def timed(order):
    s = BatchedLevenbergMarquardt(w["desc"], B)
    s.setIterations(10)
    s.setPenaltyWeights(*w["weights"])
    s.set_option("phase_cycles", 1)
    x0, xf = w["x0"][order], w["xf"][order]
    s.set_instance_data(s.init_trajectory(x0, xf), xref=xf)
    ts = []
    for _ in range(14):
        s.restore_instance_data()
        s.solve(new_run=True)
        ts.append(s.get_stats()["solve_ms"])
    pc = s.get_phase_cycles()
    _, chi2, _ = s.get_solution()
    return sorted(ts[2:])[len(ts[2:]) // 2], min(ts[2:]), pc[:, 5].copy(), float(chi2.sum())


ident = np.arange(B)
med, mn, nfact, c0 = timed(ident)
print(f"as seeded            : median {med:.4f} ms  min {mn:.4f}  factorisations per instance min / mean / max {nfact.min()} / {nfact.mean():.2f} / {nfact.max()}  chi2 sum {c0:.6f}")
desc = np.argsort(-nfact, kind="stable")
inter = np.empty(B, np.int64)
inter[0::2] = desc[: (B + 1) // 2]
inter[1::2] = desc[::-1][: B // 2]
for name, order in (("heaviest first", desc), ("lightest first", desc[::-1].copy()), ("heavy / light interleaved", inter)):
    med, mn, nf, c = timed(order)
    print(f"{name:21s}: median {med:.4f} ms  min {mn:.4f}  chi2 sum {c:.6f}")
