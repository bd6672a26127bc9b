"""cfg 5 (quadrotor, multiple shooting + RK4, N = 200, batch 512): a few solves -- the command rocprofv3 --kernel-trace --stats wraps.
    python tools/profile_cfg5.py [batch] [solves] [chain_variant] [nospec]      (nospec: without the reject-streak speculation -- every launch has exactly `batch` workgroups / instance rows)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from control_box_rst_amd import problems  # noqa: E402
from control_box_rst_amd.solver import BatchedLevenbergMarquardt  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
d = problems.quad_desc()
x0, xf = problems.quad_instances(B)
s = BatchedLevenbergMarquardt(d, B)
s.setPenaltyWeights(*problems.QUAD_WEIGHTS)
if len(sys.argv) > 3:
    s.set_option("chain_variant", int(sys.argv[3]))
if len(sys.argv) > 4 and sys.argv[4] == "nospec":
    s.set_option("reject_speculation", 0)
s.set_instance_data(s.init_trajectory(x0, xf), xref=xf)
s.restore_instance_data(); s.solve(new_run=True); s.synchronize()
t0 = time.perf_counter()
for _ in range(R):
    s.restore_instance_data()
    s.solve(new_run=True)
s.synchronize()
dt = (time.perf_counter() - t0) / R
st = s.get_stats()
print(f"cfg5 batch={B} N={d.N}: {dt * 1e3:.2f} ms/solve, {B * 10 / dt / 1e3:.1f} k SQP-iterations/s, passes={st['passes']}")
