"""Launch only the edge/Jacobian sweep kernel (and the factor kernel) a few times on the headline batch -- the command that
rocprofv3 --pmc / --kernel-trace wraps to attribute counters to one kernel.   python tools/profile_sweep.py [batch] [repeat]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from control_box_rst_amd import problems  # noqa: E402
from control_box_rst_amd.solver import BatchedLevenbergMarquardt  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
R = int(sys.argv[2]) if len(sys.argv) > 2 else 10
d = problems.unicycle_desc()
x0, xf = problems.unicycle_instances(B)
s = BatchedLevenbergMarquardt(d, B)
s.setPenaltyWeights(*problems.UNICYCLE_WEIGHTS)
s.set_instance_data(s.init_trajectory(x0, xf), xref=xf)
ms = s.time_sweep(with_jacobian=True, repeat=R)
b_sweep = 8 * (s.dims.nv + 2 * s.dims.n + s.dims.m + s.dims.nnz)
print(f"sweep(values+J): {ms * 1e3:.2f} us/launch, algorithmic {B * b_sweep / 1e6:.2f} MB -> {B * b_sweep / ms / 1e6:.1f} GB/s")
mf = s.time_factor(repeat=R)
print(f"factor: {mf * 1e3:.2f} us/launch")
