"""cfg 5: where the chain kernel's cycles go (diagnostics).  python tools/chain_timeline.py [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from control_box_rst_amd import problems
from control_box_rst_amd.solver import BatchedLevenbergMarquardt
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
d = problems.quad_desc()
x0, xf = problems.quad_instances(B)
s = BatchedLevenbergMarquardt(d, B)
s.setPenaltyWeights(*problems.QUAD_WEIGHTS)
s.set_instance_data(s.init_trajectory(x0, xf), xref=xf)
ms, tl = s.time_factor(repeat=5, timeline=True)
names = ["wait+combine", "stacked pass", "lds+mfma+stores", "meeting block", "back-substitution", "epilogue"]
print(f"factor launch group: {ms:.3f} ms;  chain kernel, wave 0 of instance 0 (cycles): " + " | ".join(f"{n} {v}" for n, v in zip(names, tl[:6])), " total", sum(tl[:6]))
