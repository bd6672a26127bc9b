"""Block-tridiagonal route (lm_bt_kernel): shader-clock timeline of one instance of the headline batch with a rate limit on the controls -- per pass the
sweep phase's and the factor phase's stamps (factor: f1 operands staged | f2 product lists | f3 defect edges + write phases | f4 cyclic reduction | f5 root |
f6 back-substitution | f7 trial iterate) -- and the mean phase cycles over the batch (diagnostics).   python tools/bt_phases.py [instance] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from control_box_rst_amd import capi
from control_box_rst_amd.solver import BatchedLevenbergMarquardt
inst = int(sys.argv[1]) if len(sys.argv) > 1 else 0
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
w = bench.workload(3, batch); d = w["desc"]
d.ctrl_dev = capi.CTRL_DEV_RATE; d.ctrl_dev_params[0] = 1.0; d.ctrl_dev_params[1] = 1.0
s = BatchedLevenbergMarquardt(d, batch); s.setIterations(10); s.setPenaltyWeights(*w["weights"])
X0 = s.init_trajectory(w["x0"], w["xf"])
for rep in range(3):
    s.set_instance_data(X0, xref=w["xf"])
    if rep == 2: s.set_option("pass_timeline", inst)
    s.solve(new_run=True); s.synchronize()
st = s.get_stats(); print("batch", batch, "instance", inst, "solve_ms", st["solve_ms"], "passes", st["passes"])
s.set_option("pass_timeline", -1); s.set_option("phase_cycles", 1)
s.set_instance_data(X0, xref=w["xf"]); s.solve(new_run=True)
pc = s.get_phase_cycles().astype(float)
print("mean cycles per phase over the batch: sweep with Jacobian %.0f (n %.0f) | residual-only sweep %.0f (n %.0f) | factor %.0f (n %.0f); per instance total: mean %.0f max %.0f" % (
    pc[:, 0].sum() / max(1, pc[:, 3].sum()), pc[:, 3].sum(), pc[:, 1].sum() / max(1, pc[:, 4].sum()), pc[:, 4].sum(), pc[:, 2].sum() / max(1, pc[:, 5].sum()), pc[:, 5].sum(),
    pc[:, :3].sum(axis=1).mean(), pc[:, :3].sum(axis=1).max()))
