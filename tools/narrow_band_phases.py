import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from control_box_rst_amd import capi
from control_box_rst_amd.solver import BatchedLevenbergMarquardt
for B in (1, 1024):
    w = bench.workload(3, B); d = w["desc"]
    d.ctrl_dev = capi.CTRL_DEV_RATE; d.ctrl_dev_params[0] = 1.0; d.ctrl_dev_params[1] = 1.0
    s = BatchedLevenbergMarquardt(d, B); s.setIterations(10); s.setPenaltyWeights(*w["weights"])
    s.set_instance_data(s.init_trajectory(w["x0"], w["xf"]), xref=w["xf"]); s.solve(new_run=True); s.synchronize()
    s.restore_instance_data()
    ms_f, tl = s.time_factor(repeat=3, timeline=True)
    names = ["init", "window", "factorise", "back-substitute", "trial iterate"]
    print(f"batch {B}: assemble+factor {ms_f:.3f} ms per launch; phases (cycles): " + " | ".join(f"{nm} {tl[i + 1] - tl[i]}" for i, nm in enumerate(names)), flush=True)
