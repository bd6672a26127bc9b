timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_adapt.py -m gpu -q -x -k "horizon_lengths or cross256" 2>&1 | tail -12
python tools/opt_probe.py lag_priority=1 2>&1 | grep -v amdgpu.ids
python - <<'PY'
import time, numpy as np
from control_box_rst_amd import problems
from control_box_rst_amd.solver import BatchedLevenbergMarquardt
for N,B in ((256,256),(300,256),(1000,256)):
    d=problems.unicycle_desc(N=N); x0,xf=problems.unicycle_instances(B)
    s=BatchedLevenbergMarquardt(d,B); s.setIterations(10); s.setPenaltyWeights(*problems.UNICYCLE_WEIGHTS)
    s.set_instance_data(s.init_trajectory(x0,xf), xref=xf)
    ts=[]
    for r in range(4):
        s.restore_instance_data(); s.solve(); ts.append(s.get_stats()["solve_ms"])
    print("N",N,"batch",B,"solve_ms",min(ts))
PY
