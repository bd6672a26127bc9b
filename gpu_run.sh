cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python - <<'PY'
import sys; sys.path.insert(0,'.')
import numpy as np
from control_box_rst_amd import problems
from control_box_rst_amd.solver import BatchedLevenbergMarquardt
d=problems.unicycle_desc()
for B in (1,256,1024,2048,4096):
    x0,xf=problems.unicycle_instances(B)
    s=BatchedLevenbergMarquardt(d,B); s.setPenaltyWeights(10,10,10)
    s.set_instance_data(s.init_trajectory(x0,xf), xref=xf)
    ms, tl = s.time_factor(repeat=20, timeline=True)
    sw = s.time_sweep(True, 20); sv = s.time_sweep(False, 20)
    print(f"B={B}: factor {ms*1e3:.1f} us  sweep(J) {sw*1e3:.1f} us  sweep(values) {sv*1e3:.1f} us; timeline(cycles) {[tl[i+1]-tl[i] for i in range(7)]}")
PY
python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ['value','ms_per_step','batch_steps_per_s','solve_stats','kernel_split_ms','roofline']})"
