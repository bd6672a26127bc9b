cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12
python - <<'PY'
import sys,time; sys.path.insert(0,'.')
import numpy as np
from control_box_rst_amd import problems
from control_box_rst_amd.solver import BatchedLevenbergMarquardt
d=problems.quad_desc(); B=512
x0,xf=problems.quad_instances(B)
s=BatchedLevenbergMarquardt(d,B); s.setIterations(10); s.setPenaltyWeights(*problems.QUAD_WEIGHTS)
X0=s.init_trajectory(x0,xf); X0[:,12::16]=9.81; s.set_instance_data(X0,xref=xf)
s.set_profiling(True); s.solve(); print('cfg5 B=512 N=200:', s.get_stats())
print('sweep J', s.time_sweep(True,5)*1e3,'us  factor', s.time_factor(5)*1e3, 'us')
PY
