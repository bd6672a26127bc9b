cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python tools/profile_cfg5.py 512 3
python - <<PY
import sys; sys.path.insert(0,'.')
from control_box_rst_amd import problems
from control_box_rst_amd.solver import BatchedLevenbergMarquardt
d=problems.quad_desc(); B=512
x0,xf=problems.quad_instances(B)
s=BatchedLevenbergMarquardt(d,B); s.setPenaltyWeights(*problems.QUAD_WEIGHTS)
s.set_instance_data(s.init_trajectory(x0,xf), xref=xf)
ms=s.time_sweep(True,5); b=8*(s.dims.nv+2*s.dims.n+s.dims.m+s.dims.nnz)
print("cfg5 sweep ms/launch", ms, "values-only", s.time_sweep(False,5), "factor ms", s.time_factor(repeat=5), "B_sweep", b, "GB/s", B*b/ms/1e6)
PY
