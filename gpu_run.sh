cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
CORBO_HIP_LOOP=0 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python - <<PY
import sys, time, os; sys.path.insert(0,'.')
from control_box_rst_amd import problems
from control_box_rst_amd.solver import BatchedLevenbergMarquardt
d=problems.unicycle_desc()
for B in (1, 1024):
    x0,xf=problems.unicycle_instances(B)
    s=BatchedLevenbergMarquardt(d,B); s.setPenaltyWeights(10,10,10)
    s.set_instance_data(s.init_trajectory(x0,xf), xref=xf)
    for _ in range(3): s.restore_instance_data(); s.solve(new_run=True)
    s.synchronize(); t0=time.perf_counter()
    for _ in range(30): s.restore_instance_data(); s.solve(new_run=True)
    s.synchronize(); dt=(time.perf_counter()-t0)/30
    print(f"B={B}: {dt*1e3:.4f} ms/solve  {B*10/dt/1e6:.2f} M iter/s", flush=True)
    os.environ["CORBO_HIP_PASS_TIMELINE"]="0"; s.restore_instance_data(); s.solve(new_run=True); s.synchronize(); del os.environ["CORBO_HIP_PASS_TIMELINE"]
PY
