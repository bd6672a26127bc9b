cd $GRAFT_REPO_ROOT
python - <<PY
import sys, time, os; sys.path.insert(0,'.')
from control_box_rst_amd import problems
from control_box_rst_amd.solver import BatchedLevenbergMarquardt
d=problems.unicycle_desc()
for B in (1, 1):
    x0,xf=problems.unicycle_instances(B)
    s=BatchedLevenbergMarquardt(d,B); s.setPenaltyWeights(10,10,10)
    s.set_instance_data(s.init_trajectory(x0,xf), xref=xf)
    ms, tl = s.time_factor(repeat=10, timeline=True)
    print(f"B={B}: CR {tl[4]-tl[3]}; level h=4: setup {tl[8]-tl[3]}?? loads {tl[9]-tl[8]} fma {tl[10]-tl[9]} chol {tl[11]-tl[10]} solves {tl[12]-tl[11]} stores {tl[13]-tl[12]} barrier {tl[14]-tl[13]}")
PY
