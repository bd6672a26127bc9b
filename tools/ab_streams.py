import sys, time, threading; sys.path.insert(0,'.')
import numpy as np
from control_box_rst_amd import problems
from control_box_rst_amd.solver import BatchedLevenbergMarquardt
d=problems.unicycle_desc()
def mk(B, first):
    x0,xf=problems.unicycle_instances(B, first=first)
    s=BatchedLevenbergMarquardt(d,B); s.setIterations(10); s.setPenaltyWeights(10,10,10)
    s.set_instance_data(s.init_trajectory(x0,xf), xref=xf); return s
for nsplit in (1,2,4):
    B=1024//nsplit
    hs=[mk(B, i*B) for i in range(nsplit)]
    def work(s, n):
        for _ in range(n):
            s.restore_instance_data(); s.solve()
    for s in hs: work(s,2)
    t=time.perf_counter()
    th=[threading.Thread(target=work,args=(s,20)) for s in hs]
    [x.start() for x in th]; [x.join() for x in th]
    for s in hs: s.synchronize()
    dt=(time.perf_counter()-t)/20
    print(f"split {nsplit} x {B}: {dt*1e3:.3f} ms per 1024-batch solve -> {10240/dt/1e6:.2f} M iter/s")
