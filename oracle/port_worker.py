"""Test infrastructure (CPU baseline leg of bench.py): one worker process of the all-cores run of the C restatement.

    python oracle/port_worker.py <first> <count> <iterations>      -> one JSON line {"n", "iterations", "seconds"}

Solves `count` seeded cfg-3 instances (global indices first .. first+count-1, same generator as the GPU run) sequentially with
oracle/liboracle.so.  bench.py starts one worker per host core at the same time (processes, not threads: in-process threads of
the C library were measured not to scale on the build container).
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[0] = ROOT  # replaces the script directory: `oracle` must resolve to the package, not to oracle/oracle.py
import numpy as np  # noqa: E402

from control_box_rst_amd import capi, problems  # noqa: E402  (ctypes structures only; the HIP library is not loaded)
from oracle import oracle as O  # noqa: E402


def main():
    first, count, iterations = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    d = problems.unicycle_desc()
    x0, xf = problems.unicycle_instances(count, first=first)
    o = capi.LmOpts()
    o.iterations = iterations
    o.weight_eq, o.weight_ineq, o.weight_bounds = problems.UNICYCLE_WEIGHTS
    o.adapt_factor_eq = o.adapt_factor_ineq = o.adapt_factor_bounds = 1.0
    o.adapt_max_eq = o.adapt_max_ineq = o.adapt_max_bounds = 500.0
    p = O.OracleProblem(d)
    X = np.stack([p.init_trajectory(x0[b], xf[b]) for b in range(count)])
    t0 = time.perf_counter()
    O.solve_batch(d, X, xf, o)
    print(json.dumps({"n": count, "iterations": iterations, "seconds": time.perf_counter() - t0}))


if __name__ == "__main__":
    main()
