"""Concurrency stress (development aid): N processes repeat create / set_instance_data / eval on small random problems and compare every
result with the first one of the same problem.  python tools/stress_eval.py [procs] [rounds]"""
import multiprocessing as mp
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, rounds, q):
    import numpy as np
    from control_box_rst_amd import problems
    from control_box_rst_amd.solver import BatchedLevenbergMarquardt
    bad = 0
    rng = np.random.default_rng(rank)
    ref = {}
    for r in range(rounds):
        N = int(rng.integers(5, 60))
        key = (N, r % 7)
        d = problems.unicycle_desc(N=N) if (r % 2 or os.environ.get('STRESS_ONLY') == 'unicycle') else problems.vdp_desc(N=N)
        B = 2
        s = BatchedLevenbergMarquardt(d, B)
        s.setPenaltyWeights(5, 5, 5)
        rs = np.random.default_rng(1000 + r % 7)
        x0 = rs.uniform(-1, 1, (B, d.nx)); xf = rs.uniform(-1, 1, (B, d.nx))
        X0 = s.init_trajectory(x0, xf)
        s.set_instance_data(X0, xref=xf)
        v, j = s.eval()
        s.setIterations(3); s.solve(); X, c, st = s.get_solution()
        got = (v.copy(), j.copy(), X.copy())
        k2 = (key, r % 2)
        if k2 in ref:
            if not all(np.array_equal(a, b, equal_nan=False) for a, b in zip(ref[k2], got)):
                bad += 1
        else:
            if not np.isfinite(v).all():
                bad += 1
            ref[k2] = got
        s.close()
    q.put((rank, bad))


if __name__ == "__main__":
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    mp.set_start_method("spawn")
    q = mp.Queue()
    ps = [mp.Process(target=worker, args=(i, rounds, q)) for i in range(procs)]
    [p.start() for p in ps]
    res = [q.get() for _ in ps]
    [p.join() for p in ps]
    print("mismatches per process:", sorted(res))
