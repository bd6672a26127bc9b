"""Discrete-event model of the headline launch (1024 instances, four workgroups per CU): what could a different placement of the instances buy?

Pass sequences per instance come from the oracle (inner passes per outer iteration: rejects + the accepted one); a workgroup that shares its CU
with r - 1 others runs a pass in T1 (1 + a (r - 1)) cycles (T1 = 45 k reject / 52 k accept alone, 59 - 100 k with four: a = 0.2).
Compared: the shipping placement (instance i on CU i mod 256, everything resident from the start), an idealised tail migration (whenever two
CUs differ by >= 2 residents the longest remaining chain moves, at a cost in k-cycles), the throughput bound and the longest chain alone.
    python tools/schedule_model.py [a] [migration cost]
"""
import sys
import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))


def pass_sequences():
    import bench
    from oracle import oracle as O
    from control_box_rst_amd import capi
    w = bench.workload(3, 1024)
    opts = capi.default_lm_opts(10, *w["weights"])
    out = []
    for b in range(1024):
        p = O.OracleProblem(w["desc"])
        p.set_data(p.init_trajectory(w["x0"][b], w["xf"][b]), xref=w["xf"][b])
        _, _, tr = p.solve(opts)
        out.append([t["inner_passes"] for t in tr])
    return np.array(out)


def simulate(work, a, migrate=False, cost=6.0, ncu=256):
    slow = lambda r: 1 + a * (r - 1)
    rem = [sum(w) for w in work]
    cu = [[] for _ in range(ncu)]
    for i in range(len(work)):
        cu[i % ncu].append(i)
    t, moves, alive = 0.0, 0, len(work)
    while alive:
        dt = min(min(rem[i] for i in c) * slow(len(c)) for c in cu if c)
        for c in cu:
            if not c:
                continue
            s = slow(len(c))
            for i in c:
                rem[i] -= dt / s
            for i in [i for i in c if rem[i] <= 1e-9]:
                c.remove(i)
                alive -= 1
        t += dt
        while migrate:
            lens = [len(c) for c in cu]
            hi, lo = int(np.argmax(lens)), int(np.argmin(lens))
            if lens[hi] - lens[lo] < 2:
                break
            i = max(cu[hi], key=lambda j: rem[j])
            cu[hi].remove(i)
            cu[lo].append(i)
            rem[i] += cost
            moves += 1
    return t, moves


if __name__ == "__main__":
    a = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
    cost = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
    seqs = pass_sequences()
    work = [sum(([45.0] * (n - 1) + [52.0] for n in s), []) for s in seqs]
    tot = sum(sum(w) for w in work)
    base, _ = simulate(work, a)
    mig, moves = simulate(work, a, True, cost)
    print(f"passes {int(seqs.sum())} (min {seqs.sum(1).min()}, mean {seqs.sum(1).mean():.1f}, max {seqs.sum(1).max()})")
    print(f"shipping placement {base:.0f} k cycles | tail migration (cost {cost:.0f} k) {mig:.0f} k ({moves} moves, {100 * (1 - mig / base):.1f} %) | "
          f"throughput bound {tot * (1 + 3 * a) / 4 / 256:.0f} k | longest chain alone {max(sum(w) for w in work):.0f} k")
