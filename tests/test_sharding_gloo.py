"""world_size-2 gloo test of the multi-GPU path (CPU): contiguous batch sharding, statistic reduction, max-time and
trajectory gather.  The per-rank compute is stood in for by the oracle (test infrastructure) -- on the GPU box the same
sharding code drives the HIP solver in bench.py."""
import os
import socket

import numpy as np
import pytest

from control_box_rst_amd import capi, problems, sharding


def test_shard_bounds_cover_without_overlap():
    for g, w in [(8192, 8), (1024, 1), (10, 3), (7, 8), (0, 2)]:
        seen = []
        for r in range(w):
            first, count = sharding.shard_bounds(g, w, r)
            seen += list(range(first, first + count))
        assert seen == list(range(g))
    with pytest.raises(ValueError):
        sharding.shard_bounds(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, global_batch, tmp):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        d = problems.unicycle_desc(N=12)
        opts = capi.default_lm_opts(4, 10, 10, 10)
        first, count = sharding.shard_bounds(global_batch, world, rank)
        x0, xf = problems.unicycle_instances(count, first=first)
        p = O.OracleProblem(d)
        X0 = np.stack([p.init_trajectory(x0[b], xf[b]) for b in range(count)])
        X, chi2, status = O.solve_batch(d, X0, xf, opts)
        local = {k: 0 for k in sharding.STAT_KEYS}
        local["lm_iterations"] = count * opts.iterations
        red = sharding.reduce_stats(local, float(chi2.sum()), int((status <= 1).sum()), dist)
        tmax = sharding.reduce_max(1.0 + rank, dist)
        tall = sharding.gather_scalars(1.0 + rank, dist)
        allx = sharding.gather_trajectories(X, global_batch, dist)
        np.savez(os.path.join(tmp, f"r{rank}.npz"), allx=allx, tmax=tmax, tall=np.array(tall), iters=red["lm_iterations"], chi2=red["chi2_sum"], ok=red["ok_instances"])
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process(tmp_path, oracle_mod):
    import torch.multiprocessing as mp
    G, world = 5, 2  # uneven split: 3 + 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, G, str(tmp_path)), nprocs=world, join=True)
    d = problems.unicycle_desc(N=12)
    opts = capi.default_lm_opts(4, 10, 10, 10)
    x0, xf = problems.unicycle_instances(G)
    p = oracle_mod.OracleProblem(d)
    X0 = np.stack([p.init_trajectory(x0[b], xf[b]) for b in range(G)])
    X, chi2, status = oracle_mod.solve_batch(d, X0, xf, opts)
    for r in range(world):
        z = np.load(tmp_path / f"r{r}.npz")
        assert np.array_equal(z["allx"], X)                # gathered global result == single-process result, bit for bit
        assert z["tmax"] == 2.0                            # MAX over ranks
        assert np.array_equal(z["tall"], [1.0, 2.0])       # every rank's own value, in rank order
        assert z["iters"] == G * opts.iterations and z["ok"] == G
        assert abs(z["chi2"] - chi2.sum()) < 1e-9
