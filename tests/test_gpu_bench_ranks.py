"""bench.py's N > 1 path end to end on the 1-GPU box: two ranks launched exactly as the driver launches them
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 ...`), both placed on GPU 0 with gloo
standing in for RCCL (CORBO_BENCH_TEST_SHARED_GPU=1).  Checks the contract of the one JSON line and that the whole-job numbers
are the sum over the ranks' shards."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, batch=64, steps=3, warmup=1):
    env = dict(os.environ, CORBO_BENCH_TEST_SHARED_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    args = ["bench.py", "--gpus", str(world), "--steps", str(steps), "--warmup", str(warmup), "--batch", str(batch), "--no-cpu-baseline"]
    if world == 1:
        cmd = [sys.executable] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + args
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # rank 0 prints ONE JSON line, the other ranks none
    return json.loads(lines[0])


@pytest.mark.gpu
def test_two_ranks_line_and_whole_job_totals():
    one = _run(1, batch=128)
    two = _run(2, batch=64)                              # weak scaling: 64 per rank -> the same 128 global instances
    for j, n in ((one, 1), (two, 2)):
        assert j["n_gpus"] == n and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak"
        assert j["unit"] == "SQP-iterations/s" and j["dtype"] == "f64" and j["vs_baseline"] is None and j["higher_is_better"] is True
        assert j["config"]["global_batch"] == 128
        assert abs(j["value"] - j["solve_stats"]["lm_iterations"] * j["steps"] / (j["ms_per_step"] * 1e-3 * j["steps"])) <= 1e-6 * j["value"]
        assert set(j["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
        assert len(j["ms_per_step_ranks"]) == n                              # every rank's own time: a straggler GPU is visible in the line
        assert j["gather"]["matches_host_copy"] is True and j["gather"]["bytes"] == 128 * 498 * 8   # device-side all-gather of the trajectories
    # rank r owns global instances [64 r, 64 r + 64): the union is the single-rank batch, so the totals agree exactly
    # (integers) / to rounding of the summation order (chi2)
    for k in ("lm_iterations", "accepted", "rejected", "factorizations", "ok_instances"):
        assert one["solve_stats"][k] == two["solve_stats"][k], k
    assert abs(one["solve_stats"]["chi2_sum"] - two["solve_stats"]["chi2_sum"]) <= 1e-9 * abs(one["solve_stats"]["chi2_sum"])
