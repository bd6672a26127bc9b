"""Batch sharding across the GPUs of one node.

OCP instances are independent NLPs (the reference solves them one after another,
src/tasks/src/benchmark_task_varying_initial_state.cpp:74-99), so the batch is the sharding unit: rank r owns a contiguous
slice of the global batch and runs the whole inner loop on its own GPU.  There is NO collective on the data path; RCCL
(torch.distributed backend "nccl") / gloo is used only for the barrier, the max-over-ranks time, the reduction of the
solution statistics and an optional all-gather of the final trajectories.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np


def shard_bounds(global_batch: int, world: int, rank: int) -> Tuple[int, int]:
    """(first, count) of the contiguous slice owned by `rank`; remainders go to the low ranks."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(global_batch, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


STAT_KEYS = ("lm_iterations", "accepted_steps", "rejected_steps", "jacobian_sweeps", "residual_sweeps", "factorizations", "counted_iterations")


def reduce_stats(local: Dict[str, float], chi2_sum: float, ok_instances: int, dist=None, device="cpu") -> Dict[str, float]:
    """SUM-reduce the per-rank solver statistics (tens of bytes: one tiny all-reduce)."""
    import torch
    vec = torch.tensor([float(local[k]) for k in STAT_KEYS] + [float(chi2_sum), float(ok_instances)], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
    out = {k: float(vec[i].item()) for i, k in enumerate(STAT_KEYS)}
    out["chi2_sum"] = float(vec[len(STAT_KEYS)].item())
    out["ok_instances"] = float(vec[len(STAT_KEYS) + 1].item())
    return out


def reduce_max(value: float, dist=None, device="cpu") -> float:
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_scalars(value: float, dist=None, device="cpu"):
    """Every rank's value, in rank order (one tiny all-gather): makes a straggler GPU visible in the one bench line."""
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(value)]
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return [float(p.item()) for p in parts]


def gather_trajectories(x_local: np.ndarray, global_batch: int, dist=None, device="cpu") -> np.ndarray:
    """All-gather the final trajectories [count][nv] of every rank into [global_batch][nv] (outside any timed region)."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return np.array(x_local, copy=True)
    world, rank = dist.get_world_size(), dist.get_rank()
    nv = x_local.shape[1]
    maxc = max(shard_bounds(global_batch, world, r)[1] for r in range(world))
    pad = torch.zeros((maxc, nv), dtype=torch.float64, device=device)
    pad[: x_local.shape[0]] = torch.as_tensor(x_local, dtype=torch.float64, device=device)
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    out = np.empty((global_batch, nv))
    for r in range(world):
        first, count = shard_bounds(global_batch, world, r)
        out[first:first + count] = parts[r][:count].cpu().numpy()
    return out


def gather_trajectories_device(solver, global_batch: int, dist=None):
    """All-gather of the final trajectories straight from device memory: every rank contributes its handle's resident iterate array (a
    torch view of the library's HBM buffer, BatchedLevenbergMarquardt.device_tensor) to ONE collective (RCCL all_gather_into_tensor over
    xGMI under backend "nccl"); no host round trip.  Uneven shards are padded to the largest one.  Returns a CUDA tensor
    [global_batch][nv] on every rank."""
    import torch
    solver.synchronize()
    view = solver.device_tensor()                      # [count][row_stride], aliases the handle's buffer
    nv = solver.dims.nv
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return view[:, :nv].clone()
    world = dist.get_world_size()
    maxc = max(shard_bounds(global_batch, world, r)[1] for r in range(world))
    send = view
    if view.shape[0] != maxc:
        send = torch.zeros((maxc, view.shape[1]), dtype=view.dtype, device=view.device)
        send[: view.shape[0]] = view
    recv = torch.empty((world * maxc, view.shape[1]), dtype=view.dtype, device=view.device)
    try:
        dist.all_gather_into_tensor(recv, send.contiguous())
    except (RuntimeError, NotImplementedError):   # (backends without the flat variant, e.g. gloo in the shared-GPU test)
        parts = list(recv.view(world, maxc, view.shape[1]).unbind(0))
        dist.all_gather(parts, send.contiguous())
    out = torch.empty((global_batch, nv), dtype=view.dtype, device=view.device)
    for r in range(world):
        first, count = shard_bounds(global_batch, world, r)
        out[first:first + count] = recv[r * maxc: r * maxc + count, :nv]
    return out
