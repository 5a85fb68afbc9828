"""Multi-GPU layout of the rollout: scenarios are independent, so ranks shard them and never exchange data on the
rollout path; the ONLY collective is one all-reduce (SUM) of the packed metric accumulators after the rollouts
(RCCL over xGMI with backend "nccl", gloo in CPU tests).  Payload ~10 KB: latency-bound, one call, no ring of
per-metric calls (the reference has no merge step at all: policy_evaluator.py:466-490,578-593)."""
from __future__ import annotations

import numpy as np
import torch


def shard_ids(rank: int, world: int, per_rank: int):
    """Global scenario ids of this rank: interleaved (r, r+W, r+2W, ...) so a fixed global id maps to the same
    scenario and the same sampling-noise stream whatever the world size."""
    return [rank + i * world for i in range(per_rank)]


TILT_SWEEP = (-20.0, -10.0, -5.0, 0.0, 5.0, 10.0, 20.0, 30.0)      # the eight tilt values of BASELINE configs[4]


def rank_plan(rank: int, world: int, per_rank: int, tilt_sweep: bool = False):
    """What rank `rank` of `world` rolls: (global scenario ids, per-scenario tilt triples [per_rank, 3] or None).  The tilt of a
    scenario follows its GLOBAL id (id % 8 picks the sweep value for goal = vehicle = road-edge tilt), so the sweep — like the
    scenarios and their noise streams — does not depend on the world size."""
    ids = shard_ids(rank, world, per_rank)
    tilt = None
    if tilt_sweep:
        tilt = np.repeat(np.asarray(TILT_SWEEP)[np.asarray(ids) % len(TILT_SWEEP)][:, None], 3, axis=1)
    return ids, tilt


def gather_scalar(dist, value: float, device="cpu"):
    """[value of rank 0, ..., value of rank W-1] on every rank (all_gather of one double); [value] without a process group."""
    if dist is None:
        return [float(value)]
    mine = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]


def allreduce_metrics(acc, device="cpu"):
    """In-place SUM over all ranks of a MetricAccumulators; no-op without an initialised process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return acc
    vec = torch.from_numpy(acc.pack()).to(device)
    dist.all_reduce(vec, op=dist.ReduceOp.SUM)
    return acc.unpack(vec.cpu().numpy())
