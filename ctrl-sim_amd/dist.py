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


def allreduce_metrics(acc, device="cpu"):
    """In-place SUM over all ranks of a MetricAccumulators; no-op without an initialised process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return acc
    vec = torch.from_numpy(acc.pack()).to(device)
    dist.all_reduce(vec, op=dist.ReduceOp.SUM)
    return acc.unpack(vec.cpu().numpy())
