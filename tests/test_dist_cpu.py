"""world_size-2 gloo test of the only collective of the rollout: the packed metric all-reduce (CPU, no GPU)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ctrlsim_amd import metrics, spec, scenarios
from ctrlsim_amd.dist import allreduce_metrics, shard_ids


def _fake_rollout(cfg, scn, seed):
    rs = np.random.RandomState(seed)
    N, T1 = scn.N, cfg.nocturne.steps + 1
    st = np.zeros((N, T1, 8))
    st[..., :2] = np.stack([scn.x, scn.y], 1)[:, None] + np.cumsum(rs.normal(0, 0.5, (N, T1, 2)), 1)
    st[..., 2:4] = rs.normal(0, 3, (N, T1, 2)); st[..., 4] = rs.uniform(-3, 3, (N, T1)); st[..., 7] = 1
    coll = (rs.uniform(size=(N, T1, 2)) < 0.01).astype(np.uint8)
    gt = np.zeros((N, T1, 5)); gt[..., :2] = st[..., :2] + rs.normal(0, 1, (N, T1, 2)); gt[..., 3] = 5; gt[..., 4] = 1
    return st, coll, rs.uniform(-10, 10, (N, T1)), gt


def _accumulate(cfg, ids):
    acc = metrics.MetricAccumulators()
    for i in ids:
        scn = scenarios.make_scenario(1, i, n_agents=6, n_polylines=8)
        st, coll, accel, gt = _fake_rollout(cfg, scn, 100 + i)
        acc.add_scenario(st, coll, accel, gt, scn.goal_pos.astype(float), scn.goal_heading.astype(float),
                         scn.goal_speed.astype(float), cfg)
    return acc


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = spec.make_cfg(nocturne__steps=20, nocturne__history_steps=1)
    acc = _accumulate(cfg, shard_ids(rank, world, 3))
    allreduce_metrics(acc)
    if rank == 0:
        q.put(acc.pack())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_metric_allreduce_equals_single_process(world):
    """The only collective of a multi-GPU run — one SUM all-reduce of the packed accumulators — over `world` gloo ranks holding
    the interleaved scenario shards (shard_ids) equals the single-process accumulation over the union."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=90)
        assert p.exitcode == 0
    cfg = spec.make_cfg(nocturne__steps=20, nocturne__history_steps=1)
    ref = _accumulate(cfg, range(3 * world))              # ids 0..3W-1 == union of the interleaved shards
    np.testing.assert_allclose(got, ref.pack(), rtol=1e-12, atol=1e-12)
    m, lines = metrics.MetricAccumulators().unpack(got).compute()
    assert set(m) == {"goal", "collision_rate", "offroad_rate", "fde", "ade", "lin_speed_jsd", "ang_speed_jsd",
                      "accel_jsd", "nearest_dist_jsd"}     # evaluators/policy_evaluator.py:251-305


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_shard_ids_partition(world):
    """Interleaved shards: every global id on exactly one rank, and a given id at the same place of its rank's list for every
    world size that divides it the same way (id = rank + i * world)."""
    shards = [shard_ids(r, world, 5) for r in range(world)]
    assert sorted(sum(shards, [])) == list(range(5 * world))
    for r, sh in enumerate(shards):
        assert all(g % world == r for g in sh)
