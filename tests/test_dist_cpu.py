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


def _bench_dry(world, scenarios_per_rank, extra_env=None, extra_args=()):
    """bench.py --dry-run launched EXACTLY as the driver launches N > 1 (python -m torch.distributed.run ... bench.py --gpus N ...)."""
    import json
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    args = ["bench.py", "--gpus", str(world), "--dry-run", "--scenarios", str(scenarios_per_rank), "--agents", "4", "--polylines", "8",
            "--steps", "3", "--warmup", "1", "--rollout-steps", "20", *extra_args]
    cmd = [sys.executable] + (["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                               "--master-port", str(port)] if world > 1 else []) + args
    env = dict(os.environ, OMP_NUM_THREADS="1", **(extra_env or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                     # ONE JSON line, from rank 0
    return json.loads(lines[0]), time.time() - t0


def test_bench_rank_flow_dry_run_world_8():
    """Round 5 (round-4 review, item 7): no 8-GPU node is available to the builder, so everything rank-dependent in bench.py is rehearsed on
    the CPU at world size 8 over gloo, launched as the driver launches it: interleaved scenario ids, the tilt sweep following the GLOBAL
    scenario id, ranks 3 and 5 halving their model batch after an out-of-memory at engine construction while the others keep theirs, the
    barriers around the timed region, MAX / gather of the elapsed times (rank 3 is a straggler by construction), the SUM all-reduce of the
    1250-double metric vector, the parting barrier, and rank 0 alone spending seconds on its CPU sample AFTER the others have left the
    process group (no collective is pending, so no rendezvous timeout can fire)."""
    from ctrlsim_amd.dist import TILT_SWEEP
    out, wall = _bench_dry(8, 6, {"CTRLSIM_BENCH_DRY_OOM_RANKS": "3,5", "CTRLSIM_BENCH_DRY_CPU_S": "4"}, ("--tilt-sweep",))
    c = out["config"]
    assert out["dry_run"] is True and out["value"] is None and out["n_gpus"] == 8
    assert c["scenario_ids_rank0"] == [0, 8, 16, 24, 32, 40]
    assert c["tilt_rank0"] == [TILT_SWEEP[i % 8] for i in c["scenario_ids_rank0"]]
    assert c["model_batch_contexts_per_rank"] == [1024, 1024, 1024, 512, 1024, 512, 1024, 1024] and c["model_batch_reduced"] is True
    per = c["rank_elapsed_s"]["per_rank"]
    assert len(per) == 8 and per.index(max(per)) == 3 and c["rank_elapsed_s"]["max"] == max(per) and c["rank_elapsed_s"]["min"] == min(per)
    assert out["ms_per_step"] * out["steps"] * 1e-3 >= max(per)                  # the reported time is the MAX over ranks (barrier to barrier)
    assert "world 8" in c["collective"] and "gloo" in c["collective"]
    assert out["agent_steps_counted"] == 6 * 4 * 20 * 8
    # the reduced metric vector equals ONE process rolling the union of the ids (0 .. 47), whatever the sharding
    one, _ = _bench_dry(1, 48)
    np.testing.assert_allclose(out["metric_vector"], one["metric_vector"], rtol=1e-12, atol=1e-12)
    assert one["config"]["scenario_ids_rank0"] == list(range(48))
