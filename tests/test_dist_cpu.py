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


def _bench_dry(world, scenarios_per_rank, extra_env=None, extra_args=(), launcher=True, tmp="/tmp"):
    """bench.py --dry-run launched EXACTLY as the driver launches N > 1 (python -m torch.distributed.run ... bench.py --gpus N ...), or —
    launcher=False — as plain `python bench.py --gpus N`, which must start the N ranks itself.  Returns (detail dict, wall seconds)
    after checking the ONE stdout line: short enough for the driver's parser, valid JSON, naming the detail file."""
    import json
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    detail = os.path.join(str(tmp), f"bench_detail_dry_{world}_{int(launcher)}_{os.getpid()}.json")
    args = ["bench.py", "--gpus", str(world), "--dry-run", "--scenarios", str(scenarios_per_rank), "--agents", "4", "--polylines", "8",
            "--steps", "3", "--warmup", "1", "--rollout-steps", "20", "--detail-file", detail, *extra_args]
    cmd = [sys.executable] + (["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                               "--master-port", str(port)] if (world > 1 and launcher) else []) + args
    env = dict(os.environ, OMP_NUM_THREADS="1", **(extra_env or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                     # ONE JSON line, from rank 0
    assert r.stdout.strip().splitlines()[-1] == lines[0] and len(lines[0]) < 4096     # the LAST stdout line, < 4 KB (round 5's was 27 KB)
    short = json.loads(lines[0])
    assert short["n_gpus"] == world and short["detail_file"] == detail
    with open(detail) as f:
        out = json.load(f)
    os.remove(detail)
    assert out["n_gpus"] == short["n_gpus"] and out["agent_steps_counted"] == short["agent_steps_counted"]
    return out, time.time() - t0


def test_bench_gpus_flag_means_n_ranks_without_a_launcher():
    """`python bench.py --gpus 8` with no torchrun around it starts 8 ranks itself (round-5 review: it used to run ONE rank and print
    n_gpus 1), and a --gpus that disagrees with the launcher's world size is refused."""
    import subprocess
    import sys
    out, _ = _bench_dry(8, 3, launcher=False)
    assert out["n_gpus"] == 8 and out["agent_steps_counted"] == 3 * 4 * 20 * 8
    assert "world 8" in out["config"]["collective"] and len(out["config"]["rank_elapsed_s"]["per_rank"]) == 8
    assert out["config"]["scenario_ids_rank0"] == [0, 8, 16]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "4", "--dry-run", "--scenarios", "2", "--agents", "4", "--polylines", "8", "--steps", "1"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE = 1" in r.stderr


def test_bench_short_line_of_a_full_report_stays_under_the_limit():
    """bench.short_line on a detail dict as large as round 5's (per-kernel rows of both streams, 16 per-class attention rows, satellites,
    long notes): < 4 KB, valid JSON, carries the contract's fields + roofline of the dominant kernel + cpu_baseline."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    row = lambda kind, share: {"kernel": "x" * 300, "kind": kind, "achieved": 333.3333333, "frac": 0.4000001, "unit": "TFLOP/s", "avg_launch_ms": 1.234567,
                               "launches": 1234, "time_share_of_step": share, "algorithmic_hbm_bytes_per_launch": 1.23456789e9, "traffic": 1.3e9,
                               "hbm_rate_at_algorithmic_bytes_TBps": 2.0}
    detail = {"metric": "agent-steps/sec (closed-loop rollout), 64 agents x 90 steps", "value": 142631.123456, "unit": "agent-steps/s", "n_gpus": 8, "steps": 20,
              "warmup": 5, "ms_per_step": 4137.123456, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype_short": "f32 (f16x3 split-operand MFMA, fp32 accumulate)",
              "dtype": "y" * 200, "data": "synthetic",
              "config": {"workload_tag": "BASELINE.json configs[2] per GPU: 2048 synthetic scenarios/GPU x 64 agents x 90 steps, 512 polylines, CtRL-Sim base model (random init)",
                         "workload": "w" * 600, "scenarios_per_gpu": 2048, "agents": 64, "rollout_steps": 90, "polylines": 512, "model_batch_contexts": 1024, "lanes": 2,
                         "parallelism": "scenario-sharded x8", "rank_elapsed_s": {"per_rank": [80.0] * 8}},
              "roofline": {"bound": "mfma", "peak": 833.3333, "unit": "TFLOP/s", "frac": 0.33, "kernels": [row("ffn_fused", 0.27), row("attention_causal", 0.2)] + [row("linear_plain", 0.01)] * 12,
                           "kernels_on_side_streams": [row("ffn_fused", 0.01)] * 14, "causal_attention_by_size_class": [{"n": "z" * 200}] * 16,
                           "end_to_end": {"achieved": 224.3, "frac": 0.27}, "note": "n" * 3000},
              "cpu_baseline": {"value": 10.9, "unit": "agent-steps/s", "cores": 64, "kind": "port", "sample": "s" * 500, "sample_short": "3 scenarios x 64 agents x 3 steps (126 focal-group steps, 61.0 s, oracle port, 64 threads)", "note": "n" * 600},
              "parity_spot_check": {"identical": True, "note": "n" * 300}, "rollout_metrics": {"goal": 0.1}}
    assert len(json.dumps(detail)) > 20000
    line = json.dumps(bench.short_line(detail, "bench_detail.json"))
    assert len(line) < 2048 < bench.SHORT_LINE_LIMIT
    s = json.loads(line)
    assert s["value"] == pytest.approx(142631.1, rel=1e-6) and s["roofline"]["kernel"] == "ffn_fused" and s["roofline"]["frac"] == pytest.approx(0.4, rel=1e-3)
    assert s["roofline"]["bound"] == "mfma" and s["roofline"]["traffic"] == pytest.approx(1.3e9) and s["roofline"]["attention_causal_frac"] is not None
    assert s["cpu_baseline"] == {"value": 10.9, "unit": "agent-steps/s", "cores": 64, "kind": "port", "sample": detail["cpu_baseline"]["sample_short"]}
    assert s["config"]["workload"].startswith("BASELINE.json configs[2]") and "model" not in s["config"] and s["dtype"].startswith("f32")


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
