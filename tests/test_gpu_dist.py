"""The job's ONE collective on the hardware that exists: bench.py with torch.distributed initialised on backend "nccl" (= RCCL) at
world size 1 — init_process_group, the barriers around the timed region, the MAX all-reduce of the elapsed time and the SUM
all-reduce of the ctrlsim_metrics_pack vector all run through RCCL on the single GPU, exactly the calls the 8-GPU run makes
(the reference has no merge step: evaluators/policy_evaluator.py:466-490,578-593 write one JSON per partition)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra_env, *args):
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    env = dict(os.environ, **extra_env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--scenarios", "6", "--steps", "2", "--warmup", "0",
                        "--agents", "12", "--polylines", "40", "--rollout-steps", "8", "--max-ctx", "32", "--no-cpu-baseline",
                        "--spot-check", "2", "--detail-file", os.path.join(ROOT, "gpurun_out", "bench_detail_test.json"), *args], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    return _short_and_detail(r.stdout)


def _short_and_detail(stdout):
    """The run's ONE stdout line (what the driver parses: short, complete) and, returned, the detail file it names."""
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    # exactly one JSON line, and it is the LAST line of stdout (RCCL's version banner, printed through C stdio, must not follow it)
    assert sum(ln.startswith("{") for ln in lines) == 1 and lines[-1].startswith("{"), stdout[-2000:]
    assert len(lines[-1]) < 4096, len(lines[-1])
    short = json.loads(lines[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "parity_spot_check", "detail_file"):
        assert k in short, k
    assert "workload" in short["config"] and "model" not in short["config"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in short["roofline"], k
    with open(os.path.join(ROOT, short["detail_file"])) as f:
        detail = json.load(f)
    assert short["value"] == pytest.approx(detail["value"], rel=1e-5) and short["n_gpus"] == detail["n_gpus"]
    assert short["parity_spot_check"]["identical"] == detail["parity_spot_check"]["identical"]
    return detail


def test_bench_collective_runs_through_rccl_at_world_size_one():
    plain = _bench({})
    rccl = _bench({"CTRLSIM_BENCH_FORCE_DIST": "1"})
    assert "nccl" in rccl["config"]["collective"] and "none" in plain["config"]["collective"]
    assert rccl["n_gpus"] == 1 and rccl["value"] > 0
    # the all-reduced metric vector of one rank is the rank's own vector: same rollout metrics with and without RCCL
    # (the device accumulators add doubles with atomics: the last bit may differ from run to run)
    for k, v in plain["rollout_metrics"].items():
        assert rccl["rollout_metrics"][k] == pytest.approx(v, rel=1e-12, abs=1e-15), k
    for out in (plain, rccl):
        assert out["parity_spot_check"]["identical"] is True, out["parity_spot_check"]
        r = out["roofline"]          # (an 8-step rollout is all K/V-cached steps: its kernels run on the lanes' side streams)
        assert (r["kernels"] or r["kernels_on_side_streams"]) and r["end_to_end"]["frac"] > 0
        assert len(out["config"]["size_classes"]) == 16


def test_bench_at_world_size_two_on_the_one_gpu_that_exists():
    """Round 4: bench.py's rank > 0 branches — interleaved scenario ids, the MAX all-reduce of the elapsed time, the SUM
    all-reduce of the metric vector, the parting barrier — run with TWO ranks, launched exactly as the driver launches N > 1
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...`), both on
    the single GPU with gloo collectives (CTRLSIM_BENCH_DEBUG_SHARED_GPU=1: RCCL cannot put two ranks on one device).  Rank r rolls the
    global scenarios r, r + 2, ...: the reduced rollout metrics must equal a world-size-1 run over the union of those ids, `value` must
    count both ranks, rank 0 alone prints the JSON line and both ranks exit cleanly (the reference's equivalent is a merge-less file
    partition, evaluators/policy_evaluator.py:466-490,578-593)."""
    import socket
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    common = ["--steps", "1", "--warmup", "0", "--agents", "12", "--polylines", "40", "--rollout-steps", "8", "--max-ctx", "32",
              "--no-cpu-baseline", "--spot-check", "2", "--detail-file", os.path.join(ROOT, "gpurun_out", "bench_detail_test2.json")]
    env = dict(os.environ, CTRLSIM_BENCH_DEBUG_SHARED_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scenarios", "3", *common],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])            # rank 1 left cleanly too
    two = _short_and_detail(r.stdout)                                         # rank 0 only
    one = _bench({}, "--scenarios", "6", "--steps", "1")                      # ids 0..5 = the union of rank 0's (0, 2, 4) and rank 1's (1, 3, 5)
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and "gloo" in two["config"]["collective"]
    assert two["config"]["workload"].startswith("3 synthetic") and two["config"]["scenarios_per_gpu"] == 3
    assert two["config"]["parallelism"] == "scenario-sharded x2"
    # whole-job aggregate: 2 ranks x 3 scenarios x 12 vehicles x 8 steps over the max-over-ranks time
    assert two["value"] == pytest.approx(2 * 3 * 12 * 8 / (two["ms_per_step"] * two["steps"] / 1e3), rel=1e-6)
    for k, v in one["rollout_metrics"].items():
        assert two["rollout_metrics"][k] == pytest.approx(v, rel=1e-12, abs=1e-15), k
    assert two["parity_spot_check"]["identical"] is True
