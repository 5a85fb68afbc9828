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
    env = dict(os.environ, **extra_env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--scenarios", "6", "--steps", "2", "--warmup", "0",
                        "--agents", "12", "--polylines", "40", "--rollout-steps", "8", "--max-ctx", "32", "--no-cpu-baseline",
                        "--spot-check", "2", *args], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


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
