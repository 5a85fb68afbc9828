"""Throughput of the PLUGIN route (BASELINE configs[0]'s analogue): PolicyEvaluator -> AutoregressivePolicy.predict / act -> Simulation.step,
one scenario at a time through the reference-shaped surface (per step: the history mirrored to the device, one policy step, a
device -> host read of the sampled actions, one simulator step, a read of the new state row).  Plumbing, not the product's fast path
(RolloutEngine, bench.py) — this gives it a number.   usage: python tools/facade_rate.py [scenarios=3] [agents=8] [steps=20] [batched]"""
import sys
import time

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import ctrlsim_amd  # noqa: F401
from ctrlsim_amd import spec
from ctrlsim_amd.models import CtRLSim
from ctrlsim_amd.policies import AutoregressivePolicy
from ctrlsim_amd.evaluators import PolicyEvaluator

S = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
T = int(sys.argv[3]) if len(sys.argv) > 3 else 20
cfg = spec.make_cfg(nocturne__steps=T, nocturne__history_steps=1)
cfg.eval["synthetic"] = dict(num_scenarios=S + 1, n_agents=N, n_polylines=200, seed=7, extent=60.0)
cfg.eval.num_files_to_evaluate = (S + 1) * cfg.eval.partitions
model = CtRLSim(cfg, seed=0, device="cuda:0")
pol = cfg.eval.policy
policy = AutoregressivePolicy(cfg=cfg, model_path="", model=model, use_rtg=pol.use_rtg, predict_rtgs=pol.predict_rtgs,
                              discretize_rtgs=pol.discretize_rtgs, real_time_rewards=pol.real_time_rewards,
                              privileged_return=pol.privileged_return, max_return=pol.max_return, min_return=pol.min_return,
                              key_dict={"next_acceleration": "next_acceleration", "next_steering": "next_steering", "rtgs": "rtgs"},
                              tilt_dict={"tilt": True, "goal_tilt": 0, "veh_veh_tilt": 0, "veh_edge_tilt": 0}, name=pol.model,
                              action_temperature=pol.action_temperature, nucleus_sampling=pol.nucleus_sampling,
                              nucleus_threshold=pol.nucleus_threshold)
BATCHED = len(sys.argv) > 4 and sys.argv[4] == "batched"      # round 5: every scene of the evaluation in one RolloutEngine batch (the default of evaluate_policy)
cfg.eval["batched"] = BATCHED
cfg1 = spec.make_cfg(nocturne__steps=T, nocturne__history_steps=1)
cfg1.eval["batched"] = BATCHED
cfg1.eval["synthetic"] = dict(num_scenarios=1, n_agents=N, n_polylines=200, seed=7, extent=60.0)
cfg1.eval.num_files_to_evaluate = cfg1.eval.partitions
PolicyEvaluator(cfg1, policy).evaluate_policy()              # warm-up: first launches, allocations
t0 = time.perf_counter()
EV = PolicyEvaluator(cfg, policy)
m, _ = EV.evaluate_policy()
el = time.perf_counter() - t0
n_scn = S + 1
print(f"plugin route ({'batched: one RolloutEngine batch' if BATCHED else 'per-scenario loop'}): {n_scn} scenarios x {N} vehicles x {T} steps, full model, in {el:.2f} s = {n_scn * N * T / el:.0f} agent-steps/s "
      f"({el / (n_scn * T) * 1e3:.1f} ms per scenario-step)")
if BATCHED:
    print("  batched route, seconds by phase of the step loop:", {k: round(v, 3) for k, v in getattr(EV, "batched_timing", {}).items()})
