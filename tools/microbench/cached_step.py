"""Repeats ONE K/V-cached policy step (t = 31) of a 51-scenario lane back to back and reports the GPU time per repetition."""
import sys, time
sys.path.insert(0, '.')
import torch
import ctrlsim_amd  # noqa
from ctrlsim_amd import spec, weights, scenarios
from ctrlsim_amd.engine import RolloutEngine
S = 51
cfg = spec.make_cfg(); d = spec.Dims(cfg)
eng = RolloutEngine(cfg, weights.generate(d, 0), 'cuda:0', max_ctx=512, seed=0, lanes=1)
eng.load_scenarios(scenarios.make_batch(0, range(S), n_agents=64, n_polylines=512), steps=90)
eng.reset(0, S); eng.run(31, s0=0, s1=S); torch.cuda.synchronize()
L = eng.lanes[0]
eng._main = torch.cuda.current_stream(eng.device)
side, L.side = L.side, None
eng._enqueue_groups(L, 31, 0, S); hist, _ = eng._await_groups(L)
chunks = eng._chunks(hist, 0)
print("cstep chunks", [(a, b, c) for a, b, c in chunks])
(s0, s1, counts) = chunks[0]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for _ in range(5):
    eng._chunk_step_cached(L, s0, s1, counts, 31)
torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    eng._chunk_step_cached(L, s0, s1, counts, 31)
b.record(); torch.cuda.synchronize()
print(f"cstep cached policy step t=31, {sum(counts)} contexts: {a.elapsed_time(b) / reps:.3f} ms per step")
