import sys; sys.path.insert(0,'.'); sys.path.insert(0,'oracle')
import numpy as np, torch
import ctrlsim_amd
from ctrlsim_amd import spec, weights, scenarios
from ctrlsim_amd.engine import RolloutEngine
cfg = spec.make_cfg(nocturne__steps=90, nocturne__history_steps=1)
d = spec.Dims(cfg); w = weights.generate(d, 0)
scns = scenarios.make_batch(0, range(48), n_agents=64, n_polylines=512)
eng = RolloutEngine(cfg, w, "cuda:0", max_ctx=512, seed=0, lanes=1)
eng.load_scenarios(scns, steps=90)
hist = np.zeros((90, 26), np.int64)
orig = eng._enqueue_groups
def spy(L, t, s0, s1, compare=False):
    orig(L, t, s0, s1, compare)
    torch.cuda.synchronize()
    ng = eng.n_groups[s0:s1].cpu().numpy(); ids = eng.grp_ids[s0:s1].cpu().numpy()
    for s in range(s1 - s0):
        for g in range(ng[s]):
            hist[t, bin(int(ids[s, g]) & (2**64 - 1)).count("1")] += 1
eng._enqueue_groups = spy
eng.run(90)
for t in (0, 16, 31, 32, 45, 60, 75, 89):
    h = hist[t]; n = np.arange(26)
    print(t, "contexts", h.sum(), "mean n", round((h * n).sum() / h.sum(), 2), h.tolist())
tot = hist[32:].sum(0)
print("sliding phase n distribution:", tot.tolist(), "mean", (tot * np.arange(26)).sum() / tot.sum())
np.save("gpurun_out/nhist.npy", hist)
