"""Does a kernel damage workgroups of OTHER kernels on its CU?  A victim kernel (tests/pollute/pollute.hip) holds patterns in its
LDS and registers and re-checks them for `ms` milliseconds on a side stream while the main stream runs the kernel under test
back to back.     python tools/victim_probe.py [aggressor=gemm_ln|gemm|ffn|attn|none] [ms=200] [victim_lds_kb=60]"""
import ctypes as C
import sys

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import torch

import ctrlsim_amd  # noqa: F401
import gpu_utils
from gpu_utils import DEV

agg = sys.argv[1] if len(sys.argv) > 1 else "gemm_ln"
ms = int(sys.argv[2]) if len(sys.argv) > 2 else 200
kb = int(sys.argv[3]) if len(sys.argv) > 3 else 60
lib = C.CDLL(gpu_utils.build_pollute_lib())
log = torch.zeros(8 + 6 * 64, dtype=torch.int32, device=DEV)
side = torch.cuda.Stream()
rs = np.random.RandomState(0)
M = 65536
A = gpu_utils.dev(rs.randn(M, 256).astype(np.float32))
W = torch.from_numpy((rs.randn(256, 256) * 0.05).astype(np.float32))
W3 = torch.from_numpy((rs.randn(768, 256) * 0.05).astype(np.float32))
R = gpu_utils.dev(rs.randn(M, 256).astype(np.float32))
g, b = gpu_utils.dev(np.ones(256, np.float32)), gpu_utils.dev(np.zeros(256, np.float32))
from ctrlsim_amd import _lib
from ctrlsim_amd.pack import split3_planes
p_ = _lib.ptr


def make(Wt, resid, ln):
    Wc = Wt.numpy()
    planes = torch.from_numpy(split3_planes(Wc).view(np.int16).copy()).to(DEV)
    out = torch.empty(M, Wc.shape[0], device=DEV)

    def call():
        _lib.check(_lib.lib().ctrlsim_gemm_nt_bf16x6(p_(A), A.stride(0), p_(planes), Wc.shape[0], 0, None, p_(R) if resid else None,
                                                     R.stride(0) if resid else 0, p_(out), out.stride(0), M, Wc.shape[0], 256, 0,
                                                     p_(g) if ln else None, p_(b) if ln else None, _lib.stream_ptr()), "gemm")
        return out
    return call


if agg == "gemm_ln":
    f = make(W, True, True)
elif agg == "gemm":
    f = make(W3, False, False)
elif agg == "rollout":            # the whole closed-loop step (every kernel of the forward pass) on the main stream
    from ctrlsim_amd import spec, weights, scenarios
    from ctrlsim_amd.engine import RolloutEngine
    cfg = spec.make_cfg(nocturne__steps=90, nocturne__history_steps=1)
    eng = RolloutEngine(cfg, weights.generate(spec.Dims(cfg), 0), DEV, max_ctx=64, seed=3, lanes=1)
    eng.load_scenarios([scenarios.make_scenario(7, i, n_agents=64, n_polylines=512) for i in range(3)], steps=90)

    def f():
        eng.reset(0, 3)
        eng.run(90)
        return None
elif agg == "none":
    f = lambda: None
else:
    raise SystemExit("aggressor " + agg)
f(); torch.cuda.synchronize()
ref = f()
ref = None if ref is None else ref.clone()
torch.cuda.synchronize()
gbuf = torch.zeros(256 * 2048 * 20, device=DEV)
assert lib.victim_launch(ms, 256, kb * 1024, 12345, C.c_void_p(log.data_ptr()), C.c_void_p(gbuf.data_ptr()), C.c_void_p(side.cuda_stream)) == 0
ev = torch.cuda.Event(); ev.record(side)
n = bad = 0
while not ev.query():
    out = f()
    n += 1
    if ref is not None and n % 8 == 0:
        bad += int((out != ref).any())
torch.cuda.synchronize()
l = log.cpu().numpy().view(np.uint32)
print(f"aggressor {agg}: {n} launches beside a {kb} KB victim for {ms} ms; victim iterations {l[2]}, LDS mismatches {l[0]}, "
      f"register mismatches {l[1]}, arithmetic mismatches {l[3]}, cross-wave global hand-over mismatches {l[4]} (own read-back {l[5]}); aggressor outputs that differ from a solo run: {bad}")
for k in range(min(int(l[0]), 48)):
    e = l[8 + 6 * k: 14 + 6 * k]
    print(f"  LDS  block {e[0]} word {e[1]} want {e[2]:#010x} got {e[3]:#010x} iteration {e[4]}")
for k in range(min(int(l[1]), 16)):
    e = l[8 + 6 * (48 + k): 14 + 6 * (48 + k)]
    print(f"  REG  block {e[0]} lane*64+reg {e[1]} want {e[2]:#010x} got {e[3]:#010x} iteration {e[4]}")
