"""Per-segment accounting of the shipped feed-forward mode (MODE 1: out-projection + LayerNorm in front, ctrlsim_ffn_fused_pre), like qp_stamps.py:
apply tools/microbench/patches/pre_stamp.patch to a scratch copy of csrc/ffn_fused.hip, build with CTRLSIM_VARIANT=preS CTRLSIM_EXTRA_DEFS=-DQP_STAMP,
run with CTRLSIM_LIB=tools/microbench/variants/all_preS.so."""
import ctypes, sys
sys.path.insert(0, '.')
import numpy as np
import torch
import ctrlsim_amd  # noqa: F401
from ctrlsim_amd import _lib
from ctrlsim_amd.pack import ffn_planes_pre
DEV = 'cuda:0'
lib = _lib.lib(); p = _lib.ptr; st = _lib.stream_ptr()
raw = ctypes.CDLL(_lib.LIB_PATH)
g = torch.Generator().manual_seed(1)
F = 1024
Wo = torch.randn(256, 256, generator=g) * 0.07; W1 = torch.randn(F, 256, generator=g) * 0.05; W2 = torch.randn(256, F, generator=g) * 0.05
bo, g0, be0, b2, gam, bet = (torch.randn(256, generator=g).to(DEV) * s for s in (0.3, 1.0, 0.2, 0.4, 1.0, 0.1))
b1 = torch.randn(F, generator=g).to(DEV) * 0.3
dev = lambda a: torch.from_numpy(a.view(np.int16).copy()).to(DEV)
wod, w1q, w2d = (dev(a) for a in ffn_planes_pre(Wo.numpy(), W1.numpy(), W2.numpy(), 1))
M = int(sys.argv[1]) if len(sys.argv) > 1 else 256 * 2304
O = torch.randn(M, 256, generator=g).to(DEV); R = torch.randn(M, 256, generator=g).to(DEV); Y = torch.empty_like(R)
call = lambda: lib.ctrlsim_ffn_fused_pre(p(O), 256, p(R), 256, p(wod), p(bo), p(g0), p(be0), p(w1q), p(b1), p(w2d), p(b2), p(gam), p(bet), p(Y), 256, M, F, st)
for _ in range(10): call()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
raw.qp_stamps_read(buf)
n = 30
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(n): call()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / n
raw.qp_stamps_read(buf)
names = ["rows requested, waited for, converted", "barrier (DMA blocks 0, 1)", "leading product (8 blocks; residual rows arrive)", "LayerNorm + fragments", "feed-forward block (64 blocks)", "epilogue (LDS transposition, LayerNorm, stores)"]
rb = n * ((M + 127) // 128)
tot = sum(buf[i] for i in range(6))
for i, nm in enumerate(names):
    print(f"{nm:52s} {buf[i] / rb:9.0f} ticks per row block  {100 * buf[i] / tot:5.1f} %")
print(f"total {tot / rb:.0f} ticks per row block; {ms:.3f} ms per launch of {M} rows (stamped build)")
