"""Measured proxies for the fusion proposals sized in profiles/r05_fusion_k256.md (sustained, 589 824 rows):
  (b) out-proj + LN2 as a leading product inside the fused FFN  ~  the fused FFN with 4 more hidden blocks' worth of matrix work and weight
      stream (F = 1152 instead of 1024: +12.5 % MFMAs, +12.5 % LDS-DMA bytes — the leading 256 x 256 product has exactly the cost of 4 half-pairs)
  (a) out-proj + LN1 + cross-Q in one row-stationary kernel     ~  the row-stationary kernel as a plain Linear with N = 512 output columns (rows read
      once, two 256-column products' worth of weight stream and MFMAs, 2 KB per row of fp32 stores) — without the LayerNorm dependency between
      the two products, i.e. a LOWER bound of the fused kernel's time
against the kernels they would replace.  and, since both were built later in the round, the fused kernels themselves.  usage: python tools/microbench/fusion_proxies.py [B=256]"""
import sys
import time
sys.path.insert(0, '.')
import numpy as np
import torch
import ctrlsim_amd  # noqa: F401
from ctrlsim_amd import _lib
from ctrlsim_amd.pack import split3_planes, ffn_planes, row_blocks

DEV = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M = B * 2304
lib = _lib.lib(); p = _lib.ptr; st = _lib.stream_ptr()
g = torch.randn(256, device=DEV)


def sustained(fn, secs=1.2):
    def burst():
        n, t0 = 0, time.perf_counter()
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        while True:
            for _ in range(8):
                fn()
            n += 8
            torch.cuda.synchronize()
            if time.perf_counter() - t0 > secs:
                break
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / n
    burst()
    return burst()


X = torch.randn(M, 256, device=DEV); Y = torch.empty_like(X); R = torch.randn(M, 256, device=DEV)
b2 = torch.randn(256, device=DEV)
res = {}
for F in (1024, 1152, 1280):
    W1 = torch.randn(F, 256) * 0.05; W2 = torch.randn(256, F) * 0.05
    w1p, w2p = ffn_planes(W1.numpy(), W2.numpy())
    w1d = torch.from_numpy(w1p.view(np.int16).copy()).to(DEV); w2d = torch.from_numpy(w2p.view(np.int16).copy()).to(DEV)
    b1 = torch.randn(F, device=DEV)
    res[f"ffn F={F}"] = sustained(lambda: lib.ctrlsim_ffn_fused(p(X), 256, p(w1d), p(b1), p(w2d), p(b2), p(g), p(g), p(Y), 256, M, F, st))
W = torch.randn(256, 256) * 0.05
planes = torch.from_numpy(split3_planes(W.numpy()).view(np.int16).copy()).to(DEV)
res["ws256 out-proj + residual + LN"] = sustained(lambda: lib.ctrlsim_gemm_nt_bf16x6(p(X), 256, p(planes), 256, 0, p(b2), p(R), 256, p(Y), 256, M, 256, 256, 0, p(g), p(g), st))
res["ws256 plain 256 -> 256"] = sustained(lambda: lib.ctrlsim_gemm_nt_bf16x6(p(X), 256, p(planes), 256, 0, p(b2), None, 0, p(Y), 256, M, 256, 256, 0, None, None, st))
for N in (256, 512):
    Wn = torch.randn(N, 256) * 0.05; bn = torch.randn(N, device=DEV)
    blk = torch.from_numpy(row_blocks(Wn.numpy(), 1).view(np.int16).copy()).to(DEV)
    C = torch.empty(M, N, device=DEV)
    res[f"row-stationary plain 256 -> {N}"] = sustained(lambda: lib.ctrlsim_gemm_kv_blocks(p(X), 256, p(blk), p(bn), p(C), N, M, N, None, 0, 0, 0, st))
    del C
# the fused kernels as built (round 5, options 3 = 2 / 3)
from ctrlsim_amd.pack import ffn_planes_pre, outproj_q_planes
O = torch.randn(M, 256, device=DEV)
W1 = torch.randn(1024, 256) * 0.05; W2 = torch.randn(256, 1024) * 0.05; Wq = torch.randn(256, 256) * 0.05
dev = lambda a: torch.from_numpy(a.view(np.int16).copy()).to(DEV)
wod, w1q, w2d = (dev(a) for a in ffn_planes_pre(W.numpy(), W1.numpy(), W2.numpy(), 1))
b1 = torch.randn(1024, device=DEV)
res["BUILT (b): out-proj + LN + ffn F=1024"] = sustained(lambda: lib.ctrlsim_ffn_fused_pre(p(O), 256, p(X), 256, p(wod), p(b2), p(g), p(g), p(w1q), p(b1), p(w2d), p(b2), p(g), p(g), p(Y), 256, M, 1024, st))
wod2, wqd = (dev(a) for a in outproj_q_planes(W.numpy(), Wq.numpy(), 1))
Q = torch.empty(M, 256, device=DEV)
res["BUILT (a): out-proj + LN + q"] = sustained(lambda: lib.ctrlsim_outproj_ln_q(p(O), 256, p(X), 256, p(wod2), p(b2), p(g), p(g), p(wqd), p(b2), p(Y), 256, p(Q), 256, M, st))
for k, v in res.items():
    print(f"{k:40s} {v:7.3f} ms")
print(f"(b) proxy: FFN(1152) - FFN(1024) = {res['ffn F=1152'] - res['ffn F=1024']:.3f} ms of extra fused work against {res['ws256 out-proj + residual + LN']:.3f} ms for the separate kernel")
print(f"(a) proxy: row-stationary 256 -> 512 = {res['row-stationary plain 256 -> 512']:.3f} ms against {res['ws256 out-proj + residual + LN'] + res['ws256 plain 256 -> 256']:.3f} ms for the two separate kernels")
