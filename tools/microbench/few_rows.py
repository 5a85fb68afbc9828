"""Few-row Linear(256 -> 256) launches (second pass, last layer on the queried rows, K/V-cached steps: M = 8-25 k rows per launch in the rollout):
the weight-stationary kernel (a persistent workgroup per CU loads its 256 KB of weight fragments into registers first) against the tiled kernel,
per launch, alone on the device.  ctrlsim_set_option(6, mask): bit 1 = weight-stationary kernel for M < 16384 rows, bit 0 for larger ones.
Usage: python tools/microbench/few_rows.py"""
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
import ctrlsim_amd  # noqa: F401
from ctrlsim_amd import _lib
from ctrlsim_amd.pack import split3_planes

DEV = 'cuda:0'
lib = _lib.lib(); p = _lib.ptr; st = _lib.stream_ptr()
W = torch.randn(256, 256) * 0.05
planes = torch.from_numpy(split3_planes(W.numpy()).view(np.int16).copy()).to(DEV)
bb = torch.randn(256, device=DEV); g = torch.randn(256, device=DEV)


def t(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for M in (2048, 4096, 8192, 12288, 16000, 16384, 24576, 32768, 65536, 131072):
    A = torch.randn(M, 256, device=DEV); R = torch.randn(M, 256, device=DEV); Cm = torch.empty(M, 256, device=DEV)
    row = [f"M = {M:6d}"]
    for name, call in (("out-proj + residual + LayerNorm", lambda: lib.ctrlsim_gemm_nt_bf16x6(p(A), 256, p(planes), 256, 0, p(bb), p(R), 256, p(Cm), 256, M, 256, 256, 0, p(g), p(g), st)),
                       ("plain", lambda: lib.ctrlsim_gemm_nt_bf16x6(p(A), 256, p(planes), 256, 0, p(bb), None, 0, p(Cm), 256, M, 256, 256, 0, None, None, st))):
        for mask, tag in ((15, "ws"), (12, "tiled")):
            lib.ctrlsim_set_option(6, mask)
            row.append(f"{name} {tag}: {t(call):7.1f} us")
    lib.ctrlsim_set_option(6, 15)
    print(" | ".join(row), flush=True)
