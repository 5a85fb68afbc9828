"""Shader clock and socket power under EACH hot kernel (round 6: the part is power-bound under these kernels — profiles/r06_power_clocks.md).
Every kernel is run back to back for `secs` seconds at the rollout shape (B contexts x 2304 token rows, random data) while a host thread
samples `rocm-smi --showclocks --showpower` once a second; the first two seconds of samples are dropped (the clock settles).
Usage: python tools/microbench/clock_by_kernel.py [B=256] [secs=6]
Prints per kernel: ms per launch, fp32-equivalent TFLOP/s (or GB/s), median sclk MHz, median socket W."""
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, '.'); sys.path.insert(0, 'oracle'); sys.path.insert(0, 'tests')
import numpy as np
import torch

import ctrlsim_amd  # noqa: F401
from ctrlsim_amd import _lib, spec, weights
from ctrlsim_amd.pack import split3_planes, ffn_planes, row_blocks

DEV = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SECS = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
lib = _lib.lib(); p = _lib.ptr; st = _lib.stream_ptr()
L = 2304; M = B * L


class Sampler:
    def __init__(self):
        self.samples, self._stop = [], threading.Event()
        self.th = threading.Thread(target=self.run, daemon=True); self.th.start()

    def run(self):
        while not self._stop.wait(1.0):
            try:
                out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            except (OSError, subprocess.SubprocessError):
                continue
            m = re.search(r"sclk clock level:[^(]*\((\d+)Mhz\)", out); w = re.search(r"Power \(W\):\s*([\d.]+)", out)
            if m and w:
                self.samples.append((time.perf_counter(), int(m.group(1)), float(w.group(1))))

    def stop(self):
        self._stop.set(); self.th.join(timeout=15)
        return self.samples


def measure(name, fn, work, unit):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    smp = Sampler()
    t0 = time.perf_counter(); n = 0
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    while time.perf_counter() - t0 < SECS:
        for _ in range(8):
            fn()
        n += 8
        torch.cuda.synchronize()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    s = [x for x in smp.stop() if x[0] - t0 > 2.0]
    clk = sorted(x[1] for x in s); pw = sorted(x[2] for x in s)
    med = lambda v: v[len(v) // 2] if v else float('nan')
    print(f"{name:44s} {ms:8.3f} ms  {work / ms / 1e9:8.1f} {unit}  sclk {med(clk):5} MHz  {med(pw):6} W  ({len(s)} samples)", flush=True)


g = torch.randn(256, device=DEV)
# fused feed-forward block
F = 1024
X = torch.randn(M, 256, device=DEV); W1 = torch.randn(F, 256) * 0.05; W2 = torch.randn(256, F) * 0.05
b1 = torch.randn(F, device=DEV); b2 = torch.randn(256, device=DEV)
w1p, w2p = ffn_planes(W1.numpy(), W2.numpy())
w1d = torch.from_numpy(w1p.view(np.int16).copy()).to(DEV); w2d = torch.from_numpy(w2p.view(np.int16).copy()).to(DEV)
Y = torch.empty_like(X)
measure('ffn_fused (MODE 0)', lambda: lib.ctrlsim_ffn_fused(p(X), 256, p(w1d), p(b1), p(w2d), p(b2), p(g), p(g), p(Y), 256, M, F, st), 4.0 * M * 256 * F, 'TF-eq')
# Linears
A = X; W = torch.randn(768, 256) * 0.05; bb = torch.randn(768, device=DEV)
planes = torch.from_numpy(split3_planes(W.numpy()).view(np.int16).copy()).to(DEV)
R = torch.randn(M, 256, device=DEV); Cm = torch.empty(M, 256, device=DEV)
measure('gemm_ws256 out-proj + residual + LayerNorm', lambda: lib.ctrlsim_gemm_nt_bf16x6(p(A), 256, p(planes), 768, 0, p(bb), p(R), 256, p(Cm), 256, M, 256, 256, 0, p(g), p(g), st),
        2.0 * M * 256 * 256, 'TF-eq')
measure('gemm_ws256 plain 256 -> 256', lambda: lib.ctrlsim_gemm_nt_bf16x6(p(A), 256, p(planes), 768, 0, p(bb), None, 0, p(Cm), 256, M, 256, 256, 0, None, None, st),
        2.0 * M * 256 * 256, 'TF-eq')
nkt = 36
blocks = torch.from_numpy(row_blocks(W.numpy(), 1).view(np.int16).copy()).to(DEV)
C3 = torch.empty(M, 768, device=DEV)
img = torch.zeros(B * 8 * nkt * 4096 * 2, dtype=torch.int16, device=DEV)
measure('inproj_rs (QKV + K/V images)', lambda: lib.ctrlsim_gemm_kv_blocks(p(A), 256, p(blocks), p(bb), p(C3), 768, M, 768, p(img), L, nkt, 256, st), 2.0 * M * 768 * 256, 'TF-eq')
del C3, R, Cm
# attention
qkv = torch.randn(B, L, 768, device=DEV); O = torch.empty(B, L, 256, device=DEV)
lib.ctrlsim_kv_split(qkv.data_ptr() + 1024, qkv.data_ptr() + 2048, 768, L * 768, None, B, L, nkt, p(img), st)
tbl = torch.zeros(lib.ctrlsim_attention_mask_table_bytes(L, nkt) // 8, dtype=torch.int64, device=DEV)
lib.ctrlsim_attention_mask_table(L, L, 24, 0, L, nkt, p(tbl), st)
pairs = 72 * 72 * 32 * 31 / 2 + 32 * 24 * (3 * 24 + 3)
measure('attention causal (mask table) L = 2304', lambda: lib.ctrlsim_attention_tbl(p(qkv), 768, L * 768, p(img), nkt, p(O), 256, L * 256, B, L, L, 24, 0, 1, p(tbl), st),
        pairs * 128 * 8 * B, 'TF-eq')
Q = torch.randn(B, L, 256, device=DEV); KV = torch.randn(B, 224, 512, device=DEV)
pad = torch.zeros(B, 224, dtype=torch.uint8, device=DEV)
img2 = torch.zeros(B * 8 * 4 * 4096 * 2, dtype=torch.int16, device=DEV)
lib.ctrlsim_kv_split(p(KV), KV.data_ptr() + 1024, 512, 224 * 512, None, B, 224, 4, p(img2), st)
measure('attention cross (resident keys) Lk = 224', lambda: lib.ctrlsim_attention_presplit(0, p(Q), 256, L * 256, p(img2), 4, p(O), 256, L * 256, None, p(pad), B, L, 224, 24, st),
        L * 224 * 128 * 8 * B, 'TF-eq')
del qkv, O, Q, KV, img, img2, X, Y
# map encoder
from ctrlsim_amd.engine import HipModel, CtxBuffers
cfg = spec.make_cfg(); d = spec.Dims(cfg)
model = HipModel(cfg, weights.generate(d, 0), DEV)
Bm = 4 * B
rs = np.random.RandomState(0)
npts = rs.randint(20, d.NP + 1, (Bm, d.P))
ex = (np.arange(d.NP)[None, None] < npts[..., None]).astype(np.float32)
rp = np.concatenate([rs.randn(Bm, d.P, d.NP, 2).astype(np.float32) * 20 * ex[..., None], ex[..., None]], -1)
cb = CtxBuffers(d, Bm, DEV)
cb.road_pts.copy_(torch.from_numpy(rp).to(DEV))
out = torch.empty(Bm * d.P, d.D, device=DEV); padm = torch.empty(Bm, d.P, dtype=torch.uint8, device=DEV)
measure(f'map_pool ({Bm} contexts, ragged)', lambda: lib.ctrlsim_map_pool(model.handle, Bm, p(cb.road_pts), p(out), p(padm), st), Bm * d.P * 1.5e6, 'TF (tagged)')
