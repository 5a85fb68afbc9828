"""Sustained-clock micro-benchmark of the hot kernels at the ROLLOUT batch (B contexts x 2304 tokens), through the C ABI.

Each kernel is run back to back for `secs` seconds after a warm-up of the same length, so the part sits at the clock it
sustains in the rollout (burst timings of a few launches ran 5-15 % fast in round 1 and rewarded the wrong changes).
Usage: python tools/microbench/sustained.py [B=384] [secs=1.5] [filter]
Prints one line per kernel: ms per launch, fp32-equivalent TFLOP/s, fraction of the split-operand roof (2500 / NPROD)."""
import os
import sys
import time

sys.path.insert(0, '.'); sys.path.insert(0, 'oracle'); sys.path.insert(0, 'tests')
import numpy as np
import torch

import ctrlsim_amd  # noqa: F401
from ctrlsim_amd import _lib
from ctrlsim_amd.pack import split3_planes, ffn_planes

DEV = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 384
SECS = float(sys.argv[2]) if len(sys.argv) > 2 else 1.5
FILT = sys.argv[3] if len(sys.argv) > 3 else ''
OPTIONS = [kv.split('=') for kv in filter(None, os.environ.get('CTRLSIM_OPTIONS', '').split(','))]   # '<option>=<value>,...'
lib = _lib.lib(); p = _lib.ptr; st = _lib.stream_ptr()
for _k, _v in OPTIONS:
    lib.ctrlsim_set_option(int(_k), int(_v))
NPROD = 3 if lib.ctrlsim_split_scheme() == 1 else 6
ROOF = 2500.0 / NPROD


def sustained(fn):
    def burst(secs):
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
    burst(SECS)
    return burst(SECS)


def report(name, ms, flops):
    tf = flops / ms / 1e9
    print(f'{name:34s} {ms:8.3f} ms  {tf:7.1f} TF-eq  {tf / ROOF:.3f} of roof', flush=True)


L = 2304; M = B * L; A3 = 72; T = 32; Aa = 24
g = torch.randn(256, device=DEV)
if 'attn' in FILT or not FILT:
    nkt = 36
    qkv = torch.randn(B, L, 768, device=DEV)
    O = torch.empty(B, L, 256, device=DEV)
    pairs = A3 * A3 * T * (T - 1) / 2 + T * Aa * (3 * Aa + 3)
    img = torch.zeros(B * 8 * nkt * 4096 * (2 if NPROD == 3 else 3), dtype=torch.int16, device=DEV)
    lib.ctrlsim_kv_split(qkv.data_ptr() + 1024, qkv.data_ptr() + 2048, 768, L * 768, None, B, L, nkt, p(img), st)
    f = lambda: lib.ctrlsim_attention_presplit(1, p(qkv), 768, L * 768, p(img), nkt, p(O), 256, L * 256, None, None, B, L, L, 24, st)
    report(f'attn causal presplit L={L}', sustained(f), pairs * 128 * 8 * B)
    tbl = torch.zeros(lib.ctrlsim_attention_mask_table_bytes(L, nkt) // 8, dtype=torch.int64, device=DEV)
    lib.ctrlsim_attention_mask_table(L, L, 24, 0, L, nkt, p(tbl), st)
    f = lambda: lib.ctrlsim_attention_tbl(p(qkv), 768, L * 768, p(img), nkt, p(O), 256, L * 256, B, L, L, 24, 0, 1, p(tbl), st)
    report(f'attn causal mask-table L={L}', sustained(f), pairs * 128 * 8 * B)
    Q = torch.randn(B, L, 256, device=DEV); KV = torch.randn(B, 224, 512, device=DEV)
    pad = torch.zeros(B, 224, dtype=torch.uint8, device=DEV)
    img2 = torch.zeros(B * 8 * 4 * 4096 * (2 if NPROD == 3 else 3), dtype=torch.int16, device=DEV)
    lib.ctrlsim_kv_split(p(KV), KV.data_ptr() + 1024, 512, 224 * 512, None, B, 224, 4, p(img2), st)
    f = lambda: lib.ctrlsim_attention_presplit(0, p(Q), 256, L * 256, p(img2), 4, p(O), 256, L * 256, None, p(pad), B, L, 224, 24, st)
    report('attn cross presplit Lk=224', sustained(f), L * 224 * 128 * 8 * B)
    del qkv, O, img, Q, KV, img2
if 'few' in FILT:
    # few-query causal calls (second pass / last layer / K/V-cached steps): Bf contexts x 8 heads, one query block each
    T = 32
    for (Bf, Lq_) in ((77, 24), (77, 96), (512, 24)):
        nkt = 36
        qkv = torch.randn(Bf, Lq_, 768, device=DEV); O = torch.empty(Bf, Lq_, 256, device=DEV)
        img = torch.randn(Bf * 8 * nkt * 4096 * (2 if NPROD == 3 else 3), device=DEV).to(torch.float16).view(torch.int16)
        qpos = torch.arange(L - Lq_, L, dtype=torch.int32, device=DEV)
        f = lambda: lib.ctrlsim_attention_presplit(1, p(qkv), 768, Lq_ * 768, p(img), nkt, p(O), 256, Lq_ * 256, p(qpos), None, Bf, Lq_, L, 24, st)
        report(f'attn few-query B={Bf} Lq={Lq_} keys={L}', sustained(f), Lq_ * L * 128 * 8 * Bf)
        del qkv, O, img
if 'compact' in FILT:
    # compact contexts (representative slot): Actx slots -> Areg = Actx - 1 regular + 1 representative of multiplicity 25 - Actx
    T = 32
    for Actx in (int(a) for a in os.environ.get('SUSTAINED_CLASSES', '4,8,12,16,20').split(',')):
        Ar = Actx - 1
        Lreg = T * 3 * Ar; Lq = Lreg + 3 * T
        nkt = (Lreg + 63) // 64 + 2
        Bc = max(64, int(B * 2304 / Lq))
        qkv = torch.randn(Bc, Lq, 768, device=DEV); O = torch.empty(Bc, Lq, 256, device=DEV)
        img = torch.randn(Bc * 8 * nkt * 4096 * (2 if NPROD == 3 else 3), device=DEV).to(torch.float16).view(torch.int16)
        A3 = 3 * Ar
        pairs = A3 * A3 * T * (T - 1) / 2 + T * Ar * (3 * Ar + 3)
        pairs += A3 * (3 * T * (T - 1) / 2 + T) + 3 * A3 * T * (T - 1) / 2 + 3 * Ar * T + 9 * T * (T - 1) / 2 + 6 * T
        f = lambda: lib.ctrlsim_attention_compact(p(qkv), 768, Lq * 768, p(img), nkt, p(O), 256, Lq * 256, None, Bc, Lq, Lreg, Ar,
                                                  3 * T, 25 - Actx, Lreg, st)
        report(f'attn compact Actx={Actx} L={Lq} B={Bc}', sustained(f), pairs * 128 * 8 * Bc)
        tbl = torch.zeros(lib.ctrlsim_attention_mask_table_bytes(Lq, nkt) // 8, dtype=torch.int64, device=DEV)
        lib.ctrlsim_attention_mask_table(Lq, Lreg, Ar, 3 * T, Lreg, nkt, p(tbl), st)
        f = lambda: lib.ctrlsim_attention_tbl(p(qkv), 768, Lq * 768, p(img), nkt, p(O), 256, Lq * 256, Bc, Lq, Lreg, Ar, 3 * T, 25 - Actx,
                                              p(tbl), st)
        report(f'attn mask-table Actx={Actx} L={Lq} B={Bc}', sustained(f), pairs * 128 * 8 * Bc)
        del qkv, O, img
if 'gemm' in FILT or not FILT:
    for (N, K, relu, res, ln, name) in [(768, 256, 0, 0, 0, 'qkv (fp32 out)'), (256, 256, 0, 0, 0, 'cross-q'), (256, 256, 0, 1, 1, 'out+res+LN')]:
        A = torch.randn(M, K, device=DEV); W = torch.randn(N, K) * 0.05; b = torch.randn(N, device=DEV)
        R = torch.randn(M, N, device=DEV) if res else None; Cm = torch.empty(M, N, device=DEV)
        planes = torch.from_numpy(split3_planes(W.numpy()).view(np.int16).copy()).to(DEV)
        f = lambda: lib.ctrlsim_gemm_nt_bf16x6(p(A), K, p(planes), N, 0, p(b), p(R), N if res else 0, p(Cm), N, M, N, K, relu,
                                               p(g) if ln else None, p(g) if ln else None, st)
        report(f'gemm {name} N={N} K={K}', sustained(f), 2.0 * M * N * K)
        del A, R, Cm
if 'gemm' in FILT or not FILT:
    nkt = 36
    A = torch.randn(M, 256, device=DEV); W = torch.randn(768, 256) * 0.05; b = torch.randn(768, device=DEV)
    planes = torch.from_numpy(split3_planes(W.numpy()).view(np.int16).copy()).to(DEV)
    Cm = torch.empty(M, 768, device=DEV)
    img = torch.zeros(B * 8 * nkt * 4096 * (2 if NPROD == 3 else 3), dtype=torch.int16, device=DEV)
    f = lambda: lib.ctrlsim_gemm_nt_kv(p(A), 256, p(planes), 768, 0, p(b), p(Cm), 768, M, 768, 256, p(img), L, nkt, 256, st)
    report('gemm qkv + K/V images N=768 K=256', sustained(f), 2.0 * M * 768 * 256)
    if NPROD == 3:
        from ctrlsim_amd.pack import row_blocks
        blocks = torch.from_numpy(row_blocks(W.numpy(), 1).view(np.int16).copy()).to(DEV)
        f = lambda: lib.ctrlsim_gemm_kv_blocks(p(A), 256, p(blocks), p(b), p(Cm), 768, M, 768, p(img), L, nkt, 256, st)
        report('gemm qkv + K/V images, row-stationary', sustained(f), 2.0 * M * 768 * 256)
        Mq = M // 4
        f = lambda: lib.ctrlsim_gemm_nt_kv(p(A), 256, p(planes), 768, 0, p(b), p(Cm), 768, Mq, 768, 256, p(img), L, nkt, 256, st)
        report('gemm qkv + K/V images, quarter rows', sustained(f), 2.0 * Mq * 768 * 256)
        f = lambda: lib.ctrlsim_gemm_kv_blocks(p(A), 256, p(blocks), p(b), p(Cm), 768, Mq, 768, p(img), L, nkt, 256, st)
        report('gemm qkv + K/V images, row-stationary, quarter rows', sustained(f), 2.0 * Mq * 768 * 256)
    if NPROD == 3:
        Wq = torch.randn(256, 256) * 0.05; bq = torch.randn(256, device=DEV)
        blq = torch.from_numpy(row_blocks(Wq.numpy(), 1).view(np.int16).copy()).to(DEV)
        Cq = torch.empty(M, 256, device=DEV)
        f = lambda: lib.ctrlsim_gemm_kv_blocks(p(A), 256, p(blq), p(bq), p(Cq), 256, M, 256, None, 0, 0, 0, st)
        report('gemm cross-q, row-stationary', sustained(f), 2.0 * M * 256 * 256)
        del Cq
    del A, Cm, img
if 'ffn' in FILT or not FILT:
    F = 1024
    X = torch.randn(M, 256, device=DEV); W1 = torch.randn(F, 256) * 0.05; W2 = torch.randn(256, F) * 0.05
    b1 = torch.randn(F, device=DEV); b2 = torch.randn(256, device=DEV)
    w1p, w2p = ffn_planes(W1.numpy(), W2.numpy())
    w1d = torch.from_numpy(w1p.view(np.int16).copy()).to(DEV); w2d = torch.from_numpy(w2p.view(np.int16).copy()).to(DEV)
    Y = torch.empty_like(X)
    f = lambda: lib.ctrlsim_ffn_fused(p(X), 256, p(w1d), p(b1), p(w2d), p(b2), p(g), p(g), p(Y), 256, M, F, st)
    report('ffn fused', sustained(f), 4.0 * M * 256 * F)
