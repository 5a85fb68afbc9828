"""Known-byte launches of the four hot MFMA kernels for the FETCH_SIZE / WRITE_SIZE calibration of profiles/r05_pmc_traffic.md:
every launch streams M = 589 824 activation rows (604 MB per 1 KB of row bytes: past the 256 MiB Infinity Cache), weights are < 4 MB
(L2 / MALL-resident), so the bytes each kernel MUST move are known per operand.  Prints one JSON line with those byte counts; run it
under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes) and divide (tools/pmc_calib_summary.py)."""
import json
import sys
sys.path.insert(0, '.')
import numpy as np
import torch

import ctrlsim_amd  # noqa: F401
from ctrlsim_amd import _lib
from ctrlsim_amd.pack import split3_planes, row_blocks, ffn_planes

DEV = 'cuda:0'; B, L = 256, 2304; M = B * L; nkt = 36; REPS = 3
lib = _lib.lib(); p = _lib.ptr; st = _lib.stream_ptr()
NPL = 2 if lib.ctrlsim_split_scheme() == 1 else 3
KB = 1024.0 * M
g = torch.randn(256, device=DEV)
expect = {}

# 1. in_proj + K/V images, row-stationary: rows in once; fp32 query rows + K / V plane entries out
A = torch.randn(M, 256, device=DEV); W = torch.randn(768, 256) * 0.05; b = torch.randn(768, device=DEV)
blocks = torch.from_numpy(row_blocks(W.numpy(), 1).view(np.int16).copy()).to(DEV)
Cm = torch.empty(M, 768, device=DEV); img = torch.zeros(B * 8 * nkt * 4096 * NPL, dtype=torch.int16, device=DEV)
for _ in range(REPS):
    lib.ctrlsim_gemm_kv_blocks(p(A), 256, p(blocks), p(b), p(Cm), 768, M, 768, p(img), L, nkt, 256, st)
torch.cuda.synchronize()
expect["inproj_rs_kernel"] = {"read": KB, "write": KB + 2 * NPL * 256 * 2.0 * M, "note": "A rows; Q rows fp32 + K/V plane entries"}
del Cm, img

# 2. weight-stationary Linear 256 -> 256: plain, and with residual + LayerNorm
planes = torch.from_numpy(split3_planes((torch.randn(256, 256) * 0.05).numpy()).view(np.int16).copy()).to(DEV)
bq = torch.randn(256, device=DEV); R = torch.randn(M, 256, device=DEV); C2 = torch.empty(M, 256, device=DEV)
for _ in range(REPS):
    lib.ctrlsim_gemm_nt_bf16x6(p(A), 256, p(planes), 256, 0, p(bq), None, 0, p(C2), 256, M, 256, 256, 0, None, None, st)
torch.cuda.synchronize()
for _ in range(REPS):
    lib.ctrlsim_gemm_nt_bf16x6(p(A), 256, p(planes), 256, 0, p(bq), p(R), 256, p(C2), 256, M, 256, 256, 0, p(g), p(g), st)
torch.cuda.synchronize()
expect["gemm_ws256_kernel plain"] = {"read": KB, "write": KB, "match": "gemm_ws256_kernelILb0ELb0ELb0E"}
expect["gemm_ws256_kernel res+LN"] = {"read": 2 * KB, "write": KB, "match": "gemm_ws256_kernelILb0ELb1ELb1E"}

# 3. fused feed-forward block: x rows as operand (+ as residual: second read, cache hit or not), y rows out
F = 1024
W1 = torch.randn(F, 256) * 0.05; W2 = torch.randn(256, F) * 0.05
w1p, w2p = ffn_planes(W1.numpy(), W2.numpy())
w1d = torch.from_numpy(w1p.view(np.int16).copy()).to(DEV); w2d = torch.from_numpy(w2p.view(np.int16).copy()).to(DEV)
b1 = torch.randn(F, device=DEV)
for _ in range(REPS):
    lib.ctrlsim_ffn_fused(p(A), 256, p(w1d), p(b1), p(w2d), p(bq), p(g), p(g), p(C2), 256, M, F, st)
torch.cuda.synchronize()
expect["ffn_fused_bf16x6_kernel"] = {"read": KB, "read_max": 2 * KB, "write": KB, "note": "x read as operand and again as residual"}
del R

# 4. causal attention, staged form with the mask table (B contexts of 2304 rows): Q rows in, O rows out, K/V images read by 18 query blocks
qkv = torch.randn(B, L, 768, device=DEV); O = C2.view(B, L, 256)
img = torch.zeros(B * 8 * nkt * 4096 * NPL, dtype=torch.int16, device=DEV)
lib.ctrlsim_kv_split(qkv.data_ptr() + 1024, qkv.data_ptr() + 2048, 768, L * 768, None, B, L, nkt, p(img), st)
tbl = torch.zeros(lib.ctrlsim_attention_mask_table_bytes(L, nkt) // 8, dtype=torch.int64, device=DEV)
lib.ctrlsim_attention_mask_table(L, L, 24, 0, L, nkt, p(tbl), st)
torch.cuda.synchronize()
for _ in range(REPS):
    lib.ctrlsim_attention_tbl(p(qkv), 768, L * 768, p(img), nkt, p(O), 256, L * 256, B, L, L, 24, 0, 1, p(tbl), st)
torch.cuda.synchronize()
img_bytes = float(img.numel() * 2)
expect["attention staged causal"] = {"read": KB + img_bytes, "read_max": KB + img_bytes * 9.5, "write": KB, "match": "attention_bf16x6_kernel<1, true, true",
                                     "note": "Q rows (1 KB of 3 KB-strided rows) + each K/V image once; query blocks re-read images (L2 / MALL)"}
# 5. few-query streaming form: 24 queries per context at the end of the window, all keys
Lq = 24; Bf = 1024
qf = torch.randn(Bf, Lq, 768, device=DEV); Of = torch.empty(Bf, Lq, 256, device=DEV)
imgf = torch.randn(Bf * 8 * nkt * 4096 * NPL // 2, device=DEV).view(torch.int16)
qpos = torch.arange(L - Lq, L, dtype=torch.int32, device=DEV)
for _ in range(REPS):
    lib.ctrlsim_attention_presplit(1, p(qf), 768, Lq * 768, p(imgf), nkt, p(Of), 256, Lq * 256, p(qpos), None, Bf, Lq, L, 24, st)
torch.cuda.synchronize()
expect["attention streaming few-query"] = {"read": float(imgf.numel() * 2) + Bf * Lq * 1024.0, "write": Bf * Lq * 1024.0, "match": "attention_bf16x6_kernel<1, true, false",
                                           "note": "every K/V image once (one wave per context, head, 32 queries)"}
print(json.dumps({"M": M, "reps": REPS, "expect": expect}))
