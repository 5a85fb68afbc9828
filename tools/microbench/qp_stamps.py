"""Per-segment cycle accounting of the QP mode of the fused feed-forward kernel (ctrlsim_outproj_ln_q): wave 0 lane 0 of every workgroup adds
s_memtime deltas per row block.  The stamps are NOT in the shipped source: apply tools/microbench/patches/qp_stamp.patch to a scratch copy of
csrc/ffn_fused.hip first (git apply; git checkout afterwards), then
  CTRLSIM_VARIANT=qpS  CTRLSIM_EXTRA_DEFS="-DQP_STAMP"                  python ctrl-sim_amd/csrc/build.py     # as shipped + stamps
  CTRLSIM_VARIANT=qpS2 CTRLSIM_EXTRA_DEFS="-DQP_STAMP -DQP_SKIP_LD=1"   python ctrl-sim_amd/csrc/build.py     # without the row loads
  CTRLSIM_VARIANT=qpS1 CTRLSIM_EXTRA_DEFS="-DQP_STAMP -DQP_SKIP_ST=1"   python ctrl-sim_amd/csrc/build.py     # without the stores (q products become dead code)
usage: CTRLSIM_LIB=tools/microbench/variants/all_qpS.so python tools/microbench/qp_stamps.py   (profiles/r05_fusion_k256.md holds the round-5 numbers)"""
import ctypes, sys
sys.path.insert(0, '.')
import numpy as np
import torch
import ctrlsim_amd  # noqa: F401
from ctrlsim_amd import _lib
from ctrlsim_amd.pack import outproj_q_planes
DEV = 'cuda:0'
lib = _lib.lib(); p = _lib.ptr; st = _lib.stream_ptr()
raw = ctypes.CDLL(_lib.LIB_PATH)
g = torch.Generator().manual_seed(1)
Wo = torch.randn(256, 256, generator=g) * 0.07; Wq = torch.randn(256, 256, generator=g) * 0.09
bo, g0, be0, bq = (torch.randn(256, generator=g).to(DEV) * s for s in (0.3, 1.0, 0.2, 0.4))
dev = lambda a: torch.from_numpy(a.view(np.int16).copy()).to(DEV)
wod, wqd = (dev(a) for a in outproj_q_planes(Wo.numpy(), Wq.numpy(), 1))
M = 256 * 2304
O = torch.randn(M, 256, generator=g).to(DEV); R = torch.randn(M, 256, generator=g).to(DEV)
X1 = torch.empty_like(R); Q = torch.empty_like(R)
call = lambda: lib.ctrlsim_outproj_ln_q(p(O), 256, p(R), 256, p(wod), p(bo), p(g0), p(be0), p(wqd), p(bq), p(X1), 256, p(Q), 256, M, st)
for _ in range(20): call()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
raw.qp_stamps_read(buf)
n = 50
for _ in range(n): call()
torch.cuda.synchronize()
raw.qp_stamps_read(buf)
names = ["rows converted (O loads waited)", "barrier (DMA blocks 0,1)", "leading product (8 blocks)", "LayerNorm + fragments", "q products (8 blocks) + stores", "last q stores"]
rb = n * M / 128
tot = sum(buf[i] for i in range(6))
for i, nm in enumerate(names):
    print(f"{nm:40s} {buf[i] / rb:9.0f} ticks per row block  {100 * buf[i] / tot:5.1f} %")
print(f"total {tot / rb:.0f} ticks per row block (ticks of the constant-rate counter; divide by the measured kernel time per row block for their length)")
