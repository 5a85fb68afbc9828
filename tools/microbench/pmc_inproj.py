"""Three launches each of the in_proj with K/V images at 589 824 rows through the weight-stationary kernel (gemm_ws256_kernel<..,KV>) and the
row-stationary one (inproj_rs_kernel): the target of the rocprofv3 --pmc passes summarised in profiles/r04_c_pmc_inproj.md."""
import sys
sys.path.insert(0, '.')
import torch, numpy as np
import ctrlsim_amd  # noqa
from ctrlsim_amd import _lib
from ctrlsim_amd.pack import split3_planes, row_blocks
DEV = 'cuda:0'; B, L = 256, 2304; M = B * L; nkt = 36
lib = _lib.lib(); p = _lib.ptr; st = _lib.stream_ptr()
A = torch.randn(M, 256, device=DEV); W = torch.randn(768, 256) * 0.05; b = torch.randn(768, device=DEV)
planes = torch.from_numpy(split3_planes(W.numpy()).view(np.int16).copy()).to(DEV)
blocks = torch.from_numpy(row_blocks(W.numpy(), 1).view(np.int16).copy()).to(DEV)
Cm = torch.empty(M, 768, device=DEV); img = torch.zeros(B * 8 * nkt * 8192, dtype=torch.int16, device=DEV)
for _ in range(3):
    lib.ctrlsim_gemm_nt_kv(p(A), 256, p(planes), 768, 0, p(b), p(Cm), 768, M, 768, 256, p(img), L, nkt, 256, st)
torch.cuda.synchronize()
for _ in range(3):
    lib.ctrlsim_gemm_kv_blocks(p(A), 256, p(blocks), p(b), p(Cm), 768, M, 768, p(img), L, nkt, 256, st)
torch.cuda.synchronize()
