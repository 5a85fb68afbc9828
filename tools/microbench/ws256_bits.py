"""Outputs of the weight-stationary 256 -> 256 Linear variants (plain / ReLU / residual / LayerNorm) at a few row counts, hashed — to compare two
builds bit for bit (CTRLSIM_LIB=...).  usage: python tools/microbench/ws256_bits.py"""
import hashlib, sys
sys.path.insert(0, '.')
import numpy as np
import torch
import ctrlsim_amd  # noqa: F401
from ctrlsim_amd import _lib
from ctrlsim_amd.pack import split3_planes
DEV = 'cuda:0'
lib = _lib.lib(); p = _lib.ptr; st = _lib.stream_ptr()
g = torch.Generator().manual_seed(3)
W = torch.randn(256, 256, generator=g) * 0.06
planes = torch.from_numpy(split3_planes(W.numpy()).view(np.int16).copy()).to(DEV)
bias, gam, bet = (torch.randn(256, generator=g).to(DEV) for _ in range(3))
h = hashlib.sha256()
for M in (1, 31, 32, 33, 8192, 8192 + 17, 20000, 16384 * 3 + 5, 300000):
    X = torch.randn(M, 256, generator=g).to(DEV); R = torch.randn(M, 256, generator=g).to(DEV)
    for relu, resid, ln in ((0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 1, 1), (1, 0, 1), (0, 0, 1), (1, 1, 1)):
        Y = torch.full((M, 320), 7.0, device=DEV)
        _lib.check(lib.ctrlsim_gemm_nt_bf16x6(p(X), 256, p(planes), 256, 0, p(bias), p(R) if resid else None, 256, p(Y), 320, M, 256, 256, relu,
                                              p(gam) if ln else None, p(bet) if ln else None, st))
        torch.cuda.synchronize()
        assert (Y[:, 256:] == 7.0).all()
        h.update(Y.cpu().numpy().tobytes())
        if resid:                                  # in place over the residual rows
            Z = R.clone()
            _lib.check(lib.ctrlsim_gemm_nt_bf16x6(p(X), 256, p(planes), 256, 0, p(bias), p(Z), 256, p(Z), 256, M, 256, 256, relu,
                                                  p(gam) if ln else None, p(bet) if ln else None, st))
            torch.cuda.synchronize()
            assert torch.equal(Z, Y[:, :256]), (M, relu, resid, ln)
print("sha256 of all outputs:", h.hexdigest())
