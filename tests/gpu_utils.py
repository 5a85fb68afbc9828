import ctypes as C

import numpy as np
import torch

from ctrlsim_amd import _lib

DEV = "cuda:0"


def dev(a, dt=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dt is not None:
        t = t.to(dt)
    return t.to(DEV)


def gemm(A, W, bias=None, R=None, relu=False):
    M, K = A.shape
    N = W.shape[0]
    Cm = torch.empty(M, N, device=DEV)
    p = _lib.ptr
    _lib.check(_lib.lib().ctrlsim_gemm_nt(p(A), A.stride(0), p(W), W.stride(0), p(bias), p(R), R.stride(0) if R is not None else 0,
                                          p(Cm), N, M, N, K, int(relu), _lib.stream_ptr()), "gemm")
    return Cm


def gemm_bf16x6(A, W, bias=None, R=None, relu=False, n0=0, n=None, ln=None, out=None):
    """W: full [Ntot,K] float32 tensor (CPU or GPU); rows [n0, n0+n) are multiplied."""
    from ctrlsim_amd.pack import split3_planes
    Wc = W.detach().cpu().numpy()
    planes = torch.from_numpy(split3_planes(Wc).view(np.int16).copy()).to(DEV)
    M, K = A.shape
    n = Wc.shape[0] - n0 if n is None else n
    Cm = torch.empty(M, n, device=DEV) if out is None else out      # out may alias R (fused-LN in-place update)
    p = _lib.ptr
    g, be = ln if ln is not None else (None, None)
    _lib.check(_lib.lib().ctrlsim_gemm_nt_bf16x6(p(A), A.stride(0), p(planes), Wc.shape[0], n0, p(bias), p(R),
                                                 R.stride(0) if R is not None else 0, p(Cm), Cm.stride(0), M, n, K, int(relu),
                                                 p(g), p(be), _lib.stream_ptr()), "gemm_bf16x6")
    return Cm
