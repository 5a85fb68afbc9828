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
