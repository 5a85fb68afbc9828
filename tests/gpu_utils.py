import ctypes as C

import numpy as np
import torch

from ctrlsim_amd import _lib

DEV = "cuda:0"


def build_pollute_lib():
    """tests/pollute/pollute.hip -> tests/pollute/libpollute.so (hipcc cross-compiles without a GPU; __graft_entry__.build() calls
    this so that the library travels to the GPU box prebuilt)."""
    import os
    import subprocess
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pollute")
    src, so = os.path.join(here, "pollute.hip"), os.path.join(here, "libpollute.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, src])
    return so


def delay_simulator_steps(eng, delay_us):
    """Every simulator step of `eng` first waits delay_us on its stream (tests/pollute's spin kernel): on a two-lane engine the
    step then runs underneath LATER kernels of the other lane's policy step than it naturally would."""
    lib = C.CDLL(build_pollute_lib())
    sink = torch.zeros(4, dtype=torch.int32, device=DEV)
    inner = eng.sim_step

    def delayed(t, act_f64=None, s0=0, s1=None, stream=None):
        st = _lib.stream_ptr() if stream is None else stream
        assert lib.spin_launch(int(delay_us), C.c_void_p(sink.data_ptr()), C.c_void_p(st)) == 0
        return inner(t, act_f64=act_f64, s0=s0, s1=s1, stream=stream)
    eng.sim_step = delayed
    eng._delay_sink = sink


class Polluter:
    """Context manager: launches pollute_kernel in a loop on its own stream from a host thread while the body runs (fresh waves of
    the kernels under test then start on register files and LDS full of NaN patterns)."""

    def __init__(self, blocks=1024):
        self.lib = C.CDLL(build_pollute_lib())
        self.sink = torch.zeros(4, dtype=torch.int32, device=DEV)
        self.stream = torch.cuda.Stream()
        self.blocks, self.launches = blocks, 0

    def _loop(self):
        k = 0
        while not self._stop.is_set():
            rc = self.lib.pollute_launch(C.c_void_p(self.sink.data_ptr()), k, self.blocks, C.c_void_p(self.stream.cuda_stream))
            assert rc == 0, rc
            k += 1
            if k % 8 == 0:
                self.stream.synchronize()
        self.launches = k

    def __enter__(self):
        import threading
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._loop)
        self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._th.join()
        self.stream.synchronize()


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
