"""Per-segment cycle accounting of the row-stationary in_proj (library built with -DRS_TIMING): python tools/microbench/rs_timing.py [B] [L]"""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from ctrlsim_amd import _lib
from ctrlsim_amd.pack import row_blocks
lib = _lib.lib(); p = _lib.ptr; st = _lib.stream_ptr(); DEV = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256; L = int(sys.argv[2]) if len(sys.argv) > 2 else 2304
M, nkt = B * L, (L + 63) // 64
A = torch.randn(M, 256, device=DEV); W = torch.randn(768, 256) * 0.05; b = torch.randn(768, device=DEV)
blocks = torch.from_numpy(row_blocks(W.numpy(), 1).view(np.int16).copy()).to(DEV)
Cm = torch.empty(M, 768, device=DEV); img = torch.zeros(B * 8 * nkt * 8192, dtype=torch.int16, device=DEV)
f = lambda: lib.ctrlsim_gemm_kv_blocks(p(A), 256, p(blocks), p(b), p(Cm), 768, M, 768, p(img), L, nkt, 256, st)
raw = ctypes.CDLL(os.environ["CTRLSIM_LIB"]); raw.ctrlsim_debug_rs_times.argtypes = [ctypes.c_void_p, ctypes.c_int]
for _ in range(20): f()
torch.cuda.synchronize(); raw.ctrlsim_debug_rs_times(None, 1)
n = 50
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n): f()
e1.record(); torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 8)(); raw.ctrlsim_debug_rs_times(out, 0)
t = np.array(list(out), dtype=np.float64)
waves, phases = t[6], t[5]
names = ["x load + split", "k-loop", "wait + barrier", "epilogue", "acc init", "(phases)", "(waves)", "job prologue (placement)"]
tot = t[[0, 1, 2, 3, 4, 7]].sum()
print(f"M = {M}: {e0.elapsed_time(e1) / n:.3f} ms per launch; {phases / waves:.1f} phases per stamped wave per launch x {n}")
for i in (7, 0, 4, 1, 2, 3):
    print(f"  {names[i]:28s} {t[i] / tot * 100:5.1f} %   {t[i] / phases:8.1f} ticks per phase")
