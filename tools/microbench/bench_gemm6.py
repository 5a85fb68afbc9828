"""bf16x6 GEMM micro-benchmark at rollout shapes. Usage: python tools/microbench/bench_gemm6.py [B]"""
import sys
sys.path.insert(0,'.'); sys.path.insert(0,'oracle'); sys.path.insert(0,'tests')
import torch, numpy as np
import ctrlsim_amd
from ctrlsim_amd import _lib
from ctrlsim_amd.pack import split3_planes
DEV='cuda:0'
B=int(sys.argv[1]) if len(sys.argv)>1 else 64
lib=_lib.lib(); p=_lib.ptr; st=_lib.stream_ptr()
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n
L=2304; M=B*L
g=torch.randn(256,device=DEV)
for tile in ((0, 1, 2) if len(sys.argv) > 2 else (0,)):
  lib.ctrlsim_set_option(2,tile)
  print('tile option',tile)
  for (N,K,relu,res,ln,name) in [(768,256,0,0,0,'qkv'),(256,256,0,1,0,'out+res'),(256,256,0,1,1,'out+res+LN'),(1024,256,1,0,0,'ffn1'),(256,1024,0,1,0,'ffn2+res'),(256,1024,0,1,1,'ffn2+res+LN')]:
      A=torch.randn(M,K,device=DEV); W=torch.randn(N,K)*0.05; b=torch.randn(N,device=DEV); R=torch.randn(M,N,device=DEV) if res else None; Cm=torch.empty(M,N,device=DEV)
      planes=torch.from_numpy(split3_planes(W.numpy()).view(np.int16).copy()).to(DEV)
      f=lambda: lib.ctrlsim_gemm_nt_bf16x6(p(A),K,p(planes),N,0,p(b),p(R),N if res else 0,p(Cm),N,M,N,K,relu,p(g) if ln else None,p(g) if ln else None,st)
      ms=timeit(f); print(f'gemm6 {name:14s} M={M} N={N} K={K}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.1f} TF-eq')
