import sys
sys.path.insert(0,'.'); sys.path.insert(0,'oracle'); sys.path.insert(0,'tests')
import torch, numpy as np
import ctrlsim_amd
from ctrlsim_amd import _lib
from ctrlsim_amd.pack import ffn_planes, split3_planes
DEV='cuda:0'
B=int(sys.argv[1]) if len(sys.argv)>1 else 64
lib=_lib.lib(); p=_lib.ptr; st=_lib.stream_ptr()
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n
M=B*2304; F=1024
X=torch.randn(M,256,device=DEV); W1=torch.randn(F,256)*0.05; W2=torch.randn(256,F)*0.05
b1=torch.randn(F,device=DEV); b2=torch.randn(256,device=DEV); g=torch.randn(256,device=DEV)
w1p,w2p=ffn_planes(W1.numpy(),W2.numpy())
w1d=torch.from_numpy(w1p.view(np.int16).copy()).to(DEV); w2d=torch.from_numpy(w2p.view(np.int16).copy()).to(DEV)
Y=torch.empty_like(X)
f=lambda: lib.ctrlsim_ffn_fused(p(X),256,p(w1d),p(b1),p(w2d),p(b2),p(g),p(g),p(Y),256,M,F,st)
ms=timeit(f); print(f'ffn fused M={M}: {ms:.3f} ms  {4*M*256*F/ms/1e9:.1f} TF-eq')
H=torch.empty(M,F,device=DEV)
pl1=torch.from_numpy(split3_planes(W1.numpy()).view(np.int16).copy()).to(DEV); pl2=torch.from_numpy(split3_planes(W2.numpy()).view(np.int16).copy()).to(DEV)
def two():
    lib.ctrlsim_gemm_nt_bf16x6(p(X),256,p(pl1),F,0,p(b1),None,0,p(H),F,M,F,256,1,None,None,st)
    lib.ctrlsim_gemm_nt_bf16x6(p(H),F,p(pl2),256,0,p(b2),p(X),256,p(Y),256,M,256,F,0,p(g),p(g),st)
ms=timeit(two); print(f'ffn two kernels M={M}: {ms:.3f} ms  {4*M*256*F/ms/1e9:.1f} TF-eq')
