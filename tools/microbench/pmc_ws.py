"""Three launches each of the plain and the residual + LayerNorm weight-stationary Linear at 589 824 rows: the target of the rocprofv3 --pmc
passes summarised in profiles/r03_c_pmc_ws_linear.md (counter sets as in tools/microbench/pmc.sh)."""
import sys
sys.path.insert(0,'.'); sys.path.insert(0,'oracle'); sys.path.insert(0,'tests')
import torch, numpy as np
import ctrlsim_amd
from ctrlsim_amd import _lib
from ctrlsim_amd.pack import split3_planes
DEV='cuda:0'; M=256*2304
lib=_lib.lib(); p=_lib.ptr; st=_lib.stream_ptr()
g=torch.randn(256,device=DEV)
for (relu,res,ln) in [(0,0,0),(0,1,1)]:
    N=K=256
    A=torch.randn(M,K,device=DEV); W=torch.randn(N,K)*0.05; b=torch.randn(N,device=DEV); R=torch.randn(M,N,device=DEV) if res else None; Cm=torch.empty(M,N,device=DEV)
    planes=torch.from_numpy(split3_planes(W.numpy()).view(np.int16).copy()).to(DEV)
    for _ in range(3): lib.ctrlsim_gemm_nt_bf16x6(p(A),K,p(planes),N,0,p(b),p(R),N if res else 0,p(Cm),N,M,N,K,relu,p(g) if ln else None,p(g) if ln else None,st)
    torch.cuda.synchronize()
