import sys
sys.path.insert(0,'.'); sys.path.insert(0,'oracle'); sys.path.insert(0,'tests')
import torch, numpy as np
import ctrlsim_amd
from ctrlsim_amd import _lib
from ctrlsim_amd.pack import split3_planes
DEV='cuda:0'; B=64; L=2304; M=B*L
lib=_lib.lib(); p=_lib.ptr; st=_lib.stream_ptr()
for (N,K,relu,res) in [(768,256,0,0),(1024,256,1,0),(256,1024,0,1)]:
    A=torch.randn(M,K,device=DEV); W=torch.randn(N,K)*0.05; b=torch.randn(N,device=DEV); R=torch.randn(M,N,device=DEV) if res else None; Cm=torch.empty(M,N,device=DEV)
    planes=torch.from_numpy(split3_planes(W.numpy()).view(np.int16).copy()).to(DEV)
    for _ in range(2): lib.ctrlsim_gemm_nt_bf16x6(p(A),K,p(planes),N,0,p(b),p(R),N if res else 0,p(Cm),N,M,N,K,relu,None,None,st)
    torch.cuda.synchronize()
qkv=torch.randn(B,L,768,device=DEV); O=torch.empty(B,L,256,device=DEV)
for _ in range(2): lib.ctrlsim_attention(1,p(qkv),768,L*768,qkv.data_ptr()+1024,qkv.data_ptr()+2048,768,L*768,p(O),256,L*256,None,None,B,L,L,24,st)
torch.cuda.synchronize()

nkt=36
img=torch.zeros(B*8*nkt*12288,dtype=torch.int16,device=DEV)
lib.ctrlsim_kv_split(qkv.data_ptr()+1024,qkv.data_ptr()+2048,768,L*768,None,B,L,nkt,p(img),st)
for _ in range(2): lib.ctrlsim_attention_presplit(1,p(qkv),768,L*768,p(img),nkt,p(O),256,L*256,None,None,B,L,L,24,st)
torch.cuda.synchronize()

# fused feed-forward block
from ctrlsim_amd.pack import ffn_planes
F=1024
X=torch.randn(M,256,device=DEV); W1=torch.randn(F,256)*0.05; W2=torch.randn(256,F)*0.05
b1=torch.randn(F,device=DEV); b2=torch.randn(256,device=DEV); g=torch.randn(256,device=DEV)
w1p,w2p=ffn_planes(W1.numpy(),W2.numpy())
w1d=torch.from_numpy(w1p.view(np.int16).copy()).to(DEV); w2d=torch.from_numpy(w2p.view(np.int16).copy()).to(DEV)
Y=torch.empty_like(X)
for _ in range(2): lib.ctrlsim_ffn_fused(p(X),256,p(w1d),p(b1),p(w2d),p(b2),p(g),p(g),p(Y),256,M,F,st)
torch.cuda.synchronize()
