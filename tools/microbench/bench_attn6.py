import sys
sys.path.insert(0,'.'); sys.path.insert(0,'oracle'); sys.path.insert(0,'tests')
import torch, numpy as np
import ctrlsim_amd
from ctrlsim_amd import _lib
DEV='cuda:0'
B=int(sys.argv[1]) if len(sys.argv)>1 else 64
lib=_lib.lib(); p=_lib.ptr; st=_lib.stream_ptr()
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n
L=2304; nkt=36
qkv=torch.randn(B,L,768,device=DEV); O=torch.empty(B,L,256,device=DEV)
A3=72;T=32;Aa=24; pairs=A3*A3*T*(T-1)/2+T*Aa*(3*Aa+3)
f=lambda: lib.ctrlsim_attention(1,p(qkv),768,L*768,qkv.data_ptr()+1024,qkv.data_ptr()+2048,768,L*768,p(O),256,L*256,None,None,B,L,L,24,st)
ms=timeit(f); print(f'attn causal L={L}: {ms:.3f} ms  {pairs*128*8*B/ms/1e9:.1f} TF-eq (visible pairs)')
img=torch.zeros(B*8*nkt*12288,dtype=torch.int16,device=DEV)
f=lambda: lib.ctrlsim_kv_split(qkv.data_ptr()+1024,qkv.data_ptr()+2048,768,L*768,None,B,L,nkt,p(img),st)
ms0=timeit(f); print(f'attn kv_split L={L}: {ms0:.3f} ms  {B*L*5120/ms0/1e9:.2f} TB/s')
f=lambda: lib.ctrlsim_attention_presplit(1,p(qkv),768,L*768,p(img),nkt,p(O),256,L*256,None,None,B,L,L,24,st)
ms=timeit(f); print(f'attn causal presplit L={L}: {ms:.3f} ms  {pairs*128*8*B/ms/1e9:.1f} TF-eq; with split {pairs*128*8*B/(ms+ms0)/1e9:.1f}')
Q=torch.randn(B,L,256,device=DEV); KV=torch.randn(B,224,512,device=DEV); pad=torch.zeros(B,224,dtype=torch.uint8,device=DEV)
f=lambda: lib.ctrlsim_attention(0,p(Q),256,L*256,p(KV),KV.data_ptr()+1024,512,224*512,p(O),256,L*256,None,p(pad),B,L,224,24,st)
ms=timeit(f); print(f'attn cross Lq={L} Lk=224: {ms:.3f} ms  {L*224*128*8*B/ms/1e9:.1f} TF-eq')
img2=torch.zeros(B*8*4*12288,dtype=torch.int16,device=DEV)
lib.ctrlsim_kv_split(p(KV),KV.data_ptr()+1024,512,224*512,None,B,224,4,p(img2),st)
f=lambda: lib.ctrlsim_attention_presplit(0,p(Q),256,L*256,p(img2),4,p(O),256,L*256,None,p(pad),B,L,224,24,st)
ms=timeit(f); print(f'attn cross presplit: {ms:.3f} ms  {L*224*128*8*B/ms/1e9:.1f} TF-eq')
