"""Three launches each of the shipped causal attention kernel (mask tables, staged K/V images) at the plain class (L = 2304, B = 256) and at the
compact class A' = 8 (L = 768, B = 768): the target of the SQ counter passes of profiles/r05_c_pmc_attention.md."""
import sys
sys.path.insert(0, '.')
import torch
import ctrlsim_amd  # noqa: F401
from ctrlsim_amd import _lib
DEV = 'cuda:0'
lib = _lib.lib(); p = _lib.ptr; st = _lib.stream_ptr()
NPL = 2 if lib.ctrlsim_split_scheme() == 1 else 3
T = 32
for Actx, B in ((24, 256), (8, 768)):
    plain = Actx == 24
    Ar = 24 if plain else Actx - 1
    Lreg = T * 3 * Ar; rep = 0 if plain else 3 * T; Lq = Lreg + rep
    nkt = (Lreg + 63) // 64 + (0 if plain else 2)
    qkv = torch.randn(B, Lq, 768, device=DEV); O = torch.empty(B, Lq, 256, device=DEV)
    img = torch.randn(B * 8 * nkt * 4096 * NPL // 2, device=DEV).to(torch.float16).view(torch.int16) if False else \
        (torch.randn(B * 8 * nkt * 4096 * NPL, device=DEV) * 0.5).to(torch.float16).view(torch.int16)
    tbl = torch.zeros(lib.ctrlsim_attention_mask_table_bytes(Lq, nkt) // 8, dtype=torch.int64, device=DEV)
    lib.ctrlsim_attention_mask_table(Lq, Lreg, Ar, rep, Lreg, nkt, p(tbl), st)
    torch.cuda.synchronize()
    for _ in range(3):
        lib.ctrlsim_attention_tbl(p(qkv), 768, Lq * 768, p(img), nkt, p(O), 256, Lq * 256, B, Lq, Lreg, Ar, rep, 25 - Actx if not plain else 1, p(tbl), st)
    torch.cuda.synchronize()
    del qkv, O, img, tbl
