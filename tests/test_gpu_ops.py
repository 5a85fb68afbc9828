"""GPU parity of the building-block kernels against plain PyTorch fp32 references (tolerances are fp32 rounding)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from ctrlsim_amd import _lib  # noqa: E402
import model_oracle as mo  # noqa: E402
from gpu_utils import DEV, gemm, gemm_bf16x6  # noqa: E402


@pytest.mark.parametrize("M,N,K,relu,resid", [(128, 128, 256, False, False), (300, 1050, 256, False, False),
                                              (1000, 768, 256, True, False), (777, 256, 1024, False, True),
                                              (5000, 32, 256, False, False), (24, 1000, 256, False, False),
                                              (4097, 512, 512, False, False)])
def test_gemm_nt(M, N, K, relu, resid):
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=g).to(DEV)
    W = (torch.randn(N, K, generator=g) * 0.1).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    R = torch.randn(M, N, generator=g).to(DEV) if resid else None
    out = gemm(A, W, b, R, relu)
    ref = (A.double() @ W.double().T + b.double())
    if resid:
        ref = ref + R.double()
    if relu:
        ref = ref.clamp_min(0)
    err = (out.double() - ref).abs().max().item()
    assert err < 5e-5 * math.sqrt(K / 256), err
    # transpose-detecting: asymmetric W, identity-like A picks columns of W^T
    A2 = torch.zeros(M, K, device=DEV); A2[torch.arange(M), torch.arange(M) % K] = 1.0
    out2 = gemm(A2, W, None, None, False)
    assert torch.equal(out2, W.T[torch.arange(M) % K])


@pytest.mark.parametrize("M,N,K,relu,resid,n0", [(128, 256, 256, False, False, 0), (300, 1050, 256, False, False, 0),
                                                 (1000, 768, 256, True, False, 0), (777, 256, 1024, False, True, 0),
                                                 (5000, 256, 256, False, False, 256), (24, 1000, 256, False, False, 0),
                                                 (4097, 512, 512, False, False, 256),
                                                 # one, odd and few k-steps: prologue / tail of the operand pipeline
                                                 (300, 200, 16, False, False, 0), (257, 384, 48, True, False, 0),
                                                 (1000, 130, 80, False, True, 0), (64, 128, 32, False, False, 128)])
def test_gemm_bf16x6_has_fp32_class_accuracy(M, N, K, relu, resid, n0):
    """The split-bf16 GEMM must be as accurate as the f32-input MFMA GEMM: both are compared with an fp64 reference."""
    g = torch.Generator().manual_seed(M + N + 7)
    A = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).to(DEV)     # rows of varying scale
    Wfull = (torch.randn(n0 + N, K, generator=g) * 0.1)
    b = torch.randn(N, generator=g).to(DEV)
    R = torch.randn(M, N, generator=g).to(DEV) if resid else None
    out = gemm_bf16x6(A, Wfull, b, R, relu, n0=n0, n=N)
    Wd = Wfull[n0:n0 + N].to(DEV).contiguous()
    out32 = gemm(A, Wd, b, R, relu) if K % 32 == 0 else None       # the f32-input kernel steps K by 32
    ref = A.double() @ Wd.double().T + b.double()
    if resid:
        ref = ref + R.double()
    if relu:
        ref = ref.clamp_min(0)
    scale = A.double().abs() @ Wd.double().abs().T + b.double().abs()          # sum |a||w| + |b| (+ |r|): the natural error scale
    if resid:
        scale = scale + R.double().abs()
    scale = scale.clamp_min(1e-30)
    e6 = ((out.double() - ref).abs() / scale).max().item()
    e32 = ((out32.double() - ref).abs() / scale).max().item() if out32 is not None else 1.0
    print(f"relative-to-sum|ab| error: bf16x6 {e6:.2e}  f32 mfma {e32:.2e}")
    assert e6 < 4e-7 and e6 < 8 * e32 + 1e-9, (e6, e32)


@pytest.mark.parametrize("M,K,relu,resid,ldc", [(1003, 256, False, True, 256), (130, 1024, False, True, 256),
                                                 (517, 512, True, False, 512), (24, 256, True, False, 256),
                                                 (333, 16, False, True, 256), (200, 48, False, False, 256)])
def test_gemm_bf16x6_fused_layernorm(M, K, relu, resid, ldc):
    """LayerNorm in the GEMM epilogue (in place over the residual when there is one) against torch in float64."""
    g = torch.Generator().manual_seed(M + K)
    A = torch.randn(M, K, generator=g).to(DEV)
    W = torch.randn(256, K, generator=g) * 0.1
    b = torch.randn(256, generator=g).to(DEV)
    gam = torch.randn(256, generator=g).to(DEV); bet = torch.randn(256, generator=g).to(DEV)
    buf = torch.randn(M, ldc, generator=g).to(DEV)
    R = buf if resid else None
    before = buf.clone()
    out = gemm_bf16x6(A, W, b, R, relu, ln=(gam, bet), out=buf)
    pre = A.double() @ W.to(DEV).double().T + b.double() + (before[:, :256].double() if resid else 0)
    ref = torch.nn.functional.layer_norm(pre, (256,), gam.double(), bet.double(), 1e-5)
    if relu:
        ref = ref.clamp_min(0)
    assert (out[:, :256].double() - ref).abs().max().item() < 2e-5
    if ldc > 256:
        assert torch.equal(out[:, 256:], before[:, 256:])     # columns beyond N untouched


@pytest.fixture(params=[0, 7], ids=["tiled", "weight-stationary"])
def ws_option(request):
    """OPT_GEMM_WS (csrc/common.h): every Linear(256 -> 256 G) family through the tiled kernel, or through the weight-stationary one."""
    _lib.lib().ctrlsim_set_option(6, request.param)
    yield request.param
    _lib.lib().ctrlsim_set_option(6, 15)                    # the default (common.h: weight-stationary + row-stationary in_proj)


@pytest.mark.parametrize("M,relu,resid,ln", [(5000, False, False, False), (8191, False, True, False), (4097, True, False, False),
                                              (31, False, True, True), (32, True, True, True), (33, False, False, True),
                                              (16385, False, True, True), (20000, True, False, True)])
def test_linear256_both_kernels(ws_option, M, relu, resid, ln):
    """Linear(256 -> 256) [+ residual] [+ LayerNorm] [+ ReLU] against float64, for both kernels that serve the shape (row counts
    around the 32-row block of the weight-stationary kernel and beyond one block per compute unit)."""
    g = torch.Generator().manual_seed(M)
    A = (torch.randn(M, 256, generator=g) * torch.exp(0.5 * torch.randn(M, 1, generator=g))).to(DEV)
    W = torch.randn(512, 256, generator=g) * 0.1
    b = torch.randn(256, generator=g).to(DEV)
    gam = torch.randn(256, generator=g).to(DEV); bet = torch.randn(256, generator=g).to(DEV)
    buf = torch.randn(M, 256, generator=g).to(DEV)
    before = buf.clone()
    out = gemm_bf16x6(A, W, b, buf if resid else None, relu, n0=256, n=256, ln=(gam, bet) if ln else None, out=buf)
    ref = A.double() @ W[256:].to(DEV).double().T + b.double() + (before.double() if resid else 0)
    if ln:
        ref = torch.nn.functional.layer_norm(ref, (256,), gam.double(), bet.double(), 1e-5)
    if relu:
        ref = ref.clamp_min(0)
    scale = (A.double().abs() @ W[256:].to(DEV).double().abs().T + 1).max().item() if not ln else 1.0
    assert (out.double() - ref).abs().max().item() < 2e-5 * max(1.0, scale / 50)


@pytest.mark.parametrize("M", [33, 5000, 40000])
def test_linear256_with_scattered_row_store(M):
    """Round 6: the plain Linear(256 -> 256) of the weight-stationary kernel writing result row i to row c_rows[i] of a LARGER row space
    (ctrlsim_gemm256_rows; the map encoder's last Linear fills the polyline rows of the scene-encoder source — modules/encoder.py:155-158's
    concatenation — without a copy kernel): bit-identical to Linear + row copy, entries < 0 are not stored, every other row of the
    destination is untouched."""
    from ctrlsim_amd.pack import split3_planes
    g = torch.Generator().manual_seed(M)
    A = torch.randn(M, 256, generator=g).to(DEV)
    W = torch.randn(512, 256, generator=g) * 0.1
    b = torch.randn(256, generator=g).to(DEV)
    ref = gemm_bf16x6(A, W, b, n0=256, n=256)
    rows = M + M // 8 + 24                                               # 200 polyline rows + 24 vehicle rows per context, as the real map
    perm = torch.randperm(rows, generator=g)[:M].to(torch.int32)
    perm[::97] = -1
    idx = perm.to(DEV)
    dst = torch.full((rows, 256), 7.5, device=DEV)
    planes = torch.from_numpy(split3_planes(W.numpy()).view(np.int16).copy()).to(DEV)
    p = _lib.ptr
    rc = _lib.lib().ctrlsim_gemm256_rows(p(A), 256, p(planes), 512, 256, p(b), p(dst), 256, p(idx), M, _lib.stream_ptr())
    assert rc == 0, rc                                                   # (1 = kernel not applicable: the shipped default must take it)
    torch.cuda.synchronize()
    keep = idx >= 0
    assert torch.equal(dst[idx[keep].long()], ref[keep])
    untouched = torch.ones(rows, dtype=torch.bool, device=DEV)
    untouched[idx[keep].long()] = False
    assert bool((dst[untouched] == 7.5).all())


# (160 x 288 = 46 080 rows = 1 440 row blocks of 32: more than four per compute unit, so every persistent workgroup of the weight-stationary
#  kernel runs its multi-job ring with result stores and the next blocks' requests in flight across the counted-vmcnt barriers)
@pytest.mark.parametrize("B,L,col0", [(3, 224, 256), (2, 96, 256), (5, 160, 0), (1, 2304, 256), (160, 288, 256), (150, 292, 0)])
def test_linear_with_kv_image_epilogue(ws_option, B, L, col0):
    """The in_proj Linear whose key / value columns leave as split K / V tile images (ctrlsim_gemm_nt_kv): the images must drive
    the attention kernel to the same output as images split from the fp32 result of the plain Linear, and the fp32 (query)
    columns must match it."""
    g = torch.Generator().manual_seed(B * L + col0)
    N, M, nkt = col0 + 512, B * L, (L + 63) // 64
    A = torch.randn(M, 256, generator=g).to(DEV)
    W = torch.randn(N, 256, generator=g) * 0.08
    b = torch.randn(N, generator=g).to(DEV)
    y = gemm_bf16x6(A, W, b)                                   # fp32 rows [M, N] (tiled kernel: N is not 256)
    p = _lib.ptr
    lib, st = _lib.lib(), _lib.stream_ptr()
    img_ref = _kv_images(y.data_ptr() + 4 * col0, y.data_ptr() + 4 * (col0 + 256), N, L * N, B, L, nkt)
    from ctrlsim_amd.pack import split3_planes
    planes = torch.from_numpy(split3_planes(W.numpy()).view(np.int16).copy()).to(DEV)
    C = torch.full((M, max(col0, 4)), float("nan"), device=DEV)
    img = torch.zeros_like(img_ref)                            # the caller zeroes the tiles' tails
    _lib.check(lib.ctrlsim_gemm_nt_kv(p(A), 256, p(planes), N, 0, p(b), p(C), C.stride(0), M, N, 256, p(img), L, nkt, col0, st))
    if col0:
        assert (C - y[:, :col0]).abs().max().item() < 1e-5
    Q = torch.randn(B, L, 256, generator=g).to(DEV)
    O0 = torch.zeros(B, L, 256, device=DEV); O1 = torch.zeros_like(O0)
    pad = torch.zeros(B, L, dtype=torch.uint8, device=DEV)
    for im, O in ((img_ref, O0), (img, O1)):
        _lib.check(lib.ctrlsim_attention_presplit(0, p(Q), 256, L * 256, p(im), nkt, p(O), 256, L * 256, None, p(pad), B, L, L, 1, st))
    assert torch.isfinite(O1).all() and (O0 - O1).abs().max().item() < 2e-5
    used = B * 8 * nkt * (8192 if lib.ctrlsim_split_scheme() == 1 else 12288)
    same = (img.view(-1)[:used] == img_ref.view(-1)[:used]).float().mean().item()   # the leading planes agree bit for bit almost everywhere
    assert same > 0.6, same


@pytest.mark.parametrize("B,L,col0", [(3, 224, 256), (2, 96, 256), (5, 160, 0), (1, 2304, 256), (160, 288, 256), (300, 288, 256), (450, 292, 0)])
def test_row_stationary_linear_with_kv_images(B, L, col0):
    """The same in_proj through the ROW-stationary kernel (ctrlsim_gemm_kv_blocks: rows in registers, 32-column weight blocks streamed
    through LDS; one job = 256 rows, so the cases cover a partial single job, one job per compute unit, and more than one job per
    workgroup with the counted waits in flight and a partial last job): query columns and attention driven by its K / V images against the
    plain Linear + split pass."""
    lib = _lib.lib()
    if lib.ctrlsim_split_scheme() != 1:
        pytest.skip("two-fp16-plane scheme only")
    g = torch.Generator().manual_seed(7 * B * L + col0)
    N, M, nkt = col0 + 512, B * L, (L + 63) // 64
    A = (torch.randn(M, 256, generator=g) * torch.exp(0.3 * torch.randn(M, 1, generator=g))).to(DEV)
    W = torch.randn(N, 256, generator=g) * 0.08
    b = torch.randn(N, generator=g).to(DEV)
    y = gemm_bf16x6(A, W, b)
    p, st = _lib.ptr, _lib.stream_ptr()
    img_ref = _kv_images(y.data_ptr() + 4 * col0, y.data_ptr() + 4 * (col0 + 256), N, L * N, B, L, nkt)
    from ctrlsim_amd.pack import row_blocks
    blocks = torch.from_numpy(row_blocks(W.numpy(), 1).view(np.int16).copy()).to(DEV)
    C = torch.full((M, max(col0, 4)), float("nan"), device=DEV)
    img = torch.zeros_like(img_ref)
    _lib.check(lib.ctrlsim_gemm_kv_blocks(p(A), 256, p(blocks), p(b), p(C), C.stride(0), M, N, p(img), L, nkt, col0, st))
    if col0:
        ref = A.double() @ W[:col0].to(DEV).double().T + b[:col0].double()
        scale = (A.double().abs() @ W[:col0].to(DEV).double().abs().T + 1).max().item()
        assert (C.double() - ref).abs().max().item() < 2e-5 * max(1.0, scale / 50)
        assert (C - y[:, :col0]).abs().max().item() < 1e-5 * max(1.0, scale / 50)
    Q = torch.randn(B, L, 256, generator=g).to(DEV)
    O0 = torch.zeros(B, L, 256, device=DEV); O1 = torch.zeros_like(O0)
    pad = torch.zeros(B, L, dtype=torch.uint8, device=DEV)
    for im, O in ((img_ref, O0), (img, O1)):
        _lib.check(lib.ctrlsim_attention_presplit(0, p(Q), 256, L * 256, p(im), nkt, p(O), 256, L * 256, None, p(pad), B, L, L, 1, st))
    assert torch.isfinite(O1).all() and (O0 - O1).abs().max().item() < 2e-5
    used = B * 8 * nkt * 8192
    same = (img.view(-1)[:used] == img_ref.view(-1)[:used]).float().mean().item()
    assert same > 0.6, same
    # the tails of the last tiles were left alone (the caller zeroes them)
    if L % 64:
        v = img.view(-1)[:used].view(B * 8, nkt, 2, 2, 64 * 32)       # [ctx*head][tile][K|V][plane][...]
        ktail = v[:, -1, 0].reshape(B * 8, 2, 4, 64, 8)[:, :, :, L % 64:, :]
        assert (ktail == 0).all()


@pytest.mark.parametrize("M,N", [(300, 256), (65536, 256), (70001, 512), (16385, 64)])
def test_row_stationary_plain_linear(M, N):
    """ctrlsim_gemm_kv_blocks without images: a plain Linear(256 -> N) through the row-stationary kernel, against float64."""
    lib = _lib.lib()
    if lib.ctrlsim_split_scheme() != 1:
        pytest.skip("two-fp16-plane scheme only")
    g = torch.Generator().manual_seed(M + N)
    A = (torch.randn(M, 256, generator=g) * torch.exp(0.3 * torch.randn(M, 1, generator=g))).to(DEV)
    W = torch.randn(N, 256, generator=g) * 0.1
    b = torch.randn(N, generator=g).to(DEV)
    from ctrlsim_amd.pack import row_blocks
    blocks = torch.from_numpy(row_blocks(W.numpy(), 1).view(np.int16).copy()).to(DEV)
    C = torch.full((M, N + 4), float("nan"), device=DEV)
    p = _lib.ptr
    _lib.check(lib.ctrlsim_gemm_kv_blocks(p(A), 256, p(blocks), p(b), p(C), C.stride(0), M, N, None, 0, 0, 0, _lib.stream_ptr()))
    ref = A.double() @ W.to(DEV).double().T + b.double()
    scale = (A.double().abs() @ W.to(DEV).double().abs().T + 1).max().item()
    assert (C[:, :N].double() - ref).abs().max().item() < 2e-5 * max(1.0, scale / 50)
    assert torch.isnan(C[:, N:]).all()                        # columns beyond N untouched


def test_layernorm_and_in_place():
    g = torch.Generator().manual_seed(1)
    X = torch.randn(1003, 256, generator=g).to(DEV) * 3 + 1
    R = torch.randn(1003, 256, generator=g).to(DEV)
    gam = torch.randn(256, generator=g).to(DEV); bet = torch.randn(256, generator=g).to(DEV)
    Y = torch.empty_like(X)
    p = _lib.ptr
    for relu, radd in ((0, None), (1, R)):
        _lib.check(_lib.lib().ctrlsim_layernorm256(p(X), 256, p(radd), 256, p(gam), p(bet), p(Y), 256, 1003, relu,
                                                   _lib.stream_ptr()))
        ref = torch.nn.functional.layer_norm(X + (radd if radd is not None else 0), (256,), gam, bet, 1e-5)
        if relu:
            ref = ref.relu()
        assert (Y - ref).abs().max().item() < 2e-5
    Z = X.clone()
    _lib.check(_lib.lib().ctrlsim_layernorm256(p(Z), 256, None, 0, p(gam), p(bet), p(Z), 256, 1003, 0, _lib.stream_ptr()))
    assert (Z - torch.nn.functional.layer_norm(X, (256,), gam, bet, 1e-5)).abs().max().item() < 2e-5


def _attn_ref(q, k, v, vis):  # q [B,H,Lq,32] ... vis bool [B,1|H,Lq,Lk]
    s = (q.double() @ k.double().transpose(-1, -2)) / math.sqrt(32)
    s = s.masked_fill(~vis, float("-inf"))
    return (torch.softmax(s, -1) @ v.double())


@pytest.fixture(params=[0, 1], ids=["f32mfma", "bf16x6"])
def attn_impl(request):
    _lib.lib().ctrlsim_set_option(0, request.param)
    yield request.param
    _lib.lib().ctrlsim_set_option(0, 1)


@pytest.mark.parametrize("A,T", [(24, 32), (4, 4), (6, 8), (24, 7)])
def test_attention_structured_causal_mask(A, T, attn_impl):
    B, H = 2, 8
    L = A * T * 3
    g = torch.Generator().manual_seed(A * T)
    qkv = torch.randn(B, L, 768, generator=g).to(DEV)
    O = torch.zeros(B, L, 256, device=DEV)
    p = _lib.ptr
    _lib.check(_lib.lib().ctrlsim_attention(1, p(qkv), 768, L * 768, qkv.data_ptr() + 256 * 4, qkv.data_ptr() + 512 * 4, 768,
                                            L * 768, p(O), 256, L * 256, None, None, B, L, L, A, _lib.stream_ptr()))
    q, k, v = [qkv[..., i * 256:(i + 1) * 256].view(B, L, H, 32).transpose(1, 2) for i in range(3)]
    vis = mo.causal_mask_closed_form(A, T, 3).to(DEV)[None, None]
    ref = _attn_ref(q, k, v, vis).transpose(1, 2).reshape(B, L, 256)
    assert (O.double() - ref).abs().max().item() < 2e-5
    # gathered queries: the A rtg tokens of timestep ti, compact Q/O buffers
    ti = T - 1
    pos = torch.tensor([(ti * A + a) * 3 + 1 for a in range(A)], dtype=torch.int32, device=DEV)
    qc = qkv[:, pos.long(), :].contiguous()
    Oc = torch.zeros(B, A, 256, device=DEV)
    _lib.check(_lib.lib().ctrlsim_attention(1, p(qc), 768, A * 768, qkv.data_ptr() + 256 * 4, qkv.data_ptr() + 512 * 4, 768,
                                            L * 768, p(Oc), 256, A * 256, p(pos), None, B, A, L, A, _lib.stream_ptr()))
    assert (Oc.double() - ref[:, pos.long()]).abs().max().item() < 2e-5


@pytest.mark.parametrize("kind", ["sharp", "rising", "falling"])
def test_attention_sharp_logits_exercise_speculative_and_rescale_paths(kind, attn_impl):
    """The split-operand kernel exponentiates against its current softmax base and moves the base only when a row sum reaches
    2^15 (a probability would leave the fp16 range of the split's leading plane).  Sharp logits (std 6), keys whose scores rise
    steadily (new maxima in every tile until the bound trips, again and again) and keys whose scores fall (tiny probabilities
    after a large first one) against float64."""
    B, H, A, T = 2, 8, 24, 16
    L = A * T * 3
    g = torch.Generator().manual_seed(len(kind))
    qkv = torch.randn(B, L, 768, generator=g)
    if kind == "sharp":
        qkv[..., :256] *= 6.0
    else:
        # one shared direction u: q = c u + noise, k_j = ramp_j u + noise  ->  score_j ~ c * ramp_j * |u|^2 / sqrt(32)
        u = torch.randn(H, 32, generator=g); u = u / u.norm(dim=-1, keepdim=True)
        ramp = torch.linspace(0.0, 40.0, L) * (1.0 if kind == "rising" else -1.0)
        qkv[..., :256] = 0.3 * qkv[..., :256] + (math.sqrt(32) * u).reshape(1, 1, 256)
        qkv[..., 256:512] = 0.3 * qkv[..., 256:512] + (ramp[:, None, None] * u[None]).reshape(1, L, 256)
    qkv = qkv.to(DEV)
    O = torch.zeros(B, L, 256, device=DEV)
    p = _lib.ptr
    _lib.check(_lib.lib().ctrlsim_attention(1, p(qkv), 768, L * 768, qkv.data_ptr() + 256 * 4, qkv.data_ptr() + 512 * 4, 768,
                                            L * 768, p(O), 256, L * 256, None, None, B, L, L, A, _lib.stream_ptr()))
    q, k, v = [qkv[..., i * 256:(i + 1) * 256].view(B, L, H, 32).transpose(1, 2) for i in range(3)]
    vis = mo.causal_mask_closed_form(A, T, 3).to(DEV)[None, None]
    ref = _attn_ref(q, k, v, vis).transpose(1, 2).reshape(B, L, 256)
    assert torch.isfinite(O).all()
    assert (O.double() - ref).abs().max().item() < 3e-5


@pytest.mark.parametrize("mode", [2, 3], ids=["il", "trajeglish"])
@pytest.mark.parametrize("A,T", [(24, 32), (6, 8), (24, 7)])
def test_attention_mask_variants_of_the_baselines(A, T, mode):
    """Modes 2 / 3 of the structured mask: the IL (state, action) and Trajeglish (action only) models keep the 3-slot token layout
    and the token types they lack are dead as keys.  Restricted to the live tokens the visibility must be get_causal_mask with
    2 / 1 token types (oracle closed form, pinned against the reference in tests/golden/variants.npz)."""
    B, H = 2, 8
    L = A * T * 3
    g = torch.Generator().manual_seed(A * T + mode)
    qkv = torch.randn(B, L, 768, generator=g).to(DEV)
    O = torch.zeros(B, L, 256, device=DEV)
    p = _lib.ptr
    _lib.check(_lib.lib().ctrlsim_attention(mode, p(qkv), 768, L * 768, qkv.data_ptr() + 256 * 4, qkv.data_ptr() + 512 * 4, 768,
                                            L * 768, p(O), 256, L * 256, None, None, B, L, L, A, _lib.stream_ptr()))
    live = [0, 2] if mode == 2 else [2]                              # token types the model has, in slot order
    K = len(live)
    idx = torch.tensor([(ta * 3 + k) for ta in range(T * A) for k in live], device=DEV)
    q, k, v = [qkv[..., i * 256:(i + 1) * 256].view(B, L, H, 32).transpose(1, 2) for i in range(3)]
    vis = mo.causal_mask_closed_form(A, T, K).to(DEV)[None, None]
    ref = _attn_ref(q[:, :, idx], k[:, :, idx], v[:, :, idx], vis).transpose(1, 2).reshape(B, len(idx), 256)
    assert (O[:, idx].double() - ref).abs().max().item() < 2e-5
    assert torch.isfinite(O).all()                                    # the dead rows hold finite values too
    # gathered queries (the last layer's compact rows): state tokens (IL) / action tokens (Trajeglish) of the last step
    ti, off = T - 1, (0 if mode == 2 else 2)
    pos = torch.tensor([(ti * A + a) * 3 + off for a in range(A)], dtype=torch.int32, device=DEV)
    qc = qkv[:, pos.long(), :].contiguous()
    Oc = torch.zeros(B, A, 256, device=DEV)
    _lib.check(_lib.lib().ctrlsim_attention(mode, p(qc), 768, A * 768, qkv.data_ptr() + 256 * 4, qkv.data_ptr() + 512 * 4, 768,
                                            L * 768, p(Oc), 256, A * 256, p(pos), None, B, A, L, A, _lib.stream_ptr()))
    assert (Oc.double() - O[:, pos.long()].double()).abs().max().item() < 2e-5


@pytest.mark.parametrize("A,T", [(24, 32), (6, 8), (24, 7), (5, 13), (2, 9)])
def test_attention_mask_with_attend_own_return_action(A, T):
    """Mode 5 (round 6): the CtRL-Sim mask under cfg.model.attend_own_return_action (utils/train_utils.py:114-129) — of the earlier
    timesteps a query sees the state tokens and its own agent's return / action tokens only — against float64 attention under the
    oracle's closed form (pinned to the reference's get_causal_mask: tests/golden/own_return.npz), over all rows and over the gathered
    rows of the few-row launches (state tokens of the last step, rtg tokens of the last step)."""
    B, H = 2, 8
    L = A * T * 3
    g = torch.Generator().manual_seed(A * T + 5)
    qkv = torch.randn(B, L, 768, generator=g).to(DEV)
    O = torch.zeros(B, L, 256, device=DEV)
    p = _lib.ptr
    _lib.check(_lib.lib().ctrlsim_attention(5, p(qkv), 768, L * 768, qkv.data_ptr() + 256 * 4, qkv.data_ptr() + 512 * 4, 768,
                                            L * 768, p(O), 256, L * 256, None, None, B, L, L, A, _lib.stream_ptr()))
    vis = mo.causal_mask_closed_form(A, T, 3, 0, True).to(DEV)[None, None]
    assert not torch.equal(vis, mo.causal_mask_closed_form(A, T, 3, 0, False).to(DEV)[None, None]) or A == 1
    q, k, v = [qkv[..., i * 256:(i + 1) * 256].view(B, L, H, 32).transpose(1, 2) for i in range(3)]
    ref = _attn_ref(q, k, v, vis).transpose(1, 2).reshape(B, L, 256)
    assert (O.double() - ref).abs().max().item() < 2e-5
    for off in (0, 1):                                               # the last layer's queried rows / the second pass's rtg rows
        pos = torch.tensor([((T - 1) * A + a) * 3 + off for a in range(A)], dtype=torch.int32, device=DEV)
        qc = qkv[:, pos.long(), :].contiguous()
        Oc = torch.zeros(B, A, 256, device=DEV)
        _lib.check(_lib.lib().ctrlsim_attention(5, p(qc), 768, A * 768, qkv.data_ptr() + 256 * 4, qkv.data_ptr() + 512 * 4, 768,
                                                L * 768, p(Oc), 256, A * 256, p(pos), None, B, A, L, A, _lib.stream_ptr()))
        assert (Oc.double() - ref[:, pos.long()]).abs().max().item() < 2e-5


@pytest.mark.parametrize("A,T", [(24, 32), (6, 8), (24, 7), (5, 13)])
def test_attention_mask_of_the_decision_transformer(A, T):
    """Mode 4: get_causal_mask with state_index 1 for the token order (rtg, state, action), evaluated on tokens stored in the
    slots (state, rtg, action) — i.e. the reference's mask with rows and columns permuted accordingly."""
    B, H = 2, 8
    L = A * T * 3
    g = torch.Generator().manual_seed(A * T + 4)
    qkv = torch.randn(B, L, 768, generator=g).to(DEV)
    O = torch.zeros(B, L, 256, device=DEV)
    p = _lib.ptr
    _lib.check(_lib.lib().ctrlsim_attention(4, p(qkv), 768, L * 768, qkv.data_ptr() + 256 * 4, qkv.data_ptr() + 512 * 4, 768,
                                            L * 768, p(O), 256, L * 256, None, None, B, L, L, A, _lib.stream_ptr()))
    # position of slot (state, rtg, action) token in the reference's (rtg, state, action) sequence
    perm = torch.tensor([(i // 3) * 3 + (1, 0, 2)[i % 3] for i in range(L)], device=DEV)
    vis_ref = mo.causal_mask_closed_form(A, T, 3, 1).to(DEV)
    vis = vis_ref[perm][:, perm][None, None]
    q, k, v = [qkv[..., i * 256:(i + 1) * 256].view(B, L, H, 32).transpose(1, 2) for i in range(3)]
    ref = _attn_ref(q, k, v, vis).transpose(1, 2).reshape(B, L, 256)
    assert (O.double() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("Lq,Lk", [(224, 224), (2304, 224), (10, 10), (24, 224), (130, 67)])
def test_attention_key_padding(Lq, Lk, attn_impl):
    B, H = 3, 8
    g = torch.Generator().manual_seed(Lq + Lk)
    Q = torch.randn(B, Lq, 256, generator=g).to(DEV)
    KV = torch.randn(B, Lk, 512, generator=g).to(DEV)
    pad = (torch.rand(B, Lk, generator=g) < 0.3)
    pad[:, 0] = False
    pad_d = pad.to(torch.uint8).to(DEV)
    O = torch.zeros(B, Lq, 256, device=DEV)
    p = _lib.ptr
    _lib.check(_lib.lib().ctrlsim_attention(0, p(Q), 256, Lq * 256, p(KV), KV.data_ptr() + 256 * 4, 512, Lk * 512, p(O), 256,
                                            Lq * 256, None, p(pad_d), B, Lq, Lk, 24, _lib.stream_ptr()))
    q = Q.view(B, Lq, H, 32).transpose(1, 2)
    k = KV[..., :256].reshape(B, Lk, H, 32).transpose(1, 2)
    v = KV[..., 256:].reshape(B, Lk, H, 32).transpose(1, 2)
    vis = (~pad).to(DEV)[:, None, None, :].expand(B, 1, Lq, Lk)
    ref = _attn_ref(q, k, v, vis).transpose(1, 2).reshape(B, Lq, 256)
    assert (O.double() - ref).abs().max().item() < 2e-5


def _kv_images(K_ptr, V_ptr, ldkv, kbs, B, rows, nkt, pos=None, img=None):
    img = torch.full((B * 8 * nkt * 12288,), 0x7FC0, dtype=torch.int16, device=DEV) if img is None else img   # NaN-filled
    _lib.check(_lib.lib().ctrlsim_kv_split(K_ptr, V_ptr, ldkv, kbs, _lib.ptr(pos), B, rows, nkt, _lib.ptr(img),
                                           _lib.stream_ptr()), "kv_split")
    return img


def _compact_masks(Areg, T, rep):
    """(visible, m-fold) boolean matrices [L, L] of a compact context's row order — regular rows (t, a < Areg, k) at (t Areg + a) 3 + k,
    then, with rep, the representative's rows (t, k) at Lreg + 3 t + k — written from the rules, not from the kernel: a query of step
    tq sees every key of an earlier step, of its own step the state tokens (k = 0) and its own slot's tokens up to itself
    (utils/train_utils.py:81-129); the representative stands for `mult` equal padded slots, so its keys count mult-fold, except that its
    own later tokens of the step are seen by itself only once (csrc/attention_bf16x6.hip, kernel header)."""
    rows = [(t, a, k) for t in range(T) for a in range(Areg) for k in range(3)]
    if rep:
        rows += [(t, -1, k) for t in range(T) for k in range(3)]
    n = len(rows)
    vis = torch.zeros(n, n, dtype=torch.bool); mul = torch.zeros(n, n, dtype=torch.bool)
    for i, (tq, aq, kq) in enumerate(rows):
        for j, (tk, ak, kk) in enumerate(rows):
            if tk < tq or (tk == tq and kk == 0):
                vis[i, j] = True; mul[i, j] = ak < 0
            elif tk == tq and ak == aq and kk <= kq:
                vis[i, j] = True                                  # own tokens of the step (the representative's: once)
    return vis, mul


@pytest.mark.parametrize("Actx,T", [(24, 32), (8, 32), (12, 32), (5, 7), (4, 32), (16, 9), (20, 3), (7, 1)])
def test_attention_with_mask_tables_equals_in_kernel_masks_and_float64(Actx, T):
    """Round 4: the causal launches over the token rows take their visibility masks from a per-class table (ctrlsim_attention_mask_table,
    one v_cndmask per score) instead of building them per query.  Plain (24-slot) and compact classes, full and short windows (key ranges
    that end inside a sub-tile, query groups that straddle the regular / representative boundary): against float64 with the mask written
    from the rules, and against the in-kernel-mask kernel (bit-identical for plain contexts: same arithmetic, same order)."""
    lib, st, p = _lib.lib(), _lib.stream_ptr(), _lib.ptr
    B, H = 2, 8
    rep = 1 if Actx < 24 else 0
    Areg = Actx - rep
    mult = 24 - Areg
    Lreg = T * 3 * Areg
    L = Lreg + rep * 3 * T
    rep_k0 = (Lreg + 63) // 64 * 64
    nkt = (Lreg + 63) // 64 + rep * ((3 * T + 63) // 64)
    g = torch.Generator().manual_seed(Actx * 100 + T)
    qkv = torch.randn(B, L, 768, generator=g).to(DEV)
    Kp, Vp = qkv.data_ptr() + 1024, qkv.data_ptr() + 2048
    key_pos = torch.tensor([i if i < Lreg else rep_k0 + (i - Lreg) for i in range(L)], dtype=torch.int32, device=DEV)
    nel = 8192 if lib.ctrlsim_split_scheme() == 1 else 12288
    img = torch.zeros(B * 8 * nkt * nel, dtype=torch.int16, device=DEV)
    _kv_images(Kp, Vp, 768, L * 768, B, L, nkt, pos=key_pos, img=img)
    O_old = torch.zeros(B, L, 256, device=DEV); O_new = torch.full_like(O_old, float("nan"))
    _lib.check(lib.ctrlsim_attention_compact(p(qkv), 768, L * 768, p(img), nkt, p(O_old), 256, L * 256, None, B, L, Lreg, Areg,
                                             rep * 3 * T, mult, Lreg, st))
    nbytes = lib.ctrlsim_attention_mask_table_bytes(L, nkt)
    assert nbytes > 0
    tbl = torch.full((nbytes // 8,), -1, dtype=torch.int64, device=DEV)          # all-ones: an unwritten entry would show
    _lib.check(lib.ctrlsim_attention_mask_table(L, Lreg, Areg, rep * 3 * T, Lreg, nkt, p(tbl), st))
    _lib.check(lib.ctrlsim_attention_tbl(p(qkv), 768, L * 768, p(img), nkt, p(O_new), 256, L * 256, B, L, Lreg, Areg, rep * 3 * T, mult,
                                         p(tbl), st))
    vis, mul = _compact_masks(Areg, T, rep)
    q, k, v = [qkv[..., i * 256:(i + 1) * 256].view(B, L, H, 32).transpose(1, 2) for i in range(3)]
    sc = (q.double() @ k.double().transpose(-1, -2)) / math.sqrt(32)
    sc = sc + math.log(mult) * mul.to(DEV)[None, None].double() if rep else sc
    sc = sc.masked_fill(~vis.to(DEV)[None, None], float("-inf"))
    ref = (torch.softmax(sc, -1) @ v.double()).transpose(1, 2).reshape(B, L, 256)
    assert torch.isfinite(O_new).all()
    assert (O_old.double() - ref).abs().max().item() < 2e-5
    assert (O_new.double() - ref).abs().max().item() < 2e-5
    if rep == 0:
        assert torch.equal(O_old, O_new)
    else:
        assert (O_old - O_new).abs().max().item() < 5e-6


@pytest.mark.parametrize("Actx,T,which", [(24, 32, "last3"), (24, 32, "state"), (8, 32, "last3"), (12, 9, "last3"), (5, 32, "state"),
                                          (16, 20, "mixed"), (4, 3, "last3")])
def test_attention_few_query_streaming_form(Actx, T, which):
    """Round 4, option 9: launches with at most 96 queries per context (the second pass, the last decoder layer on the queried rows, the
    K/V-cached steps) run one wave per (context, head, 32 queries) with the K / V fragments read straight from the tile images.  Same
    arithmetic in the same order as the LDS-staged form: bit-identical; and against float64 with the mask written from the rules.
    Plain and compact contexts; the queries are the last step's tokens (representative's included), its state tokens, or a mix of steps."""
    lib, st, p = _lib.lib(), _lib.stream_ptr(), _lib.ptr
    B, H = 3, 8
    rep = 1 if Actx < 24 else 0
    Areg = Actx - rep
    mult = 24 - Areg
    Lreg = T * 3 * Areg
    L = Lreg + rep * 3 * T
    rep_k0 = (Lreg + 63) // 64 * 64
    nkt = (Lreg + 63) // 64 + rep * ((3 * T + 63) // 64)
    g = torch.Generator().manual_seed(Actx * 1000 + T)
    qkv = torch.randn(B, L, 768, generator=g).to(DEV)
    Kp, Vp = qkv.data_ptr() + 1024, qkv.data_ptr() + 2048
    key_pos = torch.tensor([i if i < Lreg else rep_k0 + (i - Lreg) for i in range(L)], dtype=torch.int32, device=DEV)
    nel = 8192 if lib.ctrlsim_split_scheme() == 1 else 12288
    img = torch.zeros(B * 8 * nkt * nel, dtype=torch.int16, device=DEV)
    _kv_images(Kp, Vp, 768, L * 768, B, L, nkt, pos=key_pos, img=img)
    rows_of = lambda t, ks: [(t * Areg + a) * 3 + k for a in range(Areg) for k in ks] + ([Lreg + 3 * t + k for k in ks] if rep else [])
    if which == "last3":
        rows = rows_of(T - 1, (0, 1, 2))
    elif which == "state":
        rows = rows_of(T - 1, (0,))
    else:
        rows = rows_of(T - 1, (1, 2)) + rows_of(T // 2, (0, 2)) + rows_of(0, (0,))
    rows = rows[:96]
    Lq = len(rows)
    # the kernel takes query POSITIONS in the window layout: regular rows as they are, the representative's rows from rep_pos0 = Lreg on
    q_pos = torch.tensor(rows, dtype=torch.int32, device=DEV)
    Q = qkv[:, rows, :256].contiguous()
    outs = []
    for opt in (1, 0):
        lib.ctrlsim_set_option(9, opt)
        O = torch.full((B, Lq, 256), float("nan"), device=DEV)
        try:
            _lib.check(lib.ctrlsim_attention_compact(p(Q), 256, Lq * 256, p(img), nkt, p(O), 256, Lq * 256, p(q_pos), B, Lq, Lreg, Areg,
                                                     rep * 3 * T, mult, Lreg, st))
        finally:
            lib.ctrlsim_set_option(9, 1)
        outs.append(O)
    vis, mul = _compact_masks(Areg, T, rep)
    q = Q.view(B, Lq, H, 32).transpose(1, 2)
    k, v = [qkv[..., i * 256:(i + 1) * 256].view(B, L, H, 32).transpose(1, 2) for i in (1, 2)]
    sc = (q.double() @ k.double().transpose(-1, -2)) / math.sqrt(32)
    visq, mulq = vis[rows].to(DEV)[None, None], mul[rows].to(DEV)[None, None]
    sc = sc + math.log(mult) * mulq.double() if rep else sc
    sc = sc.masked_fill(~visq, float("-inf"))
    ref = (torch.softmax(sc, -1) @ v.double()).transpose(1, 2).reshape(B, Lq, 256)
    assert torch.isfinite(outs[0]).all()
    assert (outs[0].double() - ref).abs().max().item() < 2e-5
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("Lq,Lk", [(24, 224), (72, 224), (1, 67), (96, 130)])
def test_attention_few_query_streaming_form_key_padding(Lq, Lk):
    """The same for the key-padding mode over pre-split images (cross attention of the few-row passes)."""
    lib, st, p = _lib.lib(), _lib.stream_ptr(), _lib.ptr
    B, H = 4, 8
    g = torch.Generator().manual_seed(Lq * 7 + Lk)
    Q = torch.randn(B, Lq, 256, generator=g).to(DEV)
    KV = torch.randn(B, Lk, 512, generator=g).to(DEV)
    pad = (torch.rand(B, Lk, generator=g) < 0.3); pad[:, 0] = False
    pad[1] = False                                             # a context without padded keys: the bias path is skipped
    pad_d = pad.to(torch.uint8).to(DEV)
    nkt = (Lk + 63) // 64
    img = _kv_images(p(KV), KV.data_ptr() + 1024, 512, Lk * 512, B, Lk, nkt)
    outs = []
    for opt in (1, 0):
        lib.ctrlsim_set_option(9, opt)
        O = torch.full((B, Lq, 256), float("nan"), device=DEV)
        try:
            _lib.check(lib.ctrlsim_attention_presplit(0, p(Q), 256, Lq * 256, p(img), nkt, p(O), 256, Lq * 256, None, p(pad_d), B, Lq, Lk,
                                                      24, st))
        finally:
            lib.ctrlsim_set_option(9, 1)
        outs.append(O)
    q = Q.view(B, Lq, H, 32).transpose(1, 2)
    k = KV[..., :256].reshape(B, Lk, H, 32).transpose(1, 2)
    v = KV[..., 256:].reshape(B, Lk, H, 32).transpose(1, 2)
    vis = (~pad).to(DEV)[:, None, None, :].expand(B, 1, Lq, Lk)
    ref = _attn_ref(q, k, v, vis).transpose(1, 2).reshape(B, Lq, 256)
    assert (outs[0].double() - ref).abs().max().item() < 2e-5
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("A,T", [(24, 32), (6, 8), (24, 7)])
def test_attention_presplit_images_match_in_kernel_split(A, T):
    """K/V split once into bf16 images + DMA staging must give bit-identical output to the in-kernel split, for the
    full build, for a row-scatter (KV-cache) build, and in both mask modes."""
    B = 2
    L = A * T * 3
    nkt = (L + 63) // 64
    g = torch.Generator().manual_seed(A + T)
    qkv = torch.randn(B, L, 768, generator=g).to(DEV)
    p = _lib.ptr
    lib, st = _lib.lib(), _lib.stream_ptr()
    Kp, Vp = qkv.data_ptr() + 1024, qkv.data_ptr() + 2048
    O0 = torch.zeros(B, L, 256, device=DEV); O1 = torch.zeros_like(O0); O2 = torch.zeros_like(O0)
    _lib.check(lib.ctrlsim_attention(1, p(qkv), 768, L * 768, Kp, Vp, 768, L * 768, p(O0), 256, L * 256, None, None, B, L, L, A, st))
    img = _kv_images(Kp, Vp, 768, L * 768, B, L, nkt)
    _lib.check(lib.ctrlsim_attention_presplit(1, p(qkv), 768, L * 768, p(img), nkt, p(O1), 256, L * 256, None, None, B, L, L, A, st))
    assert torch.equal(O0, O1)
    # scatter build: zeroed images, rows written in a shuffled order through the position list
    perm = torch.randperm(L, generator=g)
    rows = qkv[:, perm, :].contiguous()
    img2 = torch.zeros_like(img)
    pos = perm.to(torch.int32).to(DEV)
    _kv_images(rows.data_ptr() + 1024, rows.data_ptr() + 2048, 768, L * 768, B, L, nkt, pos=pos, img=img2)
    _lib.check(lib.ctrlsim_attention_presplit(1, p(qkv), 768, L * 768, p(img2), nkt, p(O2), 256, L * 256, None, None, B, L, L, A, st))
    assert torch.equal(O0, O2)
    # key-padding mode over a ragged memory length
    Lk = 224 if L > 224 else L - 3
    nk2 = (Lk + 63) // 64
    KV = torch.randn(B, Lk, 512, generator=g).to(DEV)
    pad = (torch.rand(B, Lk, generator=g) < 0.3); pad[:, 0] = False
    pad_d = pad.to(torch.uint8).to(DEV)
    Q = qkv[..., :256].contiguous()
    Oa = torch.zeros(B, L, 256, device=DEV); Ob = torch.zeros_like(Oa)
    _lib.check(lib.ctrlsim_attention(0, p(Q), 256, L * 256, p(KV), KV.data_ptr() + 1024, 512, Lk * 512, p(Oa), 256, L * 256, None,
                                     p(pad_d), B, L, Lk, A, st))
    img3 = _kv_images(p(KV), KV.data_ptr() + 1024, 512, Lk * 512, B, Lk, nk2)
    _lib.check(lib.ctrlsim_attention_presplit(0, p(Q), 256, L * 256, p(img3), nk2, p(Ob), 256, L * 256, None, p(pad_d), B, L, Lk,
                                              A, st))
    assert torch.equal(Oa, Ob) and torch.isfinite(Ob).all()


@pytest.mark.parametrize("M,F", [(128, 1024), (1000, 1024), (37, 64), (4133, 512)])
def test_ffn_fused_block(M, F):
    """linear1 -> ReLU -> linear2 -> + x -> LayerNorm as one kernel (hidden tile in registers) vs torch in float64,
    out of place and in place."""
    from ctrlsim_amd.pack import ffn_planes
    g = torch.Generator().manual_seed(M + F)
    X = (torch.randn(M, 256, generator=g) * torch.exp(0.5 * torch.randn(M, 1, generator=g))).to(DEV)
    W1 = torch.randn(F, 256, generator=g) * 0.08
    W2 = torch.randn(256, F, generator=g) * 0.05
    b1 = torch.randn(F, generator=g).to(DEV) * 0.3
    b2 = torch.randn(256, generator=g).to(DEV) * 0.3
    gam = torch.randn(256, generator=g).to(DEV); bet = torch.randn(256, generator=g).to(DEV)
    w1p, w2p = ffn_planes(W1.numpy(), W2.numpy())
    w1d = torch.from_numpy(w1p.view(np.int16).copy()).to(DEV)
    w2d = torch.from_numpy(w2p.view(np.int16).copy()).to(DEV)
    Y = torch.empty_like(X)
    p = _lib.ptr
    _lib.check(_lib.lib().ctrlsim_ffn_fused(p(X), 256, p(w1d), p(b1), p(w2d), p(b2), p(gam), p(bet), p(Y), 256, M, F,
                                            _lib.stream_ptr()), "ffn_fused")
    Xd = X.double()
    h = torch.relu(Xd @ W1.to(DEV).double().T + b1.double())
    pre = Xd + h @ W2.to(DEV).double().T + b2.double()
    ref = torch.nn.functional.layer_norm(pre, (256,), gam.double(), bet.double(), 1e-5)
    err = (Y.double() - ref).abs().max().item()
    print("fused ffn max abs err", err)
    assert err < 3e-5
    Z = X.clone()
    _lib.check(_lib.lib().ctrlsim_ffn_fused(p(Z), 256, p(w1d), p(b1), p(w2d), p(b2), p(gam), p(bet), p(Z), 256, M, F,
                                            _lib.stream_ptr()), "ffn_fused in place")
    assert torch.equal(Z, Y)


@pytest.mark.parametrize("M,F", [(128, 1024), (1000, 1024), (37, 64), (4133, 512), (40000, 1024)])
def test_outproj_layernorm_ffn_as_one_kernel(M, F):
    """The post-LN block in front of a feed-forward block fused into it (ctrlsim_ffn_fused_pre): x1 = LayerNorm0(R + O Wo^T + bo),
    y = LayerNorm(x1 + W2 relu(W1 x1 + b1) + b2) — nn.TransformerDecoderLayer's multihead_attn.out_proj + norm2 + feed-forward + norm3 /
    nn.TransformerEncoderLayer's self_attn.out_proj + norm1 + feed-forward + norm2 (modules/decoder.py:16-20, encoder.py:42-46) — against
    torch in float64, out of place and with y aliasing the residual rows; and against the two-kernel route it replaces."""
    from ctrlsim_amd.pack import ffn_planes_pre, ffn_planes, split3_planes
    if _lib.lib().ctrlsim_get_option(0) != 1:
        pytest.skip("two-fp16-plane scheme only")
    g = torch.Generator().manual_seed(M + F + 1)
    O = torch.randn(M, 256, generator=g).to(DEV)
    R = (torch.randn(M, 256, generator=g) * torch.exp(0.5 * torch.randn(M, 1, generator=g))).to(DEV)
    Wo = torch.randn(256, 256, generator=g) * 0.07
    W1 = torch.randn(F, 256, generator=g) * 0.08
    W2 = torch.randn(256, F, generator=g) * 0.05
    bo, g0, be0 = (torch.randn(256, generator=g).to(DEV) * s for s in (0.3, 1.0, 0.2))
    b1 = torch.randn(F, generator=g).to(DEV) * 0.3
    b2 = torch.randn(256, generator=g).to(DEV) * 0.3
    gam = torch.randn(256, generator=g).to(DEV); bet = torch.randn(256, generator=g).to(DEV)
    wop, w1q, w2p = ffn_planes_pre(Wo.numpy(), W1.numpy(), W2.numpy(), 1)
    dev = lambda a: torch.from_numpy(a.view(np.int16).copy()).to(DEV)
    wod, w1d, w2d = dev(wop), dev(w1q), dev(w2p)
    Y = torch.empty_like(R)
    p = _lib.ptr
    call = lambda Rr, Yy: _lib.check(_lib.lib().ctrlsim_ffn_fused_pre(p(O), 256, p(Rr), 256, p(wod), p(bo), p(g0), p(be0), p(w1d), p(b1), p(w2d),
                                                                      p(b2), p(gam), p(bet), p(Yy), 256, M, F, _lib.stream_ptr()), "ffn_fused_pre")
    call(R, Y)
    x1 = torch.nn.functional.layer_norm(R.double() + O.double() @ Wo.to(DEV).double().T + bo.double(), (256,), g0.double(), be0.double(), 1e-5)
    h = torch.relu(x1 @ W1.to(DEV).double().T + b1.double())
    ref = torch.nn.functional.layer_norm(x1 + h @ W2.to(DEV).double().T + b2.double(), (256,), gam.double(), bet.double(), 1e-5)
    err = (Y.double() - ref).abs().max().item()
    print("out-proj + LN + ffn as one kernel: max abs err", err)
    assert err < 5e-5
    Z = R.clone()
    call(Z, Z)
    assert torch.equal(Z, Y)
    assert _lib.lib().ctrlsim_ffn_fused_pre(p(O), 256, p(R), 256, p(wod), p(bo), p(g0), p(be0), p(w1d), p(b1), p(w2d), p(b2), p(gam), p(bet),
                                            p(Y), 256, M, 3104, _lib.stream_ptr()) != 0          # beyond the LDS the bias tables fit in


@pytest.mark.parametrize("M", [1, 129, 33000])
def test_outproj_layernorm_ffn_with_strided_rows_and_output_over_the_attention_rows(M):
    """ctrlsim_ffn_fused_pre with three different leading dimensions (attention output 320, residual 288, output 272 floats) and with the
    output written over the ATTENTION rows (the other aliasing the kernel allows): a wave reads all of its rows before its first store, the
    rows of the next row block it requests ahead are not written by anybody else.  One row, one row past a row block, many row blocks per
    workgroup (33 000 rows = 258 row blocks on 256 persistent workgroups: the ahead-of-time requests of the last trip are clamped)."""
    from ctrlsim_amd.pack import ffn_planes_pre
    if _lib.lib().ctrlsim_get_option(0) != 1:
        pytest.skip("two-fp16-plane scheme only")
    F = 256
    g = torch.Generator().manual_seed(M)
    Ob = torch.randn(M, 320, generator=g).to(DEV); Rb = torch.randn(M, 288, generator=g).to(DEV)
    Wo = torch.randn(256, 256, generator=g) * 0.07; W1 = torch.randn(F, 256, generator=g) * 0.08; W2 = torch.randn(256, F, generator=g) * 0.05
    bo, g0, be0, b2, gam, bet = (torch.randn(256, generator=g).to(DEV) * s for s in (0.3, 1.0, 0.2, 0.3, 1.0, 0.3))
    b1 = torch.randn(F, generator=g).to(DEV) * 0.3
    dev = lambda a: torch.from_numpy(a.view(np.int16).copy()).to(DEV)
    wod, w1d, w2d = (dev(a) for a in ffn_planes_pre(Wo.numpy(), W1.numpy(), W2.numpy(), 1))
    O, R = Ob[:, :256], Rb[:, :256]
    x1 = torch.nn.functional.layer_norm(R.double() + O.double() @ Wo.to(DEV).double().T + bo.double(), (256,), g0.double(), be0.double(), 1e-5)
    h = torch.relu(x1 @ W1.to(DEV).double().T + b1.double())
    ref = torch.nn.functional.layer_norm(x1 + h @ W2.to(DEV).double().T + b2.double(), (256,), gam.double(), bet.double(), 1e-5)
    p = _lib.ptr
    Yb = torch.full((M, 272), float("nan"), device=DEV)
    run = lambda Yp, ldy: _lib.check(_lib.lib().ctrlsim_ffn_fused_pre(p(Ob), 320, p(Rb), 288, p(wod), p(bo), p(g0), p(be0), p(w1d), p(b1), p(w2d), p(b2),
                                                                      p(gam), p(bet), Yp, ldy, M, F, _lib.stream_ptr()), "ffn_fused_pre")
    run(p(Yb), 272)
    assert (Yb[:, :256].double() - ref).abs().max().item() < 5e-5 and torch.isnan(Yb[:, 256:]).all()
    keep = Ob[:, 256:].clone()
    run(p(Ob), 320)                                           # output over the attention rows
    assert torch.equal(Ob[:, :256], Yb[:, :256]) and torch.equal(Ob[:, 256:], keep)


@pytest.mark.parametrize("M", [128, 1000, 37, 4133, 40000])
def test_outproj_layernorm_query_projection_as_one_kernel(M):
    """x1 = LayerNorm0(R + O Wo^T + bo), q = x1 Wq^T + bq (ctrlsim_outproj_ln_q): nn.TransformerDecoderLayer's self_attn.out_proj + norm1 and
    the query third of multihead_attn.in_proj (modules/decoder.py:16-20) against torch in float64; x1 written over the residual rows."""
    from ctrlsim_amd.pack import outproj_q_planes
    if _lib.lib().ctrlsim_get_option(0) != 1:
        pytest.skip("two-fp16-plane scheme only")
    g = torch.Generator().manual_seed(M + 7)
    O = torch.randn(M, 256, generator=g).to(DEV)
    R = (torch.randn(M, 256, generator=g) * torch.exp(0.5 * torch.randn(M, 1, generator=g))).to(DEV)
    Wo = torch.randn(256, 256, generator=g) * 0.07
    Wq = torch.randn(256, 256, generator=g) * 0.09
    bo, g0, be0, bq = (torch.randn(256, generator=g).to(DEV) * s for s in (0.3, 1.0, 0.2, 0.4))
    wop, wqp = outproj_q_planes(Wo.numpy(), Wq.numpy(), 1)
    dev = lambda a: torch.from_numpy(a.view(np.int16).copy()).to(DEV)
    wod, wqd = dev(wop), dev(wqp)
    X1 = torch.full((M, 256), float("nan"), device=DEV); Q = torch.full((M, 320), float("nan"), device=DEV)
    p = _lib.ptr
    call = lambda Rr, Xx, Qq, ldq: _lib.check(_lib.lib().ctrlsim_outproj_ln_q(p(O), 256, p(Rr), 256, p(wod), p(bo), p(g0), p(be0), p(wqd), p(bq),
                                                                              p(Xx), 256, p(Qq), ldq, M, _lib.stream_ptr()), "outproj_ln_q")
    call(R, X1, Q, 320)
    x1 = torch.nn.functional.layer_norm(R.double() + O.double() @ Wo.to(DEV).double().T + bo.double(), (256,), g0.double(), be0.double(), 1e-5)
    q = x1 @ Wq.to(DEV).double().T + bq.double()
    e1 = (X1.double() - x1).abs().max().item(); e2 = (Q[:, :256].double() - q).abs().max().item()
    print("out-proj + LN + q as one kernel: max abs err x1", e1, "q", e2)
    assert e1 < 2e-5 and e2 < 3e-5 and torch.isnan(Q[:, 256:]).all()
    Z = R.clone(); Q2 = torch.empty(M, 256, device=DEV)
    call(Z, Z, Q2, 256)
    assert torch.equal(Z, X1) and torch.equal(Q2, Q[:, :256])


@pytest.mark.parametrize("B,NP", [(1, 100), (3, 100), (2, 127), (2, 128), (1, 200), (1, 256)])
def test_map_pool_matches_the_unfolded_front_end_in_float64(B, NP):
    """map_pool_kernel (point MLP + single-seed attention pooling with every linear stage folded at pack time: csrc/map_encoder.hip) against
    a float64 evaluation of the UNFOLDED front end written from modules/map_encoder.py:28-46 — Linear(3, 256) - LayerNorm - ReLU -
    Linear(256, 256) per point, an 8-head attention whose only query is the learned seed, key padding = missing points, a polyline without
    any point un-masks its point 0 — up to, not including, out_proj.  Polylines with 0, 1, 2, 31, 32, 33, 99 and 100 visible points, points
    missing in the middle; padding bytes exact, pooled vectors to fp32 accuracy.  Round 6 (packed-fp32 kernel: a polyline's compact point range
    starts at an even slot and an odd range ends in a zero-weight pad point): 127 / 128 points per polyline (two polylines fill the 256 slots
    with and without pads) and 200 / 256 (one polyline per workgroup)."""
    from ctrlsim_amd import spec, weights
    from ctrlsim_amd.engine import HipModel
    cfg = spec.make_cfg(dataset__waymo__max_num_road_pts_per_polyline=NP)
    d = spec.Dims(cfg)
    assert d.NP == NP
    wts = weights.generate(d, 0)
    model = HipModel(cfg, wts, DEV)
    lib, p, st = _lib.lib(), _lib.ptr, _lib.stream_ptr()
    rs = np.random.RandomState(B)
    npts = rs.randint(0, d.NP + 1, (B, d.P))
    npts[0, :8] = [0, 1, 2, 31, 32, 33, NP - 1, NP]
    npts[0, 12:16] = [NP, NP, NP - 1, NP]                        # full neighbours in one workgroup, odd + full
    ex = (np.arange(d.NP)[None, None] < npts[..., None]).astype(np.float32)
    ex[0, 9, 5:40] = 0.0                                        # holes: existence is a per-point flag, not a prefix
    ex[0, 10, ::2] = 0.0
    rp = np.concatenate([(rs.randn(B, d.P, d.NP, 2) * 60).astype(np.float32), ex[..., None]], -1)
    rp[..., :2] *= ex[..., None]
    road = torch.from_numpy(rp).to(DEV)
    out = torch.full((B * d.P, d.D), float("nan"), device=DEV)
    pad = torch.full((B, d.P), 7, dtype=torch.uint8, device=DEV)
    _lib.check(lib.ctrlsim_map_pool(model.handle, B, p(road), p(out), p(pad), st))
    torch.cuda.synchronize()
    got, got_pad = out.cpu().numpy().astype(np.float64), pad.cpu().numpy()
    # ---- the unfolded computation
    W = {k.split("map_encoder.")[1]: np.asarray(v, np.float64) for k, v in wts.items() if "map_encoder." in k}
    x = rp.reshape(-1, d.NP, 3).astype(np.float64)
    h = x @ W["road_pts_encoder.mlp.0.weight"].T + W["road_pts_encoder.mlp.0.bias"]
    h = (h - h.mean(-1, keepdims=True)) / np.sqrt(h.var(-1, keepdims=True) + 1e-5)
    h = np.maximum(h * W["road_pts_encoder.mlp.1.weight"] + W["road_pts_encoder.mlp.1.bias"], 0.0)
    h = h @ W["road_pts_encoder.mlp.3.weight"].T + W["road_pts_encoder.mlp.3.bias"]                # [polylines, NP, 256]
    Wi, bi = W["road_pts_attn_layer.in_proj_weight"], W["road_pts_attn_layer.in_proj_bias"]
    q = (Wi[:256] @ W["map_seeds"].reshape(256) + bi[:256]).reshape(8, 32)
    k = (h @ Wi[256:512].T + bi[256:512]).reshape(-1, d.NP, 8, 32)
    v = (h @ Wi[512:].T + bi[512:]).reshape(-1, d.NP, 8, 32)
    vis = x[..., 2] != 0
    empty = ~vis.any(1)
    vis[empty, 0] = True                                          # map_encoder.py:31
    sc = np.einsum("hd,pnhd->pnh", q, k) / np.sqrt(32.0)
    sc = np.where(vis[..., None], sc, -np.inf)
    a = np.exp(sc - sc.max(1, keepdims=True)); a /= a.sum(1, keepdims=True)
    ref = np.einsum("pnh,pnhd->phd", a, v).reshape(-1, 256)
    assert np.array_equal(got_pad.reshape(-1), empty.astype(np.uint8))
    assert np.isfinite(got).all()
    scale = np.abs(ref).max(1, keepdims=True) + 1e-3
    err = np.abs(got - ref) / scale
    assert err.max() < 2e-5, (err.max(), np.unravel_index(err.argmax(), err.shape))
