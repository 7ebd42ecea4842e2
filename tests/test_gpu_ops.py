"""Per-kernel parity: every libomnihip entry point (called through the C-ABI binding)
against the CPU oracle on the same seeded inputs.  Tolerances are stated per test:
bf16 outputs are compared at one-to-two bf16 ulps (2^-8 relative), fp32 paths at 1e-5."""
import math

import pytest
import torch

from omnivideo.utils import synthetic as syn
from oracle import dit as O

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32


@pytest.fixture(scope="module")
def H():
    import omnivideo._hip as H
    assert torch.cuda.is_available()
    H.lib()
    return H


def dev(t, dt=None):
    t = t.cuda()
    return t.to(dt) if dt is not None else t


def rb(x):
    return x.to(BF16).to(F32)


def assert_bf16_close(got, want, ulps=2.0, atol=1e-3, flip_frac=0.0, flip_atol=0.0):
    """|got-want| <= atol + ulps*2^-8*|want| everywhere, except that a fraction flip_frac of
    the elements may be off by up to flip_atol more: an intermediate bf16 rounding that
    lands on the other side of a tie when the fp32 input differs in its last bit."""
    got, want = got.float().cpu(), want.float().cpu()
    err = (got - want).abs()
    tol = atol + ulps * 2.0 ** -8 * want.abs()
    bad = err > tol
    assert not (err > tol + flip_atol).any(), f"max err {err.max():.4g}"
    assert bad.sum() <= flip_frac * bad.numel(), f"{int(bad.sum())}/{bad.numel()} off; max err {err.max():.4g}"


@pytest.mark.parametrize("C,M", [(256, 7), (1536, 33), (5120, 5), (4096, 9), (2048, 7), (3072, 13), (768, 2)])
def test_layernorm_modulate(H, C, M):
    x = syn.randn("ln.x", (M, C), 1, 3.0) + 0.5
    sh, sc = syn.randn("ln.sh", (C,), 1, 0.3), syn.randn("ln.sc", (C,), 1, 0.3)
    w, b = rb(1 + syn.randn("ln.w", (C,), 1, 0.1)), rb(syn.randn("ln.b", (C,), 1, 0.1))
    # plain + modulation, fp32 out
    out = torch.empty(M, C, device="cuda")
    H.layernorm_modulate(dev(x), out, 1e-6, shift=dev(sh), scale=dev(sc))
    ref = O.layer_norm(x, 1e-6) * (1 + sc) + sh
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=1e-5)
    # affine, bf16 out
    outb = torch.empty(M, C, device="cuda", dtype=BF16)
    H.layernorm_modulate(dev(x), outb, 1e-6, weight=dev(w, BF16), bias=dev(b, BF16))
    assert_bf16_close(outb, O.layer_norm(x, 1e-6, w, b), ulps=1.01, atol=1e-6)
    # affine and modulation together, rows with a stride
    wide = torch.full((M, C + 256), float("nan"), device="cuda")
    H.layernorm_modulate(dev(x), wide[:, :C], 1e-6, shift=dev(sh), scale=dev(sc), weight=dev(w, BF16), bias=dev(b, BF16))
    torch.testing.assert_close(wide[:, :C].cpu(), O.layer_norm(x, 1e-6, w, b) * (1 + sc) + sh, rtol=1e-5, atol=1e-5)
    assert torch.isnan(wide[:, C:]).all()
    # block-0 rounding
    H.layernorm_modulate(dev(x), outb, 1e-6, shift=dev(sh), scale=dev(sc), round_norm=True)
    assert_bf16_close(outb, rb(O.layer_norm(x, 1e-6)) * (1 + sc) + sh, ulps=1.01, atol=1e-6, flip_frac=1e-3,
                      flip_atol=0.04)


def test_layernorm_against_reference_module_golden(H, golden_dir):
    """The LayerNorm kernel, fp32 output, against the reference's WanLayerNorm itself (tests/golden/loop_tiny.npz;
    reference model.py:87-97): plain, and with the affine parameters of a block's norm3 (bf16-rounded weights on
    both sides of the comparison are not available there, so the affine case carries the bf16 rounding of the
    parameters: 2^-8 relative)."""
    import os
    import numpy as np
    G = np.load(os.path.join(golden_dir, "loop_tiny.npz"))
    cfg = syn.dit_config("tiny")
    sd = syn.make_unified_state_dict(cfg, 0)
    x = syn.randn("probe.x", (1, 80, cfg["dim"]), 3)[0]
    out = torch.empty(80, cfg["dim"], device="cuda")
    H.layernorm_modulate(dev(x), out, cfg["eps"])
    torch.testing.assert_close(out.cpu(), torch.from_numpy(G["layernorm_plain"]), rtol=1e-5, atol=1e-5)
    w, b = sd["wan_model.blocks.1.norm3.weight"], sd["wan_model.blocks.1.norm3.bias"]
    H.layernorm_modulate(dev(x), out, cfg["eps"], weight=dev(w, BF16), bias=dev(b, BF16))
    torch.testing.assert_close(out.cpu(), torch.from_numpy(G["layernorm_affine"]), rtol=2.0 ** -7, atol=2.0 ** -7)
    torch.testing.assert_close(out.cpu(), O.layer_norm(x, cfg["eps"], rb(w), rb(b)), rtol=1e-5, atol=1e-5)


def test_layernorm_rejects_bad_width(H):
    with pytest.raises(RuntimeError, match="multiple of 256"):
        H.layernorm_modulate(torch.zeros(4, 100, device="cuda"), torch.zeros(4, 100, device="cuda"), 1e-6)


@pytest.mark.parametrize("C,N,grid,L", [(256, 2, (3, 4, 6), 80), (1536, 12, (2, 5, 7), 70), (5120, 40, (1, 3, 4), 12),
                                          # rows shared by two waves, odd row count; 7 x 64 chunks + a partial one
                                          (3072, 24, (1, 3, 5), 15), (3840, 30, (1, 2, 3), 7)])
def test_rmsnorm_rope(H, C, N, grid, L):
    from omnivideo.modules.engine import build_rope_table
    D = C // N
    x = rb(syn.randn("rr.x", (L, C), 2, 2.0))
    w = rb(1 + syn.randn("rr.w", (C,), 2, 0.1))
    n_tok = grid[0] * grid[1] * grid[2]
    out = torch.empty(L, C, device="cuda", dtype=BF16)
    H.rmsnorm_rope(dev(x, BF16), dev(w, BF16), out, D, 1e-6, rope=build_rope_table(D, max(grid), "cuda"),
                   rows_per_sample=L, n_rope=n_tok, grid=grid)
    ref = O.rope_rotate(O.rms_norm(x, w, 1e-6, "bf16").view(L, N, D), O.rope_angles(D, grid)).reshape(L, C)
    # the kernel's fp32 1/rms can differ from the CPU's in the last bit (summation order),
    # which occasionally flips one of the two intermediate bf16 roundings
    assert_bf16_close(out, rb(ref), ulps=1.01, atol=1e-6, flip_frac=2e-3, flip_atol=0.04)
    # no rope (cross-attention q/k, VLM norm)
    H.rmsnorm_rope(dev(x, BF16), dev(w, BF16), out, D, 1e-6)
    assert_bf16_close(out, O.rms_norm(x, w, 1e-6, "bf16"), ulps=1.01, atol=1e-6, flip_frac=2e-3, flip_atol=0.04)


def test_rmsnorm_rope_batched_and_offset(H):
    """Two samples stacked (CFG batch) and a token offset (sequence-parallel shard)."""
    from omnivideo.modules.engine import build_rope_table
    C, N, D, grid = 256, 2, 128, (2, 4, 4)
    S = 40          # 32 real tokens + 8 padded rows per sample
    x = rb(syn.randn("rr2.x", (2 * S, C), 2))
    w = rb(1 + syn.randn("rr2.w", (C,), 2, 0.1))
    rope = build_rope_table(D, 4, "cuda")
    out = torch.empty(2 * S, C, device="cuda", dtype=BF16)
    H.rmsnorm_rope(dev(x, BF16), dev(w, BF16), out, D, 1e-6, rope=rope, rows_per_sample=S, n_rope=32, grid=grid)
    ang = O.rope_angles(D, grid)
    for b in range(2):
        ref = O.rope_rotate(O.rms_norm(x[b * S:(b + 1) * S], w, 1e-6, "bf16").view(S, N, D), ang).reshape(S, C)
        assert_bf16_close(out[b * S:(b + 1) * S], rb(ref), ulps=1.01, atol=1e-6)
    # shard 1 of 2 (rows 20..39 of one sample)
    out2 = torch.empty(20, C, device="cuda", dtype=BF16)
    H.rmsnorm_rope(dev(x[20:40], BF16), dev(w, BF16), out2, D, 1e-6, rope=rope, rows_per_sample=20, tok_offset=20,
                   n_rope=32, grid=grid)
    assert torch.equal(out2.cpu(), out[20:40].cpu())


@pytest.mark.parametrize("M,N,K", [(1, 128, 64), (130, 256, 192), (300, 384, 1536), (257, 1536, 8960),
                                   (777, 256, 128), (1100, 512, 1536),
                                   # >= 512 tiles of 256 x 256: the ring-pipelined large-shape kernel
                                   (8192, 4096, 128), (4999, 6656, 192), (16500, 2048, 320),
                                   # 27 x 19 = 513 tiles: the 8 XCD chunks of the tile walk are not all the same size
                                   (6912, 4864, 256)])
def test_gemm_epilogues(H, M, N, K):
    a = rb(syn.randn("g.a", (M, K), 3))
    w = rb(syn.uniform("g.w", (N, K), 1 / math.sqrt(K), 3))
    b = rb(syn.randn("g.b", (N,), 3, 0.1))
    acc = a @ w.t() + b
    A, W, B = dev(a, BF16), dev(w, BF16), dev(b, BF16)
    out = torch.empty(M, N, device="cuda", dtype=BF16)
    H.gemm(A, W, out, B, H.EPI_BF16)
    assert_bf16_close(out, acc, ulps=1.01, atol=2e-4)
    H.gemm(A, W, out, B, H.EPI_BF16_GELU)
    # the pre-activation is rounded to bf16 before GELU: where the fp32 sum sits on a rounding
    # tie (different summation order) the input moves by one bf16 ulp, so does the output
    assert_bf16_close(out, O.gelu_tanh(rb(acc)), ulps=2.01, atol=2e-3, flip_frac=2e-4, flip_atol=4e-2)
    o32 = torch.empty(M, N, device="cuda")
    H.gemm(A, W, o32, B, H.EPI_F32)
    assert_bf16_close(o32, acc, ulps=1.01, atol=2e-4)
    assert torch.equal(o32.cpu(), rb(o32.cpu()))          # holds bf16-representable values
    x0 = syn.randn("g.x0", (M, N), 3)
    gate = syn.randn("g.gate", (N,), 3)
    xr = dev(x0).clone()
    H.gemm(A, W, xr, B, H.EPI_RESID, gate=dev(gate))
    torch.testing.assert_close(xr.cpu(), x0 + rb(acc) * gate, rtol=2e-2, atol=2e-2)
    xr = dev(x0).clone()
    H.gemm(A, W, xr, None, H.EPI_RESID)
    torch.testing.assert_close(xr.cpu(), x0 + rb(a @ w.t()), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("M,N,K", [(8192, 4096, 1536), (8192, 4096, 8960), (40560, 1536, 192), (9000, 4608, 128)])
def test_gemm_large_kernel_is_repeatable(H, M, N, K):
    """The LDS buffers of the 256 x 256 kernel are re-staged while the loop runs, ordered only by counted
    waits and barriers: a staging/read race would show as run-to-run differences (the accumulation order is
    fixed, so results must be bit-identical) or as a difference from the plain product."""
    g = torch.Generator(device="cuda").manual_seed(11)
    a = torch.randn(M, K, device="cuda", generator=g).to(BF16)
    w = (torch.randn(N, K, device="cuda", generator=g) / 40).to(BF16)
    b = torch.randn(N, device="cuda", generator=g).to(BF16)
    first = torch.empty(M, N, device="cuda", dtype=BF16)
    H.gemm(a, w, first, b, H.EPI_BF16_GELU)
    out = torch.empty_like(first)
    for _ in range(20):
        out.fill_(float("nan"))
        H.gemm(a, w, out, b, H.EPI_BF16_GELU)
        assert torch.equal(out, first)
    rows = torch.arange(0, M, 97, device="cuda")
    ref = torch.nn.functional.gelu((a[rows].float() @ w.float().t() + b.float()).to(BF16).float(), approximate="tanh")
    err = (first[rows].float() - ref).abs().max().item()
    assert err <= 0.05 * max(1.0, ref.abs().max().item()), err


def test_gemm_strided_views(H):
    """Writing one third of a fused [M,3C] buffer and reading a strided A."""
    M, C = 70, 256
    a3 = rb(syn.randn("gs.a", (M, 3 * C), 4))
    w = rb(syn.uniform("gs.w", (C, C), 1 / 16, 4))
    A3 = dev(a3, BF16)
    out3 = torch.zeros(M, 3 * C, device="cuda", dtype=BF16)
    H.gemm(A3[:, C:2 * C], dev(w, BF16), out3[:, 2 * C:], None)
    assert_bf16_close(out3[:, 2 * C:], a3[:, C:2 * C] @ w.t(), ulps=1.01, atol=2e-4)
    assert float(out3[:, :2 * C].abs().max()) == 0.0


def test_gemm_rejects_bad_shapes(H):
    z = lambda *s: torch.zeros(*s, device="cuda", dtype=BF16)
    with pytest.raises(RuntimeError, match="K="):
        H.gemm(z(4, 100), z(128, 100), z(4, 128))
    with pytest.raises(RuntimeError, match="N="):
        H.gemm(z(4, 64), z(100, 64), z(4, 100))
    with pytest.raises(ValueError):
        H.gemm(z(4, 64), z(128, 128), z(4, 128))


@pytest.mark.parametrize("B,Lq,Lk,N,k_len", [
    (1, 80, 80, 2, 72),          # tiny self-attention with padded tail
    (1, 300, 48, 2, None),       # cross-attention, single partial tile
    (2, 257, 257, 3, 200),       # CFG batch, ragged q block, mask inside the last tile
    (1, 1024, 1024, 1, 1024),    # several full tiles
    (1, 33, 700, 2, 641),        # long keys, k_len one past a tile edge
    (1, 64, 64, 1, 1),           # a single valid key
])
def test_flash_attention(H, B, Lq, Lk, N, k_len):
    D = 128
    q = rb(syn.randn("fa.q", (B, Lq, N, D), 5))
    k = rb(syn.randn("fa.k", (B, Lk, N, D), 5))
    v = rb(syn.randn("fa.v", (B, Lk, N, D), 5))
    out = torch.empty(B, Lq, N, D, device="cuda", dtype=BF16)
    H.flash_attn(dev(q, BF16), dev(k, BF16), dev(v, BF16), out, k_len=k_len)
    for b in range(B):
        ref = O.attention(q[b], k[b], v[b], k_len=k_len, flow="fp32")
        # P is rounded to bf16 before the PV product and the output is bf16: 2^-8 relative on
        # |O| plus a small absolute term for cancellation in the weighted sum
        assert_bf16_close(out[b], ref, ulps=2.0, atol=6e-3)


def test_flash_attention_softmax_spike(H):
    """One key dominates from the 3rd tile on: exercises the running-max rescale."""
    D, L = 128, 320
    q = rb(syn.randn("fs.q", (1, 64, 1, D), 6))
    k = rb(syn.randn("fs.k", (1, L, 1, D), 6))
    v = rb(syn.randn("fs.v", (1, L, 1, D), 6))
    k[0, 150, 0] = rb(q[0, 7, 0] * 4.0)         # huge score for query 7 at key 150
    k[0, 300, 0] = rb(q[0, 9, 0] * 6.0)
    out = torch.empty(1, 64, 1, D, device="cuda", dtype=BF16)
    H.flash_attn(dev(q, BF16), dev(k, BF16), dev(v, BF16), out)
    ref = O.attention(q[0], k[0], v[0], flow="fp32")
    assert_bf16_close(out[0], ref, ulps=2.0, atol=6e-3)


@pytest.mark.parametrize("Lq,Lk,n_real", [(130, 512, 37), (70, 200, 63), (257, 6144, 127), (40, 96, 95), (33, 64, 0)])
def test_flash_attention_folded_padding(H, Lq, Lk, n_real):
    """Keys [n_real, Lk) are one repeated row (a zero-padded context after the embedding): reading
    keys [0, n_real] with the last one counted Lk - n_real times must equal attending all Lk keys
    (reference semantics, model.py:174), here checked against the oracle on the full key set."""
    N, D = 2, 128
    q = rb(syn.randn("fp.q", (1, Lq, N, D), 8))
    k = rb(syn.randn("fp.k", (1, Lk, N, D), 8))
    v = rb(syn.randn("fp.v", (1, Lk, N, D), 8))
    k[:, n_real:] = k[:, n_real:n_real + 1]
    v[:, n_real:] = v[:, n_real:n_real + 1]
    out = torch.empty(1, Lq, N, D, device="cuda", dtype=BF16)
    H.flash_attn(dev(q, BF16), dev(k, BF16), dev(v, BF16), out, k_len=n_real + 1, last_key_copies=Lk - n_real)
    ref = O.attention(q[0], k[0], v[0], flow="fp32")
    assert_bf16_close(out[0], ref, ulps=2.0, atol=6e-3)
    full = torch.empty_like(out)
    H.flash_attn(dev(q, BF16), dev(k, BF16), dev(v, BF16), full)
    assert_bf16_close(out[0], full[0].float().cpu(), ulps=2.0, atol=6e-3)


def test_flash_attention_strided_qkv(H):
    """q/k/v as views into one fused [B,L,3C] projection buffer."""
    B, L, N, D = 2, 130, 2, 128
    C = N * D
    qkv = rb(syn.randn("fq.qkv", (B, L, 3 * C), 7))
    Q = dev(qkv, BF16)
    out = torch.empty(B, L, C, device="cuda", dtype=BF16)
    H.flash_attn(Q[:, :, :C].unflatten(2, (N, D)), Q[:, :, C:2 * C].unflatten(2, (N, D)),
                 Q[:, :, 2 * C:].unflatten(2, (N, D)), out.unflatten(2, (N, D)), k_len=100)
    for b in range(B):
        ref = O.attention(qkv[b, :, :C].view(L, N, D), qkv[b, :, C:2 * C].view(L, N, D),
                          qkv[b, :, 2 * C:].view(L, N, D), k_len=100, flow="fp32")
        assert_bf16_close(out[b].view(L, N, D), ref, ulps=2.0, atol=6e-3)


def test_flash_attention_rejects(H):
    z = lambda *s: torch.zeros(*s, device="cuda", dtype=BF16)
    with pytest.raises(RuntimeError, match="head_dim"):
        H.flash_attn(z(1, 8, 1, 64), z(1, 8, 1, 64), z(1, 8, 1, 64), z(1, 8, 1, 64))
    with pytest.raises(RuntimeError, match="k_len"):
        H.flash_attn(z(1, 8, 1, 128), z(1, 8, 1, 128), z(1, 8, 1, 128), z(1, 8, 1, 128), k_len=9)


def test_patchify(H):
    lat = syn.randn("pf.lat", (16, 3, 8, 12), 8)
    for patch in ((1, 2, 2), (1, 4, 4)):
        ref, grid = O.patchify(lat, patch)
        rows = torch.empty(ref.shape, device="cuda", dtype=BF16)
        H.patchify(dev(lat), rows, patch)
        assert torch.equal(rows.float().cpu(), rb(ref))
        H.patchify(dev(lat, BF16), rows, patch)
        assert torch.equal(rows.float().cpu(), rb(ref))


def test_time_embedding(H):
    cfg = syn.dit_config("tiny")
    sd = O.round_weights_bf16(syn.make_unified_state_dict(cfg, 0))
    C = cfg["dim"]
    ts = torch.tensor([999.0, 731.0, 208.0], dtype=torch.float64, device="cuda")
    sin = torch.empty(256, device="cuda")
    H.time_sinusoid(ts, sin, step=1)
    torch.testing.assert_close(sin.cpu(), O.time_sinusoid(731.0, 256)[0].float(), rtol=0, atol=1e-6)
    step = torch.tensor([2], dtype=torch.int32, device="cuda")
    H.time_sinusoid(ts, sin, step_ptr=step)
    torch.testing.assert_close(sin.cpu(), O.time_sinusoid(208.0, 256)[0].float(), rtol=0, atol=1e-6)
    p = "wan_model."
    g = lambda n: dev(sd[p + n], BF16)
    e_tmp, e, e0 = (torch.empty(n, device="cuda") for n in (C, C, 6 * C))
    H.gemv_f32(sin, g("time_embedding.0.weight"), g("time_embedding.0.bias"), e_tmp)
    H.gemv_f32(e_tmp, g("time_embedding.2.weight"), g("time_embedding.2.bias"), e, silu_in=True)
    H.gemv_f32(e, g("time_projection.1.weight"), g("time_projection.1.bias"), e0, silu_in=True)
    re, re0 = O.time_embed(208.0, sd, cfg)
    torch.testing.assert_close(e.cpu(), re, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(e0.cpu().view(6, C), re0, rtol=1e-4, atol=1e-5)


def test_head_unpatchify(H):
    cfg = syn.dit_config("tiny")
    sd = O.round_weights_bf16(syn.make_unified_state_dict(cfg, 0))
    C, grid = cfg["dim"], (3, 4, 6)
    x = syn.randn("hd.x", (80, C), 9)
    e = syn.randn("hd.e", (C,), 9, 0.3)
    ref = O.head_unpatchify(x, e, sd, cfg, grid)
    hm = torch.empty(2, C, device="cuda")
    H.head_modulation(dev(sd["wan_model.head.modulation"].reshape(2, C), BF16), dev(e), hm)
    y = torch.empty(80, C, device="cuda")
    H.layernorm_modulate(dev(x), y, cfg["eps"], shift=hm[0], scale=hm[1])
    out = torch.empty(16, 3, 8, 12, device="cuda")
    H.head_unpatchify(y, dev(sd["wan_model.head.head.weight"], BF16), dev(sd["wan_model.head.head.bias"], BF16), out,
                      grid, cfg["patch_size"], cfg["out_dim"])
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("B,N,L,k_len,copies", [(1, 8, 10240, 10240, 1), (1, 8, 10200, 9999, 1), (2, 4, 10240, 8000, 37),
                                                (2, 12, 20280, 20280, 1)])
def test_attention_tail_balancing_equals_plain_call(H, B, N, L, k_len, copies):
    """With a workspace the trailing query blocks are computed in two key halves and merged: the
    result equals the plain call up to fp32 summation order, and rows outside the split blocks are
    bitwise unchanged."""
    g = torch.Generator(device="cuda").manual_seed(B * 100 + N)
    q, k, v = (torch.randn(B, L, N, 128, device="cuda", generator=g).to(BF16) for _ in range(3))
    nbytes = H.flash_attn_workspace_bytes(B, L, N, k_len)
    assert nbytes > 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    plain, bal = torch.empty_like(q), torch.full_like(q, float("nan"))
    H.flash_attn(q, k, v, plain, k_len=k_len, last_key_copies=copies)
    H.flash_attn(q, k, v, bal, k_len=k_len, last_key_copies=copies, workspace=ws)
    assert torch.isfinite(bal.float()).all()
    d = (bal.float() - plain.float()).abs()
    assert float(d.max()) < 2e-2 and float((d > 0).float().mean()) < 0.2     # a few last-bit differences in the split rows
    changed = (d.amax(dim=(0, 2, 3)) > 0).nonzero().flatten()
    assert changed.numel() > 0 and int(changed.min()) >= L // 2              # only trailing query blocks are split
    # too small a workspace, or none: exactly the plain call
    small = torch.full_like(q, float("nan"))
    H.flash_attn(q, k, v, small, k_len=k_len, last_key_copies=copies, workspace=ws[:1024])
    assert torch.equal(small, plain)
    # shapes that do not profit report 0
    assert H.flash_attn_workspace_bytes(1, 1024, 2, 1024) == 0 and H.flash_attn_workspace_bytes(2, 75600, 40, 75600) == 0


@pytest.mark.parametrize("M,C", [(515, 1536), (100, 5120), (64, 256), (101, 3072), (35, 2048)])
def test_resid_layernorm_equals_gemm_residual_epilogue_then_layernorm(H, M, C):
    """x += y * gate inside the norm kernel == the GEMM's residual epilogue followed by the norm, bit
    for bit, for every output kind."""
    g = torch.Generator(device="cuda").manual_seed(M + C)
    K = 256
    a = (torch.randn(M, K, device="cuda", generator=g)).to(BF16)
    w = (torch.randn(C, K, device="cuda", generator=g) / 16).to(BF16)
    bias = (torch.randn(C, device="cuda", generator=g) * 0.1).to(BF16)
    x0 = torch.randn(M, C, device="cuda", generator=g)
    gate, shift, scale = (torch.randn(C, device="cuda", generator=g) * 0.5 for _ in range(3))
    for use_gate in (True, False):
        gt = gate if use_gate else None
        x_ref = x0.clone()
        H.gemm(a, w, x_ref, bias, H.EPI_RESID, gate=gt)
        y = torch.empty(M, C, device="cuda", dtype=BF16)
        H.gemm(a, w, y, bias, H.EPI_BF16)
        for dt in (BF16, F32, torch.uint8):
            x = x0.clone()
            out = torch.empty(M, C, device="cuda", dtype=dt)
            ref = torch.empty(M, C, device="cuda", dtype=dt)
            if dt == torch.uint8:
                rs, rs_ref = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
                H.layernorm_modulate_fp8(x_ref, ref, rs_ref, 1e-6, shift=shift, scale=scale)
                H.resid_layernorm_modulate(x, y, out, 1e-6, res_gate=gt, shift=shift, scale=scale, row_scale=rs)
                assert torch.equal(rs, rs_ref)
            else:
                H.layernorm_modulate(x_ref, ref, 1e-6, shift=shift, scale=scale)
                H.resid_layernorm_modulate(x, y, out, 1e-6, res_gate=gt, shift=shift, scale=scale)
            assert torch.equal(x, x_ref) and torch.equal(out, ref)


def test_flash_attention_is_repeatable(H):
    """K/V tiles are re-staged by DMA right behind a barrier and the split blocks go through a workspace:
    a race in either shows as run-to-run differences (every summation order is fixed)."""
    g = torch.Generator(device="cuda").manual_seed(5)
    B, L, N = 2, 8320, 8          # 2 * 8 * 33 = 528 workgroups of 256 rows: two rounds and a split tail
    q, k, v = ((torch.randn(B, L, N, 128, device="cuda", generator=g)).to(BF16) for _ in range(3))
    nbytes = H.flash_attn_workspace_bytes(B, L, N, L - 7)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device="cuda") if nbytes else None
    first = torch.empty_like(q)
    H.flash_attn(q, k, v, first, k_len=L - 7, workspace=ws)
    out = torch.empty_like(q)
    for _ in range(10):
        out.fill_(float("nan"))
        H.flash_attn(q, k, v, out, k_len=L - 7, workspace=ws)
        assert torch.equal(out, first)


def test_reference_signature_flash_attention_lens_scale_dtype():
    """omnivideo.modules.flash_attention (reference attention.py:24-133): per-sample k_lens, full
    q_lens, q_scale, softmax_scale, and the result back in q's dtype."""
    from omnivideo.modules import flash_attention
    B, Lq, Lk, N, D = 2, 70, 90, 2, 128
    q = rb(syn.randn("fa2.q", (B, Lq, N, D), 6))
    k = rb(syn.randn("fa2.k", (B, Lk, N, D), 6))
    v = rb(syn.randn("fa2.v", (B, Lk, N, D), 6))
    k_lens = torch.tensor([90, 37])
    out = flash_attention(dev(q), dev(k), dev(v), q_lens=torch.tensor([Lq, Lq]), k_lens=k_lens, softmax_scale=0.07,
                          q_scale=0.5)
    assert out.dtype == F32 and tuple(out.shape) == (B, Lq, N, D)
    for b in range(B):
        # the oracle scales by 1/sqrt(D): fold the ratio to the requested scale into q (fp32 flow: no rounding)
        ref = O.attention(rb(q[b] * 0.5) * (0.07 * D ** 0.5), k[b], v[b], k_len=int(k_lens[b]), flow="fp32")
        assert_bf16_close(out[b], ref, ulps=3.0, atol=8e-3)
    with pytest.raises(NotImplementedError):
        flash_attention(dev(q), dev(k), dev(v), q_lens=torch.tensor([Lq, Lq - 1]))
    with pytest.raises(NotImplementedError):
        flash_attention(dev(q), dev(k), dev(v), causal=True)


@pytest.mark.parametrize("C,heads,S,nb,p", [(256, 2, 40, 2, 2), (1536, 12, 52, 2, 4), (5120, 40, 19, 2, 8), (512, 4, 36, 1, 1)])
def test_qkv_norm_rope_fused_launch_and_send_image(H, C, heads, S, nb, p):
    """One launch for the q and k thirds (own weights, RoPE) equals the two single launches bit for bit;
    with a send buffer it writes q, k, v split by destination rank - the all-to-all image that torch
    chunk/stack copies would build; gather_col_blocks is the inverse regrouping of the way back."""
    from omnivideo.modules.engine import build_rope_table
    D, grid, tok0 = C // heads, (3, 4, 6), 30
    n_tok = 72
    g = torch.Generator(device="cuda").manual_seed(C + S)
    qkv = torch.randn(nb * S, 3 * C, device="cuda", generator=g).to(BF16)
    wq = (torch.randn(C, device="cuda", generator=g) * 0.2 + 1).to(BF16)
    wk = (torch.randn(C, device="cuda", generator=g) * 0.2 + 1).to(BF16)
    rope = build_rope_table(D, max(grid), "cuda")
    ref = qkv.clone()
    for i, w in ((0, wq), (1, wk)):
        v_ = ref[:, i * C:(i + 1) * C]
        H.rmsnorm_rope(v_, w, v_, D, 1e-6, rope=rope, rows_per_sample=S, tok_offset=tok0, n_rope=n_tok, grid=grid)
    inplace = qkv.clone()
    H.qkv_norm_rope(inplace, wq, wk, D, 1e-6, rope, S, tok0, n_tok, grid)
    assert torch.equal(inplace, ref)
    Cs = C // p
    send = torch.full((3, nb, p, S, Cs), float("nan"), device="cuda", dtype=BF16)
    src = qkv.clone()
    H.qkv_norm_rope(src, wq, wk, D, 1e-6, rope, S, tok0, n_tok, grid, send=send)
    assert torch.equal(src, qkv)                                   # the input is left alone
    want = ref.view(nb, S, 3, p, Cs).permute(2, 0, 3, 1, 4)        # [part, sample, dest, row, C/p]
    assert torch.equal(send, want)
    # the way back: [p, S, Cs] blocks -> rows [S, C]
    blocks = torch.randn(p, S, Cs, device="cuda", generator=g).to(BF16)
    rows = torch.full((S, C + 8), float("nan"), device="cuda", dtype=BF16)[:, :C]
    H.gather_col_blocks(blocks, rows)
    assert torch.equal(rows, blocks.permute(1, 0, 2).reshape(S, C))
