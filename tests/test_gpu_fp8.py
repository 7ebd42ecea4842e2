"""FP8 (e4m3) mode of the linear layers.  The reference has no FP8 path; these tests pin the
kernels to their definitions (torch's float8_e4m3fn conversion, exact products of the
dequantised operands) and to this library's own bf16 entries."""
import math

import pytest
import torch

from conftest import check_rel

pytestmark = pytest.mark.gpu
BF16, F32, F8 = torch.bfloat16, torch.float32, torch.float8_e4m3fn


@pytest.fixture(scope="module")
def H():
    import omnivideo._hip as H
    H.lib()
    return H


def randn(shape, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(*shape, device="cuda", generator=g) * scale


def ref_quant(x):
    """the kernel's definition with torch ops on the CPU (IEEE division; torch's GPU kernels divide by a
    scalar through its reciprocal): scale = absmax / 448, q = e4m3(x * (1 / scale))"""
    xf = x.float().cpu()
    mx = xf.abs().amax(dim=1)
    s = torch.where(mx > 0, mx / 448.0, torch.ones_like(mx))
    q = (xf * (1.0 / s)[:, None]).clamp(-448, 448).to(F8)
    return q.view(torch.uint8).to(x.device), s.to(x.device)


@pytest.mark.parametrize("M,K", [(37, 512), (1000, 1536), (300, 8960), (64, 5120), (5, 13824), (9, 264)])
def test_quant_rows_matches_torch_float8(H, M, K):
    x = randn((M, K), 1, 3.0).to(BF16)
    x[0] = 0
    x[1, 5] = 1e4
    q, s = H.quant_rows_fp8(x)
    qr, sr = ref_quant(x)
    assert torch.equal(s, sr)
    assert torch.equal(q, qr)
    # dequantised rows are within e4m3's half-ulp (2^-4 relative) of the input
    deq = q.view(F8).float() * s[:, None]
    err = (deq - x.float()).abs()
    assert (err <= 2.0 ** -4 * x.float().abs() + s[:, None] * 2.0 ** -10).all()


@pytest.mark.parametrize("M,C", [(777, 1536), (33, 5120), (31, 3072), (5, 512)])
def test_layernorm_fp8_equals_bf16_kernel_then_quant(H, M, C):
    x = randn((M, C), 2, 2.0) + 0.3
    shift, scale = randn((C,), 3, 0.3), randn((C,), 4, 0.3)
    w, b = (randn((C,), 5, 0.2) + 1).to(BF16), randn((C,), 6, 0.1).to(BF16)
    for kw in (dict(shift=shift, scale=scale), dict(weight=w, bias=b), dict(shift=shift, scale=scale, round_norm=True)):
        ref = torch.empty(M, C, device="cuda", dtype=BF16)
        H.layernorm_modulate(x, ref, 1e-6, **kw)
        qr, sr = ref_quant(ref)
        q = torch.empty(M, C, device="cuda", dtype=torch.uint8)
        s = torch.empty(M, device="cuda")
        H.layernorm_modulate_fp8(x, q, s, 1e-6, **kw)
        assert torch.equal(s, sr) and torch.equal(q, qr)


def test_gemm_fp8_exact_on_small_integers(H):
    """integer operands: every product and partial sum is exact, so any operand-layout or k-order
    mistake shows as a wrong integer (asymmetric operands)."""
    M, N, K = 300, 512, 256
    g = torch.Generator(device="cuda").manual_seed(7)
    a = torch.randint(-3, 4, (M, K), device="cuda", generator=g).float()
    w = torch.randint(-2, 3, (N, K), device="cuda", generator=g).float()
    w[:, ::7] += 1
    a8, w8 = a.to(F8).view(torch.uint8), w.to(F8).view(torch.uint8)
    ones = torch.ones(N, device="cuda")
    out = torch.zeros(M, N, device="cuda")
    H.gemm_fp8(a8, w8, ones, out, None, H.EPI_F32)
    assert torch.equal(out, (a @ w.t()).to(BF16).float())


@pytest.mark.parametrize("M,N,K", [(1000, 512, 128), (777, 256, 192), (4096, 1536, 1536), (515, 768, 8960)])
def test_gemm_fp8_epilogues_vs_dequantised_product(H, M, N, K):
    a = randn((M, K), 11).to(BF16)
    w = randn((N, K), 12, 1 / math.sqrt(K)).to(BF16)
    bias = randn((N,), 13, 0.1).to(BF16)
    a8, sa = H.quant_rows_fp8(a)
    w8, sw = H.quant_rows_fp8(w)
    acc = ((a8.view(F8).float() * sa[:, None]).double() @ (w8.view(F8).float() * sw[:, None]).double().t()).float() \
        + bias.float()

    def close(got, want, ulps=1.01, atol=3e-4):
        err = (got.float() - want.float()).abs()
        assert (err <= atol + ulps * 2.0 ** -8 * want.float().abs()).all(), float(err.max())

    out = torch.empty(M, N, device="cuda", dtype=BF16)
    H.gemm_fp8(a8, w8, sw, out, bias, H.EPI_BF16, a_scale=sa)
    close(out, acc)
    # the quantised product stays within FP8's error of the bf16-operand product
    full = a.float() @ w.float().t() + bias.float()
    rel = float((out.float() - full).norm() / full.norm())
    assert rel < 5e-2, rel
    # uniform activation scale instead of per-row scales
    H.gemm_fp8(a8, w8, sw, out, None, H.EPI_BF16, a_scale_all=0.5)
    acc_u = ((a8.view(F8).float() * 0.5).double() @ (w8.view(F8).float() * sw[:, None]).double().t()).float()
    close(out, acc_u, ulps=2.01, atol=3e-2)     # |values| ~ 100: the same relative bounds as above
    gel = torch.empty(M, N, device="cuda", dtype=BF16)
    H.gemm_fp8(a8, w8, sw, gel, bias, H.EPI_BF16_GELU, a_scale=sa)
    want_g = torch.nn.functional.gelu(acc.to(BF16).float(), approximate="tanh")
    err = (gel.float() - want_g).abs()
    assert float(err.max()) < 4e-2 and float((err > 2e-3 + 2.0 ** -7 * want_g.abs()).float().mean()) < 1e-3
    # FP8 output = the bf16 GELU output of the same kernel, divided by out_scale and converted
    g8 = torch.empty(M, N, device="cuda", dtype=torch.uint8)
    H.gemm_fp8(a8, w8, sw, g8, bias, H.EPI_FP8_GELU, a_scale=sa, out_scale=0.25)
    assert torch.equal(g8, (gel.float() * 4.0).clamp(-448, 448).to(F8).view(torch.uint8))
    x0 = randn((M, N), 14)
    gate = randn((N,), 15)
    x = x0.clone()
    H.gemm_fp8(a8, w8, sw, x, bias, H.EPI_RESID, a_scale=sa, gate=gate)
    torch.testing.assert_close(x, x0 + acc.to(BF16).float() * gate, rtol=2e-2, atol=2e-2)
    H.gemm_fp8(a8, w8, sw, x, bias, H.EPI_RESID, a_scale=sa, gate=-gate)
    torch.testing.assert_close(x, x0, rtol=0, atol=1e-5)


def test_gemm_fp8_is_repeatable(H):
    """ring staging / fragment-read races would show as run-to-run differences"""
    M, N, K = 8192, 4096, 1536
    a8, sa = H.quant_rows_fp8(randn((M, K), 21).to(BF16))
    w8, sw = H.quant_rows_fp8(randn((N, K), 22, 1 / math.sqrt(K)).to(BF16))
    first = torch.empty(M, N, device="cuda", dtype=BF16)
    H.gemm_fp8(a8, w8, sw, first, None, H.EPI_BF16, a_scale=sa)
    out = torch.empty_like(first)
    for _ in range(20):
        out.fill_(float("nan"))
        H.gemm_fp8(a8, w8, sw, out, None, H.EPI_BF16, a_scale=sa)
        assert torch.equal(out, first)


def test_gemm_fp8_rejects_unsupported_shapes(H):
    a8 = torch.zeros(64, 128, dtype=torch.uint8, device="cuda")
    sw = torch.ones(256, device="cuda")
    with pytest.raises(RuntimeError, match="not supported"):
        H.gemm_fp8(a8, torch.zeros(200, 128, dtype=torch.uint8, device="cuda"), sw[:200],
                   torch.empty(64, 200, device="cuda", dtype=BF16))
    with pytest.raises(RuntimeError, match="not supported"):
        H.gemm_fp8(a8[:, :64].contiguous(), torch.zeros(256, 64, dtype=torch.uint8, device="cuda"), sw,
                   torch.empty(64, 256, device="cuda", dtype=BF16))
    with pytest.raises(ValueError):
        H.gemm_fp8(a8, torch.zeros(256, 128, dtype=torch.uint8, device="cuda"), sw,
                   torch.empty(64, 256, device="cuda", dtype=F32), epilogue=H.EPI_BF16)


# ------------------------------------------------------------------------------ model level

def _tiny_unified():
    from omnivideo.modules.unified_model import UnifiedWanWithMixedConditionModel
    from omnivideo.utils import synthetic as syn
    cfg = syn.dit_config("tiny")
    sd = syn.make_unified_state_dict(cfg, 0)
    m = UnifiedWanWithMixedConditionModel(
        wan_config=dict(cfg), vlm_in_dim=cfg["vlm_in_dim"], use_visual_context_adapter=True,
        visual_context_adapter_patch_size=cfg["visual_patch_size"], max_context_len=cfg["text_len"],
        eps=cfg["eps"], skip_init=True)
    m.load_state_dict(sd, strict=True)
    return cfg, m.to(BF16).cuda().eval()


def rel_l2(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm())


def test_model_forward_fp8_vs_bf16_path():
    """The tolerance of the FP8 mode is defined against this build's own bf16 path (itself checked
    against the oracle and the reference's goldens): relative L2 of one forward; the mode's budget is 6e-2, the asserts sit at three times what was observed."""
    from omnivideo.utils import synthetic as syn
    cfg, m = _tiny_unified()
    wm = m.wan_model
    lat = syn.randn("in.lat", (16, 3, 8, 12), 1).to(BF16).cuda()
    ctx = [syn.randn("in.text", (10, cfg["text_dim"]), 1).to(BF16).cuda()]
    t = torch.tensor([500])
    ref = wm([lat], t, ctx, 72)[0].clone()
    wm.linear_precision = "fp8"
    got = wm([lat], t, ctx, 72)[0].clone()
    assert wm.engine().precision == "fp8" and wm.engine().layers[0]["w_qkv"] is None
    assert torch.isfinite(got).all()
    assert wm.engine().attention == "int8"
    r = rel_l2(got, ref)
    check_rel("tiny forward, fp8 linears + int8 QK^T vs bf16 path", r, 1e-2)              # observed 3.4e-3
    assert r > 1e-4                           # different arithmetic, same function
    wm.attention_precision = "bf16"           # e4m3 linear layers alone
    r_lin = rel_l2(wm([lat], t, ctx, 72)[0], ref)
    check_rel("tiny forward, fp8 linears (bf16 attention) vs bf16 path", r_lin, 1e-2)     # observed 3.4e-3
    wm.attention_precision = "auto"
    assert torch.equal(wm([lat], t, ctx, 72)[0], got)          # deterministic
    wm.linear_precision = "bf16"
    assert torch.equal(wm([lat], t, ctx, 72)[0], ref)          # and switching back restores the bf16 path
    with pytest.raises(ValueError):
        wm.linear_precision = "fp4"
        wm([lat], t, ctx, 72)


def test_full_width_block_fp8_vs_bf16_full_grid():
    """One block of the 1.3B width over the full 20 280-token grid, both guidance branches."""
    from omnivideo.modules.engine import DiTEngine
    from omnivideo.modules.unified_model import UnifiedWanWithMixedConditionModel
    from omnivideo.utils import synthetic as syn
    cfg = syn.dit_config("1.3B", num_layers=2, text_len=512)
    sd = syn.make_unified_state_dict(cfg, 0)
    m = UnifiedWanWithMixedConditionModel(wan_config=dict(cfg), vlm_in_dim=cfg["vlm_in_dim"],
                                          use_visual_context_adapter=True,
                                          visual_context_adapter_patch_size=cfg["visual_patch_size"],
                                          max_context_len=cfg["text_len"], eps=cfg["eps"], skip_init=True)
    m.load_state_dict(sd, strict=True)
    wm = m.to(BF16).cuda().eval().wan_model
    lat = syn.randn("full.lat", (16, 13, 60, 104), 1).to(BF16).cuda()
    ctxs = [syn.randn("full.text", (20, cfg["text_dim"]), 1).to(BF16).cuda(),
            syn.randn("full.null", (126, cfg["text_dim"]), 1).to(BF16).cuda()]
    outs = {}
    for prec in ("bf16", "fp8"):
        eng = DiTEngine(wm, precision=prec)
        eng.plan((13, 30, 52), 20280, 2)
        cache = eng.encode_context(ctxs, wm.text_len)
        out = torch.empty(2, 16, 13, 60, 104, device="cuda")
        eng.forward(lat, torch.tensor([640.0], dtype=torch.float64, device="cuda"), cache, out)
        outs[prec] = out
        del eng
    r = rel_l2(outs["fp8"], outs["bf16"])
    assert torch.isfinite(outs["fp8"]).all()
    check_rel("2 blocks of 1.3B width, full grid, fp8 linears + int8 QK^T vs bf16 path", r, 6e-3)   # observed 1.7e-3


def test_generate_fp8_linear_mode_tracks_bf16_trajectory():
    """A guided 8-step sampling run through generate() with linear_precision="fp8" (graph replay
    included) stays close to the bf16 run from the same noise: the quantisation error does not
    build up over the steps."""
    from omnivideo.configs import Config, t2v_1_3B
    from omnivideo.utils import synthetic as syn
    from omnivideo.x2x_gen_unified_1_3B import OmniVideoX2XUnified1_3B
    cfg = syn.dit_config("tiny")
    sd, sp = syn.make_unified_state_dict(cfg, 0), syn.make_special_tokens(cfg, 0)
    text = syn.randn("in.text", (10, cfg["text_dim"]), 1).to(BF16)
    text_null = syn.randn("in.text_null", (6, cfg["text_dim"]), 1).to(BF16)
    visual = syn.randn("in.visual", (1, 16, 3, 8, 12), 1).cuda()
    noise = syn.randn("gen.noise", (16, 3, 8, 12), 11).to(BF16)
    lats = {}
    for prec in ("bf16", "fp8"):
        pipe = OmniVideoX2XUnified1_3B(Config(t2v_1_3B), None, vlm_in_dim=cfg["vlm_in_dim"],
                                       use_visual_context_adapter=True,
                                       visual_context_adapter_patch_size=cfg["visual_patch_size"],
                                       max_context_len=cfg["text_len"], wan_config=dict(cfg), linear_precision=prec)
        pipe.model.load_state_dict(sd, strict=True)
        pipe.model.to(BF16)
        pipe.special_tokens = {k: v.cuda().to(BF16) for k, v in sp.items()}
        assert pipe.model.wan_model.linear_precision == prec
        lats[prec] = pipe.generate("", precomputed_context=text, precomputed_context_null=text_null, visual_emb=visual,
                                   size=(96, 64), frame_num=9, shift=5.0, sampling_steps=8, guide_scale=3.0, seed=1,
                                   noise=noise, return_latents=True).float()
    r = rel_l2(lats["fp8"], lats["bf16"])
    assert torch.isfinite(lats["fp8"]).all() and r > 1e-4
    check_rel("8-step generate(), fp8 linears + int8 QK^T vs bf16 run", r, 1.5e-2)   # observed 4.8e-3
