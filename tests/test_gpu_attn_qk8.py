"""Self-attention with Q K^T on int8 operands (the attention half of the reduced-precision mode of the
largest model; the reference has no such path).  The quantiser is pinned bit for bit to its definition;
the kernel is pinned to exact softmax attention of the DEQUANTISED q and k (what it is meant to compute:
bf16-kernel tolerance), and its distance to attention on the original bf16 q and k is measured and
bounded (the cost of the 8-bit operands)."""
import pytest
import torch

from conftest import check_rel
from oracle import dit as O

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32


@pytest.fixture(scope="module")
def H():
    import omnivideo._hip as H
    H.lib()
    return H


def rnd(shape, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(*shape, device="cuda", generator=g) * scale).to(BF16)


def quantise(H, x, rps):
    B, L, N, D = x.shape
    q = torch.empty(B, L, N, D, device="cuda", dtype=torch.int8)
    s = torch.empty(B, L if rps == 1 else -(-L // rps), N, device="cuda")
    H.quant_heads_i8(x, q, s, rps)
    return q, s


def ref_quant(x, rps):
    """the definition, with torch on the CPU: block absmax / 127, rint(x * (127 / absmax))"""
    xf = x.float().cpu()
    B, L, N, D = xf.shape
    pad = (-L) % rps
    xp = torch.cat([xf, xf.new_zeros(B, pad, N, D)], 1).view(B, -1, rps, N, D)
    mx = xp.abs().amax((2, 4))                                            # [B, blocks, N]
    scale = torch.where(mx > 0, mx / 127.0, torch.ones_like(mx))
    # tensor / tensor: IEEE division (scalar / tensor goes through a rounded reciprocal in torch)
    inv = torch.where(mx > 0, torch.full_like(mx, 127.0) / mx, torch.zeros_like(mx))
    q = (xp * inv[:, :, None, :, None]).round().clamp(-127, 127).to(torch.int8).view(B, -1, N, D)[:, :L]
    return q, scale


@pytest.mark.parametrize("B,L,N,rps", [(2, 100, 3, 1), (2, 100, 3, 64), (1, 64, 12, 64), (1, 777, 5, 1), (1, 777, 5, 64)])
def test_quant_heads_matches_definition(H, B, L, N, rps):
    buf = rnd((B, L, 3 * N * 128), 1, 2.0)
    x = buf[:, :, N * 128:2 * N * 128].unflatten(2, (N, 128))             # the k third of a fused projection buffer
    x[0, 3] = 0
    q, s = quantise(H, x, rps)
    qr, sr = ref_quant(x, rps)
    assert torch.equal(s.cpu(), sr)
    assert torch.equal(q.cpu(), qr)


@pytest.mark.parametrize("B,L,N,k_len", [(1, 256, 2, 256), (2, 333, 3, 300), (1, 1500, 2, 1437), (2, 20280, 12, 20280)])
def test_qk8_attention_vs_oracle_on_dequantised_operands(H, B, L, N, k_len):
    q, k, v = rnd((B, L, N, 128), 11), rnd((B, L, N, 128), 12), rnd((B, L, N, 128), 13)
    q8, qs = quantise(H, q, 1)
    k8, ks = quantise(H, k, 64)
    ws = torch.empty(max(H.flash_attn_workspace_bytes(B, L, N, k_len), 1), dtype=torch.uint8, device="cuda")
    out = torch.full((B, L, N, 128), float("nan"), device="cuda", dtype=BF16)
    H.flash_attn_qk8(q8, qs, k8, ks, v, out, k_len=k_len, workspace=ws)
    assert torch.isfinite(out.float()).all()
    rows = torch.cat([torch.tensor([0, 31, 32, L - 1]),
                      torch.randint(0, L, (44,), generator=torch.Generator().manual_seed(3))]) if L > 400 else torch.arange(L)
    ksx = ks.repeat_interleave(64, dim=1)[:, :L]                          # tile scale of every key
    for b in range(B):
        qd = q8[b, rows].float().cpu() * qs[b, rows].cpu()[..., None]
        kd = k8[b].float().cpu() * ksx[b].cpu()[..., None]
        ref = O.attention(qd, kd, v[b].float().cpu(), k_len=k_len, flow="fp32")
        got = out[b, rows].float().cpu()
        err = (got - ref).abs()
        assert (err <= 6e-3 + 2.0 * 2.0 ** -8 * ref.abs()).all(), float(err.max())
    # against the bf16 attention kernel on the original operands: the price of 8-bit Q and K
    base = torch.empty_like(out)
    H.flash_attn(q, k, v, base, k_len=k_len)
    rel = float((out.float() - base.float()).norm() / base.float().norm())
    check_rel(f"qk8 attention vs bf16 attention, L={L} N={N}", rel, 3.5e-2)      # observed 1.2e-2 on unit-variance operands
    # determinism, and the plain grid (no workspace) differs by summation order at most
    out2 = torch.empty_like(out)
    H.flash_attn_qk8(q8, qs, k8, ks, v, out2, k_len=k_len, workspace=ws)
    assert torch.equal(out, out2)
    H.flash_attn_qk8(q8, qs, k8, ks, v, out2, k_len=k_len)
    assert float((out2.float() - out.float()).abs().max()) < 2e-2


def test_qk8_attention_rejects_bad_arguments(H):
    q, k, v = rnd((1, 128, 2, 128), 1), rnd((1, 128, 2, 128), 2), rnd((1, 128, 2, 128), 3)
    q8, qs = quantise(H, q, 1)
    k8, ks = quantise(H, k, 64)
    out = torch.empty_like(v)
    with pytest.raises(RuntimeError, match="k_len"):
        H.flash_attn_qk8(q8, qs, k8, ks, v, out, k_len=0)
    with pytest.raises(ValueError, match="scales"):
        H.flash_attn_qk8(q8, ks, k8, ks, v, out)
