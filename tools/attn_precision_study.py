#!/usr/bin/env python
"""Emulation study (torch on the GPU, no product code): how far does attention move when Q K^T runs on
8-bit operands?  Inputs are made the way the DiT makes them (projection, full-width RMSNorm with weight,
RoPE), scores and P V are then evaluated with Q/K rounded to bf16 (what the kernel does today), to
e4m3 with a scale per token and head, and to int8 with a scale per token and head for Q and per 64-key
tile and head for K (with and without subtracting the per-channel key mean first, which leaves the
softmax unchanged).  P and V stay bf16.  Prints the relative L2 distance of each variant's output to
exact fp32 attention.  Development aid for the reduced-precision mode of the largest model."""
import math
import sys

import torch

F8 = torch.float8_e4m3fn


def rope(x, L):
    D = x.shape[-1]
    pos = torch.arange(L, device=x.device, dtype=torch.float64)[:, None]
    freq = 1.0 / torch.pow(10000.0, torch.arange(0, D, 2, device=x.device, dtype=torch.float64) / D)
    ang = pos * freq
    a, b = x[..., 0::2].double(), x[..., 1::2].double()
    c, s = ang.cos()[:, None, :], ang.sin()[:, None, :]
    return torch.stack([a * c - b * s, a * s + b * c], -1).flatten(-2).float()


def attend(q, k, v, p8=False):
    """p8: the probabilities (relative to the row maximum, as a flash kernel holds them) rounded to e4m3
    before P V instead of to bf16; the row sum stays in fp32."""
    s = torch.einsum("qnd,knd->nqk", q, k) / math.sqrt(q.shape[-1])
    if not p8:
        return torch.einsum("nqk,knd->qnd", torch.softmax(s, -1).to(torch.bfloat16).float(), v)
    e = torch.exp(s - s.amax(-1, keepdim=True))
    return torch.einsum("nqk,knd->qnd", e.to(F8).float(), v) / e.sum(-1).t()[:, :, None]


def q_chan(v):               # e4m3 with one scale per (head, channel) over the whole sequence
    s = v.abs().amax(0, keepdim=True).clamp_min(1e-12) / 448.0
    return (v / s).to(F8).float() * s


def q_tok(x, kind):          # one scale per (token, head)
    mx = x.abs().amax(-1, keepdim=True).clamp_min(1e-12)
    if kind == "e4m3":
        s = mx / 448.0
        return (x / s).to(F8).float() * s
    s = mx / 127.0
    return (x / s).round().clamp(-127, 127) * s


def q_tile(x, tile=64):      # int8, one scale per (64-key tile, head)
    L, N, D = x.shape
    pad = (-L) % tile
    xp = torch.cat([x, x.new_zeros(pad, N, D)]).view(-1, tile, N, D)
    s = xp.abs().amax((1, 3), keepdim=True).clamp_min(1e-12) / 127.0
    return ((xp / s).round().clamp(-127, 127) * s).view(-1, N, D)[:L]


def main(L=4096, C=1536, N=12, sharp=1.0, outlier=0.0):
    dev, D = "cuda", 128
    g = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)
    h = r(L, C)
    wq, wk, wv = (r(C, C) / math.sqrt(C) for _ in range(3))
    nq, nk = 1 + 0.2 * r(C), 1 + 0.2 * r(C)
    bias_k = outlier * r(C) * (r(C) > 1.5)                 # a few channels with a large constant offset
    rms = lambda x: x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6)
    q = rope((rms(h @ wq.t()) * nq * sharp).view(L, N, D), L)
    k = rope((rms(h @ wk.t() + bias_k) * nk * sharp).view(L, N, D), L)
    v = (h @ wv.t()).view(L, N, D).to(torch.bfloat16).float()
    ref = attend(q, k, v)
    rel = lambda o: float((o - ref).norm() / ref.norm())
    bf = lambda x: x.to(torch.bfloat16).float()
    km = k - k.mean(0, keepdim=True)
    rows = [("bf16 Q,K (today)", rel(attend(bf(q), bf(k), v))),
            ("e4m3 Q,K per token", rel(attend(q_tok(q, "e4m3"), q_tok(k, "e4m3"), v))),
            ("int8 Q per token, K per token", rel(attend(q_tok(q, "i8"), q_tok(k, "i8"), v))),
            ("int8 Q per token, K per 64-key tile", rel(attend(q_tok(q, "i8"), q_tile(k), v))),
            ("int8, K mean removed, per 64-key tile", rel(attend(q_tok(q, "i8"), q_tile(km), v))),
            ("int8 Q K^T + e4m3 P, bf16 V", rel(attend(q_tok(q, "i8"), q_tile(k), v, p8=True))),
            ("int8 Q K^T + e4m3 P and V (per channel)", rel(attend(q_tok(q, "i8"), q_tile(k), q_chan(v), p8=True)))]
    print(f"L={L} C={C} heads={N} logit scale x{sharp} key-channel offsets x{outlier}")
    for name, e in rows:
        print(f"  {name:42s} rel L2 {e:.2e}")


if __name__ == "__main__":
    main()
    main(sharp=2.0)
    main(outlier=4.0)
    main(C=5120, N=40, L=2048)
