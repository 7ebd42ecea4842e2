"""TEST INFRASTRUCTURE ONLY - CPU oracle for the Omni-Video 2 DiT forward.

A from-scratch restatement, as stateless functions over a flat name->tensor weight
dict, of the reference's mixed-condition DiT.  It is the checker for the HIP path
(tests/, __graft_entry__.smoke, bench.py's cpu_baseline leg); the product package
never imports it.  It is pinned to the real reference by tests/golden/*.npz, which
oracle/gen_golden.py produced by running the reference's own modules on CPU.

Two precision flows:
  * ``flow="fp32"``  - everything in float32 (float64 for the sinusoid/RoPE tables),
    the reference's behaviour on CPU, where ``torch.amp.autocast('cuda', ...)`` is
    inert.  This is what the golden vectors pin.
  * ``flow="bf16"``  - same math with a round-to-bfloat16 inserted at every point
    where the reference's GPU path (bf16 params under ``autocast('cuda', bf16)``,
    reference: omnivideo/x2x_gen_unified.py:401-403) materialises a bf16 tensor.
    Values stay in float32 containers; products of bf16-representable numbers are
    exact in float32, so this models bf16-in / fp32-accumulate matrix units.

Each function cites the reference lines it restates.
"""
import math

import torch
import torch.nn.functional as F


def rb(x, flow):
    """Round to bfloat16 (kept in a float32 container) when emulating the bf16 flow."""
    if flow == "bf16":
        return x.to(torch.bfloat16).to(torch.float32)
    return x


def round_weights_bf16(sd):
    """bf16 params (reference casts the whole module: x2x_gen_unified.py:126)."""
    return {k: v.to(torch.bfloat16).to(torch.float32) for k, v in sd.items()}


# --------------------------------------------------------------------------- tables

def time_sinusoid(t, dim=256):
    """[cos | sin] of t * 10000^(-i/half), float64 (model.py:13-23)."""
    half = dim // 2
    pos = torch.as_tensor(t, dtype=torch.float64).reshape(-1)
    inv = torch.pow(10000.0, -torch.arange(half, dtype=torch.float64) / half)
    ang = pos[:, None] * inv[None, :]
    return torch.cat([ang.cos(), ang.sin()], dim=1)


def rope_angles(head_dim, grid, max_pos=1024, theta=10000.0):
    """Rotation angle per (token, complex pair): [f*h*w, head_dim/2] float64.

    The head's head_dim/2 complex pairs are split (frame, height, width) as
    (c - 2*(c//3), c//3, c//3) with c = head_dim/2, each axis using its own
    frequency ladder 1/theta^(2i/dim_axis) (model.py:26-34,42,52-57,404-411)."""
    f, h, w = grid
    c = head_dim // 2
    parts = (c - 2 * (c // 3), c // 3, c // 3)
    assert max(f, h, w) <= max_pos

    def ladder(npairs):
        dim_axis = 2 * npairs
        return 1.0 / torch.pow(theta, torch.arange(0, dim_axis, 2, dtype=torch.float64) / dim_axis)

    pf = torch.arange(f, dtype=torch.float64)[:, None] * ladder(parts[0])[None, :]
    ph = torch.arange(h, dtype=torch.float64)[:, None] * ladder(parts[1])[None, :]
    pw = torch.arange(w, dtype=torch.float64)[:, None] * ladder(parts[2])[None, :]
    ang = torch.cat([
        pf[:, None, None, :].expand(f, h, w, -1),
        ph[None, :, None, :].expand(f, h, w, -1),
        pw[None, None, :, :].expand(f, h, w, -1)], dim=-1)
    return ang.reshape(f * h * w, c)


def rope_rotate(x, angles):
    """x: [L, N, D]; rotates interleaved pairs (x[2i], x[2i+1]) of the first
    len(angles) tokens in float64, passes the rest through, returns float32
    (model.py:37-65)."""
    n_tok = angles.shape[0]
    L, N, D = x.shape
    xr = x[:n_tok].to(torch.float64).reshape(n_tok, N, D // 2, 2)
    c, s = angles.cos()[:, None, :], angles.sin()[:, None, :]
    a, b = xr[..., 0], xr[..., 1]
    out = torch.stack([a * c - b * s, a * s + b * c], dim=-1).reshape(n_tok, N, D)
    return torch.cat([out, x[n_tok:].to(torch.float64)], dim=0).float()


# --------------------------------------------------------------------------- norms

def rms_norm(x, weight, eps, flow):
    """fp32 normalise over the last dim, cast back to x's dtype, times weight
    (model.py:76-84).  With bf16 activations the normalised value is rounded
    before the multiply and the product is rounded again."""
    y = x * torch.rsqrt(x.pow(2).mean(dim=-1, keepdim=True) + eps)
    return rb(rb(y, flow) * weight, flow)


def layer_norm(x, eps, weight=None, bias=None):
    """fp32 LayerNorm, biased variance (model.py:87-97)."""
    return F.layer_norm(x, (x.shape[-1],), weight, bias, eps)


def gelu_tanh(x):
    return F.gelu(x, approximate="tanh")


def linear(x, sd, prefix, flow):
    """bf16-autocast Linear: inputs rounded, fp32 accumulate, output rounded."""
    return rb(F.linear(rb(x, flow), sd[prefix + ".weight"], sd[prefix + ".bias"]), flow)


# ----------------------------------------------------------------------- attention

def attention(q, k, v, k_len=None, flow="fp32"):
    """softmax(q k^T / sqrt(D)) v per head; q [Lq,N,D], k/v [Lk,N,D]; keys at
    index >= k_len are masked (attention.py:24-133 semantics with B=1).  In the
    bf16 flow inputs and output are rounded (the flash kernel takes and returns
    half precision: attention.py:59-83)."""
    q, k, v = rb(q, flow), rb(k, flow), rb(v, flow)
    D = q.shape[-1]
    if k_len is not None and k_len < k.shape[0]:
        k, v = k[:k_len], v[:k_len]          # masked keys get zero weight: same as dropping them
    out = torch.empty(q.shape[0], q.shape[1], v.shape[-1])
    kt = k.permute(1, 2, 0)                   # [N, D, Lk]
    vt = v.permute(1, 0, 2)                   # [N, Lk, D]
    qchunk = max(1, (1 << 27) // max(1, k.shape[0] * q.shape[1]))   # bound the score block to ~0.5 GB
    for q0 in range(0, q.shape[0], qchunk):
        s = torch.matmul(q[q0:q0 + qchunk].permute(1, 0, 2), kt) / math.sqrt(D)
        p = torch.softmax(s, dim=-1)
        out[q0:q0 + qchunk] = torch.matmul(p, vt).permute(1, 0, 2)
    return rb(out, flow)


def self_attention(h, sd, pre, num_heads, eps, angles, n_valid, flow):
    """model.py:125-154: q/k/v Linear, full-width RMSNorm on q and k, RoPE,
    attention with keys limited to the n_valid real tokens, o Linear."""
    L, C = h.shape
    D = C // num_heads
    q = rms_norm(linear(h, sd, pre + ".q", flow), sd[pre + ".norm_q.weight"], eps, flow)
    k = rms_norm(linear(h, sd, pre + ".k", flow), sd[pre + ".norm_k.weight"], eps, flow)
    v = linear(h, sd, pre + ".v", flow)
    q = rope_rotate(q.view(L, num_heads, D), angles)
    k = rope_rotate(k.view(L, num_heads, D), angles)
    a = attention(q, k, v.view(L, num_heads, D), k_len=n_valid, flow=flow)
    return linear(a.reshape(L, C), sd, pre + ".o", flow)


def cross_kv(ctx, sd, pre, eps, flow):
    """K (normalised) and V of the embedded context (model.py:170-171).  They do
    not depend on the latents or the timestep."""
    k = rms_norm(linear(ctx, sd, pre + ".k", flow), sd[pre + ".norm_k.weight"], eps, flow)
    v = linear(ctx, sd, pre + ".v", flow)
    return k, v


def cross_attention(h, kv, sd, pre, num_heads, eps, flow):
    """model.py:159-179 with context_lens=None: every padded key is attended."""
    L, C = h.shape
    D = C // num_heads
    k, v = kv
    q = rms_norm(linear(h, sd, pre + ".q", flow), sd[pre + ".norm_q.weight"], eps, flow)
    a = attention(q.view(L, num_heads, D), k.view(-1, num_heads, D),
                  v.view(-1, num_heads, D), k_len=None, flow=flow)
    return linear(a.reshape(L, C), sd, pre + ".o", flow)


# --------------------------------------------------------------------------- blocks

def block_forward(x, e0, ctx_kv, sd, pre, cfg, angles, n_valid, flow, x_is_bf16=False):
    """One WanAttentionBlock (model.py:218-258).  x [L,C] fp32 residual stream,
    e0 [6,C] fp32 (one row: the timestep is the same for every token)."""
    eps, N = cfg["eps"], cfg["num_heads"]
    e = sd[pre + ".modulation"].reshape(6, -1) + e0
    n1 = layer_norm(x, eps)
    if x_is_bf16:           # block 0 sees the bf16 patch embedding: norm output is cast back
        n1 = rb(n1, flow)
    y = self_attention(n1 * (1 + e[1]) + e[0], sd, pre + ".self_attn", N, eps, angles,
                       n_valid, flow)
    x = x + y * e[2]
    n3 = layer_norm(x, eps, sd[pre + ".norm3.weight"], sd[pre + ".norm3.bias"])
    x = x + cross_attention(n3, ctx_kv, sd, pre + ".cross_attn", N, eps, flow)
    h = layer_norm(x, eps) * (1 + e[4]) + e[3]
    h = rb(gelu_tanh(linear(h, sd, pre + ".ffn.0", flow)), flow)
    y = linear(h, sd, pre + ".ffn.2", flow)
    return x + y * e[5]


def patchify(lat, patch):
    """[Cin,T,H,W] -> [f*h*w, Cin*pt*ph*pw] rows in (f,h,w) order, columns in the
    Conv3d weight's (cin, pt, ph, pw) order (model.py:455-458)."""
    cin, T, H, W = lat.shape
    pt, ph, pw = patch
    f, h, w = T // pt, H // ph, W // pw
    u = lat[:, :f * pt, :h * ph, :w * pw].reshape(cin, f, pt, h, ph, w, pw)
    return u.permute(1, 3, 5, 0, 2, 4, 6).reshape(f * h * w, cin * pt * ph * pw), (f, h, w)


def patch_embed(lat, weight, bias, patch, flow):
    rows, grid = patchify(lat, patch)
    return rb(F.linear(rb(rows, flow), weight.reshape(weight.shape[0], -1), bias), flow), grid


def time_embed(t, sd, cfg, pre="wan_model."):
    """e [C] and e0 [6,C], fp32 end to end (model.py:467-476)."""
    s = time_sinusoid(t, cfg["freq_dim"]).float()[0]
    e = F.linear(s, sd[pre + "time_embedding.0.weight"], sd[pre + "time_embedding.0.bias"])
    e = F.linear(F.silu(e), sd[pre + "time_embedding.2.weight"], sd[pre + "time_embedding.2.bias"])
    e0 = F.linear(F.silu(e), sd[pre + "time_projection.1.weight"], sd[pre + "time_projection.1.bias"])
    return e, e0.reshape(6, -1)


def text_embed(ctx, sd, cfg, flow, pre="wan_model."):
    """Zero-pad the mixed context to text_len rows, then Linear-GELU-Linear
    (model.py:480-485).  Padded rows become bias-only features, not zeros."""
    n = ctx.shape[0]
    assert n <= cfg["text_len"]
    padded = torch.cat([ctx, ctx.new_zeros(cfg["text_len"] - n, ctx.shape[1])])
    h = rb(gelu_tanh(linear(padded, sd, pre + "text_embedding.0", flow)), flow)
    return linear(h, sd, pre + "text_embedding.2", flow)


def head_unpatchify(x, e, sd, cfg, grid, pre="wan_model."):
    """fp32 LN + 2-way modulation + Linear(C -> out*pt*ph*pw), then the
    'fhwpqrc->cfphqwr' scatter back to [Cout, F, H, W] (model.py:278-290,526-549)."""
    m = sd[pre + "head.modulation"].reshape(2, -1) + e[None, :]
    y = layer_norm(x, cfg["eps"]) * (1 + m[1]) + m[0]
    y = F.linear(y, sd[pre + "head.head.weight"], sd[pre + "head.head.bias"])
    f, h, w = grid
    pt, ph, pw = cfg["patch_size"]
    c = cfg["out_dim"]
    u = y[:f * h * w].reshape(f, h, w, pt, ph, pw, c)
    return u.permute(6, 0, 3, 1, 4, 2, 5).reshape(c, f * pt, h * ph, w * pw)


def wan_forward(lat, t, ctx, seq_len, sd, cfg, flow="fp32", num_layers=None, pre="wan_model."):
    """WanModel.forward for one sample (model.py:417-524): lat [Cin,T,H,W], scalar
    timestep t, mixed context ctx [n, text_dim] -> [Cout,T,H,W] float32."""
    x, grid = patch_embed(rb(lat, flow), sd[pre + "patch_embedding.weight"],
                          sd[pre + "patch_embedding.bias"], cfg["patch_size"], flow)
    n_valid = x.shape[0]
    assert n_valid <= seq_len
    x = torch.cat([x, x.new_zeros(seq_len - n_valid, x.shape[1])])
    e, e0 = time_embed(t, sd, cfg, pre)
    c = text_embed(rb(ctx, flow), sd, cfg, flow, pre)
    angles = rope_angles(cfg["dim"] // cfg["num_heads"], grid)
    nl = cfg["num_layers"] if num_layers is None else num_layers
    for i in range(nl):
        bp = f"{pre}blocks.{i}"
        kv = cross_kv(c, sd, bp + ".cross_attn", cfg["eps"], flow)
        x = block_forward(x, e0, kv, sd, bp, cfg, angles, n_valid, flow,
                          x_is_bf16=(i == 0 and flow == "bf16"))
    return head_unpatchify(x, e, sd, cfg, grid, pre)


def wan_forward_rows(lat, t, ctx, seq_len, sd, cfg, rows, flow="fp32", pre="wan_model."):
    """wan_forward for a ONE-layer model, evaluated only at the token rows `rows`: returns the head
    output [len(rows), pt*ph*pw*Cout] of those tokens (column order (p, q, r, c) as in
    head_unpatchify), i.e. the latent patch each token writes.  Every operator of the forward is
    row-local except self-attention, whose keys and values are needed for all tokens; with one
    layer those come straight from the patch embedding, so the cost is one K/V projection over the
    sequence plus |rows| rows of everything else.  Used where a whole forward is out of reach for the
    CPU (75 600 tokens at width 5120).  Same arithmetic, line by line, as wan_forward."""
    assert cfg["num_layers"] == 1
    eps, N = cfg["eps"], cfg["num_heads"]
    rows = torch.as_tensor(rows, dtype=torch.long)
    x, grid = patch_embed(rb(lat, flow), sd[pre + "patch_embedding.weight"],
                          sd[pre + "patch_embedding.bias"], cfg["patch_size"], flow)
    n_valid = x.shape[0]
    x = torch.cat([x, x.new_zeros(seq_len - n_valid, x.shape[1])])
    e, e0 = time_embed(t, sd, cfg, pre)
    c = text_embed(rb(ctx, flow), sd, cfg, flow, pre)
    angles = rope_angles(cfg["dim"] // cfg["num_heads"], grid)
    bp = f"{pre}blocks.0"
    D = cfg["dim"] // N
    mod = sd[bp + ".modulation"].reshape(6, -1) + e0
    n1 = layer_norm(x, eps)
    if flow == "bf16":
        n1 = rb(n1, flow)
    h = n1 * (1 + mod[1]) + mod[0]
    sp = bp + ".self_attn"
    k = rope_rotate(rms_norm(linear(h, sd, sp + ".k", flow), sd[sp + ".norm_k.weight"], eps, flow).view(-1, N, D), angles)
    v = linear(h, sd, sp + ".v", flow).view(-1, N, D)
    q = rms_norm(linear(h[rows], sd, sp + ".q", flow), sd[sp + ".norm_q.weight"], eps, flow).view(-1, N, D)
    # RoPE of the sampled rows: rope_rotate rotates row i by angles[i], so hand it the rows' own angles
    ang_r = torch.zeros(len(rows), angles.shape[1], dtype=angles.dtype)
    real = rows < angles.shape[0]
    ang_r[real] = angles[rows[real]]
    q = rope_rotate(q, ang_r)                 # a zero angle is the identity: padded rows stay unrotated
    a = attention(q, k, v, k_len=n_valid, flow=flow)
    xr = x[rows] + linear(a.reshape(len(rows), -1), sd, sp + ".o", flow) * mod[2]
    n3 = layer_norm(xr, eps, sd[bp + ".norm3.weight"], sd[bp + ".norm3.bias"])
    kv = cross_kv(c, sd, bp + ".cross_attn", eps, flow)
    xr = xr + cross_attention(n3, kv, sd, bp + ".cross_attn", N, eps, flow)
    hh = layer_norm(xr, eps) * (1 + mod[4]) + mod[3]
    hh = rb(gelu_tanh(linear(hh, sd, bp + ".ffn.0", flow)), flow)
    xr = xr + linear(hh, sd, bp + ".ffn.2", flow) * mod[5]
    m = sd[pre + "head.modulation"].reshape(2, -1) + e[None, :]
    y = layer_norm(xr, eps) * (1 + m[1]) + m[0]
    return F.linear(y, sd[pre + "head.head.weight"], sd[pre + "head.head.bias"])


def gather_token_patches(out, rows, grid, patch):
    """The [len(rows), pt*ph*pw*Cout] patches that tokens `rows` own in an unpatchified
    [Cout, F, H, W] output (inverse of the scatter in head_unpatchify)."""
    f, h, w = grid
    pt, ph, pw = patch
    c = out.shape[0]
    u = out.reshape(c, f, pt, h, ph, w, pw).permute(1, 3, 5, 2, 4, 6, 0).reshape(f * h * w, pt * ph * pw * c)
    return u[torch.as_tensor(rows, dtype=torch.long)]


# ---------------------------------------------------------------- condition builder

def vlm_project(ar, sd, cfg, flow):
    """RMSNorm(vlm_dim) + Linear(vlm_dim -> text_dim) (unified_model.py:79-80,201-202)."""
    return linear(rms_norm(rb(ar, flow), sd["vlm_norm.weight"], cfg["eps"], flow),
                  sd, "vlm_proj", flow)


def visual_adapter(vis, sd, cfg, flow, pre="visual_context_adapter."):
    """Conv3d patchify (1,4,4) -> LayerNorm -> Linear -> LayerNorm, no affine
    (visual_context_adapter.py:52-59,101-114).  vis [Cin,T,h,w] -> [n, text_dim]."""
    rows, _ = patchify(rb(vis, flow), cfg["visual_patch_size"])
    w = sd[pre + "patch_embedding.weight"]
    x = rb(F.linear(rb(rows, flow), w.reshape(w.shape[0], -1), sd[pre + "patch_embedding.bias"]), flow)
    x = layer_norm(x, cfg["eps"])
    x = linear(x, sd, pre + "projection", flow)
    return layer_norm(x, cfg["eps"])


def mixed_context(sd, cfg, flow, text=None, ar_vision=None, visual=None, special=None,
                  max_context_len=None):
    """Tight concatenation [VLM | <prp_st> text <prp_ed> | <img_st> visual <img_ed>],
    truncated to max_context_len (unified_model.py:195-276)."""
    parts = []
    if ar_vision is not None:
        parts.append(vlm_project(ar_vision.reshape(-1, ar_vision.shape[-1]), sd, cfg, flow))
    if text is not None:
        t2 = text.reshape(-1, text.shape[-1])
        parts += [special["<prp_st>"], t2, special["<prp_ed>"]] if special else [t2]
    if visual is not None:
        v = visual_adapter(visual.reshape(visual.shape[-4:]), sd, cfg, flow)
        parts += [special["<img_st>"], v, special["<img_ed>"]] if special else [v]
    if not parts:
        return torch.zeros(1, cfg["text_dim"])
    ctx = torch.cat([p.reshape(-1, cfg["text_dim"]).float() for p in parts], dim=0)
    lim = cfg["text_len"] if max_context_len is None else max_context_len
    return ctx[:lim]


def unified_forward(lat, t, sd, cfg, seq_len, flow="fp32", text=None, ar_vision=None,
                    visual=None, special=None, num_layers=None):
    """UnifiedWanWithMixedConditionModel.forward for one sample (unified_model.py:138-278)."""
    ctx = mixed_context(sd, cfg, flow, text, ar_vision, visual, special, cfg["text_len"])
    return wan_forward(lat, t, ctx, seq_len, sd, cfg, flow, num_layers)
