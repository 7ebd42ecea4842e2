"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's umT5 encoder
(omnivideo/modules/t5.py: T5Encoder with shared_pos=False, as built by umt5_xxl(encoder_only=True),
t5.py:455-468) in two dtype flows: "fp32", and "bf16" = the reference's GPU setting where the module
and every activation are bf16 (model.to(bf16), t5.py:440) and each tensor op rounds its result.

Pinned by tests/golden/t5_tiny.npz (outputs of the reference module itself, oracle/gen_golden.py).
Only tests, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import torch
import torch.nn.functional as F


def rb(x, flow):
    return x.to(torch.bfloat16).to(torch.float32) if flow == "bf16" else x


def relative_position_bucket(lq, lk, num_buckets=32, max_dist=128):
    """Bidirectional bucket index [lq, lk] (t5.py:232-263)."""
    rel_pos = torch.arange(lk).unsqueeze(0) - torch.arange(lq).unsqueeze(1)
    nb = num_buckets // 2
    rel_buckets = (rel_pos > 0).long() * nb
    rel_pos = torch.abs(rel_pos)
    max_exact = nb // 2
    large = max_exact + (torch.log(rel_pos.float() / max_exact) / math.log(max_dist / max_exact) *
                         (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return rel_buckets + torch.where(rel_pos < max_exact, rel_pos, large)


def t5_norm(x, w, eps, flow):
    """t5.py:60-65: fp32 normalise, cast to the weight's dtype, times weight."""
    y = x * torch.rsqrt(x.pow(2).mean(dim=-1, keepdim=True) + eps)
    return rb(w * rb(y, flow), flow)


def t5_gelu(x, flow):
    """t5.py:47-49 evaluated op by op (every intermediate is a tensor of the activation dtype)."""
    r = lambda t: rb(t, flow)
    # torch.pow(x, 3.0) on a bf16 tensor is base * base * base in bf16 arithmetic: two roundings
    cube = r(r(x * x) * x) if flow == "bf16" else torch.pow(x, 3.0)
    inner = r(x + r(0.044715 * cube))
    t = torch.tanh(r(math.sqrt(2.0 / math.pi) * inner))
    return r(r(0.5 * x) * r(1.0 + r(t)))


def t5_attention(h, sd, p, num_heads, bucket, mask, flow):
    """t5.py:85-119 for one sample: h [L, C], mask [L] (1 = attend)."""
    r = lambda t: rb(t, flow)
    L = h.shape[0]
    q = r(F.linear(h, sd[p + "attn.q.weight"])).view(L, num_heads, -1)
    k = r(F.linear(h, sd[p + "attn.k.weight"])).view(L, num_heads, -1)
    v = r(F.linear(h, sd[p + "attn.v.weight"])).view(L, num_heads, -1)
    bias = sd[p + "pos_embedding.embedding.weight"][bucket].permute(2, 0, 1)      # [N, L, L]
    bias = bias.clone()
    if mask is not None:
        lowest = torch.finfo(torch.bfloat16 if flow == "bf16" else torch.float32).min
        bias.masked_fill_(mask.view(1, 1, L) == 0, lowest)
    s = r(r(torch.einsum("inc,jnc->nij", q, k)) + bias)
    a = r(torch.softmax(s, dim=-1))
    o = r(torch.einsum("nij,jnc->inc", a, v)).reshape(L, -1)
    return r(F.linear(o, sd[p + "attn.o.weight"]))


def t5_ffn(h, sd, p, flow):
    """t5.py:135-140: fc2(fc1(x) * GELU(gate(x)))."""
    r = lambda t: rb(t, flow)
    g = t5_gelu(r(F.linear(h, sd[p + "ffn.gate.0.weight"])), flow)
    return r(F.linear(r(r(F.linear(h, sd[p + "ffn.fc1.weight"])) * g), sd[p + "ffn.fc2.weight"]))


def round_weights_bf16(sd):
    return {k: v.to(torch.bfloat16).to(torch.float32) for k, v in sd.items()}


def encode(ids, mask, sd, cfg, flow="fp32", eps=1e-6):
    """T5Encoder.forward (t5.py:302-311) for ids [B, L], mask [B, L] -> [B, L, dim] float32.  In the
    bf16 flow pass weights already rounded with round_weights_bf16."""
    r = lambda t: rb(t, flow)
    B, L = ids.shape
    bucket = relative_position_bucket(L, L, cfg["num_buckets"])
    out = []
    for b in range(B):
        x = sd["token_embedding.weight"][ids[b]]
        m = None if mask is None else mask[b]
        for i in range(cfg["num_layers"]):
            p = f"blocks.{i}."
            x = r(x + t5_attention(t5_norm(x, sd[p + "norm1.weight"], eps, flow), sd, p, cfg["num_heads"], bucket, m,
                                   flow))
            x = r(x + t5_ffn(t5_norm(x, sd[p + "norm2.weight"], eps, flow), sd, p, flow))
        out.append(t5_norm(x, sd["norm.weight"], eps, flow))
    return torch.stack(out)
