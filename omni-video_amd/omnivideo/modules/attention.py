"""`flash_attention` / `attention` with the reference's signature (reference:
omnivideo/modules/attention.py:24-182), backed by ovd_flash_attn_fwd_bf16."""
import torch

from .. import _hip as H

__all__ = ["flash_attention", "attention"]


def flash_attention(q, k, v, q_lens=None, k_lens=None, dropout_p=0., softmax_scale=None, q_scale=None,
                    causal=False, window_size=(-1, -1), deterministic=False, dtype=torch.bfloat16, version=None):
    """q [B,Lq,Nq,C1], k [B,Lk,Nk,C1], v [B,Lk,Nk,C2]; k_lens [B] masks the key tail.
    Returns [B,Lq,Nq,C2] in q's original dtype, computed in bf16 with fp32 accumulation.
    Built for what the DiT uses: C1 = C2 = 128, Nq = Nk, non-causal, no dropout, no
    window, q_lens = None or all equal to Lq; anything else raises."""
    if dtype != torch.bfloat16:
        raise NotImplementedError("flash_attention: only bf16 compute is built")
    if q.device.type != "cuda" or q.size(-1) > 256:
        raise AssertionError("flash_attention needs a HIP device tensor with head_dim <= 256")
    if causal or dropout_p != 0. or tuple(window_size) != (-1, -1):
        raise NotImplementedError("flash_attention: causal / dropout / window are not built")
    if q_lens is not None and any(int(n) != q.size(1) for n in q_lens.tolist()):
        # the reference unflattens the packed output back to [B, Lq] (attention.py:133), which only
        # works when every q_len equals Lq
        raise NotImplementedError("flash_attention: q_lens shorter than Lq are not built")
    if q.shape[2] != k.shape[2] or q.shape[3] != 128 or v.shape[3] != 128:
        raise NotImplementedError("flash_attention: needs Nq == Nk and head_dim 128")
    out_dtype = q.dtype
    b, lq = q.size(0), q.size(1)
    qb, kb, vb = (t.to(torch.bfloat16).contiguous() for t in (q, k, v))
    if q_scale is not None:
        qb = (qb * q_scale).to(torch.bfloat16)
    out = torch.empty_like(qb)
    if k_lens is None:
        H.flash_attn(qb, kb, vb, out, softmax_scale=softmax_scale)
    else:
        lens = [int(x) for x in k_lens.tolist()]
        for i in range(b):
            H.flash_attn(qb[i:i + 1], kb[i:i + 1], vb[i:i + 1], out[i:i + 1], k_len=lens[i],
                         softmax_scale=softmax_scale)
    return out.type(out_dtype)


def attention(q, k, v, q_lens=None, k_lens=None, dropout_p=0., softmax_scale=None, q_scale=None, causal=False,
              window_size=(-1, -1), deterministic=False, dtype=torch.bfloat16, fa_version=None):
    """reference attention.py:136-182: same entry, always the HIP kernel here."""
    return flash_attention(q=q, k=k, v=v, q_lens=q_lens, k_lens=k_lens, dropout_p=dropout_p,
                           softmax_scale=softmax_scale, q_scale=q_scale, causal=causal, window_size=window_size,
                           deterministic=deterministic, dtype=dtype, version=fa_version)
