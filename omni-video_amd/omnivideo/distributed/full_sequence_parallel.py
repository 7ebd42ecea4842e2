"""Sequence-parallel forwards with the reference's names and binding convention
(reference: omnivideo/distributed/full_sequence_parallel.py): the pipeline binds
`sp_dit_forward` onto `wan_model`, `sp_attn_forward` onto every `block.self_attn` and
`sp_cross_attn_forward` onto every `block.cross_attn` with `types.MethodType`, after
setting `.sp_group` on each (reference x2x_gen_unified.py:172-182)."""
import torch

from .. import _hip as H
from ..modules.engine import build_rope_table, make_group_comm
from .ulysses import distributed_attention
from .util import get_rank, get_world_size

BF16 = torch.bfloat16


def sp_dit_forward(self, x, t, context, seq_len, y=None):
    """WanModel.forward under sequence parallelism (reference :63-167).  Every rank passes the
    same x / t / context and gets the full result; the token rows are sharded inside the
    engine (modules/engine.py)."""
    group = getattr(self, "sp_group", None)
    comm = getattr(self, "_sp_comm", None)
    if comm is None or comm.group is not group:
        comm = make_group_comm(group) if get_world_size(group) > 1 else None
        self._sp_comm = comm
    return self._forward_impl(x, t, context, seq_len, y=y, comm=comm)


def sp_attn_forward(self, x, seq_lens, grid_sizes, freqs=None, dtype=torch.bfloat16):
    """WanSelfAttention.forward on this rank's row shard x [B, L/p, C] (reference :170-201):
    local q/k/v projections and RMSNorm, RoPE at the shard's token offset (:49-55), Ulysses
    attention, local output projection."""
    group = getattr(self, "sp_group", None)
    p, r = get_world_size(group), get_rank(group)
    B, s, C = x.shape
    N, D = self.num_heads, self.head_dim
    xb = x.to(BF16).contiguous()
    qkv = torch.empty(B, s, 3 * C, dtype=BF16, device=x.device)
    for i, w in enumerate((self.q, self.k, self.v)):
        H.gemm(xb.view(B * s, C), w.weight.data, qkv.view(B * s, 3 * C)[:, i * C:(i + 1) * C], w.bias.data)
    for b in range(B):
        grid = tuple(int(v) for v in grid_sizes[b].tolist())
        rope = build_rope_table(D, max(grid), x.device)
        for i, nw in ((0, self.norm_q), (1, self.norm_k)):
            v_ = qkv[b, :, i * C:(i + 1) * C]
            H.rmsnorm_rope(v_, nw.weight.data, v_, D, self.eps, rope=rope, rows_per_sample=s, tok_offset=r * s,
                           n_rope=grid[0] * grid[1] * grid[2], grid=grid)
    q, k, v = (qkv[:, :, i * C:(i + 1) * C].reshape(B, s, N, D) for i in range(3))
    a = distributed_attention(q, k, v, seq_lens, window_size=self.window_size, sp_group=group)
    out = torch.empty(B, s, C, dtype=BF16, device=x.device)
    H.gemm(a.reshape(B * s, C).contiguous(), self.o.weight.data, out.view(B * s, C), self.o.bias.data)
    return out


def sp_cross_attn_forward(self, x, context, context_lens, dtype=torch.bfloat16):
    """WanCrossAttention.forward with the query rows AND the context rows sharded over the
    group, as the reference does (:204-232): the shards are all-to-all'ed so each rank attends
    the full context for its head group."""
    group = getattr(self, "sp_group", None)
    B, s, C = x.shape
    N, D = self.num_heads, self.head_dim
    s_ctx = context.shape[1]
    xb, cb = x.to(BF16).contiguous(), context.to(BF16).contiguous()
    q = torch.empty(B, s, C, dtype=BF16, device=x.device)
    kv = torch.empty(B, s_ctx, 2 * C, dtype=BF16, device=x.device)
    H.gemm(xb.view(B * s, C), self.q.weight.data, q.view(B * s, C), self.q.bias.data)
    H.rmsnorm_rope(q.view(B * s, C), self.norm_q.weight.data, q.view(B * s, C), D, self.eps)
    kv2 = kv.view(B * s_ctx, 2 * C)
    H.gemm(cb.view(B * s_ctx, C), self.k.weight.data, kv2[:, :C], self.k.bias.data)
    H.gemm(cb.view(B * s_ctx, C), self.v.weight.data, kv2[:, C:], self.v.bias.data)
    H.rmsnorm_rope(kv2[:, :C], self.norm_k.weight.data, kv2[:, :C], D, self.eps)
    a = distributed_attention(q.view(B, s, N, D), kv[:, :, :C].reshape(B, s_ctx, N, D),
                              kv[:, :, C:].reshape(B, s_ctx, N, D), context_lens, sp_group=group)
    out = torch.empty(B, s, C, dtype=BF16, device=x.device)
    H.gemm(a.reshape(B * s, C).contiguous(), self.o.weight.data, out.view(B * s, C), self.o.bias.data)
    return out
