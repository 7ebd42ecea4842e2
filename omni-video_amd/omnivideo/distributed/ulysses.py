"""Ulysses sequence-parallel attention (reference: omnivideo/distributed/ulysses.py)."""
import torch.distributed as dist

from ..modules.attention import flash_attention
from .util import all_to_all


def distributed_attention(q, k, v, seq_lens, window_size=(-1, -1), sp_group=None):
    """q [B, Lq/p, Nq, C1], k [B, Lk/p, Nk, C1], v [B, Lk/p, Nk, C2] (this rank's sequence
    shard, all heads) -> [B, Lq/p, Nq, C2].  Three all-to-alls turn the shards into the full
    sequence for Nq/p heads, the HIP flash-attention kernel runs on that, one all-to-all
    restores the sequence sharding (reference ulysses.py:41-55)."""
    if not dist.is_initialized():
        raise ValueError("distributed group should be initialized.")
    p = dist.get_world_size(group=sp_group) if sp_group is not None else dist.get_world_size()
    for name, t in (("q", q), ("k", k), ("v", v)):
        if t.size(2) % p != 0:
            raise ValueError(f"{name}.num_heads={t.size(2)} must be divisible by sp_size={p}")
    q = all_to_all(q, scatter_dim=2, gather_dim=1, group=sp_group)
    k = all_to_all(k, scatter_dim=2, gather_dim=1, group=sp_group)
    v = all_to_all(v, scatter_dim=2, gather_dim=1, group=sp_group)
    x = flash_attention(q, k, v, k_lens=seq_lens, window_size=window_size)
    return all_to_all(x, scatter_dim=1, gather_dim=2, group=sp_group)
