"""Worker for test_distributed_cpu.py: one gloo rank.  Checks the collectives of
omnivideo.distributed.util and a Ulysses-sharded self-attention (oracle math + the
product's collectives) against the unsharded oracle."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "omni-video_amd"))

from omnivideo.distributed import util as U          # noqa: E402
from omnivideo.utils import synthetic as syn         # noqa: E402
from oracle import dit as O                          # noqa: E402


def main():
    dist.init_process_group("gloo", init_method="env://")
    world, rank = dist.get_world_size(), dist.get_rank()
    sp = int(os.environ.get("SP_SIZE", world))
    dp_group, group = U.init_distributed_groups(sp)
    assert U.get_sequence_parallel_size() == sp and U.get_world_size(group) == sp
    # data-parallel groups are the strided complements: ranks {r % sp == const}
    assert dist.get_world_size(dp_group) == world // sp and dist.get_rank(dp_group) == rank // sp
    r = U.get_rank(group)
    assert r == rank % sp

    # all_to_all: scatter heads / gather sequence, and back (reference util.py:95-121)
    B, L, N, D = 1, 4 * sp, 2 * sp, 8
    full = syn.randn("a2a.x", (B, L, N, D), 1)
    mine = full[:, r * (L // sp):(r + 1) * (L // sp)]
    y = U.all_to_all(mine, scatter_dim=2, gather_dim=1, group=group)
    assert torch.equal(y, full[:, :, r * (N // sp):(r + 1) * (N // sp)]), "seq-gather/head-scatter"
    z = U.all_to_all(y, scatter_dim=1, gather_dim=2, group=group)
    assert torch.equal(z, mine), "round trip"
    g = U.gather_forward(mine, dim=1, group=group)
    assert torch.equal(g, full), "gather_forward"

    # Ulysses-sharded self-attention == unsharded (rope offsets, padded tail, head split)
    cfg = syn.dit_config("tiny")
    sd = syn.make_unified_state_dict(cfg, 0)
    C, NH, grid, S = cfg["dim"], cfg["num_heads"], (3, 4, 6), 80
    x = syn.randn("probe.x", (1, S, C), 3)[0]
    pre = "wan_model.blocks.1.self_attn"
    if NH % sp == 0 and S % sp == 0:      # the head split needs sp | heads (reference ulysses.py:33-34)
        ref = O.self_attention(x, sd, pre, NH, cfg["eps"], O.rope_angles(C // NH, grid), 72, "fp32")
        s = S // sp
        xs = x[r * s:(r + 1) * s]
        Dh = C // NH
        q = O.rms_norm(O.linear(xs, sd, pre + ".q", "fp32"), sd[pre + ".norm_q.weight"], cfg["eps"], "fp32")
        k = O.rms_norm(O.linear(xs, sd, pre + ".k", "fp32"), sd[pre + ".norm_k.weight"], cfg["eps"], "fp32")
        v = O.linear(xs, sd, pre + ".v", "fp32")
        ang = O.rope_angles(Dh, grid)                       # [72, D/2]; rows past it are not rotated
        lo, hi = r * s, min((r + 1) * s, ang.shape[0])
        a_loc = ang[lo:hi] if hi > lo else ang[:0]
        q = O.rope_rotate(q.view(s, NH, Dh), a_loc)
        k = O.rope_rotate(k.view(s, NH, Dh), a_loc)
        qf = U.all_to_all(q[None], 2, 1, group)[0]
        kf = U.all_to_all(k[None], 2, 1, group)[0]
        vf = U.all_to_all(v.view(1, s, NH, Dh), 2, 1, group)[0]
        of = O.attention(qf, kf, vf, k_len=72, flow="fp32")
        o = U.all_to_all(of[None], 1, 2, group)[0]
        out = O.linear(o.reshape(s, C), sd, pre + ".o", "fp32")
        full_out = U.gather_forward(out[None], 1, group)[0]
        torch.testing.assert_close(full_out, ref, rtol=1e-4, atol=1e-5)

    # row-sharded VAE decode: the communicator's neighbour exchange and row gather, and the strip
    # arithmetic of vae2_1._Shard, on a CPU stand-in of the decoder's spatial pattern
    # (3x3 conv -> nearest 2x upsample + 3x3 conv) with uneven strips
    import torch.nn.functional as F
    from omnivideo.modules.engine import TorchGroupComm
    from omnivideo.modules.vae2_1 import _Shard, shard_rows
    comm = TorchGroupComm(group)
    assert (comm.rank, comm.size) == (r, sp)
    up, down = comm.exchange_rows(torch.full((2, 3), float(r)), torch.full((2, 3), float(r) + 0.5))
    assert (up is None) == (r == 0) and (down is None) == (r == sp - 1)
    assert up is None or torch.equal(up, torch.full((2, 3), float(r - 1) + 0.5))
    assert down is None or torch.equal(down, torch.full((2, 3), float(r + 1)))
    T, Hh, Ww, Cc = 2, 3 * sp + 1, 5, 3
    rows = shard_rows(Hh, sp)
    assert sum(rows) == Hh and max(rows) - min(rows) <= 1
    full = syn.randn("halo.x", (T, Hh, Ww, Cc), 2)
    w1, w2 = syn.randn("halo.w1", (4, Cc, 3, 3), 2) * 0.3, syn.randn("halo.w2", (2, 4, 3, 3), 2) * 0.3

    def conv(x, w):          # channels-last 3x3 convolution with zero padding
        return F.conv2d(x.permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1).contiguous()

    def up2(x):
        return x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)

    want = conv(up2(conv(full, w1)), w2)
    sh = _Shard(comm, rows)
    xs = sh.strip(full)
    assert xs.shape[1] == rows[r] + 2
    sh.refresh(xs)
    y = conv(xs, w1)
    sh.refresh(y)
    u = conv(up2(y), w2)[:, 1:-1].contiguous()
    got = sh.gather(u, scale=2)
    torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-6)
    assert torch.equal(sh.gather(xs), full)

    # data-parallel prompt sharding arithmetic (reference generate_omni_v2v.py:443-457,572-574)
    dp_world = world // sp
    dp_rank = rank // sp
    prompts = list(range(7))
    mine_p = [i for i in prompts if i % dp_world == dp_rank]
    got = [None] * world
    dist.all_gather_object(got, mine_p)
    assert sorted(set(sum(got, []))) == prompts
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
