"""Sequence-parallel engine forward on ONE GPU: p ranks run as threads of this process,
each with its own DiTEngine (its own row shard and buffers), exchanging through an
in-process communicator with the same two collectives the RCCL communicator offers.
The sharded result must equal the unsharded forward bit for bit: every kernel computes
each row / head independently of how rows and heads are distributed."""
import threading

import pytest
import torch

from omnivideo.utils import synthetic as syn

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


class _Shared:
    def __init__(self, n):
        self.n = n
        self.slots = [None] * n
        self.barrier = threading.Barrier(n)


class ThreadComm:
    def __init__(self, shared, rank):
        self.sh, self.rank, self.size = shared, rank, shared.n

    def all_to_all(self, recv, send):
        torch.cuda.synchronize()
        self.sh.slots[self.rank] = send
        self.sh.barrier.wait()
        for i in range(self.size):
            recv[i].copy_(self.sh.slots[i][self.rank])
        torch.cuda.synchronize()
        self.sh.barrier.wait()

    def all_reduce_sum(self, t):
        torch.cuda.synchronize()
        self.sh.slots[self.rank] = t.clone()
        self.sh.barrier.wait()
        t.zero_()
        for i in range(self.size):
            t.add_(self.sh.slots[i])
        torch.cuda.synchronize()
        self.sh.barrier.wait()


def build(cfg, sd):
    from omnivideo.modules.unified_model import UnifiedWanWithMixedConditionModel
    m = UnifiedWanWithMixedConditionModel(wan_config=dict(cfg), vlm_in_dim=cfg["vlm_in_dim"],
                                          use_visual_context_adapter=True,
                                          visual_context_adapter_patch_size=cfg["visual_patch_size"],
                                          max_context_len=cfg["text_len"], eps=cfg["eps"], skip_init=True)
    m.load_state_dict(sd, strict=True)
    return m.to(BF16).cuda().eval().wan_model


@pytest.mark.parametrize("heads,sp,seq_len,precision", [(2, 2, 72, "bf16"), (2, 2, 80, "bf16"), (4, 4, 80, "bf16"),
                                                        (4, 2, 76, "bf16"), (4, 2, 80, "fp8"), (2, 2, 72, "fp8")])
def test_sharded_forward_equals_unsharded(heads, sp, seq_len, precision):
    from omnivideo.modules.engine import DiTEngine as _Engine
    DiTEngine = lambda wm: _Engine(wm, precision=precision)     # fp8: e4m3 linear layers + int8 Q K^T
    cfg = syn.dit_config("tiny", dim=128 * heads, num_heads=heads, ffn_dim=512)
    sd = syn.make_unified_state_dict(cfg, 0)
    wm = build(cfg, sd)
    lat = syn.randn("in.lat", (16, 3, 8, 12), 1).cuda().to(BF16)
    ctxs = [syn.randn("in.text", (10, cfg["text_dim"]), 1).cuda().to(BF16),
            syn.randn("in.text_null", (6, cfg["text_dim"]), 1).cuda().to(BF16)]
    ts = torch.tensor([640.0], dtype=torch.float64, device="cuda")
    grid = (3, 4, 6)
    ref_eng = DiTEngine(wm)
    ref_eng.plan(grid, seq_len, 2)
    cache = ref_eng.encode_context(ctxs, wm.text_len)
    ref = torch.empty(2, 16, 3, 8, 12, device="cuda")
    ref_eng.forward(lat, ts, cache, ref)
    torch.cuda.synchronize()

    shared = _Shared(sp)
    outs, errs = [None] * sp, []

    def worker(r):
        try:
            eng = DiTEngine(wm)
            eng.plan(grid, seq_len, 2, comm=ThreadComm(shared, r))
            out = torch.empty(2, 16, 3, 8, 12, device="cuda")
            eng.forward(lat, ts, cache, out)
            torch.cuda.synchronize()
            outs[r] = out
        except Exception as e:          # noqa: BLE001
            errs.append(e)
            shared.barrier.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(sp)]
    [t.start() for t in th]
    [t.join(120) for t in th]
    assert not errs, errs
    for r in range(sp):
        assert torch.equal(outs[r], ref), f"rank {r}: max diff {(outs[r] - ref).abs().max().item()}"


def test_head_divisibility_is_enforced():
    """12 heads cannot be split 8 ways (reference ulysses.py:33-34)."""
    from omnivideo.modules.engine import DiTEngine
    cfg = syn.dit_config("tiny", dim=384, num_heads=3, ffn_dim=512)
    wm = build(cfg, syn.make_unified_state_dict(cfg, 0))
    eng = DiTEngine(wm)
    with pytest.raises(ValueError, match="divisible"):
        eng.plan((3, 4, 6), 72, 1, comm=ThreadComm(_Shared(2), 0))


# ------------------------------------------------------------------ row-sharded VAE decode

class ThreadRowComm:
    """exchange_rows / all_gather_rows of engine.TorchGroupComm between threads of this process."""

    def __init__(self, shared, rank):
        self.sh, self.rank, self.size = shared, rank, shared.n

    def _swap(self, value):
        torch.cuda.synchronize()
        self.sh.slots[self.rank] = value
        self.sh.barrier.wait()
        got = list(self.sh.slots)
        self.sh.barrier.wait()
        return got

    def exchange_rows(self, up, down):
        got = self._swap((up.clone(), down.clone()))
        from_up = got[self.rank - 1][1] if self.rank > 0 else None
        from_down = got[self.rank + 1][0] if self.rank < self.size - 1 else None
        return from_up, from_down

    def all_gather_rows(self, x, rows):
        got = self._swap(x.clone())
        assert [g.shape[1] for g in got] == list(rows)
        return torch.cat(got, dim=1)


@pytest.mark.parametrize("ranks,latent", [(2, (16, 3, 4, 6)), (3, (16, 2, 7, 4)), (4, (16, 1, 4, 4)), (2, (16, 5, 10, 6))])
def test_row_sharded_vae_decode_equals_unsharded(ranks, latent):
    """Every rank decodes a horizontal strip, exchanging one row with each neighbour before each
    spatial convolution; the gathered video equals the single-GPU decode bit for bit (uneven
    strips, strips of one row, several temporal chunks)."""
    from omnivideo.modules.vae2_1 import Wan2_1_VAE
    cfg = syn.vae_config("tiny")
    sd = syn.make_vae_state_dict(cfg, 0)

    def make():
        m = Wan2_1_VAE(z_dim=cfg["z_dim"], vae_pth=None, device="cuda", dim=cfg["dim"])
        m.model.load_state_dict(sd, strict=True)
        return m

    v = make()
    per_rank = [make() for _ in range(ranks)]        # one model (and feature cache) per rank, as in real ranks
    z = syn.randn("vae.shard", latent, 3).cuda()
    ref = v.decode([z])[0]
    ref_frames = v.decode_frames([z])[0]
    shared = _Shared(ranks)
    outs, errs = [None] * ranks, []

    def worker(r):
        try:
            torch.cuda.set_device(0)
            comm = ThreadRowComm(shared, r)
            outs[r] = (per_rank[r].decode([z], comm=comm)[0], per_rank[r].decode_frames([z], comm=comm)[0])
        except Exception as e:      # surfaces in the main thread
            errs.append(e)
            shared.barrier.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(ranks)]
    [t.start() for t in th]
    [t.join(timeout=120) for t in th]
    assert not errs, errs
    for r in range(ranks):
        assert torch.equal(outs[r][0], ref), f"rank {r}"
        assert torch.equal(outs[r][1], ref_frames), f"rank {r}"
