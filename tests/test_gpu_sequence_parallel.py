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


@pytest.mark.parametrize("heads,sp,seq_len", [(2, 2, 72), (2, 2, 80), (4, 4, 80), (4, 2, 76)])
def test_sharded_forward_equals_unsharded(heads, sp, seq_len):
    from omnivideo.modules.engine import DiTEngine
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
