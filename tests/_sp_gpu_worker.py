"""One sequence-parallel rank of tests/test_gpu_sp_processes.py: a fresh process on cuda:0.  Several
ranks share the one GPU of the test box, so the process group runs on gloo and the product picks its
host-staged communicator (RCCL refuses two ranks on one device); everything else - group creation,
method binding, row sharding, RoPE offsets, head scatter, context broadcast, row-sharded VAE decode -
is the code a multi-GPU run executes.  Results go to <outdir>/rank<r>.pt for the parent to compare."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "omni-video_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import _sp_case as case                              # noqa: E402


def case_generate(outdir, rank, world):
    pipe, cfg = case.build_pipeline(use_usp=True, sp_size=world, rank=rank)
    assert pipe.sp_size == world and pipe.sp_comm is not None
    wm = pipe.high_noise_model.wan_model
    assert wm.forward.__func__.__name__ == "sp_dit_forward"
    assert wm.blocks[0].self_attn.forward.__func__.__name__ == "sp_attn_forward"
    assert wm.blocks[0].cross_attn.forward.__func__.__name__ == "sp_cross_attn_forward"
    lat, video = case.run_generate(pipe, cfg)                 # graphs between eager collectives, or one graph
    if os.environ.get("OVD_SP_COMM") == "peer":
        from omnivideo.distributed.peer_mapped import PeerMappedComm
        assert isinstance(pipe.sp_comm, PeerMappedComm) and pipe.sp_comm.barriers > 0
    pipe.use_graph = False
    lat_eager, _ = case.run_generate(pipe, cfg)               # every launch eager
    if lat is not None:
        assert torch.equal(lat, lat_eager), "segmented graph replay differs from the eager step"
    # the model-level entry the pipeline binds (sp_dit_forward): every rank gets the full output
    kw = case.generate_inputs(cfg)
    t = torch.tensor([640.0], device="cuda")
    y = wm(kw["noise"].cuda()[None].unbind(0), t, [kw["precomputed_context"].cuda()], 72)[0]
    torch.cuda.synchronize()
    torch.save(dict(lat=None if lat is None else lat.cpu(), video=None if video is None else video.cpu(),
                    dit=y.cpu()), os.path.join(outdir, f"rank{rank}.pt"))


def case_attention(outdir, rank, world):
    """sp_attn_forward / sp_cross_attn_forward / distributed_attention on this rank's row shard."""
    import types
    from omnivideo.distributed import util as U
    from omnivideo.distributed.full_sequence_parallel import sp_attn_forward, sp_cross_attn_forward
    from omnivideo.distributed.ulysses import distributed_attention
    from omnivideo.modules.attention import flash_attention
    pipe, cfg = case.build_pipeline(use_usp=False)
    group = U.init_sequence_parallel_group(world)
    blk = pipe.high_noise_model.wan_model.blocks[1]
    x, ctx, grid, seq_lens = case.attention_inputs(cfg)
    B, L, C = x.shape
    s, sc = L // world, ctx.shape[1] // world
    N, D = blk.self_attn.num_heads, blk.self_attn.head_dim
    res = {}
    for m in (blk.self_attn, blk.cross_attn):
        m.sp_group = group
    sa = types.MethodType(sp_attn_forward, blk.self_attn)
    ca = types.MethodType(sp_cross_attn_forward, blk.cross_attn)
    res["self"] = sa(x[:, rank * s:(rank + 1) * s].contiguous(), seq_lens, grid).cpu()
    res["cross"] = ca(x[:, rank * s:(rank + 1) * s].contiguous(), ctx[:, rank * sc:(rank + 1) * sc].contiguous(),
                      None).cpu()
    q, k, v = (t.view(B, L, N, D) for t in (x, x.flip(1).contiguous(), (x * 0.5).contiguous()))
    sl = slice(rank * s, (rank + 1) * s)
    res["ulysses"] = distributed_attention(q[:, sl].contiguous(), k[:, sl].contiguous(), v[:, sl].contiguous(),
                                           seq_lens, sp_group=group).cpu()
    res["ulysses_ref_rows"] = flash_attention(q, k, v, k_lens=seq_lens)[:, sl].cpu()
    torch.cuda.synchronize()
    torch.save(res, os.path.join(outdir, f"rank{rank}.pt"))


def case_peer(outdir, rank, world):
    """The peer-mapped communicator on its own: all-to-all, sum and barrier, eagerly and replayed from one
    graph several times in a row, against values every rank can compute; then a barrier that one rank
    joins too late, which has to end with the status word set on every rank instead of hanging."""
    from omnivideo.distributed.peer_mapped import PeerMappedComm
    comm = PeerMappedComm(None, timeout_s=20)
    dev = "cuda"
    n = 3 * 1024 + 8                                   # elements per chunk (16-byte multiple, not a power of two)
    send = comm.alloc((world, n), torch.bfloat16)
    recv = torch.empty(world, n, dtype=torch.bfloat16, device=dev)
    tot = torch.empty(4099, dtype=torch.float32, device=dev)          # odd length: scalar tail of the sum
    val = lambda src, dst, it: (torch.arange(n, device=dev) % 251 + 7 * src + 3 * dst + it).to(torch.bfloat16)

    def fill(it):
        for j in range(world):
            send[j].copy_(val(rank, j, it))
        tot.copy_(torch.arange(4099, device=dev, dtype=torch.float32) * (rank + 1) + it)

    def exchange():
        comm.all_to_all(recv, send)
        comm.all_reduce_sum(tot)

    def verify(it):
        for j in range(world):
            assert torch.equal(recv[j], val(j, rank, it)), f"all_to_all chunk {j} iteration {it}"
        want = sum(torch.arange(4099, device=dev, dtype=torch.float32) * (r + 1) + it for r in range(world))
        assert torch.equal(tot, want), f"all_reduce_sum iteration {it}"

    fill(0); exchange(); verify(0)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        exchange()
    for it in range(1, 5):
        fill(it)
        comm.barrier()                                 # peers have verified the previous round before send changes
        g.replay()
        verify(it)
        comm.barrier()
    comm.check()
    torch.cuda.synchronize()
    dist.barrier()
    # A barrier that the last rank joins too late: the ranks that gave up waiting (1 s) mark themselves as
    # failed in every peer's flag array and go on; the late rank finds their epochs already there, passes - and
    # must find the marks as well.  check() then has to raise on EVERY rank (a rank that raised alone would
    # leave the others blocked in their next collective).
    import time
    comm.timeout_s = 1.0
    if rank == world - 1:
        time.sleep(3.0)
    comm.barrier()
    torch.cuda.synchronize()
    own = comm.status()
    assert own != 0, "the failed barrier left this rank's status word clear"
    assert (own == world if rank < world - 1 else own > 64), f"rank {rank}: status word {own}"
    dist.barrier()
    try:
        comm.check()
    except RuntimeError as err:
        assert "barrier failed" in str(err) and "every rank" in str(err), str(err)
    else:
        raise AssertionError(f"rank {rank}: check() passed although a barrier of the group timed out")
    comm.barrier()                                     # later barriers return at once instead of waiting
    torch.cuda.synchronize()
    torch.save(dict(ok=True), os.path.join(outdir, f"rank{rank}.pt"))


def engine_case(precision, comm=None, replays=0, shape="tiny4", overlap=None):
    """One DiT forward, optionally sharded over `comm` and replayed from a graph.  shape "tiny4": a 4-head
    tiny model over seq_len 80 (8 padding rows; the inputs of tests/test_gpu_sequence_parallel.py); "a14b":
    one layer of the A14B width (C = 5120, 40 heads, FFN 13 824) on the full 81-frame 720p grid, 75 600
    tokens (the inputs of tests/test_gpu_fullsize_a14b.py)."""
    from omnivideo.modules.engine import DiTEngine
    from omnivideo.modules.unified_model import UnifiedWanWithMixedConditionModel
    from omnivideo.utils import synthetic as syn
    BF16 = torch.bfloat16
    if shape == "a14b":
        cfg = syn.dit_config("A14B", num_layers=1, text_len=512)
        lat_shape, grid, seq_len = (16, 21, 90, 160), (21, 45, 80), 75600
    else:
        cfg = syn.dit_config("tiny", dim=512, num_heads=4, ffn_dim=512)
        lat_shape, grid, seq_len = (16, 3, 8, 12), (3, 4, 6), 80
    m = UnifiedWanWithMixedConditionModel(wan_config=dict(cfg), vlm_in_dim=cfg["vlm_in_dim"],
                                          use_visual_context_adapter=True,
                                          visual_context_adapter_patch_size=cfg["visual_patch_size"],
                                          max_context_len=cfg["text_len"], eps=cfg["eps"], skip_init=True)
    m.load_state_dict(syn.make_unified_state_dict(cfg, 0), strict=True)
    wm = m.to(BF16).cuda().eval().wan_model
    lat = syn.randn("in.lat", lat_shape, 1).cuda().to(BF16)
    ctxs = [syn.randn("in.text", (10, cfg["text_dim"]), 1).cuda().to(BF16),
            syn.randn("in.text_null", (6, cfg["text_dim"]), 1).cuda().to(BF16)]
    ts = torch.tensor([640.0], dtype=torch.float64, device="cuda")
    eng = DiTEngine(wm, precision=precision)
    if overlap is not None:                # False: the serial order of exchanges and attention
        eng.sp_overlap = overlap
    eng.plan(grid, seq_len, 2, comm=comm)
    cache = eng.encode_context(ctxs, wm.text_len)
    out = torch.empty(2, *lat_shape, device="cuda")
    eng.forward(lat, ts, cache, out)
    torch.cuda.synchronize()
    first = out.clone()
    if replays:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            eng.forward(lat, ts, cache, out)
        for _ in range(replays):
            out.fill_(float("nan"))
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, first), "graph replay of the sharded forward differs from the eager one"
    return first


def case_peer_engine(outdir, rank, world):
    """The sharded engine forward over `world` processes with the peer-mapped communicator (kernels write to
    and read from the peers), eager and as one replayed graph, bf16 and reduced precision."""
    from omnivideo.distributed.peer_mapped import PeerMappedComm
    res = {}
    for precision in ("bf16", "fp8"):
        for direct in (True, False):
            comm = PeerMappedComm(None, timeout_s=20, direct=direct)
            res[f"{precision}_{'direct' if direct else 'pull'}"] = engine_case(precision, comm, replays=3).cpu()
            if direct:     # the same exchange with both guidance branches in one attention launch, nothing beside it
                res[f"{precision}_direct_serial"] = engine_case(precision, comm, replays=2, overlap=False).cpu()
            comm.check()
            comm.close()
    torch.save(res, os.path.join(outdir, f"rank{rank}.pt"))


def case_peer_a14b(outdir, rank, world):
    """One A14B-width layer on the 75 600-token grid over `world` ranks (8: the per-rank shapes of the
    81-frame 720p workload - 9 450 rows, 5 heads, exchange buffers of 0.6 GB addressed on 8 peers)."""
    from omnivideo.distributed.peer_mapped import PeerMappedComm
    comm = PeerMappedComm(None, timeout_s=60)
    res = {}
    for precision in ("bf16", "fp8"):             # fp8: e4m3 linear layers + int8 Q K^T, the 81-frame V2V workload's mode
        res[precision] = engine_case(precision, comm, replays=1, shape="a14b").cpu()
        comm.check()
    if rank in (0, world - 1):                    # every rank holds the full output; two of them are compared
        torch.save(res, os.path.join(outdir, f"rank{rank}.pt"))
    comm.close()


def main():
    which, outdir = sys.argv[1], sys.argv[2]
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="env://")
    rank, world = dist.get_rank(), dist.get_world_size()
    with torch.no_grad():
        {"generate": case_generate, "attention": case_attention, "peer": case_peer,
         "peer_engine": case_peer_engine, "peer_a14b": case_peer_a14b}[which](outdir, rank, world)
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
