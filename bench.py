#!/usr/bin/env python
"""Headline benchmark: denoised frames/s of the Omni-Video 2 sampling loop on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one sampling step of the workload: two DiT forwards (conditional and
unconditional guidance branches, run as one batch of 2), the guidance combine and the
UniPC update - replayed from a HIP graph.  Default workload = BASELINE.json configs[1]:
OmniVideo2-1.3B text-to-video, 49 frames at 832x480, 50 UniPC steps, bf16, synthetic
inputs and random-init weights of that architecture (no checkpoints or datasets exist
here).  value = frames per second of whole-job throughput = n_gpus * 49 / (50 * step time);
with N GPUs every rank samples its own video (the reference's prompt-level data
parallelism, tools/inference/generate_omni_v2v.py:443-457,572-574), no data-path
collective, so scaling is weak.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "omni-video_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch                                    # noqa: E402
import torch.distributed as dist                # noqa: E402

WORKLOADS = {
    # name: model, latent [C,T,h,w], frames, sampling steps, text_len, context rows (text, vlm, null text)
    "cfg2": dict(desc="OmniVideo2-1.3B T2V 49f 832x480 50 steps bf16", model="1.3B", latent=(16, 13, 60, 104),
                 frames=49, steps=50, text_len=6144, n_text=20, n_vlm=60, n_null=126, visual=False),
    "cfg3": dict(desc="OmniVideo2-1.3B V2V 49f 832x480 40 steps bf16", model="1.3B", latent=(16, 13, 60, 104),
                 frames=49, steps=40, text_len=6272, n_text=550, n_vlm=700, n_null=126, visual=True),
    "cfg4": dict(desc="OmniVideo2-A14B T2V 81f 1280x720 50 steps bf16 (one expert resident in the bench)", model="A14B",
                 latent=(16, 21, 90, 160), frames=81, steps=50, text_len=6144, n_text=20, n_vlm=60, n_null=126,
                 visual=False),
    "cfg5": dict(desc="OmniVideo2-A14B V2V 81f 1280x720 40 steps, e4m3 linear layers + bf16 attention "
                      "(one expert resident in the bench)", model="A14B", latent=(16, 21, 90, 160), frames=81, steps=40,
                 text_len=6144, n_text=550, n_vlm=700, n_null=126, visual=True, precision="fp8"),
    "cfg1": dict(desc="OmniVideo2-1.3B T2V 17f 256x256 20 steps (bf16 on GPU)", model="1.3B", latent=(16, 5, 32, 32),
                 frames=17, steps=20, text_len=512, n_text=20, n_vlm=60, n_null=126, visual=False),
}
PEAK_BF16_TFLOPS = 2500.0        # MI355X dense bf16 MFMA peak


def dit_flops_per_forward(cfg, L, Lk, Lk_read=None):
    """2*MACs per forward of one guidance branch.  Algorithmic (the reference's own work): per
    layer L*(6C^2 + 2CF) + Lk*2C^2 linear, 4*L^2*C self-attention, 4*L*Lk*C cross-attention, plus
    patch / head / text embeddings.  Executed (Lk_read given): the text embedding and the
    cross-attention K/V projections run once per video outside the step, and cross-attention
    reads Lk_read keys (the run of identical zero-padding keys is folded into one weighted key)."""
    C, F, nl = cfg["dim"], cfg["ffn_dim"], cfg["num_layers"]
    self_att = 4 * L * L * C
    if Lk_read is None:
        lin = 2 * (L * (6 * C * C + 2 * C * F) + Lk * 2 * C * C)
        cross = 4 * L * Lk * C
        embed = 2 * (L * 64 * C + L * C * 64 + Lk * (cfg["text_dim"] * C + C * C))
    else:
        lin = 2 * L * (6 * C * C + 2 * C * F)
        cross = 4 * L * Lk_read * C
        embed = 2 * (L * 64 * C + L * C * 64)
    return nl * (lin + self_att + cross) + embed, nl * self_att


def build_model(cfg, text_len, device):
    from omnivideo.modules.unified_model import UnifiedWanWithMixedConditionModel
    torch.manual_seed(0)
    with torch.device(device):
        m = UnifiedWanWithMixedConditionModel(
            wan_config=dict(cfg), vlm_in_dim=cfg["vlm_in_dim"], use_visual_context_adapter=True,
            visual_context_adapter_patch_size=cfg["visual_patch_size"], max_context_len=text_len, skip_init=True)
    return m.to(torch.bfloat16).eval()


def cpu_baseline(cfg, wl, threads):
    """The CPU oracle (oracle/dit.py, fp32 flow = the reference's CPU path) timed on this
    box's host cores on a bounded sample: ONE DiT block of one guidance branch at the
    workload's shapes, extrapolated to layers x 2 branches x steps."""
    from omnivideo.utils import synthetic as syn
    from oracle import dit as O
    torch.set_num_threads(threads)
    one = dict(cfg, num_layers=1)
    sd = syn.make_unified_state_dict(one, 0)
    C = cfg["dim"]
    lat = wl["latent"]
    grid = (lat[1], lat[2] // 2, lat[3] // 2)
    L = grid[0] * grid[1] * grid[2]
    g = torch.Generator().manual_seed(0)
    x = torch.randn(L, C, generator=g)
    ctx = torch.randn(wl["text_len"], C, generator=g)
    e0 = torch.randn(6, C, generator=g) * 0.3
    ang = O.rope_angles(C // cfg["num_heads"], grid)
    reps = 2                                   # the block for both guidance branches: ~10-30 s of CPU work
    t0 = time.time()
    with torch.no_grad():
        for _ in range(reps):
            kv = O.cross_kv(ctx, sd, "wan_model.blocks.0.cross_attn", cfg["eps"], "fp32")
            O.block_forward(x, e0, kv, sd, "wan_model.blocks.0", cfg, ang, L, "fp32")
    dt = time.time() - t0
    per_video = dt / reps * cfg["num_layers"] * 2 * wl["steps"]
    return {"value": wl["frames"] / per_video, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"1 of {cfg['num_layers']} DiT blocks for both guidance branches, L={L}, Lk={wl['text_len']}, "
                      f"fp32: {dt:.1f} s; extrapolated x{cfg['num_layers']} layers x{wl['steps']} steps",
            "seconds_measured": dt}


class _CpuStagedComm:
    """Development aid for --rehearse-on-one-gpu: the communicator interface of
    omnivideo.modules.engine.TorchGroupComm over gloo, staging device tensors through the host
    (gloo's all-to-all and point-to-point paths take CPU tensors)."""

    def __init__(self):
        self.size, self.rank = dist.get_world_size(), dist.get_rank()

    def all_to_all(self, recv, send):
        src = send.cpu().contiguous().view(torch.uint8).reshape(self.size, -1)       # bytes: any element type
        r = torch.empty_like(src)
        dist.all_to_all_single(r, src)
        recv.copy_(r.view(recv.dtype).reshape(recv.shape))

    def all_reduce_sum(self, t):
        c = t.cpu()
        dist.all_reduce(c)
        t.copy_(c)

    def exchange_rows(self, up, down):
        ops, from_up, from_down = [], None, None
        if self.rank > 0:
            from_up = torch.empty(up.shape, dtype=up.dtype)
            ops += [dist.P2POp(dist.isend, up.cpu(), self.rank - 1), dist.P2POp(dist.irecv, from_up, self.rank - 1)]
        if self.rank < self.size - 1:
            from_down = torch.empty(down.shape, dtype=down.dtype)
            ops += [dist.P2POp(dist.isend, down.cpu(), self.rank + 1), dist.P2POp(dist.irecv, from_down, self.rank + 1)]
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return (None if from_up is None else from_up.to(up.device),
                None if from_down is None else from_down.to(down.device))

    def all_gather_rows(self, x, rows):
        mx = max(rows)
        pad = torch.zeros((x.shape[0], mx) + tuple(x.shape[2:]), dtype=x.dtype)
        pad[:, :x.shape[1]] = x.cpu()
        parts = [torch.empty_like(pad) for _ in range(self.size)]
        dist.all_gather(parts, pad)
        return torch.cat([p[:, :n] for p, n in zip(parts, rows)], dim=1).to(x.device)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg2", choices=list(WORKLOADS))
    ap.add_argument("--precision", choices=["bf16", "fp8"], default=None,
                    help="arithmetic of the large linear layers (default: the workload's; fp8 only for cfg5)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vae", action="store_true", help="skip the VAE decode leg")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--parallel", choices=["auto", "dp", "sp"], default="auto",
                    help="with --gpus N > 1: dp = one video per GPU (no data-path collective); sp = one video "
                         "over all N GPUs, Ulysses sequence parallelism; auto = sp for the A14B workloads "
                         "(40 heads), dp for the 1.3B ones (12 heads do not divide by 8)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="development aid: every rank uses cuda:0 and the gloo backend, to exercise the "
                         "multi-rank control flow on a one-GPU box (numbers are meaningless)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = "cpu" if args.rehearse_on_one_gpu else dev       # where the max-over-ranks reductions run
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="gloo" if args.rehearse_on_one_gpu else "nccl", init_method="env://")

    from omnivideo import _hip as H
    from omnivideo.utils import synthetic as syn
    from omnivideo.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler
    from omnivideo.x2x_gen_unified import DenoiseLoop
    H.lib()

    wl = WORKLOADS[args.workload]
    cfg = syn.dit_config(wl["model"], text_len=wl["text_len"])
    sp = world > 1 and (args.parallel == "sp" or (args.parallel == "auto" and wl["model"] == "A14B"))
    comm = None
    if sp:
        if cfg["num_heads"] % world:
            raise SystemExit(f"sequence parallelism needs the {cfg['num_heads']} heads to divide by {world} ranks")
        from omnivideo.modules.engine import TorchGroupComm
        comm = _CpuStagedComm() if args.rehearse_on_one_gpu else TorchGroupComm(None)
    K, W = args.steps, args.warmup
    if 1 + W + K > wl["steps"]:
        raise SystemExit(f"1 setup + {W} warmup + {K} timed steps exceed the workload's {wl['steps']} sampling steps")
    BF16 = torch.bfloat16
    model = build_model(cfg, wl["text_len"], dev)
    precision = args.precision or wl.get("precision", "bf16")
    model.wan_model.linear_precision = precision
    # data parallel: a different video per rank; sequence parallel: every rank holds the same inputs
    g = torch.Generator(device=dev).manual_seed(1818 + (0 if sp else 1000 * rank))
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    special = {k: rn(1, cfg["text_dim"]).to(BF16) for k in ("<img_st>", "<img_ed>", "<prp_st>", "<prp_ed>")}
    ctx_c, ar_c = [rn(wl["n_text"], cfg["text_dim"]).to(BF16)], [rn(wl["n_vlm"], cfg["vlm_in_dim"]).to(BF16)]
    ctx_u = [rn(wl["n_null"], cfg["text_dim"]).to(BF16)]
    ar_u = [(torch.zeros(2, cfg["vlm_in_dim"], device=dev) + 1e-6).to(BF16)]
    visual = rn(1, *wl["latent"]) if wl["visual"] else None
    noise = rn(*wl["latent"]).to(BF16)

    with torch.no_grad():
        mixed_c = model.build_mixed_context(1, ctx_c, ar_c, visual, special)[0]
        mixed_u = model.build_mixed_context(1, ctx_u, ar_u, visual, special)[0]
        sched = FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
        sched.set_timesteps(wl["steps"], device="cpu", shift=5.0)
        lat = noise.clone()
        seq_len = wl["latent"][1] * (wl["latent"][2] // 2) * (wl["latent"][3] // 2)
        use_graph = not args.no_graph and not sp          # collectives are not captured
        loop = DenoiseLoop([model] * wl["steps"], lat, {id(model): (mixed_c, mixed_u)}, seq_len, sched, 3.0,
                           use_graph=use_graph, comm=comm)
        run = loop.run

        def fence():
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        run(1)                      # setup step: lazy initialisation + graph capture (never timed)
        run(W)
        fence()
        t0 = time.perf_counter()
        run(K)
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], device=red_dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        assert torch.isfinite(lat.float()).all(), "latents went non-finite"

        # ---- roofline of the dominant kernel (self-attention): HIP events around each launch
        # of a second, eager pass of K steps on the same stream
        roof = None
        if rank == 0 or sp:          # under sequence parallelism the extra steps hold collectives: all ranks
            eng = loop.engines[id(model)]
            eng.timing = []
            for i in range(K):
                loop._step(id(model))
            torch.cuda.synchronize()
            ms = [a.elapsed_time(b) for a, b in eng.timing]
            eng.timing = None
            L = seq_len
            # batch of 2 guidance branches; a sequence-parallel rank runs all rows for its share of the heads
            flops_launch = 4.0 * L * L * cfg["dim"] * 2 / (world if sp else 1)
            avg_ms = sum(ms) / len(ms)
            ach = flops_launch / (avg_ms * 1e-3) / 1e12
            # HBM-side bytes per launch cannot be counted from here: they come from separate
            # rocprofv3 --pmc passes over the same kernel at the cfg2 shape, kept under profiles/
            traffic, tpath = None, os.path.join(ROOT, "profiles", "r1_attn_self_pmc_traffic.json")
            if args.workload in ("cfg2", "cfg3") and os.path.exists(tpath):
                traffic = json.load(open(tpath))["traffic_bytes_per_launch"]
            roof = {"bound": "mfma", "kernel": "attn_pipe_kernel_w8 (self-attention)", "achieved": round(ach, 1),
                    "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                    "traffic": traffic, "launches_timed": len(ms), "avg_launch_ms": round(avg_ms, 4),
                    "flop_per_launch": flops_launch}

    # ---- second-largest kernel family: the FFN-up GEMM (+ GELU) of one layer on the step's own buffers
    roof2 = None
    if rank == 0 and precision == "bf16" and not sp:
        eng = loop.engines[id(model)]
        Lw = eng.layers[0]
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        with torch.no_grad():
            for a, b in ev:
                a.record()
                H.gemm(eng.h, Lw["w_f0"], eng.ffn, Lw["b_f0"], H.EPI_BF16_GELU)
                b.record()
        torch.cuda.synchronize()
        ms2 = sorted(a.elapsed_time(b) for a, b in ev)[len(ev) // 2]
        fl2 = 2.0 * eng.h.shape[0] * cfg["ffn_dim"] * cfg["dim"]
        roof2 = {"bound": "mfma", "kernel": "gemm256_kernel<GELU> (FFN-up, one layer)", "achieved": round(fl2 / ms2 / 1e9, 1),
                 "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(fl2 / ms2 / 1e9 / PEAK_BF16_TFLOPS, 4),
                 "median_launch_ms": round(ms2, 4), "flop_per_launch": fl2}

    # ---- VAE decode of one video's latent (random-init full-width VAE), reported beside the loop
    vae_s = None
    if not args.no_vae:
        from omnivideo.modules.vae2_1 import Wan2_1_VAE
        with torch.no_grad(), torch.device(dev):
            vae = Wan2_1_VAE(z_dim=16, vae_pth=None, device=dev)
        z = torch.randn(*wl["latent"], device=dev, generator=g)
        with torch.no_grad():
            vae.decode([z], comm=comm)           # under sequence parallelism: row-sharded over the group
            fence()
            t0 = time.perf_counter()
            vae.decode([z], comm=comm)
            fence()
            vae_s = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([vae_s], device=red_dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            vae_s = float(tt.item())
        del vae

    ms_step = dt / K * 1e3
    videos = 1 if sp else world                  # videos in flight on the node
    value = videos * wl["frames"] / (wl["steps"] * ms_step * 1e-3)
    if rank == 0:
        cache = loop.caches[id(model)]
        fl_fwd, fl_self = dit_flops_per_forward(cfg, seq_len, wl["text_len"])
        fl_exec, _ = dit_flops_per_forward(cfg, seq_len, wl["text_len"], cache.k_len)
        out = {
            "metric": "denoised frames/sec (whole node)", "value": round(value, 4), "unit": "frames/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(ms_step, 3), "higher_is_better": True,
            "scaling": "strong" if sp else "weak", "vs_baseline": None,
            "dtype": "bf16" if precision == "bf16" else "fp8 (e4m3 linear layers, bf16 attention, fp32 residual stream)",
            "data": "synthetic",
            "config": {"workload": wl["desc"], "linear_precision": precision, "latent": list(wl["latent"]), "tokens": seq_len,
                       "context_len": wl["text_len"], "context_keys_read": cache.k_len, "sampling_steps": wl["steps"], "guide_scale": 3.0,
                       "parallelism": (f"sp{world}" if sp else f"dp{world}") if world > 1 else "single",
                       "scope": "sampling loop (2 DiT forwards + CFG + UniPC per step); per-video context "
                                "encode and VAE decode are outside the timed steps",
                       "graph_replay": use_graph},
            # algorithmic = the reference's work per step (2 forwards); executed = what the kernels ran
            "step_tflops": round(2 * fl_fwd / (ms_step * 1e-3) / 1e12, 1),
            "step_tflops_executed": round(2 * fl_exec / (ms_step * 1e-3) / 1e12, 1),
            "step_mfma_frac_executed": round(2 * fl_exec / (ms_step * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
            "roofline": roof,
        }
        if roof2 is not None:
            out["roofline_gemm"] = roof2
        if vae_s is not None:
            out["vae_decode_s_per_video"] = round(vae_s, 3)
            out["value_with_vae_decode"] = round(videos * wl["frames"] / (wl["steps"] * ms_step * 1e-3 + vae_s), 4)
        if not args.no_cpu_baseline and world == 1:      # the CPU leg is timed at N = 1 only
            out["cpu_baseline"] = cpu_baseline(cfg, wl, args.cpu_threads)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
