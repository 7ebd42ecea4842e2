#!/usr/bin/env python
"""Headline benchmark: denoised frames/s of the Omni-Video 2 sampling loop on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...`, or plain:
    with no WORLD_SIZE in the environment the script starts its own N ranks that way)

A "step" is one sampling step of the workload: two DiT forwards (conditional and
unconditional guidance branches, run as one batch of 2), the guidance combine and the
UniPC update - replayed from a HIP graph.  Default workload = BASELINE.json configs[1]:
OmniVideo2-1.3B text-to-video, 49 frames at 832x480, 50 UniPC steps, bf16, synthetic
inputs and random-init weights of that architecture (no checkpoints or datasets exist
here).  value = frames per second of whole-job throughput = n_gpus * 49 / (50 * step time + the
per-video context encode); with N GPUs every rank samples its own video (the reference's prompt-level data
parallelism, tools/inference/generate_omni_v2v.py:443-457,572-574), no data-path
collective, so scaling is weak.

Rank 0 prints ONE JSON line.  With N > 1 and the companion jobs enabled it prints that line as soon as the
headline measurement is complete and, once the companions have ended, the same line again with their results
merged in: the last line on stdout is always the most complete one, and the first one is there whatever happens
to the companions.  Progress ("[bench] phase ...") goes to stderr.

Every sequence-parallel number is gated: after the warm-up steps all ranks run the forward of the current step
with a reduced layer count sharded over the group, rank 0 runs the same forward unsharded on its own GPU, and
the two outputs have to be equal bit for bit (the sharded forward is bitwise equal to the unsharded one by
construction); after the timed steps the latents have to be identical on every rank.  The line carries
"sp_verified"; when it is false there is no "value".
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
    "cfg4": dict(desc="OmniVideo2-A14B T2V 81f 1280x720 50 steps bf16, both experts resident", model="A14B",
                 latent=(16, 21, 90, 160), frames=81, steps=50, text_len=6144, n_text=20, n_vlm=60, n_null=126,
                 visual=False),
    "cfg5": dict(desc="OmniVideo2-A14B V2V 81f 1280x720 40 steps, e4m3 linear layers + int8 Q K^T, "
                      "both experts resident", model="A14B", latent=(16, 21, 90, 160), frames=81, steps=40,
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


def build_model(cfg, text_len, device, seed=0):
    """Random-init model of the workload's architecture, constructed in bf16 on the device (no fp32 staging
    copy: at 14 B parameters that copy is 53 GiB the allocator would keep cached)."""
    from omnivideo.modules.unified_model import UnifiedWanWithMixedConditionModel
    torch.manual_seed(seed)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device(device):
            m = UnifiedWanWithMixedConditionModel(
                wan_config=dict(cfg), vlm_in_dim=cfg["vlm_in_dim"], use_visual_context_adapter=True,
                visual_context_adapter_patch_size=cfg["visual_patch_size"], max_context_len=text_len, skip_init=True)
    finally:
        torch.set_default_dtype(prev)
    return m.to(torch.bfloat16).eval()


def cpu_baseline(cfg, wl, threads):
    """The CPU oracle (oracle/dit.py, fp32 flow = the reference's CPU path) timed on this box's host
    cores on a bounded sample of the workload: whole WanModel forwards of one guidance branch at the
    workload's shapes (patch / time / text embedding, blocks, head, unpatchify) with 0 layers and
    with 1 layer.  A full forward is taken as t(0 layers) + num_layers x (t(1 layer) - t(0 layers)),
    a video as steps x 2 branches of that: the layer count and the step count are extrapolated, the
    per-forward work outside the blocks is measured.  The condition builder (per-video) and the
    sampler update (a few MB of elementwise work per step) are left out."""
    from omnivideo.utils import synthetic as syn
    from oracle import dit as O
    torch.set_num_threads(threads)
    one = dict(cfg, num_layers=1)
    sd = {k: v for k, v in syn.make_unified_state_dict(one, 0).items() if k.startswith("wan_model.")}
    g = torch.Generator().manual_seed(0)
    lat = torch.randn(*wl["latent"], generator=g)
    ctx = torch.randn(wl["n_text"] + wl["n_vlm"] + 2, cfg["text_dim"], generator=g)
    L = wl["latent"][1] * (wl["latent"][2] // 2) * (wl["latent"][3] // 2)
    times = []
    with torch.no_grad():
        for nl in (0, 1):
            t0 = time.time()
            O.wan_forward(lat, 500.0, ctx, L, sd, one, "fp32", num_layers=nl)
            times.append(time.time() - t0)
    per_forward = times[0] + cfg["num_layers"] * max(times[1] - times[0], 0.0)
    per_video = per_forward * 2 * wl["steps"]
    return {"value": wl["frames"] / per_video, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"one guidance branch, L={L}, Lk={wl['text_len']}, fp32: whole forward with 0 layers "
                      f"{times[0]:.1f} s and with 1 layer {times[1]:.1f} s; forward = t0 + {cfg['num_layers']} x (t1 - t0) "
                      f"= {per_forward:.0f} s, video = {wl['steps']} steps x 2 branches (layers and steps extrapolated; "
                      "condition builder and sampler update not included)",
            "seconds_measured": round(sum(times), 2)}


def measure_attention_traffic(B, L, N, limit_s=75):
    """HBM-side bytes of one self-attention launch at this shape, counted now: two `rocprofv3 --pmc` passes
    (FETCH_SIZE and WRITE_SIZE do not fit one pass) over a child process that runs only that launch
    (tools/bench_kernels.py attnbal).  On gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so it
    is doubled; WRITE_SIZE is exact; both are reported in KiB.  Returns (bytes, description) or raises."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        raise RuntimeError("rocprofv3 not found")
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        raise RuntimeError("this run is itself being profiled")
    kib = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        with tempfile.TemporaryDirectory(dir="/tmp") as d:
            env = dict(os.environ, TMPDIR="/tmp", OVD_BENCH_ATTN_SHAPE=f"{B},{L},{N}")
            cmd = [prof, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "t", "--", sys.executable,
                   os.path.join(ROOT, "tools", "bench_kernels.py"), "attnbal"]
            proc = subprocess.Popen(cmd, env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                    start_new_session=True)
            try:
                proc.wait(timeout=limit_s)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)
                proc.wait()
                raise RuntimeError(f"{counter} pass killed after {limit_s} s")
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if proc.returncode != 0 or not files:
                raise RuntimeError(f"{counter} pass: exit status {proc.returncode}, {len(files)} counter files")
            vals = [float(r["Counter_Value"]) for f in files for r in csv.DictReader(open(f))
                    if r["Counter_Name"] == counter and "attn_pipe_kernel" in r["Kernel_Name"]]
            if not vals:
                raise RuntimeError(f"{counter} pass: no attention launch in the counter file")
            kib[counter] = sum(vals) / len(vals)
    nbytes = int((2.0 * kib["FETCH_SIZE"] + kib["WRITE_SIZE"]) * 1024)
    return nbytes, (f"measured by this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the same launch in a child "
                    f"process (B{B} L{L} N{N}; FETCH_SIZE {kib['FETCH_SIZE']:.0f} KiB doubled for gfx950, WRITE_SIZE "
                    f"{kib['WRITE_SIZE']:.0f} KiB; the tail-merge kernel's ~24 MB not included)")


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N fresh ranks under torch.distributed.run
    (this process has not touched the GPU and never does), pass their output through and exit with
    their status."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


_LAUNCH_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK",
               "ROLE_WORLD_SIZE", "ROLE_NAME", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID",
               "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_USE_AGENT_STORE",
               "TORCHELASTIC_ERROR_FILE", "TORCH_NCCL_ASYNC_ERROR_HANDLING")


_T0 = time.time()


def phase(name, **kw):
    """Progress trail on stderr (rank 0 of a job speaks for it): which phase was entered, when, and how much
    device memory was free.  A job that is killed at a time limit leaves its last phase behind."""
    if int(os.environ.get("RANK", 0)) != 0 and not os.environ.get("OVD_BENCH_TRACE_ALL_RANKS"):
        return
    extra = "".join(f" {k}={v}" for k, v in kw.items())
    try:
        free, total = torch.cuda.mem_get_info()
        mem = f" free={free / 2**30:.1f}GiB of {total / 2**30:.0f}"
    except Exception:                    # noqa: BLE001 - before the device is set up
        mem = ""
    who = f" rank={os.environ['RANK']}" if os.environ.get("OVD_BENCH_TRACE_ALL_RANKS") and "RANK" in os.environ else ""
    print(f"[bench] phase {name} t={time.time() - _T0:.1f}s{mem}{extra}{who}", file=sys.stderr, flush=True)


def last_phase(stderr_text):
    lines = [ln for ln in stderr_text.splitlines() if ln.startswith("[bench] phase ")]
    return lines[-1][len("[bench] phase "):] if lines else None


def companion_run(workload, n, extra, limit_s):
    """After the headline measurement, on rank 0 alone (the other ranks have exited and this process has
    released its buffers): the sequence-parallel configuration of the metric - one `workload` video over
    all n GPUs - as a separate job of n fresh ranks (this script again, self-launched; `extra` selects the
    communicators), bounded in time and isolated from the headline line, which is on stdout before this
    starts.  A job that is killed reports the phase it had reached and whatever line it had printed by then."""
    import signal
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in _LAUNCH_ENV}
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(n), "--workload", workload, "--parallel", "sp",
           "--steps", "2", "--warmup", "1", "--no-vae", "--no-cpu-baseline", "--companion", "off"] + extra
    t0 = time.time()
    try:
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                start_new_session=True)
        killed = False
        try:
            so, se = proc.communicate(timeout=limit_s)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)
            so, se = proc.communicate()
            killed = True
        lines = [ln for ln in (so or "").splitlines() if ln.startswith("{")]
        res = {}
        if lines:
            r = json.loads(lines[-1])
            keep = ("value", "unit", "n_gpus", "ms_per_step", "scaling", "dtype", "context_encode_s_per_video",
                    "step_tflops", "roofline", "sp_verified", "sp_verification", "sp_comm_runs", "error")
            res = {k: r.get(k) for k in keep if k in r}
            res["config"] = {k: r["config"].get(k) for k in ("workload", "parallelism", "sp_comm", "experts_resident",
                                                              "timed_schedule_steps", "graph_replay")}
        if killed:
            res["stopped"] = f"killed after {limit_s:.0f} s in phase: {last_phase(se or '')}"
        elif proc.returncode != 0 or not lines:
            res["stopped"] = f"exit status {proc.returncode} in phase: {last_phase(se or '')}"
            res["stderr_tail"] = (se or "")[-600:]
        res["wall_s"] = round(time.time() - t0, 1)
        return res
    except Exception as err:          # noqa: BLE001 - the headline line must survive anything here
        return {"stopped": f"{type(err).__name__}: {err}"}


def bits_checksum(t):
    """Two 64-bit checksums of a tensor's bit pattern: the sum of its words and a position-weighted sum."""
    w = t.contiguous().view(torch.int16 if t.element_size() == 2 else torch.int32).to(torch.int64).view(-1)
    idx = torch.arange(w.numel(), device=w.device, dtype=torch.int64) % 65521 + 1
    return torch.stack([w.sum(), (w * idx).sum()])


def install_corruption(comm, engines):
    """Test hook (--test-corrupt-exchange): the last rank of the group damages what it receives in every
    exchange, the way a stale or torn read of a peer's buffer would.  The verification has to notice."""
    if comm.rank != comm.size - 1:
        return
    if getattr(comm, "direct", False):
        orig = comm.barrier

        def barrier():
            orig()
            for eng in engines:
                eng.sp_qkv.view(-1)[:64].add_(1.0)
        comm.barrier = barrier
    else:
        orig_a2a = comm.all_to_all

        def all_to_all(recv, send):
            orig_a2a(recv, send)
            recv.view(-1)[:64].add_(1.0)
        comm.all_to_all = all_to_all


def verify_sequence_parallel(loop, key, comm, rank, layers, all_reduce_min):
    """All ranks: the forward of the current step with the first `layers` layers, sharded over `comm`.
    Rank 0 of the group: the same forward unsharded (own workspaces, shared weights and context cache).
    Bitwise comparison on rank 0, and every rank's (full) output has to carry the same bits.
    Returns (ok, description)."""
    import copy
    eng, cache = loop.engines[key], loop.caches[key]
    full, ws = eng.layers, eng.attn_ws
    k = max(1, min(int(layers), len(full)))
    out_sp = torch.empty_like(loop.out)
    # Both forwards run without the attention kernel's tail-balancing workspace: whether a launch splits its last
    # query blocks over two key halves depends on its grid, which differs between the sharded and the unsharded
    # launch, and a split block sums in a different order.  Without the workspace every query block is computed
    # the same way on both sides and the comparison is exact.
    eng.layers, eng.attn_ws = full[:k], None
    try:
        eng.forward(loop.lat, loop.state["tsteps"], cache, out_sp, step_ptr=loop.step_ptr)
    finally:
        eng.layers, eng.attn_ws = full, ws
    torch.cuda.synchronize()
    ok, note = True, ""
    if rank == 0:
        ref = copy.copy(eng)
        ref._shape = None
        ref.layers = full[:k]
        ref.plan(eng.grid, eng.seq_len, eng.nb, comm=None)
        ref.attn_ws = None
        out_ref = torch.empty_like(loop.out)
        ref.forward(loop.lat, loop.state["tsteps"], cache, out_ref, step_ptr=loop.step_ptr)
        torch.cuda.synchronize()
        ok = bool(torch.equal(out_sp, out_ref))
        if not ok:
            d = (out_sp - out_ref).abs()
            note = f"; sharded vs unsharded: {int((d != 0).sum())} of {d.numel()} elements differ, max |diff| {float(d.max()):.3e}"
        del ref, out_ref
    same = all_reduce_min(bits_checksum(out_sp))
    return (ok and same), (f"{k}-layer forward at the current step, sharded over {comm.size} ranks vs unsharded on rank 0: "
                           f"{'bitwise equal' if ok else 'DIFFERENT'}; output identical on every rank: {same}{note}")


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
    ap.add_argument("--no-vae", action="store_true", help="skip the VAE legs (decode; encode of the source clip for V2V)")
    ap.add_argument("--no-traffic", action="store_true",
                    help="skip the two rocprofv3 counter passes that measure the attention launch's HBM-side bytes")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--parallel", choices=["auto", "dp", "sp"], default="auto",
                    help="with --gpus N > 1: dp = one video per GPU (no data-path collective); sp = one video "
                         "over all N GPUs, Ulysses sequence parallelism; auto = sp for the A14B workloads "
                         "(40 heads), dp for the 1.3B ones (12 heads do not divide by 8)")
    ap.add_argument("--sp-size", type=int, default=None,
                    help="ranks per sequence-parallel group (consecutive ranks); the N / sp-size groups sample one "
                         "video each (reference: tools/inference/generate_omni_v2v.py:443-457, "
                         "omnivideo/distributed/util.py:43-66).  Default: N under --parallel sp, else 1")
    ap.add_argument("--companion", default="auto",
                    help="with --gpus N > 1 and the default (data-parallel 1.3B) workload, rank 0 afterwards also runs "
                         "the sequence-parallel configuration of the metric (cfg4: A14B 81f 720p over all N GPUs, "
                         "when 40 heads divide by N) as a separate bounded job and reports it under "
                         "'sequence_parallel_companion'; 'off' disables, a workload name selects another one")
    ap.add_argument("--companion-budget", type=float, default=240.0, help="seconds for the companion job")
    ap.add_argument("--sp-comm", default="auto",
                    help="communicator(s) of the sequence-parallel group, comma-separated, measured in this order: "
                         "rccl = all-to-alls of the collective library between graph segments; peer = peer-mapped "
                         "device memory, the step is one graph (omnivideo/distributed/peer_mapped.py); host = staged "
                         "through host memory (one-GPU rehearsals); auto = OVD_SP_COMM, else rccl (host under "
                         "--rehearse-on-one-gpu).  The line's value comes from the first one whose result verifies; "
                         "a single 'peer' that fails verification falls back to rccl")
    ap.add_argument("--verify-layers", type=int, default=2,
                    help="layers of the sharded-vs-unsharded forward comparison that gates sequence-parallel numbers")
    ap.add_argument("--test-corrupt-exchange", choices=["off", "peer", "all"], default="off", help=argparse.SUPPRESS)
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="development aid: every rank uses cuda:0 and the gloo backend, to exercise the "
                         "multi-rank control flow on a one-GPU box (numbers are meaningless)")
    ap.add_argument("--layers", type=int, default=None,
                    help="development aid: build the model with this many layers (the line says so; not a result)")
    ap.add_argument("--experts", type=int, default=None, choices=[1, 2],
                    help="development aid: experts held resident (default: 2 for A14B, 1 for 1.3B)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)
    # a rank that stops making progress leaves its Python stack on stderr every 150 s
    import faulthandler
    faulthandler.dump_traceback_later(float(os.environ.get("OVD_BENCH_STACK_DUMP_S", "150")), repeat=True, file=sys.stderr)
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = "cpu" if args.rehearse_on_one_gpu else dev       # where the max-over-ranks reductions run
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="gloo" if args.rehearse_on_one_gpu else "nccl", init_method="env://")
    phase("start", workload=args.workload, world=world)

    from omnivideo import _hip as H
    from omnivideo.modules.engine import make_group_comm, planned_bytes
    from omnivideo.utils import synthetic as syn
    from omnivideo.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler
    from omnivideo.x2x_gen_unified import DenoiseLoop
    H.lib()

    wl = WORKLOADS[args.workload]
    cfg = syn.dit_config(wl["model"], text_len=wl["text_len"])
    if args.layers is not None:
        cfg = dict(cfg, num_layers=int(args.layers))
    sp_size = args.sp_size
    if sp_size is None:
        sp_size = world if (world > 1 and (args.parallel == "sp" or (args.parallel == "auto" and wl["model"] == "A14B"))) else 1
    sp = sp_size > 1
    if world % sp_size:
        raise SystemExit(f"--sp-size {sp_size} does not divide the {world} ranks")
    if sp and cfg["num_heads"] % sp_size:
        raise SystemExit(f"sequence parallelism needs the {cfg['num_heads']} heads to divide by {sp_size} ranks")
    dp_rank, n_groups = rank // sp_size, world // sp_size
    sp_group = None
    if sp and world > 1:
        if sp_size == world:
            from omnivideo.distributed.util import init_sequence_parallel_group
            sp_group = init_sequence_parallel_group(sp_size)
        else:                            # data-parallel groups of sequence-parallel groups
            from omnivideo.distributed.util import init_distributed_groups
            _, sp_group = init_distributed_groups(sp_size)
    sp_rank = dist.get_rank(sp_group) if sp else 0
    K, W = args.steps, args.warmup
    if W + K > wl["steps"]:
        raise SystemExit(f"{W} warmup + {K} timed steps exceed the workload's {wl['steps']} sampling steps")
    BF16 = torch.bfloat16
    precision = args.precision or wl.get("precision", "bf16")
    n_experts = args.experts or (2 if wl["model"] == "A14B" else 1)
    seq_len = wl["latent"][1] * (wl["latent"][2] // 2) * (wl["latent"][3] // 2)
    grid = (wl["latent"][1], wl["latent"][2] // 2, wl["latent"][3] // 2)

    # ---- what this rank is about to allocate, against what the device has: refuse before allocating
    # (ranks that rehearse on one GPU share its memory; the unsharded verification forward runs on rank 0)
    plan = planned_bytes(cfg, grid, seq_len, 2, wl["text_len"], sp=sp_size, precision=precision,
                         attn_workspace=H.flash_attn_workspace_bytes(2, seq_len, cfg["num_heads"] // sp_size, seq_len))
    per_rank = n_experts * plan["total"]
    verify_extra = planned_bytes(cfg, grid, seq_len, 2, wl["text_len"], sp=1, precision=precision,
                                 attn_workspace=H.flash_attn_workspace_bytes(2, seq_len, cfg["num_heads"], seq_len))["workspaces"] if sp else 0
    if world > 1:
        dist.barrier()                   # every rank looks at the device before any of them has allocated
    free_b, total_b = torch.cuda.mem_get_info()
    if world > 1:
        dist.barrier()
    sharing = world if args.rehearse_on_one_gpu else 1
    need = sharing * per_rank + verify_extra
    phase("plan", per_rank_GiB=round(per_rank / 2**30, 1), verify_GiB=round(verify_extra / 2**30, 1),
          ranks_on_this_gpu=sharing, need_GiB=round(need / 2**30, 1))
    if need > 0.97 * free_b:
        msg = (f"{args.workload} needs {per_rank / 2**30:.1f} GiB per rank ({n_experts} expert(s): parameters "
               f"{plan['parameters'] / 2**30:.1f}, engine copies {plan['engine_copies'] / 2**30:.1f}, context cache "
               f"{plan['context_cache'] / 2**30:.1f}, workspaces {plan['workspaces'] / 2**30:.1f} GiB each)"
               + (f" + {verify_extra / 2**30:.1f} GiB for the unsharded verification forward" if sp else "")
               + f"; {sharing} rank(s) on this GPU need {need / 2**30:.1f} GiB, {free_b / 2**30:.1f} GiB are free: refused")
        if rank == 0:
            print(json.dumps({"metric": "denoised frames/sec (whole node)", "value": None, "unit": "frames/s", "n_gpus": world,
                              "error": msg, "config": {"workload": wl["desc"]}}), flush=True)
        raise SystemExit(msg)

    # the A14B pipeline holds two experts, switched where the timestep falls below 875
    # (reference x2x_gen_unified.py:228-258,407); the 1.3B pipeline one model
    phase("build")
    experts = [build_model(cfg, wl["text_len"], dev, seed) for seed in range(n_experts)]
    for m in experts:
        m.wan_model.linear_precision = precision
    model = experts[0]
    torch.cuda.empty_cache()
    # data parallel: a different video per group; sequence parallel: every rank of a group holds the same inputs
    g = torch.Generator(device=dev).manual_seed(1818 + 1000 * dp_rank)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    special = {k: rn(1, cfg["text_dim"]).to(BF16) for k in ("<img_st>", "<img_ed>", "<prp_st>", "<prp_ed>")}
    ctx_c, ar_c = [rn(wl["n_text"], cfg["text_dim"]).to(BF16)], [rn(wl["n_vlm"], cfg["vlm_in_dim"]).to(BF16)]
    ctx_u = [rn(wl["n_null"], cfg["text_dim"]).to(BF16)]
    ar_u = [(torch.zeros(2, cfg["vlm_in_dim"], device=dev) + 1e-6).to(BF16)]
    noise = rn(*wl["latent"]).to(BF16)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(*vals):
        if world == 1:
            return vals
        tt = torch.tensor(vals, device=red_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return tuple(float(v) for v in tt.tolist())

    def group_all(flag):
        """True on every rank of the sequence-parallel group iff `flag` is true on all of them."""
        tt = torch.tensor([1 if flag else 0], device=red_dev, dtype=torch.int32)
        dist.all_reduce(tt, op=dist.ReduceOp.MIN, group=sp_group)
        return bool(tt.item())

    def same_in_group(vec):
        lo = vec.to(red_dev).clone()
        hi = lo.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=sp_group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=sp_group)
        return bool((lo == hi).all())

    # ---- V2V: the source clip goes through the VAE encoder first (reference tools/inference/
    # generate_omni_v2v.py:609-614); its latents are the visual condition of both guidance branches
    vae, vae_enc_s, visual = None, None, None
    if wl["visual"]:
        if args.no_vae:
            visual = rn(1, *wl["latent"])
        else:
            phase("vae-encode")
            from omnivideo.modules.vae2_1 import Wan2_1_VAE
            with torch.no_grad(), torch.device(dev):
                vae = Wan2_1_VAE(z_dim=16, vae_pth=None, device=dev)
            lt, lh, lw = wl["latent"][1:]
            clip = (torch.rand(3, 4 * (lt - 1) + 1, 8 * lh, 8 * lw, device=dev, generator=g) * 2 - 1)
            with torch.no_grad():
                vae.encode([clip])
                fence()
                t0 = time.perf_counter()
                z_src = vae.encode([clip])[0]
                fence()
                vae_enc_s = time.perf_counter() - t0
            (vae_enc_s,) = max_over_ranks(vae_enc_s)
            assert tuple(z_src.shape) == tuple(wl["latent"]), z_src.shape
            visual = z_src[None].float()
            del clip

    with torch.no_grad():
        phase("context")
        pairs = {id(m): (m.build_mixed_context(1, ctx_c, ar_c, visual, special)[0],
                         m.build_mixed_context(1, ctx_u, ar_u, visual, special)[0]) for m in experts}
        sched = FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
        sched.set_timesteps(wl["steps"], device="cpu", shift=5.0)
        by_step = [experts[0] if (t >= 875 or len(experts) == 1) else experts[1] for t in sched.timesteps.tolist()]
        n_high = sum(1 for m in by_step if m is experts[0]) if len(experts) == 2 else None
        use_graph = not args.no_graph      # under sequence parallelism: graphs between eager collectives
        # two experts: the timed steps straddle the switch, so that both experts' weights, context caches
        # and graphs are live in the timed region
        first = 0 if n_high is None else max(0, min(n_high - W - (K + 1) // 2, wl["steps"] - W - K))
        timed_steps = list(range(first + W, first + W + K))

        def measure(comm, kind):
            """One complete measurement over one communicator (None: no sequence parallelism)."""
            r = {"kind": kind}
            lat = noise.clone()
            phase("plan-engines", comm=kind)
            loop = DenoiseLoop(by_step, lat, pairs, seq_len, sched, 3.0, use_graph=use_graph, comm=comm)
            if comm is not None and (args.test_corrupt_exchange == "all" or
                                     (args.test_corrupt_exchange == "peer" and kind == "peer")):
                install_corruption(comm, list(loop.engines.values()))
            # setup (never timed as steps): lazy initialisation and one graph capture per expert
            fence()
            phase("capture", comm=kind)
            t0 = time.perf_counter()
            loop.prepare()
            fence()
            r["capture_s"] = time.perf_counter() - t0
            # per-video set-up the reference spreads over every step: context assembly is done above; the text
            # embedding and every layer's cross-attention K/V projection are timed here and charged to the video
            phase("encode-context", comm=kind)
            fence()
            t0 = time.perf_counter()
            for m in experts:
                loop.engines[id(m)].encode_context(list(pairs[id(m)]), m.wan_model.text_len)
            fence()
            ctx_s = time.perf_counter() - t0
            phase("warmup", comm=kind)
            loop.seek(first)
            loop.run(W)
            fence()
            if comm is not None:
                phase("verify", comm=kind)
                ok, note = verify_sequence_parallel(loop, id(by_step[first + W]), comm, sp_rank, args.verify_layers,
                                                    same_in_group)
                ok = group_all(ok)
                r["verified"], r["verification"] = ok, note
                fence()
            phase("timed-steps", comm=kind)
            t0 = time.perf_counter()
            loop.run(K)
            fence()
            dt = time.perf_counter() - t0
            r["dt"], r["ctx_s"] = max_over_ranks(dt, ctx_s)
            assert torch.isfinite(lat.float()).all(), "latents went non-finite"
            if comm is not None:
                agree = same_in_group(bits_checksum(lat))
                r["verified"] = group_all(r["verified"] and agree)
                r["verification"] += f"; latents after the timed steps identical on every rank: {agree}"
                if hasattr(comm, "check"):
                    try:
                        comm.check()
                    except RuntimeError as err:
                        r["verified"] = False
                        r["verification"] += f"; {err}"
            # ---- roofline of the dominant kernel (self-attention): HIP events around each launch
            # of a second, eager pass of K steps on the same stream
            roof = None
            phase("roofline", comm=kind)
            if rank == 0 or sp:          # under sequence parallelism the extra steps hold collectives: all ranks
                eng = loop.engines[id(model)]
                eng.timing = []
                loop.seek(first)
                for i in range(K):
                    loop._step(id(model))
                torch.cuda.synchronize()
                ms = [ev[0].elapsed_time(ev[1]) for ev in eng.timing]
                branches = [ev[2] for ev in eng.timing]
                eng.timing = None
                L = seq_len
                # a launch covers one or both guidance branches; a sequence-parallel rank runs all rows for its
                # share of the heads
                flops_total = sum(4.0 * L * L * cfg["dim"] * nbr / sp_size for nbr in branches)
                flops_launch = flops_total / len(ms)
                avg_ms = sum(ms) / len(ms)
                ach = flops_total / (sum(ms) * 1e-3) / 1e12
                # HBM-side bytes per launch: separate rocprofv3 --pmc passes over the same kernel in a child process
                traffic, tsrc = None, None
                if world == 1 and not args.no_traffic and eng.attention == "bf16":
                    try:
                        traffic, tsrc = measure_attention_traffic(2, L, cfg["num_heads"])
                    except Exception as err:          # noqa: BLE001 - the counters are an extra, the line must survive
                        tsrc = f"live counter passes failed ({type(err).__name__}: {err}); "
                for name in (() if traffic is not None else ("r2_attn_self_pmc_traffic.json", "r1_attn_self_pmc_traffic.json")):
                    tpath = os.path.join(ROOT, "profiles", name)
                    if args.workload in ("cfg2", "cfg3") and not sp and os.path.exists(tpath):
                        traffic = json.load(open(tpath))["traffic_bytes_per_launch"]
                        tsrc = (tsrc or "") + f"offline rocprofv3 --pmc passes at this shape, profiles/{name} (not measured by this run)"
                        break
                int8_attn = eng.attention == "int8"
                # int8 Q K^T runs at twice the bf16 rate, P V at the bf16 rate, half of the FLOPs each:
                # the launch's matrix-time roof is 1 / (0.5 / 5000 + 0.5 / 2500) = 3333 TFLOP/s
                peak = 1.0 / (0.5 / (2 * PEAK_BF16_TFLOPS) + 0.5 / PEAK_BF16_TFLOPS) if int8_attn else PEAK_BF16_TFLOPS
                roof = {"bound": "mfma", "kernel": ("attn_i8_kernel_w8" if int8_attn else "attn_pipe_kernel_w8") + " (self-attention)",
                        "achieved": round(ach, 1), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                        "traffic": traffic, "traffic_source": tsrc, "launches_timed": len(ms), "avg_launch_ms": round(avg_ms, 4),
                        "flop_per_launch": flops_launch}

            r["roof"], r["k_len"] = roof, loop.caches[id(model)].k_len
            if comm is None:
                r["loop"] = loop                     # the GEMM roofline below runs on the step's own buffers
            return r

        runs = []
        if not sp:
            res = measure(None, None)
        else:
            kinds = [k.strip() for k in args.sp_comm.split(",") if k.strip()]
            fallback = "host" if args.rehearse_on_one_gpu else "rccl"
            if kinds == ["peer"]:
                kinds = ["peer", fallback + "?"]             # "?": only if nothing before it verified
            res = None
            for kind in kinds:
                if kind.endswith("?"):
                    if res is not None and res["verified"]:
                        continue
                    kind = kind[:-1]
                comm = make_group_comm(sp_group, None if kind == "auto" else kind)
                cur = measure(comm, type(comm).__name__ if kind == "auto" else kind)
                runs.append({"sp_comm": cur["kind"], "class": type(comm).__name__, "sp_verified": cur["verified"],
                             "ms_per_step": round(cur["dt"] / K * 1e3, 3), "verification": cur["verification"],
                             "payload": getattr(comm, "payload", None),
                             "roofline_frac": cur["roof"]["frac"] if cur["roof"] else None})
                if res is None or (cur["verified"] and not res["verified"]):
                    res = cur
                # the engines' exchange buffers belong to this communicator: forget them before it goes
                for m in experts:
                    m.wan_model.engine()._shape = None
                fence()
                if hasattr(comm, "close"):
                    comm.close()
                comm = None
        loop, dt, ctx_s, capture_s, roof = res.get("loop"), res["dt"], res["ctx_s"], res["capture_s"], res["roof"]
        sp_ok = res.get("verified", True)

    # ---- second-largest kernel family: the FFN-up GEMM (+ GELU) of one layer on the step's own buffers
    roof2 = None
    if rank == 0 and precision == "bf16" and not sp:
        eng = loop.engines[id(model)]
        Lw = eng.layers[0]
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        with torch.no_grad():
            # the device has idled (counter passes in a child process, host work): the first launches after an
            # idle period run at a low clock, so spend 0.3 s on the same launch before timing it
            t_warm = time.perf_counter()
            while time.perf_counter() - t_warm < 0.3:
                for _ in range(8):
                    H.gemm(eng.h, Lw["w_f0"], eng.ffn, Lw["b_f0"], H.EPI_BF16_GELU)
                torch.cuda.synchronize()
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
        # all six linear layers of one block, each on the step's own buffers (median of 10 launches each)
        Mrows, C, F = eng.h.shape[0], cfg["dim"], cfg["ffn_dim"]
        qkv2, att2 = eng.qkv.view(Mrows, 3 * C), eng.att.view(Mrows, C)
        calls = {
            "qkv": (lambda: H.gemm(eng.h, Lw["w_qkv"], qkv2, Lw["b_qkv"], H.EPI_BF16), 2.0 * Mrows * 3 * C * C),
            "o": (lambda: H.gemm(att2, Lw["w_o"], eng.res, Lw["b_o"], H.EPI_BF16), 2.0 * Mrows * C * C),
            "cross_q": (lambda: H.gemm(eng.h, Lw["w_cq"], qkv2[:, :C], Lw["b_cq"], H.EPI_BF16), 2.0 * Mrows * C * C),
            "cross_o": (lambda: H.gemm(att2, Lw["w_co"], eng.res, Lw["b_co"], H.EPI_BF16), 2.0 * Mrows * C * C),
            "ffn_up_gelu": (lambda: H.gemm(eng.h, Lw["w_f0"], eng.ffn, Lw["b_f0"], H.EPI_BF16_GELU), 2.0 * Mrows * F * C),
            "ffn_down": (lambda: H.gemm(eng.ffn, Lw["w_f2"], eng.res, Lw["b_f2"], H.EPI_BF16), 2.0 * Mrows * C * F),
        }
        per, tot_ms, tot_fl = {}, 0.0, 0.0
        with torch.no_grad():
            for name, (fn, fl) in calls.items():
                evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
                for a, b in evs:
                    a.record()
                    fn()
                    b.record()
                torch.cuda.synchronize()
                m = sorted(a.elapsed_time(b) for a, b in evs)[len(evs) // 2]
                per[name] = round(fl / m / 1e9, 1)
                tot_ms += m
                tot_fl += fl
        roof2["block_linears"] = {"achieved": round(tot_fl / tot_ms / 1e9, 1), "frac": round(tot_fl / tot_ms / 1e9 / PEAK_BF16_TFLOPS, 4),
                                  "ms": round(tot_ms, 4), "flop": tot_fl, "tflops_each": per}

    # ---- VAE decode of one video's latent (random-init full-width VAE), reported beside the loop
    vae_s = None
    if not args.no_vae:
        phase("vae-decode")
        from omnivideo.modules.vae2_1 import Wan2_1_VAE
        if vae is None:
            with torch.no_grad(), torch.device(dev):
                vae = Wan2_1_VAE(z_dim=16, vae_pth=None, device=dev)
        z = torch.randn(*wl["latent"], device=dev, generator=g)
        comm = make_group_comm(sp_group, "auto") if sp else None       # row-sharded decode over the group
        vae_flop = [0.0]
        with torch.no_grad():
            # the untimed first decode also counts the convolutions' FLOPs (2 x output voxels x taps x Cin x Cout, as
            # fp32 work) through the two wrappers every convolution of the VAE goes through
            import omnivideo.modules.vae2_1 as vae_mod
            conv_plain, conv_split = vae_mod.H.vae_conv3d, vae_mod.H.vae_conv3d_ps

            def counted(fn):
                def run(x, w, *a, **kw):
                    y = fn(x, w, *a, **kw)
                    taps, cout, cin = w.shape
                    vae_flop[0] += 2.0 * (y.numel() / y.shape[-1]) * taps * cout * cin
                    return y
                return run
            vae_mod.H.vae_conv3d, vae_mod.H.vae_conv3d_ps = counted(conv_plain), counted(conv_split)
            try:
                vae.decode([z], comm=comm)       # under sequence parallelism: row-sharded over the group
            finally:
                vae_mod.H.vae_conv3d, vae_mod.H.vae_conv3d_ps = conv_plain, conv_split
            fence()
            t0 = time.perf_counter()
            vae.decode([z], comm=comm)
            fence()
            vae_s = time.perf_counter() - t0
        (vae_s,) = max_over_ranks(vae_s)
    vae = None

    ms_step = dt / K * 1e3
    videos = n_groups                             # videos in flight on the node
    # a video = its sampling steps + its context encode (reference work hoisted out of the steps)
    value = videos * wl["frames"] / (wl["steps"] * ms_step * 1e-3 + ctx_s)
    out = None
    if rank == 0:
        phase("report")
        k_len = res["k_len"]
        fl_fwd, fl_self = dit_flops_per_forward(cfg, seq_len, wl["text_len"])
        fl_exec, _ = dit_flops_per_forward(cfg, seq_len, wl["text_len"], k_len) if k_len is not None else (None, None)
        par = "single"
        if world > 1:
            par = f"sp{sp_size}" if sp_size == world else f"dp{world}" if not sp else f"dp{n_groups} x sp{sp_size}"
        out = {
            "metric": "denoised frames/sec (whole node)", "value": round(value, 4) if sp_ok else None, "unit": "frames/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(ms_step, 3), "higher_is_better": True,
            "scaling": "strong" if (sp and n_groups == 1) else "weak", "vs_baseline": None,
            "dtype": "bf16" if precision == "bf16" else "fp8 (e4m3 linear layers, int8 Q K^T in self-attention with bf16 P V, fp32 residual stream)",
            "data": "synthetic",
            "config": {"workload": wl["desc"], "linear_precision": precision, "latent": list(wl["latent"]), "tokens": seq_len,
                       "context_len": wl["text_len"], "context_keys_read": k_len, "sampling_steps": wl["steps"], "guide_scale": 3.0,
                       "parallelism": par,
                       "sp_comm": (res["kind"] if sp else None),
                       "scope": "per video: sampling_steps x the timed step (2 DiT forwards + CFG + UniPC) + the context "
                                "encode (text embedding, cross-attention K/V of every layer and expert); VAE legs "
                                "reported separately; graph capture is once per process and shape",
                       "experts_resident": len(experts), "timed_schedule_steps": timed_steps,
                       "high_noise_steps": n_high, "graph_replay": use_graph},
            "context_encode_s_per_video": round(ctx_s, 4), "graph_capture_s_per_process": round(capture_s, 3),
            "value_loop_only": round(videos * wl["frames"] / (wl["steps"] * ms_step * 1e-3), 4) if sp_ok else None,
            # algorithmic = the reference's work per step (2 forwards); executed = what the kernels ran
            "step_tflops": round(n_groups * 2 * fl_fwd / (ms_step * 1e-3) / 1e12, 1),
            "roofline": roof,
        }
        out["device_memory"] = {"planned_GiB_per_rank": round(per_rank / 2**30, 2),
                                "peak_allocated_GiB": round(torch.cuda.max_memory_allocated() / 2**30, 2),
                                "note": "planned = parameters + engine copies + context caches + workspaces of the experts "
                                        "(engine.planned_bytes), computed before anything was allocated; the peak also holds "
                                        "the VAE legs, the verification forward and the graphs' private pools"}
        if args.layers is not None:
            out["config"]["layers_override"] = args.layers
            out["value"] = out["value_loop_only"] = None
            out["error"] = "development run with a reduced layer count: not a result"
        if fl_exec is not None:
            out["step_tflops_executed"] = round(2 * fl_exec / (ms_step * 1e-3) / 1e12, 1)
            out["step_mfma_frac_executed"] = round(2 * fl_exec / (ms_step * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4)
        if sp:
            out["sp_verified"] = sp_ok
            out["sp_verification"] = res["verification"]
            out["sp_comm_runs"] = runs
            if not sp_ok:
                out["error"] = "the sequence-parallel result did not verify against the unsharded forward: no value"
            elif runs and runs[0]["sp_comm"] != res["kind"]:
                out["sp_comm_fallback"] = f"{runs[0]['sp_comm']} failed verification; the value is from {res['kind']}"
        if roof2 is not None:
            out["roofline_gemm"] = roof2
        if vae_enc_s is not None:
            out["vae_encode_s_per_video"] = round(vae_enc_s, 3)
        if vae_s is not None:
            out["vae_decode_s_per_video"] = round(vae_s, 3)
            if not sp and vae_flop[0] > 0:
                # every fp32 product of the convolutions is three bf16 MFMA products of hi / lo operand halves
                ach = 3.0 * vae_flop[0] / vae_s / 1e12
                out["roofline_vae"] = {"bound": "mfma", "kernel": "whole VAE decode (conv3d_ws_kernel holds 85 % of its FLOPs)",
                                       "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                       "frac": round(ach / PEAK_BF16_TFLOPS, 4), "fp32_flop_per_decode": vae_flop[0],
                                       "note": "split-bf16 arithmetic: 3 bf16 MFMA products per fp32 product; wall time of "
                                               "the decode, all kernels included"}
            out["value_with_vae_decode"] = round(videos * wl["frames"] / (wl["steps"] * ms_step * 1e-3 + ctx_s + vae_s), 4) if sp_ok else None
            if vae_enc_s is not None:        # V2V: encode of the source clip + loop + decode
                out["value_with_vae"] = round(videos * wl["frames"] / (wl["steps"] * ms_step * 1e-3 + ctx_s + vae_s + vae_enc_s), 4) if sp_ok else None
    comp = args.companion
    if comp == "auto":
        comp = "cfg4" if (world > 1 and not sp and args.workload == "cfg2" and not args.rehearse_on_one_gpu) else "off"
    will_run_companion = comp != "off" and world > 1
    if rank == 0 and will_run_companion:
        out["sequence_parallel_companion"] = {"pending": "this line was printed before the companion job started"}
        print(json.dumps(out), flush=True)
    if rank == 0 and not args.no_cpu_baseline and world == 1:      # the CPU leg is timed at N = 1 only
        phase("cpu-baseline")
        out["cpu_baseline"] = cpu_baseline(cfg, wl, args.cpu_threads)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if will_run_companion:
            if syn.dit_config(WORKLOADS[comp]["model"])["num_heads"] % world == 0:
                # free this process's buffers; the other ranks leave on their own once the group is gone
                eng = Lw = roof = loop = res = lat = None                     # noqa: F841 - drop every reference to device buffers
                del experts, model, pairs
                import gc
                gc.collect()
                torch.cuda.empty_cache()
                time.sleep(3.0)
                phase("companion", workload=comp)
                # one job, both communicators measured on the same resident models: the collective library
                # first, then peer-mapped memory (whole step as one graph)
                extra = ["--rehearse-on-one-gpu", "--sp-comm", "host,peer"] if args.rehearse_on_one_gpu else ["--sp-comm", "rccl,peer"]
                out["sequence_parallel_companion"] = companion_run(comp, world, extra, limit_s=args.companion_budget)
            else:
                out["sequence_parallel_companion"] = {"skipped": f"{comp}: heads do not divide by {world} ranks"}
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
