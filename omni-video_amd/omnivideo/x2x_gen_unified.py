"""Sampling pipeline with the reference's entry point (reference:
omnivideo/x2x_gen_unified.py: OmniVideoX2XUnified.__init__ :26-44, .generate :261-278).

What changes underneath, with unchanged arithmetic per element:
  * both guidance branches run as one batch of 2 through a DiTEngine (engine.py), with
    the mixed context assembled, embedded and projected to per-layer cross-attention
    K/V once per video instead of once per layer per forward per step;
  * guidance combine + UniPC update is one fused launch per step;
  * a whole step (2 forwards + update) is captured once per expert into a HIP graph and
    replayed; the step index lives on the device, so the loop has no host
    synchronisation (the reference syncs on `t.item()` every step, :235); under sequence
    parallelism the step is a chain of graphs with the collectives issued between them;
  * both experts stay resident in HBM (288 GB): no CPU offload round trips.
Not provided: FSDP sharding.  The umT5 text encoder (modules/t5.py) is attached when
`checkpoint_dir` holds its checkpoint and tokenizer files; otherwise text conditions come in as
tensors (`precomputed_context`, and `precomputed_context_null` for the negative prompt), or
through any `self.text_encoder` callable with the reference's `(texts, device) -> list[Tensor]`
convention.
"""
import logging
import math
import os
import random
import sys

import torch
import torch.distributed as dist

from . import _hip as H
from .modules.unified_model import UnifiedWanWithMixedConditionModel
from .utils.fm_solvers import FlowDPMSolverMultistepScheduler, get_sampling_sigmas, retrieve_timesteps
from .utils.fm_solvers_unipc import FlowUniPCMultistepScheduler

BF16, F32 = torch.bfloat16, torch.float32


class _SegmentedGraph:
    """One sampling step of a sequence-parallel rank as a chain of HIP graphs with the collectives
    between them: the launches from one collective to the next are captured into a graph each, the
    collectives themselves are recorded as calls and issued eagerly at replay, in the same order on
    every rank.  (A graph cannot hold a collective that is staged through the host, and keeping RCCL
    calls outside the graphs makes the replayed step independent of the communicator in use.)"""

    class _Recorder:
        def __init__(self, owner, comm):
            self._o, self._c = owner, comm
            self.size, self.rank = comm.size, comm.rank

        def __getattr__(self, name):
            fn = getattr(self._c, name)

            def call(*args):
                self._o._close()
                self._o.items.append((fn, args))
                self._o._open()
            return call

    def __init__(self, step_fn, engine):
        self.items = []
        self._ctx = self._g = None
        comm = engine.comm
        engine.comm = self._Recorder(self, comm)
        try:
            self._open()
            step_fn()
            self._close()
        finally:
            engine.comm = comm

    def _open(self):
        self._g = torch.cuda.CUDAGraph()
        self._ctx = torch.cuda.graph(self._g, capture_error_mode="thread_local")
        self._n0 = H.CALLS
        self._ctx.__enter__()

    def _close(self):
        self._ctx.__exit__(None, None, None)
        if H.CALLS > self._n0:                  # segments without a launch (two collectives in a row) are dropped
            self.items.append((self._g, None))
        self._g = self._ctx = None

    def replay(self):
        for item, args in self.items:
            if args is None:
                item.replay()
            else:
                item(*args)


class DenoiseLoop:
    """The body of the reference's `for t in timesteps` loop (x2x_gen_unified.py:474-492)
    for one latent, as graph-replayable device work."""

    def __init__(self, models_by_step, lat, ctx_pairs, seq_len, scheduler, guide_scale, use_graph=True,
                 comm=None):
        """models_by_step: list (one per step) of the UnifiedWan model serving that step;
        ctx_pairs: {id(model): (mixed_cond, mixed_uncond)} context tensors [n, text_dim]."""
        self.lat = lat
        self.sched = scheduler
        self.guide = float(guide_scale)
        self.use_graph = use_graph
        self.models_by_step = models_by_step
        dev = lat.device
        wm0 = models_by_step[0].wan_model
        pt, ph, pw = wm0.patch_size
        self.grid = (lat.shape[1] // pt, lat.shape[2] // ph, lat.shape[3] // pw)
        self.out = torch.empty(2, wm0.out_dim, *lat.shape[1:], dtype=F32, device=dev)
        self.step_ptr = torch.zeros(1, dtype=torch.int32, device=dev)
        self.state = scheduler.prepare(lat)
        self.done = 0
        self.engines, self.caches, self.graphs = {}, {}, {}
        for m in models_by_step:
            if id(m) in self.engines:
                continue
            eng = m.wan_model.engine()
            eng.plan(self.grid, seq_len, 2, comm=comm)
            self.engines[id(m)] = eng
            self.caches[id(m)] = eng.encode_context(list(ctx_pairs[id(m)]), m.wan_model.text_len)

    def _step(self, key):
        eng = self.engines[key]
        eng.forward(self.lat, self.state["tsteps"], self.caches[key], self.out, step_ptr=self.step_ptr)
        self.sched.step_cfg(self.out[0], self.out[1], self.guide, self.lat, step_ptr=self.step_ptr)
        H.step_advance(self.step_ptr)

    def _snapshot(self):
        st = self.state
        bufs = [self.lat, self.step_ptr] + [st[k] for k in ("last", "m1", "m2") if k in st]
        return [(b, b.clone()) for b in bufs]

    def prepare(self):
        """Captures the step graph of every expert now (instead of at each expert's first step): one
        eager step to warm lazy initialisation, then the capture.  The latents, the sampler history and
        the step counter are restored afterwards, so the run that follows is unaffected."""
        if not self.use_graph:
            return
        for key in self.engines:
            if key in self.graphs:
                continue
            saved = self._snapshot()
            self._step(key)
            for b, c in saved:
                b.copy_(c)
            torch.cuda.synchronize()
            comm = self.engines[key].comm
            if comm is not None and not getattr(comm, "capturable", False):
                # sequence parallel over a collective library: graphs between the collectives, collectives eager
                g = _SegmentedGraph(lambda: self._step(key), self.engines[key])
            else:
                g = torch.cuda.CUDAGraph()
                # thread_local: other threads of the process (e.g. a process group's watchdog polling its
                # events) must not be able to invalidate the capture
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self._step(key)
            self.graphs[key] = g

    def seek(self, i):
        """Continue from step i of the schedule (the device-side step counter follows)."""
        self.done = int(i)
        self.step_ptr.fill_(int(i))

    def step_once(self, i):
        key = id(self.models_by_step[i])
        if not self.use_graph:
            self._step(key)
            return
        if key not in self.graphs:
            self.prepare()
        self.graphs[key].replay()

    def run(self, num_steps=None):
        """Runs the next `num_steps` steps (default: all that remain)."""
        end = len(self.models_by_step) if num_steps is None else self.done + num_steps
        while self.done < end:
            self.step_once(self.done)
            self.done += 1
        return self.lat


class OmniVideoX2XUnified:

    def __init__(self, config, checkpoint_dir, vlm_in_dim=4096, device_id=0, rank=0, t5_fsdp=False, dit_fsdp=False,
                 use_usp=False, sp_size=None, t5_cpu=False, use_visual_context_adapter=False,
                 visual_context_adapter_patch_size=(1, 4, 4), max_context_len=None, init_on_cpu=True,
                 wan_config=None, fsdp_shard_size=None, *, linear_precision="bf16"):
        """linear_precision (extension): "bf16" = the reference's arithmetic; "fp8" runs the six large
        linear layers of every DiT block on e4m3 operands (WanModel.linear_precision)."""
        self.linear_precision = linear_precision
        self.device = torch.device(f"cuda:{device_id}")
        self.config, self.rank, self.t5_cpu, self.dit_fsdp, self.init_on_cpu = config, rank, t5_cpu, dit_fsdp, init_on_cpu
        if dit_fsdp or t5_fsdp:
            raise NotImplementedError("FSDP sharding is not part of this build: both experts fit one MI355X")
        self.num_train_timesteps = config.num_train_timesteps
        self.boundary = config.get("boundary", 0.875)
        self.param_dtype = config.param_dtype
        self.vae_stride, self.patch_size = config.vae_stride, config.patch_size
        self.text_encoder = self._build_text_encoder(config, checkpoint_dir)     # None without its files
        self.vae = self._build_vae(os.path.join(checkpoint_dir, config.vae_checkpoint) if checkpoint_dir else None)
        self._init_sequence_parallel(use_usp, sp_size)
        self._build_models(wan_config, checkpoint_dir, vlm_in_dim, use_visual_context_adapter,
                           visual_context_adapter_patch_size, max_context_len)
        if self.use_usp:
            self._apply_usp({id(m): m for m in (self.high_noise_model, self.low_noise_model)}.values())
        self.sample_neg_prompt = config.sample_neg_prompt_en
        self.special_tokens = self._load_special_tokens(checkpoint_dir)
        self.use_graph = True

    # ------------------------------------------------------------------ construction helpers
    def _build_text_encoder(self, config, checkpoint_dir):
        """reference :79-86: umT5-XXL from `t5_checkpoint` / `t5_tokenizer` under checkpoint_dir."""
        if not checkpoint_dir:
            return None
        ckpt = os.path.join(checkpoint_dir, config.t5_checkpoint)
        tok = os.path.join(checkpoint_dir, config.t5_tokenizer)
        if not (os.path.exists(ckpt) and os.path.isdir(tok)):
            logging.info("no umT5 checkpoint/tokenizer under checkpoint_dir: pass precomputed_context tensors")
            return None
        from .modules.t5 import T5EncoderModel
        return T5EncoderModel(text_len=config.text_len, dtype=config.t5_dtype, device=self.device,
                              checkpoint_path=ckpt, tokenizer_path=tok)

    def _build_vae(self, vae_pth):
        try:
            from .modules.vae2_1 import Wan2_1_VAE
        except ImportError:
            return None
        if vae_pth is None or not os.path.exists(vae_pth):
            logging.info("no VAE checkpoint found: generate() will return latents unless self.vae is set")
            return None
        return Wan2_1_VAE(vae_pth=vae_pth, device=self.device)

    def _init_sequence_parallel(self, use_usp, sp_size):
        self.use_usp = use_usp
        if use_usp:
            from .distributed.util import get_rank, get_sequence_parallel_size, init_sequence_parallel_group
            self.sp_group = init_sequence_parallel_group(self.config.get("sp_size", 1) if sp_size is None else sp_size)
            self.sp_size = get_sequence_parallel_size()
            self.sp_rank = get_rank(self.sp_group)
            from .modules.engine import make_group_comm
            with torch.cuda.device(self.device):      # a peer-mapped communicator allocates its buffers here
                self.sp_comm = make_group_comm(self.sp_group) if self.sp_size > 1 else None
        else:
            self.sp_group, self.sp_size, self.sp_rank, self.sp_comm = None, 1, 0, None

    def _apply_usp(self, models):
        """Binds the sequence-parallel forwards as the reference does (x2x_gen_unified.py:172-182)."""
        import types
        from .distributed.full_sequence_parallel import sp_attn_forward, sp_cross_attn_forward, sp_dit_forward
        for m in models:
            m.wan_model.sp_group = self.sp_group
            for block in m.wan_model.blocks:
                block.self_attn.sp_group = self.sp_group
                block.cross_attn.sp_group = self.sp_group
                block.sp_group = self.sp_group
                block.self_attn.forward = types.MethodType(sp_attn_forward, block.self_attn)
                block.cross_attn.forward = types.MethodType(sp_cross_attn_forward, block.cross_attn)
            m.wan_model.forward = types.MethodType(sp_dit_forward, m.wan_model)

    def _new_model(self, wan_config, vlm_in_dim, use_vca, vca_patch, max_context_len, checkpoint_dir=None,
                   subfolder=None):
        """wan_config given: random-init model of that configuration, to be filled by load_state_dict (the
        route of the reference CLI, tools/inference/generate_omni_v2v.py:495-557; reference :112-139).
        wan_config None: the DiT is read from the diffusers-layout directory `checkpoint_dir/subfolder`
        (reference :140-169 -> modules/unified_model.py:42-43); the condition builder's own modules keep their
        initial values until a checkpoint is loaded into them, as in the reference."""
        kw = dict(vlm_in_dim=vlm_in_dim, precision_dtype=self.param_dtype, use_usp=self.use_usp,
                  use_visual_context_adapter=use_vca, visual_context_adapter_patch_size=vca_patch,
                  max_context_len=max_context_len)
        if wan_config is None:
            if not checkpoint_dir:
                raise ValueError("without wan_config the models are read from checkpoint_dir: pass one of the two")
            m = UnifiedWanWithMixedConditionModel.from_pretrained(wan_ckpt_dir=checkpoint_dir, wan_subfolder=subfolder, **kw)
        else:
            m = UnifiedWanWithMixedConditionModel(wan_config=wan_config, skip_init=True, **kw)
        m = m.to(self.param_dtype).to(self.device).eval()
        m.wan_model.linear_precision = self.linear_precision
        return m

    def _build_models(self, wan_config, checkpoint_dir, vlm_in_dim, use_vca, vca_patch, max_context_len):
        cfg = self.config
        self.high_noise_model = self._new_model(wan_config, vlm_in_dim, use_vca, vca_patch, max_context_len, checkpoint_dir,
                                                cfg.get("high_noise_checkpoint", "high_noise_model"))
        self.low_noise_model = self._new_model(wan_config, vlm_in_dim, use_vca, vca_patch, max_context_len, checkpoint_dir,
                                               cfg.get("low_noise_checkpoint", "low_noise_model"))

    def _load_special_tokens(self, checkpoint_dir):
        """The delimiter-token embeddings (reference :215-226).  The reference unpickles
        `special_tokens.pkl`; unpickling executes code from the file, so here the dictionary is read
        with a loader that executes nothing: `special_tokens.pt` (a `torch.save` of the same dict) if
        present, else `special_tokens.pkl` if it happens to be a torch archive.  A raw pickle is
        refused with the one-line conversion its owner has to run once."""
        if not checkpoint_dir:
            return None
        pt, pkl = (os.path.join(checkpoint_dir, n) for n in ("special_tokens.pt", "special_tokens.pkl"))
        path = pt if os.path.exists(pt) else pkl if os.path.exists(pkl) else None
        if path is None:
            return None
        try:
            toks = torch.load(path, map_location="cpu", weights_only=True)
        except Exception as err:      # raw pickle: "Invalid magic number" / unsupported global
            raise RuntimeError(
                f"{path} is not a torch.save archive that loads with weights_only=True ({type(err).__name__}: "
                f"{str(err).splitlines()[0][:120]}).  This build does not unpickle checkpoint files.  Re-save "
                "the dictionary once, in an environment you trust the file in: "
                "python -c \"import pickle, torch; torch.save(pickle.load(open('special_tokens.pkl','rb')), "
                "'special_tokens.pt')\" - special_tokens.pt is looked for first.") from err
        if not isinstance(toks, dict):
            raise TypeError("special_tokens should be a dictionary")
        return {k: v.to(self.param_dtype).to(self.device) for k, v in toks.items()}

    def _prepare_model_for_timestep(self, t, boundary, offload_model=False):
        """reference :228-258 without the offload traffic: both experts stay on the device."""
        return self.high_noise_model if float(t) >= boundary else self.low_noise_model

    # ------------------------------------------------------------------ generate
    def generate(self, input_prompt, precomputed_context=None, ar_vision_input=None, visual_emb=None,
                 size=(1280, 720), frame_num=81, shift=5.0, sample_solver="unipc", sampling_steps=50,
                 guide_scale=5.0, n_prompt="", seed=-1, precision_dtype=torch.bfloat16, offload_model=True,
                 classifier_free_ratio=0.0, unconditioned_context=None, condition_mode="auto",
                 precomputed_context_null=None, noise=None, return_latents=False):
        """reference :261-540.  Extra keyword-only extensions: `precomputed_context_null`
        (negative-prompt embedding when no text encoder is attached), `noise` (use these
        initial latents instead of drawing them on the device), `return_latents`."""
        if precision_dtype != torch.bfloat16:
            raise NotImplementedError("the HIP path computes in bf16 (the reference's GPU setting)")
        if sample_solver not in ("unipc", "dpm++"):
            raise NotImplementedError("Unsupported solver.")
        if condition_mode not in ("auto", "full"):
            raise NotImplementedError("only condition_mode 'auto'/'full' reaches the DiT in the reference")
        F_ = frame_num
        z_dim = self.high_noise_model.wan_model.in_dim
        target_shape = (z_dim, (F_ - 1) // self.vae_stride[0] + 1, size[1] // self.vae_stride[1],
                        size[0] // self.vae_stride[2])
        seq_len = math.ceil((target_shape[2] * target_shape[3]) / (self.patch_size[1] * self.patch_size[2]) *
                            target_shape[1] / self.sp_size) * self.sp_size
        if n_prompt == "":
            n_prompt = self.sample_neg_prompt
        seed = seed if seed >= 0 else random.randint(0, sys.maxsize)
        seed_g = torch.Generator(device=self.device)
        seed_g.manual_seed(seed)

        def to2d(x):
            if x is None:
                return None
            if isinstance(x, list) and len(x) == 1:
                x = x[0]
            if not isinstance(x, torch.Tensor):
                x = torch.tensor(x)
            if x.dim() == 3 and x.size(0) == 1:
                x = x.squeeze(0)
            return x

        if precomputed_context is not None:
            context = [to2d(precomputed_context).to(self.device, dtype=precision_dtype)]
        elif self.text_encoder is not None:
            context = [t.to(self.device) for t in self.text_encoder([input_prompt], self.device)]
        else:
            raise ValueError("no text encoder attached: pass precomputed_context")
        if precomputed_context_null is not None:
            context_null = [to2d(precomputed_context_null).to(self.device, dtype=precision_dtype)]
        elif self.text_encoder is not None:
            context_null = [t.to(self.device) for t in self.text_encoder([n_prompt], self.device)]
        else:
            raise ValueError("no text encoder attached: pass precomputed_context_null for the negative prompt")
        if self.use_usp and dist.is_initialized() and self.sp_size > 1:
            src = self.rank - self.sp_rank
            for ctx in (context, context_null):
                for i in range(len(ctx)):
                    dist.broadcast(ctx[i], src=src, group=self.sp_group)

        if noise is None:
            noise = torch.randn(*target_shape, dtype=precision_dtype, device=self.device, generator=seed_g)
        lat = noise.to(self.device, precision_dtype).contiguous().clone()
        assert tuple(lat.shape) == tuple(target_shape)

        with torch.no_grad():
            boundary = self.boundary * self.num_train_timesteps
            if sample_solver == "unipc":        # reference :409-416
                sched = FlowUniPCMultistepScheduler(num_train_timesteps=self.num_train_timesteps, shift=1,
                                                    use_dynamic_shifting=False)
                sched.set_timesteps(sampling_steps, device="cpu", shift=shift)
            else:                               # reference :417-426
                sched = FlowDPMSolverMultistepScheduler(num_train_timesteps=self.num_train_timesteps, shift=1,
                                                        use_dynamic_shifting=False)
                retrieve_timesteps(sched, device="cpu", sigmas=get_sampling_sigmas(sampling_steps, shift))
                # DPM-Solver++ samples are float32 after the first step (fm_solvers.py:793); the
                # bf16 noise converts exactly, and the DiT rounds its input to bf16 either way
                lat = lat.float()
            # null AR vision: two rows of 1e-6 (reference :433-449)
            ar_null = None
            if ar_vision_input is not None:
                vlm_dim = ar_vision_input[0].size(-1) if isinstance(ar_vision_input, list) else ar_vision_input.size(-1)
                if isinstance(unconditioned_context, dict) and "uncond_ar_vision" in unconditioned_context:
                    unc = unconditioned_context["uncond_ar_vision"]
                else:
                    unc = torch.zeros(2, vlm_dim, device=self.device, dtype=precision_dtype) + 1e-6
                ar_null = [unc for _ in ar_vision_input] if isinstance(ar_vision_input, list) else unc
            models = [self._prepare_model_for_timestep(t, boundary) for t in sched.timesteps.tolist()]
            pairs = {}
            for m in models:
                if id(m) not in pairs:
                    c = m.build_mixed_context(1, context, ar_vision_input, visual_emb, self.special_tokens)[0]
                    u = m.build_mixed_context(1, context_null, ar_null, visual_emb, self.special_tokens)[0]
                    pairs[id(m)] = (c, u)
            # with sequence parallelism the step replays as graphs between eager collectives (one graph with
            # the peer-mapped communicator)
            loop = DenoiseLoop(models, lat, pairs, seq_len, sched, guide_scale, use_graph=self.use_graph,
                               comm=self.sp_comm)
            lat = loop.run()
            if hasattr(self.sp_comm, "check"):
                self.sp_comm.check()             # a peer-mapped barrier that gave up waiting invalidates the run
            if dist.is_initialized():
                dist.barrier()
            is_decode_rank = (self.sp_rank == 0) if self.use_usp else (self.rank == 0)
            if return_latents or self.vae is None:
                return lat if is_decode_rank else None
            if self.sp_comm is not None and lat.shape[2] >= self.sp_comm.size:
                # the reference decodes on the first rank of the group while the others wait
                # (x2x_gen_unified.py:512-524); here every rank decodes a strip of rows and all of
                # them receive the video, which only the first one returns
                videos = self.vae.decode([lat], comm=self.sp_comm)
            else:
                videos = self.vae.decode([lat]) if is_decode_rank else [None]
        if dist.is_initialized():
            dist.barrier()
        return videos[0] if is_decode_rank else None
