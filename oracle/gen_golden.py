"""TEST INFRASTRUCTURE ONLY - writes tests/golden/*.npz by running the REFERENCE's own
modules (loaded by oracle/ref_loader.py from /root/reference) on CPU in float32.

Only outputs are stored: weights and inputs are regenerated from their names by
omnivideo.utils.synthetic, so the fixtures stay small.  Run from the repo root:

    python oracle/gen_golden.py [dit] [loop] [unipc] [dpm] [t5] [vae] [resample] [frames]

The reference tree exists only in the build container; the committed .npz files are
what travels.
"""
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "omni-video_amd"))

from omnivideo.utils import synthetic as syn      # noqa: E402
from oracle import ref_loader                     # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
warnings.filterwarnings("ignore", message=".*Disabling autocast.*")


def tiny_inputs(seed=1):
    cfg = syn.dit_config("tiny")
    return dict(
        lat=syn.randn("in.lat", (16, 3, 8, 12), seed),
        text=syn.randn("in.text", (10, cfg["text_dim"]), seed),
        text_null=syn.randn("in.text_null", (6, cfg["text_dim"]), seed),
        ar=syn.randn("in.ar", (1, 7, cfg["vlm_in_dim"]), seed),
        visual=syn.randn("in.visual", (1, 16, 3, 8, 12), seed),
    )


def build_ref_unified(ns, cfg, sd, max_context_len):
    m = ns.unified_model.UnifiedWanWithMixedConditionModel(
        wan_config=dict(cfg), vlm_in_dim=cfg["vlm_in_dim"], use_visual_context_adapter=True,
        visual_context_adapter_patch_size=cfg["visual_patch_size"],
        max_context_len=max_context_len, eps=cfg["eps"], skip_init=True)
    missing, unexpected = m.load_state_dict(sd, strict=True)
    return m.eval()


def gen_dit():
    ns = ref_loader.load()
    cfg = syn.dit_config("tiny")
    sd = syn.make_unified_state_dict(cfg, seed=0)
    special = syn.make_special_tokens(cfg, seed=0)
    inp = tiny_inputs()
    out = {}
    with torch.no_grad():
        m = build_ref_unified(ns, cfg, sd, cfg["text_len"])
        # (a) text only, no delimiters, seq_len == token count
        out["text_only"] = m([inp["lat"]], t=torch.tensor([731]), context=[inp["text"]],
                             seq_len=72)[0]
        # (b) everything + delimiter tokens + padded sequence (seq_len > f*h*w)
        out["full_special_pad"] = m([inp["lat"]], t=torch.tensor([999]), context=[inp["text"]],
                                    ar_vision_input=[inp["ar"][0]], visual_emb=inp["visual"],
                                    special_token_dict=special, seq_len=80)[0]
        # (c) null branch: negative text, 2-row 1e-6 AR stand-in (x2x_gen_unified.py:444)
        ar_null = torch.zeros(2, cfg["vlm_in_dim"]) + 1e-6
        out["null_branch"] = m([inp["lat"]], t=torch.tensor([208]), context=[inp["text_null"]],
                               ar_vision_input=[ar_null], visual_emb=inp["visual"],
                               special_token_dict=special, seq_len=72)[0]
        # (d) truncation: context longer than max_context_len
        m30 = build_ref_unified(ns, cfg, sd, 30)
        out["truncated_ctx30"] = m30([inp["lat"]], t=torch.tensor([500]), context=[inp["text"]],
                                     ar_vision_input=[inp["ar"][0]], visual_emb=inp["visual"],
                                     special_token_dict=special, seq_len=72)[0]
        # per-operator probes
        wm = m.wan_model
        C, N = cfg["dim"], cfg["num_heads"]
        q = syn.randn("probe.q", (1, 80, N, C // N), 3)
        grid = torch.tensor([[3, 4, 6]])
        out["rope_apply"] = ns.model.rope_apply(q, grid, wm.freqs)[0]
        out["sinusoid_731"] = ns.model.sinusoidal_embedding_1d(256, torch.tensor([731.]))[0]
        x = syn.randn("probe.x", (1, 80, C), 3)
        blk = wm.blocks[1]
        out["rmsnorm"] = blk.self_attn.norm_q(x)[0]
        out["self_attn"] = blk.self_attn(x, torch.tensor([72]), grid, wm.freqs)[0]
        ctxe = syn.randn("probe.ctx", (1, 48, C), 3)
        out["cross_attn"] = blk.cross_attn(x, ctxe, None)[0]
        e0 = syn.randn("probe.e0", (1, 1, 6, C), 3, 0.3).expand(1, 80, 6, C)
        out["block"] = blk(x, e0, torch.tensor([72]), grid, wm.freqs, ctxe, None)[0]
        e = syn.randn("probe.e", (1, 1, C), 3, 0.3).expand(1, 80, C)
        out["head"] = wm.head(x, e)[0]
        out["visual_adapter"] = m.visual_context_adapter(inp["visual"])[0]
        out["vlm_proj"] = m.vlm_proj(m.vlm_norm(inp["ar"][0]))
    np.savez_compressed(os.path.join(GOLD, "dit_tiny.npz"),
                        **{k: v.detach().numpy() for k, v in out.items()})
    print("dit_tiny.npz:", {k: tuple(v.shape) for k, v in out.items()})


def linear_model(lat, t, which):
    """Deterministic stand-in for the DiT in sampler trajectories."""
    a = 0.9 if which == "c" else 0.7
    return (a * lat.float() + 0.001 * float(t) * torch.sin(lat.float() * 3)).float()


def gen_unipc():
    ns = ref_loader.load()
    out = {}
    for steps, shift in ((20, 5.0), (40, 5.0), (50, 5.0), (50, 12.0)):
        s = ns.unipc.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1,
                                                 use_dynamic_shifting=False)
        s.set_timesteps(steps, device="cpu", shift=shift)
        out[f"timesteps_{steps}_{shift:g}"] = s.timesteps.numpy()
        out[f"sigmas_{steps}_{shift:g}"] = s.sigmas.numpy()
    for dt_name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        for steps in (20, 3):
            s = ns.unipc.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1,
                                                     use_dynamic_shifting=False)
            s.set_timesteps(steps, device="cpu", shift=5.0)
            lat = syn.randn("unipc.noise", (16, 2, 4, 6), 7).to(dt)
            traj = []
            for t in s.timesteps:
                c = linear_model(lat, t, "c")
                u = linear_model(lat, t, "u")
                v = u + 3.0 * (c - u)
                lat = s.step(v.unsqueeze(0), t, lat.unsqueeze(0), return_dict=False)[0].squeeze(0)
                traj.append(lat.float().clone())
            out[f"traj_{dt_name}_{steps}"] = torch.stack(traj).numpy()
    np.savez_compressed(os.path.join(GOLD, "unipc.npz"), **out)
    print("unipc.npz:", {k: tuple(v.shape) for k, v in out.items()})


def gen_dpm():
    ns = ref_loader.load()
    out = {}
    for steps, shift in ((20, 5.0), (40, 5.0), (50, 5.0), (50, 12.0)):
        s = ns.dpm.FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
        ts, _ = ns.dpm.retrieve_timesteps(s, device="cpu", sigmas=ns.dpm.get_sampling_sigmas(steps, shift))
        out[f"timesteps_{steps}_{shift:g}"] = ts.numpy()
        out[f"sigmas_{steps}_{shift:g}"] = s.sigmas.numpy()
    for dt_name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        for steps in (20, 3, 1):
            s = ns.dpm.FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
            ts, _ = ns.dpm.retrieve_timesteps(s, device="cpu", sigmas=ns.dpm.get_sampling_sigmas(steps, 5.0))
            lat = syn.randn("dpm.noise", (16, 2, 4, 6), 7).to(dt)
            traj = []
            for t in ts:
                c = linear_model(lat, t, "c")
                u = linear_model(lat, t, "u")
                v = u + 3.0 * (c - u)
                lat = s.step(v.unsqueeze(0), t, lat.unsqueeze(0), return_dict=False)[0].squeeze(0)
                assert lat.dtype == torch.float32
                traj.append(lat.float().clone())
            out[f"traj_{dt_name}_{steps}"] = torch.stack(traj).numpy()
    np.savez_compressed(os.path.join(GOLD, "dpm.npz"), **out)
    print("dpm.npz:", {k: tuple(v.shape) for k, v in out.items()})


def gen_loop():
    """The reference's denoising loop (x2x_gen_unified.py:474-492) restated line by line around the reference's
    OWN modules - UnifiedWanWithMixedConditionModel for both guidance branches, FlowUniPCMultistepScheduler
    constructed and stepped as at :409-416 and :486-491 - on CPU in fp32 (`generate` itself hard-codes a CUDA
    device and generator, :48,339).  Also WanLayerNorm with and without its affine parameters (model.py:87-97)."""
    ns = ref_loader.load()
    cfg = syn.dit_config("tiny")
    sd = syn.make_unified_state_dict(cfg, seed=0)
    special = syn.make_special_tokens(cfg, seed=0)
    inp = tiny_inputs()
    out = {}
    with torch.no_grad():
        m = build_ref_unified(ns, cfg, sd, cfg["text_len"])
        sched = ns.unipc.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
        sched.set_timesteps(4, device="cpu", shift=5.0)
        ar_null = torch.zeros(2, cfg["vlm_in_dim"]) + 1e-6
        arg_c = dict(context=[inp["text"]], ar_vision_input=[inp["ar"][0]], visual_emb=inp["visual"],
                     special_token_dict=special, seq_len=72)
        arg_null = dict(context=[inp["text_null"]], ar_vision_input=[ar_null], visual_emb=inp["visual"],
                        special_token_dict=special, seq_len=72)
        latents = [syn.randn("gen.noise", (16, 3, 8, 12), 11)]
        traj = []
        for t in sched.timesteps:
            timestep = torch.stack([t])
            noise_pred_cond = m(latents, t=timestep, **arg_c)[0]
            noise_pred_uncond = m(latents, t=timestep, **arg_null)[0]
            noise_pred = noise_pred_uncond + 3.0 * (noise_pred_cond - noise_pred_uncond)
            temp_x0 = sched.step(noise_pred.unsqueeze(0), t, latents[0].unsqueeze(0), return_dict=False)[0]
            latents = [temp_x0.squeeze(0)]
            traj.append(latents[0].clone())
        out["cfg_unipc_traj_4"] = torch.stack(traj)
        out["timesteps_4"] = sched.timesteps.clone()
        x = syn.randn("probe.x", (1, 80, cfg["dim"]), 3)
        ln = ns.model.WanLayerNorm(cfg["dim"], eps=cfg["eps"], elementwise_affine=False)
        out["layernorm_plain"] = ln(x)[0]
        out["layernorm_affine"] = m.wan_model.blocks[1].norm3(x)[0]
    np.savez_compressed(os.path.join(GOLD, "loop_tiny.npz"), **{k: v.detach().numpy() for k, v in out.items()})
    print("loop_tiny.npz:", {k: tuple(v.shape) for k, v in out.items()})


def gen_t5():
    """Outputs of the reference T5Encoder (tiny configuration, shared_pos=False, eval mode) in
    float32 and with the module cast to bf16, for two batches of token ids."""
    ns = ref_loader.load()
    cfg = syn.t5_config("tiny")
    sd = syn.make_t5_state_dict(cfg, 0)
    out = {}
    for name, (B, L, lens) in {"short": (2, 40, [40, 23]), "long": (1, 300, [277])}.items():
        ids, mask = syn.make_t5_ids(cfg, B, L, lens, seed=3)
        for dt_name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
            m = ns.t5.T5Encoder(vocab=cfg["vocab_size"], dim=cfg["dim"], dim_attn=cfg["dim_attn"],
                                dim_ffn=cfg["dim_ffn"], num_heads=cfg["num_heads"], num_layers=cfg["num_layers"],
                                num_buckets=cfg["num_buckets"], shared_pos=False, dropout=0.1)
            m.load_state_dict(sd, strict=True)
            m = m.to(dt).eval()
            with torch.no_grad():
                out[f"{name}_{dt_name}"] = m(ids, mask).float().numpy()
    np.savez_compressed(os.path.join(GOLD, "t5_tiny.npz"), **out)
    print("t5_tiny.npz:", {k: tuple(v.shape) for k, v in out.items()})


def gen_frames():
    """Pillow itself (the library torchvision's Resize calls on PIL images) on seeded frames:
    crop with the oracle's box, Image.resize(BILINEAR); only the resized bytes are stored."""
    from PIL import Image
    from oracle import video as V
    out = {}
    for name, (T, Hh, W, th, tw) in syn.FRAME_CASES.items():
        frames = syn.make_frames(name, T, Hh, W)
        top, left, ch, cw = V.center_crop_box(Hh, W, (th, tw))
        res = [np.asarray(Image.fromarray(f[top:top + ch, left:left + cw]).resize((tw, th), Image.BILINEAR))
               for f in frames]
        out[name] = np.stack(res)
    np.savez_compressed(os.path.join(GOLD, "frames.npz"), **out)
    print("frames.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    which = sys.argv[1:] or ["dit", "loop", "unipc", "dpm", "t5", "vae", "resample", "frames"]
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    if "dit" in which:
        gen_dit()
    if "loop" in which:
        gen_loop()
    if "unipc" in which:
        gen_unipc()
    if "dpm" in which:
        gen_dpm()
    if "t5" in which:
        gen_t5()
    if "frames" in which:
        gen_frames()
    if "vae" in which:
        from oracle import gen_golden_vae
        gen_golden_vae.main()
    if "resample" in which:
        from oracle import gen_golden_vae
        gen_golden_vae.resample_probes()
