"""The fused guidance + UniPC kernel and the whole denoising loop (eager and graph-replayed)
against the oracle sampler driving the oracle DiT."""
import os

import pytest
import torch

from conftest import check_rel

from omnivideo.utils import synthetic as syn
from oracle import dit as O
from oracle import dpm as D
from oracle import unipc as U

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32


def linear_model(lat, t, which):
    a = 0.9 if which == "c" else 0.7
    return (a * lat.float() + 0.001 * float(t) * torch.sin(lat.float() * 3)).float()


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("steps", [20, 3, 1])
def test_fused_step_kernel_matches_oracle(dt, steps):
    from omnivideo.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler
    s = FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
    s.set_timesteps(steps, device="cuda", shift=5.0)
    orc = U.UniPC(steps, 5.0)
    x = syn.randn("unipc.noise", (16, 2, 4, 6), 7).to(dt)
    xo, xg = x.clone(), x.cuda().clone()
    for t in orc.timesteps:
        # both sides see the same model outputs (computed from the oracle's state) so the
        # comparison isolates the update rule
        c, u = linear_model(xo, t, "c"), linear_model(xo, t, "u")
        xg.copy_(xo)
        s.step_cfg(c.cuda(), u.cuda(), 3.0, xg)
        xo = orc.step(u + 3.0 * (c - u), xo)
        if dt == F32:
            torch.testing.assert_close(xg.cpu(), xo, rtol=1e-5, atol=1e-6)
        else:
            assert torch.equal(xg.cpu(), xo)


def test_scheduler_step_interface_matches_oracle():
    """`step(model_output, t, sample)` with the reference's calling convention."""
    from omnivideo.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler
    s = FlowUniPCMultistepScheduler()
    s.set_timesteps(5, device="cuda", shift=5.0)
    orc = U.UniPC(5, 5.0)
    x = syn.randn("unipc.noise", (1, 16, 2, 4, 6), 7)
    xo, xg = x.clone(), x.cuda()
    for t in s.timesteps:
        v = linear_model(xo, t, "c")
        xg = s.step(v.cuda(), t, xo.cuda(), return_dict=False)[0]
        xo = orc.step(v, xo)
        torch.testing.assert_close(xg.cpu(), xo, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("steps", [20, 3, 1])
def test_dpm_fused_step_kernel_matches_oracle(dt, steps):
    """Guidance + DPM-Solver++ step in one launch: same float32 operations in the same order as
    the reference's tensor expressions, so the trajectory is bit-identical to the oracle's."""
    from omnivideo.utils.fm_solvers import (FlowDPMSolverMultistepScheduler, get_sampling_sigmas,
                                            retrieve_timesteps)
    s = FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
    ts, n = retrieve_timesteps(s, device="cuda", sigmas=get_sampling_sigmas(steps, 5.0))
    orc = D.DPMpp(steps, 5.0)
    assert n == steps and torch.equal(ts.cpu(), orc.timesteps) and torch.equal(s.sigmas, orc.sigmas)
    xo = syn.randn("dpm.noise", (16, 2, 4, 6), 7).to(dt)
    xg = torch.empty(xo.shape, dtype=F32, device="cuda")
    for i, t in enumerate(orc.timesteps):
        c, u = linear_model(xo, t, "c"), linear_model(xo, t, "u")
        s.step_cfg(c.cuda(), u.cuda(), 3.0, xo.cuda(), out=xg)
        xo = orc.step(u + 3.0 * (c - u), xo)
        assert xo.dtype == F32 and torch.equal(xg.cpu(), xo), i


def test_dpm_scheduler_step_interface_matches_oracle():
    from omnivideo.utils.fm_solvers import FlowDPMSolverMultistepScheduler
    s = FlowDPMSolverMultistepScheduler()
    s.set_timesteps(5, device="cuda", shift=5.0)
    x = syn.randn("dpm.noise", (1, 16, 2, 4, 6), 7).to(BF16)
    with pytest.raises(TypeError, match="fp32"):
        s.step_cfg(x.float().cuda(), x.float().cuda(), 1.0, x.cuda())
    xg = s.step(x.float().cuda(), s.timesteps[0], x.cuda(), return_dict=False)[0]
    assert xg.dtype == F32
    sig = s.sigmas
    x0 = x - sig[0] * x.float()
    h = (torch.log(1 - sig[1]) - torch.log(sig[1])) - (torch.log(1 - sig[0]) - torch.log(sig[0]))
    ref = (sig[1] / sig[0]) * x.float() - ((1 - sig[1]) * (torch.exp(-h) - 1.0)) * x0
    assert torch.equal(xg.cpu(), ref)


def _pipeline(cfg, sd, sp):
    from omnivideo.configs import Config, t2v_1_3B
    from omnivideo.x2x_gen_unified_1_3B import OmniVideoX2XUnified1_3B
    conf = Config(t2v_1_3B)
    pipe = OmniVideoX2XUnified1_3B(conf, None, vlm_in_dim=cfg["vlm_in_dim"], use_visual_context_adapter=True,
                                   visual_context_adapter_patch_size=cfg["visual_patch_size"],
                                   max_context_len=cfg["text_len"], wan_config=dict(cfg))
    pipe.model.load_state_dict(sd, strict=True)
    pipe.model.to(BF16)
    pipe.special_tokens = {k: v.cuda().to(BF16) for k, v in sp.items()}
    return pipe


@pytest.mark.parametrize("use_graph", [False, True])
def test_generate_loop_matches_oracle(use_graph):
    """3-step CFG sampling of the tiny model (V2V-style mixed context) through
    `generate()`: latents vs the oracle loop in its bf16 flow."""
    cfg = syn.dit_config("tiny")
    sd = syn.make_unified_state_dict(cfg, 0)
    sdb = O.round_weights_bf16(sd)
    sp = syn.make_special_tokens(cfg, 0)
    text, text_null = syn.randn("in.text", (10, cfg["text_dim"]), 1), syn.randn("in.text_null", (6, cfg["text_dim"]), 1)
    ar, visual = syn.randn("in.ar", (1, 7, cfg["vlm_in_dim"]), 1), syn.randn("in.visual", (1, 16, 3, 8, 12), 1)
    noise = syn.randn("gen.noise", (16, 3, 8, 12), 11).to(BF16)
    pipe = _pipeline(cfg, sd, sp)
    pipe.use_graph = use_graph
    steps, shift, guide = 4, 5.0, 3.0
    lat = pipe.generate("", precomputed_context=text.to(BF16), precomputed_context_null=text_null.to(BF16),
                        ar_vision_input=[ar[0].cuda().to(BF16)], visual_emb=visual.cuda(), size=(96, 64),
                        frame_num=9, shift=shift, sampling_steps=steps, guide_scale=guide, seed=1, noise=noise,
                        return_latents=True)
    assert lat.dtype == BF16 and tuple(lat.shape) == (16, 3, 8, 12)
    ar_null = torch.zeros(2, cfg["vlm_in_dim"]) + 1e-6
    rbw = lambda t: t.to(BF16).float()
    fc = lambda l, t: O.unified_forward(l.float(), t, sdb, cfg, 72, "bf16", text=rbw(text), ar_vision=rbw(ar),
                                        visual=visual, special=sp)
    fu = lambda l, t: O.unified_forward(l.float(), t, sdb, cfg, 72, "bf16", text=rbw(text_null),
                                        ar_vision=rbw(ar_null), visual=visual, special=sp)
    ref = U.sample_loop(fc, fu, noise.clone(), steps, shift, guide)
    rel = float((lat.float().cpu() - ref.float()).norm() / ref.float().norm())
    check_rel("test_generate_loop_matches_oracle", rel, 6e-3)      # observed 1.9e-3


def test_generate_loop_vs_reference_loop_golden(golden_dir):
    """The same 4-step guided sampling against the trajectory of the REFERENCE's own modules driven as its loop
    drives them (tests/golden/loop_tiny.npz, fp32 on CPU): the distance is that of the bf16 flow from the fp32
    flow accumulated over four steps - the pinned leg of the loop-level parity."""
    import numpy as np
    cfg = syn.dit_config("tiny")
    sd, sp = syn.make_unified_state_dict(cfg, 0), syn.make_special_tokens(cfg, 0)
    text, text_null = syn.randn("in.text", (10, cfg["text_dim"]), 1), syn.randn("in.text_null", (6, cfg["text_dim"]), 1)
    ar, visual = syn.randn("in.ar", (1, 7, cfg["vlm_in_dim"]), 1), syn.randn("in.visual", (1, 16, 3, 8, 12), 1)
    noise = syn.randn("gen.noise", (16, 3, 8, 12), 11)
    pipe = _pipeline(cfg, sd, sp)
    lat = pipe.generate("", precomputed_context=text.to(BF16), precomputed_context_null=text_null.to(BF16),
                        ar_vision_input=[ar[0].cuda().to(BF16)], visual_emb=visual.cuda(), size=(96, 64),
                        frame_num=9, shift=5.0, sampling_steps=4, guide_scale=3.0, seed=1, noise=noise.to(BF16),
                        return_latents=True)
    want = torch.from_numpy(np.load(os.path.join(golden_dir, "loop_tiny.npz"))["cfg_unipc_traj_4"])[-1]
    rel = float((lat.float().cpu() - want).norm() / want.norm())
    check_rel("generate 4 steps vs the reference loop's fp32 golden", rel, 1.4e-2)      # observed 4.5e-3


def test_generate_loop_dpmpp_matches_oracle():
    """`generate(sample_solver="dpm++")` (x2x_gen_unified.py:417-426): float32 latents vs the oracle
    DiT driven by the oracle DPM-Solver++ loop."""
    cfg = syn.dit_config("tiny")
    sd = syn.make_unified_state_dict(cfg, 0)
    sdb = O.round_weights_bf16(sd)
    sp = syn.make_special_tokens(cfg, 0)
    text, text_null = syn.randn("in.text", (10, cfg["text_dim"]), 1), syn.randn("in.text_null", (6, cfg["text_dim"]), 1)
    noise = syn.randn("gen.noise", (16, 3, 8, 12), 11).to(BF16)
    pipe = _pipeline(cfg, sd, sp)
    steps, shift, guide = 4, 5.0, 3.0
    lat = pipe.generate("", precomputed_context=text.to(BF16), precomputed_context_null=text_null.to(BF16),
                        size=(96, 64), frame_num=9, shift=shift, sample_solver="dpm++", sampling_steps=steps,
                        guide_scale=guide, seed=1, noise=noise, return_latents=True)
    assert lat.dtype == F32 and tuple(lat.shape) == (16, 3, 8, 12)
    rbw = lambda t: t.to(BF16).float()
    fc = lambda l, t: O.unified_forward(l.float(), t, sdb, cfg, 72, "bf16", text=rbw(text), special=sp)
    fu = lambda l, t: O.unified_forward(l.float(), t, sdb, cfg, 72, "bf16", text=rbw(text_null), special=sp)
    ref = D.sample_loop(fc, fu, noise.clone(), steps, shift, guide)
    rel = float((lat.cpu() - ref).norm() / ref.norm())
    check_rel("test_generate_loop_dpmpp_matches_oracle", rel, 2e-3)      # observed 6.7e-4
    with pytest.raises(NotImplementedError):
        pipe.generate("", precomputed_context=text.to(BF16), precomputed_context_null=text_null.to(BF16),
                      size=(96, 64), frame_num=9, sample_solver="euler", noise=noise, return_latents=True)


def test_two_expert_switch_at_noise_boundary():
    """The two-model pipeline (x2x_gen_unified.py:228-258,407): the high-noise expert serves the
    steps with t >= 875, the low-noise expert the rest; both stay resident and each replays its
    own captured graph.  Checked against the oracle loop switching weights at the same boundary."""
    from omnivideo.configs import Config, t2v_A14B
    from omnivideo.x2x_gen_unified import OmniVideoX2XUnified
    cfg = syn.dit_config("tiny")
    sds = [syn.make_unified_state_dict(cfg, seed) for seed in (0, 1)]
    sp = syn.make_special_tokens(cfg, 0)
    pipe = OmniVideoX2XUnified(Config(t2v_A14B), None, vlm_in_dim=cfg["vlm_in_dim"], use_visual_context_adapter=True,
                               visual_context_adapter_patch_size=cfg["visual_patch_size"],
                               max_context_len=cfg["text_len"], wan_config=dict(cfg))
    assert pipe.high_noise_model is not pipe.low_noise_model
    pipe.high_noise_model.load_state_dict(sds[0], strict=True)
    pipe.low_noise_model.load_state_dict(sds[1], strict=True)
    pipe.high_noise_model.to(BF16)
    pipe.low_noise_model.to(BF16)
    pipe.special_tokens = {k: v.cuda().to(BF16) for k, v in sp.items()}
    text, text_null = syn.randn("in.text", (10, cfg["text_dim"]), 1), syn.randn("in.text_null", (6, cfg["text_dim"]), 1)
    noise = syn.randn("gen.noise", (16, 3, 8, 12), 11).to(BF16)
    steps, shift, guide = 6, 5.0, 3.0
    _, ts = U.schedule(steps, shift)
    n_high = int((ts >= 875).sum())
    assert 0 < n_high < steps                   # the run really crosses the boundary
    lat = pipe.generate("", precomputed_context=text.to(BF16), precomputed_context_null=text_null.to(BF16),
                        size=(96, 64), frame_num=9, shift=shift, sampling_steps=steps, guide_scale=guide, seed=1,
                        noise=noise, return_latents=True)
    sdb = [O.round_weights_bf16(sd) for sd in sds]
    rbw = lambda t: t.to(BF16).float()
    pick = lambda t: sdb[0] if t >= 875 else sdb[1]
    fc = lambda l, t: O.unified_forward(l.float(), t, pick(t), cfg, 72, "bf16", text=rbw(text), special=sp)
    fu = lambda l, t: O.unified_forward(l.float(), t, pick(t), cfg, 72, "bf16", text=rbw(text_null), special=sp)
    ref = U.sample_loop(fc, fu, noise.clone(), steps, shift, guide)
    rel = float((lat.float().cpu() - ref.float()).norm() / ref.float().norm())
    check_rel("test_two_expert_switch_at_noise_boundary", rel, 7e-3)      # observed 2.2e-3
    # one expert for every step gives a different answer: the switch is observable
    one = U.sample_loop(lambda l, t: O.unified_forward(l.float(), t, sdb[1], cfg, 72, "bf16", text=rbw(text), special=sp),
                        lambda l, t: O.unified_forward(l.float(), t, sdb[1], cfg, 72, "bf16", text=rbw(text_null),
                                                       special=sp), noise.clone(), steps, shift, guide)
    assert float((one.float() - ref.float()).norm() / ref.float().norm()) > 10 * rel


def test_graph_replay_equals_eager_bitwise():
    cfg = syn.dit_config("tiny")
    sd, sp = syn.make_unified_state_dict(cfg, 0), syn.make_special_tokens(cfg, 0)
    text, text_null = syn.randn("in.text", (10, cfg["text_dim"]), 1), syn.randn("in.text_null", (6, cfg["text_dim"]), 1)
    noise = syn.randn("gen.noise", (16, 3, 8, 12), 11).to(BF16)
    outs = []
    for use_graph in (False, True):
        pipe = _pipeline(cfg, sd, sp)
        pipe.use_graph = use_graph
        outs.append(pipe.generate("", precomputed_context=text.to(BF16), precomputed_context_null=text_null.to(BF16),
                                  size=(96, 64), frame_num=9, sampling_steps=6, guide_scale=3.0, seed=1, noise=noise,
                                  return_latents=True).clone())
    assert torch.equal(outs[0], outs[1])


def test_generate_returns_decoded_video_end_to_end():
    """generate() with a VAE attached: sampling loop + decode in one call.  The video equals the VAE
    decode of the latents the same call returns with return_latents=True, and tracks the oracle loop +
    oracle VAE decode."""
    from omnivideo.modules.vae2_1 import Wan2_1_VAE
    from oracle import vae as V
    cfg = syn.dit_config("tiny")
    sd = syn.make_unified_state_dict(cfg, 0)
    sp = syn.make_special_tokens(cfg, 0)
    vcfg = syn.vae_config("tiny")
    vsd = syn.make_vae_state_dict(vcfg, 0)
    text, text_null = syn.randn("in.text", (10, cfg["text_dim"]), 1), syn.randn("in.text_null", (6, cfg["text_dim"]), 1)
    noise = syn.randn("gen.noise", (16, 3, 8, 12), 11).to(BF16)
    pipe = _pipeline(cfg, sd, sp)
    pipe.vae = Wan2_1_VAE(z_dim=vcfg["z_dim"], vae_pth=None, device="cuda", dim=vcfg["dim"])
    pipe.vae.model.load_state_dict(vsd, strict=True)
    kw = dict(precomputed_context=text.to(BF16), precomputed_context_null=text_null.to(BF16), size=(96, 64),
              frame_num=9, shift=5.0, sampling_steps=3, guide_scale=3.0, seed=1, noise=noise)
    video = pipe.generate("", **kw)
    assert video.dtype == F32 and tuple(video.shape) == (3, 9, 64, 96)
    assert float(video.min()) >= -1.0 and float(video.max()) <= 1.0
    lat = pipe.generate("", return_latents=True, **kw)
    assert torch.equal(video, pipe.vae.decode([lat])[0])
    sdb = O.round_weights_bf16(sd)
    rbw = lambda t: t.to(BF16).float()
    fc = lambda l, t: O.unified_forward(l.float(), t, sdb, cfg, 72, "bf16", text=rbw(text))
    fu = lambda l, t: O.unified_forward(l.float(), t, sdb, cfg, 72, "bf16", text=rbw(text_null))
    ref_lat = U.sample_loop(fc, fu, noise.clone(), 3, 5.0, 3.0)
    lay = V.vae_layout(vcfg["dim"], vcfg["z_dim"], vcfg["dim_mult"], vcfg["num_res_blocks"], vcfg["temperal_downsample"])
    with torch.no_grad():
        ref_video = V.decode(ref_lat.float(), vsd, lay).clamp(-1, 1)
    rel = float((video.cpu() - ref_video).norm() / ref_video.norm())
    check_rel("test_generate_returns_decoded_video_end_to_end", rel, 3e-2)      # observed 1.04e-2


def test_v2v_input_path_frames_to_latents_to_generate():
    """The V2V input chain of the reference's CLI (tools/inference/generate_omni_v2v.py:597-640): source
    frames -> crop/resize/normalise -> VAE encode -> visual_emb -> generate(), against the same chain in
    the oracle (Pillow-exact resize, oracle encode, oracle loop)."""
    from omnivideo.modules.vae2_1 import Wan2_1_VAE
    from omnivideo.utils.utils import transform_frames_to_tensor
    from oracle import vae as V
    from oracle import video as OV
    cfg = syn.dit_config("tiny")
    sd, sp = syn.make_unified_state_dict(cfg, 0), syn.make_special_tokens(cfg, 0)
    vcfg = syn.vae_config("tiny")
    vsd = syn.make_vae_state_dict(vcfg, 0)
    lay = V.vae_layout(vcfg["dim"], vcfg["z_dim"], vcfg["dim_mult"], vcfg["num_res_blocks"], vcfg["temperal_downsample"])
    frames = syn.make_frames("v2v.src", 9, 70, 110)
    pipe = _pipeline(cfg, sd, sp)
    pipe.vae = Wan2_1_VAE(z_dim=vcfg["z_dim"], vae_pth=None, device="cuda", dim=vcfg["dim"])
    pipe.vae.model.load_state_dict(vsd, strict=True)
    ft = transform_frames_to_tensor(frames, (64, 96))                       # [T, 3, h, w] on the device
    assert torch.equal(ft.cpu(), OV.transform_frames_to_tensor(frames, (64, 96)))
    latent = pipe.vae.encode([ft.transpose(0, 1)])[0]                        # [16, 3, 8, 12]
    assert tuple(latent.shape) == (16, 3, 8, 12)
    with torch.no_grad():
        ref_latent = V.encode(OV.transform_frames_to_tensor(frames, (64, 96)).transpose(0, 1).contiguous(), vsd, lay)
    torch.testing.assert_close(latent.cpu(), ref_latent, rtol=1e-3, atol=1e-4)
    text, text_null = syn.randn("in.text", (10, cfg["text_dim"]), 1), syn.randn("in.text_null", (6, cfg["text_dim"]), 1)
    noise = syn.randn("gen.noise", (16, 3, 8, 12), 11).to(BF16)
    lat = pipe.generate("", precomputed_context=text.to(BF16), precomputed_context_null=text_null.to(BF16),
                        visual_emb=latent[None], size=(96, 64), frame_num=9, shift=5.0, sampling_steps=3,
                        guide_scale=3.0, seed=1, noise=noise, return_latents=True)
    sdb = O.round_weights_bf16(sd)
    rbw = lambda t: t.to(BF16).float()
    vis = ref_latent[None]
    fc = lambda l, t: O.unified_forward(l.float(), t, sdb, cfg, 72, "bf16", text=rbw(text), visual=vis, special=sp)
    fu = lambda l, t: O.unified_forward(l.float(), t, sdb, cfg, 72, "bf16", text=rbw(text_null), visual=vis, special=sp)
    ref = U.sample_loop(fc, fu, noise.clone(), 3, 5.0, 3.0)
    rel = float((lat.float().cpu() - ref.float()).norm() / ref.float().norm())
    check_rel("test_v2v_input_path_frames_to_latents_to_generate", rel, 6e-3)      # observed 1.8e-3
