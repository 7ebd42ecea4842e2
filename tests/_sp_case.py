"""Inputs and pipelines shared by tests/test_gpu_sp_processes.py (single process) and its workers
(tests/_sp_gpu_worker.py, one process per sequence-parallel rank): the same tiny two-expert pipeline
with a tiny VAE, built from seeds so every process holds identical weights and inputs."""
import torch

from omnivideo.utils import synthetic as syn

BF16 = torch.bfloat16
GEN = dict(size=(96, 64), frame_num=9, shift=5.0, sampling_steps=6, guide_scale=3.0, seed=1)


def build_pipeline(use_usp, sp_size=None, rank=0):
    from omnivideo.configs import Config, t2v_A14B
    from omnivideo.modules.vae2_1 import Wan2_1_VAE
    from omnivideo.x2x_gen_unified import OmniVideoX2XUnified
    cfg = syn.dit_config("tiny")
    vcfg = syn.vae_config("tiny")
    pipe = OmniVideoX2XUnified(Config(t2v_A14B), None, vlm_in_dim=cfg["vlm_in_dim"], rank=rank, use_usp=use_usp,
                               sp_size=sp_size, use_visual_context_adapter=True,
                               visual_context_adapter_patch_size=cfg["visual_patch_size"],
                               max_context_len=cfg["text_len"], wan_config=dict(cfg))
    for m, seed in ((pipe.high_noise_model, 0), (pipe.low_noise_model, 1)):
        m.load_state_dict(syn.make_unified_state_dict(cfg, seed), strict=True)
        m.to(BF16)
    pipe.special_tokens = {k: v.cuda().to(BF16) for k, v in syn.make_special_tokens(cfg, 0).items()}
    pipe.vae = Wan2_1_VAE(z_dim=vcfg["z_dim"], vae_pth=None, device="cuda", dim=vcfg["dim"])
    pipe.vae.model.load_state_dict(syn.make_vae_state_dict(vcfg, 0), strict=True)
    return pipe, cfg


def generate_inputs(cfg):
    text = syn.randn("in.text", (10, cfg["text_dim"]), 1).to(BF16)
    text_null = syn.randn("in.text_null", (6, cfg["text_dim"]), 1).to(BF16)
    ar = syn.randn("in.ar", (1, 7, cfg["vlm_in_dim"]), 1)[0].cuda().to(BF16)
    visual = syn.randn("in.visual", (1, 16, 3, 8, 12), 1).cuda()
    noise = syn.randn("gen.noise", (16, 3, 8, 12), 11).to(BF16)
    return dict(precomputed_context=text, precomputed_context_null=text_null, ar_vision_input=[ar],
                visual_emb=visual, noise=noise, **GEN)


def run_generate(pipe, cfg):
    """(latents, video) of the same sampling run; None on ranks that do not return a result."""
    kw = generate_inputs(cfg)
    lat = pipe.generate("", return_latents=True, **kw)
    video = pipe.generate("", **kw)
    return lat, video


def attention_inputs(cfg, B=2, L=80, Lc=48):
    x = syn.randn("sp.x", (B, L, cfg["dim"]), 5).cuda().to(BF16)
    ctx = syn.randn("sp.ctx", (B, Lc, cfg["dim"]), 6).cuda().to(BF16)
    grid = torch.tensor([[3, 4, 6]] * B)
    seq_lens = torch.tensor([72] * B)
    return x, ctx, grid, seq_lens
