#!/usr/bin/env python
"""BASELINE configuration 1 end to end, both ways, on one box: OmniVideo2-1.3B text-to-video, 17 frames at
256x256, 20 UniPC steps, guidance 3.0 (reference: x2x_gen_unified_1_3B.py through generate(),
x2x_gen_unified.py:261-540; samples/t2v_example.jsonl has no checkpoint here, so weights are the seeded
synthetic ones and the text condition is a seeded tensor).

  * CPU leg: the oracle (oracle/dit.py, unipc.py, vae.py; fp32 flow = the reference's CPU path) runs the whole
    job - 20 steps x 2 forwards of the full 30-layer model, then the VAE decode - on the host cores, timed;
  * GPU leg: `OmniVideoX2XUnified1_3B.generate` (HIP kernels, bf16 as the reference's GPU setting) on the
    same weights, noise and conditions, timed.

Prints one JSON object: both wall times, frames/s, and the relative L2 distance of the final latents and of
the decoded videos (bf16 GPU flow against fp32 CPU flow after 20 guided steps).  Not part of the test suite
(minutes of CPU); its output is kept under profiles/."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "omni-video_amd")):
    sys.path.insert(0, p)
import torch                                            # noqa: E402

from omnivideo.utils import synthetic as syn            # noqa: E402
from oracle import dit as O                             # noqa: E402
from oracle import unipc as U                           # noqa: E402
from oracle import vae as V                             # noqa: E402

BF16 = torch.bfloat16


def main(steps=20, threads=16, full_vae=True):
    torch.set_num_threads(threads)
    cfg = syn.dit_config("1.3B", text_len=512)
    sd = syn.make_unified_state_dict(cfg, 0)
    vcfg = syn.vae_config("wan2.1" if full_vae else "tiny")
    vsd = syn.make_vae_state_dict(vcfg, 0)
    lay = V.vae_layout(vcfg["dim"], vcfg["z_dim"], vcfg["dim_mult"], vcfg["num_res_blocks"], vcfg["temperal_downsample"])
    text = syn.randn("cfg1.text", (20, cfg["text_dim"]), 1)
    text_null = syn.randn("cfg1.null", (126, cfg["text_dim"]), 1)
    noise = syn.randn("cfg1.noise", (16, 5, 32, 32), 7).to(BF16)
    shift, guide = 5.0, 3.0

    # ---- GPU leg
    from omnivideo.configs import Config, t2v_1_3B
    from omnivideo.modules.vae2_1 import Wan2_1_VAE
    from omnivideo.x2x_gen_unified_1_3B import OmniVideoX2XUnified1_3B
    pipe = OmniVideoX2XUnified1_3B(Config(t2v_1_3B), None, vlm_in_dim=cfg["vlm_in_dim"], use_visual_context_adapter=True,
                                   visual_context_adapter_patch_size=cfg["visual_patch_size"],
                                   max_context_len=cfg["text_len"], wan_config=dict(cfg))
    pipe.model.load_state_dict(sd, strict=True)
    pipe.model.to(BF16)
    pipe.vae = Wan2_1_VAE(z_dim=vcfg["z_dim"], vae_pth=None, device="cuda", dim=vcfg["dim"])
    pipe.vae.model.load_state_dict(vsd, strict=True)
    kw = dict(precomputed_context=text.to(BF16), precomputed_context_null=text_null.to(BF16), size=(256, 256),
              frame_num=17, shift=shift, sampling_steps=steps, guide_scale=guide, seed=1, noise=noise)
    with torch.no_grad():
        pipe.generate("", **kw)                          # warm: library load, graph capture
        torch.cuda.synchronize()
        t0 = time.time()
        video = pipe.generate("", **kw)
        torch.cuda.synchronize()
        gpu_s = time.time() - t0
        lat = pipe.generate("", return_latents=True, **kw).float().cpu()
    video = video.cpu()

    # ---- CPU leg (fp32 flow, fp32 weights: the reference's CPU path)
    rbw = lambda t: t.to(BF16).float()                   # the same bf16-rounded condition tensors the GPU leg got
    fc = lambda l, t: O.wan_forward(l.float(), t, rbw(text), 1280, sd, cfg, "fp32")
    fu = lambda l, t: O.wan_forward(l.float(), t, rbw(text_null), 1280, sd, cfg, "fp32")
    sdw = {k: v for k, v in sd.items()}
    t0 = time.time()
    with torch.no_grad():
        ref_lat = U.sample_loop(fc, fu, noise.float(), steps, shift, guide)
        loop_s = time.time() - t0
        ref_video = V.decode(ref_lat.float(), vsd, lay).clamp(-1, 1)
    cpu_s = time.time() - t0
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
    print(json.dumps({
        "config": "BASELINE configs[0]: OmniVideo2-1.3B T2V 17 frames 256x256, %d UniPC steps, guidance 3.0, synthetic weights" % steps,
        "cpu": {"kind": "port (oracle, fp32 flow)", "cores": threads, "wall_s": round(cpu_s, 1), "loop_s": round(loop_s, 1),
                "frames_per_s": round(17 / cpu_s, 4)},
        "gpu": {"dtype": "bf16", "wall_s": round(gpu_s, 3), "frames_per_s": round(17 / gpu_s, 2),
                "includes": "context encode, 20 graph-replayed steps, VAE decode (graph capture excluded: warm run first)"},
        "speedup": round(cpu_s / gpu_s, 1),
        "rel_l2_latents_gpu_bf16_vs_cpu_fp32": rel(lat, ref_lat),
        "rel_l2_video_gpu_vs_cpu": rel(video, ref_video),
        "note": "the CPU leg samples in fp32 (the reference's CPU path), the GPU leg in bf16 with bf16 latents re-quantised "
                "every step (the reference's GPU path): the distance is the two flows' divergence over the whole trajectory"}))


if __name__ == "__main__":
    main(steps=int(sys.argv[1]) if len(sys.argv) > 1 else 20)
