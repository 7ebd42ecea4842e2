"""TEST INFRASTRUCTURE ONLY - golden vectors for the causal 3D VAE, produced by the
reference's own WanVAE_ (omnivideo/modules/vae2_1.py) with synthetic weights at a reduced
width (dim=32).  Run through `python oracle/gen_golden.py vae`."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "omni-video_amd"))

from omnivideo.utils import synthetic as syn      # noqa: E402
from oracle import ref_loader                     # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def build(ns):
    V = ns.vae2_1
    cfg = syn.vae_config("tiny")
    sd = syn.make_vae_state_dict(cfg, 0)
    m = V.WanVAE_(dim=cfg["dim"], z_dim=cfg["z_dim"], dim_mult=list(cfg["dim_mult"]),
                  num_res_blocks=cfg["num_res_blocks"], attn_scales=[],
                  temperal_downsample=list(cfg["temperal_downsample"]), dropout=0.0).eval()
    m.load_state_dict(sd, strict=True)
    return m


def resample_probes():
    """The four Resample modes (vae2_1.py:65-159) of the tiny VAE as separate probes, streamed over three
    consecutive chunks each with the module's own feature cache: downsample2d / downsample3d are the encoder's
    (chunks of 1, 4, 2 frames: the first passes the temporal convolution untouched, :142-158), upsample3d /
    upsample2d the decoder's (three chunks of one latent frame: the first stores the 'Rep' sentinel and skips the
    temporal convolution, the second convolves against zero left context, the third against the cached frames,
    :102-136).  -> tests/golden/vae_resample.npz"""
    ns = ref_loader.load()
    V = ns.vae2_1
    m = build(ns)
    mods = {}
    for name, seq in (("encoder.downsamples", m.encoder.downsamples), ("decoder.upsamples", m.decoder.upsamples)):
        for i, layer in enumerate(seq):
            if isinstance(layer, V.Resample):
                mods.setdefault(layer.mode, (f"{name}.{i}", layer))
    assert sorted(mods) == ["downsample2d", "downsample3d", "upsample2d", "upsample3d"], sorted(mods)
    out = {}
    with torch.no_grad():
        for mode, (name, layer) in sorted(mods.items()):
            cin = layer.dim
            chunks = ((0, 1), (1, 5), (5, 7)) if mode.startswith("down") else ((0, 1), (1, 2), (2, 3))
            x = syn.randn(f"vae.resample.{mode}", (1, cin, chunks[-1][1], 6, 8), 5)
            cache, ys = [None], []
            for a, b in chunks:
                ys.append(layer(x[:, :, a:b], cache, [0]))
            out[mode] = torch.cat(ys, 2)[0]
            out[mode + "_module"] = np.array(name)
    np.savez_compressed(os.path.join(GOLD, "vae_resample.npz"), **{k: (v.numpy() if hasattr(v, "numpy") else v)
                                                                   for k, v in out.items()})
    print("vae_resample.npz:", {k: tuple(getattr(v, "shape", ())) for k, v in out.items()})


def main():
    ns = ref_loader.load()
    V = ns.vae2_1
    cfg = syn.vae_config("tiny")
    sd = syn.make_vae_state_dict(cfg, 0)
    m = V.WanVAE_(dim=cfg["dim"], z_dim=cfg["z_dim"], dim_mult=list(cfg["dim_mult"]),
                  num_res_blocks=cfg["num_res_blocks"], attn_scales=[],
                  temperal_downsample=list(cfg["temperal_downsample"]), dropout=0.0).eval()
    m.load_state_dict(sd, strict=True)
    mean = torch.tensor([-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
                         0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921])
    std = torch.tensor([2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
                        3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160])
    scale = [mean, 1.0 / std]
    out = {}
    with torch.no_grad():
        for frames, hw in ((5, (32, 48)), (9, (32, 32)), (1, (16, 16))):
            vid = syn.randn(f"vae.video{frames}", (3, frames, *hw), 5).clamp(-1, 1)
            out[f"encode_{frames}"] = m.encode(vid[None], scale)[0]
        for tl, hw in ((2, (4, 6)), (3, (4, 4)), (1, (2, 2))):
            z = syn.randn(f"vae.z{tl}", (16, tl, *hw), 5)
            out[f"decode_{tl}"] = m.decode(z[None], scale)[0].float().clamp_(-1, 1)
        # streaming probes: one ResidualBlock and the four Resample modes over 3 chunks
        x = syn.randn("vae.probe", (1, 32, 7, 6, 8), 5)
        rb = m.encoder.downsamples[0]
        cache, idx, ys = [None] * 2, None, []
        for a, b in ((0, 1), (1, 5), (5, 7)):
            idx = [0]
            ys.append(rb(x[:, :, a:b], cache, idx))
        out["resblock_stream"] = torch.cat(ys, 2)[0]
    np.savez_compressed(os.path.join(GOLD, "vae_tiny.npz"), **{k: v.numpy() for k, v in out.items()})
    print("vae_tiny.npz:", {k: tuple(v.shape) for k, v in out.items()})
