"""The VAE oracle (oracle/vae.py) against golden vectors made by the reference's WanVAE_
(tests/golden/vae_tiny.npz via oracle/gen_golden_vae.py): streaming encode (chunks
1,4,4), per-frame decode with the temporal-upsampling first-chunk rule, and a residual
block streamed over three ragged chunks."""
import os

import numpy as np
import pytest
import torch

from omnivideo.utils import synthetic as syn
from oracle import vae as V


@pytest.fixture(scope="module")
def G(golden_dir):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, "vae_tiny.npz")).items()}


@pytest.fixture(scope="module")
def W():
    cfg = syn.vae_config("tiny")
    return cfg, syn.make_vae_state_dict(cfg, 0), V.vae_layout(cfg["dim"], cfg["z_dim"], cfg["dim_mult"],
                                                              cfg["num_res_blocks"], cfg["temperal_downsample"])


@pytest.mark.parametrize("frames,hw", [(5, (32, 48)), (9, (32, 32)), (1, (16, 16))])
def test_encode(G, W, frames, hw):
    cfg, sd, lay = W
    vid = syn.randn(f"vae.video{frames}", (3, frames, *hw), 5).clamp(-1, 1)
    with torch.no_grad():
        z = V.encode(vid, sd, lay)
    torch.testing.assert_close(z, G[f"encode_{frames}"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("tl,hw", [(2, (4, 6)), (3, (4, 4)), (1, (2, 2))])
def test_decode(G, W, tl, hw):
    cfg, sd, lay = W
    z = syn.randn(f"vae.z{tl}", (16, tl, *hw), 5)
    with torch.no_grad():
        x = V.decode(z, sd, lay)
    assert tuple(x.shape) == (3, 4 * tl - 3, 8 * hw[0], 8 * hw[1])
    torch.testing.assert_close(x, G[f"decode_{tl}"], rtol=1e-4, atol=1e-5)


def test_resblock_streaming(G, W):
    cfg, sd, lay = W
    x = syn.randn("vae.probe", (1, 32, 7, 6, 8), 5)[0]
    cache, ys = {}, []
    with torch.no_grad():
        for a, b in ((0, 1), (1, 5), (5, 7)):
            ys.append(V.res_block(x[:, a:b], sd, "encoder.downsamples.0", cache))
    torch.testing.assert_close(torch.cat(ys, 1), G["resblock_stream"], rtol=1e-4, atol=1e-5)


@pytest.fixture(scope="module")
def GR(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "vae_resample.npz")).items())


@pytest.mark.parametrize("mode", ["downsample2d", "downsample3d", "upsample2d", "upsample3d"])
def test_resample_modes_streamed_over_three_chunks(GR, W, mode):
    """Each Resample mode on its own, three consecutive chunks through the oracle's feature cache, against the
    reference module's output (oracle/gen_golden_vae.py: resample_probes): the first-chunk pass-through of the
    temporal down-sampling, the 'Rep' sentinel / zero-context / cached-context sequence of the temporal
    up-sampling (reference vae2_1.py:102-158)."""
    cfg, sd, lay = W
    name = str(GR[mode + "_module"])
    want = torch.from_numpy(GR[mode])
    chunks = ((0, 1), (1, 5), (5, 7)) if mode.startswith("down") else ((0, 1), (1, 2), (2, 3))
    cin = sd[name + ".resample.1.weight"].shape[1]
    x = syn.randn(f"vae.resample.{mode}", (1, cin, chunks[-1][1], 6, 8), 5)[0]
    fn = V.downsample if mode.startswith("down") else V.upsample
    cache, ys = {}, []
    with torch.no_grad():
        for a, b in chunks:
            ys.append(fn(x[:, a:b], sd, name, mode, cache))
    got = torch.cat(ys, 1)
    assert tuple(got.shape) == tuple(want.shape)
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-5)
