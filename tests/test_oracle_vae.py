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
