"""The CPU oracle (oracle/dit.py, fp32 flow) against golden vectors produced by the
reference's own modules (oracle/gen_golden.py).  Tolerance: fp32 round-off only."""
import os

import numpy as np
import pytest
import torch

from omnivideo.utils import synthetic as syn
from oracle import dit as O

RTOL, ATOL = 2e-4, 2e-5


@pytest.fixture(scope="module")
def G(golden_dir):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, "dit_tiny.npz")).items()}


@pytest.fixture(scope="module")
def W():
    cfg = syn.dit_config("tiny")
    return cfg, syn.make_unified_state_dict(cfg, 0), syn.make_special_tokens(cfg, 0)


def inputs(seed=1):
    cfg = syn.dit_config("tiny")
    return dict(lat=syn.randn("in.lat", (16, 3, 8, 12), seed),
                text=syn.randn("in.text", (10, cfg["text_dim"]), seed),
                text_null=syn.randn("in.text_null", (6, cfg["text_dim"]), seed),
                ar=syn.randn("in.ar", (1, 7, cfg["vlm_in_dim"]), seed),
                visual=syn.randn("in.visual", (1, 16, 3, 8, 12), seed))


def close(a, b, rtol=RTOL, atol=ATOL):
    torch.testing.assert_close(a.float(), b.float(), rtol=rtol, atol=atol)


def test_sinusoid(G):
    close(O.time_sinusoid(731.0, 256)[0].float(), G["sinusoid_731"].float(), 1e-6, 1e-6)


def test_rope(G):
    q = syn.randn("probe.q", (1, 80, 2, 128), 3)[0]
    close(O.rope_rotate(q, O.rope_angles(128, (3, 4, 6))), G["rope_apply"], 1e-6, 1e-6)
    # rows past f*h*w pass through untouched
    assert torch.equal(O.rope_rotate(q, O.rope_angles(128, (3, 4, 6)))[72:], q[72:])


def test_rmsnorm(G, W):
    cfg, sd, _ = W
    x = syn.randn("probe.x", (1, 80, 256), 3)[0]
    close(O.rms_norm(x, sd["wan_model.blocks.1.self_attn.norm_q.weight"], cfg["eps"], "fp32"), G["rmsnorm"])


def test_self_attn(G, W):
    cfg, sd, _ = W
    x = syn.randn("probe.x", (1, 80, 256), 3)[0]
    y = O.self_attention(x, sd, "wan_model.blocks.1.self_attn", 2, cfg["eps"],
                         O.rope_angles(128, (3, 4, 6)), 72, "fp32")
    close(y, G["self_attn"])


def test_cross_attn(G, W):
    cfg, sd, _ = W
    x = syn.randn("probe.x", (1, 80, 256), 3)[0]
    c = syn.randn("probe.ctx", (1, 48, 256), 3)[0]
    kv = O.cross_kv(c, sd, "wan_model.blocks.1.cross_attn", cfg["eps"], "fp32")
    close(O.cross_attention(x, kv, sd, "wan_model.blocks.1.cross_attn", 2, cfg["eps"], "fp32"), G["cross_attn"])


def test_block(G, W):
    cfg, sd, _ = W
    x = syn.randn("probe.x", (1, 80, 256), 3)[0]
    c = syn.randn("probe.ctx", (1, 48, 256), 3)[0]
    e0 = syn.randn("probe.e0", (1, 1, 6, 256), 3, 0.3)[0, 0]
    kv = O.cross_kv(c, sd, "wan_model.blocks.1.cross_attn", cfg["eps"], "fp32")
    y = O.block_forward(x, e0, kv, sd, "wan_model.blocks.1", cfg, O.rope_angles(128, (3, 4, 6)), 72, "fp32")
    close(y, G["block"])


def test_head(G, W):
    cfg, sd, _ = W
    x = syn.randn("probe.x", (1, 80, 256), 3)[0]
    e = syn.randn("probe.e", (1, 1, 256), 3, 0.3)[0, 0]
    m = sd["wan_model.head.modulation"].reshape(2, -1) + e[None]
    y = O.layer_norm(x, cfg["eps"]) * (1 + m[1]) + m[0]
    y = torch.nn.functional.linear(y, sd["wan_model.head.head.weight"], sd["wan_model.head.head.bias"])
    close(y, G["head"])


def test_condition_pieces(G, W):
    cfg, sd, _ = W
    inp = inputs()
    close(O.visual_adapter(inp["visual"][0], sd, cfg, "fp32"), G["visual_adapter"])
    close(O.vlm_project(inp["ar"][0], sd, cfg, "fp32"), G["vlm_proj"])


def test_forward_text_only(G, W):
    cfg, sd, _ = W
    inp = inputs()
    y = O.unified_forward(inp["lat"], 731, sd, cfg, 72, "fp32", text=inp["text"])
    close(y, G["text_only"])


def test_forward_full_special_padded(G, W):
    cfg, sd, sp = W
    inp = inputs()
    y = O.unified_forward(inp["lat"], 999, sd, cfg, 80, "fp32", text=inp["text"], ar_vision=inp["ar"],
                          visual=inp["visual"], special=sp)
    close(y, G["full_special_pad"])


def test_forward_null_branch(G, W):
    cfg, sd, sp = W
    inp = inputs()
    ar_null = torch.zeros(2, cfg["vlm_in_dim"]) + 1e-6
    y = O.unified_forward(inp["lat"], 208, sd, cfg, 72, "fp32", text=inp["text_null"], ar_vision=ar_null,
                          visual=inp["visual"], special=sp)
    close(y, G["null_branch"])


def test_forward_truncated_context(G, W):
    cfg, sd, sp = W
    inp = inputs()
    cfg30 = dict(cfg, text_len=30)
    y = O.unified_forward(inp["lat"], 500, sd, cfg30, 72, "fp32", text=inp["text"], ar_vision=inp["ar"],
                          visual=inp["visual"], special=sp)
    close(y, G["truncated_ctx30"])


def test_bf16_flow_stays_near_fp32(G, W):
    """The bf16 emulation must stay within bf16 tolerance of the fp32 golden."""
    cfg, sd, sp = W
    inp = inputs()
    y = O.unified_forward(inp["lat"], 999, O.round_weights_bf16(sd), cfg, 80, "bf16", text=inp["text"],
                          ar_vision=inp["ar"], visual=inp["visual"], special=sp)
    rel = (y - G["full_special_pad"]).norm() / G["full_special_pad"].norm()
    assert rel < 3e-2, rel


@pytest.mark.parametrize("flow", ["fp32", "bf16"])
def test_sampled_row_forward_equals_whole_forward(flow):
    """wan_forward_rows (one layer, sampled token rows; what the A14B-geometry GPU test compares
    against) reproduces the same rows of wan_forward exactly, padded sequence tail included."""
    import torch
    from omnivideo.utils import synthetic as syn
    from oracle import dit as O
    cfg = syn.dit_config("tiny", num_layers=1)
    sd = syn.make_unified_state_dict(cfg, 0)
    if flow == "bf16":
        sd = O.round_weights_bf16(sd)
    lat = syn.randn("rows.lat", (16, 3, 8, 12), 1)
    ctx = syn.randn("rows.text", (10, cfg["text_dim"]), 1)
    full = O.wan_forward(lat, 640, ctx, 80, sd, cfg, flow)
    rows = [0, 1, 17, 40, 70, 71]
    got = O.wan_forward_rows(lat, 640, ctx, 80, sd, cfg, rows, flow)
    want = O.gather_token_patches(full, rows, (3, 4, 6), cfg["patch_size"])
    # fp32: summation order of the row-subset matmuls only; bf16 flow: that order can flip a rounding
    tol = dict(rtol=1e-5, atol=1e-6) if flow == "fp32" else dict(rtol=2e-3, atol=5e-5)
    torch.testing.assert_close(got, want, **tol)


@pytest.fixture(scope="module")
def GL(golden_dir):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, "loop_tiny.npz")).items()}


def test_layernorm(GL, W):
    """WanLayerNorm without and with affine parameters (reference model.py:87-97; norm3 of a block)."""
    cfg, sd, _ = W
    x = syn.randn("probe.x", (1, 80, cfg["dim"]), 3)[0]
    close(O.layer_norm(x, cfg["eps"]), GL["layernorm_plain"])
    close(O.layer_norm(x, cfg["eps"], sd["wan_model.blocks.1.norm3.weight"], sd["wan_model.blocks.1.norm3.bias"]),
          GL["layernorm_affine"])


def test_guided_unipc_trajectory_through_the_reference_loop(GL, W):
    """Four guided UniPC steps of the tiny mixed-condition model: the oracle's loop (forward of both guidance
    branches, combine, scheduler step) against the trajectory the reference's own modules produce when driven
    as x2x_gen_unified.py:474-492 drives them (oracle/gen_golden.py: gen_loop), after every step."""
    from oracle import unipc as U
    cfg, sd, special = W
    inp = inputs()
    ar_null = torch.zeros(2, cfg["vlm_in_dim"]) + 1e-6
    sched = U.UniPC(4, 5.0)
    assert [int(t) for t in sched.timesteps] == GL["timesteps_4"].tolist()
    lat = syn.randn("gen.noise", (16, 3, 8, 12), 11)
    with torch.no_grad():
        for i, t in enumerate(sched.timesteps):
            c = O.unified_forward(lat, int(t), sd, cfg, 72, "fp32", text=inp["text"], ar_vision=inp["ar"][0],
                                  visual=inp["visual"], special=special)
            u = O.unified_forward(lat, int(t), sd, cfg, 72, "fp32", text=inp["text_null"], ar_vision=ar_null,
                                  visual=inp["visual"], special=special)
            lat = sched.step(u + 3.0 * (c - u), lat)
            close(lat, GL["cfg_unipc_traj_4"][i], 5e-4, 5e-5)
