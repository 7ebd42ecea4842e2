"""Deterministic synthetic weights and inputs (no checkpoints ship with this repo).

Every tensor is drawn from a numpy PCG64 stream seeded by (seed, crc32(name)), so the
same name always yields the same values on any machine and independent of creation
order.  Parameter names follow the checkpoint layout of the reference's
``UnifiedWanWithMixedConditionModel`` (``wan_model.*``, ``vlm_norm.*``, ``vlm_proj.*``,
``visual_context_adapter.*``; reference: omnivideo/modules/unified_model.py:79-90,
omnivideo/modules/model.py:118-123,202-216,273-276,384-401).
"""
import math
import zlib

import numpy as np
import torch


def _rng(seed, name):
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))


def randn(name, shape, seed=0, scale=1.0):
    a = _rng(seed, name).standard_normal(size=tuple(shape), dtype=np.float32)
    return torch.from_numpy(a * np.float32(scale))


def uniform(name, shape, bound, seed=0):
    a = _rng(seed, name).random(size=tuple(shape), dtype=np.float32)
    return torch.from_numpy((a * 2 - 1) * np.float32(bound))


def dit_config(name="1.3B", **over):
    """Architecture hyper-parameters (reference: omnivideo/configs/wan_t2v_1_3B.py:19-40,
    wan_t2v_A14B.py:19-42, shared_config.py:10)."""
    base = dict(patch_size=(1, 2, 2), in_dim=16, out_dim=16, freq_dim=256, text_dim=4096,
                text_len=512, eps=1e-6, vlm_in_dim=2048,
                visual_patch_size=(1, 4, 4), use_visual_context_adapter=True)
    if name == "1.3B":
        base.update(dim=1536, ffn_dim=8960, num_heads=12, num_layers=30)
    elif name == "A14B":
        base.update(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40)
    elif name == "tiny":
        base.update(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=256,
                    vlm_in_dim=256, text_len=48)
    else:
        raise ValueError(name)
    base.update(over)
    return base


def _linear(sd, prefix, out_f, in_f, seed, bias=True):
    bound = 1.0 / math.sqrt(in_f)
    sd[prefix + ".weight"] = uniform(prefix + ".weight", (out_f, in_f), bound, seed)
    if bias:
        sd[prefix + ".bias"] = uniform(prefix + ".bias", (out_f,), bound, seed)


def _norm_weight(sd, name, dim, seed):
    sd[name] = 1.0 + randn(name, (dim,), seed, 0.1)


def make_unified_state_dict(cfg, seed=0):
    """fp32 state dict for the mixed-condition DiT."""
    C, F, T = cfg["dim"], cfg["ffn_dim"], cfg["text_dim"]
    pt, ph, pw = cfg["patch_size"]
    sd = {}
    w = "wan_model."
    fan = cfg["in_dim"] * pt * ph * pw
    sd[w + "patch_embedding.weight"] = uniform(w + "patch_embedding.weight",
                                               (C, cfg["in_dim"], pt, ph, pw),
                                               1 / math.sqrt(fan), seed)
    sd[w + "patch_embedding.bias"] = uniform(w + "patch_embedding.bias", (C,),
                                             1 / math.sqrt(fan), seed)
    _linear(sd, w + "text_embedding.0", C, T, seed)
    _linear(sd, w + "text_embedding.2", C, C, seed)
    _linear(sd, w + "time_embedding.0", C, cfg["freq_dim"], seed)
    _linear(sd, w + "time_embedding.2", C, C, seed)
    _linear(sd, w + "time_projection.1", 6 * C, C, seed)
    for i in range(cfg["num_layers"]):
        b = f"{w}blocks.{i}."
        for att in ("self_attn", "cross_attn"):
            for p in "qkvo":
                _linear(sd, f"{b}{att}.{p}", C, C, seed)
            _norm_weight(sd, f"{b}{att}.norm_q.weight", C, seed)
            _norm_weight(sd, f"{b}{att}.norm_k.weight", C, seed)
        _norm_weight(sd, b + "norm3.weight", C, seed)
        sd[b + "norm3.bias"] = randn(b + "norm3.bias", (C,), seed, 0.1)
        _linear(sd, b + "ffn.0", F, C, seed)
        _linear(sd, b + "ffn.2", C, F, seed)
        sd[b + "modulation"] = randn(b + "modulation", (1, 6, C), seed, C ** -0.5)
    _linear(sd, w + "head.head", cfg["out_dim"] * pt * ph * pw, C, seed)
    sd[w + "head.modulation"] = randn(w + "head.modulation", (1, 2, C), seed, C ** -0.5)
    _norm_weight(sd, "vlm_norm.weight", cfg["vlm_in_dim"], seed)
    _linear(sd, "vlm_proj", T, cfg["vlm_in_dim"], seed)
    if cfg.get("use_visual_context_adapter", True):
        vt, vh, vw = cfg["visual_patch_size"]
        fan = cfg["in_dim"] * vt * vh * vw
        v = "visual_context_adapter."
        sd[v + "patch_embedding.weight"] = uniform(v + "patch_embedding.weight",
                                                   (C, cfg["in_dim"], vt, vh, vw),
                                                   1 / math.sqrt(fan), seed)
        sd[v + "patch_embedding.bias"] = uniform(v + "patch_embedding.bias", (C,),
                                                 1 / math.sqrt(fan), seed)
        _linear(sd, v + "projection", T, C, seed)
    return sd


def make_special_tokens(cfg, seed=0):
    """Four learned delimiter rows (reference: unified_model.py:249-252)."""
    return {k: randn("special." + k, (1, cfg["text_dim"]), seed)
            for k in ("<img_st>", "<img_ed>", "<prp_st>", "<prp_ed>")}


# ------------------------------------------------------------------------------ VAE

def vae_config(name="wan2.1", **over):
    """Wan 2.1 VAE hyper-parameters (reference: omnivideo/modules/vae2_1.py:596-603)."""
    base = dict(dim=96, z_dim=16, dim_mult=(1, 2, 4, 4), num_res_blocks=2,
                temperal_downsample=(False, True, True))
    if name == "tiny":
        base.update(dim=32)
    base.update(over)
    return base


def _conv(sd, name, cout, cin, k, seed):
    fan = cin * k[0] * k[1] * k[2] if len(k) == 3 else cin * k[0] * k[1]
    bound = 1.0 / math.sqrt(fan)
    sd[name + ".weight"] = uniform(name + ".weight", (cout, cin) + tuple(k), bound, seed)
    sd[name + ".bias"] = uniform(name + ".bias", (cout,), bound, seed)


def _vae_res(sd, name, cin, cout, seed):
    sd[name + ".residual.0.gamma"] = 1.0 + randn(name + ".residual.0.gamma", (cin, 1, 1, 1), seed, 0.1)
    _conv(sd, name + ".residual.2", cout, cin, (3, 3, 3), seed)
    sd[name + ".residual.3.gamma"] = 1.0 + randn(name + ".residual.3.gamma", (cout, 1, 1, 1), seed, 0.1)
    _conv(sd, name + ".residual.6", cout, cout, (3, 3, 3), seed)
    if cin != cout:
        _conv(sd, name + ".shortcut", cout, cin, (1, 1, 1), seed)


def _vae_attn(sd, name, c, seed):
    sd[name + ".norm.gamma"] = 1.0 + randn(name + ".norm.gamma", (c, 1, 1), seed, 0.1)
    _conv(sd, name + ".to_qkv", 3 * c, c, (1, 1), seed)
    _conv(sd, name + ".proj", c, c, (1, 1), seed)


def make_vae_state_dict(cfg, seed=0):
    """fp32 state dict with the key names of the reference's WanVAE_ module tree
    (vae2_1.py:264-507): encoder / decoder / conv1 / conv2."""
    dim, z, mult, nres = cfg["dim"], cfg["z_dim"], tuple(cfg["dim_mult"]), cfg["num_res_blocks"]
    tdown = tuple(cfg["temperal_downsample"])
    sd = {}
    dims = [dim * u for u in (1,) + mult]
    _conv(sd, "encoder.conv1", dims[0], 3, (3, 3, 3), seed)
    idx = 0
    for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
        for _ in range(nres):
            _vae_res(sd, f"encoder.downsamples.{idx}", cin, cout, seed)
            cin = cout
            idx += 1
        if i != len(mult) - 1:
            n = f"encoder.downsamples.{idx}"
            _conv(sd, n + ".resample.1", cout, cout, (3, 3), seed)
            if tdown[i]:
                _conv(sd, n + ".time_conv", cout, cout, (3, 1, 1), seed)
            idx += 1
    top = dims[-1]
    _vae_res(sd, "encoder.middle.0", top, top, seed)
    _vae_attn(sd, "encoder.middle.1", top, seed)
    _vae_res(sd, "encoder.middle.2", top, top, seed)
    sd["encoder.head.0.gamma"] = 1.0 + randn("encoder.head.0.gamma", (top, 1, 1, 1), seed, 0.1)
    _conv(sd, "encoder.head.2", 2 * z, top, (3, 3, 3), seed)
    _conv(sd, "conv1", 2 * z, 2 * z, (1, 1, 1), seed)
    _conv(sd, "conv2", z, z, (1, 1, 1), seed)
    ddims = [dim * u for u in (mult[-1],) + mult[::-1]]
    tup = tdown[::-1]
    _conv(sd, "decoder.conv1", ddims[0], z, (3, 3, 3), seed)
    _vae_res(sd, "decoder.middle.0", ddims[0], ddims[0], seed)
    _vae_attn(sd, "decoder.middle.1", ddims[0], seed)
    _vae_res(sd, "decoder.middle.2", ddims[0], ddims[0], seed)
    idx = 0
    for i, (cin, cout) in enumerate(zip(ddims[:-1], ddims[1:])):
        if i in (1, 2, 3):
            cin = cin // 2
        for _ in range(nres + 1):
            _vae_res(sd, f"decoder.upsamples.{idx}", cin, cout, seed)
            cin = cout
            idx += 1
        if i != len(mult) - 1:
            n = f"decoder.upsamples.{idx}"
            _conv(sd, n + ".resample.1", cout // 2, cout, (3, 3), seed)
            if tup[i]:
                _conv(sd, n + ".time_conv", cout * 2, cout, (3, 1, 1), seed)
            idx += 1
    sd["decoder.head.0.gamma"] = 1.0 + randn("decoder.head.0.gamma", (ddims[-1], 1, 1, 1), seed, 0.1)
    _conv(sd, "decoder.head.2", 3, ddims[-1], (3, 3, 3), seed)
    return sd


# ---------------------------------------------------------------------------- umT5 text encoder

def t5_config(name="umt5-xxl", **over):
    """Encoder hyper-parameters (reference: omnivideo/modules/t5.py:455-468)."""
    if name == "umt5-xxl":
        base = dict(vocab_size=256384, dim=4096, dim_attn=4096, dim_ffn=10240, num_heads=64, num_layers=24,
                    num_buckets=32)
    elif name == "tiny":
        base = dict(vocab_size=300, dim=256, dim_attn=256, dim_ffn=512, num_heads=4, num_layers=2, num_buckets=32)
    else:
        raise KeyError(name)
    base.update(over)
    return base


def make_t5_state_dict(cfg, seed=0):
    """Parameter names of the reference's T5Encoder with shared_pos=False (t5.py:266-311)."""
    sd = {}
    d, da, df = cfg["dim"], cfg["dim_attn"], cfg["dim_ffn"]
    sd["token_embedding.weight"] = randn("t5.token_embedding", (cfg["vocab_size"], d), seed)
    for i in range(cfg["num_layers"]):
        p = f"blocks.{i}."
        sd[p + "norm1.weight"] = 1.0 + randn("t5." + p + "norm1", (d,), seed, 0.1)
        for n_ in ("q", "k", "v"):
            sd[p + f"attn.{n_}.weight"] = randn("t5." + p + n_, (da, d), seed, (d * (da // cfg["num_heads"])) ** -0.25
                                                if n_ != "v" else d ** -0.5)
        sd[p + "attn.o.weight"] = randn("t5." + p + "o", (d, da), seed, da ** -0.5)
        sd[p + "norm2.weight"] = 1.0 + randn("t5." + p + "norm2", (d,), seed, 0.1)
        sd[p + "ffn.gate.0.weight"] = randn("t5." + p + "gate", (df, d), seed, d ** -0.5)
        sd[p + "ffn.fc1.weight"] = randn("t5." + p + "fc1", (df, d), seed, d ** -0.5)
        sd[p + "ffn.fc2.weight"] = randn("t5." + p + "fc2", (d, df), seed, df ** -0.5)
        sd[p + "pos_embedding.embedding.weight"] = randn("t5." + p + "pos", (cfg["num_buckets"], cfg["num_heads"]),
                                                         seed, 0.5)
    sd["norm.weight"] = 1.0 + randn("t5.norm", (d,), seed, 0.1)
    return sd


def make_t5_ids(cfg, batch, length, lens, seed=0):
    """Token ids [batch, length] and the prefix mask the tokenizer would return (1 = real token)."""
    ids = torch.from_numpy(_rng(seed, "t5.ids").integers(0, cfg["vocab_size"], size=(batch, length)))
    mask = (torch.arange(length)[None, :] < torch.tensor(lens)[:, None]).long()
    return ids * mask, mask


# name -> (frames, height, width, target height, target width) of the frame-preprocessing cases
FRAME_CASES = {
    "down_wide": (2, 45, 80, 24, 40),
    "down_tall_crop": (2, 70, 50, 24, 40),
    "up": (1, 20, 30, 32, 48),
    "odd_crop": (1, 37, 61, 16, 28),
    "odd_crop2": (1, 39, 63, 16, 28),
    "same": (1, 24, 40, 24, 40),
    "big_reduce": (1, 200, 356, 24, 40),
    "portrait": (1, 64, 36, 40, 24),
}


def make_frames(name, T, Hh, W, seed=0):
    """uint8 frames [T, H, W, 3]: full-range noise with saturated patches (exercises the clip)."""
    f = _rng(seed, "frames." + name).integers(0, 256, size=(T, Hh, W, 3)).astype("uint8")
    f[:, : Hh // 4, : W // 4] = 255
    f[:, -(Hh // 4):, -(W // 4):] = 0
    return f
