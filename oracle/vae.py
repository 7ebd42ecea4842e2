"""TEST INFRASTRUCTURE ONLY - CPU oracle for the Wan 2.1 causal 3D VAE (float32).

A from-scratch restatement of omnivideo/modules/vae2_1.py as functions over a flat
name->tensor weight dict (the reference's state_dict names).  Tensors are [C,T,H,W]
(batch of one).  The streaming semantics are kept exactly: the encoder consumes frame
chunks 1,4,4,...; the decoder consumes one latent frame per call; every causal conv
keeps the last two input frames of the previous chunk as left context
(vae2_1.py:27-35,201-219); the temporal up/down-samplers have their first-chunk special
cases (vae2_1.py:100-159).  Caches are keyed by layer name instead of a running index.
Product code never imports this file.
"""
import torch
import torch.nn.functional as F

CACHE_T = 2


def vae_layout(dim=96, z_dim=16, dim_mult=(1, 2, 4, 4), num_res_blocks=2,
               temperal_downsample=(False, True, True)):
    """Module tree of WanVAE_ as (kind, name, args) lists for encoder and decoder
    (vae2_1.py:264-471)."""
    enc, dec = [], []
    dims = [dim * u for u in (1,) + tuple(dim_mult)]
    idx = 0
    for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
        for _ in range(num_res_blocks):
            enc.append(("res", f"encoder.downsamples.{idx}", (cin, cout)))
            cin = cout
            idx += 1
        if i != len(dim_mult) - 1:
            mode = "downsample3d" if temperal_downsample[i] else "downsample2d"
            enc.append((mode, f"encoder.downsamples.{idx}", (cout,)))
            idx += 1
    top = dims[-1]
    ddims = [dim * u for u in (dim_mult[-1],) + tuple(dim_mult[::-1])]
    t_up = tuple(temperal_downsample[::-1])
    idx = 0
    for i, (cin, cout) in enumerate(zip(ddims[:-1], ddims[1:])):
        if i in (1, 2, 3):
            cin = cin // 2
        for _ in range(num_res_blocks + 1):
            dec.append(("res", f"decoder.upsamples.{idx}", (cin, cout)))
            cin = cout
            idx += 1
        if i != len(dim_mult) - 1:
            mode = "upsample3d" if t_up[i] else "upsample2d"
            dec.append((mode, f"decoder.upsamples.{idx}", (cout,)))
            idx += 1
    return dict(enc=enc, dec=dec, top=top, dec_top=ddims[0], dec_out=ddims[-1], z_dim=z_dim, dim=dim)


def causal_conv(x, w, b, cache=None, stride=1):
    """CausalConv3d.forward (vae2_1.py:27-35): left-pad time by 2*pad_t (cache frames first,
    zeros for the rest), symmetric spatial padding; x [C,T,H,W]."""
    kt, kh, kw = w.shape[2:]
    pt = kt - 1 if kt == 3 and stride == 1 else 0
    if cache is not None and pt > 0:
        x = torch.cat([cache, x], dim=1)
        pt -= cache.shape[1]
    x = F.pad(x, (kw // 2, kw // 2, kh // 2, kh // 2, pt, 0))
    return F.conv3d(x[None], w, b, stride=stride)[0]


def rms_norm(x, gamma):
    """RMS_norm (vae2_1.py:50-53): L2-normalise over channels, times sqrt(C) * gamma."""
    c = x.shape[0]
    return F.normalize(x, dim=0) * (c ** 0.5) * gamma.reshape(c, 1, 1, 1)


def _next_cache(x, old):
    """Last two frames of x, completed with the previous cache's last frame when x has
    only one (vae2_1.py:206-213)."""
    c = x[:, -CACHE_T:].clone()
    if c.shape[1] < 2 and old is not None:
        c = torch.cat([old[:, -1:], c], dim=1)
    return c


def cached_conv(x, sd, name, cache):
    old = cache.get(name)
    y = causal_conv(x, sd[name + ".weight"], sd[name + ".bias"], old)
    cache[name] = _next_cache(x, old)
    return y


def res_block(x, sd, name, cache):
    """ResidualBlock.forward (vae2_1.py:201-219)."""
    h = x
    if name + ".shortcut.weight" in sd:
        h = causal_conv(x, sd[name + ".shortcut.weight"], sd[name + ".shortcut.bias"])
    y = F.silu(rms_norm(x, sd[name + ".residual.0.gamma"]))
    y = cached_conv(y, sd, name + ".residual.2", cache)
    y = F.silu(rms_norm(y, sd[name + ".residual.3.gamma"]))
    y = cached_conv(y, sd, name + ".residual.6", cache)
    return y + h


def attn_block(x, sd, name):
    """AttentionBlock.forward (vae2_1.py:239-261): per-frame single-head attention over
    h*w positions with head_dim = C."""
    c, t, h, w = x.shape
    y = rms_norm(x, sd[name + ".norm.gamma"])
    y = y.permute(1, 0, 2, 3)                                     # [t,c,h,w]
    qkv = F.conv2d(y, sd[name + ".to_qkv.weight"], sd[name + ".to_qkv.bias"])
    q, k, v = qkv.reshape(t, 3 * c, h * w).permute(0, 2, 1).chunk(3, dim=-1)
    o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
    o = o.permute(0, 2, 1).reshape(t, c, h, w)
    o = F.conv2d(o, sd[name + ".proj.weight"], sd[name + ".proj.bias"])
    return o.permute(1, 0, 2, 3) + x


def conv2d_frames(x, w, b, stride=1, pad=1):
    return F.conv2d(x.permute(1, 0, 2, 3), w, b, stride=stride, padding=pad).permute(1, 0, 2, 3)


def upsample(x, sd, name, mode, cache):
    """Resample.forward, upsample modes (vae2_1.py:100-140)."""
    c = x.shape[0]
    if mode == "upsample3d":
        key = name + ".time_conv"
        if key not in cache:
            cache[key] = "Rep"                      # first chunk: no temporal upsampling
        else:
            old = cache[key]
            nxt = x[:, -CACHE_T:].clone()
            if nxt.shape[1] < 2:
                pad = torch.zeros_like(nxt) if isinstance(old, str) else old[:, -1:]
                nxt = torch.cat([pad, nxt], dim=1)
            y = causal_conv(x, sd[key + ".weight"], sd[key + ".bias"], None if isinstance(old, str) else old)
            cache[key] = nxt
            t = y.shape[1]
            y = y.reshape(2, c, t, *y.shape[2:])
            x = torch.stack((y[0], y[1]), dim=2).reshape(c, 2 * t, *y.shape[3:])
    x = F.interpolate(x.permute(1, 0, 2, 3), scale_factor=(2.0, 2.0), mode="nearest-exact")
    x = F.conv2d(x, sd[name + ".resample.1.weight"], sd[name + ".resample.1.bias"], padding=1)
    return x.permute(1, 0, 2, 3)


def downsample(x, sd, name, mode, cache):
    """Resample.forward, downsample modes (vae2_1.py:137-159)."""
    y = F.pad(x.permute(1, 0, 2, 3), (0, 1, 0, 1))
    y = F.conv2d(y, sd[name + ".resample.1.weight"], sd[name + ".resample.1.bias"], stride=2).permute(1, 0, 2, 3)
    if mode == "downsample3d":
        key = name + ".time_conv"
        if key not in cache:
            cache[key] = y.clone()                   # first chunk passes through
        else:
            old = cache[key]
            nxt = y[:, -1:].clone()
            y = F.conv3d(torch.cat([old[:, -1:], y], dim=1)[None], sd[key + ".weight"], sd[key + ".bias"],
                         stride=(2, 1, 1))[0]
            cache[key] = nxt
    return y


def encoder_chunk(x, sd, lay, cache):
    """Encoder3d.forward on one chunk (vae2_1.py:317-365)."""
    x = cached_conv(x, sd, "encoder.conv1", cache)
    for kind, name, _ in lay["enc"]:
        x = res_block(x, sd, name, cache) if kind == "res" else downsample(x, sd, name, kind, cache)
    x = res_block(x, sd, "encoder.middle.0", cache)
    x = attn_block(x, sd, "encoder.middle.1")
    x = res_block(x, sd, "encoder.middle.2", cache)
    x = F.silu(rms_norm(x, sd["encoder.head.0.gamma"]))
    return cached_conv(x, sd, "encoder.head.2", cache)


def decoder_chunk(x, sd, lay, cache):
    """Decoder3d.forward on one latent frame (vae2_1.py:422-471)."""
    x = cached_conv(x, sd, "decoder.conv1", cache)
    x = res_block(x, sd, "decoder.middle.0", cache)
    x = attn_block(x, sd, "decoder.middle.1")
    x = res_block(x, sd, "decoder.middle.2", cache)
    for kind, name, _ in lay["dec"]:
        x = res_block(x, sd, name, cache) if kind == "res" else upsample(x, sd, name, kind, cache)
    x = F.silu(rms_norm(x, sd["decoder.head.0.gamma"]))
    return cached_conv(x, sd, "decoder.head.2", cache)


LATENT_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
               0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921]
LATENT_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
              3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160]


def encode(video, sd, lay):
    """Wan2_1_VAE.encode for one video [3,T,H,W] -> normalised mu [z,T',h,w]
    (vae2_1.py:515-541,646-654)."""
    cache = {}
    t = video.shape[1]
    outs = []
    for i in range(1 + (t - 1) // 4):
        chunk = video[:, :1] if i == 0 else video[:, 1 + 4 * (i - 1):1 + 4 * i]
        outs.append(encoder_chunk(chunk, sd, lay, cache))
    out = torch.cat(outs, dim=1)
    mu = causal_conv(out, sd["conv1.weight"], sd["conv1.bias"])[:lay["z_dim"]]
    mean = torch.tensor(LATENT_MEAN[:lay["z_dim"]]).reshape(-1, 1, 1, 1)
    inv_std = (1.0 / torch.tensor(LATENT_STD[:lay["z_dim"]])).reshape(-1, 1, 1, 1)
    return (mu - mean) * inv_std


def decode(z, sd, lay):
    """Wan2_1_VAE.decode for one latent [z,T,h,w] -> video [3,4T-3,8h,8w] clamped to [-1,1]
    (vae2_1.py:543-567,656-662)."""
    mean = torch.tensor(LATENT_MEAN[:lay["z_dim"]]).reshape(-1, 1, 1, 1)
    inv_std = (1.0 / torch.tensor(LATENT_STD[:lay["z_dim"]])).reshape(-1, 1, 1, 1)
    z = z / inv_std + mean
    x = causal_conv(z, sd["conv2.weight"], sd["conv2.bias"])
    cache = {}
    outs = [decoder_chunk(x[:, i:i + 1], sd, lay, cache) for i in range(x.shape[1])]
    return torch.cat(outs, dim=1).float().clamp_(-1, 1)
