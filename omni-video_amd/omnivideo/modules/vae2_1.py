"""Wan 2.1 causal 3D VAE with the reference's class names, state-dict keys and call
signatures (reference: omnivideo/modules/vae2_1.py), executed by libomnihip's
channels-last fp32 kernels.

The module tree only owns parameters (so released `Wan2.1_VAE.pth` state dicts load by
name).  Activations live as [T,H,W,C] fp32 on the device.  The streaming semantics of the
reference are kept: the encoder consumes frame chunks 1,4,4,..., the decoder one latent
frame per call, every causal conv sees the last two input frames of the previous chunk.
Instead of copying those frames into a feature cache (vae2_1.py:206-215), each conv keeps
references to the previous chunk's input tensor and hands the kernel two frame pointers;
RMS norm + SiLU, the nearest 2x upsample, zero padding, bias, residual add and the
temporal frame interleave are fused into the convolution kernel.
"""
import logging
import os

import torch
import torch.nn as nn

from .. import _hip as H

__all__ = ["Wan2_1_VAE"]

CACHE_T = 2
F32 = torch.float32


class CausalConv3d(nn.Conv3d):
    """reference vae2_1.py:16-35 (parameter container; see `_Runner.conv`)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._padding = (self.padding[2], self.padding[2], self.padding[1], self.padding[1], 2 * self.padding[0], 0)
        self.padding = (0, 0, 0)


class RMS_norm(nn.Module):
    """reference vae2_1.py:38-53"""

    def __init__(self, dim, channel_first=True, images=True, bias=False):
        super().__init__()
        shape = (dim, 1, 1) if images else (dim, 1, 1, 1)
        self.channel_first, self.scale = channel_first, dim ** 0.5
        self.gamma = nn.Parameter(torch.ones(shape if channel_first else (dim,)))
        self.bias = 0.


class Resample(nn.Module):
    """reference vae2_1.py:65-98"""

    def __init__(self, dim, mode):
        assert mode in ("none", "upsample2d", "upsample3d", "downsample2d", "downsample3d")
        super().__init__()
        self.dim, self.mode = dim, mode
        if mode in ("upsample2d", "upsample3d"):
            self.resample = nn.Sequential(nn.Identity(), nn.Conv2d(dim, dim // 2, 3, padding=1))
            if mode == "upsample3d":
                self.time_conv = CausalConv3d(dim, dim * 2, (3, 1, 1), padding=(1, 0, 0))
        elif mode in ("downsample2d", "downsample3d"):
            self.resample = nn.Sequential(nn.Identity(), nn.Conv2d(dim, dim, 3, stride=(2, 2)))
            if mode == "downsample3d":
                self.time_conv = CausalConv3d(dim, dim, (3, 1, 1), stride=(2, 1, 1), padding=(0, 0, 0))
        else:
            self.resample = nn.Identity()


class ResidualBlock(nn.Module):
    """reference vae2_1.py:185-199"""

    def __init__(self, in_dim, out_dim, dropout=0.0):
        super().__init__()
        self.in_dim, self.out_dim = in_dim, out_dim
        self.residual = nn.Sequential(
            RMS_norm(in_dim, images=False), nn.SiLU(), CausalConv3d(in_dim, out_dim, 3, padding=1),
            RMS_norm(out_dim, images=False), nn.SiLU(), nn.Dropout(dropout),
            CausalConv3d(out_dim, out_dim, 3, padding=1))
        self.shortcut = CausalConv3d(in_dim, out_dim, 1) if in_dim != out_dim else nn.Identity()


class AttentionBlock(nn.Module):
    """reference vae2_1.py:222-237"""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        self.norm = RMS_norm(dim)
        self.to_qkv = nn.Conv2d(dim, dim * 3, 1)
        self.proj = nn.Conv2d(dim, dim, 1)
        nn.init.zeros_(self.proj.weight)


class Encoder3d(nn.Module):
    """reference vae2_1.py:264-315"""

    def __init__(self, dim=128, z_dim=4, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
                 temperal_downsample=[True, True, False], dropout=0.0):
        super().__init__()
        if attn_scales:
            raise NotImplementedError("attention at intermediate scales is not used by Wan 2.1 and not built")
        dims = [dim * u for u in [1] + list(dim_mult)]
        self.conv1 = CausalConv3d(3, dims[0], 3, padding=1)
        downs = []
        for i, (in_dim, out_dim) in enumerate(zip(dims[:-1], dims[1:])):
            for _ in range(num_res_blocks):
                downs.append(ResidualBlock(in_dim, out_dim, dropout))
                in_dim = out_dim
            if i != len(dim_mult) - 1:
                downs.append(Resample(out_dim, "downsample3d" if temperal_downsample[i] else "downsample2d"))
        self.downsamples = nn.Sequential(*downs)
        self.middle = nn.Sequential(ResidualBlock(out_dim, out_dim, dropout), AttentionBlock(out_dim),
                                    ResidualBlock(out_dim, out_dim, dropout))
        self.head = nn.Sequential(RMS_norm(out_dim, images=False), nn.SiLU(), CausalConv3d(out_dim, z_dim, 3, padding=1))


class Decoder3d(nn.Module):
    """reference vae2_1.py:368-420"""

    def __init__(self, dim=128, z_dim=4, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
                 temperal_upsample=[False, True, True], dropout=0.0):
        super().__init__()
        if attn_scales:
            raise NotImplementedError("attention at intermediate scales is not used by Wan 2.1 and not built")
        dims = [dim * u for u in [dim_mult[-1]] + list(dim_mult[::-1])]
        self.conv1 = CausalConv3d(z_dim, dims[0], 3, padding=1)
        self.middle = nn.Sequential(ResidualBlock(dims[0], dims[0], dropout), AttentionBlock(dims[0]),
                                    ResidualBlock(dims[0], dims[0], dropout))
        ups = []
        for i, (in_dim, out_dim) in enumerate(zip(dims[:-1], dims[1:])):
            if i in (1, 2, 3):
                in_dim = in_dim // 2
            for _ in range(num_res_blocks + 1):
                ups.append(ResidualBlock(in_dim, out_dim, dropout))
                in_dim = out_dim
            if i != len(dim_mult) - 1:
                ups.append(Resample(out_dim, "upsample3d" if temperal_upsample[i] else "upsample2d"))
        self.upsamples = nn.Sequential(*ups)
        self.head = nn.Sequential(RMS_norm(out_dim, images=False), nn.SiLU(), CausalConv3d(out_dim, 3, 3, padding=1))


class _Ctx:
    """Left temporal context of one causal conv: the input frames at time -1 and -2 of the
    previous chunk (tensor views [H,W,C]) and their per-voxel inverse norms."""
    __slots__ = ("p1", "p2", "r1", "r2", "rep")

    def __init__(self):
        self.p1 = self.p2 = self.r1 = self.r2 = None
        self.rep = False

    def push(self, x, rn):
        """x [T,H,W,C] just consumed: the new context is its last two frames, completed with the
        old last frame when T == 1 (reference vae2_1.py:206-213)."""
        if x.shape[0] >= 2:
            self.p2, self.p1 = x[-2], x[-1]
            self.r2, self.r1 = (rn[-2], rn[-1]) if rn is not None else (None, None)
        else:
            self.p2, self.p1 = self.p1, x[-1]
            self.r2, self.r1 = self.r1, (rn[-1] if rn is not None else None)


def shard_rows(total, parts):
    """Row counts of `total` rows split over `parts` ranks, as even as possible, larger first."""
    q, r = divmod(total, parts)
    return [q + (i < r) for i in range(parts)]


class _Shard:
    """Row-sharded execution of the decoder over the ranks of a sequence-parallel group (the
    reference decodes on the group's first rank only while the others idle,
    x2x_gen_unified.py:512-524).  Rank r owns a horizontal strip of every activation, held with
    one extra row above and below ([T, rows + 2, W, C]).  Before a convolution with spatial extent
    the two extra rows are refreshed: the neighbour's adjacent row, or zeros at the image border -
    exactly what the unsharded convolution sees there (its zero padding).  Everything without
    spatial extent (1x1x1 and temporal convolutions, norms, residual adds) runs on the strip as
    is; the values in the extra rows between refreshes are never read by a valid output."""

    def __init__(self, comm, rows):
        self.comm, self.rows = comm, list(rows)

    def refresh(self, x):
        up, down = self.comm.exchange_rows(x[:, 1].contiguous(), x[:, -2].contiguous())
        if up is None:
            x[:, 0].zero_()
        else:
            x[:, 0].copy_(up)
        if down is None:
            x[:, -1].zero_()
        else:
            x[:, -1].copy_(down)

    def gather(self, x, scale=1):
        """strip with its extra rows -> the full-height tensor (on every rank)."""
        return self.comm.all_gather_rows(x[:, 1:-1].contiguous(), [n * scale for n in self.rows])

    def strip(self, full, scale=1):
        """full-height tensor -> this rank's strip with (stale) extra rows."""
        r0 = sum(self.rows[:self.comm.rank]) * scale
        n = self.rows[self.comm.rank] * scale
        out = full.new_zeros((full.shape[0], n + 2) + tuple(full.shape[2:]))
        lo, hi = max(r0 - 1, 0), min(r0 + n + 1, full.shape[1])
        out[:, lo - (r0 - 1):lo - (r0 - 1) + hi - lo] = full[:, lo:hi]
        return out


class _Runner:
    """Executes the encoder / decoder chunk by chunk on channels-last tensors."""

    def __init__(self, model):
        self.m = model
        self.ctx = {}
        self._w = {}
        self.shard = None           # a _Shard while a row-sharded decode is running

    def reset(self):
        self.ctx = {}

    def weight(self, conv):
        """[Cout,Cin,(KT,)KH,KW] -> fp32 [taps, Cout, Cin padded to 4]; cached per parameter version."""
        key = id(conv)
        w = conv.weight
        ver = (w.data_ptr(), w._version)
        hit = self._w.get(key)
        if hit is None or hit[0] != ver:
            wd = w.detach().to(F32)
            if wd.dim() == 4:
                wd = wd.unsqueeze(2)
            cout, cin = wd.shape[:2]
            cp = (cin + 3) // 4 * 4
            t = wd.permute(2, 3, 4, 0, 1).reshape(-1, cout, cin)
            if cp != cin:
                t = torch.cat([t, t.new_zeros(t.shape[0], cout, cp - cin)], dim=2)
            t = t.contiguous()
            # the kernel's default path multiplies split-bf16 operands: hand it the weights already split
            # (the exact-fp32 path, OVD_VAE_EXACT_FP32=1, takes the fp32 weights)
            tw = t if os.environ.get("OVD_VAE_EXACT_FP32", "0") not in ("", "0") else H.SplitWeight(t)
            hit = (ver, tw, conv.bias.detach().to(F32).contiguous(), tuple(wd.shape[2:]), t)
            self._w[key] = hit
        return hit[1], hit[2], hit[3]

    def weight_f32(self, conv):
        """the fp32 [taps, Cout, Cin] weights (for callers that slice them)"""
        self.weight(conv)
        return self._w[id(conv)][4]

    def conv(self, x, conv, name=None, norm=None, silu=False, resid=None, **kw):
        """Causal conv with optional fused RMS-norm(+SiLU) on the input.  `name` keys the
        temporal context; None = stateless (1x1x1 convs, per-frame 2D convs)."""
        w, b, k = self.weight(conv)
        if self.shard is not None and k[1] > 1:
            self.shard.refresh(x)
        c = self.ctx.setdefault(name, _Ctx()) if name is not None else None
        gamma = norm.gamma.detach().reshape(-1).to(F32).contiguous() if norm is not None else None
        if norm is not None and isinstance(w, H.SplitWeight) and k[1:] == (3, 3) and x.shape[-1] % 32 == 0 \
                and kw.get("pad_hw", 0) == 1 and not (set(kw) - {"pad_t", "pad_hw"}):
            # Normalised + activated convolutions (the bulk of the FLOPs): norm, SiLU and the split into bf16
            # hi | lo happen once per layer here, and the kernel's input staging is then pure copies.  The
            # temporal context is kept in that form too.
            xs = H.vae_act_split(x, gamma, silu)
            y = H.vae_conv3d_ps(xs, w, b, k, prev2=None if c is None else c.p2, prev1=None if c is None else c.p1,
                                resid=resid, pad_t=kw.get("pad_t", 0), pad_hw=1)
            if c is not None:
                c.push(xs, None)
            return y
        rn = H.vae_voxel_rnorm(x) if norm is not None else None
        p1 = p2 = r1 = r2 = None
        if c is not None:
            p1, p2, r1, r2 = c.p1, c.p2, c.r1, c.r2
        y = H.vae_conv3d(x, w, b, k, prev2=p2, prev1=p1, rnorm=rn, rnorm_prev2=r2, rnorm_prev1=r1, gamma=gamma,
                         silu=silu, resid=resid, **kw)
        if c is not None:
            c.push(x, rn)
        return y

    # ------------------------------------------------------------------ blocks
    def res_block(self, x, blk, name):
        """reference ResidualBlock.forward (vae2_1.py:201-219)"""
        r = blk.residual
        h = x if isinstance(blk.shortcut, nn.Identity) else self.conv(x, blk.shortcut)
        y = self.conv(x, r[2], name + ".residual.2", norm=r[0], silu=True, pad_t=2, pad_hw=1)
        return self.conv(y, r[6], name + ".residual.6", norm=r[3], silu=True, resid=h, pad_t=2, pad_hw=1)

    def attn_block(self, x, blk):
        """reference AttentionBlock.forward (vae2_1.py:239-261): per frame, one head of
        width C over the h*w positions; q/k/v are 1x1 convs of the RMS-normed input."""
        if self.shard is not None:      # attention spans the whole frame: gather, compute, keep the own strip
            sh, self.shard = self.shard, None
            try:
                return sh.strip(self.attn_block(sh.gather(x), blk))
            finally:
                self.shard = sh
        T, Hh, Ww, C = x.shape
        L = Hh * Ww
        if L % 4:
            raise NotImplementedError("VAE attention needs h*w to be a multiple of 4")
        wqkv, bq, _ = self.weight(blk.to_qkv)
        rn = H.vae_voxel_rnorm(x)
        gamma = blk.norm.gamma.detach().reshape(-1).to(F32).contiguous()
        # q | k | v in one launch (every output channel is its own dot product: the same values as three launches)
        qkv = H.vae_conv3d(x, wqkv, bq, (1, 1, 1), rnorm=rn, gamma=gamma)
        q, k, v = (qkv[..., i * C:(i + 1) * C].contiguous() for i in range(3))
        wp, bp, _ = self.weight(blk.proj)
        out = torch.empty_like(x)
        for t in range(T):
            s = H.vae_conv3d(q[t].reshape(1, 1, L, C), k[t].reshape(1, L, C), None, (1, 1, 1)).reshape(L, L)
            H.softmax_rows(s, C ** -0.5)
            vt = v[t].reshape(L, C).t().contiguous().reshape(1, C, L)
            o = H.vae_conv3d(s.reshape(1, 1, L, L), vt, None, (1, 1, 1))
            H.vae_conv3d(o.reshape(1, Hh, Ww, C), wp, bp, (1, 1, 1), resid=x[t:t + 1], out=out[t:t + 1])
        return out

    def upsample(self, x, rs, name):
        """reference Resample.forward, upsample modes (vae2_1.py:100-140)"""
        if rs.mode == "upsample3d":
            key = name + ".time_conv"
            c = self.ctx.get(key)
            if c is None:
                c = self.ctx[key] = _Ctx()
                c.rep = True                        # first chunk: no temporal upsampling
            else:
                w, b, k = self.weight(rs.time_conv)
                if c.rep:                           # second chunk: zero left context
                    y = H.vae_conv3d(x, w, b, k, pad_t=2, time_interleave=True)
                    c.rep = False
                else:
                    y = H.vae_conv3d(x, w, b, k, prev2=c.p2, prev1=c.p1, pad_t=2, time_interleave=True)
                c.push(x, None)
                x = y
        w, b, k = self.weight(rs.resample[1])
        if self.shard is not None:
            self.shard.refresh(x)
        if isinstance(w, H.SplitWeight) and x.shape[-1] % 32 == 0:
            # the source image split into bf16 hi | lo once (a quarter of the output's voxels), staged by pure copies
            y = H.vae_conv3d_ps(H.vae_act_split(x), w, b, k, pad_hw=1, upsample2=True)
        else:
            y = H.vae_conv3d(x, w, b, k, pad_hw=1, upsample2=True)
        # sharded: [rows + 2] -> 2 rows + 4: the outermost row on each side is dropped, leaving one extra row
        return y[:, 1:-1].contiguous() if self.shard is not None else y

    def downsample(self, x, rs, name):
        """reference Resample.forward, downsample modes (vae2_1.py:137-159)"""
        w, b, k = self.weight(rs.resample[1])
        y = H.vae_conv3d(x, w, b, k, stride_hw=2, pad_hw=0)
        if rs.mode == "downsample3d":
            key = name + ".time_conv"
            c = self.ctx.get(key)
            if c is None:
                c = self.ctx[key] = _Ctx()
                c.p1 = y[-1]                        # first chunk passes through
            else:
                wt, bt, kt = self.weight(rs.time_conv)
                nxt = y[-1]
                y2 = H.vae_conv3d(y, wt, bt, kt, prev1=c.p1, stride_t=2, pad_t=1)
                c.p1 = nxt                          # a view: keeps its chunk's storage alive
                y = y2
        return y

    def encoder_chunk(self, x):
        e = self.m.encoder
        x = self.conv(x, e.conv1, "encoder.conv1", pad_t=2, pad_hw=1)
        for i, layer in enumerate(e.downsamples):
            n = f"encoder.downsamples.{i}"
            x = self.res_block(x, layer, n) if isinstance(layer, ResidualBlock) else self.downsample(x, layer, n)
        x = self.res_block(x, e.middle[0], "encoder.middle.0")
        x = self.attn_block(x, e.middle[1])
        x = self.res_block(x, e.middle[2], "encoder.middle.2")
        return self.conv(x, e.head[2], "encoder.head.2", norm=e.head[0], silu=True, pad_t=2, pad_hw=1)

    def decoder_chunk(self, x):
        d = self.m.decoder
        x = self.conv(x, d.conv1, "decoder.conv1", pad_t=2, pad_hw=1)
        x = self.res_block(x, d.middle[0], "decoder.middle.0")
        x = self.attn_block(x, d.middle[1])
        x = self.res_block(x, d.middle[2], "decoder.middle.2")
        for i, layer in enumerate(d.upsamples):
            n = f"decoder.upsamples.{i}"
            x = self.res_block(x, layer, n) if isinstance(layer, ResidualBlock) else self.upsample(x, layer, n)
        return self.conv(x, d.head[2], "decoder.head.2", norm=d.head[0], silu=True, pad_t=2, pad_hw=1)


class WanVAE_(nn.Module):
    """reference vae2_1.py:482-588"""

    def __init__(self, dim=128, z_dim=4, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
                 temperal_downsample=[True, True, False], dropout=0.0):
        super().__init__()
        self.dim, self.z_dim, self.dim_mult, self.num_res_blocks = dim, z_dim, dim_mult, num_res_blocks
        self.attn_scales, self.temperal_downsample = attn_scales, temperal_downsample
        self.temperal_upsample = temperal_downsample[::-1]
        self.encoder = Encoder3d(dim, z_dim * 2, dim_mult, num_res_blocks, attn_scales, self.temperal_downsample, dropout)
        self.conv1 = CausalConv3d(z_dim * 2, z_dim * 2, 1)
        self.conv2 = CausalConv3d(z_dim, z_dim, 1)
        self.decoder = Decoder3d(dim, z_dim, dim_mult, num_res_blocks, attn_scales, self.temperal_upsample, dropout)
        self._runner = None

    def runner(self):
        if self._runner is None:
            self._runner = _Runner(self)
        return self._runner

    def clear_cache(self):
        if self._runner is not None:
            self._runner.reset()

    @staticmethod
    def _scale_vectors(scale, z, device):
        mean, inv_std = scale
        if not isinstance(mean, torch.Tensor):
            mean = torch.full((z,), float(mean))
            inv_std = torch.full((z,), float(inv_std))
        return mean.to(device, F32).contiguous(), inv_std.to(device, F32).contiguous()

    def encode(self, x, scale):
        """x [1,3,T,H,W] -> normalised mu [1,z,T',h,w] (reference vae2_1.py:515-541)."""
        dev = self.conv1.weight.device
        if x.shape[0] != 1:
            raise NotImplementedError("batch of one, as the pipeline calls it")
        run = self.runner()
        run.reset()
        vid = x[0].to(dev, F32).contiguous()
        t = vid.shape[1]
        outs = []
        for i in range(1 + (t - 1) // 4):
            sl = vid[:, :1] if i == 0 else vid[:, 1 + 4 * (i - 1):1 + 4 * i]
            outs.append(run.encoder_chunk(H.vae_to_channels_last(sl.contiguous(), c_pad=4)))
        out = torch.cat(outs, dim=0)
        mu = run.conv(out, self.conv1)
        mean, inv_std = self._scale_vectors(scale, self.z_dim, dev)
        z = H.vae_from_channels_last(mu, channels=self.z_dim, pre=-mean, mul=inv_std)
        run.reset()
        return z[None]

    def decode(self, z, scale, as_frames=False, comm=None):
        """z [1,z,T,h,w] -> video [1,3,4T-3,8h,8w] (reference vae2_1.py:543-567).  With
        `as_frames` the decoder's channels-last output is converted straight to the uint8
        [4T-3,8h,8w,3] frames that the reference's cache_video hands to its video writer
        (utils/utils.py:175-182), without materialising the planar fp32 video.
        comm: a communicator of `size` ranks (rank, size, exchange_rows, all_gather_rows; e.g.
        engine.TorchGroupComm over the sequence-parallel group): every rank passes the same z,
        decodes a horizontal strip of the video (see _Shard) and receives the full result."""
        dev = self.conv1.weight.device
        if z.shape[0] != 1:
            raise NotImplementedError("batch of one, as the pipeline calls it")
        run = self.runner()
        run.reset()
        mean, inv_std = self._scale_vectors(scale, self.z_dim, dev)
        zc = H.vae_to_channels_last(z[0].to(dev, F32).contiguous(), mul=(1.0 / inv_std).contiguous(), post=mean)
        x = run.conv(zc, self.conv2)
        if comm is not None and comm.size > 1:
            if x.shape[1] < comm.size:
                raise ValueError(f"cannot shard {x.shape[1]} latent rows over {comm.size} ranks")
            run.shard = _Shard(comm, shard_rows(x.shape[1], comm.size))
            x = run.shard.strip(x)
        try:
            outs = [run.decoder_chunk(x[i:i + 1]) for i in range(x.shape[0])]
            out = torch.cat(outs, dim=0)
            if run.shard is not None:
                out = run.shard.gather(out, scale=2 ** (len(self.dim_mult) - 1))
        finally:
            run.shard = None
        run.reset()
        if as_frames:
            return H.video_to_uint8(out, channels_last=True)
        return H.vae_from_channels_last(out)[None]


def _video_vae(pretrained_path=None, z_dim=None, device="cpu", **kwargs):
    """reference vae2_1.py:591-615"""
    cfg = dict(dim=96, z_dim=z_dim, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
               temperal_downsample=[False, True, True], dropout=0.0)
    cfg.update(**kwargs)
    model = WanVAE_(**cfg)
    if pretrained_path is not None:
        logging.info(f"loading {pretrained_path}")
        model.load_state_dict(torch.load(pretrained_path, map_location="cpu", weights_only=True))
    return model


class Wan2_1_VAE:
    """reference vae2_1.py:618-662"""

    def __init__(self, z_dim=16, vae_pth="cache/vae_step_411000.pth", dtype=torch.float, device="cuda", **cfg):
        if dtype != torch.float:
            raise NotImplementedError("the VAE kernels compute in fp32 (the reference's setting)")
        self.dtype, self.device = dtype, device
        mean = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
                0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921]
        std = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
               3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160]
        self.mean = torch.tensor(mean, dtype=dtype, device=device)
        self.std = torch.tensor(std, dtype=dtype, device=device)
        self.scale = [self.mean, 1.0 / self.std]
        self.model = _video_vae(pretrained_path=vae_pth, z_dim=z_dim, **cfg).eval().requires_grad_(False).to(device)

    def encode(self, videos):
        """videos: list of [3,T,H,W] -> list of fp32 [z,T',h,w]."""
        return [self.model.encode(u.unsqueeze(0), self.scale).float().squeeze(0) for u in videos]

    def decode(self, zs, comm=None):
        """zs: list of [z,T,h,w] -> list of fp32 [3,4T-3,8h,8w] clamped to [-1,1].  comm (extension):
        decode row-sharded over the ranks of a communicator, see WanVAE_.decode."""
        return [self.model.decode(u.unsqueeze(0), self.scale, comm=comm).float().clamp_(-1, 1).squeeze(0) for u in zs]

    def decode_frames(self, zs, comm=None):
        """zs: list of [z,T,h,w] -> list of uint8 [4T-3,8h,8w,3] frames: decode() followed by the
        tensor half of cache_video (reference utils/utils.py:175-182) in one pass."""
        return [self.model.decode(u.unsqueeze(0), self.scale, as_frames=True, comm=comm) for u in zs]
