"""VAE parity on the GPU: the fused channels-last conv kernel against torch's fp32 conv on
the CPU for every mode the VAE uses, then whole encode / decode of the reduced-width VAE
through the reference-named `Wan2_1_VAE` surface against the oracle and against the
goldens made by the reference's own WanVAE_.

Two arithmetic paths, both run here: the default computes every fp32 product as three bf16 MFMA
products of the operands' hi / lo halves (relative error ~2^-16 per product, finer than the TF32
that cuDNN convolutions default to, coarser than exact fp32) and is held to rtol 2e-4 per layer and
1e-3 per whole encode / decode; OVD_VAE_EXACT_FP32=1 selects exact-fp32 MFMA (only the summation order
differs from the reference) and is held to rtol 2e-5 / 1e-4.  The switch is read once per process, so
the exact path runs in a fresh child process (test_exact_fp32_path_in_fresh_process)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from omnivideo.utils import synthetic as syn
from oracle import vae as V

pytestmark = pytest.mark.gpu
EXACT = os.environ.get("OVD_VAE_EXACT_FP32", "0") not in ("", "0")
RTOL, ATOL = (2e-5, 2e-6) if EXACT else (2e-4, 2e-5)          # one layer
RTOL_NET, ATOL_NET = (1e-4, 1e-5) if EXACT else (1e-3, 1e-4)    # whole encoder / decoder


@pytest.fixture(scope="module")
def H():
    import omnivideo._hip as H
    H.lib()
    return H


def cl(x):           # [C,T,H,W] -> [T,H,W,C] on the device
    return x.permute(1, 2, 3, 0).contiguous().cuda()


def cf(y):           # back
    return y.cpu().permute(3, 0, 1, 2).contiguous()


def prep_w(w):
    if w.dim() == 4:
        w = w.unsqueeze(2)
    co, ci = w.shape[:2]
    return w.permute(2, 3, 4, 0, 1).reshape(-1, co, ci).contiguous().cuda(), tuple(w.shape[2:])


@pytest.mark.parametrize("cin,cout,T,Hh,Ww", [(32, 32, 3, 9, 17), (64, 96, 1, 8, 16), (16, 128, 2, 5, 7), (4, 32, 4, 12, 20),
                                              (96, 3, 2, 10, 33), (128, 160, 1, 4, 6)])
def test_causal_conv333_with_context(H, cin, cout, T, Hh, Ww):
    x = syn.randn("cv.x", (cin, T, Hh, Ww), 1)
    ctx = syn.randn("cv.ctx", (cin, 2, Hh, Ww), 1)
    w = syn.uniform("cv.w", (cout, cin, 3, 3, 3), (27 * cin) ** -0.5, 1)
    b = syn.randn("cv.b", (cout,), 1, 0.1)
    wk, k = prep_w(w)
    for ncache in (0, 1, 2):
        cache = ctx[:, 2 - ncache:] if ncache else None
        ref = V.causal_conv(x, w, b, cache)
        c = cl(ctx)
        y = H.vae_conv3d(cl(x), wk, b.cuda(), k, prev2=c[0] if ncache == 2 else None,
                         prev1=c[1] if ncache >= 1 else None, pad_t=2, pad_hw=1)
        torch.testing.assert_close(cf(y), ref, rtol=RTOL, atol=ATOL)


def test_conv_fused_norm_silu_residual(H):
    cin, cout, T, Hh, Ww = 64, 64, 2, 9, 11
    x = syn.randn("cn.x", (cin, T, Hh, Ww), 2)
    ctx = syn.randn("cn.ctx", (cin, 2, Hh, Ww), 2)
    g = 1 + syn.randn("cn.g", (cin,), 2, 0.1)
    w = syn.uniform("cn.w", (cout, cin, 3, 3, 3), (27 * cin) ** -0.5, 2)
    b = syn.randn("cn.b", (cout,), 2, 0.1)
    res = syn.randn("cn.r", (cout, T, Hh, Ww), 2)
    act = lambda t: F.silu(V.rms_norm(t, g))
    ref = V.causal_conv(act(x), w, b, act(ctx)) + res
    wk, k = prep_w(w)
    xc, cc = cl(x), cl(ctx)
    rn, rc = H.vae_voxel_rnorm(xc), H.vae_voxel_rnorm(cc)
    torch.testing.assert_close(rn.cpu(), 1.0 / x.norm(dim=0).clamp_min(1e-12), rtol=1e-5, atol=1e-7)
    y = H.vae_conv3d(xc, wk, b.cuda(), k, prev2=cc[0], prev1=cc[1], rnorm=rn, rnorm_prev2=rc[0], rnorm_prev1=rc[1],
                     gamma=g.cuda(), silu=True, resid=cl(res), pad_t=2, pad_hw=1)
    torch.testing.assert_close(cf(y), ref, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("Hh,Ww", [(6, 10), (8, 16), (5, 9)])
def test_upsample_conv_and_downsample_conv(H, Hh, Ww):
    c = 64
    x = syn.randn("ud.x", (c, 2, Hh, Ww), 3)
    w = syn.uniform("ud.w", (c // 2, c, 3, 3), (9 * c) ** -0.5, 3)
    b = syn.randn("ud.b", (c // 2,), 3, 0.1)
    up = F.interpolate(x.permute(1, 0, 2, 3), scale_factor=(2.0, 2.0), mode="nearest-exact")
    ref = F.conv2d(up, w, b, padding=1).permute(1, 0, 2, 3)
    wk, k = prep_w(w)
    y = H.vae_conv3d(cl(x), wk, b.cuda(), k, pad_hw=1, upsample2=True)
    torch.testing.assert_close(cf(y), ref, rtol=RTOL, atol=ATOL)
    w2 = syn.uniform("ud.w2", (c, c, 3, 3), (9 * c) ** -0.5, 3)
    ref = F.conv2d(F.pad(x.permute(1, 0, 2, 3), (0, 1, 0, 1)), w2, None, stride=2).permute(1, 0, 2, 3)
    wk, k = prep_w(w2)
    y = H.vae_conv3d(cl(x), wk, None, k, stride_hw=2, pad_hw=0)
    torch.testing.assert_close(cf(y), ref, rtol=RTOL, atol=ATOL)


@pytest.mark.skipif(EXACT, reason="split weights belong to the split-bf16 path")
@pytest.mark.parametrize("cin,cout,T,Hh,Ww,nctx", [(32, 32, 3, 9, 17, 2), (64, 96, 1, 8, 16, 1), (96, 96, 2, 33, 18, 2),
                                                   (96, 3, 2, 10, 33, 0), (128, 160, 1, 4, 6, 2), (16, 128, 2, 5, 7, 2)])
def test_split_weight_and_presplit_routes(H, cin, cout, T, Hh, Ww, nctx):
    """The producer / consumer kernel: (a) pre-split weights + fp32 input (norm, SiLU, residual fused) against the
    reference arithmetic; (b) the pre-activated, pre-split input route, bit-identical to (a).  Patches of 16 x 16 that
    the frame does not fill, channel counts that are no multiple of the 96 / 64 / 32 a workgroup makes, 0 - 2 context
    frames."""
    x = syn.randn("sw.x", (cin, T, Hh, Ww), 6)
    ctx = syn.randn("sw.ctx", (cin, 2, Hh, Ww), 6)
    g = 1 + syn.randn("sw.g", (cin,), 6, 0.1)
    w = syn.uniform("sw.w", (cout, cin, 3, 3, 3), (27 * cin) ** -0.5, 6)
    b = syn.randn("sw.b", (cout,), 6, 0.1)
    res = syn.randn("sw.r", (cout, T, Hh, Ww), 6)
    act = lambda t: F.silu(V.rms_norm(t, g))
    ref = V.causal_conv(act(x), w, b, act(ctx[:, 2 - nctx:]) if nctx else None) + res
    wk, k = prep_w(w)
    sw = H.SplitWeight(wk)
    xc, cc = cl(x), cl(ctx)
    rn, rc = H.vae_voxel_rnorm(xc), H.vae_voxel_rnorm(cc)
    p2, p1 = (cc[0] if nctx == 2 else None), (cc[1] if nctx >= 1 else None)
    y = H.vae_conv3d(xc, sw, b.cuda(), k, prev2=p2, prev1=p1, rnorm=rn, rnorm_prev2=None if p2 is None else rc[0],
                     rnorm_prev1=None if p1 is None else rc[1], gamma=g.cuda(), silu=True, resid=cl(res), pad_t=2, pad_hw=1)
    torch.testing.assert_close(cf(y), ref, rtol=RTOL, atol=ATOL)
    if cin % 32 == 0:
        xs, cs = H.vae_act_split(xc, g.cuda(), True), H.vae_act_split(cc, g.cuda(), True)
        y2 = H.vae_conv3d_ps(xs, sw, b.cuda(), k, prev2=None if p2 is None else cs[0], prev1=None if p1 is None else cs[1],
                             resid=cl(res), pad_t=2, pad_hw=1)
        assert torch.equal(y2, y)


@pytest.mark.skipif(EXACT, reason="split weights belong to the split-bf16 path")
@pytest.mark.parametrize("Hh,Ww", [(6, 10), (8, 16), (5, 9), (17, 12)])
def test_upsample_conv_on_split_routes(H, Hh, Ww):
    """Resample's Upsample(2x nearest) + Conv2d on pre-split weights, from the fp32 image and from its pre-split form
    (the route the decoder takes): equal to each other bit for bit and to torch's fp32 result within the path's tolerance."""
    c = 64
    x = syn.randn("us.x", (c, 2, Hh, Ww), 3)
    w = syn.uniform("us.w", (c // 2, c, 3, 3), (9 * c) ** -0.5, 3)
    b = syn.randn("us.b", (c // 2,), 3, 0.1)
    up = F.interpolate(x.permute(1, 0, 2, 3), scale_factor=(2.0, 2.0), mode="nearest-exact")
    ref = F.conv2d(up, w, b, padding=1).permute(1, 0, 2, 3)
    wk, k = prep_w(w)
    sw = H.SplitWeight(wk)
    y = H.vae_conv3d(cl(x), sw, b.cuda(), k, pad_hw=1, upsample2=True)
    torch.testing.assert_close(cf(y), ref, rtol=RTOL, atol=ATOL)
    y2 = H.vae_conv3d_ps(H.vae_act_split(cl(x)), sw, b.cuda(), k, pad_hw=1, upsample2=True)
    assert torch.equal(y2, y)


def test_time_convs(H):
    c, T, Hh, Ww = 32, 3, 5, 6
    x = syn.randn("tc.x", (c, T, Hh, Ww), 4)
    ctx = syn.randn("tc.ctx", (c, 2, Hh, Ww), 4)
    # temporal upsampler: (3,1,1) conv to 2C channels, frames interleaved (vae2_1.py:127-136)
    w = syn.uniform("tc.w", (2 * c, c, 3, 1, 1), (3 * c) ** -0.5, 4)
    b = syn.randn("tc.b", (2 * c,), 4, 0.1)
    y = V.causal_conv(x, w, b, ctx).reshape(2, c, T, Hh, Ww)
    ref = torch.stack((y[0], y[1]), dim=2).reshape(c, 2 * T, Hh, Ww)
    wk, k = prep_w(w)
    cc = cl(ctx)
    got = H.vae_conv3d(cl(x), wk, b.cuda(), k, prev2=cc[0], prev1=cc[1], pad_t=2, time_interleave=True)
    torch.testing.assert_close(cf(got), ref, rtol=RTOL, atol=ATOL)
    # temporal downsampler: (3,1,1) stride 2 over [last cached frame | chunk] (vae2_1.py:155-156)
    x4 = syn.randn("tc.x4", (c, 4, Hh, Ww), 4)
    w2 = syn.uniform("tc.w2", (c, c, 3, 1, 1), (3 * c) ** -0.5, 4)
    ref = F.conv3d(torch.cat([ctx[:, -1:], x4], 1)[None], w2, None, stride=(2, 1, 1))[0]
    wk, k = prep_w(w2)
    got = H.vae_conv3d(cl(x4), wk, None, k, prev1=cc[1], stride_t=2, pad_t=1)
    torch.testing.assert_close(cf(got), ref, rtol=RTOL, atol=ATOL)


def test_layout_and_softmax(H):
    z = syn.randn("lo.z", (16, 3, 4, 5), 5)
    a, m, p = syn.randn("lo.a", (16,), 5), syn.randn("lo.m", (16,), 5), syn.randn("lo.p", (16,), 5)
    y = H.vae_to_channels_last(z.cuda(), c_pad=20, pre=a.cuda(), mul=m.cuda(), post=p.cuda())
    ref = ((z + a.view(-1, 1, 1, 1)) * m.view(-1, 1, 1, 1) + p.view(-1, 1, 1, 1)).permute(1, 2, 3, 0)
    torch.testing.assert_close(y.cpu()[..., :16], ref, rtol=1e-6, atol=1e-6)
    assert float(y[..., 16:].abs().max()) == 0.0
    back = H.vae_from_channels_last(y, channels=16, clamp=True)
    torch.testing.assert_close(back.cpu(), ref.permute(3, 0, 1, 2).clamp(-1, 1), rtol=1e-6, atol=1e-6)
    s = syn.randn("lo.s", (37, 300), 5, 3.0)
    got = H.softmax_rows(s.cuda().clone(), 0.25)
    torch.testing.assert_close(got.cpu(), torch.softmax(s * 0.25, dim=-1), rtol=1e-5, atol=1e-7)


def test_conv_rejects(H):
    x = torch.zeros(1, 4, 4, 6, device="cuda")
    with pytest.raises(RuntimeError, match="multiple of 4"):
        H.vae_conv3d(x, torch.zeros(27, 8, 6, device="cuda"), None, (3, 3, 3), pad_t=2, pad_hw=1)
    with pytest.raises(ValueError):
        H.vae_conv3d(x[..., :4].contiguous(), torch.zeros(9, 8, 4, device="cuda"), None, (3, 3, 3))


# --------------------------------------------------------------------------------- whole model

@pytest.fixture(scope="module")
def G(golden_dir):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, "vae_tiny.npz")).items()}


@pytest.fixture(scope="module")
def vae():
    from omnivideo.modules.vae2_1 import Wan2_1_VAE
    cfg = syn.vae_config("tiny")
    v = Wan2_1_VAE(z_dim=cfg["z_dim"], vae_pth=None, device="cuda", dim=cfg["dim"])
    sd = syn.make_vae_state_dict(cfg, 0)
    v.model.load_state_dict(sd, strict=True)
    lay = V.vae_layout(cfg["dim"], cfg["z_dim"], cfg["dim_mult"], cfg["num_res_blocks"], cfg["temperal_downsample"])
    return v, sd, lay


def test_state_dict_keys(vae):
    v, sd, _ = vae
    assert set(v.model.state_dict().keys()) == set(sd.keys())


@pytest.mark.parametrize("frames,hw", [(5, (32, 48)), (9, (32, 32)), (1, (16, 16))])
def test_encode(vae, G, frames, hw):
    v, sd, lay = vae
    vid = syn.randn(f"vae.video{frames}", (3, frames, *hw), 5).clamp(-1, 1)
    z = v.encode([vid.cuda()])[0]
    torch.testing.assert_close(z.cpu(), G[f"encode_{frames}"], rtol=RTOL_NET, atol=ATOL_NET)
    with torch.no_grad():
        torch.testing.assert_close(z.cpu(), V.encode(vid, sd, lay), rtol=RTOL_NET, atol=ATOL_NET)


@pytest.mark.parametrize("tl,hw", [(2, (4, 6)), (3, (4, 4)), (1, (2, 2))])
def test_decode(vae, G, tl, hw):
    v, sd, lay = vae
    z = syn.randn(f"vae.z{tl}", (16, tl, *hw), 5)
    x = v.decode([z.cuda()])[0]
    assert tuple(x.shape) == (3, 4 * tl - 3, 8 * hw[0], 8 * hw[1])
    torch.testing.assert_close(x.cpu(), G[f"decode_{tl}"], rtol=RTOL_NET, atol=ATOL_NET)


@pytest.mark.parametrize("mode", ["downsample2d", "downsample3d", "upsample2d", "upsample3d"])
def test_resample_modes_streamed_over_three_chunks(vae, golden_dir, mode):
    """The runner's Resample paths (first-chunk pass-through of the temporal down-sampling; 'Rep' / zero-context
    / cached-context sequence of the temporal up-sampling), three consecutive chunks each, against the
    reference module's own output (tests/golden/vae_resample.npz; reference vae2_1.py:102-158)."""
    v, sd, lay = vae
    GR = dict(np.load(os.path.join(golden_dir, "vae_resample.npz")).items())
    name, want = str(GR[mode + "_module"]), torch.from_numpy(GR[mode])
    chunks = ((0, 1), (1, 5), (5, 7)) if mode.startswith("down") else ((0, 1), (1, 2), (2, 3))
    layer = v.model.get_submodule(name)
    assert layer.mode == mode
    cin = sd[name + ".resample.1.weight"].shape[1]
    x = syn.randn(f"vae.resample.{mode}", (1, cin, chunks[-1][1], 6, 8), 5)[0]
    run = v.model.runner()
    run.reset()
    fn = run.downsample if mode.startswith("down") else run.upsample
    with torch.no_grad():
        ys = [fn(cl(x[:, a:b]), layer, name) for a, b in chunks]
    run.reset()
    got = cf(torch.cat(ys, dim=0))
    assert tuple(got.shape) == tuple(want.shape)
    torch.testing.assert_close(got, want, rtol=RTOL, atol=ATOL)


def test_decode_encode_round_trip_shapes(vae):
    """decode -> encode keeps the latent geometry (size-independent property)."""
    v, _, _ = vae
    z = syn.randn("vae.rt", (16, 3, 4, 6), 6)
    x = v.decode([z.cuda()])[0]
    z2 = v.encode([x])[0]
    assert tuple(z2.shape) == tuple(z.shape) and torch.isfinite(z2).all()


def test_video_to_uint8_matches_reference_formula():
    """uint8 frames: clamp, (x - lo) / (hi - lo), * 255, truncate (reference utils/utils.py:177-184
    with make_grid's normalize for one video), from planar and from channels-last input."""
    import omnivideo._hip as H
    from omnivideo.utils.utils import video_to_frames, cache_video
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(3, 5, 34, 50, device="cuda", generator=g) * 0.8
    x[0, 0, 0, :6] = torch.tensor([-1.0, 1.0, 0.0, -5.0, 5.0, 1.0 - 2.0 ** -24], device="cuda")
    for lo, hi in ((-1.0, 1.0), (0.0, 1.0), (-0.5, 2.0)):
        t = x.clamp(lo, hi)
        want = ((t - lo) / max(hi - lo, 1e-5) * 255).type(torch.uint8).permute(1, 2, 3, 0).contiguous()
        got = video_to_frames(x[None], value_range=(lo, hi))
        assert got.dtype == torch.uint8 and torch.equal(got, want)
        xcl = torch.zeros(5, 34, 50, 4, device="cuda")
        xcl[..., :3] = x.permute(1, 2, 3, 0)
        assert torch.equal(H.video_to_uint8(xcl, channels_last=True, value_range=(lo, hi)), want)
    seen = {}
    name = cache_video(x[None], save_file="/tmp/unused.mp4", fps=16, writer=lambda p, f, fps: seen.update(p=p, f=f, fps=fps))
    assert name == "/tmp/unused.mp4" and seen["fps"] == 16 and seen["f"].shape == (5, 34, 50, 3)
    assert cache_video(x[None], retry=2, writer=lambda p, f, fps: 1 / 0) is None


def test_decode_frames_equals_decode_then_convert(vae):
    v, _, _ = vae
    z = syn.randn("vae.z3", (16, 3, 4, 4), 5).cuda()
    vid = v.decode([z])[0]
    frames = v.decode_frames([z])[0]
    want = ((vid.clamp(-1, 1) + 1.0) / 2.0 * 255).type(torch.uint8).permute(1, 2, 3, 0)
    assert torch.equal(frames, want)


@pytest.mark.parametrize("name", list(syn.FRAME_CASES))
def test_transform_frames_to_tensor_bit_exact(golden_dir, name):
    """crop + Pillow-exact resize + ToTensor/Normalize on the device vs Pillow's bytes (golden) mapped
    through the reference's fp32 arithmetic, and vs the CPU restatement: bit for bit."""
    from omnivideo.utils.utils import transform_frames_to_tensor
    from oracle import video as OV
    T, Hh, W, th, tw = syn.FRAME_CASES[name]
    frames = syn.make_frames(name, T, Hh, W)
    got = transform_frames_to_tensor(frames, (th, tw))
    assert tuple(got.shape) == (T, 3, th, tw) and got.transpose(0, 1).is_contiguous()
    gold = torch.from_numpy(np.load(os.path.join(golden_dir, "frames.npz"))[name].astype(np.int64))
    want = OV.normalize_lut()[gold].permute(0, 3, 1, 2)
    assert torch.equal(got.cpu(), want)
    assert torch.equal(got.cpu(), OV.transform_frames_to_tensor(frames, (th, tw)))


def test_transform_frames_full_size_properties():
    """720p source -> 480p target at 49 frames: constant frames stay constant, the result is within
    [-1, 1], frames are independent, and sampled frames equal the CPU restatement."""
    from omnivideo.utils.utils import transform_frames_to_tensor
    from oracle import video as OV
    g = torch.Generator().manual_seed(5)
    frames = torch.randint(0, 256, (49, 720, 1280, 3), generator=g, dtype=torch.uint8)
    frames[7] = 93
    out = transform_frames_to_tensor(frames, (480, 832))
    assert tuple(out.shape) == (49, 3, 480, 832)
    lut = OV.normalize_lut()
    assert torch.equal(out[7].cpu(), torch.full((3, 480, 832), float(lut[93])))
    assert float(out.min()) >= -1.0 and float(out.max()) <= 1.0
    for i in (0, 48):
        assert torch.equal(out[i].cpu(), OV.transform_frames_to_tensor(frames[i:i + 1].numpy(), (480, 832))[0])
    assert torch.equal(transform_frames_to_tensor(frames[20:22], (480, 832)), out[20:22])


def test_frames_resize_rejects_bad_input():
    import omnivideo._hip as H
    lut = torch.zeros(256, device="cuda")
    with pytest.raises(ValueError):
        H.frames_resize_normalize(torch.zeros(1, 8, 8, 3, device="cuda"), (0, 0, 8, 8), (4, 4), lut)
    with pytest.raises(RuntimeError, match="crop window"):
        H.frames_resize_normalize(torch.zeros(1, 8, 8, 3, dtype=torch.uint8, device="cuda"), (2, 0, 8, 8), (4, 4), lut)


def test_conv333_large_frame(H):
    """A large frame (thousands of tiles, several grid rounds), ragged at both edges."""
    cin, cout, T, Hh, Ww = 32, 96, 2, 250, 421
    x = syn.randn("cvl.x", (cin, T, Hh, Ww), 1)
    ctx = syn.randn("cvl.ctx", (cin, 2, Hh, Ww), 1)
    w = syn.uniform("cvl.w", (cout, cin, 3, 3, 3), (27 * cin) ** -0.5, 1)
    b = syn.randn("cvl.b", (cout,), 1, 0.1)
    wk, k = prep_w(w)
    ref = V.causal_conv(x, w, b, ctx)
    c = cl(ctx)
    y = H.vae_conv3d(cl(x), wk, b.cuda(), k, prev2=c[0], prev1=c[1], pad_t=2, pad_hw=1)
    # 20 M outputs, each a sum of 864 products: the far tail of the summation-order differences of two
    # fp32 sums reaches 2.5e-6 next to a zero crossing, hence twice the one-layer absolute tolerance
    torch.testing.assert_close(cf(y), ref, rtol=RTOL, atol=2 * ATOL)


@pytest.mark.skipif(EXACT, reason="this is the child process")
def test_exact_fp32_path_in_fresh_process():
    """The same layer / encode / decode tests on the exact-fp32 matrix path at its tighter tolerance."""
    import subprocess
    import sys
    env = dict(os.environ, OVD_VAE_EXACT_FP32="1")
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k",
                          "conv or encode or decode", "-p", "no:cacheprovider"], env=env, capture_output=True, text=True,
                         timeout=600)
    assert res.returncode == 0, res.stdout[-4000:] + res.stderr[-2000:]
    assert " passed" in res.stdout and "failed" not in res.stdout, res.stdout[-1000:]
