#!/usr/bin/env python
"""Per-layer timing of the VAE decode (and encode): every ovd_vae_conv3d_f32 call timed with HIP events and
grouped by shape.  Development aid."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "omni-video_amd"))
import torch                                   # noqa: E402
from omnivideo import _hip as H                # noqa: E402
from omnivideo.modules.vae2_1 import Wan2_1_VAE  # noqa: E402

agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
orig = H.vae_conv3d


def timed(x, w, bias, kernel, **kw):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    y = orig(x, w, bias, kernel, **kw)
    b.record()
    torch.cuda.synchronize()
    key = (tuple(x.shape[1:]), w.shape[1], tuple(kernel), bool(kw.get("upsample2")), kw.get("stride_hw", 1),
           kw.get("rnorm") is not None)
    e = agg[key]
    e[0] += 1
    e[1] += a.elapsed_time(b)
    e[2] += 2.0 * y.numel() / y.shape[-1] * w.shape[0] * w.shape[1] * w.shape[2]
    return y


def report(title):
    tot = sum(v[1] for v in agg.values())
    print(f"--- {title}: {tot:.1f} ms in conv kernels")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"in[H,W,C]={k[0]} Cout={k[1]} k={k[2]} ups={int(k[3])} s={k[4]} norm={int(k[5])}: n={v[0]:4d} "
              f"{v[1]:8.1f} ms  {v[2] / v[1] / 1e9:7.1f} TFLOP/s (fp32 work)")
    agg.clear()


if __name__ == "__main__":
    with torch.device("cuda"):
        vae = Wan2_1_VAE(z_dim=16, vae_pth=None, device="cuda")
    z = torch.randn(16, 13, 60, 104, device="cuda")
    x = vae.decode([z])[0]
    H.vae_conv3d = timed
    import omnivideo.modules.vae2_1 as V
    V.H.vae_conv3d = timed
    x = vae.decode([z])[0]
    report("decode 49f 480p")
    vae.encode([x])
    report("encode 49f 480p")
