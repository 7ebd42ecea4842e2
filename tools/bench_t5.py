#!/usr/bin/env python
"""Times the umT5-XXL encoder (random weights) on one prompt of 512 tokens.  Development aid."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "omni-video_amd"))
import torch
from omnivideo.modules.t5 import umt5_xxl

m = umt5_xxl(encoder_only=True, dtype=torch.bfloat16, device="cuda").eval()
for p in m.parameters():
    if p.dim() == 1:
        p.data.fill_(1.0)
    else:
        p.data.normal_(0, p.shape[-1] ** -0.5 if p.shape[0] != 32 else 0.5)
ids = torch.randint(0, 256384, (1, 512), device="cuda")
mask = (torch.arange(512, device="cuda")[None] < 77).long()
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    y = m(ids, mask)
    torch.cuda.synchronize()
    print(f"umT5-XXL encode [1, 512]: {(time.time() - t0) * 1e3:.1f} ms, finite={bool(torch.isfinite(y).all())}, "
          f"|y| max {float(y.abs().max()):.2f}", flush=True)
