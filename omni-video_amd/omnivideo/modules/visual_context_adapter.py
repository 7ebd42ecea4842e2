"""Source-latent context adapter (reference: omnivideo/modules/visual_context_adapter.py):
Conv3d patchify (1,4,4) -> LayerNorm -> Linear(hidden -> out_dim) -> LayerNorm, run as
ovd_patchify + ovd_gemm_bf16 + ovd_layernorm_modulate_* launches."""
import math

import torch
import torch.nn as nn

from .. import _hip as H

BF16, F32 = torch.bfloat16, torch.float32


class VisualContextAdapter(nn.Module):

    def __init__(self, patch_size=(1, 4, 4), in_channels=16, hidden_dim=2048, out_dim=4098, eps=1e-6):
        super().__init__()
        self.patch_size, self.in_channels = tuple(patch_size), in_channels
        self.hidden_dim, self.out_dim, self.eps = hidden_dim, out_dim, eps
        self.patch_embedding = nn.Conv3d(in_channels, hidden_dim, kernel_size=self.patch_size,
                                         stride=self.patch_size)
        self.norm1 = nn.LayerNorm(hidden_dim, eps=eps, elementwise_affine=False)
        self.projection = nn.Linear(hidden_dim, out_dim)
        self.norm2 = nn.LayerNorm(out_dim, eps=eps, elementwise_affine=False)
        self.apply(self._init_weights)

    def _init_weights(self, module):
        # reference visual_context_adapter.py:64-75
        if isinstance(module, nn.Conv3d):
            nn.init.xavier_uniform_(module.weight.flatten(1))
            if module.bias is not None:
                nn.init.zeros_(module.bias)
        elif isinstance(module, nn.Linear):
            nn.init.xavier_uniform_(module.weight)
            if module.bias is not None:
                nn.init.zeros_(module.bias)

    def forward(self, x):
        """x: list of [C,F,H,W] | [B,C,F,H,W] | [C,F,H,W] -> fp32 [B, N, out_dim]
        (reference visual_context_adapter.py:77-116; the last LayerNorm yields fp32 under
        autocast)."""
        if isinstance(x, list):
            x = torch.stack(x)
        elif x.dim() == 4:
            x = x.unsqueeze(0)
        w = self.patch_embedding.weight.data
        dev = w.device
        if dev.type != "cuda" or w.dtype != BF16:
            raise RuntimeError("VisualContextAdapter needs bf16 parameters on a HIP device")
        B, Cin, T, Hh, Ww = x.shape
        pt, ph, pw = self.patch_size
        n = (T // pt) * (Hh // ph) * (Ww // pw)
        k = Cin * pt * ph * pw
        out = torch.empty(B, n, self.out_dim, dtype=F32, device=dev)
        rows = torch.empty(n, k, dtype=BF16, device=dev)
        a = torch.empty(n, self.hidden_dim, dtype=F32, device=dev)
        hb = torch.empty(n, self.hidden_dim, dtype=BF16, device=dev)
        p = torch.empty(n, self.out_dim, dtype=F32, device=dev)
        for b in range(B):
            xb = x[b].to(dev)
            xb = xb.contiguous() if xb.dtype in (BF16, F32) else xb.float().contiguous()
            H.patchify(xb, rows, self.patch_size)
            H.gemm(rows, w.reshape(self.hidden_dim, k), a, self.patch_embedding.bias.data, H.EPI_F32)
            H.layernorm_modulate(a, hb, self.eps)
            H.gemm(hb, self.projection.weight.data, p, self.projection.bias.data, H.EPI_F32)
            H.layernorm_modulate(p, out[b], self.eps)
        return out
