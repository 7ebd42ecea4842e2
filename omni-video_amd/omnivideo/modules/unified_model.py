"""Mixed-condition wrapper around the DiT (reference: omnivideo/modules/unified_model.py):
projects VLM hidden states, runs the visual context adapter on the source latents, tightly
concatenates [VLM | <prp_st> text <prp_ed> | <img_st> visual <img_ed>], truncates to
max_context_len and calls the DiT.

None of that depends on the latents or the timestep, so the assembled context is cached
per set of input tensors and the DiT's own cache (embedded context + per-layer
cross-attention K/V) is hit on every sampling step after the first."""
import logging
import os
import random
from typing import Dict, List, Optional, Union

import torch
import torch.nn as nn

from .. import _hip as H
from .model import WanModel, WanRMSNorm
from .visual_context_adapter import VisualContextAdapter

BF16, F32 = torch.bfloat16, torch.float32


def _ident(t):
    if t is None:
        return None
    if isinstance(t, (list, tuple)):
        return tuple(_ident(u) for u in t)
    if isinstance(t, dict):
        return tuple((k, _ident(v)) for k, v in sorted(t.items()))
    return (t.data_ptr(), t._version, tuple(t.shape), str(t.dtype))


class UnifiedWanWithMixedConditionModel(nn.Module):

    def __init__(self, wan_model_or_ckpt_dir: Union[WanModel, str, None] = None, wan_subfolder: Optional[str] = None,
                 vlm_in_dim: int = 4096, precision_dtype: torch.dtype = torch.float32, device_id: int = 0,
                 rank: int = 0, dit_fsdp: bool = False, use_usp: bool = False,
                 use_visual_context_adapter: bool = False, visual_context_adapter_patch_size: tuple = None,
                 max_context_len: int = None, eps: float = 1e-6, wan_config: Optional[Dict] = None,
                 skip_init: bool = False):
        super().__init__()
        if isinstance(wan_model_or_ckpt_dir, str):
            # reference unified_model.py:42-43: a diffusers-layout directory (config.json + weights)
            self.wan_model = WanModel.from_pretrained(wan_model_or_ckpt_dir, subfolder=wan_subfolder)
        elif isinstance(wan_model_or_ckpt_dir, WanModel):
            self.wan_model = wan_model_or_ckpt_dir
        elif wan_config is not None:
            # reference unified_model.py:46-69
            logging.info(f"Initializing WanModel with random weights from config (skip_init={skip_init})")
            keys = dict(patch_size=(1, 2, 2), dim=2048, ffn_dim=8192, freq_dim=256, num_heads=16, num_layers=32,
                        window_size=(-1, -1), qk_norm=True, cross_attn_norm=True, eps=1e-6)
            params = {k: wan_config.get(k, d) for k, d in keys.items()}
            for k in ("model_type", "text_len", "in_dim", "text_dim", "out_dim"):
                if k in wan_config:
                    params[k] = wan_config[k]
            self.wan_model = WanModel(**params, skip_init=skip_init)
        else:
            raise ValueError("Either wan_model_or_ckpt_dir or wan_config must be provided")

        if max_context_len is not None:
            self.wan_model.text_len = max_context_len
        self.max_context_len = self.wan_model.text_len
        self.vlm_norm = WanRMSNorm(vlm_in_dim, eps=eps)
        self.vlm_proj = nn.Linear(vlm_in_dim, self.wan_model.text_dim)
        if use_visual_context_adapter:
            self.visual_context_adapter = VisualContextAdapter(
                patch_size=self.wan_model.patch_size if visual_context_adapter_patch_size is None
                else visual_context_adapter_patch_size,
                in_channels=self.wan_model.in_dim, hidden_dim=self.wan_model.dim,
                out_dim=self.wan_model.text_dim, eps=eps)
        else:
            self.visual_context_adapter = None
        self._mixed_key = None
        self._mixed = None

    @classmethod
    def from_pretrained(cls, wan_ckpt_dir, wan_subfolder=None, **kw):
        return cls(wan_model_or_ckpt_dir=wan_ckpt_dir, wan_subfolder=wan_subfolder, **kw)

    def enable_gradient_checkpointing(self):
        self.wan_model.enable_gradient_checkpointing()

    def reset_wan_text_len(self, new_len: int):
        self.wan_model.text_len = new_len

    # ------------------------------------------------------------------ context pieces
    def _project_vlm(self, ar):
        """WanRMSNorm + Linear (reference unified_model.py:196-206); ar [n, vlm_dim]."""
        w = self.vlm_proj.weight.data
        a = ar.reshape(-1, ar.shape[-1]).to(w.device, BF16).contiguous()
        nrm = torch.empty_like(a)
        H.rmsnorm_rope(a, self.vlm_norm.weight.data, nrm, 0, self.vlm_norm.eps)
        out = torch.empty(a.shape[0], w.shape[0], dtype=BF16, device=w.device)
        H.gemm(nrm, w, out, self.vlm_proj.bias.data)
        return out

    def build_mixed_context(self, batch_size, context=None, ar_vision_input=None, visual_emb=None,
                            special_token_dict=None):
        """Tight concatenation per sample (reference unified_model.py:195-276)."""
        dev = self.vlm_proj.weight.device
        text_dim = self.wan_model.text_dim

        def pick(item, idx=None):
            # reference unified_model.py:225-241
            if item is None:
                return None
            if isinstance(item, list) and idx is not None:
                item = item[idx]
            if isinstance(item, torch.Tensor):
                if idx is not None and item.dim() >= 3 and item.size(0) == batch_size:
                    item = item[idx]
                if item.dim() == 3:
                    item = item[0]
                elif item.dim() == 1:
                    item = item.unsqueeze(0)
            return item

        vlm_out = None
        if ar_vision_input is not None:
            if isinstance(ar_vision_input, list):
                vlm_out = [None if a is None else self._project_vlm(a) for a in ar_vision_input]
            else:
                vlm_out = self._project_vlm(ar_vision_input).reshape(*ar_vision_input.shape[:-1], text_dim)
        vis_out = None
        if visual_emb is not None and self.visual_context_adapter is not None:
            if isinstance(visual_emb, list):
                vis_out = [None if v is None else self.visual_context_adapter(v).squeeze(0) for v in visual_emb]
            else:
                vis_out = self.visual_context_adapter(visual_emb)
        elif visual_emb is not None:
            vis_out = [None] * batch_size
        mixed = []
        for idx in range(batch_size):
            parts = []
            vlm_item, ctx_item, vis_item = pick(vlm_out, idx), pick(context, idx), pick(vis_out, idx)
            if special_token_dict is not None:
                tok = {k: pick(special_token_dict.get(k)) for k in ("<img_st>", "<img_ed>", "<prp_st>", "<prp_ed>")}
                if vlm_item is not None:
                    parts.append(vlm_item)
                if ctx_item is not None:
                    parts += [tok["<prp_st>"], ctx_item, tok["<prp_ed>"]]
                if vis_item is not None:
                    parts += [tok["<img_st>"], vis_item, tok["<img_ed>"]]
            else:
                parts = [p for p in (vlm_item, ctx_item, vis_item) if p is not None]
            parts = [p for p in parts if p is not None]
            if parts:
                new_ctx = torch.cat([p.to(dev) for p in parts], dim=0)
                if self.max_context_len is not None and new_ctx.shape[0] > self.max_context_len:
                    new_ctx = new_ctx[:self.max_context_len]
                mixed.append(new_ctx)
            else:
                mixed.append(torch.zeros(1, text_dim, device=dev, dtype=BF16))
        return mixed

    def forward(self, x, t, context=None, aligned_emb=None, ar_vision_input=None, visual_emb=None, seq_len=None,
                special_token_dict=None, classifier_free_ratio=None, unconditioned_context=None,
                condition_mode="auto"):
        """reference unified_model.py:138-278 (inference behaviour: classifier_free_ratio of
        None or 0 never drops a condition)."""
        if condition_mode == "auto":
            condition_mode = "full"
        if condition_mode != "full":
            return None     # the reference falls through without a return for other modes
        batch_size = x.size(0) if isinstance(x, torch.Tensor) else len(x)
        # the reference draws 2*B python randoms per call, also at inference
        # (unified_model.py:161-162): keep the global RNG stream in step with it
        for _ in range(2 * batch_size):
            random.random()
        if classifier_free_ratio is not None and classifier_free_ratio > 0:
            raise NotImplementedError("training-time condition dropout is not part of the sampling path")
        key = (_ident(context), _ident(ar_vision_input), _ident(visual_emb), _ident(special_token_dict),
               self.max_context_len, tuple((p.data_ptr(), p._version) for n, p in self.named_parameters()
                                           if not n.startswith("wan_model.")))
        if key != self._mixed_key:
            self._mixed = self.build_mixed_context(batch_size, context, ar_vision_input, visual_emb,
                                                   special_token_dict)
            self._mixed_key = key
            self._mixed_refs = (context, ar_vision_input, visual_emb, special_token_dict)
        return self.wan_model(x, t=t, context=self._mixed, seq_len=seq_len)

    def save_pretrained(self, save_dir: str):
        """reference unified_model.py:280-285: the DiT as a diffusers-layout directory `wan_model/`, the
        condition builder's own modules as state-dict archives beside it."""
        os.makedirs(save_dir, exist_ok=True)
        self.wan_model.save_pretrained(os.path.join(save_dir, "wan_model"))
        if self.visual_context_adapter:
            torch.save(self.visual_context_adapter.state_dict(),
                       os.path.join(save_dir, "visual_context_adapter_pytorch_model.bin"))
        torch.save(self.vlm_norm.state_dict(), os.path.join(save_dir, "vlm_norm_pytorch_model.bin"))
        torch.save(self.vlm_proj.state_dict(), os.path.join(save_dir, "vlm_proj_pytorch_model.bin"))

    def enable_eval(self):
        self.eval()

    def enable_train(self):
        raise NotImplementedError("this package implements the sampling path only")
