"""Wan-style video DiT with the reference's module names, parameter names and call
signatures (reference: omnivideo/modules/model.py), executed by hand-written HIP kernels.

The nn.Module tree exists to own parameters under the checkpoint's key names
(`blocks.{i}.self_attn.q.weight`, ... so `load_state_dict` of released weights works);
no torch compute op runs in any forward here.  Parameters are expected in bf16 on a HIP
device (what the reference's pipeline does: x2x_gen_unified.py:126), activations follow
the reference's mixed-precision flow: bf16 GEMM/attention operands, fp32 residual stream,
norms, modulation and head.
"""
import math

import torch
import torch.nn as nn

from .. import _hip as H
from .attention import flash_attention
from .engine import DiTEngine, build_rope_table

__all__ = ["WanModel"]

BF16, F32 = torch.bfloat16, torch.float32


def sinusoidal_embedding_1d(dim, position):
    """[cos | sin] features in float64 (reference model.py:13-23). position: 1-D tensor
    on the HIP device."""
    assert dim % 2 == 0
    pos = position.to(torch.float64).reshape(-1).contiguous()
    out = torch.empty(pos.numel(), dim, dtype=F32, device=pos.device)
    for i in range(pos.numel()):
        H.time_sinusoid(pos, out[i], step=i)
    return out.to(torch.float64)


def rope_params(max_seq_len, dim, theta=10000):
    """Complex rotation table [max_seq_len, dim/2] (reference model.py:26-34); host-side
    float64, used to seed the device cos/sin tables."""
    assert dim % 2 == 0
    freqs = torch.outer(torch.arange(max_seq_len, dtype=torch.float64),
                        1.0 / torch.pow(theta, torch.arange(0, dim, 2, dtype=torch.float64) / dim))
    return torch.polar(torch.ones_like(freqs), freqs)


def _as_bf16_rows(x):
    return x if x.dtype == BF16 else x.to(BF16)


def rope_apply(x, grid_sizes, freqs=None):
    """3-axis RoPE on [B, L, N, D] (reference model.py:37-65); returns float32 like the
    reference.  `freqs` is accepted for signature compatibility; the device table is
    rebuilt from the same formula."""
    B, L, N, D = x.shape
    xb = _as_bf16_rows(x).reshape(B, L, N * D).contiguous()
    out = torch.empty_like(xb)
    for b in range(B):
        grid = tuple(int(v) for v in grid_sizes[b].tolist())
        H.rmsnorm_rope(xb[b], None, out[b], D, 0.0, rope=build_rope_table(D, max(grid), x.device),
                       rows_per_sample=L, n_rope=grid[0] * grid[1] * grid[2], grid=grid)
    # the rotated value is produced in bf16: the attention entry casts it to bf16 next
    # anyway (reference attention.py:59-83)
    return out.view(B, L, N, D).float()


class WanRMSNorm(nn.Module):
    """reference model.py:68-84"""

    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.dim, self.eps = dim, eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        xb = _as_bf16_rows(x).contiguous()
        out = torch.empty_like(xb)
        H.rmsnorm_rope(xb, self.weight.data, out, head_dim=0, eps=self.eps)
        return out


class WanLayerNorm(nn.LayerNorm):
    """reference model.py:87-97 (fp32 statistics, result in the input dtype)"""

    def __init__(self, dim, eps=1e-6, elementwise_affine=False):
        super().__init__(dim, elementwise_affine=elementwise_affine, eps=eps)

    def forward(self, x):
        xf = x.to(F32).contiguous()
        out = torch.empty_like(xf, dtype=x.dtype if x.dtype in (BF16, F32) else F32)
        H.layernorm_modulate(xf, out, self.eps, weight=self.weight.data if self.elementwise_affine else None,
                             bias=self.bias.data if self.elementwise_affine else None)
        return out


class WanSelfAttention(nn.Module):
    """reference model.py:100-154"""

    def __init__(self, dim, num_heads, window_size=(-1, -1), qk_norm=True, eps=1e-6):
        assert dim % num_heads == 0
        super().__init__()
        self.dim, self.num_heads, self.head_dim = dim, num_heads, dim // num_heads
        self.window_size, self.qk_norm, self.eps = window_size, qk_norm, eps
        self.q, self.k, self.v, self.o = (nn.Linear(dim, dim) for _ in range(4))
        self.norm_q = WanRMSNorm(dim, eps=eps) if qk_norm else nn.Identity()
        self.norm_k = WanRMSNorm(dim, eps=eps) if qk_norm else nn.Identity()

    def _check(self):
        if not self.qk_norm or tuple(self.window_size) != (-1, -1):
            raise NotImplementedError("only qk_norm=True with global attention is built (the released configs)")

    def forward(self, x, seq_lens, grid_sizes, freqs=None):
        """x [B, L, C] -> [B, L, C] bf16.  seq_lens [B] real token counts, grid_sizes [B,3]."""
        return self._run(x, seq_lens, grid_sizes)

    def _run(self, x, seq_lens, grid_sizes, resid=None, gate=None):
        """resid: fp32 [B,L,C] updated in place as resid += out * gate (the o-projection's
        fused epilogue) instead of returning the bf16 projection."""
        self._check()
        B, L, C = x.shape
        N, D = self.num_heads, self.head_dim
        dev = x.device
        xb = _as_bf16_rows(x).contiguous()
        qkv = torch.empty(B, L, 3 * C, dtype=BF16, device=dev)
        att = torch.empty(B, L, C, dtype=BF16, device=dev)
        out = torch.empty(B, L, C, dtype=BF16, device=dev)
        for i, w in enumerate((self.q, self.k, self.v)):
            H.gemm(xb.view(B * L, C), w.weight.data, qkv.view(B * L, 3 * C)[:, i * C:(i + 1) * C], w.bias.data)
        for b in range(B):
            grid = tuple(int(v) for v in grid_sizes[b].tolist())
            n_tok = grid[0] * grid[1] * grid[2]
            rope = build_rope_table(D, max(grid), dev)
            for i, nw in ((0, self.norm_q), (1, self.norm_k)):
                v_ = qkv[b, :, i * C:(i + 1) * C]
                H.rmsnorm_rope(v_, nw.weight.data, v_, D, self.eps, rope=rope, rows_per_sample=L, n_rope=n_tok,
                               grid=grid)
            q4, k4, v4 = (qkv[b:b + 1, :, i * C:(i + 1) * C].unflatten(2, (N, D)) for i in range(3))
            H.flash_attn(q4, k4, v4, att[b:b + 1].unflatten(2, (N, D)), k_len=int(seq_lens[b]))
        if resid is not None:
            H.gemm(att.view(B * L, C), self.o.weight.data, resid.view(B * L, C), self.o.bias.data, H.EPI_RESID,
                   gate=gate)
            return resid
        H.gemm(att.view(B * L, C), self.o.weight.data, out.view(B * L, C), self.o.bias.data)
        return out


class WanCrossAttention(WanSelfAttention):
    """reference model.py:157-179"""

    def forward(self, x, context, context_lens):
        return self._run(x, context, context_lens)

    def _run(self, x, context, context_lens, resid=None):
        self._check()
        B, L, C = x.shape
        N, D = self.num_heads, self.head_dim
        Lk, dev = context.shape[1], x.device
        xb, cb = _as_bf16_rows(x).contiguous(), _as_bf16_rows(context).contiguous()
        q = torch.empty(B, L, C, dtype=BF16, device=dev)
        kv = torch.empty(B, Lk, 2 * C, dtype=BF16, device=dev)
        att, out = torch.empty_like(q), torch.empty_like(q)
        H.gemm(xb.view(B * L, C), self.q.weight.data, q.view(B * L, C), self.q.bias.data)
        H.rmsnorm_rope(q.view(B * L, C), self.norm_q.weight.data, q.view(B * L, C), D, self.eps)
        kv2 = kv.view(B * Lk, 2 * C)
        H.gemm(cb.view(B * Lk, C), self.k.weight.data, kv2[:, :C], self.k.bias.data)
        H.gemm(cb.view(B * Lk, C), self.v.weight.data, kv2[:, C:], self.v.bias.data)
        H.rmsnorm_rope(kv2[:, :C], self.norm_k.weight.data, kv2[:, :C], D, self.eps)
        for b in range(B):
            kl = Lk if context_lens is None else int(context_lens[b])
            H.flash_attn(q[b:b + 1].unflatten(2, (N, D)), kv[b:b + 1, :, :C].unflatten(2, (N, D)),
                         kv[b:b + 1, :, C:].unflatten(2, (N, D)), att[b:b + 1].unflatten(2, (N, D)), k_len=kl)
        if resid is not None:
            H.gemm(att.view(B * L, C), self.o.weight.data, resid.view(B * L, C), self.o.bias.data, H.EPI_RESID)
            return resid
        H.gemm(att.view(B * L, C), self.o.weight.data, out.view(B * L, C), self.o.bias.data)
        return out


class WanAttentionBlock(nn.Module):
    """reference model.py:182-258"""

    def __init__(self, dim, ffn_dim, num_heads, window_size=(-1, -1), qk_norm=True, cross_attn_norm=False,
                 eps=1e-6):
        super().__init__()
        self.dim, self.ffn_dim, self.num_heads = dim, ffn_dim, num_heads
        self.window_size, self.qk_norm, self.cross_attn_norm, self.eps = window_size, qk_norm, cross_attn_norm, eps
        self.norm1 = WanLayerNorm(dim, eps)
        self.self_attn = WanSelfAttention(dim, num_heads, window_size, qk_norm, eps)
        self.norm3 = WanLayerNorm(dim, eps, elementwise_affine=True) if cross_attn_norm else nn.Identity()
        self.cross_attn = WanCrossAttention(dim, num_heads, (-1, -1), qk_norm, eps)
        self.norm2 = WanLayerNorm(dim, eps)
        self.ffn = nn.Sequential(nn.Linear(dim, ffn_dim), nn.GELU(approximate="tanh"), nn.Linear(ffn_dim, dim))
        self.modulation = nn.Parameter(torch.randn(1, 6, dim) / dim ** 0.5)

    def forward(self, x, e, seq_lens, grid_sizes, freqs, context, context_lens):
        """x [B,L,C] (fp32 residual stream, or bf16 for the first block), e fp32 [B,1|L,6,C]
        (identical for every token of a sample) -> fp32 [B,L,C]."""
        assert e.dtype == F32
        if not self.cross_attn_norm:
            raise NotImplementedError("cross_attn_norm=False is not built (released configs use True)")
        B, L, C = x.shape
        dev = x.device
        round_norm = x.dtype == BF16
        xs = x.to(F32).contiguous().clone()
        h = torch.empty(B, L, C, dtype=BF16, device=dev)
        ffn = torch.empty(B * L, self.ffn_dim, dtype=BF16, device=dev)
        for b in range(B):
            mod = (self.modulation.data.float().reshape(6, C) + e[b, 0].reshape(6, C)).contiguous()
            xb = xs[b]
            H.layernorm_modulate(xb, h[b], self.eps, shift=mod[0], scale=mod[1], round_norm=round_norm)
            self.self_attn._run(h[b:b + 1], seq_lens[b:b + 1], grid_sizes[b:b + 1], resid=xs[b:b + 1],
                                gate=mod[2])
            H.layernorm_modulate(xb, h[b], self.eps, weight=self.norm3.weight.data, bias=self.norm3.bias.data)
            self.cross_attn._run(h[b:b + 1], context[b:b + 1],
                                 None if context_lens is None else context_lens[b:b + 1], resid=xs[b:b + 1])
            H.layernorm_modulate(xb, h[b], self.eps, shift=mod[3], scale=mod[4])
            fb = ffn[b * L:(b + 1) * L]
            H.gemm(h[b], self.ffn[0].weight.data, fb, self.ffn[0].bias.data, H.EPI_BF16_GELU)
            H.gemm(fb, self.ffn[2].weight.data, xb, self.ffn[2].bias.data, H.EPI_RESID, gate=mod[5])
        return xs


class Head(nn.Module):
    """reference model.py:261-290"""

    def __init__(self, dim, out_dim, patch_size, eps=1e-6):
        super().__init__()
        self.dim, self.out_dim, self.patch_size, self.eps = dim, out_dim, patch_size, eps
        self.norm = WanLayerNorm(dim, eps)
        self.head = nn.Linear(dim, math.prod(patch_size) * out_dim)
        self.modulation = nn.Parameter(torch.randn(1, 2, dim) / dim ** 0.5)

    def forward(self, x, e):
        """x [B, L, C], e fp32 [B, L, C] or [B, 1, C] with the same row for every token of a sample
        (the timestep embedding) -> fp32 [B, L, prod(patch_size) * out_dim] (reference model.py:278-290):
        fp32 LayerNorm, 2-way modulation and the projection in exact fp32 on the matrix cores."""
        assert e.dtype == F32
        B, L, C = x.shape
        pt, ph, pw = self.patch_size
        xf = x.to(F32).contiguous()
        e = e.to(x.device)
        hmod = torch.empty(2, C, dtype=F32, device=x.device)
        y = torch.empty(L, C, dtype=F32, device=x.device)
        tmp = torch.empty(self.out_dim, L * pt, ph, pw, dtype=F32, device=x.device)
        outs = []
        for b in range(B):
            eb = e[b]
            if eb.shape[0] > 1 and not bool((eb == eb[:1]).all()):
                raise NotImplementedError("Head: a different modulation row per token is not built "
                                          "(the model passes one timestep embedding per sample)")
            H.head_modulation(self.modulation.data.reshape(2, C), eb[0].contiguous(), hmod)
            H.layernorm_modulate(xf[b], y, self.eps, shift=hmod[0], scale=hmod[1])
            # the fused projection writes token l's (p, q, r, c) outputs to tmp[c, l*pt + p, q, r]
            H.head_unpatchify(y, self.head.weight.data, self.head.bias.data, tmp, (L, 1, 1), self.patch_size,
                              self.out_dim)
            outs.append(tmp.view(self.out_dim, L, pt, ph, pw).permute(1, 2, 3, 4, 0).reshape(L, -1).clone())
        return torch.stack(outs)


class WanModel(nn.Module):
    """reference model.py:293-549.  `forward` keeps the reference signature; the work is
    done by a DiTEngine (engine.py) built lazily from the current parameters."""

    def __init__(self, model_type="t2v", patch_size=(1, 2, 2), text_len=512, in_dim=16, dim=2048, ffn_dim=8192,
                 freq_dim=256, text_dim=4096, out_dim=16, num_heads=16, num_layers=32, window_size=(-1, -1),
                 qk_norm=True, cross_attn_norm=True, eps=1e-6, use_gradient_checkpointing=False, skip_init=False):
        super().__init__()
        assert model_type in ["t2v", "i2v", "ti2v"]
        self.model_type = model_type
        self.use_gradient_checkpointing = use_gradient_checkpointing
        self.patch_size, self.text_len, self.in_dim, self.dim = tuple(patch_size), text_len, in_dim, dim
        self.ffn_dim, self.freq_dim, self.text_dim, self.out_dim = ffn_dim, freq_dim, text_dim, out_dim
        self.num_heads, self.num_layers, self.window_size = num_heads, num_layers, window_size
        self.qk_norm, self.cross_attn_norm, self.eps = qk_norm, cross_attn_norm, eps
        self.patch_embedding = nn.Conv3d(in_dim, dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.text_embedding = nn.Sequential(nn.Linear(text_dim, dim), nn.GELU(approximate="tanh"),
                                            nn.Linear(dim, dim))
        self.time_embedding = nn.Sequential(nn.Linear(freq_dim, dim), nn.SiLU(), nn.Linear(dim, dim))
        self.time_projection = nn.Sequential(nn.SiLU(), nn.Linear(dim, dim * 6))
        self.blocks = nn.ModuleList([
            WanAttentionBlock(dim, ffn_dim, num_heads, window_size, qk_norm, cross_attn_norm, eps)
            for _ in range(num_layers)])
        self.head = Head(dim, out_dim, self.patch_size, eps)
        assert (dim % num_heads) == 0 and (dim // num_heads) % 2 == 0
        d = dim // num_heads
        self.freqs = torch.cat([rope_params(1024, d - 4 * (d // 6)), rope_params(1024, 2 * (d // 6)),
                                rope_params(1024, 2 * (d // 6))], dim=1)
        if not skip_init:
            self.init_weights()
        self._engine = None
        self._engine_key = None
        self._ctx_key = None
        self._ctx_cache = None
        # "bf16" = the reference's arithmetic; "fp8" = e4m3 operands for the six large linear layers of
        # every block (an extension: the reference has no such mode).  Takes effect at the next forward.
        self.linear_precision = "bf16"
        # self-attention operands: "auto" = int8 Q K^T in the fp8 mode, bf16 otherwise; "bf16" / "int8" force one
        self.attention_precision = "auto"

    # ---- diffusers-layout directories -----------------------------------------------
    # The reference's WanModel is a diffusers ModelMixin / ConfigMixin (model.py:293-303): `from_pretrained(dir,
    # subfolder=...)` reads <dir>/<subfolder>/config.json (the constructor arguments except `ignore_for_config`)
    # and the weights beside it, `save_pretrained(dir)` writes both.  The same files are read and written here
    # without the diffusers package, with loaders that execute nothing from the files.
    CONFIG_NAME = "config.json"
    WEIGHTS_NAME = "diffusion_pytorch_model"
    ignore_for_config = ["patch_size", "cross_attn_norm", "qk_norm", "text_dim", "window_size"]
    _config_keys = ("model_type", "text_len", "in_dim", "dim", "ffn_dim", "freq_dim", "out_dim", "num_heads",
                    "num_layers", "eps", "use_gradient_checkpointing")

    @property
    def config(self):
        return {k: getattr(self, k) for k in self._config_keys}

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder=None, torch_dtype=None, **overrides):
        """reference: ModelMixin.from_pretrained as called at modules/unified_model.py:42-43.  Reads config.json
        and `diffusion_pytorch_model.safetensors` (one file, or shards listed by
        `diffusion_pytorch_model.safetensors.index.json`) or `diffusion_pytorch_model.bin` (a torch.save archive,
        loaded with weights_only=True).  A local directory only: nothing is downloaded."""
        import json
        import os
        d = os.path.join(pretrained_model_name_or_path, subfolder) if subfolder else pretrained_model_name_or_path
        cfg_path = os.path.join(d, cls.CONFIG_NAME)
        if not os.path.isfile(cfg_path):
            raise OSError(f"{d} does not hold a {cls.CONFIG_NAME}: not a diffusers-layout WanModel directory "
                          "(hub names are not resolved here, pass a local path)")
        with open(cfg_path) as fh:
            raw = json.load(fh)
        accepted = set(cls._config_keys) | set(cls.ignore_for_config)
        kw = {k: v for k, v in raw.items() if k in accepted}
        kw.update(overrides)
        for k in ("patch_size", "window_size"):
            if k in kw:
                kw[k] = tuple(kw[k])
        model = cls(**kw, skip_init=True)
        state = {}
        st, idx, bn = (os.path.join(d, cls.WEIGHTS_NAME + e) for e in (".safetensors", ".safetensors.index.json", ".bin"))
        if os.path.isfile(st) or os.path.isfile(idx):
            from safetensors.torch import load_file
            files = [st]
            if not os.path.isfile(st):
                with open(idx) as fh:
                    files = sorted({os.path.join(d, f) for f in json.load(fh)["weight_map"].values()})
            for f in files:
                state.update(load_file(f, device="cpu"))
        elif os.path.isfile(bn):
            state = torch.load(bn, map_location="cpu", weights_only=True)
        else:
            raise OSError(f"no {cls.WEIGHTS_NAME}.safetensors (or .safetensors.index.json / .bin) in {d}")
        missing, unexpected = model.load_state_dict(state, strict=False)
        if missing or unexpected:
            raise RuntimeError(f"{d}: checkpoint does not match the configuration: {len(missing)} missing keys "
                               f"(first: {missing[:3]}), {len(unexpected)} unexpected (first: {unexpected[:3]})")
        if torch_dtype is not None:
            model = model.to(torch_dtype)
        return model.eval()

    def save_pretrained(self, save_directory, safe_serialization=True, **unused):
        """reference: ModelMixin.save_pretrained as called at modules/unified_model.py:280-282."""
        import json
        import os
        os.makedirs(save_directory, exist_ok=True)
        cfg = dict(self.config, _class_name=type(self).__name__)
        with open(os.path.join(save_directory, self.CONFIG_NAME), "w") as fh:
            json.dump(cfg, fh, indent=2, sort_keys=True)
        state = {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}
        if safe_serialization:
            from safetensors.torch import save_file
            save_file(state, os.path.join(save_directory, self.WEIGHTS_NAME + ".safetensors"), metadata={"format": "pt"})
        else:
            torch.save(state, os.path.join(save_directory, self.WEIGHTS_NAME + ".bin"))

    # ---- engine management --------------------------------------------------------
    def _weights_key(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def engine(self):
        key = (self._weights_key(), self.linear_precision, self.attention_precision)
        if self._engine is None or self._engine_key != key:
            self._engine = None                      # release the old copies before building new ones
            self._engine = DiTEngine(self, precision=self.linear_precision,
                                     attention=None if self.attention_precision == "auto" else self.attention_precision)
            self._engine_key = key
            self._ctx_key = None
        return self._engine

    def context_cache(self, contexts):
        """Embedded context + per-layer cross K/V, reused while the same context tensors
        (same storage, same version) and text_len are presented."""
        eng = self.engine()
        key = (self.text_len,) + tuple((c.data_ptr(), c._version, tuple(c.shape)) for c in contexts)
        if self._ctx_key != key:
            self._ctx_cache = eng.encode_context(contexts, self.text_len)
            self._ctx_key = key
            self._ctx_refs = list(contexts)   # keep the storages alive so data_ptr stays unique
        return self._ctx_cache

    def forward(self, x, t, context, seq_len, y=None):
        """x: list of [Cin,F,H,W]; t: [B]; context: list of [L,text_dim]; returns a list of
        fp32 [Cout,F,H,W] (reference model.py:417-524)."""
        return self._forward_impl(x, t, context, seq_len, y=y)

    def _forward_impl(self, x, t, context, seq_len, y=None, comm=None):
        """comm: sequence-parallel communicator (distributed/full_sequence_parallel.py)."""
        if self.model_type == "i2v":
            assert y is not None
        if y is not None:
            x = [torch.cat([u, v], dim=0) for u, v in zip(x, y)]
        if t.dim() == 2:
            # [B, L]: the reference broadcasts one timestep per sample over the tokens
            # (model.py:467-476); every token of a sample must carry the same value
            if not bool((t == t[:, :1]).all()):
                raise NotImplementedError("timesteps that differ between the tokens of a sample are not built")
            t = t[:, 0]
        elif t.dim() != 1:
            raise ValueError(f"t must have shape [B] or [B, L], got {tuple(t.shape)}")
        eng = self.engine()
        outs = []
        tt = t.to(eng.device, torch.float64).contiguous()
        for i, u in enumerate(x):
            u = u.to(eng.device)
            u = u.contiguous() if u.dtype in (BF16, F32) else u.float().contiguous()
            pt, ph, pw = self.patch_size
            grid = (u.shape[1] // pt, u.shape[2] // ph, u.shape[3] // pw)
            assert grid[0] * grid[1] * grid[2] <= seq_len
            eng.plan(grid, seq_len, 1, comm=comm)
            cache = self.context_cache([context[i]])
            out = torch.empty(1, self.out_dim, grid[0] * pt, grid[1] * ph, grid[2] * pw, dtype=F32,
                              device=eng.device)
            eng.forward(u, tt, cache, out, step=i)
            outs.append(out[0])
        return outs

    def unpatchify(self, x, grid_sizes):
        """List of [L, prod(patch_size) * Cout] head outputs -> list of [Cout, F, H, W] (reference
        model.py:526-549).  A layout change only; the forward does not come through here (the head
        kernel scatters its result directly, ovd_head_unpatchify)."""
        c = self.out_dim
        out = []
        for u, v in zip(x, grid_sizes.tolist() if torch.is_tensor(grid_sizes) else grid_sizes):
            f, h, w = (int(i) for i in v)
            pt, ph, pw = self.patch_size
            u = u[:f * h * w].view(f, h, w, pt, ph, pw, c).permute(6, 0, 3, 1, 4, 2, 5)
            out.append(u.reshape(c, f * pt, h * ph, w * pw))
        return out

    def init_weights(self):
        """reference model.py:551-573"""
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
        nn.init.xavier_uniform_(self.patch_embedding.weight.flatten(1))
        for m in self.text_embedding.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, std=.02)
        for m in self.time_embedding.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, std=.02)
        nn.init.zeros_(self.head.head.weight)

    def enable_gradient_checkpointing(self):
        self.use_gradient_checkpointing = True

    def disable_gradient_checkpointing(self):
        self.use_gradient_checkpointing = False
