"""umT5-XXL text encoder on the HIP kernels, with the reference's names (reference:
omnivideo/modules/t5.py): `T5Encoder` (shared_pos=False, the only form the pipeline builds,
t5.py:455-468), `umt5_xxl(encoder_only=True)`, `T5EncoderModel`.

Parameter names equal the reference's (`token_embedding`, `blocks.{i}.{norm1,attn.{q,k,v,o},norm2,
ffn.{gate.0,fc1,fc2},pos_embedding.embedding}`, `norm`) so the released `models_t5_umt5-xxl-enc-
bf16.pth` loads with `load_state_dict`.  The forward runs in the reference's GPU dtype flow:
module and activations in bf16, every tensor op rounding its result (t5.py:440,60-65,111-113).

Per layer: RMSNorm kernel -> fused q|k|v GEMM -> bias/softmax attention kernel -> o GEMM ->
bf16 residual add -> RMSNorm -> fused gate|fc1 GEMM -> gated-GELU kernel -> fc2 GEMM -> add.
The relative-position bucket table depends only on the length and is built once per length on
the host with the reference's formula (t5.py:232-263).
"""
import logging
import math

import torch
import torch.nn as nn

from .. import _hip as H

BF16 = torch.bfloat16

__all__ = ["T5Encoder", "T5EncoderModel", "umt5_xxl"]


class T5LayerNorm(nn.Module):
    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.dim, self.eps = dim, eps
        self.weight = nn.Parameter(torch.ones(dim))


class T5Attention(nn.Module):
    def __init__(self, dim, dim_attn, num_heads, dropout=0.1):
        assert dim_attn % num_heads == 0
        super().__init__()
        self.dim, self.dim_attn, self.num_heads, self.head_dim = dim, dim_attn, num_heads, dim_attn // num_heads
        self.q = nn.Linear(dim, dim_attn, bias=False)
        self.k = nn.Linear(dim, dim_attn, bias=False)
        self.v = nn.Linear(dim, dim_attn, bias=False)
        self.o = nn.Linear(dim_attn, dim, bias=False)


class _Gate(nn.Sequential):
    """`ffn.gate.0` holds the weight, as in the reference's nn.Sequential(Linear, GELU)."""
    def __init__(self, dim, dim_ffn):
        super().__init__(nn.Linear(dim, dim_ffn, bias=False))


class T5FeedForward(nn.Module):
    def __init__(self, dim, dim_ffn, dropout=0.1):
        super().__init__()
        self.dim, self.dim_ffn = dim, dim_ffn
        self.gate = _Gate(dim, dim_ffn)
        self.fc1 = nn.Linear(dim, dim_ffn, bias=False)
        self.fc2 = nn.Linear(dim_ffn, dim, bias=False)


class T5RelativeEmbedding(nn.Module):
    def __init__(self, num_buckets, num_heads, bidirectional, max_dist=128):
        super().__init__()
        self.num_buckets, self.num_heads, self.bidirectional, self.max_dist = num_buckets, num_heads, bidirectional, max_dist
        self.embedding = nn.Embedding(num_buckets, num_heads)

    def bucket_table(self, lq, lk):
        """reference :232-263 (bidirectional), as an int32 [lq, lk] table."""
        rel_pos = torch.arange(lk).unsqueeze(0) - torch.arange(lq).unsqueeze(1)
        if not self.bidirectional:
            raise NotImplementedError("only the encoder's bidirectional buckets are built")
        nb = self.num_buckets // 2
        rel_buckets = (rel_pos > 0).long() * nb
        rel_pos = torch.abs(rel_pos)
        max_exact = nb // 2
        large = max_exact + (torch.log(rel_pos.float() / max_exact) / math.log(self.max_dist / max_exact) *
                             (nb - max_exact)).long()
        large = torch.min(large, torch.full_like(large, nb - 1))
        return (rel_buckets + torch.where(rel_pos < max_exact, rel_pos, large)).to(torch.int32)


class T5SelfAttention(nn.Module):
    def __init__(self, dim, dim_attn, dim_ffn, num_heads, num_buckets, shared_pos=True, dropout=0.1):
        super().__init__()
        if shared_pos:
            raise NotImplementedError("umT5 uses per-block position embeddings (shared_pos=False)")
        self.norm1 = T5LayerNorm(dim)
        self.attn = T5Attention(dim, dim_attn, num_heads, dropout)
        self.norm2 = T5LayerNorm(dim)
        self.ffn = T5FeedForward(dim, dim_ffn, dropout)
        self.pos_embedding = T5RelativeEmbedding(num_buckets, num_heads, bidirectional=True)


class T5Encoder(nn.Module):
    def __init__(self, vocab, dim, dim_attn, dim_ffn, num_heads, num_layers, num_buckets, shared_pos=True,
                 dropout=0.1):
        super().__init__()
        self.dim, self.dim_attn, self.dim_ffn = dim, dim_attn, dim_ffn
        self.num_heads, self.num_layers, self.num_buckets, self.shared_pos = num_heads, num_layers, num_buckets, shared_pos
        if dim_attn // num_heads != 64:
            raise NotImplementedError("the attention kernel is built for head_dim 64 (umT5)")
        self.token_embedding = vocab if isinstance(vocab, nn.Embedding) else nn.Embedding(vocab, dim)
        self.blocks = nn.ModuleList([T5SelfAttention(dim, dim_attn, dim_ffn, num_heads, num_buckets, shared_pos, dropout)
                                     for _ in range(num_layers)])
        self.norm = T5LayerNorm(dim)
        self._fused = None
        self._buckets = {}

    # fused projection weights are views built once from the parameters
    def _prepare(self):
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._fused is not None and self._fused[0] == key:
            return self._fused[1]
        for p in self.parameters():
            if p.dtype != BF16 or not p.is_cuda:
                raise RuntimeError("T5Encoder runs on the HIP device with bf16 parameters: call .to(torch.bfloat16).cuda()")
        layers = []
        for blk in self.blocks:
            layers.append(dict(
                n1=blk.norm1.weight.detach(), n2=blk.norm2.weight.detach(),
                w_qkv=torch.cat([blk.attn.q.weight, blk.attn.k.weight, blk.attn.v.weight]).detach().contiguous(),
                w_o=blk.attn.o.weight.detach().contiguous(),
                w_gf=torch.cat([blk.ffn.gate[0].weight, blk.ffn.fc1.weight]).detach().contiguous(),
                w_2=blk.ffn.fc2.weight.detach().contiguous(),
                pos=blk.pos_embedding.embedding.weight.detach().contiguous()))
        self._fused = (key, layers)
        return layers

    def _bucket(self, L, device):
        if L not in self._buckets:
            self._buckets[L] = self.blocks[0].pos_embedding.bucket_table(L, L).to(device).contiguous()
        return self._buckets[L]

    @torch.no_grad()
    def forward(self, ids, mask=None):
        """ids [B, L] int64, mask [B, L] (1 = token; a prefix mask, as the tokenizer returns) ->
        [B, L, dim] bf16 (reference :302-311)."""
        layers = self._prepare()
        dev = self.token_embedding.weight.device
        ids = ids.to(dev)
        B, L = ids.shape
        k_lens = None
        if mask is not None:
            mask = mask.to(dev)
            k_lens = mask.gt(0).sum(dim=1).to(torch.int32)
            if not torch.equal(mask.gt(0), torch.arange(L, device=dev)[None, :] < k_lens[:, None]):
                raise NotImplementedError("only prefix masks (real tokens first, padding after) are built")
            k_lens = k_lens.contiguous()
        bucket = self._bucket(L, dev)
        M, C, A, F = B * L, self.dim, self.dim_attn, self.dim_ffn
        x = self.token_embedding.weight.detach()[ids].reshape(M, C).contiguous()       # gather: data movement
        h = torch.empty(M, C, dtype=BF16, device=dev)
        qkv = torch.empty(B, L, 3 * A, dtype=BF16, device=dev)
        att = torch.empty(B, L, A, dtype=BF16, device=dev)
        y = torch.empty(M, C, dtype=BF16, device=dev)
        gf = torch.empty(M, 2 * F, dtype=BF16, device=dev)
        g = torch.empty(M, F, dtype=BF16, device=dev)
        for lw in layers:
            H.rmsnorm_rope(x, lw["n1"], h, 64, self.blocks[0].norm1.eps)
            H.gemm(h, lw["w_qkv"], qkv.view(M, 3 * A), None, H.EPI_BF16)
            H.t5_attention(qkv, lw["pos"], bucket, k_lens, att, self.num_heads)
            H.gemm(att.view(M, A), lw["w_o"], y, None, H.EPI_BF16)
            H.add_bf16(x, y)
            H.rmsnorm_rope(x, lw["n2"], h, 64, self.blocks[0].norm2.eps)
            H.gemm(h, lw["w_gf"], gf, None, H.EPI_BF16)
            H.t5_gated_gelu(gf, g)
            H.gemm(g, lw["w_2"], y, None, H.EPI_BF16)
            H.add_bf16(x, y)
        H.rmsnorm_rope(x, self.norm.weight.detach(), h, 64, self.norm.eps)
        return h.view(B, L, C)


def umt5_xxl(encoder_only=True, return_tokenizer=False, dtype=BF16, device="cuda", **kwargs):
    """reference :414-468 for the encoder."""
    if not encoder_only or return_tokenizer:
        raise NotImplementedError("only the encoder is built; construct the tokenizer separately")
    cfg = dict(vocab_size=256384, dim=4096, dim_attn=4096, dim_ffn=10240, num_heads=64, encoder_layers=24,
               decoder_layers=24, num_buckets=32, shared_pos=False, dropout=0.1)
    cfg.update(**kwargs)
    with torch.device(device):
        model = T5Encoder(vocab=cfg["vocab_size"], dim=cfg["dim"], dim_attn=cfg["dim_attn"], dim_ffn=cfg["dim_ffn"],
                          num_heads=cfg["num_heads"], num_layers=cfg["encoder_layers"],
                          num_buckets=cfg["num_buckets"], shared_pos=cfg["shared_pos"], dropout=cfg["dropout"])
    return model.to(dtype=dtype, device=device)


class T5EncoderModel:
    """reference :471-512.  The tokenizer is built from `tokenizer_path` (a local directory with
    the umT5 tokenizer files) or passed in as any callable (texts, return_mask=True,
    add_special_tokens=True) -> (ids, mask).  Without one, call `encode_ids`."""

    def __init__(self, text_len, dtype=BF16, device=None, checkpoint_path=None, tokenizer_path=None, shard_fn=None,
                 tokenizer=None, **t5_kwargs):
        if shard_fn is not None:
            raise NotImplementedError("FSDP sharding is not part of this build")
        self.text_len, self.dtype = text_len, dtype
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.checkpoint_path, self.tokenizer_path = checkpoint_path, tokenizer_path
        self.model = umt5_xxl(encoder_only=True, dtype=dtype, device=self.device, **t5_kwargs).eval().requires_grad_(False)
        if checkpoint_path is not None:
            logging.info(f"loading {checkpoint_path}")
            self.model.load_state_dict(torch.load(checkpoint_path, map_location="cpu", weights_only=True))
        if tokenizer is None and tokenizer_path is not None:
            from .tokenizers import HuggingfaceTokenizer
            tokenizer = HuggingfaceTokenizer(name=tokenizer_path, seq_len=text_len, clean="whitespace")
        self.tokenizer = tokenizer

    def encode_ids(self, ids, mask):
        seq_lens = mask.gt(0).sum(dim=1).long()
        context = self.model(ids, mask)
        return [u[:v] for u, v in zip(context, seq_lens)]

    def __call__(self, texts, device):
        if self.tokenizer is None:
            raise RuntimeError("no tokenizer attached: pass tokenizer=... or use encode_ids(ids, mask)")
        ids, mask = self.tokenizer(texts, return_mask=True, add_special_tokens=True)
        return self.encode_ids(ids.to(device), mask.to(device))
