"""umT5 encoder on the HIP kernels (omnivideo/modules/t5.py here) against the oracle in its bf16
flow and the reference's own outputs (tests/golden/t5_tiny.npz), plus the three T5 kernels."""
import os

import numpy as np
import pytest
import torch

from omnivideo.utils import synthetic as syn
from oracle import t5 as T

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32
CASES = {"short": (2, 40, [40, 23]), "long": (1, 300, [277])}


@pytest.fixture(scope="module")
def H():
    import omnivideo._hip as H
    H.lib()
    return H


def rb(x):
    return x.to(BF16).to(F32)


def build(cfg, sd):
    from omnivideo.modules.t5 import T5Encoder
    m = T5Encoder(vocab=cfg["vocab_size"], dim=cfg["dim"], dim_attn=cfg["dim_attn"], dim_ffn=cfg["dim_ffn"],
                  num_heads=cfg["num_heads"], num_layers=cfg["num_layers"], num_buckets=cfg["num_buckets"],
                  shared_pos=False)
    m.load_state_dict(sd, strict=True)
    return m.to(BF16).cuda().eval()


def test_gated_gelu_and_add_are_bit_exact(H):
    g = rb(syn.randn("t5g.gate", (70, 512), 1, 1.7))
    f = rb(syn.randn("t5g.fc1", (70, 512), 1))
    h = torch.cat([g, f], dim=1).cuda().to(BF16)
    out = torch.empty(70, 512, device="cuda", dtype=BF16)
    H.t5_gated_gelu(h, out)
    want = rb(f * T.t5_gelu(g, "bf16"))
    # tanhf on the device and on the host may differ in the last fp32 bit before the bf16 rounding
    assert float((out.float().cpu() != want).float().mean()) < 1e-3
    torch.testing.assert_close(out.float().cpu(), want, rtol=2 ** -7, atol=1e-6)
    x, y = rb(syn.randn("t5a.x", (33, 256), 1)), rb(syn.randn("t5a.y", (33, 256), 2))
    xg = x.cuda().to(BF16)
    H.add_bf16(xg, y.cuda().to(BF16))
    assert torch.equal(xg.float().cpu(), rb(x + y))


@pytest.mark.parametrize("L,lens", [(40, [40, 23]), (130, [130, 1]), (300, [277, 300])])
def test_attention_kernel_vs_oracle(H, L, lens):
    cfg = syn.t5_config("tiny")
    N = cfg["num_heads"]
    A = N * 64
    qkv = rb(syn.randn("t5att.qkv", (2, L, 3 * A), 1, 0.5))
    pos = rb(syn.randn("t5att.pos", (32, N), 1, 0.5))
    bucket = T.relative_position_bucket(L, L, 32)
    out = torch.empty(2, L, A, device="cuda", dtype=BF16)
    H.t5_attention(qkv.cuda().to(BF16), pos.cuda().to(BF16), bucket.to(torch.int32).cuda(),
                   torch.tensor(lens, dtype=torch.int32, device="cuda"), out, N)
    for b in range(2):
        q, k, v = (qkv[b, :, i * A:(i + 1) * A].view(L, N, 64) for i in range(3))
        bias = pos[bucket].permute(2, 0, 1).clone()
        bias[:, :, lens[b]:] = torch.finfo(BF16).min
        s = rb(rb(torch.einsum("inc,jnc->nij", q, k)) + bias)
        a = rb(torch.softmax(s, dim=-1))
        want = rb(torch.einsum("nij,jnc->inc", a, v)).reshape(L, A)
        got = out[b].float().cpu()
        # bf16 ties in the rounded scores / probabilities under a different fp32 summation order
        torch.testing.assert_close(got, want, rtol=2 ** -6, atol=2e-3)
        assert float((got != want).float().mean()) < 0.05


@pytest.mark.parametrize("name", list(CASES))
def test_encoder_vs_oracle_and_reference(golden_dir, name):
    cfg = syn.t5_config("tiny")
    sd = syn.make_t5_state_dict(cfg, 0)
    m = build(cfg, sd)
    ids, mask = syn.make_t5_ids(cfg, *CASES[name], seed=3)
    got = m(ids.cuda(), mask.cuda()).float().cpu()
    orc = T.encode(ids, mask, T.round_weights_bf16(sd), cfg, "bf16")
    ref = torch.from_numpy(dict(np.load(os.path.join(golden_dir, "t5_tiny.npz")))[f"{name}_bf16"])
    # only the rows of real tokens are kept by the pipeline (t5.py:512); padded query rows are
    # computed too and compared as well
    assert float((got - orc).norm() / orc.norm()) < 6e-3
    assert float((got - ref).norm() / ref.norm()) < 8e-3
    # against the fp32 reference the HIP path is as far away as the reference's own bf16 run
    ref32 = torch.from_numpy(dict(np.load(os.path.join(golden_dir, "t5_tiny.npz")))[f"{name}_fp32"])
    err = lambda a: float((a - ref32).norm() / ref32.norm())
    assert err(got) < 1.2 * err(ref) + 1e-3, (err(got), err(ref))


def test_encoder_model_wrapper_and_mask_semantics():
    from omnivideo.modules.t5 import T5EncoderModel
    cfg = syn.t5_config("tiny")
    enc = T5EncoderModel(text_len=32, device="cuda", vocab_size=cfg["vocab_size"], dim=cfg["dim"],
                         dim_attn=cfg["dim_attn"], dim_ffn=cfg["dim_ffn"], num_heads=cfg["num_heads"],
                         encoder_layers=cfg["num_layers"], num_buckets=cfg["num_buckets"])
    enc.model.load_state_dict(syn.make_t5_state_dict(cfg, 0), strict=True)
    enc.model.to(BF16)
    ids, mask = syn.make_t5_ids(cfg, 2, 32, [32, 11], seed=5)
    outs = enc.encode_ids(ids.cuda(), mask.cuda())
    assert [tuple(o.shape) for o in outs] == [(32, cfg["dim"]), (11, cfg["dim"])]
    # the valid rows do not depend on the padding behind them
    alone = enc.encode_ids(ids[1:, :11].cuda(), mask[1:, :11].cuda())[0]
    assert torch.equal(alone, outs[1])
    with pytest.raises(RuntimeError, match="tokenizer"):
        enc(["a prompt"], "cuda")
    with pytest.raises(NotImplementedError, match="prefix"):
        enc.model(ids.cuda(), torch.tensor([[1, 0] * 16, [1] * 32]).cuda())


def test_generate_encodes_prompts_with_the_attached_text_encoder():
    """generate() without precomputed contexts calls self.text_encoder for the prompt and the negative
    prompt (reference x2x_gen_unified.py:336-352); the result equals passing the same embeddings in."""
    from omnivideo.configs import Config, t2v_1_3B
    from omnivideo.modules.t5 import T5EncoderModel
    from omnivideo.x2x_gen_unified_1_3B import OmniVideoX2XUnified1_3B
    dcfg = syn.dit_config("tiny")
    tcfg = syn.t5_config("tiny")
    assert tcfg["dim"] == dcfg["text_dim"]

    def tokenizer(texts, return_mask=True, add_special_tokens=True):
        """stand-in for HuggingfaceTokenizer: bytes as ids, end token 1, padded to 32 with a prefix mask"""
        ids, mask = torch.zeros(len(texts), 32, dtype=torch.long), torch.zeros(len(texts), 32, dtype=torch.long)
        for i, t in enumerate(texts):
            toks = [2 + (b % 250) for b in t.encode()][:31] + [1]
            ids[i, :len(toks)] = torch.tensor(toks)
            mask[i, :len(toks)] = 1
        return ids, mask

    enc = T5EncoderModel(text_len=32, device="cuda", tokenizer=tokenizer, vocab_size=tcfg["vocab_size"],
                         dim=tcfg["dim"], dim_attn=tcfg["dim_attn"], dim_ffn=tcfg["dim_ffn"],
                         num_heads=tcfg["num_heads"], encoder_layers=tcfg["num_layers"],
                         num_buckets=tcfg["num_buckets"])
    enc.model.load_state_dict(syn.make_t5_state_dict(tcfg, 0), strict=True)
    enc.model.to(BF16)
    pipe = OmniVideoX2XUnified1_3B(Config(t2v_1_3B), None, vlm_in_dim=dcfg["vlm_in_dim"],
                                   use_visual_context_adapter=True,
                                   visual_context_adapter_patch_size=dcfg["visual_patch_size"],
                                   max_context_len=dcfg["text_len"], wan_config=dict(dcfg))
    pipe.model.load_state_dict(syn.make_unified_state_dict(dcfg, 0), strict=True)
    pipe.model.to(BF16)
    pipe.text_encoder = enc
    noise = syn.randn("gen.noise", (16, 3, 8, 12), 11).to(BF16)
    kw = dict(size=(96, 64), frame_num=9, shift=5.0, sampling_steps=2, guide_scale=3.0, seed=1, noise=noise,
              return_latents=True)
    prompt, neg = "a person playing guitar", "blurry, low quality"
    a = pipe.generate(prompt, n_prompt=neg, **kw)
    c, u = enc([prompt], "cuda")[0], enc([neg], "cuda")[0]
    assert c.shape[0] == len(prompt) + 1 and u.shape[0] == len(neg) + 1
    b = pipe.generate("", precomputed_context=c, precomputed_context_null=u, **kw)
    assert torch.isfinite(a.float()).all() and torch.equal(a, b)
