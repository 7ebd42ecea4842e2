"""The diffusers-layout directory route of the model constructors (reference modules/unified_model.py:42-43,
x2x_gen_unified.py:140-169): `WanModel.save_pretrained` / `from_pretrained` round trip on CPU - config.json
carries the constructor arguments the reference registers (model.py:298-301 ignores patch_size, qk_norm,
cross_attn_norm, text_dim, window_size), the weights come back bit for bit from a single safetensors file,
from shards behind an index file, and from a .bin archive read with weights_only=True."""
import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "omni-video_amd"))

from omnivideo.modules.model import WanModel                                   # noqa: E402
from omnivideo.modules.unified_model import UnifiedWanWithMixedConditionModel    # noqa: E402
from omnivideo.utils import synthetic as syn                                   # noqa: E402


def tiny():
    cfg = syn.dit_config("tiny")
    keys = ("patch_size", "dim", "ffn_dim", "freq_dim", "num_heads", "num_layers", "text_len", "in_dim", "text_dim",
            "out_dim", "eps")
    torch.manual_seed(3)
    return WanModel(**{k: cfg[k] for k in keys if k in cfg}, skip_init=True), cfg


def same_weights(a, b):
    sa, sb = a.state_dict(), b.state_dict()
    assert sorted(sa) == sorted(sb)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


@pytest.mark.parametrize("safe", [True, False])
def test_save_and_from_pretrained_round_trip(tmp_path, safe):
    m, cfg = tiny()
    m.save_pretrained(str(tmp_path / "wan_model"), safe_serialization=safe)
    raw = json.load(open(tmp_path / "wan_model" / "config.json"))
    assert raw["dim"] == cfg["dim"] and raw["num_layers"] == cfg["num_layers"] and raw["_class_name"] == "WanModel"
    assert not set(raw) & set(WanModel.ignore_for_config)
    back = WanModel.from_pretrained(str(tmp_path), subfolder="wan_model", text_dim=cfg["text_dim"])
    assert back.text_len == m.text_len and back.num_heads == m.num_heads and not back.training
    same_weights(m, back)


def test_sharded_safetensors_and_unified_wrapper(tmp_path):
    from safetensors.torch import save_file
    m, cfg = tiny()
    d = tmp_path / "high_noise_model"
    m.save_pretrained(str(d))
    os.remove(d / "diffusion_pytorch_model.safetensors")
    items = sorted(m.state_dict().items())
    half = len(items) // 2
    wmap = {}
    for i, part in enumerate((items[:half], items[half:])):
        name = f"diffusion_pytorch_model-0000{i + 1}-of-00002.safetensors"
        save_file({k: v.contiguous() for k, v in part}, str(d / name))
        wmap.update({k: name for k, _ in part})
    json.dump({"metadata": {}, "weight_map": wmap}, open(d / "diffusion_pytorch_model.safetensors.index.json", "w"))
    if cfg["text_dim"] != 4096:              # text_dim is not part of config.json (reference ignore_for_config)
        raw = json.load(open(d / "config.json"))
        raw["text_dim"] = cfg["text_dim"]
        json.dump(raw, open(d / "config.json", "w"))
    u = UnifiedWanWithMixedConditionModel.from_pretrained(str(tmp_path), wan_subfolder="high_noise_model",
                                                          vlm_in_dim=cfg["vlm_in_dim"], max_context_len=48)
    same_weights(m, u.wan_model)
    assert u.wan_model.text_len == 48 and u.max_context_len == 48


def test_missing_directory_and_mismatched_checkpoint_raise(tmp_path):
    with pytest.raises(OSError, match="config.json"):
        WanModel.from_pretrained(str(tmp_path / "nowhere"))
    m, cfg = tiny()
    m.save_pretrained(str(tmp_path))
    raw = json.load(open(tmp_path / "config.json"))
    raw["num_layers"] = cfg["num_layers"] + 1
    json.dump(raw, open(tmp_path / "config.json", "w"))
    with pytest.raises(RuntimeError, match="does not match the configuration"):
        WanModel.from_pretrained(str(tmp_path), text_dim=cfg["text_dim"])


@pytest.mark.gpu
def test_pipeline_without_wan_config_reads_the_directory_and_matches_the_config_route(tmp_path):
    """`OmniVideoX2XUnified1_3B(config, checkpoint_dir, wan_config=None)` (reference x2x_gen_unified_1_3B.py:
    121-136) reads the DiT from `checkpoint_dir/<config.checkpoint>`; with the condition builder's weights
    loaded on top, `generate` gives the same latents as the wan_config + load_state_dict route, bit for bit."""
    from omnivideo.configs import Config, t2v_1_3B
    from omnivideo.x2x_gen_unified_1_3B import OmniVideoX2XUnified1_3B
    BF16 = torch.bfloat16
    cfg = syn.dit_config("tiny")
    sd = syn.make_unified_state_dict(cfg, 0)
    conf = Config(t2v_1_3B)
    kw = dict(vlm_in_dim=cfg["vlm_in_dim"], use_visual_context_adapter=True,
              visual_context_adapter_patch_size=cfg["visual_patch_size"], max_context_len=cfg["text_len"])
    a = OmniVideoX2XUnified1_3B(conf, None, wan_config=dict(cfg), **kw)
    a.model.load_state_dict(sd, strict=True)
    a.model.to(BF16)
    sub = conf.get("checkpoint", None)
    a.model.wan_model.float().save_pretrained(str(tmp_path / sub) if sub else str(tmp_path))
    a.model.to(BF16)
    d = tmp_path / sub if sub else tmp_path
    raw = json.load(open(d / "config.json"))
    raw.update(text_dim=cfg["text_dim"], patch_size=list(cfg["patch_size"]))
    json.dump(raw, open(d / "config.json", "w"))
    b = OmniVideoX2XUnified1_3B(conf, str(tmp_path), wan_config=None, **kw)
    rest = {k: v for k, v in sd.items() if not k.startswith("wan_model.")}
    missing, unexpected = b.model.load_state_dict(rest, strict=False)
    assert not unexpected and all(k.startswith("wan_model.") for k in missing)
    b.model.to(BF16)
    sp = {k: v.cuda().to(BF16) for k, v in syn.make_special_tokens(cfg, 0).items()}
    a.special_tokens = b.special_tokens = sp
    text, null = syn.randn("in.text", (10, cfg["text_dim"]), 1).to(BF16), syn.randn("in.text_null", (6, cfg["text_dim"]), 1).to(BF16)
    noise = syn.randn("gen.noise", (16, 3, 8, 12), 11).to(BF16)
    run = lambda p: p.generate("", precomputed_context=text, precomputed_context_null=null, size=(96, 64), frame_num=9,
                               shift=5.0, sampling_steps=3, guide_scale=3.0, seed=1, noise=noise, return_latents=True)
    assert torch.equal(run(a), run(b))
