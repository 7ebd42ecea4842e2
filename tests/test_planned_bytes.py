"""`engine.planned_bytes` - what bench.py compares with the device's free memory BEFORE allocating anything -
against what the modules really hold: the DiT's parameter bytes exactly (CPU), and on a GPU the engine copies,
the context cache and the workspaces that `DiTEngine` allocates (allocator-measured, within 2 %)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "omni-video_amd"))

from omnivideo.modules.engine import planned_bytes          # noqa: E402
from omnivideo.modules.model import WanModel                # noqa: E402
from omnivideo.utils import synthetic as syn                # noqa: E402

KEYS = ("patch_size", "dim", "ffn_dim", "freq_dim", "num_heads", "num_layers", "text_len", "in_dim", "text_dim", "out_dim", "eps")


@pytest.mark.parametrize("name,over", [("tiny", {}), ("1.3B", dict(num_layers=2)), ("A14B", dict(num_layers=1))])
def test_parameter_bytes_are_exact(name, over):
    cfg = syn.dit_config(name, **over)
    with torch.device("meta"):
        m = WanModel(**{k: cfg[k] for k in KEYS if k in cfg}, skip_init=True)
    want = 2 * sum(p.numel() for p in m.parameters())
    got = planned_bytes(cfg, (2, 2, 2), 8, 2, cfg["text_len"])["parameters"]
    assert got == want, (got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["bf16", "fp8"])
def test_engine_allocations_match_the_plan(precision):
    from omnivideo import _hip as H
    from omnivideo.modules.engine import DiTEngine
    cfg = syn.dit_config("1.3B", num_layers=2, text_len=512)
    grid, seq_len, nb = (5, 16, 16), 1280, 2
    with torch.device("cuda"):
        m = WanModel(**{k: cfg[k] for k in KEYS if k in cfg}, skip_init=True).to(torch.bfloat16)
    plan = planned_bytes(cfg, grid, seq_len, nb, cfg["text_len"], precision=precision,
                         attn_workspace=H.flash_attn_workspace_bytes(nb, seq_len, cfg["num_heads"], seq_len))
    torch.cuda.synchronize()
    a0 = torch.cuda.memory_allocated()
    eng = DiTEngine(m, precision=precision)
    a1 = torch.cuda.memory_allocated()
    eng.plan(grid, seq_len, nb)
    a2 = torch.cuda.memory_allocated()
    cache = eng.encode_context([torch.randn(20, cfg["text_dim"], device="cuda").to(torch.bfloat16)] * nb, cfg["text_len"])
    torch.cuda.synchronize()
    a3 = torch.cuda.memory_allocated()
    assert abs((a1 - a0) - plan["engine_copies"]) <= 0.02 * plan["engine_copies"] + (1 << 20), (a1 - a0, plan["engine_copies"])
    assert abs((a2 - a1) - plan["workspaces"]) <= 0.02 * plan["workspaces"] + (1 << 20), (a2 - a1, plan["workspaces"])
    # the cache itself; the plan also counts the transient embedding buffers of encode_context
    assert cache.kv.numel() * 2 <= plan["context_cache"] and (a3 - a2) <= plan["context_cache"] + (1 << 20)
