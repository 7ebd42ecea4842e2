"""TEST INFRASTRUCTURE ONLY - loads the reference's hot-path modules by file path.

Used by ``oracle/gen_golden.py`` (golden-vector generation) and by the CPU tests that
pin the oracle restatement to the real reference.  It only works where
``/root/reference`` exists (the build container); nothing on the GPU box imports it.

The reference package cannot be imported as a package here: ``omnivideo/__init__.py``
pulls in ``easydict``, ``diffusers``, ``ftfy`` and a T5 module whose class body calls
``torch.cuda.current_device()``.  The hot-path files themselves only need two
``diffusers`` *mixins* that carry no arithmetic (``ModelMixin`` is ``nn.Module`` plus
checkpoint I/O; ``ConfigMixin``/``register_to_config`` record constructor kwargs in
``self.config``).  Minimal equivalents of those mixins are registered in
``sys.modules`` and the files are loaded with ``importlib`` under a synthetic package
name so that their relative imports resolve among themselves.

``flash_attention`` in the reference asserts a CUDA device and calls the third-party
``flash_attn`` wheel (``omnivideo/modules/attention.py:54,96-130``); on CPU it is
replaced by exact softmax attention with the same ``k_lens`` masking, which is the
math of the reference's own fallback (``attention.py:167-182``) kept in the input
dtype.
"""
import dataclasses
import enum
import functools
import importlib.util
import inspect
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REF_ROOT = os.environ.get("OMNI_REFERENCE_ROOT", "/root/reference")
_PKG = "_omni_ref"


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "omnivideo", "modules"))


class _Cfg(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def _register_to_config(init):
    @functools.wraps(init)
    def wrapped(self, *args, **kwargs):
        sig = inspect.signature(init)
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = _Cfg({k: v for k, v in bound.arguments.items() if k != "self"})
        object.__setattr__(self, "_cfg_store", cfg)
        init(self, *args, **kwargs)
    return wrapped


class _ConfigMixin:
    @property
    def config(self):
        return self._cfg_store

    def register_to_config(self, **kw):
        self._cfg_store.update(kw)


class _ModelMixin(nn.Module):
    pass


class _SchedulerMixin:
    pass


@dataclasses.dataclass
class _SchedulerOutput:
    prev_sample: torch.Tensor


class _Karras(enum.Enum):
    UniPCMultistepScheduler = 1


def _install_stubs():
    if "diffusers" in sys.modules and getattr(sys.modules["diffusers"], "_omni_stub", False):
        return
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m._omni_stub = True
        sys.modules[name] = m
        return m
    mod("diffusers")
    mod("diffusers.configuration_utils", ConfigMixin=_ConfigMixin,
        register_to_config=_register_to_config)
    mod("diffusers.models")
    mod("diffusers.models.modeling_utils", ModelMixin=_ModelMixin)
    mod("diffusers.schedulers")
    mod("diffusers.schedulers.scheduling_utils", KarrasDiffusionSchedulers=_Karras,
        SchedulerMixin=_SchedulerMixin, SchedulerOutput=_SchedulerOutput)
    mod("diffusers.utils", deprecate=lambda *a, **k: None,
        is_scipy_available=lambda: False)
    mod("diffusers.utils.torch_utils", randn_tensor=lambda shape, generator=None, device=None, dtype=None:
        torch.randn(shape, generator=generator, device=device, dtype=dtype))


def exact_attention(q, k, v, q_lens=None, k_lens=None, dropout_p=0., softmax_scale=None,
                    q_scale=None, causal=False, window_size=(-1, -1), deterministic=False,
                    dtype=torch.bfloat16, version=None):
    """CPU stand-in for the flash-attn call: same signature as the reference's
    ``flash_attention`` (attention.py:24-38), same [B, L, N, D] layout, masks keys
    at positions >= k_lens[b], returns q's dtype."""
    assert not causal and q_lens is None and q_scale is None
    b, lq, lk = q.size(0), q.size(1), k.size(1)
    mask = None
    if k_lens is not None:
        mask = (torch.arange(lk)[None, :] < k_lens.view(-1, 1).to(torch.long))
        mask = mask[:, None, None, :].expand(b, 1, lq, lk)
    out = F.scaled_dot_product_attention(
        q.transpose(1, 2), k.transpose(1, 2).to(q.dtype), v.transpose(1, 2).to(q.dtype),
        attn_mask=mask, scale=softmax_scale)
    return out.transpose(1, 2).contiguous().type(q.dtype)


def _load(subdir, fname, modname):
    full = f"{_PKG}.{subdir}.{modname}"
    if full in sys.modules:
        return sys.modules[full]
    path = os.path.join(REF_ROOT, "omnivideo", subdir, fname)
    spec = importlib.util.spec_from_file_location(full, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[full] = m
    spec.loader.exec_module(m)
    return m


@functools.lru_cache(maxsize=None)
def load():
    """Returns a namespace with the reference's hot-path modules:
    .model .attention .unified_model .visual_context_adapter .vae2_1 .unipc .dpm .t5"""
    if not available():
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    _install_stubs()
    for name in (_PKG, f"{_PKG}.modules", f"{_PKG}.utils"):
        if name not in sys.modules:
            p = types.ModuleType(name)
            sub = name.split(".")[-1] if "." in name else ""
            p.__path__ = [os.path.join(REF_ROOT, "omnivideo", sub)]
            sys.modules[name] = p
    ns = types.SimpleNamespace()
    ns.attention = _load("modules", "attention.py", "attention")
    ns.model = _load("modules", "model.py", "model")
    ns.model.flash_attention = exact_attention
    ns.visual_context_adapter = _load("modules", "visual_context_adapter.py",
                                      "visual_context_adapter")
    ns.unified_model = _load("modules", "unified_model.py", "unified_model")
    ns.vae2_1 = _load("modules", "vae2_1.py", "vae2_1")
    ns.unipc = _load("utils", "fm_solvers_unipc.py", "fm_solvers_unipc")
    ns.dpm = _load("utils", "fm_solvers.py", "fm_solvers")
    # t5.py evaluates torch.cuda.current_device() as a default argument at class-definition time
    # (t5.py:477), which raises on a CPU-only box; answer 0 while the module is being loaded.  Its
    # `from .tokenizers import ...` needs ftfy: a stub module stands in (the tokenizer is not used).
    import torch.cuda as _tc
    _saved = _tc.current_device
    _tc.current_device = lambda: 0
    tok = types.ModuleType(f"{_PKG}.modules.tokenizers")
    tok.HuggingfaceTokenizer = None
    sys.modules.setdefault(f"{_PKG}.modules.tokenizers", tok)
    try:
        ns.t5 = _load("modules", "t5.py", "t5")
    finally:
        _tc.current_device = _saved
    return ns
