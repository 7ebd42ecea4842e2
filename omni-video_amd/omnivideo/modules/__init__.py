"""Operator surface of the reference's `omnivideo.modules` for the sampling path
(reference: omnivideo/modules/__init__.py:9-20).  The T5 / tokenizer / Wan2.2-VAE names
of the reference are outside the accelerated path and are not provided."""
from .attention import flash_attention
from .model import WanModel
from .visual_context_adapter import VisualContextAdapter

__all__ = ["WanModel", "flash_attention", "VisualContextAdapter"]
