"""Operator surface of the reference's `omnivideo.modules` for the sampling path and the
text encoder in front of it (reference: omnivideo/modules/__init__.py:9-20).  The T5 decoder and
the Wan2.2 VAE of the reference are outside this build and are not provided."""
from .attention import flash_attention
from .model import WanModel
from .t5 import T5Encoder, T5EncoderModel
from .tokenizers import HuggingfaceTokenizer
from .vae2_1 import Wan2_1_VAE
from .visual_context_adapter import VisualContextAdapter

__all__ = ["Wan2_1_VAE", "WanModel", "flash_attention", "VisualContextAdapter", "T5Encoder", "T5EncoderModel",
           "HuggingfaceTokenizer"]
