"""Tokenizer wrapper with the reference's interface (reference: omnivideo/modules/tokenizers.py).
`transformers.AutoTokenizer` reads the tokenizer files from a local directory; nothing is fetched.
`ftfy` (mojibake repair, tokenizers.py:12) is used when it is importable; without it the text is
only HTML-unescaped and a warning is logged once, since unusual inputs may then tokenize
differently from the reference."""
import html
import logging
import string

import regex as re

__all__ = ["HuggingfaceTokenizer"]

try:
    import ftfy
except ImportError:          # pragma: no cover - depends on the installation
    ftfy = None
_warned = False


def basic_clean(text):
    global _warned
    if ftfy is not None:
        text = ftfy.fix_text(text)
    elif not _warned:
        logging.warning("ftfy is not installed: prompts are not mojibake-repaired before tokenization")
        _warned = True
    text = html.unescape(html.unescape(text))
    return text.strip()


def whitespace_clean(text):
    return re.sub(r"\s+", " ", text).strip()


def canonicalize(text, keep_punctuation_exact_string=None):
    text = text.replace("_", " ")
    table = str.maketrans("", "", string.punctuation)
    if keep_punctuation_exact_string:
        text = keep_punctuation_exact_string.join(part.translate(table)
                                                  for part in text.split(keep_punctuation_exact_string))
    else:
        text = text.translate(table)
    return re.sub(r"\s+", " ", text.lower()).strip()


class HuggingfaceTokenizer:

    def __init__(self, name, seq_len=None, clean=None, **kwargs):
        assert clean in (None, "whitespace", "lower", "canonicalize")
        from transformers import AutoTokenizer
        self.name, self.seq_len, self.clean = name, seq_len, clean
        self.tokenizer = AutoTokenizer.from_pretrained(name, local_files_only=True, **kwargs)
        self.vocab_size = self.tokenizer.vocab_size

    def __call__(self, sequence, **kwargs):
        return_mask = kwargs.pop("return_mask", False)
        _kwargs = {"return_tensors": "pt"}
        if self.seq_len is not None:
            _kwargs.update({"padding": "max_length", "truncation": True, "max_length": self.seq_len})
        _kwargs.update(**kwargs)
        if isinstance(sequence, str):
            sequence = [sequence]
        if self.clean:
            sequence = [self._clean(u) for u in sequence]
        ids = self.tokenizer(sequence, **_kwargs)
        return (ids.input_ids, ids.attention_mask) if return_mask else ids.input_ids

    def _clean(self, text):
        if self.clean == "whitespace":
            return whitespace_clean(basic_clean(text))
        if self.clean == "lower":
            return whitespace_clean(basic_clean(text)).lower()
        return canonicalize(basic_clean(text))
