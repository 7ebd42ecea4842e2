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


_PUNCT = str.maketrans("", "", string.punctuation)
_SPACES = re.compile(r"\s+")


def basic_clean(text):
    """Mojibake repair (when ftfy is present), two rounds of HTML unescaping, outer whitespace off
    (reference tokenizers.py:12-15)."""
    global _warned
    if ftfy is not None:
        text = ftfy.fix_text(text)
    elif not _warned:
        logging.warning("ftfy is not installed: prompts are not mojibake-repaired before tokenization")
        _warned = True
    return html.unescape(html.unescape(text)).strip()


def whitespace_clean(text):
    """Runs of whitespace become one space (reference tokenizers.py:18-21)."""
    return _SPACES.sub(" ", text).strip()


def canonicalize(text, keep_punctuation_exact_string=None):
    """Underscores to spaces, punctuation dropped (except the given literal), lower case, single
    spaces (reference tokenizers.py:24-35)."""
    text = text.replace("_", " ")
    pieces = text.split(keep_punctuation_exact_string) if keep_punctuation_exact_string else [text]
    text = (keep_punctuation_exact_string or "").join(p.translate(_PUNCT) for p in pieces)
    return _SPACES.sub(" ", text.lower()).strip()


_CLEANERS = {
    None: None,
    "whitespace": lambda t: whitespace_clean(basic_clean(t)),
    "lower": lambda t: whitespace_clean(basic_clean(t)).lower(),
    "canonicalize": lambda t: canonicalize(basic_clean(t)),
}


class HuggingfaceTokenizer:
    """`tokenizer(texts, return_mask=False, **hf_kwargs)` -> ids (and mask), padded and truncated to
    `seq_len` when one is given (reference tokenizers.py:38-81). Files come from the local directory
    `name` only."""

    def __init__(self, name, seq_len=None, clean=None, **kwargs):
        if clean not in _CLEANERS:
            raise AssertionError(f"clean must be one of {sorted(k for k in _CLEANERS if k)} or None")
        from transformers import AutoTokenizer
        self.name, self.seq_len, self.clean = name, seq_len, clean
        self._clean = _CLEANERS[clean]
        self.tokenizer = AutoTokenizer.from_pretrained(name, local_files_only=True, **kwargs)
        self.vocab_size = self.tokenizer.vocab_size

    def __call__(self, sequence, return_mask=False, **kwargs):
        texts = [sequence] if isinstance(sequence, str) else list(sequence)
        if self._clean is not None:
            texts = [self._clean(t) for t in texts]
        options = {"return_tensors": "pt"}
        if self.seq_len is not None:
            options.update(padding="max_length", truncation=True, max_length=self.seq_len)
        options.update(kwargs)
        enc = self.tokenizer(texts, **options)
        return (enc.input_ids, enc.attention_mask) if return_mask else enc.input_ids
