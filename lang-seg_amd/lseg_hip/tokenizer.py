"""Label strings -> CLIP token ids (host side; replaces the `clip.tokenize` call sites at
modules/models/lseg_net.py:158,163-164).

Resolution order:
  1. the `clip` package if importable (exactly what the reference uses);
  2. the in-tree byte-pair encoder if the CLIP vocabulary file is available
     (env LSEG_BPE_VOCAB or <pkg>/bpe_simple_vocab_16e6.txt.gz) -- restates the published
     CLIP SimpleTokenizer algorithm ([3P] openai/CLIP@04f4dc2 clip/simple_tokenizer.py);
  3. `synthetic_tokens` (hash ids) with a warning -- only good for synthetic-weight runs.
"""
import gzip
import html
import os
import re
import warnings
from functools import lru_cache
from typing import List, Sequence, Union

import torch

from .synth import synthetic_tokens, SOT_TOKEN, EOT_TOKEN

_HERE = os.path.dirname(os.path.abspath(__file__))


def _vocab_path():
    p = os.environ.get("LSEG_BPE_VOCAB", os.path.join(_HERE, "bpe_simple_vocab_16e6.txt.gz"))
    return p if os.path.exists(p) else None


@lru_cache()
def _bytes_to_unicode():
    # reversible byte <-> printable-unicode table of the GPT-2/CLIP BPE
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + \
        list(range(ord("\xae"), ord("\xff") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


class BPETokenizer:
    """Byte-pair encoder over the CLIP vocabulary (49152 merges -> 49408 ids)."""

    def __init__(self, vocab_path: str):
        merges = gzip.open(vocab_path).read().decode("utf-8").split("\n")
        merges = [tuple(m.split()) for m in merges[1:49152 - 256 - 2 + 1]]
        b2u = _bytes_to_unicode()
        vocab = list(b2u.values())
        vocab = vocab + [v + "</w>" for v in vocab]
        vocab += ["".join(m) for m in merges]
        vocab += ["<|startoftext|>", "<|endoftext|>"]
        self.enc = {v: i for i, v in enumerate(vocab)}
        self.ranks = {m: i for i, m in enumerate(merges)}
        self.b2u = b2u
        self.cache = {"<|startoftext|>": "<|startoftext|>", "<|endoftext|>": "<|endoftext|>"}
        try:
            import regex
            self.pat = regex.compile(
                r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""",
                regex.IGNORECASE)
        except ImportError:      # ASCII-only approximation
            self.pat = re.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[A-Za-z]+|[0-9]|[^\sA-Za-z0-9]+",
                                  re.IGNORECASE)

    def _bpe(self, token: str) -> str:
        if token in self.cache:
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        while len(word) > 1:
            pairs = {(word[i], word[i + 1]) for i in range(len(word) - 1)}
            best = min(pairs, key=lambda p: self.ranks.get(p, float("inf")))
            if best not in self.ranks:
                break
            a, b = best
            out, i = [], 0
            while i < len(word):
                if i < len(word) - 1 and word[i] == a and word[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(word[i])
                    i += 1
            word = tuple(out)
        res = " ".join(word)
        self.cache[token] = res
        return res

    def encode(self, text: str) -> List[int]:
        try:
            import ftfy
            text = ftfy.fix_text(text)
        except ImportError:
            pass
        text = html.unescape(html.unescape(text))
        text = re.sub(r"\s+", " ", text).strip().lower()
        ids = []
        for tok in self.pat.findall(text):
            tok = "".join(self.b2u[b] for b in tok.encode("utf-8"))
            ids.extend(self.enc[t] for t in self._bpe(tok).split(" "))
        return ids


    def decode(self, ids: Sequence[int]) -> str:
        """[3P] SimpleTokenizer.decode: vocabulary strings joined, byte table inverted, '</w>' -> ' '."""
        if not hasattr(self, "dec"):
            self.dec = {i: v for v, i in self.enc.items()}
            self.u2b = {u: b for b, u in self.b2u.items()}
        text = "".join(self.dec[int(i)] for i in ids)
        return bytearray(self.u2b[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")


_bpe = None
_allow_synth = None          # None: read LSEG_SYNTHETIC_TOKENS


def allow_synthetic_tokens(enabled: bool = True) -> None:
    """Opt in to (or out of) hash-based stand-in token ids when no real CLIP tokenizer is available (synthetic-weight runs only)."""
    global _allow_synth
    _allow_synth = bool(enabled)


def synthetic_tokens_allowed() -> bool:
    if _allow_synth is not None:
        return _allow_synth
    return os.environ.get("LSEG_SYNTHETIC_TOKENS", "") not in ("", "0")


def tokenize(texts: Union[str, Sequence[str]], context_length: int = 77, vocab: int = 49408) -> torch.Tensor:
    """int64 [K, context_length] = [SOT] + bpe(text) + [EOT], zero padded (same contract as
    clip.tokenize; raises if a text does not fit)."""
    global _bpe
    if isinstance(texts, str):
        texts = [texts]
    texts = list(texts)
    if vocab >= 49408:
        try:
            import clip  # noqa: F401
            return clip.tokenize(texts, context_length=context_length).to(torch.int64)
        except ImportError:
            pass
        vp = _vocab_path()
        if vp is not None:
            if _bpe is None:
                _bpe = BPETokenizer(vp)
            out = torch.zeros((len(texts), context_length), dtype=torch.int64)
            for i, t in enumerate(texts):
                ids = [SOT_TOKEN] + _bpe.encode(t) + [EOT_TOKEN]
                if len(ids) > context_length:
                    raise RuntimeError(f"Input {t} is too long for context length {context_length}")
                out[i, :len(ids)] = torch.tensor(ids)
            return out
        # No real tokenizer here.  Hash-based ids are only meaningful with SYNTHETIC weights (tests, bench.py, smoke: random-init
        # towers): with a real checkpoint they would score nonsense prompts and produce wrong masks without any error.  So this is a
        # hard failure unless the caller opted in (allow_synthetic_tokens(True) / LSEG_SYNTHETIC_TOKENS=1), never a warning.
        if not synthetic_tokens_allowed():
            raise RuntimeError(
                "clip.tokenize is unavailable: neither the `clip` package nor CLIP's BPE vocabulary (bpe_simple_vocab_16e6.txt.gz; set "
                "LSEG_BPE_VOCAB or place it next to lseg_hip/tokenizer.py) was found.  Refusing to fall back to hash-based synthetic token "
                "ids: with a real checkpoint they give wrong masks.  For synthetic-weight runs call "
                "lseg_hip.tokenizer.allow_synthetic_tokens(True) or set LSEG_SYNTHETIC_TOKENS=1.")
    return synthetic_tokens(texts, vocab, context_length)
