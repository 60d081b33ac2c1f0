"""Byte-level BPE tokenizer for CLIP prompts: the host-side replacement of ``clip.tokenize`` (clip/clip.py:199-239) and
``clip/simple_tokenizer.py:62-132`` for the trainer plugins, so that class names need no Python-side fixture and the
plugin does not have to import the reference's ``clip`` package.

The algorithm is OpenAI CLIP's published one: lower-cased, whitespace-normalised text is split by a fixed pattern, every
piece is mapped byte-wise to printable unicode, then adjacent symbols are merged greedily in the rank order of the merge
table ``bpe_simple_vocab_16e6.txt.gz`` (48 894 merges; ids: 256 byte symbols, the same 256 with the end-of-word mark,
one id per merge, then <|startoftext|> = 49406 and <|endoftext|> = 49407).  The merge table is DATA that ships with every
CLIP checkout (and with the reference, clip/); it is not part of this repository.  It is looked up, in this order, at an
explicit path, ``$MUDPT_BPE_VOCAB``, next to the CLIP checkpoint, and inside an importable ``clip`` package directory.
Pinned by tests/golden/tokenizer_cases.json (ids recorded from the reference's own tokenizer).
"""
from __future__ import annotations

import functools
import gzip
import html
import importlib.util
import os
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch

try:  # \p{L} / \p{N} classes need the third-party ``regex`` module (what the reference uses, simple_tokenizer.py:7)
    import regex as _re
    _PIECES = _re.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+", _re.IGNORECASE)
except ImportError:  # pragma: no cover - stdlib approximation: letters = word chars minus digits / underscore
    import re as _re
    _PIECES = _re.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[^\W\d_]+|\d|(?:[^\s\w]|_)+", _re.IGNORECASE)

VOCAB_FILE = "bpe_simple_vocab_16e6.txt.gz"
N_MERGES = 49152 - 256 - 2  # simple_tokenizer.py:67
SOT, EOT = "<|startoftext|>", "<|endoftext|>"
END = "</w>"


def find_vocab(explicit: Optional[str] = None, near: Optional[str] = None) -> str:
    """Path of the merge table, or a RuntimeError that says where it was looked for."""
    tried: List[str] = []
    cands = [explicit, os.environ.get("MUDPT_BPE_VOCAB")]
    if near:
        cands.append(os.path.join(os.path.dirname(os.path.abspath(near)), VOCAB_FILE))
    try:
        spec = importlib.util.find_spec("clip")  # a CLIP checkout on sys.path: its data file, without importing the package
        for loc in (spec.submodule_search_locations or []) if spec else []:
            cands.append(os.path.join(loc, VOCAB_FILE))
    except (ImportError, ValueError):
        pass
    for c in cands:
        if c:
            tried.append(c)
            if os.path.isfile(c):
                return c
    raise RuntimeError(f"CLIP's BPE merge table {VOCAB_FILE} not found (looked at: {tried or 'nowhere'}); set MUDPT_BPE_VOCAB "
                       "to the file that ships with CLIP (clip/bpe_simple_vocab_16e6.txt.gz)")


@functools.lru_cache(maxsize=1)
def _byte_symbols() -> Dict[int, str]:
    """The reversible byte -> printable-unicode table of GPT-2 / CLIP: printable latin-1 bytes map to themselves, the 68
    others to code points 256, 257, ... in byte order."""
    keep = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    table, extra = {}, 0
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
        else:
            table[b] = chr(256 + extra)
            extra += 1
    return table


class BPETokenizer:
    def __init__(self, vocab_path: Optional[str] = None, near: Optional[str] = None):
        path = find_vocab(vocab_path, near)
        with gzip.open(path, "rt", encoding="utf-8") as f:
            lines = f.read().split("\n")
        merges = [tuple(l.split()) for l in lines[1:1 + N_MERGES]]  # line 0 is a header
        if len(merges) != N_MERGES or any(len(m) != 2 for m in merges):
            raise RuntimeError(f"{path} is not CLIP's merge table ({len(merges)} usable merges, expected {N_MERGES})")
        self.rank: Dict[Tuple[str, str], int] = {m: i for i, m in enumerate(merges)}
        # ids in the order the reference builds its vocabulary (simple_tokenizer.py:69-74): byte symbols in the ORDER OF
        # bytes_to_unicode()'s value list (printable bytes first, then the remapped ones), the same with </w>, merges, specials
        keep = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
        order = keep + [b for b in range(256) if b not in keep]
        sym = _byte_symbols()
        base = [sym[b] for b in order]
        vocab = base + [s + END for s in base] + ["".join(m) for m in merges] + [SOT, EOT]
        self.ids: Dict[str, int] = {s: i for i, s in enumerate(vocab)}
        self.sot_id, self.eot_id = self.ids[SOT], self.ids[EOT]
        self._cache: Dict[str, List[int]] = {}

    def _merge(self, symbols: List[str]) -> List[str]:
        """Greedy BPE: repeatedly fuse every occurrence of the adjacent pair with the lowest merge rank."""
        while len(symbols) > 1:
            best, best_rank = None, None
            for pair in zip(symbols, symbols[1:]):
                r = self.rank.get(pair)
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = pair, r
            if best is None:
                break
            out, i = [], 0
            while i < len(symbols):
                if i + 1 < len(symbols) and symbols[i] == best[0] and symbols[i + 1] == best[1]:
                    out.append(best[0] + best[1])
                    i += 2
                else:
                    out.append(symbols[i])
                    i += 1
            symbols = out
        return symbols

    def _piece_ids(self, piece: str) -> List[int]:
        hit = self._cache.get(piece)
        if hit is not None:
            return hit
        if piece in (SOT, EOT):
            ids = [self.ids[piece]]
        else:
            sym = _byte_symbols()
            chars = [sym[b] for b in piece.encode("utf-8")]
            chars[-1] += END
            ids = [self.ids[s] for s in self._merge(chars)]
        self._cache[piece] = ids
        return ids

    def encode(self, text: str) -> List[int]:
        """simple_tokenizer.py:121-127 without ftfy (a no-op on the ASCII / well-formed unicode class names of the datasets):
        html-unescape twice, collapse whitespace, lower-case, split, BPE."""
        text = html.unescape(html.unescape(text)).strip()
        text = " ".join(text.split()).lower()
        out: List[int] = []
        for piece in _PIECES.findall(text):
            out.extend(self._piece_ids(piece))
        return out

    def __call__(self, texts: Iterable[str], context_length: int = 77, truncate: bool = False) -> torch.Tensor:
        """clip.tokenize (clip/clip.py:199-239): [n, context_length] int32, SOT + ids + EOT, zero padded; a prompt that does not
        fit raises RuntimeError unless truncate (then the last kept token becomes EOT)."""
        if isinstance(texts, str):
            texts = [texts]
        texts = list(texts)
        result = torch.zeros(len(texts), context_length, dtype=torch.int32)
        for i, t in enumerate(texts):
            ids = [self.sot_id] + self.encode(t) + [self.eot_id]
            if len(ids) > context_length:
                if not truncate:
                    raise RuntimeError(f"Input {t} is too long for context length {context_length}")
                ids = ids[:context_length]
                ids[-1] = self.eot_id
            result[i, :len(ids)] = torch.tensor(ids, dtype=torch.int32)
        return result


_default: Optional[BPETokenizer] = None


def tokenize(texts: Sequence[str], context_length: int = 77, truncate: bool = False, vocab_path: Optional[str] = None,
             near: Optional[str] = None) -> torch.Tensor:
    """Module-level ``clip.tokenize`` equivalent with a lazily built default tokenizer."""
    global _default
    if _default is None or vocab_path:
        tok = BPETokenizer(vocab_path, near)
        if not vocab_path:
            _default = tok
        return tok(texts, context_length, truncate)
    return _default(texts, context_length, truncate)
