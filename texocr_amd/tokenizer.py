"""Byte-level BPE tokenizer next to the generate() path (reference tokenizer/tokenizer.py:7-238).

Only ``decode`` is needed to turn generated ids into LaTeX (reference model/ocr_model.py:104-107); ``encode`` is
provided so that the vocabulary file round-trips and the on-disk format is pinned.  The format is the
reference's: three lines -- vocab size, ``repr`` of the special-token dict, ``repr`` of the merges dict
(tokenizer.py:110-125).  The reference ``eval()``s those lines; here they are parsed with
``ast.literal_eval`` (same accepted inputs for well-formed files, no code execution).
"""
from __future__ import annotations

import ast
import re as _re
from typing import Dict, List, Tuple

import regex

# GPT-4 style split pattern used by the reference (tokenizer.py:5)
SPLIT_PATTERN = r"""'(?i:[sdmt]|ll|ve|re)|[^\r\n\p{L}\p{N}]?+\p{L}+| ?\p{N}{1,3}| ?[^\s\p{L}\p{N}]++[\r\n]*|\s*[\r\n]|\s+(?!\S)|\s+"""


class RegExTokenizer:
    def __init__(self, vocab_size: int = 800, pattern: str = SPLIT_PATTERN, special_tokens: Dict[str, int] = None):
        self.vocab_size = vocab_size
        self.split_pattern = pattern
        self.re_pattern = regex.compile(pattern)
        self.special_tokens: Dict[str, int] = dict(special_tokens or {})
        self.bp_merges: Dict[Tuple[int, int], int] = {}
        self._rebuild()

    # ---- vocabulary -----------------------------------------------------------------------------
    def _rebuild(self) -> None:
        vocab = {i: bytes([i]) for i in range(256)}                       # tokenizer.py:18
        for (i, j), tid in self.bp_merges.items():                          # insertion order = merge order (:21-22)
            vocab[tid] = vocab[i] + vocab[j]
        for tok, tid in self.special_tokens.items():                        # :25-26
            vocab[tid] = tok.encode("utf-8")
        self.vocab = vocab
        self.inv_special_tokens = {v: k for k, v in self.special_tokens.items()}

    def load(self, path: str) -> None:
        with open(path, "r") as f:
            self.vocab_size = int(f.readline())
            self.special_tokens = ast.literal_eval(f.readline())
            self.bp_merges = ast.literal_eval(f.readline())
        if not isinstance(self.special_tokens, dict) or not isinstance(self.bp_merges, dict):
            raise ValueError("malformed tokenizer file")
        self._rebuild()

    @classmethod
    def from_tables(cls, vocab_size: int, special_tokens: Dict[str, int], merges) -> "RegExTokenizer":
        """Build from a merge table [(a, b, new_id), ...] in merge order."""
        tk = cls(vocab_size=vocab_size, special_tokens=special_tokens)
        tk.bp_merges = {(int(a), int(b)): int(t) for a, b, t in merges}
        tk._rebuild()
        return tk

    def save(self, path: str) -> None:
        with open(path, "w") as f:
            f.write(f"{self.vocab_size}\n{self.special_tokens}\n{self.bp_merges}\n")

    # ---- decode ----------------------------------------------------------------------------------
    def decode_list(self, tokens: List[int]) -> List[str]:
        out = []
        for t in tokens:
            t = int(t)
            if t in self.inv_special_tokens:
                out.append(self.inv_special_tokens[t])
            elif t in self.vocab:
                out.append(self.vocab[t].decode("utf-8", errors="replace"))   # tokenizer.py:234
            else:
                raise ValueError(f"Token {t} not found in vocabulary.")
        return out

    def decode(self, tokens: List[int]) -> str:
        return "".join(self.decode_list(tokens))

    # ---- encode ----------------------------------------------------------------------------------
    def _encode_split(self, split: str) -> List[int]:
        """Greedy BPE: repeatedly merge the adjacent pair with the lowest merge id (tokenizer.py:197-216)."""
        ids = list(split.encode("utf-8"))
        while len(ids) >= 2:
            best, best_rank = None, None
            for pair in zip(ids, ids[1:]):
                r = self.bp_merges.get(pair)
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = pair, r
            if best is None:
                break
            merged, i = [], 0
            while i < len(ids):
                if i < len(ids) - 1 and (ids[i], ids[i + 1]) == best:
                    merged.append(best_rank)
                    i += 2
                else:
                    merged.append(ids[i])
                    i += 1
            ids = merged
        return ids

    def _encode_text(self, text: str) -> List[int]:
        ids: List[int] = []
        for split in regex.findall(self.re_pattern, text):
            ids.extend(self._encode_split(split))
        return ids

    def encode(self, text: str) -> List[int]:
        if not self.special_tokens:
            return self._encode_text(text)
        pat = "(" + "|".join(regex.escape(t) for t in self.special_tokens) + ")"   # tokenizer.py:177-178
        ids: List[int] = []
        for part in regex.split(pat, text):
            if part in self.special_tokens:
                ids.append(self.special_tokens[part])
            else:
                ids.extend(self._encode_text(part))
        return ids


def process_output(output: str) -> str:
    """LaTeX post-processing of the wrapper (reference utils.py:73-79): keep one space after a control word that is
    followed by a letter/digit, drop all other whitespace."""
    output = _re.sub(r"(\\[a-zA-Z]+)\s+([a-zA-Z0-9])", r"\1<SPACE>\2", output)
    output = _re.sub(r"\s+", "", output)
    return output.replace("<SPACE>", " ")
