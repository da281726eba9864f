"""Text -> token ids.

The reference leaves tokenisation to the third-party model
(providers/qwen.py:247-258 passes raw strings).  When ``model_path`` holds a
``tokenizer.json`` it is used through the ``tokenizers`` library; otherwise — and
always for synthetic-weight runs — a deterministic word hash into the text
vocabulary stands in ("Synthetic inputs").
"""
from __future__ import annotations

import os
import re
import zlib
from typing import List

_WORD = re.compile(r"\w+|[^\w\s]", re.UNICODE)


class HashTokenizer:
    """One id per word / punctuation mark: crc32(word) folded into [0, n_plain)."""

    def __init__(self, text_vocab: int, reserved_top: int = 4096):
        # keep clear of the control ids that live at the top of the vocabulary
        self.n_plain = max(16, text_vocab - reserved_top if text_vocab > 2 * reserved_top else (text_vocab * 3) // 4)

    def encode(self, text: str) -> List[int]:
        return [zlib.crc32(w.lower().encode("utf-8")) % self.n_plain for w in _WORD.findall(text)]


class FileTokenizer:
    def __init__(self, path: str):
        from tokenizers import Tokenizer
        self.tk = Tokenizer.from_file(path)

    def encode(self, text: str) -> List[int]:
        return list(self.tk.encode(text).ids)


def load_tokenizer(model_path: str, text_vocab: int):
    p = os.path.join(model_path, "tokenizer.json") if os.path.isdir(model_path) else None
    if p and os.path.exists(p):
        return FileTokenizer(p)
    return HashTokenizer(text_vocab)
