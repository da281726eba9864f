"""CPU oracle of the token sampler (TEST INFRASTRUCTURE ONLY — see oracle/postprocess.py header).

The reference delegates sampling to the third-party ``qwen_tts`` model, which
draws from torch's global RNG seeded by ``BaseTTS._set_seeds``
(base_tts.py:142-149, called per segment at :764).  That stream cannot be
reproduced on a GPU batch, so this build defines its own: a counter-based
uniform keyed by (seed, item index, frame, code group), and a fully ordered
top-k / top-p / inverse-CDF draw.  **Parity unpinned** against the reference's
sampling (no reference test or fixture pins it); the HIP sampler is pinned to
this file bit-for-bit on the uniform and to token equality on the draw.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

M32 = 0xFFFFFFFF


def mix32(h: int) -> int:
    h &= M32
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & M32
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & M32
    h ^= h >> 16
    return h


def uniform(seed: int, item: int, frame: int, group: int) -> np.float32:
    """float32 in (0, 1): the one random number a draw consumes."""
    a = mix32((seed & M32) ^ 0x85EBCA6B)
    b = mix32(a + item * 0x9E3779B1)
    c = mix32(b + frame * 0x85EBCA77)
    d = mix32(c + group * 0xC2B2AE3D + ((seed >> 32) & M32))
    return np.float32((np.float32(d >> 8) + np.float32(0.5)) * np.float32(1.0 / (1 << 24)))


@dataclass
class SamplingParams:
    do_sample: bool = False
    temperature: float = 0.9
    top_k: int = 50            # 1..64 when sampling
    top_p: float = 1.0
    repetition_penalty: float = 1.0


def scan_f32(p: np.ndarray) -> np.ndarray:
    """Inclusive prefix sums of up to 64 float32 values in the order the GPU wave computes them: a Kogge-Stone scan, offsets
    1, 2, 4, ... 32, every addition rounded to float32.  (A defined order, not the left-to-right one: float addition is not
    associative, and the draw has to be the same token on both sides.)"""
    x = np.zeros(64, dtype=np.float32)
    x[: p.shape[0]] = p
    d = 1
    while d < 64:
        y = x.copy()
        y[d:] = (x[d:] + x[:-d]).astype(np.float32)
        x = y
        d *= 2
    return x[: p.shape[0]]


def draw(logits: np.ndarray, sp: SamplingParams, u: np.float32, suppress: np.ndarray | None = None,
         seen: np.ndarray | None = None) -> int:
    """One token from one row of float32 logits.

    suppress: bool[V], True = token forbidden.  seen: bool[V], tokens already
    emitted by this sequence (repetition penalty).  All arithmetic float32, in
    the order written here (running sums: scan_f32) — the HIP kernel follows the same order.
    """
    l = np.array(logits, dtype=np.float32, copy=True)
    if seen is not None and sp.repetition_penalty != 1.0:
        pen = np.float32(sp.repetition_penalty)
        l = np.where(seen, np.where(l > 0, l / pen, l * pen), l).astype(np.float32)
    if suppress is not None:
        l[suppress] = -np.inf
    if not sp.do_sample:
        return int(np.argmax(l))            # first maximum = lowest index
    if not (1 <= sp.top_k <= 64):
        raise ValueError("sampling needs 1 <= top_k <= 64")
    l = (l / np.float32(sp.temperature)).astype(np.float32)
    order = np.lexsort((np.arange(l.shape[0]), -l))[: sp.top_k]      # logit desc, index asc
    order = order[np.isfinite(l[order])]
    if order.size == 0:
        return int(np.argmax(l))
    m = l[order[0]]
    p = np.exp((l[order] - m).astype(np.float32)).astype(np.float32)
    cum = scan_f32(p)                              # cum[j] = p[0] + ... + p[j] in the scan's evaluation order
    total = cum[-1]
    keep = p.shape[0]
    if sp.top_p < 1.0:
        lim = np.float32(np.float32(sp.top_p) * total)
        reach = np.nonzero(cum >= lim)[0]
        if reach.size:
            keep = int(reach[0]) + 1
            total = cum[keep - 1]
    target = np.float32(u * total)
    over = np.nonzero(cum[:keep] > target)[0]
    return int(order[int(over[0])]) if over.size else int(order[keep - 1])
