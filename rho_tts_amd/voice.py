"""Voice-conditioning container, audio loading and the benchmark's synthetic reference clip.

Reference audio is encoded on the GPU (``Engine.conditioning_from_audio`` -> ``rt_voice_encode``: conv encoder, transformer,
residual vector quantiser, speaker head; reference call site providers/qwen.py:253-258).  Nothing in this package fabricates
conditioning: the stand-in prompt generator some decode-path tests use lives with them (tests/fake_voice.py).
"""
from __future__ import annotations

import wave
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch



@dataclass
class VoiceConditioning:
    language: str
    speaker: Optional[str]                 # built-in voice name, or None
    speaker_embed: Optional[torch.Tensor]  # [hidden] float32, or None
    ref_text_ids: List[int]
    ref_codes: Optional[torch.Tensor]      # [T_ref, n_groups] int64, or None


def load_audio(path: str, target_sr: int) -> np.ndarray:
    """Mono float32 in [-1, 1] at ``target_sr`` from a PCM .wav (stdlib) or a .npy array."""
    if path.endswith(".npy"):
        x = np.load(path, allow_pickle=False).astype(np.float32).reshape(-1)
        sr = target_sr
    else:
        with wave.open(path, "rb") as wf:
            sr, n, ch, sw = wf.getframerate(), wf.getnframes(), wf.getnchannels(), wf.getsampwidth()
            raw = wf.readframes(n)
        if sw == 2:
            x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
        elif sw == 4:
            x = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
        elif sw == 1:
            x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
        else:
            raise ValueError(f"unsupported sample width {sw} in {path}")
        if ch > 1:
            x = x.reshape(-1, ch).mean(axis=1)
    if sr != target_sr and x.size:
        pos = np.arange(int(round(x.size * target_sr / sr)), dtype=np.float64) * (sr / target_sr)
        x = np.interp(pos, np.arange(x.size, dtype=np.float64), x).astype(np.float32)
    return np.ascontiguousarray(x, dtype=np.float32)


def synthetic_reference_clip(seconds: float = 30.0, sr: int = 24000, seed: int = 789) -> np.ndarray:
    """SURVEY.md 8d's benchmark voice: 5 harmonics of a 120 Hz glottal-like tone, 4 Hz AM, N(0, 0.01^2) noise."""
    n = int(seconds * sr)
    t = np.arange(n, dtype=np.float64) / sr
    x = sum((0.5 / (k + 1)) * np.sin(2 * np.pi * 120.0 * (k + 1) * t) for k in range(5))
    x = x * (0.6 + 0.4 * np.sin(2 * np.pi * 4.0 * t)) * 0.3
    x = x + np.random.default_rng(seed).normal(0.0, 0.01, n)
    return x.astype(np.float32)
