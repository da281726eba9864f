"""Voice-conditioning containers and audio loading, plus a SYNTHETIC code generator for decode-path tests.

The product path encodes reference audio on the GPU (``Engine.conditioning_from_audio`` -> ``rt_voice_encode``: conv encoder,
transformer, residual vector quantiser, speaker head; reference call site providers/qwen.py:253-258).  ``conditioning_from_audio``
below is NOT that: it derives a deterministic prompt of the right SHAPE from block energies (one frame of ``n_groups`` codes per
1920 samples, a hashed embedding) without running any model, and is kept only so that decode-path parity tests and the CPU
oracle can be driven by a voice prompt that does not depend on the encoder under test (tests/test_model_shapes_gpu.py).
"""
from __future__ import annotations

import wave
import zlib
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch

from .config import ModelConfig
from .weights import hash_uniform


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


def conditioning_from_audio(cfg: ModelConfig, audio: np.ndarray, ref_text_ids: List[int], language: str = "english",
                            max_frames: Optional[int] = None) -> VoiceConditioning:
    hop, G, cb = cfg.codec.total_upsample, cfg.n_groups, cfg.codec.codebook_size
    x = np.asarray(audio, dtype=np.float32).reshape(-1)
    T = x.size // hop
    if max_frames is not None:
        T = min(T, max_frames)
    if T < 1:
        raise ValueError("reference audio is shorter than one codec frame")
    fr = x[: T * hop].reshape(T, hop)
    sub = hop // G
    blocks = fr[:, : sub * G].reshape(T, G, sub).astype(np.float64)
    e = np.log10((blocks ** 2).mean(axis=2) + 1e-10)                               # [T, G] in about [-10, 0]
    zc = (np.diff(np.signbit(blocks), axis=2) != 0).mean(axis=2)                   # [T, G] zero-crossing rate
    q = np.clip((e + 8.0) / 8.0, 0.0, 1.0) * 0.75 + np.clip(zc * 4.0, 0.0, 1.0) * 0.25
    codes = np.minimum((q * cb).astype(np.int64), cb - 1)
    stats = np.round(np.concatenate([e.mean(axis=0), zc.mean(axis=0), [float(T)]]) * 64).astype(np.int64)
    seed = zlib.crc32(stats.tobytes()) | (0x5EA7 << 32)
    emb = next(hash_uniform(cfg.talker.hidden, seed))[1] * (0.05 * 12 ** 0.5)
    return VoiceConditioning(language, None, emb.to(torch.float32), list(ref_text_ids), torch.from_numpy(codes))


def synthetic_reference_clip(seconds: float = 30.0, sr: int = 24000, seed: int = 789) -> np.ndarray:
    """SURVEY.md 8d's benchmark voice: 5 harmonics of a 120 Hz glottal-like tone, 4 Hz AM, N(0, 0.01^2) noise."""
    n = int(seconds * sr)
    t = np.arange(n, dtype=np.float64) / sr
    x = sum((0.5 / (k + 1)) * np.sin(2 * np.pi * 120.0 * (k + 1) * t) for k in range(5))
    x = x * (0.6 + 0.4 * np.sin(2 * np.pi * 4.0 * t)) * 0.3
    x = x + np.random.default_rng(seed).normal(0.0, 0.01, n)
    return x.astype(np.float32)
