"""TEST SCAFFOLDING, not product code: a stand-in voice prompt of the right SHAPE.

Derives a deterministic prompt from block energies (one frame of ``n_groups`` codes per codec frame, a hashed speaker
embedding) without running any model, so that decode-path parity tests and the CPU oracle can be driven by a voice prompt that
does not depend on the audio encoder under test (tests/test_model_shapes_gpu.py).  The product path encodes reference audio on
the GPU (``Engine.conditioning_from_audio`` -> ``rt_voice_encode``); nothing under ``rho_tts_amd/`` can import this file.
"""
import zlib
from typing import List, Optional

import numpy as np
import torch

from rho_tts_amd.config import ModelConfig
from rho_tts_amd.voice import VoiceConditioning
from rho_tts_amd.weights import hash_uniform


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
