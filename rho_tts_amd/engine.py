"""Host-side engine: one model on one GPU, batched text -> waveform.

This is the piece the provider and ``bench.py`` share.  Everything numeric
happens behind the C ABI (``librho_tts_amd.so``); this file only tokenises,
sizes batches, and moves pointers.  No CPU fallback exists: constructing an
Engine without a gfx950 GPU raises ``NativeUnavailable``.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _native
from ._native_model import NativeModel, RtSampling
from .config import ModelConfig, resolve
from .tokenizer import load_tokenizer
from .voice import VoiceConditioning, load_audio
from .weights import load_safetensors, synthetic_state

FRAME_SECONDS_PER_WORD = 0.35   # synthetic weights never emit EOS, so length is fixed by policy


@dataclass
class GenerationParams:
    do_sample: bool = True
    temperature: float = 0.9
    top_k: int = 50
    top_p: float = 1.0
    repetition_penalty: float = 1.05
    predictor_do_sample: Optional[bool] = None
    predictor_temperature: float = 0.9
    predictor_top_k: int = 50
    predictor_top_p: float = 1.0
    max_seconds: float = 60.0

    def talker(self) -> RtSampling:
        return RtSampling(int(self.do_sample), self.temperature, self.top_k, self.top_p, self.repetition_penalty)

    def predictor(self) -> RtSampling:
        ds = self.do_sample if self.predictor_do_sample is None else self.predictor_do_sample
        return RtSampling(int(ds), self.predictor_temperature, self.predictor_top_k, self.predictor_top_p, 1.0)


def pick_rows(plan: Sequence[int], max_batch: int) -> int:
    """Decode rows for a work list with estimated lengths ``plan`` on an engine of ``max_batch`` rows.  A frame on 33..64 rows
    costs up to ~1.4x a frame on 32 (measured), and a list-scheduled queue on R rows takes about
    max(longest, sum / R) frames: 32 busy rows with a queue beat 64 half-idle ones only when the lengths are ragged enough - so
    the narrower schedule is taken only when this model puts it clearly (10 %) ahead; n <= max_batch texts of similar length
    stay one static batch."""
    n = len(plan)
    full = min(n, max_batch)
    if full <= 32 or not plan:
        return full

    def cost(rows: int) -> float:
        frames = max(float(max(plan)), float(sum(plan)) / rows)
        return frames * (1.0 + 0.4 * max(0, rows - 32) / 32.0)
    return 32 if cost(32) < 0.9 * cost(full) else full


SYNTHETIC_ENV = "RHO_TTS_AMD_SYNTHETIC"


def find_checkpoint(model_path: str) -> bool:
    return os.path.isdir(model_path) and any(f.endswith(".safetensors") for f in os.listdir(model_path))


class Engine:
    """``model_path`` is a local directory with ``*.safetensors`` (+ optional ``config.json`` / ``tokenizer.json``), as the
    reference's ``from_pretrained`` accepts (providers/qwen.py:160-165).  There is no network here, so a hub id cannot be
    resolved: without a local checkpoint the engine REFUSES to start unless seeded synthetic weights were asked for
    explicitly (``synthetic=True`` or ``RHO_TTS_AMD_SYNTHETIC=1`` - benchmarks and parity tests) - it never hands random-weight
    noise to a caller who asked for a real model."""

    def __init__(self, model_path: str = "Qwen/Qwen3-TTS-12Hz-1.7B-Base", device_ordinal: int = 0, max_batch: int = 32,
                 weight_seed: int = 789, cfg: Optional[ModelConfig] = None, max_positions: Optional[int] = None,
                 synthetic: Optional[bool] = None):
        has_ckpt = find_checkpoint(model_path)
        if synthetic is None:
            synthetic = (not has_ckpt) and os.environ.get(SYNTHETIC_ENV, "") not in ("", "0")
        if not has_ckpt and not synthetic:
            raise ValueError(                                  # a configuration error: never retried by the pipeline (base_tts.py:786-787)
                f"no local checkpoint at {model_path!r}: expected a directory holding *.safetensors (this build cannot download "
                f"{model_path!r}).  For seeded synthetic weights of that architecture (benchmarks, parity tests) pass "
                f"synthetic=True or set {SYNTHETIC_ENV}=1.")
        self.cfg = cfg or resolve(model_path)
        self.model_path = model_path
        self.device_ordinal = device_ordinal
        self.device = torch.device(f"cuda:{device_ordinal}")
        self.ctx = _native.Context(device_ordinal)          # raises NativeUnavailable without a gfx950 GPU
        self.max_batch = max_batch
        has_ckpt = has_ckpt and not synthetic
        self.synthetic = not has_ckpt
        self.ignore_eos: Optional[bool] = None              # None: synthetic weights never emit EOS -> fixed lengths; real weights stop at EOS
        with torch.cuda.device(self.device):
            if has_ckpt:
                state = load_safetensors(self.cfg, model_path, device=self.device)
            else:
                state = synthetic_state(self.cfg, weight_seed, device=self.device)
            # (created after the load: a checkpoint without audio-encoder tensors has cleared cfg.codec.enc_filters by now)
            self.model = NativeModel(self.ctx, self.cfg, max_batch=max_batch, max_positions=max_positions)
            self.model.load_state(state)
            del state
            torch.cuda.empty_cache()
        self.tokenizer = load_tokenizer(model_path, self.cfg.text_vocab)
        self.voice: Optional[VoiceConditioning] = None
        self.params = GenerationParams()
        # Synthetic weights never emit end-of-sequence, so every length is imposed (frames_for).  A real checkpoint ends where
        # its end-of-sequence falls, somewhere around the planner's estimate: length_jitter = e emulates that - every text ends
        # at estimate x (1 + e (2u - 1)), u a hash of the text (the same alone and in any batch) - while schedules are still
        # planned on the estimates.  Benchmarks and scheduling tests only; 0 = lengths as estimated.
        self.length_jitter = 0.0

    def close(self) -> None:
        if getattr(self, "model", None) is not None:
            self.model.close()
            self.model = None
        if getattr(self, "ctx", None) is not None:
            self.ctx.close()
            self.ctx = None

    # ------------------------------------------------------------------ voice
    def set_voice(self, v: VoiceConditioning) -> int:
        self.voice = v
        return self.model.set_voice(v.language, v.speaker, v.speaker_embed, v.ref_text_ids, v.ref_codes)

    def conditioning_from_audio(self, audio_or_path, ref_text: str, language: str = "english") -> VoiceConditioning:
        """The conditioning front-end (the reference re-runs it on every call via ref_audio=path, qwen.py:253-258): the clip is
        encoded ON the GPU - conv encoder, transformer, residual vector quantiser, speaker head (rt_voice_encode) - into the
        reference codec frames and the speaker embedding of the prompt.  A clip may take at most half of the KV rows."""
        if self.cfg.codec.enc_filters <= 0:
            raise ValueError("this checkpoint ships no audio encoder (conditioning front-end): voice cloning from reference audio is "
                             "unavailable - use a built-in speaker, or a checkpoint that includes the speech-tokenizer encoder")
        audio = load_audio(audio_or_path, self.cfg.sample_rate) if isinstance(audio_or_path, str) else np.asarray(audio_or_path, np.float32)
        codes, spk = self.model.encode_voice(audio, max_frames=self.model.max_positions // 2)
        return VoiceConditioning(language, None, spk, self.tokenizer.encode(ref_text), codes)

    def set_voice_from_audio(self, audio_or_path, ref_text: str, language: str = "english") -> int:
        return self.set_voice(self.conditioning_from_audio(audio_or_path, ref_text, language))

    def set_builtin_voice(self, speaker: str, language: str = "english") -> int:
        return self.set_voice(VoiceConditioning(language, speaker, None, [], None))

    # ------------------------------------------------------------------ sizing
    def frames_for(self, text: str, n_tokens: int) -> int:
        cap = int(self.params.max_seconds * self.cfg.frame_rate)
        if self.synthetic:
            n_words = max(1, len(text.split()))
            rate = min(self.cfg.frame_rate, 12.5)          # test configs have toy codecs with absurd frame rates
            return max(2, min(cap, int(round(rate * FRAME_SECONDS_PER_WORD * n_words))))
        return max(2, min(cap, 8 + 6 * n_tokens))            # eos-terminated; generous cap of ~0.5 s per token

    def frames_actual(self, text: str, n_tokens: int) -> int:
        f = self.frames_for(text, n_tokens)
        if self.synthetic and self.length_jitter > 0:
            import zlib
            u = (zlib.crc32(text.encode()) & 0xFFFFFF) / float(1 << 24)
            f = max(2, int(round(f * (1.0 + self.length_jitter * (2.0 * u - 1.0)))))
        return f

    # ------------------------------------------------------------------ generate
    def generate_codes(self, texts: Sequence[str], seed: int, item_ids: Optional[Sequence[int]] = None, cancel_flag=None,
                       max_frames: Optional[Sequence[int]] = None, max_rows: int = 0) -> List[torch.Tensor]:
        if self.voice is None:
            raise ValueError("no voice set: reference audio (Base models) or a built-in speaker (CustomVoice) is required")
        ids = [self.tokenizer.encode(t) for t in texts]
        frames = list(max_frames) if max_frames is not None else [self.frames_actual(t, len(i)) for t, i in zip(texts, ids)]
        limit = self.model.max_positions - self.model.prefix_len() - 12     # (queued items step a few positions past their last frame)
        for i, f in zip(ids, frames):
            if len(i) + 2 + f > limit:
                raise RuntimeError(f"length: text of {len(i)} tokens + {f} frames exceeds the {limit} free KV rows")
        ignore_eos = self.synthetic if self.ignore_eos is None else bool(self.ignore_eos)
        return self.model.generate(ids, frames, self.params.talker(), self.params.predictor(), seed=seed, item_ids=item_ids,
                                   ignore_eos=ignore_eos, cancel_flag=cancel_flag, max_rows=max_rows)

    def vocode(self, codes: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        """Codec decoder with the reference architecture's chunking (chunk_frames + left_context_frames)."""
        c = self.cfg.codec
        out: List[Optional[torch.Tensor]] = [None] * len(codes)
        short = [i for i, x in enumerate(codes) if 0 < x.shape[0] <= c.chunk_frames]
        for i0 in range(0, len(short), self.max_batch):
            grp = short[i0:i0 + self.max_batch]
            for i, w in zip(grp, self.model.code2wav([codes[i] for i in grp])):
                out[i] = w
        for i, x in enumerate(codes):
            if x.shape[0] == 0:
                out[i] = torch.zeros(0, device=self.device)
            elif x.shape[0] > c.chunk_frames:
                parts, start = [], 0
                while start < x.shape[0]:
                    end = min(start + c.chunk_frames, x.shape[0])
                    ctx = c.left_context_frames if start - c.left_context_frames > 0 else start
                    w = self.model.code2wav([x[start - ctx:end]])[0]
                    parts.append(w[ctx * c.total_upsample:])
                    start = end
                out[i] = torch.cat(parts)
        return out  # type: ignore[return-value]

    # ------------------------------------------------------------------ sub-segment streaming
    def stream_codes(self, text: str, seed: int = 789, item_id: int = 0, first_chunk: int = 12, chunk: int = 36, cancel_flag=None,
                     max_frames: Optional[int] = None):
        """One text decoded in pieces (rt_generate_begin / _step / _peek / _end): yields ``(codes [n, n_groups] int64, last)`` as
        the frames come off the GPU - ``first_chunk`` frames first (time to first audio), then ``chunk`` at a time.  The codes
        are those of the one-call ``generate_codes`` for the same (text, seed, item_id), however they are cut."""
        if self.voice is None:
            raise ValueError("no voice set: reference audio (Base models) or a built-in speaker (CustomVoice) is required")
        ids = self.tokenizer.encode(text)
        frames = int(max_frames) if max_frames is not None else self.frames_actual(text, len(ids))
        limit = self.model.max_positions - self.model.prefix_len() - 12
        if len(ids) + 2 + frames > limit:
            raise RuntimeError(f"length: text of {len(ids)} tokens + {frames} frames exceeds the {limit} free KV rows")
        ignore_eos = self.synthetic if self.ignore_eos is None else bool(self.ignore_eos)
        self.model.generate_begin([ids], [frames], self.params.talker(), self.params.predictor(), seed=seed, item_ids=[item_id],
                                  ignore_eos=ignore_eos, cancel_flag=cancel_flag)
        try:
            sent, step = 0, max(1, int(first_chunk))
            while True:
                _, done = self.model.generate_step(step)
                codes, _ = self.model.generate_peek(0, sent, frames)
                sent += int(codes.shape[0])
                if codes.shape[0] or done:
                    yield codes, done
                if done:
                    return
                step = max(1, int(chunk))
        finally:
            self.model.generate_end()

    def stream_wav(self, text: str, seed: int = 789, item_id: int = 0, first_chunk: int = 12, chunk: int = 36, cancel_flag=None,
                   max_frames: Optional[int] = None):
        """Raw waveform chunks (GPU float32) of one text while it is still being decoded: every batch of new codec frames is
        vocoded with ``left_context_frames`` of the frames before it - the codec decoder's own chunked decode, with the chunk
        boundaries where the frames arrive.  The pieces are cut by ABSOLUTE sample position: n frames decode to
        ``wav_length(n) = n * up - d`` samples (the transposed convs trim both sides; d = 555 for the 1920x decoder), so a piece
        decoded from frame ``start - ctx`` on begins at sample ``emitted - (start - ctx) * up`` of its decode - the left context
        regenerates the d samples the previous piece could not produce yet - and the pieces played back to back are exactly
        ``wav_length(total frames)`` samples, without a gap at the boundaries.  Yields ``(wav, last)``."""
        c = self.cfg.codec
        up = c.total_upsample
        have: Optional[torch.Tensor] = None
        emitted = 0                                            # samples handed out so far = absolute position of the next one
        for codes, last in self.stream_codes(text, seed, item_id, first_chunk, chunk, cancel_flag, max_frames):
            if codes.shape[0] == 0:
                if last:
                    yield torch.zeros(0, device=self.device), True
                continue
            start = 0 if have is None else int(have.shape[0])
            have = codes if have is None else torch.cat([have, codes])
            ctx = c.left_context_frames if start - c.left_context_frames > 0 else start
            w = self.model.code2wav([have[start - ctx:]])[0]
            off = max(0, emitted - (start - ctx) * up)
            piece = w[off:]
            emitted += int(piece.numel())
            yield piece, last

    def plan_batches(self, frames: Sequence[int]) -> List[List[int]]:
        """Batches of ``max_batch`` text indices.  One batch keeps arrival order; more are bucketed by frame budget, longest
        first: a batch decodes until its longest member is done, so similar lengths keep its rows busy (dist.bucket_batches)."""
        from .dist import bucket_batches
        return bucket_batches(frames, self.max_batch)

    def pick_rows(self, plan: Sequence[int]) -> int:
        return pick_rows(plan, self.max_batch)

    def synthesize(self, texts: Sequence[str], seed: int = 789, item_ids: Optional[Sequence[int]] = None, cancel_flag=None,
                   max_frames: Optional[Sequence[int]] = None, stats: Optional[dict] = None,
                   continuous: Optional[bool] = None, plan_frames: Optional[Sequence[int]] = None) -> List[torch.Tensor]:
        """Raw waveforms (GPU float32, 1-D) for any number of texts, in the order of ``texts``.
        The RNG stream of a text is its ``item_ids`` entry (default: its index), so the result does not depend on how the
        texts are scheduled.  More texts than ``max_batch`` are decoded with continuous batching (``continuous``, default on
        when the decode rows allow it): ONE rt_generate call with the texts queued longest first, finished rows handed to the
        next queued text, then the codec decoder over batches sorted by the lengths actually produced.  ``continuous=False``
        cuts static batches bucketed by frame budget instead (plan_batches).  ``plan_frames``: the length ESTIMATES the
        schedule is planned with when they differ from the budgets ``max_frames`` (a real checkpoint ends at end-of-sequence,
        somewhere below its budget).  ``stats`` (optional dict) receives kept / launched row-frames of the schedule."""
        n = len(texts)
        ids = list(item_ids) if item_ids is not None else list(range(n))
        if max_frames is not None:
            frames = [int(f) for f in max_frames]
        else:
            frames = [self.frames_actual(t, len(self.tokenizer.encode(t))) for t in texts]
        if plan_frames is not None:
            plan = [int(f) for f in plan_frames]
        elif max_frames is None and self.synthetic and self.length_jitter > 0:
            plan = [self.frames_for(t, len(self.tokenizer.encode(t))) for t in texts]      # the estimates, not the jittered lengths
        else:
            plan = frames
        wavs: List[Optional[torch.Tensor]] = [None] * n
        rows = self.pick_rows(plan)
        if continuous is None:
            continuous = self.max_batch <= 64 and (n > self.max_batch or rows < min(n, self.max_batch))
        if continuous and self.max_batch > 64:
            raise ValueError(f"continuous batching decodes on at most 64 rows (engine built with max_batch={self.max_batch})")
        if continuous and (n > self.max_batch or rows < n):
            order = sorted(range(n), key=lambda i: (-plan[i], i))          # longest first: short items fill the tail of the schedule
            codes = self.generate_codes([texts[i] for i in order], seed, [ids[i] for i in order], cancel_flag, [frames[i] for i in order],
                                        max_rows=rows if rows < min(n, self.max_batch) else 0)
            st = self.model.generate_stats()
            by_len = sorted(range(n), key=lambda j: (-int(codes[j].shape[0]), j))
            for j, w in zip(by_len, self.vocode([codes[j] for j in by_len])):
                wavs[order[j]] = w
            if stats is not None:
                stats["frames"] = stats.get("frames", 0) + st["frames_kept"]
                stats["padded_frames"] = stats.get("padded_frames", 0) + st["frames_run"] * st["rows"]
                stats["batches"] = stats.get("batches", 0) + 1
                stats["hand_overs"] = stats.get("hand_overs", 0) + st["hand_overs"]
            return wavs  # type: ignore[return-value]
        batches = self.plan_batches(plan)
        for idx in batches:
            codes = self.generate_codes([texts[i] for i in idx], seed, [ids[i] for i in idx], cancel_flag, [frames[i] for i in idx])
            for i, w in zip(idx, self.vocode(codes)):
                wavs[i] = w
        if stats is not None:
            stats["frames"] = stats.get("frames", 0) + sum(frames)
            stats["padded_frames"] = stats.get("padded_frames", 0) + sum(max(frames[i] for i in idx) * len(idx) for idx in batches)
            stats["batches"] = stats.get("batches", 0) + len(batches)
        return wavs  # type: ignore[return-value]

    # ------------------------------------------------------------------ post
    def post_process(self, items: Sequence[Sequence[torch.Tensor]], params: _native.PostParams):
        return self.ctx.post_process(params, items)
