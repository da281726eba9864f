"""Host-side mirror of the reference's plugin interface for the generation path.

When the ``rho_tts`` package is importable the provider subclasses ITS ``BaseTTS`` and registers
with ITS ``TTSFactory`` (see ``api.py``) — that is the drop-in.  On machines without it (the GPU
box, CI) this module supplies the same names with the same argument meaning and error behaviour,
written from the interface description in SURVEY.md section 8b, so that the parity tests read like
the reference's own.  Only what the hot path touches is mirrored; validation (drift / STT),
isolation and the UI are out of scope.

Reference interfaces mirrored (paths relative to /root/reference/src/rho_tts/):
  exceptions.py:9-31, cancellation.py:14-65, result.py:14-34, provider_info.py:12-27,
  factory.py:43-176 (registry part), base_tts.py:36-1196 (constructor attributes, text helpers,
  generate / async_generate / stream, wav fallback writer).
"""
from __future__ import annotations

import asyncio
import logging
import os
import random
import threading
import time
import traceback
import wave
from abc import ABC, abstractmethod
from dataclasses import dataclass, field
from typing import Callable, Dict, Generator, List, Optional, Tuple, Type, Union

import numpy as np
import torch

logger = logging.getLogger("rho_tts_amd")

SUPPORTED_FORMATS = {"wav", "mp3", "flac", "ogg"}


# ----------------------------------------------------------------------------- errors
class RhoTTSError(Exception):
    """Root of the library's exception tree."""


class ProviderNotFoundError(RhoTTSError):
    pass


class ModelLoadError(RhoTTSError):
    pass


class AudioGenerationError(RhoTTSError):
    pass


class FormatConversionError(RhoTTSError):
    pass


class CancelledException(RhoTTSError):
    pass


# ----------------------------------------------------------------------------- cancellation
class CancellationToken:
    """Cooperative, thread-safe cancel flag polled at text / segment / iteration granularity."""

    def __init__(self):
        self._flag = threading.Event()
        self._guard = threading.Lock()

    def cancel(self) -> None:
        with self._guard:
            self._flag.set()

    def is_cancelled(self) -> bool:
        return self._flag.is_set()

    def raise_if_cancelled(self, message: Optional[str] = None) -> None:
        if self._flag.is_set():
            raise CancelledException(message or "Task was cancelled")

    def reset(self) -> None:
        with self._guard:
            self._flag.clear()


# ----------------------------------------------------------------------------- data classes
@dataclass
class GenerationResult:
    path: Optional[str] = None
    audio: Optional[torch.Tensor] = None
    sample_rate: int = 0
    duration_sec: float = 0.0
    segments_count: int = 0
    format: str = "wav"
    drift_prob: Optional[float] = None
    text_similarity: Optional[float] = None
    decay_ratio: Optional[float] = None


@dataclass
class VoiceInfo:
    id: str
    name: str
    language: str = "English"
    is_builtin: bool = True


@dataclass
class ProviderInfo:
    name: str
    supports_voice_cloning: bool = False
    supported_languages: List[str] = field(default_factory=list)
    builtin_voices: List[VoiceInfo] = field(default_factory=list)


# ----------------------------------------------------------------------------- text helpers
def split_text_into_segments(text: str, max_chars: int, force_sentence_split: bool) -> List[str]:
    """Sentence / word / hard-cut segmentation with the reference's exact outcomes (base_tts.py:538-585),
    including its quirks: sentences are told apart from the last one BY VALUE, a sentence that follows a
    pending segment is never word-wrapped, and an over-long single word is cut at ``max_chars``."""
    sentences = text.split(". ")
    tail = sentences[-1]
    forced = force_sentence_split and len(sentences) > 1
    done: List[str] = []
    pending = ""

    def flush():
        nonlocal pending
        if pending:
            done.append(pending.strip())
        pending = ""

    for raw in sentences:
        sent = raw if raw == tail else raw + ". "
        fits = len(pending) + len(sent) <= max_chars
        if fits and not forced:
            pending += sent
            continue
        if pending:
            flush()
            pending = sent
            continue
        if len(sent) <= max_chars:
            done.append(sent.strip())
            continue
        for word in sent.split():                       # word wrap of an over-long leading sentence
            if len(pending) + len(word) + 1 > max_chars:
                if pending:
                    flush()
                    pending = word
                else:
                    done.append(word[:max_chars])
            else:
                pending = f"{pending} {word}" if pending else word
    if pending.strip():
        done.append(pending.strip())
    return done


def apply_phonetic_mapping(text: str, mapping: Dict[str, str]) -> str:
    for src, dst in mapping.items():          # insertion order, plain substring replacement (base_tts.py:187-200)
        text = text.replace(src, dst)
    return text


def write_wav_pcm16(path: str, audio: torch.Tensor, sample_rate: int) -> None:
    """Mono 16-bit PCM with the reference's fallback conversion: clip to [-1, 1], times 32767, truncate
    (base_tts.py:664-671)."""
    a = audio.detach().reshape(-1).to("cpu", torch.float32).numpy()
    pcm = (np.clip(a, -1.0, 1.0) * 32767).astype(np.int16)
    with wave.open(path, "wb") as wf:
        wf.setnchannels(1)
        wf.setsampwidth(2)
        wf.setframerate(sample_rate)
        wf.writeframes(pcm.tobytes())


# ----------------------------------------------------------------------------- base class
class BaseTTS(ABC):
    """Provider contract: implement ``_generate_audio`` and ``sample_rate``; everything else is shared."""

    MAX_MODEL_CHARS = 3000
    BYTES_PER_CHAR_ESTIMATE = 500_000

    def __init__(self, device: str = "cuda", seed: int = 789, deterministic: bool = False,
                 phonetic_mapping: Optional[Dict[str, str]] = None):
        self.device = device
        self.seed = seed
        self.deterministic = deterministic
        self.phonetic_mapping = dict(phonetic_mapping) if phonetic_mapping is not None else {}
        self._set_seeds()
        self.max_chars_per_segment = 800
        self.max_iterations = 1
        self.accent_drift_threshold = 0.17
        self.text_similarity_threshold = 0.85
        self.sound_decay_threshold = 0.3
        self.max_decay_retries = 3
        self.silence_threshold_db = -50.0
        self.crossfade_duration_sec = 0.05
        self.trim_silence = True
        self.fade_duration_sec = 0.02
        self.force_sentence_split = True
        self.inter_sentence_pause_sec = 0.1
        self.voice_id: Optional[str] = None
        self.drift_model_path: Optional[str] = None
        self.auto_sort_good_threshold = None
        self.auto_sort_bad_threshold = None
        self.auto_sort_good_dir = None
        self.auto_sort_bad_dir = None
        self._max_chars_explicit = False
        self._max_model_chars = self.MAX_MODEL_CHARS
        self._voice_encoder = None
        self.reference_embedding = None

    def close(self) -> None:
        pass

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        self.close()
        return False

    @classmethod
    def provider_info(cls) -> ProviderInfo:
        return ProviderInfo(name=cls.__name__)

    # -- seeds / sizing ---------------------------------------------------------
    def _set_seeds(self) -> None:
        random.seed(self.seed)
        np.random.seed(self.seed % (2 ** 32))
        torch.manual_seed(self.seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(self.seed)

    def _get_available_memory_bytes(self) -> int:
        if str(self.device).startswith("cuda") and torch.cuda.is_available():
            free, _ = torch.cuda.mem_get_info()
            return int(free)
        try:
            import psutil
            return int(psutil.virtual_memory().available)
        except ImportError:
            try:
                return os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_AVPHYS_PAGES")
            except (ValueError, OSError):
                return 8 * 1024 ** 3

    def _compute_max_chars(self) -> int:
        if self._max_chars_explicit:
            return self.max_chars_per_segment
        per_char = self.BYTES_PER_CHAR_ESTIMATE
        budget = int(self._get_available_memory_bytes() / per_char) if per_char > 0 else self._max_model_chars
        return max(int(min(self._max_model_chars, budget) * 0.8), 200)

    def _apply_phonetic_mapping(self, text: str) -> str:
        return apply_phonetic_mapping(text, self.phonetic_mapping)

    def _split_text_into_segments(self, text: str, max_chars: int) -> List[str]:
        return split_text_into_segments(text, max_chars, self.force_sentence_split)

    # -- provider contract --------------------------------------------------------
    @abstractmethod
    def _generate_audio(self, text: Union[str, List[str]], **kwargs) -> Union[torch.Tensor, List[torch.Tensor]]:
        ...

    @property
    @abstractmethod
    def sample_rate(self) -> int:
        ...

    def _post_process_audio(self, audio: torch.Tensor) -> torch.Tensor:
        return audio

    @abstractmethod
    def _run_pipeline(self, texts: List[str], cancellation_token: CancellationToken,
                      progress_callback: Optional[Callable[[str], None]] = None
                      ) -> List[Optional[Tuple[torch.Tensor, int, dict]]]:
        ...

    # -- output helpers -----------------------------------------------------------
    def _apply_speed_pitch(self, audio: torch.Tensor, speed: float, pitch_semitones: float) -> torch.Tensor:
        try:
            import torchaudio
        except ImportError as e:
            raise RuntimeError("speed / pitch adjustment needs torchaudio (out of the hot path's scope)") from e
        if speed != 1.0:
            audio = torchaudio.functional.resample(audio.unsqueeze(0) if audio.dim() == 1 else audio,
                                                   int(self.sample_rate * speed), self.sample_rate)
            if audio.dim() == 2 and audio.shape[0] == 1:
                audio = audio.squeeze(0)
        if pitch_semitones != 0.0:
            one_d = audio.dim() == 1
            audio = torchaudio.functional.pitch_shift(audio.unsqueeze(0) if one_d else audio, self.sample_rate, pitch_semitones)
            if one_d:
                audio = audio.squeeze(0)
        return audio

    def _save_wav(self, path: str, audio: torch.Tensor, sample_rate: int) -> None:
        write_wav_pcm16(path, audio, sample_rate)

    @staticmethod
    def _convert_format(wav_path: str, target_format: str) -> str:
        try:
            from pydub import AudioSegment
        except ImportError:
            raise FormatConversionError("pydub is required for format conversion. Install with: pip install pydub")
        try:
            out = wav_path.rsplit(".", 1)[0] + f".{target_format}"
            AudioSegment.from_wav(wav_path).export(out, format=target_format)
            os.remove(wav_path)
            return out
        except Exception as e:  # noqa: BLE001
            raise FormatConversionError(f"Failed to convert to {target_format}: {e}")

    # -- public API -----------------------------------------------------------------
    def generate(self, texts: Union[str, List[str]], output_path: Optional[str] = None,
                 cancellation_token: Optional[CancellationToken] = None, format: str = "wav", speed: float = 1.0,
                 pitch_semitones: float = 0.0, progress_callback: Optional[Callable[[str], None]] = None):
        """str -> GenerationResult | None; list -> list (None per failed item) | None if every item failed.
        Unsupported format and ValueError propagate; cancellation and anything else return None."""
        if format not in SUPPORTED_FORMATS:
            raise FormatConversionError(f"Unsupported format '{format}'. Supported: {', '.join(sorted(SUPPORTED_FORMATS))}")
        single = isinstance(texts, str)
        batch = [texts] if single else list(texts)
        try:
            token = cancellation_token or CancellationToken()
            produced = self._run_pipeline(batch, token, progress_callback)
            results: List[Optional[GenerationResult]] = []
            for idx, item in enumerate(produced):
                if item is None:
                    results.append(None)
                    continue
                audio, n_segments, meta = item
                if speed != 1.0 or pitch_semitones != 0.0:
                    audio = self._apply_speed_pitch(audio, speed, pitch_semitones)
                n_samples = audio.shape[-1] if audio.dim() == 2 else audio.numel()
                res = GenerationResult(audio=audio, sample_rate=self.sample_rate, duration_sec=n_samples / self.sample_rate,
                                       segments_count=n_segments, format=format, drift_prob=meta.get("drift_prob"),
                                       text_similarity=meta.get("text_similarity"), decay_ratio=meta.get("decay_ratio"))
                if output_path is not None:
                    try:
                        target = output_path if single else f"{output_path}_{idx}.wav"
                        wav_path = target if format == "wav" else (target.rsplit(".", 1)[0] + ".wav" if "." in target else target + ".wav")
                        mono = audio.detach()                # (a provider may convert to PCM on the device before the copy)
                        self._save_wav(wav_path, mono.unsqueeze(0) if mono.dim() == 1 else mono, self.sample_rate)
                        res.path = wav_path if format == "wav" else self._convert_format(wav_path, format)
                    except FormatConversionError:
                        raise
                    except Exception as e:  # noqa: BLE001
                        logger.error(f"Failed to save audio for item {idx}: {e}")
                        res.path = None
                results.append(res)
            if not any(r is not None for r in results):
                logger.error("All text items failed to generate")
                return None
            return results[0] if single else results
        except CancelledException as e:
            logger.warning(f"Generation cancelled: {e}")
            return None
        except (FormatConversionError, ValueError):
            raise
        except Exception as e:  # noqa: BLE001
            logger.error(f"Error in TTS generation: {e}")
            traceback.print_exc()
            return None

    async def async_generate(self, texts, output_path=None, cancellation_token=None, format="wav", speed=1.0,
                             pitch_semitones=0.0, progress_callback=None):
        loop = asyncio.get_running_loop()
        return await loop.run_in_executor(None, lambda: self.generate(
            texts, output_path=output_path, cancellation_token=cancellation_token, format=format, speed=speed,
            pitch_semitones=pitch_semitones, progress_callback=progress_callback))

    def stream(self, text: str, cancellation_token: Optional[CancellationToken] = None, speed: float = 1.0,
               pitch_semitones: float = 0.0) -> Generator[GenerationResult, None, None]:
        """One result per segment, no crossfade, per-segment failures skipped (base_tts.py:1132-1190)."""
        token = cancellation_token or CancellationToken()
        segments = self._split_text_into_segments(self._apply_phonetic_mapping(text), self._compute_max_chars())
        for n, seg in enumerate(segments):
            if token.is_cancelled():
                return
            self._set_seeds()
            try:
                audio = self._post_process_audio(self._generate_audio(seg))
            except Exception as e:  # noqa: BLE001
                logger.warning(f"Segment {n + 1} failed: {e}")
                continue
            audio = self._apply_fades(self._remove_dc_offset(self._trim_silence(audio, True, True)), True, True)
            if speed != 1.0 or pitch_semitones != 0.0:
                audio = self._apply_speed_pitch(audio, speed, pitch_semitones)
            n_samples = audio.shape[-1] if audio.dim() == 2 else audio.numel()
            yield GenerationResult(audio=audio, sample_rate=self.sample_rate, duration_sec=n_samples / self.sample_rate,
                                   segments_count=1, format="wav")


# ----------------------------------------------------------------------------- registry
class TTSFactory:
    """Class-level provider registry (factory.py:43-176 without the isolation fallback)."""

    _providers: Dict[str, Type[BaseTTS]] = {}

    @classmethod
    def register_provider(cls, name: str, provider_class: Type[BaseTTS]) -> None:
        if not (isinstance(provider_class, type) and issubclass(provider_class, BaseTTS)):
            raise TypeError(f"{provider_class} must inherit from BaseTTS")
        cls._providers[name] = provider_class

    @classmethod
    def get_tts_instance(cls, provider: str = "qwen", **kwargs) -> BaseTTS:
        if provider in cls._providers:
            return cls._providers[provider](**kwargs)
        known = ", ".join(cls.list_providers()) or "(none registered)"
        raise ProviderNotFoundError(f"Unknown TTS provider: '{provider}'. Available providers: {known}.")

    @classmethod
    def list_providers(cls) -> List[str]:
        return sorted(cls._providers)

    @classmethod
    def get_provider_info(cls, provider: str) -> ProviderInfo:
        if provider in cls._providers:
            return cls._providers[provider].provider_info()
        known = ", ".join(cls.list_providers()) or "(none registered)"
        raise ProviderNotFoundError(f"Unknown TTS provider: '{provider}'. Available providers: {known}.")

    @classmethod
    def list_voices(cls, provider: str) -> List[VoiceInfo]:
        return cls.get_provider_info(provider).builtin_voices
