"""CPU oracle of the speech-to-text path.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference path: validation/stt/stt_validator.py:42-148.  Its model is Whisper "tiny"; its own fallback implementation is
transformers' ``AutoModelForSpeechSeq2Seq`` + ``AutoProcessor`` (:85-107) - and transformers IS importable in the build container
and on the GPU box.  So, unlike the TTS model, this oracle is not a sibling: the log-mel front-end is transformers'
``WhisperFeatureExtractor`` itself and the network is transformers' ``WhisperForConditionalGeneration`` itself, built from a
``WhisperConfig`` (no download) and loaded with the same seeded weights the HIP path gets.

Parity status:
  * log-mel features, encoder, decoder, logits: PINNED to the reference's own dependency (run live in the tests; a committed
    fixture - tests/golden/stt_golden.npz, written by tests/golden/make_stt_golden.py - guards against a transformers upgrade).
  * greedy decoding: the reference's pipeline calls ``model.generate`` of a PRETRAINED checkpoint, whose generation config
    (forced language / task ids, suppressed ids) comes with the download.  No checkpoint is available offline, so the decoding
    rule is restated here: forced prefix, argmax over the ids the suppression rule allows, stop at end-of-sequence.
  * resampler (TTS rate -> 16 kHz): PARITY UNPINNED - the reference's pipeline decodes its temporary WAV through ffmpeg, absent
    here.  ``resample`` restates the windowed-sinc definition of csrc/stt.hip (stt_resampler) in float64.
  * trained weights: none offline; seeded synthetic weights only (the architecture is checked, not the transcription quality).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch


def resample(x: np.ndarray, sr_in: int, sr_out: int, width: float = 6.0, rolloff: float = 0.99) -> np.ndarray:
    """Windowed-sinc resampling, float64: y[n] = sum_t x[t] g(t - n sr_in / sr_out), g = low-pass at rolloff x the lower Nyquist
    under a Hann window spanning `width` zero crossings (csrc/stt.hip stt_resampler / k_resample, same taps)."""
    if sr_in == sr_out:
        return np.asarray(x, dtype=np.float32)
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    g = math.gcd(sr_in, sr_out)
    L, M = sr_out // g, sr_in // g
    base = min(sr_in, sr_out) * rolloff
    half = int(math.ceil(width * sr_in / base))
    taps = 2 * half + 1
    h = np.zeros((L, taps), dtype=np.float64)
    for p in range(L):
        t = (np.arange(taps, dtype=np.float64) - half - p / L) * base / sr_in
        w = np.cos(t * np.pi / width / 2.0) ** 2
        sinc = np.where(t == 0.0, 1.0, np.sin(np.pi * t) / np.where(t == 0.0, 1.0, np.pi * t))
        h[p] = np.where(np.abs(t) < width, sinc * w * base / sr_in, 0.0)
    h = h.astype(np.float32).astype(np.float64)                       # the device holds the taps in float32
    n_out = (x.shape[0] * L + M - 1) // M
    xp = np.concatenate([np.zeros(half), x, np.zeros(half + 2)])
    n = np.arange(n_out, dtype=np.int64)
    basei = (n * M) // L
    phase = (n * M) % L
    idx = basei[:, None] + np.arange(taps)[None, :]                   # (+ half from the left padding, - half from the tap origin)
    return (xp[idx] * h[phase]).sum(axis=1).astype(np.float32)


def feature_extractor(cfg):
    from transformers import WhisperFeatureExtractor
    return WhisperFeatureExtractor(feature_size=cfg.n_mels, sampling_rate=cfg.sample_rate, hop_length=cfg.hop, chunk_length=cfg.chunk_seconds,
                                   n_fft=cfg.n_fft)


def log_mel(cfg, pcm16k: np.ndarray) -> torch.Tensor:
    """[n_mels][frames]: WhisperFeatureExtractor on a 16-kHz clip (padded / cut to one chunk), feature_extraction_whisper.py."""
    fe = feature_extractor(cfg)
    return fe(np.asarray(pcm16k, dtype=np.float32), sampling_rate=cfg.sample_rate, return_tensors="pt").input_features[0]


def build(cfg, state):
    """transformers' Whisper of the configured shape with ``state`` (transformers' own tensor names) as float32 weights."""
    from transformers import WhisperConfig, WhisperForConditionalGeneration
    hc = WhisperConfig(vocab_size=cfg.vocab, num_mel_bins=cfg.n_mels, encoder_layers=cfg.enc_layers, decoder_layers=cfg.dec_layers,
                       encoder_attention_heads=cfg.heads, decoder_attention_heads=cfg.heads, d_model=cfg.d_model, encoder_ffn_dim=cfg.ffn,
                       decoder_ffn_dim=cfg.ffn, max_source_positions=cfg.n_ctx, max_target_positions=cfg.n_text_ctx, eos_token_id=cfg.eos_id,
                       pad_token_id=cfg.eos_id, bos_token_id=cfg.eos_id, decoder_start_token_id=int(cfg.prefix[0]), attn_implementation="eager")
    m = WhisperForConditionalGeneration(hc).eval()
    sd = {k: v.detach().to("cpu", torch.float32) for k, v in state.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and set(missing) <= {"proj_out.weight"}, (missing, unexpected)       # proj_out is tied to embed_tokens
    return m


@torch.no_grad()
def encode(model, mel: torch.Tensor) -> torch.Tensor:
    return model.model.encoder(mel[None]).last_hidden_state[0]


@torch.no_grad()
def greedy(model, cfg, mel: torch.Tensor, max_new: Optional[int] = None) -> Tuple[List[int], torch.Tensor]:
    """Forced prefix, then argmax over the allowed ids (ids >= suppress_from never, eos excepted; begin_suppress not as the first
    generated token), lowest index on ties, until end-of-sequence.  Returns (ids after the prefix, logits behind the prefix)."""
    enc = model.model.encoder(mel[None])
    toks = list(int(t) for t in cfg.prefix)
    out: List[int] = []
    first = None
    never = torch.zeros(cfg.vocab, dtype=torch.bool)
    if cfg.suppress_from > 0:
        never[cfg.suppress_from:] = True
        never[cfg.eos_id] = False
    for t in getattr(cfg, "suppress_tokens", ()):                    # a generation config's `suppress_tokens`
        if 0 <= int(t) < cfg.vocab and int(t) != cfg.eos_id:
            never[int(t)] = True
    budget = min(int(max_new or cfg.max_new_tokens), cfg.n_text_ctx - len(cfg.prefix))
    for step in range(budget):
        lg = model(encoder_outputs=enc, decoder_input_ids=torch.tensor([toks])).logits[0, -1].float()
        if first is None:
            first = lg.clone()
        bad = never.clone()
        if step == 0:
            for t in cfg.begin_suppress:
                bad[int(t)] = True
        lg = lg.masked_fill(bad, float("-inf"))
        tok = int(torch.argmax(lg))                                   # (torch.argmax returns the first maximal index)
        if tok == cfg.eos_id:
            break
        out.append(tok)
        toks.append(tok)
    return out, first


def transcribe_windows(model, cfg, pcm: np.ndarray, sr: int, max_tokens: Optional[int] = None) -> List[int]:
    """Audio of any length: consecutive ``chunk_seconds`` windows of the INPUT, each resampled, featurised, encoded and decoded
    greedily on its own behind the forced prefix (at most ``max_new_tokens`` ids per window), ids concatenated - the definition
    rt_stt_transcribe implements (the reference's transcribers walk 30-s windows too, stt_validator.py:133-141, at seek positions
    taken from timestamp tokens, which the forced <|notimestamps|> prefix rules out here: long-form parity is unpinned)."""
    pcm = np.asarray(pcm, dtype=np.float32)
    win = int(cfg.chunk_seconds) * int(sr)
    n_win = max(1, -(-len(pcm) // win))
    cap = int(max_tokens or n_win * cfg.max_new_tokens)
    out: List[int] = []
    for w in range(n_win):
        if len(out) >= cap:
            break
        x = pcm[w * win:(w + 1) * win]
        x16 = resample(x, sr, cfg.sample_rate) if sr != cfg.sample_rate else x
        ids, _ = greedy(model, cfg, log_mel(cfg, x16), min(cap - len(out), cfg.max_new_tokens))
        out += ids
    return out
