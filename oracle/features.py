"""TEST INFRASTRUCTURE - CPU restatement of the 30 hand-crafted dimensions of the drift classifier's feature vector.

Reference: ``extract_features`` / ``_estimate_formants`` (validation/classifier/trainer.py:23-96): 13 MFCC means + 13 MFCC
standard deviations, F0 mean / std over the voiced frames of ``librosa.pyin``, and the first two LPC formants of a mid-file
frame.  (The other 256 dimensions are resemblyzer's pretrained speaker embedding: no weights offline, not restated.)

**PARITY UNPINNED.**  ``librosa`` (pyproject.toml:53, ``>=0.10``) and ``resemblyzer`` are not installed here and the reference
holds no feature fixtures, so nothing below could be checked against the reference's own numbers.  What is restated is the
PUBLISHED behaviour of the librosa 0.10 calls the reference makes, with their defaults:

  librosa.load(path, sr=16000)          mono float32 at 16 kHz.  librosa resamples with soxr_hq (absent): the windowed-sinc
                                        resampler of this build is used instead (oracle/whisper.py ``resample``) - own definition.
  librosa.feature.mfcc(y, sr, n_mfcc=13)   STFT n_fft 2048, hop 512, periodic Hann, center=True with ZERO padding (0.10's
                                        ``pad_mode="constant"``), power spectrum, 128 slaney-scale slaney-normalised mel filters
                                        0 .. sr/2, ``power_to_db`` (ref 1, amin 1e-10, top_db 80), DCT-II "ortho" -> rows 0..12.
  librosa.pyin(y, fmin=C2, fmax=C7)      probabilistic YIN, frame 2048 / window 1024 / hop 512, 100 thresholds under a Beta(2, 18)
                                        prior, Boltzmann(2) prior over the troughs, 0.1-semitone pitch bins, triangular transition
                                        window of +-50 bins, voicing switch probability 0.01, Viterbi decoding.  The reference does
                                        NOT pass ``sr``: pyin assumes its default 22050 Hz for audio sampled at 16000 Hz, so every
                                        F0 it reports is 22050 / 16000 times the acoustic one.  Restated as written.
  librosa.lpc(frame, order=18)           Burg's method (Marple's recursion) on the pre-emphasised (0.97), symmetric-Hann-windowed
                                        25-ms frame around the middle sample; formants = angles of the roots in the upper half
                                        plane, 90 Hz < f < sr / 4, the lowest two.

Only tests/ may import this file; the product's extractor is rho_tts_amd/features.py (GPU front halves in csrc/features.hip).
Every function here is written for clarity (loops, float64), not speed.
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np

from .whisper import resample

SR = 16000
N_FFT, HOP, N_MELS, N_MFCC = 2048, 512, 128, 13
PYIN_SR = 22050                                   # librosa.pyin's default, which the reference leaves in place (trainer.py:52)
FMIN = 440.0 * 2.0 ** ((36 - 69) / 12.0)          # librosa.note_to_hz('C2') = 65.406 Hz  (MIDI 36)
FMAX = 440.0 * 2.0 ** ((96 - 69) / 12.0)          # librosa.note_to_hz('C7') = 2093.005 Hz (MIDI 96)
FRAME, WIN, PHOP = 2048, 1024, 512                # pyin: frame_length, win_length = frame // 2, hop_length = frame // 4
LPC_ORDER = max(12, SR // 1000 + 2)               # 18


# ------------------------------------------------------------------------------------------------ MFCC
def hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) / (np.log(6.4) / 27.0), f / (200.0 / 3.0))


def mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), (200.0 / 3.0) * m)


def mel_filterbank(sr: int = SR, n_fft: int = N_FFT, n_mels: int = N_MELS) -> np.ndarray:
    """[n_mels][n_fft // 2 + 1]: librosa.filters.mel(sr, n_fft, n_mels, fmin=0, fmax=sr/2, htk=False, norm='slaney')."""
    fft_f = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(0.0), hz_to_mel(sr / 2.0), n_mels + 2))
    fb = np.zeros((n_mels, fft_f.shape[0]))
    for i in range(n_mels):
        lower = (fft_f - mel_f[i]) / (mel_f[i + 1] - mel_f[i])
        upper = (mel_f[i + 2] - fft_f) / (mel_f[i + 2] - mel_f[i + 1])
        fb[i] = np.maximum(0.0, np.minimum(lower, upper)) * (2.0 / (mel_f[i + 2] - mel_f[i]))
    return fb


def dct_matrix(n_out: int = N_MFCC, n_in: int = N_MELS) -> np.ndarray:
    """Rows 0 .. n_out - 1 of the orthonormal DCT-II (scipy.fftpack.dct(type=2, norm='ortho'))."""
    k = np.arange(n_out)[:, None]
    n = np.arange(n_in)[None, :]
    m = np.cos(np.pi * k * (2 * n + 1) / (2.0 * n_in)) * math.sqrt(2.0 / n_in)
    m[0] *= math.sqrt(0.5)
    return m


def mfcc(y16k: np.ndarray) -> np.ndarray:
    """[13][frames] as librosa.feature.mfcc(y=y, sr=16000, n_mfcc=13)."""
    y = np.asarray(y16k, dtype=np.float64)
    yp = np.concatenate([np.zeros(N_FFT // 2), y, np.zeros(N_FFT // 2)])
    n_frames = 1 + (yp.shape[0] - N_FFT) // HOP
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(N_FFT) / N_FFT)         # periodic Hann (get_window('hann', n, fftbins=True))
    fb = mel_filterbank()
    logmel = np.zeros((N_MELS, n_frames))
    for f in range(n_frames):
        spec = np.fft.rfft(yp[f * HOP: f * HOP + N_FFT] * win)
        logmel[:, f] = 10.0 * np.log10(np.maximum(1e-10, fb @ (spec.real ** 2 + spec.imag ** 2)))
    logmel = np.maximum(logmel, logmel.max() - 80.0)                          # power_to_db(top_db=80)
    return dct_matrix() @ logmel


# ------------------------------------------------------------------------------------------------ pYIN
def pitch_geometry() -> Tuple[int, int, int, int]:
    """(min_period, max_period, bins per semitone, pitch bins) of pyin(fmin=C2, fmax=C7, sr=22050, resolution=0.1)."""
    min_period = max(int(math.floor(PYIN_SR / FMAX)), 1)
    max_period = min(int(math.ceil(PYIN_SR / FMIN)), FRAME - WIN - 1)
    bps = int(math.ceil(1.0 / 0.1))
    n_bins = int(math.floor(12 * bps * math.log2(FMAX / FMIN))) + 1
    return min_period, max_period, bps, n_bins


def cmnd_frames(y16k: np.ndarray) -> np.ndarray:
    """[frames][max_period - min_period + 1] cumulative-mean-normalised difference (librosa.core.pitch
    ``_cumulative_mean_normalized_difference``): d(tau) = sum_{j=1..1024} (x[j] - x[j + tau])^2 over the zero-padded
    (center=True) frames, d'(tau) = d(tau) / mean(d(1 .. tau)), for tau = min_period .. max_period."""
    min_p, max_p, _, _ = pitch_geometry()
    y = np.asarray(y16k, dtype=np.float64)
    yp = np.concatenate([np.zeros(FRAME // 2), y, np.zeros(FRAME // 2)])
    n_frames = 1 + (yp.shape[0] - FRAME) // PHOP
    out = np.zeros((n_frames, max_p - min_p + 1))
    tiny = np.finfo(np.float64).tiny
    for f in range(n_frames):
        x = yp[f * PHOP: f * PHOP + FRAME]
        d = np.zeros(max_p + 1)
        for tau in range(1, max_p + 1):
            diff = x[1: WIN + 1] - x[1 + tau: WIN + 1 + tau]
            d[tau] = float(np.dot(diff, diff))
        cum_mean = np.cumsum(d[1:]) / np.arange(1, max_p + 1)                 # mean of d(1 .. tau)
        out[f] = d[min_p: max_p + 1] / (cum_mean[min_p - 1: max_p] + tiny)
    return out


def parabolic_shifts(frame: np.ndarray) -> np.ndarray:
    s = np.zeros_like(frame)
    a = (frame[:-2] + frame[2:] - 2.0 * frame[1:-1]) / 2.0
    b = (frame[2:] - frame[:-2]) / 2.0
    s[1:-1] = -b / (2.0 * a + np.finfo(np.float64).tiny)
    s[np.abs(s) > 1.0] = 0.0
    return s


def beta_cdf_2_18(x: float) -> float:
    """Regularised incomplete beta I_x(2, 18) in closed form (integer parameters): 1 - (1 - x)^18 (1 + 18 x)."""
    return 1.0 - (1.0 - x) ** 18 * (1.0 + 18.0 * x)


def observation_probs(cm: np.ndarray) -> np.ndarray:
    """[2 P][frames]: per frame the probability mass pyin puts on every pitch bin (voiced states 0 .. P - 1) and, spread evenly,
    on the unvoiced states (librosa.core.pitch ``__pyin_helper``)."""
    min_p, _, bps, P = pitch_geometry()
    n_frames, n_lags = cm.shape
    thr = np.linspace(0.0, 1.0, 101)
    beta_probs = np.diff(np.array([beta_cdf_2_18(t) for t in thr]))          # prior mass of each of the 100 thresholds
    obs = np.zeros((2 * P, n_frames))
    for f in range(n_frames):
        fr = cm[f]
        trough = np.zeros(n_lags, dtype=bool)
        trough[1:-1] = (fr[1:-1] < fr[:-2]) & (fr[1:-1] <= fr[2:])             # util.localmin
        trough[-1] = fr[-1] < fr[-2]
        trough[0] = fr[0] < fr[1]
        idx = np.nonzero(trough)[0]
        if idx.size:
            heights = fr[idx]
            probs = np.zeros(idx.size)
            for k in range(100):                                               # every threshold votes with its prior mass
                below = heights < thr[k + 1]
                n = int(below.sum())
                if n == 0:
                    continue
                pos = np.cumsum(below) - 1                                     # rank among the troughs below the threshold
                pmf = (1.0 - math.exp(-2.0)) * np.exp(-2.0 * pos) / (1.0 - math.exp(-2.0 * n))    # boltzmann.pmf(pos, 2, n)
                probs += np.where(below, pmf, 0.0) * beta_probs[k]
            gmin = int(np.argmin(heights))
            n_below_min = int(np.count_nonzero(~(heights[gmin] < thr[1:])))
            probs[gmin] += 0.01 * float(beta_probs[:n_below_min].sum())        # no_trough_prob
            shifts = parabolic_shifts(fr)
            for j, p in zip(idx, probs):
                if p == 0.0:
                    continue
                period = min_p + j + shifts[j]
                b = int(np.clip(np.round(12 * bps * np.log2((PYIN_SR / period) / FMIN)), 0, P))
                obs[b, f] = p                                                   # (a later trough in the same bin overwrites, as numpy's fancy assignment does)
        voiced = min(1.0, max(0.0, float(obs[:P, f].sum())))
        obs[P:, f] = (1.0 - voiced) / P
    return obs


def transition_matrix() -> np.ndarray:
    """[2 P][2 P]: kron(voicing switch, triangular pitch transition of +-50 bins) - sequence.transition_local / transition_loop."""
    _, _, bps, P = pitch_geometry()
    width = int(round(35.92 * 12 * PHOP / PYIN_SR)) * bps + 1
    hw = width // 2
    local = np.zeros((P, P))
    for i in range(P):
        for j in range(max(0, i - hw), min(P, i + hw + 1)):
            local[i, j] = (hw + 1 - abs(i - j)) / (hw + 1.0)                    # scipy 'triangle' window of odd length
        local[i] /= local[i].sum()
    return np.kron(np.array([[0.99, 0.01], [0.01, 0.99]]), local)


def viterbi(obs: np.ndarray, trans: np.ndarray, p_init: np.ndarray) -> np.ndarray:
    eps = np.finfo(np.float64).tiny
    lt, lo, li = np.log(trans + eps), np.log(obs.T + eps), np.log(p_init + eps)
    T, S = lo.shape
    val = np.zeros((T, S))
    ptr = np.zeros((T, S), dtype=np.int64)
    val[0] = lo[0] + li
    for t in range(1, T):
        cand = val[t - 1][None, :] + lt.T                                       # [to][from]
        ptr[t] = np.argmax(cand, axis=1)
        val[t] = lo[t] + cand[np.arange(S), ptr[t]]
    states = np.zeros(T, dtype=np.int64)
    states[-1] = int(np.argmax(val[-1]))
    for t in range(T - 2, -1, -1):
        states[t] = ptr[t + 1, states[t + 1]]
    return states


def pyin_f0(y16k: np.ndarray) -> np.ndarray:
    """f0 per frame, NaN where unvoiced: librosa.pyin(y, fmin=C2, fmax=C7) with every other argument at its default."""
    _, _, bps, P = pitch_geometry()
    obs = observation_probs(cmnd_frames(y16k))
    p_init = np.zeros(2 * P)
    p_init[P:] = 1.0 / P
    states = viterbi(obs, transition_matrix(), p_init)
    freqs = FMIN * 2.0 ** (np.arange(P) / (12.0 * bps))
    f0 = freqs[states % P]
    f0[states >= P] = np.nan
    return f0


# ------------------------------------------------------------------------------------------------ LPC formants
def mid_frame(y16k: np.ndarray) -> np.ndarray:
    y = np.asarray(y16k, dtype=np.float32)
    y_pre = np.append(y[0], y[1:] - np.float32(0.97) * y[:-1]).astype(np.float32)     # float32, as the reference computes it
    n = int(0.025 * SR)
    c = y_pre.shape[0] // 2
    fr = y_pre[max(0, c - n // 2): c + n // 2].astype(np.float64)
    return fr * np.hanning(fr.shape[0])


def burg_lpc(frame: np.ndarray, order: int = LPC_ORDER) -> np.ndarray:
    """librosa.lpc: Burg's method as in section III of Marple's paper (float64)."""
    eps = np.finfo(np.float64).tiny
    a = np.zeros(order + 1)
    a[0] = 1.0
    prev = a.copy()
    fwd, bwd = frame[1:].copy(), frame[:-1].copy()
    den = float(np.dot(fwd, fwd) + np.dot(bwd, bwd))
    for i in range(order):
        k = -2.0 * float(np.dot(bwd, fwd)) / (den + eps)
        prev, a = a, prev
        for j in range(1, i + 2):
            a[j] = prev[j] + k * prev[i - j + 1]
        a[0] = 1.0
        f_tmp = fwd
        fwd = fwd + k * bwd
        bwd = bwd + k * f_tmp
        den = (1.0 - k * k) * den - bwd[-1] ** 2 - fwd[0] ** 2
        fwd, bwd = fwd[1:], bwd[:-1]
    return a


def formants_from_lpc(a: np.ndarray) -> Tuple[float, float]:
    roots = np.roots(a)
    roots = roots[roots.imag > 0]
    freqs = np.sort(np.angle(roots) * (SR / (2.0 * np.pi)))
    freqs = freqs[(freqs > 90) & (freqs < SR / 4)]
    return (float(freqs[0]) if freqs.size > 0 else 0.0), (float(freqs[1]) if freqs.size > 1 else 0.0)


# ------------------------------------------------------------------------------------------------ the 30 dimensions
def handcrafted_features(audio: np.ndarray, sr: int) -> np.ndarray:
    """[mfcc mean x 13 | mfcc std x 13 | f0 mean | f0 std | F1 | F2]: elements 256 .. 285 of trainer.extract_features' vector."""
    y = resample(np.asarray(audio, dtype=np.float32).reshape(-1), sr, SR)
    m = mfcc(y)
    f0 = pyin_f0(y)
    v = f0[~np.isnan(f0)]
    f1, f2 = formants_from_lpc(burg_lpc(mid_frame(y)))
    return np.concatenate([m.mean(axis=1), m.std(axis=1), [float(v.mean()) if v.size else 0.0, float(v.std()) if v.size else 0.0, f1, f2]])
