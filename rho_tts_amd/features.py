"""Hand-crafted drift-classifier features on the GPU.

The reference's accent-drift classifier scores a 286-dimensional vector per segment (validation/classifier/trainer.py:23-68):
resemblyzer's 256-d speaker embedding + 13 MFCC means + 13 MFCC standard deviations + F0 mean / std (librosa.pyin) + the first two
LPC formants, all computed by librosa on a temporary WAV (base_tts.py:821-830).  This module produces the 30 hand-crafted
dimensions from the waveform in HBM: the per-sample work runs in csrc/features.hip behind ``rt_features_extract`` (resampler,
MFCC, the pYIN difference function, Burg LPC); what is left for the host is arithmetic on a few kilobytes - pYIN's trough
statistics and Viterbi pass over the [frames][329] difference function, and the roots of one degree-18 polynomial.

The 256-d embedding is NOT produced: resemblyzer's network is a pretrained checkpoint (no weights offline), and the classifier
itself is a pickled scikit-learn model this build will not load.  ``make_drift_scorer`` therefore takes the classifier as a
callable on the 30 dimensions (or on whatever vector the caller assembles around them).

Definitions: librosa 0.10's defaults as the reference calls them, restated (oracle/features.py says which) - including the
reference's own quirk of leaving ``librosa.pyin``'s default ``sr=22050`` in place for 16-kHz audio, so that every F0 is
22050 / 16000 times the acoustic one.  Parity with librosa itself is UNPINNED (not installable here).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Callable, Optional

import numpy as np
import torch

from . import _native

SR = 16000
PITCH_SR = 22050                                   # what librosa.pyin assumes when `sr` is not passed (trainer.py:52)
FMIN = 440.0 * 2.0 ** ((36 - 69) / 12.0)           # note_to_hz('C2')
FMAX = 440.0 * 2.0 ** ((96 - 69) / 12.0)           # note_to_hz('C7')
HOP = 512
BINS_PER_SEMITONE = 10                             # resolution 0.1
LPC_ORDER = max(12, SR // 1000 + 2)
N_THRESHOLDS, NO_TROUGH_PROB, SWITCH_PROB, MAX_TRANSITION_RATE = 100, 0.01, 0.01, 35.92

_DECLARED = False


def _declare(lib: C.CDLL) -> None:
    global _DECLARED
    if _DECLARED:
        return
    vp, i32, i64, pd = C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_double)
    lib.rt_features_create.argtypes = [vp, C.POINTER(vp)]
    lib.rt_features_destroy.argtypes = [vp]
    lib.rt_features_geometry.argtypes = [i32, C.c_double, C.c_double, C.POINTER(i32), C.POINTER(i32)]
    lib.rt_features_extract.argtypes = [vp, vp, i64, i32, i32, i32, i32, pd, C.POINTER(i32), pd, i32, C.POINTER(i32), pd]
    _DECLARED = True


# ------------------------------------------------------------------------------------------------ pYIN, the host half
def _beta_probs() -> np.ndarray:
    """Prior mass of each threshold bin under Beta(2, 18): I_x(2, 18) = 1 - (1 - x)^18 (1 + 18 x)."""
    x = np.linspace(0.0, 1.0, N_THRESHOLDS + 1)
    return np.diff(1.0 - (1.0 - x) ** 18 * (1.0 + 18.0 * x))


def observation_log_probs(cmnd: np.ndarray, min_period: int, n_bins: int) -> np.ndarray:
    """log of pyin's observation matrix, [frames][2 n_bins] (voiced pitch bins, then the unvoiced states), from the
    cumulative-mean-normalised difference function [frames][lags]: troughs, thresholds below each trough, Boltzmann(2) prior over
    the troughs under a threshold, Beta(2, 18) prior over the thresholds, parabolic refinement of the trough's lag."""
    T, n_lags = cmnd.shape
    thr = np.linspace(0.0, 1.0, N_THRESHOLDS + 1)[1:]
    beta = _beta_probs()
    tiny = np.finfo(np.float64).tiny
    obs = np.zeros((T, 2 * n_bins))
    # parabolic shifts for every lag of every frame
    a = (cmnd[:, :-2] + cmnd[:, 2:] - 2.0 * cmnd[:, 1:-1]) / 2.0
    b = (cmnd[:, 2:] - cmnd[:, :-2]) / 2.0
    shifts = np.zeros_like(cmnd)
    shifts[:, 1:-1] = -b / (2.0 * a + tiny)
    shifts[np.abs(shifts) > 1.0] = 0.0
    trough = np.zeros_like(cmnd, dtype=bool)
    trough[:, 1:-1] = (cmnd[:, 1:-1] < cmnd[:, :-2]) & (cmnd[:, 1:-1] <= cmnd[:, 2:])
    trough[:, -1] = cmnd[:, -1] < cmnd[:, -2]
    trough[:, 0] = cmnd[:, 0] < cmnd[:, 1]
    e2 = math.exp(-2.0)
    for t in range(T):
        idx = np.flatnonzero(trough[t])
        if idx.size:
            h = cmnd[t, idx]
            below = h[:, None] < thr[None, :]                                   # [troughs][thresholds]
            n = below.sum(axis=0)
            pos = np.cumsum(below, axis=0) - 1
            with np.errstate(divide="ignore", invalid="ignore"):
                prior = np.where(below, (1.0 - e2) * np.exp(-2.0 * pos) / (1.0 - np.exp(-2.0 * np.maximum(n, 1)))[None, :], 0.0)
            probs = prior @ beta
            g = int(np.argmin(h))
            probs[g] += NO_TROUGH_PROB * float(beta[: int(np.count_nonzero(~below[g]))].sum())
            keep = probs != 0.0
            period = min_period + idx[keep] + shifts[t, idx[keep]]
            bins = np.clip(np.round(12 * BINS_PER_SEMITONE * np.log2((PITCH_SR / period) / FMIN)), 0, n_bins).astype(np.int64)
            obs[t, bins] = probs[keep]                                            # (ascending lag: a later trough in the same bin wins)
        voiced = min(1.0, max(0.0, float(obs[t, :n_bins].sum())))
        obs[t, n_bins:] = (1.0 - voiced) / n_bins
    return np.log(obs + tiny)


def viterbi_banded(log_obs: np.ndarray, n_bins: int) -> np.ndarray:
    """Most likely state path of pyin's HMM: 2 x n_bins states, pitch transitions a triangle over +-hw bins (row-normalised at the
    edges), voicing kept with probability 0.99.  The transition matrix is never formed: the best predecessor of (voicing v, bin j)
    is searched over the 2 x (2 hw + 1) states that can reach it, lowest state index first on ties (np.argmax's rule on the dense
    matrix, which oracle/features.py builds)."""
    T = log_obs.shape[0]
    hw = (int(round(MAX_TRANSITION_RATE * 12 * HOP / PITCH_SR)) * BINS_PER_SEMITONE + 1) // 2
    tiny = np.finfo(np.float64).tiny
    j = np.arange(n_bins)
    # local[i][j] = (hw + 1 - |i - j|) / (hw + 1) / rowsum(i) for |i - j| <= hw
    lo_i, hi_i = np.maximum(0, j - hw), np.minimum(n_bins - 1, j + hw)
    tri = lambda d: (hw + 1.0 - np.abs(d)) / (hw + 1.0)                           # noqa: E731
    rowsum = np.array([tri(np.arange(lo_i[i], hi_i[i] + 1) - i).sum() for i in range(n_bins)])
    offs = np.arange(-hw, hw + 1)
    src = j[None, :] + offs[:, None]                                             # [offset][to] = from-bin
    ok = (src >= 0) & (src < n_bins)
    srcc = np.clip(src, 0, n_bins - 1)
    w_local = np.where(ok, tri(offs)[:, None] / rowsum[srcc], 0.0)               # local[from][to]
    S = 2 * n_bins
    val = np.full(S, -np.inf)
    ptr = np.zeros((T, S), dtype=np.int64)
    p_init = np.zeros(S)
    p_init[n_bins:] = 1.0 / n_bins
    val = log_obs[0] + np.log(p_init + tiny)
    neg = np.log(tiny)                                                            # log(0 + tiny): what the dense matrix holds outside the band
    for t in range(1, T):
        new = np.empty(S)
        for v_to in (0, 1):
            best = np.full(n_bins, -np.inf)
            arg = np.zeros(n_bins, dtype=np.int64)
            for v_from in (0, 1):                                                 # ascending state index: voiced block first
                lt = np.log(w_local * (1.0 - SWITCH_PROB if v_from == v_to else SWITCH_PROB) + tiny)
                cand = np.where(ok, val[v_from * n_bins + srcc] + lt, -np.inf)    # [offset][to]
                k = np.argmax(cand, axis=0)                                       # first maximum = lowest from-bin
                c = cand[k, j]
                take = c > best
                arg = np.where(take, v_from * n_bins + srcc[k, j], arg)
                best = np.where(take, c, best)
            # states outside the band reach (v_to, j) with log(tiny): they win only if everything inside is worse
            out_best = float(val.max()) + neg
            if np.any(out_best > best):
                dense_from = int(np.argmax(val))
                arg = np.where(out_best > best, dense_from, arg)
                best = np.maximum(best, out_best)
            new[v_to * n_bins: (v_to + 1) * n_bins] = log_obs[t, v_to * n_bins: (v_to + 1) * n_bins] + best
            ptr[t, v_to * n_bins: (v_to + 1) * n_bins] = arg
        val = new
    states = np.zeros(T, dtype=np.int64)
    states[-1] = int(np.argmax(val))
    for t in range(T - 2, -1, -1):
        states[t] = ptr[t + 1, states[t + 1]]
    return states


def f0_from_cmnd(cmnd: np.ndarray, min_period: int) -> np.ndarray:
    """F0 per frame (NaN = unvoiced) from the difference function: the back half of librosa.pyin."""
    n_bins = int(math.floor(12 * BINS_PER_SEMITONE * math.log2(FMAX / FMIN))) + 1
    states = viterbi_banded(observation_log_probs(np.asarray(cmnd, dtype=np.float64), min_period, n_bins), n_bins)
    f0 = FMIN * 2.0 ** ((states % n_bins) / (12.0 * BINS_PER_SEMITONE))
    f0[states >= n_bins] = np.nan
    return f0


def formants_from_lpc(a: np.ndarray):
    """(F1, F2): the two lowest root angles of the LPC polynomial between 90 Hz and sr / 4 (trainer.py:88-96)."""
    roots = np.roots(np.asarray(a, dtype=np.float64))
    roots = roots[roots.imag > 0]
    freqs = np.sort(np.angle(roots) * (SR / (2.0 * np.pi)))
    freqs = freqs[(freqs > 90) & (freqs < SR / 4)]
    return (float(freqs[0]) if freqs.size > 0 else 0.0), (float(freqs[1]) if freqs.size > 1 else 0.0)


# ------------------------------------------------------------------------------------------------ the extractor
class HandcraftedFeatures:
    """One ``rt_features`` on the context (GPU, stream) of the engine whose output it scores."""

    def __init__(self, ctx: "_native.Context"):
        self.ctx, self.lib = ctx, ctx.lib
        _declare(self.lib)
        h = C.c_void_p()
        ctx.check(self.lib.rt_features_create(ctx.handle, C.byref(h)), "rt_features_create")
        self.handle = h
        lo, hi = C.c_int32(), C.c_int32()
        if self.lib.rt_features_geometry(PITCH_SR, FMIN, FMAX, C.byref(lo), C.byref(hi)) != 0:
            raise ValueError("rt_features_geometry refused the pitch range")
        self.min_period, self.max_period = int(lo.value), int(hi.value)

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.rt_features_destroy(self.handle)
            self.handle = None

    def raw(self, audio, sample_rate: int):
        """(mfcc mean+std [26], cmnd [frames][lags], lpc [order + 1]) - what comes off the GPU."""
        x = audio if isinstance(audio, torch.Tensor) else torch.as_tensor(np.asarray(audio, dtype=np.float32))
        x = x.detach().to(device=f"cuda:{self.ctx.device_ordinal}", dtype=torch.float32).reshape(-1).contiguous()
        if x.numel() < 2:
            raise ValueError("feature extraction needs at least two samples")
        torch.cuda.current_stream(x.device).synchronize()
        n16 = -(-int(x.numel()) * SR // int(sample_rate)) if int(sample_rate) != SR else int(x.numel())
        cap = 2 + n16 // HOP
        n_lags = self.max_period - self.min_period + 1
        stats = (C.c_double * 26)()
        cmnd = np.zeros((cap, n_lags), dtype=np.float64)
        lpc = (C.c_double * (LPC_ORDER + 1))()
        nm, npf = C.c_int32(), C.c_int32()
        self.ctx.check(self.lib.rt_features_extract(self.handle, C.c_void_p(x.data_ptr()), x.numel(), int(sample_rate), self.min_period, self.max_period,
                                                    LPC_ORDER, stats, C.byref(nm), cmnd.ctypes.data_as(C.POINTER(C.c_double)), cap, C.byref(npf), lpc),
                       "rt_features_extract")
        return np.array(stats, dtype=np.float64), cmnd[: npf.value], np.array(lpc, dtype=np.float64)

    def __call__(self, audio, sample_rate: int) -> np.ndarray:
        """[13 MFCC means | 13 MFCC stds | F0 mean | F0 std | F1 | F2] = elements 256..285 of the reference's feature vector."""
        stats, cmnd, lpc = self.raw(audio, sample_rate)
        f0 = f0_from_cmnd(cmnd, self.min_period)
        v = f0[~np.isnan(f0)]
        f1, f2 = formants_from_lpc(lpc)
        return np.concatenate([stats, [float(v.mean()) if v.size else 0.0, float(v.std()) if v.size else 0.0, f1, f2]])


def make_drift_scorer(extractor: HandcraftedFeatures, classifier: Callable[[np.ndarray], float],
                      embed: Optional[Callable[[torch.Tensor, int], np.ndarray]] = None) -> Callable[[torch.Tensor, int], float]:
    """A ``drift_scorer`` hook for the provider (provider.BatchedPipeline): ``(audio tensor, sample_rate) -> probability``.
    ``classifier`` maps the feature vector to the accent-drift probability (the reference's is a pickled scikit-learn model,
    ``predict_proba(...)[0, 1]``); ``embed`` optionally supplies the speaker-embedding dimensions that precede the hand-crafted ones
    in the reference's layout."""
    def score(audio: torch.Tensor, sample_rate: int) -> float:
        f = extractor(audio, sample_rate)
        if embed is not None:
            f = np.concatenate([np.asarray(embed(audio, sample_rate), dtype=np.float64).reshape(-1), f])
        return float(classifier(f))
    return score
