"""Drift-classifier features (reference: validation/classifier/trainer.py:23-96) on the CPU: the oracle's MFCC
against the in-container siblings of the librosa calls (transformers.audio_utils for the mel / dB front end, scipy's DCT - the
function librosa itself calls), probabilistic YIN and the Burg LPC on signals with known answers, and the product's host half
(rho_tts_amd/features.py: trough statistics, banded Viterbi, LPC roots) against the oracle's dense restatement.
librosa is absent: parity with the reference's own numbers is UNPINNED (oracle/features.py header)."""
import numpy as np
import pytest

from oracle import features as OF
from rho_tts_amd import features as PF


def voiced(seconds=1.2, sr=24000, f0=220.0, seed=0):
    t = np.arange(int(seconds * sr)) / sr
    x = 0.3 * sum(np.sin(2 * np.pi * f0 * (k + 1) * t) / (k + 1) for k in range(5)) * (0.6 + 0.4 * np.sin(2 * np.pi * 3 * t))
    x[: int(0.1 * sr)] = 0.0
    return (x + 1e-3 * np.random.default_rng(seed).standard_normal(x.shape[0])).astype(np.float32)


def test_mfcc_matches_the_siblings_of_the_librosa_calls():
    import scipy.fft
    from transformers import audio_utils as AU
    y = OF.resample(voiced(), 24000, 16000)
    m = OF.mfcc(y)
    fb = AU.mel_filter_bank(1025, 128, 0.0, 8000.0, 16000, norm="slaney", mel_scale="slaney")
    S = AU.spectrogram(y.astype(np.float64), AU.window_function(2048, "hann", periodic=True), 2048, 512, fft_length=2048, power=2.0, center=True,
                       pad_mode="constant", mel_filters=fb, log_mel="dB", reference=1.0, min_value=1e-10, db_range=80.0, dtype=np.float64)
    want = scipy.fft.dct(S, axis=0, type=2, norm="ortho")[:13]
    assert m.shape == want.shape == (13, 1 + y.shape[0] // 512)
    assert float(np.abs(m - want).max()) < 1e-4 * float(np.abs(want).max())
    assert np.allclose(OF.mel_filterbank().T, fb, atol=1e-7)


def test_pyin_reports_the_reference_s_22050_hz_quirk():
    """The reference calls librosa.pyin without `sr` on 16-kHz audio (trainer.py:52): a 220-Hz voice is reported 22050 / 16000
    times too high.  Restated as written - the classifier was trained on exactly these numbers."""
    y = OF.resample(voiced(f0=220.0), 24000, 16000)
    f0 = OF.pyin_f0(y)
    v = f0[~np.isnan(f0)]
    assert v.size >= 0.7 * f0.size and np.isnan(f0[0])                      # the silent lead-in is unvoiced
    assert abs(float(np.median(v)) / (220.0 * 22050 / 16000) - 1.0) < 0.01   # within two 0.1-semitone bins
    assert OF.pitch_geometry() == (10, 338, 10, 601)


def test_lpc_formants_of_a_known_resonator():
    """White noise through a two-resonance all-pole filter: the Burg LPC's root angles sit at the resonances."""
    import scipy.signal
    sr = 16000
    poles = []
    for f, bw in ((700.0, 80.0), (1800.0, 120.0)):
        r = np.exp(-np.pi * bw / sr)
        poles += [r * np.exp(2j * np.pi * f / sr), r * np.exp(-2j * np.pi * f / sr)]
    a = np.real(np.poly(poles))
    y = scipy.signal.lfilter([1.0], a, np.random.default_rng(3).standard_normal(sr)).astype(np.float32)
    y *= 0.1 / np.abs(y).max()
    f1, f2 = OF.formants_from_lpc(OF.burg_lpc(OF.mid_frame(y)))
    got = PF.formants_from_lpc(OF.burg_lpc(OF.mid_frame(y)))
    assert got == (f1, f2)
    cand = np.array([f1, f2])
    assert np.abs(cand - 700.0).min() < 120.0 or np.abs(cand - 1800.0).min() < 200.0    # (order 18 on one 25-ms frame: coarse)
    # Burg against the definition: the predictor minimises forward + backward error, reflection coefficients below one
    fr = OF.mid_frame(y)
    a18 = OF.burg_lpc(fr)
    assert a18[0] == 1.0 and np.all(np.abs(np.roots(a18)) < 1.0)


@pytest.mark.parametrize("f0,seed", [(150.0, 1), (320.0, 2)])
def test_product_host_half_equals_the_oracle(f0, seed):
    """rho_tts_amd/features.py never forms the 1202 x 1202 transition matrix and vectorises the trough statistics; on the oracle's
    own difference function it must decode the same path, frame for frame."""
    y = OF.resample(voiced(seconds=0.9, f0=f0, seed=seed), 24000, 16000)
    cm = OF.cmnd_frames(y)
    want = OF.pyin_f0(y)
    got = PF.f0_from_cmnd(cm, OF.pitch_geometry()[0])
    assert got.shape == want.shape
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.allclose(got[~np.isnan(got)], want[~np.isnan(want)], rtol=0, atol=1e-9)
    P = OF.pitch_geometry()[3]
    lo = PF.observation_log_probs(cm, OF.pitch_geometry()[0], P)
    assert np.allclose(np.exp(lo) - np.finfo(np.float64).tiny, OF.observation_probs(cm).T, atol=1e-12)


def test_constants_agree():
    assert (PF.FMIN, PF.FMAX, PF.PITCH_SR, PF.LPC_ORDER) == (OF.FMIN, OF.FMAX, OF.PYIN_SR, OF.LPC_ORDER)
