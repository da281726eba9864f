"""The drift-classifier feature extractor on the GPU (csrc/features.hip behind rt_features_extract, host half in
rho_tts_amd/features.py) against the CPU oracle (oracle/features.py - PARITY UNPINNED against librosa, which is absent): MFCC
statistics, the pYIN difference function, the Burg LPC, and the 30 numbers end to end, on clips at the TTS rate."""
import numpy as np
import pytest
import torch

from oracle import features as OF
from rho_tts_amd import _native
from rho_tts_amd import features as PF
from tests.test_oracle_features import voiced

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = _native.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("seconds,f0,sr", [(1.2, 220.0, 24000), (3.4, 140.0, 24000), (0.7, 300.0, 16000)])
def test_features_match_the_oracle(ctx, seconds, f0, sr):
    x = voiced(seconds, sr, f0, seed=int(f0))
    ex = PF.HandcraftedFeatures(ctx)
    try:
        stats, cmnd, lpc = ex.raw(torch.from_numpy(x).cuda(), sr)
        y = OF.resample(x, sr, 16000)
        m = OF.mfcc(y)
        want = np.concatenate([m.mean(axis=1), m.std(axis=1)])
        assert float(np.abs(stats - want).max()) < 2e-3, np.abs(stats - want).max()         # (dB-scale values up to ~600; float32 samples)
        cm = OF.cmnd_frames(y)
        assert cmnd.shape == cm.shape
        assert float(np.abs(cmnd - cm).max()) < 1e-6 * max(1.0, float(np.abs(cm).max()))
        a = OF.burg_lpc(OF.mid_frame(y))
        assert float(np.abs(lpc - a).max()) < 1e-5 * float(np.abs(a).max())
        got = ex(torch.from_numpy(x).cuda(), sr)
        ref = OF.handcrafted_features(x, sr)
        assert got.shape == ref.shape == (30,)
        assert float(np.abs(got[:26] - ref[:26]).max()) < 2e-3
        assert abs(got[26] - ref[26]) < 1e-6 * ref[26] and abs(got[27] - ref[27]) < 1e-6 * max(1.0, ref[27])    # the same Viterbi path
        assert abs(got[28] - ref[28]) < 0.5 and abs(got[29] - ref[29]) < 0.5                                     # Hz
        assert abs(got[26] / (f0 * 22050 / 16000) - 1.0) < 0.02                                                   # (the reference's sr quirk)
    finally:
        ex.close()


def test_scorer_hook_runs_on_the_waveform_in_hbm(ctx):
    ex = PF.HandcraftedFeatures(ctx)
    try:
        seen = []

        def classifier(f):
            seen.append(f.shape)
            return 0.25
        score = PF.make_drift_scorer(ex, classifier, embed=lambda a, sr: np.zeros(256))
        x = torch.from_numpy(voiced(1.0, 24000, 180.0)).cuda()
        assert score(x, 24000) == 0.25 and seen == [(286,)]                   # the reference's layout: 256 + 30
        with pytest.raises(ValueError):
            ex(torch.zeros(1), 24000)
    finally:
        ex.close()


@pytest.mark.parametrize("kind", ["silence", "short", "noise"])
def test_feature_edge_cases(ctx, kind):
    """All-zero audio (no voiced frame: F0 statistics 0, LPC = [1, 0, ...], no formant), a clip shorter than one analysis frame,
    and white noise (pYIN mostly unvoiced): the extractor and the oracle agree, nothing is NaN."""
    sr = 24000
    if kind == "silence":
        x = np.zeros(int(0.6 * sr), dtype=np.float32)
    elif kind == "short":
        x = voiced(0.05, sr, 200.0, seed=4)[-1000:]                     # 1000 samples at 24 kHz = 667 at 16 kHz: one padded frame + a partial one
    else:
        x = (0.1 * np.random.default_rng(5).standard_normal(int(0.5 * sr))).astype(np.float32)
    ex = PF.HandcraftedFeatures(ctx)
    try:
        got = ex(torch.from_numpy(x).cuda(), sr)
        ref = OF.handcrafted_features(x, sr)
        assert got.shape == (30,) and np.all(np.isfinite(got)) and np.all(np.isfinite(ref))
        assert float(np.abs(got[:26] - ref[:26]).max()) < 5e-3
        assert abs(got[26] - ref[26]) <= 1e-6 * max(1.0, abs(ref[26])) and abs(got[27] - ref[27]) <= 1e-6 * max(1.0, abs(ref[27]))
        assert abs(got[28] - ref[28]) < 1.0 and abs(got[29] - ref[29]) < 1.0
        if kind == "silence":
            assert got[26] == 0.0 and got[27] == 0.0 and got[28] == 0.0 and got[29] == 0.0
    finally:
        ex.close()
