"""The conditioning front-end on the GPU (rt_voice_encode / rt_model_set_voice_pcm through the C ABI) against the CPU oracle
(oracle/encoder.py - PARITY UNPINNED against qwen-tts, cross-checked against transformers' Mimi in tests/test_oracle_encoder.py).

Codes are integers.  The quantiser kernel evaluates its distances in the oracle's defined float32 order, so given the same
input vectors its argmin is the oracle's bit for bit (checked directly).  End to end the vectors themselves come out of
MFMA convolutions (float32-faithful split precision, ~1e-6 relative) rather than CPU float32, so a code may differ only where
two codebook entries are within that noise of each other: the test demands exact equality on the committed clips and, were
a near-tie ever to flip, that the oracle's two distances differ by < 1e-4 relative (nothing else is accepted)."""
import numpy as np
import pytest
import torch

from oracle import encoder as E
from rho_tts_amd import config, weights

pytestmark = pytest.mark.gpu
torch.set_num_threads(8)


@pytest.fixture(scope="module")
def ctx():
    from rho_tts_amd import _native
    c = _native.Context(0)
    yield c
    c.close()


def build(ctx, cfg, max_positions=None):
    from rho_tts_amd._native_model import NativeModel
    state = weights.synthetic_state(cfg, 789)
    nm = NativeModel(ctx, cfg, max_batch=4, max_positions=max_positions)
    nm.load_state({k: v.cuda() for k, v in state.items()})
    return nm, {k: v.float() for k, v in state.items() if k.startswith("enc.")}


def clip(cfg, n_frames, seed):
    g = np.random.default_rng(seed)
    n = n_frames * cfg.codec.total_upsample
    t = np.arange(n) / cfg.sample_rate
    x = 0.3 * np.sin(2 * np.pi * 140.0 * t) * (0.6 + 0.4 * np.sin(2 * np.pi * 3.0 * t)) + 0.05 * g.standard_normal(n)
    return x.astype(np.float32)


def check_codes(cfg, W, pcm, got, want, mid):
    """exact, or (never seen so far) a flip between two entries the oracle itself rates within 1e-4 of each other"""
    if torch.equal(got, want):
        return 0
    bad = 0
    Q = cfg.codec.num_quantizers
    sem = (mid["emb"] @ W["enc.vq.semantic.input_proj.weight"].T).numpy()
    aco = (mid["emb"] @ W["enc.vq.acoustic.input_proj.weight"].T).numpy()
    for t in range(got.shape[0]):
        if torch.equal(got[t], want[t]):
            continue
        q = int((got[t] != want[t]).nonzero()[0])                    # first differing level; later levels follow from it
        res = sem[t] if q == 0 else aco[t].copy()
        for lv in range(1, q):
            res = res - W[f"enc.vq.codebook.{lv}"].numpy()[int(want[t, lv])]
        cb = W[f"enc.vq.codebook.{q}"].numpy()
        d = ((res[None] - cb[[int(got[t, q]), int(want[t, q])]]) ** 2).sum(1)
        assert abs(d[0] - d[1]) <= 1e-4 * d[1], (t, q, d)
        bad += 1
    return bad


@pytest.mark.parametrize("preset,n_frames", [("tiny", 37), ("small", 50)])
def test_encoder_codes_and_speaker_embedding_match_oracle(ctx, preset, n_frames):
    cfg = config.PRESETS[preset]()
    nm, W = build(ctx, cfg)
    try:
        pcm = clip(cfg, n_frames, 1)
        codes, spk = nm.encode_voice(pcm)
        want, spk_o, mid = E.encode(W, cfg, pcm, return_intermediates=True)
        assert codes.shape == want.shape == (n_frames, cfg.codec.num_quantizers)
        assert check_codes(cfg, W, pcm, codes, want, mid) == 0            # exact on this clip
        assert float((spk - spk_o).abs().max()) < 1e-4 * max(1.0, float(spk_o.abs().max()))
        # trailing samples that do not fill a frame are dropped; a cap on the frames is honoured
        c2, _ = nm.encode_voice(np.concatenate([pcm, np.zeros(5, np.float32)]))
        assert torch.equal(c2, codes)
        c3, _ = nm.encode_voice(pcm, max_frames=11)
        assert torch.equal(c3, codes[:11])                                 # causal encoder: a prefix of the clip gives a prefix of the codes
        with pytest.raises(ValueError):
            nm.encode_voice(pcm[:3])
    finally:
        nm.close()


def test_set_voice_pcm_equals_encode_then_set_voice(ctx):
    """rt_model_set_voice_pcm == rt_voice_encode + rt_model_set_voice: same prefix length, same generated codes."""
    from rho_tts_amd._native_model import RtSampling
    cfg = config.tiny()
    nm, W = build(ctx, cfg)
    try:
        pcm = clip(cfg, 21, 2)
        ref_text = [5, 6, 7, 8]
        codes, spk = nm.encode_voice(pcm)
        n1 = nm.set_voice("english", None, spk, ref_text, codes)
        a = nm.generate([[10, 11, 12], [20, 21]], [6, 5], RtSampling(1, 0.9, 50, 1.0, 1.05), seed=3)
        n2, codes2 = nm.set_voice_pcm(pcm, "english", ref_text)
        b = nm.generate([[10, 11, 12], [20, 21]], [6, 5], RtSampling(1, 0.9, 50, 1.0, 1.05), seed=3)
        assert n1 == n2 and torch.equal(codes, codes2)
        assert all(torch.equal(x, y) for x, y in zip(a, b))
        # a different clip is a different voice
        nm.set_voice_pcm(clip(cfg, 21, 9), "english", ref_text)
        c = nm.generate([[10, 11, 12], [20, 21]], [6, 5], RtSampling(1, 0.9, 50, 1.0, 1.05), seed=3)
        assert not all(torch.equal(x, y) for x, y in zip(a, c))
    finally:
        nm.close()


def test_encoder_at_real_dimensions(ctx):
    """The 1.7B / 0.6B presets' encoder (64..1024-channel conv stages, strides 4/5/6/8, 512-wide 8-layer transformer, 16 x 2048
    codebooks of 256 dims): 4 s of audio, codes against the oracle."""
    cfg = config.PRESETS["0.6b"]()
    from rho_tts_amd._native_model import NativeModel
    state = weights.synthetic_state(cfg, 789, device="cuda")
    nm = NativeModel(ctx, cfg, max_batch=2, max_positions=256)
    try:
        nm.load_state(state)
        W = {k: v.float().cpu() for k, v in state.items() if k.startswith("enc.")}
        del state
        pcm = clip(cfg, 50, 3)
        codes, spk = nm.encode_voice(pcm)
        want, spk_o, mid = E.encode(W, cfg, pcm, return_intermediates=True)
        flips = check_codes(cfg, W, pcm, codes, want, mid)
        print(f"\\nreal-dim encoder: {codes.shape[0]} frames x {codes.shape[1]} codebooks, {flips} near-tie frames")
        assert flips <= 1
        assert float((spk - spk_o).abs().max()) < 1e-4 * max(1.0, float(spk_o.abs().max()))
        assert len(set(codes[:, 0].tolist())) > 10
    finally:
        nm.close()
