"""Host-side pieces of the speech-to-text path against the reference's own dependency (transformers' Whisper; reference:
validation/stt/stt_validator.py:85-107) - CPU only: the mel filter bank and window the HIP front-end is fed, the resampler
definition, the oracle's greedy rule, and the committed fixture (tests/golden/stt_golden.npz, make_stt_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import whisper as OW
from rho_tts_amd import stt as S

HERE = os.path.dirname(os.path.abspath(__file__))


def clip(seconds, sr, seed):
    g = np.random.default_rng(seed)
    t = np.arange(int(seconds * sr)) / sr
    f0 = 110.0 + 40.0 * seed
    x = sum((0.3 / (k + 1)) * np.sin(2 * np.pi * f0 * (k + 1) * t + k) for k in range(6)) * (0.55 + 0.45 * np.sin(2 * np.pi * (2.0 + seed) * t))
    return (x + 0.01 * g.standard_normal(t.shape[0])).astype(np.float32)


def test_mel_filters_and_window_equal_the_feature_extractors():
    from transformers.audio_utils import mel_filter_bank, window_function
    for n_fft, n_mels in ((400, 80), (400, 16), (512, 128)):
        want = mel_filter_bank(num_frequency_bins=1 + n_fft // 2, num_mel_filters=n_mels, min_frequency=0.0, max_frequency=8000.0, sampling_rate=16000,
                               norm="slaney", mel_scale="slaney")
        got = S.mel_filters(n_fft, n_mels, 16000)
        assert got.shape == want.shape and float(np.abs(got - want.astype(np.float32)).max()) < 1e-9
        assert float(np.abs(S.hann_window(n_fft) - window_function(n_fft, "hann").astype(np.float32)).max()) < 3e-7
        assert np.array_equal(S.hann_window(n_fft), torch.hann_window(n_fft).numpy())       # what the torch path of the extractor multiplies by


def test_resampler_definition():
    """24 kHz -> 16 kHz: a tone below the new Nyquist survives with its amplitude and phase, a tone above it is removed; the
    length is ceil(n * 2 / 3); identical rates are the identity.  (Parity unpinned: the reference's pipeline resamples with ffmpeg.)"""
    sr_in, sr_out, n = 24000, 16000, 24000
    t = np.arange(n) / sr_in
    y = OW.resample((0.5 * np.sin(2 * np.pi * 440.0 * t)).astype(np.float32), sr_in, sr_out)
    assert y.shape[0] == (n * 2 + 2) // 3
    to = np.arange(y.shape[0]) / sr_out
    mid = slice(200, -200)
    assert float(np.abs(y[mid] - 0.5 * np.sin(2 * np.pi * 440.0 * to[mid])).max()) < 2e-3
    hi = OW.resample((0.5 * np.sin(2 * np.pi * 10500.0 * t)).astype(np.float32), sr_in, sr_out)
    assert float(np.abs(hi[mid]).max()) < 5e-3
    x = clip(0.5, 16000, 1)
    assert np.array_equal(OW.resample(x, 16000, 16000), x)
    from scipy.signal import resample_poly
    z = clip(1.0, sr_in, 2)
    assert float(np.abs(OW.resample(z, sr_in, sr_out)[mid] - resample_poly(z.astype(np.float64), 2, 3)[mid]).max()) < 0.02


def test_oracle_matches_the_committed_fixture():
    """The oracle (transformers' feature extractor + model, live) against arrays written by tests/golden/make_stt_golden.py in the
    build container: a transformers upgrade that changed either would show here, not as a mysterious GPU mismatch."""
    g = np.load(os.path.join(HERE, "golden", "stt_golden.npz"))
    cfg = S.tiny_test_config()
    x16 = OW.resample(clip(1.3, 24000, 3), 24000, 16000)
    assert np.array_equal(x16, g["tiny_pcm16k"])
    mel = OW.log_mel(cfg, x16)
    assert float((mel - torch.from_numpy(g["tiny_log_mel"])).abs().max()) < 1e-5
    model = OW.build(cfg, S.synthetic_state(cfg, 789))
    ids, first = OW.greedy(model, cfg, mel)
    assert ids == g["tiny_ids"].tolist()
    assert float((first - torch.from_numpy(g["tiny_first_logits"])).abs().max()) < 1e-4
    assert float((OW.encode(model, mel) - torch.from_numpy(g["tiny_enc"])).abs().max()) < 1e-4
    # the decoding rule: nothing at or above suppress_from but end-of-sequence, none of begin_suppress first
    assert all(t < cfg.suppress_from for t in ids) and (not ids or ids[0] not in cfg.begin_suppress)
    full = np.load(os.path.join(HERE, "golden", "stt_golden.npz"))
    c2 = S.SttConfig()
    mel2 = OW.log_mel(c2, OW.resample(clip(2.0, 24000, 4), 24000, 16000))
    assert mel2.shape == (80, 3000) and float((mel2[:, :400] - torch.from_numpy(full["w_log_mel_head"])).abs().max()) < 1e-5


def test_to_native_layouts():
    """The re-layout the library is fed: fused q/k/v with a zero k bias, the stride-2 conv as [0 | W0 | W1 | W2]."""
    cfg = S.tiny_test_config()
    st = S.synthetic_state(cfg, 789)
    nat = S.to_native(st, cfg)
    D = cfg.d_model
    assert nat["enc.l0.wqkv"].shape == (3 * D, D) and torch.equal(nat["enc.l0.wqkv"][D:2 * D], st["model.encoder.layers.0.self_attn.k_proj.weight"])
    assert float(nat["enc.l0.bqkv"][D:2 * D].abs().max()) == 0.0 and float(nat["dec.l1.cbkv"][:D].abs().max()) == 0.0
    w2 = st["model.encoder.conv2.weight"]
    c2 = nat["enc.conv2"]
    assert c2.shape == (D, 4 * D) and float(c2[:, :D].float().abs().max()) == 0.0 and torch.equal(c2[:, D:2 * D], w2[:, :, 0]) and torch.equal(c2[:, 3 * D:], w2[:, :, 2])
    assert nat["enc.conv1"].shape == (D, 3 * cfg.n_mels) and torch.equal(nat["enc.conv1"][:, cfg.n_mels:2 * cfg.n_mels], st["model.encoder.conv1.weight"][:, :, 1])
    # a strided conv by the 2-tap GEMM layout equals torch's conv1d
    x = torch.randn(1, D, 2 * cfg.n_ctx)
    want = torch.nn.functional.conv1d(x, w2.float(), stride=2, padding=1)[0].t()
    rows = x[0].t().reshape(cfg.n_ctx, 2 * D)
    prev = torch.cat([torch.zeros(1, 2 * D), rows[:-1]])
    got = torch.cat([prev, rows], dim=1) @ c2.float().t()
    assert float((got - want).abs().max()) < 1e-4
    assert len(nat) == len(set(nat)) and all(t.dtype in (torch.bfloat16, torch.float32) for t in nat.values())


def write_sidecar_configs(d, cfg):
    """What a Whisper checkpoint directory carries beside config.json (values of ``cfg``)."""
    import json
    with open(os.path.join(d, "generation_config.json"), "w") as f:
        json.dump({"eos_token_id": cfg.eos_id, "decoder_start_token_id": int(cfg.prefix[0]), "max_new_tokens": cfg.max_new_tokens,
                   "forced_decoder_ids": [[i + 1, int(t)] for i, t in enumerate(cfg.prefix[1:])], "begin_suppress_tokens": list(cfg.begin_suppress)}, f)
    with open(os.path.join(d, "preprocessor_config.json"), "w") as f:
        json.dump({"chunk_length": cfg.chunk_seconds, "feature_size": cfg.n_mels, "hop_length": cfg.hop, "n_fft": cfg.n_fft,
                   "sampling_rate": cfg.sample_rate}, f)


def test_checkpoint_directory_round_trip(tmp_path):
    """The real-checkpoint path of the speech-to-text model, exercised offline: transformers' own `save_pretrained` layout
    (model.safetensors with its tensor names, config.json) is read back by `stt.load_checkpoint` / `SttConfig.from_hf`."""
    import json
    cfg = S.tiny_test_config()
    state = S.synthetic_state(cfg, 789)
    model = OW.build(cfg, state)
    d = str(tmp_path / "whisper-test")
    model.save_pretrained(d, safe_serialization=True)
    js = json.load(open(os.path.join(d, "config.json")))
    c2 = S.SttConfig.from_hf(js)
    assert (c2.d_model, c2.heads, c2.ffn, c2.enc_layers, c2.dec_layers, c2.n_mels, c2.n_ctx, c2.n_text_ctx, c2.vocab) == \
           (cfg.d_model, cfg.heads, cfg.ffn, cfg.enc_layers, cfg.dec_layers, cfg.n_mels, cfg.n_ctx, cfg.n_text_ctx, cfg.vocab)
    assert c2.eos_id == cfg.eos_id and c2.chunk_seconds == cfg.chunk_seconds and c2.prefix[0] == cfg.prefix[0]
    assert all(t < cfg.vocab for t in c2.begin_suppress)
    write_sidecar_configs(d, cfg)                                  # generation_config.json + preprocessor_config.json
    assert S.SttConfig.from_dir(d) == cfg
    whisper_tiny = {"d_model": 384, "encoder_attention_heads": 6, "encoder_ffn_dim": 1536, "encoder_layers": 4, "decoder_layers": 4,
                    "num_mel_bins": 80, "max_source_positions": 1500, "max_target_positions": 448, "vocab_size": 51865, "eos_token_id": 50257,
                    "decoder_start_token_id": 50258, "forced_decoder_ids": [[1, 50259], [2, 50359], [3, 50363]], "begin_suppress_tokens": [220, 50257]}
    assert S.SttConfig.from_hf(whisper_tiny) == S.SttConfig()      # the published whisper-tiny config.json keys -> the defaults
    named = dict(whisper_tiny, forced_decoder_ids=None)
    gen = {"lang_to_id": {"<|en|>": 50259, "<|de|>": 50261}, "task_to_id": {"transcribe": 50359, "translate": 50358}, "no_timestamps_token_id": 50363}
    assert S.SttConfig.from_hf(named, gen).prefix == S.SttConfig().prefix
    # the generation_config.json Whisper checkpoints SHIP has `forced_decoder_ids: [[1, null], [2, 50359]]` (language
    # decided at generation time) beside the named ids: <|en|> and <|notimestamps|> must not be dropped
    shipped = dict(gen, forced_decoder_ids=[[1, None], [2, 50359]], suppress_tokens=[1, 2, 7, 50358, 50257, 99999],
                   begin_suppress_tokens=[220, 50257], max_length=448)
    c3 = S.SttConfig.from_hf(named, shipped)
    assert c3.prefix == (50258, 50259, 50359, 50363)
    assert c3.suppress_tokens == (1, 2, 7, 50358)                  # in range, end-of-sequence never suppressed
    assert S.SttConfig.from_hf(dict(whisper_tiny, forced_decoder_ids=[[1, None], [2, 50359]])).prefix == (50258, 50259, 50359, 50363)
    # a German default in the checkpoint's forced ids does not override the reference's language="en"
    assert S.SttConfig.from_hf(named, dict(gen, forced_decoder_ids=[[1, 50261], [2, 50359], [3, 50363]])).prefix == (50258, 50259, 50359, 50363)
    # a vocabulary this build knows nothing about: only what the files say
    other = {"decoder_start_token_id": 900, "vocab_size": 1000, "eos_token_id": 899, "forced_decoder_ids": [[1, None], [2, 905]]}
    assert S.SttConfig.from_hf(other).prefix == (900, 905)
    back = S.load_checkpoint(cfg, d)
    assert set(back) == set(state)
    for k in state:
        assert torch.equal(back[k].float(), state[k].float()), k
    with pytest.raises(ValueError, match="checkpoint shape|missing"):
        S.load_checkpoint(S.SttConfig(), d)                        # another architecture: shapes / tensors do not fit
