"""On-GPU speech-to-text (reference: validation/stt/stt_validator.py:42-148 and the temporary-WAV round trip of
base_tts.py:821-830) against transformers' Whisper - the reference's own fallback dependency (:85-107), built from a config with
the same seeded weights: log-mel features <= 1e-4, encoder states at float32 accuracy, logits behind the forced prefix, and the
greedy token ids EQUAL, on synthetic clips at the TTS sample rate; then the validation loop of the provider running on the
device with no temporary file."""
import numpy as np
import pytest
import torch

from oracle import whisper as OW
from rho_tts_amd import _native
from rho_tts_amd import stt as S
from tests.test_oracle_whisper import clip

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = _native.Context(0)
    yield c
    c.close()


def check_model(ctx, cfg, clips, sr, max_new):
    state = S.synthetic_state(cfg, 789)
    nat = S.NativeSTT(ctx, cfg, {k: v.cuda() for k, v in state.items()})
    model = OW.build(cfg, state)
    try:
        for x in clips:
            x16 = OW.resample(x, sr, cfg.sample_rate)
            mel_o = OW.log_mel(cfg, x16)
            mel_g = nat.log_mel(x, sr).cpu()
            assert mel_g.shape == mel_o.shape
            err = float((mel_g - mel_o).abs().max())
            assert err < 1e-4, ("log-mel", err)
            enc_o = OW.encode(model, mel_o)
            enc_g = nat.encode(x, sr).cpu()
            e_rms = float((enc_g - enc_o).pow(2).mean().sqrt() / enc_o.pow(2).mean().sqrt())
            assert e_rms < 2e-4 and float((enc_g - enc_o).abs().max()) < 5e-3 * float(enc_o.abs().max()), ("encoder", e_rms)
            _, first_o = OW.greedy(model, cfg, mel_o, max_new)
            ids_o = OW.transcribe_windows(model, cfg, x, sr, max_new)       # (clips longer than a chunk: window by window)
            ids_g, first_g = nat.transcribe_ids(x, sr, max_tokens=max_new, first_logits=True)
            l_err = float((first_g.cpu() - first_o).abs().max()) / float(first_o.std())
            assert l_err < 2e-3, ("logits", l_err)
            assert ids_g == ids_o, (ids_g, ids_o)
            assert len(ids_o) > 0
    finally:
        nat.close()


def test_small_model_every_stage(ctx):
    """A two-layer model with 2-s chunks: clips shorter and longer than a chunk, silence, a click at the chunk's end."""
    cfg = S.tiny_test_config()
    sr = 24000
    z = np.zeros(int(0.7 * sr), dtype=np.float32)
    edge = clip(2.0, sr, 5).copy()
    edge[-3:] = 0.9
    check_model(ctx, cfg, [clip(1.3, sr, 3), clip(0.4, sr, 1), clip(2.6, sr, 2), z, edge, clip(1.0, 16000, 4)[: 16000]], sr, 10)


def test_whisper_tiny_dimensions_eight_clips(ctx):
    """d_model 384, 4 + 4 layers, 6 heads, 80 mels, 1500 positions, vocabulary 51865: eight synthetic clips of 0.8 ... 9 s at the
    24-kHz TTS rate - log-mel <= 1e-4, encoder states, first logits, and every greedy token id equal to transformers' Whisper."""
    cfg = S.SttConfig(max_new_tokens=16)
    sr = 24000
    clips = [clip(d, sr, k) for k, d in enumerate((0.8, 1.7, 2.5, 3.3, 4.1, 5.0, 6.4, 9.0))]
    check_model(ctx, cfg, clips, sr, 12)


def test_long_audio_is_transcribed_window_by_window(ctx):
    """a segment longer than one chunk used to be cut to the chunk silently.  5.3 s on 2-s chunks = three windows, each
    decoded behind the forced prefix: the ids are the concatenation of the oracle's per-window greedy ids, and a `suppress_tokens`
    list (generation_config.json) removes exactly those ids on both sides."""
    import dataclasses
    cfg = S.tiny_test_config()
    sr = 24000
    x = clip(5.3, sr, 11)
    state = S.synthetic_state(cfg, 789)
    model = OW.build(cfg, state)
    nat = S.NativeSTT(ctx, cfg, {k: v.cuda() for k, v in state.items()})
    try:
        ids = nat.transcribe_ids(x, sr)
        want = OW.transcribe_windows(model, cfg, x, sr)
        assert ids == want and len(ids) > cfg.max_new_tokens                 # more than one window's worth
        one = nat.transcribe_ids(x[: 2 * sr], sr)
        assert ids[: len(one)] == one                                         # the first window alone gives the same head
        assert nat.transcribe_ids(x, sr, max_tokens=15) == want[:15]           # the caller's cap spans the windows
    finally:
        nat.close()
    banned = tuple(sorted(set(want)))[:6]                                      # ids the model likes: now never produced
    cfg2 = dataclasses.replace(cfg, suppress_tokens=banned)
    nat = S.NativeSTT(ctx, cfg2, {k: v.cuda() for k, v in state.items()})
    try:
        ids2 = nat.transcribe_ids(x, sr)
        assert ids2 == OW.transcribe_windows(model, cfg2, x, sr) and not set(ids2) & set(banned) and ids2 != want
    finally:
        nat.close()


def test_native_rate_input_skips_the_resampler(ctx):
    cfg = S.tiny_test_config()
    state = S.synthetic_state(cfg, 789)
    nat = S.NativeSTT(ctx, cfg, {k: v.cuda() for k, v in state.items()})
    try:
        x = clip(1.1, 16000, 7)
        assert float((nat.log_mel(x, 16000).cpu() - OW.log_mel(cfg, x)).abs().max()) < 1e-4
        with pytest.raises(ValueError):
            nat.log_mel(x, 10)
    finally:
        nat.close()


def test_validation_runs_on_the_device_without_a_file(monkeypatch):
    """max_iterations > 1 with both tensor-level hooks set: drift score and transcription are computed from the waveform in HBM -
    the reference's temporary WAV (base_tts.py:821-830) is never written - and the text score reaches the metadata."""
    import tempfile
    from rho_tts_amd.provider import MI355XQwenTTS
    t = MI355XQwenTTS(device="cuda", speaker="Vivian", model_path="x/CustomVoice-small", batch_size=4, max_iterations=2)
    try:
        eng = t._load_engine()
        tr = S.WhisperTranscriber(eng.ctx, synthetic=True, cfg=S.tiny_test_config())
        seen = []

        def transcribe(audio, sr):
            assert audio.is_cuda
            text = tr(audio, sr)
            seen.append(text)
            return text
        t.transcriber = transcribe
        t.drift_scorer = lambda audio, sr: 0.01
        t.text_similarity_threshold = 0.0                                    # random weights transcribe nothing: every score is accepted

        def no_files(*a, **k):
            raise AssertionError("validation wrote a temporary file")
        monkeypatch.setattr(tempfile, "mkstemp", no_files)
        res = t.generate(["A short sentence to validate.", "And another one."])
        assert res is not None and all(r is not None and r.audio.numel() > 0 for r in res)
        assert len(seen) == 2 and all(s and s.startswith("<") for s in seen)
        assert all(r.text_similarity is not None and r.drift_prob == 0.01 for r in res)
        with pytest.raises(ValueError):
            S.WhisperTranscriber(eng.ctx, model_dir="/nonexistent/whisper-tiny")         # never a silent synthetic fallback
        tr.close()
    finally:
        t.close()


def test_transcriber_from_a_checkpoint_directory(ctx, tmp_path):
    """`WhisperTranscriber(model_dir=...)`: transformers' save_pretrained layout + a tokenizer.json -> the same ids as the seeded
    state it was written from, decoded to text through the tokenizer (the path a real whisper-tiny directory takes)."""
    from tokenizers import Tokenizer
    from tokenizers.models import WordLevel
    from tokenizers.pre_tokenizers import Whitespace
    cfg = S.tiny_test_config()
    state = S.synthetic_state(cfg, 789)
    d = str(tmp_path / "whisper-test")
    OW.build(cfg, state).save_pretrained(d, safe_serialization=True)
    tok = Tokenizer(WordLevel({f"w{i}": i for i in range(cfg.vocab)}, unk_token="w0"))
    tok.pre_tokenizer = Whitespace()
    tok.save(d + "/tokenizer.json")
    from tests.test_oracle_whisper import write_sidecar_configs
    write_sidecar_configs(d, cfg)
    x = clip(1.2, 24000, 8)
    tr = S.WhisperTranscriber(ctx, model_dir=d)
    assert tr.cfg == cfg
    ref = S.WhisperTranscriber(ctx, synthetic=True, cfg=cfg)
    try:
        ids = tr.ids(torch.from_numpy(x).cuda(), 24000)
        assert ids == ref.ids(x, 24000) and len(ids) > 0
        text = tr(torch.from_numpy(x).cuda(), 24000)
        assert text.split() == [f"w{i}" for i in ids]
        assert ref(x, 24000) == " ".join(f"<{i}>" for i in ids)
    finally:
        tr.close()
        ref.close()


def test_configurations_the_kernels_cannot_serve_are_refused(ctx):
    """`rt_stt_create` checks what the kernels index with: a forced id outside the vocabulary (it would read past the embedding
    table), a chunk that is not 2 n_ctx frames, a head width without an attention instantiation."""
    import dataclasses
    cfg = S.tiny_test_config()
    state = S.synthetic_state(cfg, 789)
    for bad, what in ((dataclasses.replace(cfg, prefix=(291, cfg.vocab + 5)), "prefix id"),
                      (dataclasses.replace(cfg, chunk_seconds=3), "unsupported configuration"),
                      (dataclasses.replace(cfg, heads=4), "unsupported configuration")):
        with pytest.raises(ValueError, match=what):
            S.NativeSTT(ctx, bad, state)
