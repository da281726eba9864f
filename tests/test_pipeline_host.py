"""(Runs against whichever host API rho_tts_amd.api resolved: the mirror normally, the reference's own package when it is
importable - tests/test_dropin_reference.py runs this whole file in that mode.)
Host logic of the provider against the reference-generated pipeline fixtures
(tests/golden/pipeline_golden.json, produced by tests/golden/make_golden.py from the reference's own
BaseTTS._run_pipeline with a deterministic fake provider).  The numeric leaves are supplied by the CPU
oracle here (no GPU in this suite); tests/test_provider_gpu.py repeats the comparison with the HIP leaves."""
import json
import threading

import numpy as np
import pytest
import torch

from oracle import postprocess as OP
from rho_tts_amd import api, hostapi
from rho_tts_amd.provider import BatchedPipeline

SR = 24000
torch.set_num_threads(1)


def fake_wave(text: str) -> np.ndarray:          # must stay identical to tests/golden/make_golden.py
    n = 12000 + 480 * (len(text) % 50)
    f = 180.0 + 7.0 * (sum(map(ord, text)) % 40)
    i = np.arange(n, dtype=np.float64)
    lead = 1200 + 37 * (len(text) % 11)
    env = np.ones(n)
    env[:lead] = 0.0
    env[-lead:] = 0.0
    return ((0.25 * np.sin(2 * np.pi * f * i / SR) + 0.005) * env).astype(np.float32)


class OracleLeaves:
    def _finish_items(self, items):
        p = OP.PostParams(sample_rate=self.sample_rate, sound_decay_threshold=self.sound_decay_threshold,
                          inter_sentence_pause_sec=self.inter_sentence_pause_sec)
        return [OP.finish_item(list(it), p, loudness=False) for it in items]


class Fake(BatchedPipeline, OracleLeaves, api.BaseTTS):
    def __init__(self, batch_size=1, **kw):
        super().__init__(device="cpu", **kw)
        self.batch_size = batch_size
        self.calls, self.fail_on, self.oom_on, self.value_error_on = [], set(), set(), set()

    def _generate_audio(self, text, **kwargs):
        texts = [text] if isinstance(text, str) else list(text)
        out = []
        for t in texts:
            self.calls.append(t)
            if t in self.value_error_on:
                raise ValueError("bad configuration")
            if t in self.fail_on:
                raise OSError("synthetic failure")
            if t in self.oom_on:
                raise RuntimeError("HIP out of memory")
            out.append(torch.from_numpy(fake_wave(t)))
        return out[0] if isinstance(text, str) else out

    @property
    def sample_rate(self):
        return SR


def run(tts, texts, token=None, cb=None):
    res = tts._run_pipeline(texts, token or api.CancellationToken(), cb)
    rec = []
    for r in res:
        if r is None:
            rec.append(None)
        else:
            a, nseg, meta = r
            a = a.reshape(-1).numpy()
            rec.append({"len": int(a.shape[0]), "segments": int(nseg), "abs_sum": float(np.abs(a.astype(np.float64)).sum()),
                        "decay_ratio": float(meta["decay_ratio"]), "meta_keys": sorted(meta.keys())})
    return rec


def same(got, want):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        if w is None:
            assert g is None
            continue
        assert g["len"] == w["len"] and g["segments"] == w["segments"] and g["meta_keys"] == w["meta_keys"]
        assert abs(g["abs_sum"] - w["abs_sum"]) < 1e-6 * w["abs_sum"]
        assert abs(g["decay_ratio"] - w["decay_ratio"]) < 1e-9


@pytest.mark.parametrize("bs", [1, 3, 32])
def test_pipeline_cases_match_reference(golden_pipe, bs):
    g = golden_pipe
    t = Fake(batch_size=bs); t._max_chars_explicit = True
    same(run(t, g["single_one_segment"]["texts"]), g["single_one_segment"]["out"])
    assert t.calls == g["single_one_segment"]["calls"]
    t = Fake(batch_size=bs); t._max_chars_explicit = True
    same(run(t, g["three_segments_forced"]["texts"]), g["three_segments_forced"]["out"])
    assert t.calls == g["three_segments_forced"]["calls"]
    t = Fake(batch_size=bs); t._max_chars_explicit = True; t.force_sentence_split = False; t.max_chars_per_segment = 30
    same(run(t, g["max_chars_30"]["texts"]), g["max_chars_30"]["out"])
    assert t.calls == g["max_chars_30"]["calls"]
    t = Fake(batch_size=bs); t._max_chars_explicit = True; t.phonetic_mapping = {"exocrine": "exo-crene"}
    same(run(t, g["phonetic"]["texts"]), g["phonetic"]["out"])
    assert t.calls == g["phonetic"]["calls"]                         # the mapped text reaches _generate_audio


@pytest.mark.parametrize("bs", [1, 4, 32])
def test_failures_are_isolated_per_item(golden_pipe, bs):
    c = golden_pipe["eight_one_fails"]
    t = Fake(batch_size=bs); t._max_chars_explicit = True
    t.fail_on, t.oom_on = {c["texts"][3]}, {c["texts"][6]}
    msgs = []
    same(run(t, c["texts"], cb=msgs.append), c["out"])
    assert msgs == c["progress"]
    assert sorted(set(t.calls)) == sorted(set(c["calls"]))           # every text was attempted


def test_split_text_matches_reference(golden_pipe):
    for key, want in golden_pipe["split_text"].items():
        fs, mc, txt = key.split("|", 2)
        assert hostapi.split_text_into_segments(txt, int(mc), bool(int(fs))) == want, key


def test_value_error_propagates_and_other_runtime_errors_too():
    t = Fake(); t._max_chars_explicit = True
    t.value_error_on = {"b"}
    with pytest.raises(ValueError):
        t._run_pipeline(["a", "b"], api.CancellationToken())
    with pytest.raises(ValueError):
        t.generate(["a", "b"])                                        # generate() re-raises ValueError (base_tts.py:1096-1097)

    class Boom(Fake):
        def _generate_audio(self, text, **kw):
            raise RuntimeError("device lost")
    with pytest.raises(RuntimeError):
        Boom()._run_pipeline(["a"], api.CancellationToken())
    assert Boom().generate("a") is None                                # ... but generate() swallows it and returns None


def test_cancellation_raises_inside_and_returns_none_outside():
    t = Fake(batch_size=2); t._max_chars_explicit = True
    tok = api.CancellationToken()
    tok.cancel()
    with pytest.raises(api.CancelledException):
        t._run_pipeline(["a", "b"], tok)
    assert t.generate(["a", "b"], cancellation_token=tok) is None
    tok2 = api.CancellationToken()
    t2 = Fake(batch_size=1); t2._max_chars_explicit = True
    seen = []

    def cb(msg):
        seen.append(msg)
        if len(seen) == 2:
            tok2.cancel()
    with pytest.raises(api.CancelledException):
        t2._run_pipeline(["One.", "Two.", "Three.", "Four."], tok2, cb)
    assert len(t2.calls) == 2                                          # cancelled between segments, nothing further generated


def test_decay_retry_regenerates_whole_item_then_gives_best():
    class Decaying(Fake):
        def _generate_audio(self, text, **kw):
            out = super()._generate_audio(text, **kw)
            fix = lambda a: a * torch.linspace(1.0, 0.0, a.numel()) if self.seed == 789 else a
            return fix(out) if isinstance(text, str) else [fix(a) for a in out]
    t = Decaying(); t._max_chars_explicit = True
    res = t._run_pipeline(["Hello world"], api.CancellationToken())
    assert len(t.calls) == 2 and t.seed != 789                          # one regeneration with a wall-clock seed
    assert res[0][2]["decay_ratio"] >= 0.3

    class Always(Fake):
        def _generate_audio(self, text, **kw):
            out = super()._generate_audio(text, **kw)
            f = lambda a: a * torch.linspace(1.0, 0.0, a.numel())
            return f(out) if isinstance(text, str) else [f(a) for a in out]
    t = Always(); t._max_chars_explicit = True
    res = t._run_pipeline(["Hello world"], api.CancellationToken())
    assert len(t.calls) == t.max_decay_retries and res[0] is not None and res[0][2]["decay_ratio"] < 0.3


def test_validation_loop_only_runs_when_max_iterations_gt_1():
    calls = {"drift": 0, "text": 0}

    class V(Fake):
        def _validate_accent_drift(self, path):
            calls["drift"] += 1
            return (0.5, False) if calls["drift"] == 1 else (0.05, True)

        def _validate_text_match(self, path, text):
            calls["text"] += 1
            return True, 0.93, text
    t = V(); t._max_chars_explicit = True
    t._run_pipeline(["Hello"], api.CancellationToken())
    assert calls == {"drift": 0, "text": 0}                              # max_iterations == 1: no validation at all
    t = V(); t._max_chars_explicit = True; t.max_iterations = 3
    res = t._run_pipeline(["Hello"], api.CancellationToken())
    assert calls == {"drift": 2, "text": 1} and len(t.calls) == 2
    assert res[0][2]["drift_prob"] == 0.05 and res[0][2]["text_similarity"] == 0.93
    assert sorted(res[0][2]) == ["decay_ratio", "drift_prob", "text_similarity"]


def test_generate_modes_and_files(tmp_path):
    t = Fake(batch_size=8); t._max_chars_explicit = True
    r = t.generate("Hello there")
    assert isinstance(r, api.GenerationResult) and r.path is None and r.sample_rate == SR and r.segments_count == 1
    assert abs(r.duration_sec - r.audio.numel() / SR) < 1e-9 and r.decay_ratio is not None and r.drift_prob is None
    rs = t.generate(["One", "Two"], output_path=str(tmp_path / "out"))
    assert [x.path for x in rs] == [str(tmp_path / "out_0.wav"), str(tmp_path / "out_1.wav")]
    import wave
    with wave.open(rs[0].path) as wf:
        assert wf.getframerate() == SR and wf.getnchannels() == 1 and wf.getsampwidth() == 2
        pcm = np.frombuffer(wf.readframes(wf.getnframes()), dtype="<i2")
    assert np.array_equal(pcm, OP.pcm16(rs[0].audio))
    one = t.generate("Solo", output_path=str(tmp_path / "solo.wav"))
    assert one.path == str(tmp_path / "solo.wav")
    with pytest.raises(api.FormatConversionError):
        t.generate("x", format="aiff")
    t.fail_on = {"bad"}
    assert t.generate("bad") is None and t.generate(["bad", "bad"]) is None
    mixed = t.generate(["bad", "good"])
    assert mixed[0] is None and mixed[1] is not None


def test_factory_registration_contract():
    saved = dict(api.TTSFactory._providers)
    try:
        api.TTSFactory.register_provider("fake", Fake)
        assert "fake" in api.TTSFactory.list_providers()
        assert isinstance(api.TTSFactory.get_tts_instance("fake", batch_size=2), Fake)
        with pytest.raises(TypeError):
            api.TTSFactory.register_provider("nope", dict)
        with pytest.raises(api.ProviderNotFoundError):
            api.TTSFactory.get_tts_instance("missing")
        api.TTSFactory.register_provider("fake", Fake)              # last write wins, silently
    finally:
        api.TTSFactory._providers = saved


def test_async_and_token_threading():
    import asyncio
    t = Fake(); t._max_chars_explicit = True
    r = asyncio.run(t.async_generate("Hello"))
    assert r is not None and r.audio.numel() > 0
    tok = api.CancellationToken()
    th = threading.Thread(target=tok.cancel)
    th.start(); th.join()
    assert tok.is_cancelled()
    tok.reset()
    assert not tok.is_cancelled()
    with pytest.raises(api.CancelledException):
        tok.cancel(); tok.raise_if_cancelled()


def test_provider_constructor_contract_without_gpu():
    from rho_tts_amd.provider import MI355XQwenTTS, PROVIDER_NAME, register
    with pytest.raises(ValueError):
        MI355XQwenTTS(reference_audio="x.wav")                          # reference_text is required with reference_audio
    with pytest.raises(ValueError):
        MI355XQwenTTS(device="cpu")                                     # no CPU path, said loudly
    p = MI355XQwenTTS(speaker="Ryan", model_path="Qwen/Qwen3-TTS-12Hz-1.7B-CustomVoice")
    # defaults = QwenTTS.__init__'s (providers/qwen.py:59-60) except batch_size (dead configuration there, the decode rows here)
    assert p.sample_rate == 24000 and p.batch_size == 32 and p.force_sentence_split is False and p.max_iterations == 10
    assert (p.accent_drift_threshold, p.text_similarity_threshold, p.sound_decay_threshold, p.seed) == (0.17, 0.85, 0.3, 789)
    assert p.voice_cloning is False and p.provider_info().supports_voice_cloning
    from rho_tts_amd import api
    saved = dict(api.TTSFactory._providers)
    try:
        assert register() == PROVIDER_NAME and PROVIDER_NAME in api.TTSFactory.list_providers()
        assert issubclass(MI355XQwenTTS, api.BaseTTS)
    finally:
        api.TTSFactory._providers = saved


# ---- scripted validators: identical to tests/golden/make_golden.py (which cannot be imported here: it imports the reference)
VALIDATION_TEXTS = ["Accepted on the third try", "Fine at once", "Never good enough at all"]
DRIFT_SCRIPT = {VALIDATION_TEXTS[0]: [0.5, 0.1, 0.3], VALIDATION_TEXTS[1]: [0.05], VALIDATION_TEXTS[2]: [0.4, 0.6, 0.2]}
SIM_SCRIPT = {VALIDATION_TEXTS[0]: [0.5, 0.9], VALIDATION_TEXTS[1]: [0.95], VALIDATION_TEXTS[2]: []}


def install_scripted_validators(t, drift_threshold=0.35, sim_threshold=0.85):
    import wave
    by_len = {int(fake_wave(x).shape[0]): x for x in VALIDATION_TEXTS}
    t.drift_calls, t.text_calls = [], []
    n_drift, n_text = {}, {}

    def drift(path):
        with wave.open(path, "rb") as wf:
            text = by_len[wf.getnframes()]
        k = n_drift.get(text, 0)
        n_drift[text] = k + 1
        d = DRIFT_SCRIPT[text][k]
        t.drift_calls.append([text, d])
        return d, d < drift_threshold

    def text_match(path, text):
        k = n_text.get(text, 0)
        n_text[text] = k + 1
        s = SIM_SCRIPT[text][k]                 # (raises for the third text: a validator that fails counts as a failed attempt)
        t.text_calls.append([text, s])
        return s >= sim_threshold, s, "transcribed " + text

    t._validate_accent_drift = drift
    t._validate_text_match = text_match
    t._auto_sort_audio = lambda path, drift_prob: None
    t._log_text_diff = lambda a, b: None


@pytest.mark.parametrize("bs", [1, 2, 32])
def test_validation_retries_keep_the_references_scores(golden_pipe, bs):
    """max_iterations = 3 with scripted validators: the audio kept and the reported drift_prob (MINIMUM over the attempts, even
    when a later attempt is the accepted one) / text_similarity (LAST one computed) equal the reference's _run_pipeline
    (base_tts.py:821-906), including a validator that raises after the drift was recorded."""
    c = golden_pipe["validated_retries"]
    t = Fake(batch_size=bs); t._max_chars_explicit = True
    t.max_iterations = 3
    install_scripted_validators(t)
    got = run_with_scores(t, c["texts"])
    same(got, c["out"])
    for g, w in zip(got, c["out"]):
        for k in ("drift_prob", "text_similarity"):
            assert (k in g) == (k in w) and (k not in w or abs(g[k] - w[k]) < 1e-12), (k, g, w)
    # every attempt the reference made was made (batching interleaves the texts, so compare per text)
    for text in c["texts"]:
        assert [d for tx, d in t.drift_calls if tx == text] == [d for tx, d in c["drift_calls"] if tx == text]
        assert [s for tx, s in t.text_calls if tx == text] == [s for tx, s in c["text_calls"] if tx == text]
        assert t.calls.count(text) == c["calls"].count(text)


@pytest.mark.parametrize("bs", [1, 2, 32])
def test_auto_sort_runs_without_validation_retries(golden_pipe, bs):
    """max_iterations = 1 with an auto-sort directory (base_tts.py:801-818): every accepted segment is scored once and handed
    to _auto_sort_audio, nothing is retried, no text match runs and no score reaches the metadata - as the reference does."""
    c = golden_pipe["auto_sort_single_pass"]
    t = Fake(batch_size=bs); t._max_chars_explicit = True
    t.max_iterations = 1
    t.auto_sort_good_dir = "/nonexistent/good"
    install_scripted_validators(t)
    sorted_calls = []
    t._auto_sort_audio = lambda path, drift_prob: sorted_calls.append(float(drift_prob))
    got = run_with_scores(t, c["texts"])
    same(got, c["out"])
    assert all(g["meta_keys"] == ["decay_ratio"] for g in got)
    assert sorted(map(tuple, t.drift_calls)) == sorted(map(tuple, c["drift_calls"])) and t.text_calls == c["text_calls"] == []
    assert sorted(sorted_calls) == sorted(c["sorted"]) and t.calls == c["calls"]
    # without an auto-sort directory no validator is touched at all
    t2 = Fake(batch_size=bs); t2._max_chars_explicit = True
    install_scripted_validators(t2)
    run_with_scores(t2, c["texts"])
    assert t2.drift_calls == [] and t2.text_calls == []


def run_with_scores(tts, texts):
    res = tts._run_pipeline(texts, api.CancellationToken(), None)
    rec = []
    for a, nseg, meta in res:
        a = a.reshape(-1).numpy()
        r = {"len": int(a.shape[0]), "segments": int(nseg), "abs_sum": float(np.abs(a.astype(np.float64)).sum()),
             "decay_ratio": float(meta["decay_ratio"]), "meta_keys": sorted(meta.keys())}
        r.update({k: float(meta[k]) for k in ("drift_prob", "text_similarity") if k in meta})
        rec.append(r)
    return rec
