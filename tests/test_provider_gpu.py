"""The provider end to end on the GPU: reference pipeline fixtures with the HIP leaves, and the real
model path (tiny config) through generate() / stream() / factory registration."""
import json
import os
import wave

import numpy as np
import pytest
import torch

from oracle import postprocess as OP
from rho_tts_amd import api, config
from rho_tts_amd.provider import BatchedPipeline, HipAudioLeaves, MI355XQwenTTS, register

pytestmark = pytest.mark.gpu
SR = 24000

from tests.test_pipeline_host import fake_wave, run, same  # noqa: E402


class HipFake(BatchedPipeline, HipAudioLeaves, api.BaseTTS):
    """Reference-style fake generator + the HIP post-processing leaves."""

    def __init__(self, batch_size=4, device="cpu"):
        super().__init__(device=device)
        self.batch_size, self.calls, self.out_device = batch_size, [], device
        from rho_tts_amd import _native
        self._c = _native.Context(0)

    def _native_ctx(self):
        return self._c

    def _generate_audio(self, text, **kw):
        texts = [text] if isinstance(text, str) else list(text)
        self.calls += texts
        out = [torch.from_numpy(fake_wave(t)).to("cuda" if self.out_device != "cpu" else "cpu") for t in texts]
        return out[0] if isinstance(text, str) else out

    def _finish_items(self, items):          # the reference's fake has no loudness stage (base _post_process_audio is a no-op)
        from rho_tts_amd import _native
        outs, stats = self._post([list(i) for i in items], _native.POST_PIPELINE & ~_native.POST_LOUDNESS)
        return [(o, s.decay_ratio, bool(s.decay_ok)) for o, s in zip(outs, stats)]

    @property
    def sample_rate(self):
        return SR


def rec(res):
    out = []
    for r in res:
        a, nseg, meta = r
        a = a.reshape(-1).cpu().numpy()
        out.append({"len": int(a.shape[0]), "segments": int(nseg), "abs_sum": float(np.abs(a.astype(np.float64)).sum()),
                    "decay_ratio": float(meta["decay_ratio"]), "meta_keys": sorted(meta.keys())})
    return out


@pytest.mark.parametrize("dev", ["cpu", "cuda"])
def test_reference_pipeline_fixtures_with_hip_leaves(golden_pipe, dev):
    g = golden_pipe
    for case, setup in (("single_one_segment", {}), ("three_segments_forced", {}),
                        ("max_chars_30", {"force_sentence_split": False, "max_chars_per_segment": 30})):
        t = HipFake(device=dev)
        t._max_chars_explicit = True
        for k, v in setup.items():
            setattr(t, k, v)
        got = rec(t._run_pipeline(g[case]["texts"], api.CancellationToken()))
        for a, w in zip(got, g[case]["out"]):
            assert a["len"] == w["len"] and a["segments"] == w["segments"]
            assert abs(a["abs_sum"] - w["abs_sum"]) < 1e-5 * w["abs_sum"]
            assert abs(a["decay_ratio"] - w["decay_ratio"]) < 1e-5
        assert t.calls == g[case]["calls"]


def test_leaf_methods_behave_like_the_reference(golden_post):
    t = HipFake()
    x = torch.from_numpy(golden_post["leaf/k1/x"])
    assert t._trim_silence(x).shape == (24240,)
    assert t._trim_silence(x.unsqueeze(0)).shape == (24240,)                    # 2-D in -> 1-D out (base_tts.py:392)
    assert t._trim_silence(torch.zeros(24000)).shape == (1, 240)                # all-silent quirk (:379-380)
    assert abs(float(t._remove_dc_offset(x).mean())) < 1e-7
    f = t._apply_fades(torch.ones(2000))
    assert float(f[0]) < 0.01 and float(f[-1]) < 0.01 and float(f[1000]) == 1.0
    assert t._apply_fades(torch.ones(100)).tolist() == [1.0] * 100              # shorter than two fades: untouched
    assert t._smooth_segment_join([]) is None
    r, ok = t._validate_sound_decay(torch.from_numpy(golden_post["loud/k4/x"]))
    assert abs(r - 0.393182) < 1e-5 and ok
    assert t._validate_sound_decay(torch.zeros(0)) == (1.0, True)
    t.trim_silence = False
    assert t._trim_silence(x).shape == x.shape


@pytest.fixture(scope="module")
def tiny_provider(tmp_path_factory):
    d = tmp_path_factory.mktemp("voice")
    ref = d / "ref.wav"
    i = np.arange(SR * 2)
    pcm = (0.3 * np.sin(2 * np.pi * 150 * i / SR) * (0.6 + 0.4 * np.sin(2 * np.pi * 3 * i / SR)) * 32767).astype("<i2")
    with wave.open(str(ref), "wb") as wf:
        wf.setnchannels(1); wf.setsampwidth(2); wf.setframerate(SR); wf.writeframes(pcm.tobytes())
    saved = dict(api.TTSFactory._providers)
    name = register()
    p = api.TTSFactory.get_tts_instance(name, reference_audio=str(ref), reference_text="a short reference sentence",
                                        model_path="tiny", batch_size=4)
    yield p
    p.close()
    api.TTSFactory._providers = saved


def test_generate_through_the_factory(tiny_provider, tmp_path):
    p = tiny_provider
    assert isinstance(p, MI355XQwenTTS) and p.voice_cloning
    r = p.generate("Hello there general test of the path")
    assert isinstance(r, api.GenerationResult) and r.sample_rate == 24000 and r.segments_count == 1
    assert r.audio.dim() == 1 and r.audio.numel() > 100 and r.decay_ratio is not None
    assert abs(r.duration_sec - r.audio.numel() / 24000) < 1e-9
    rms_db = 20 * np.log10(float(torch.sqrt(torch.mean(r.audio.float() ** 2))) + 1e-12)
    assert abs(rms_db + 23.0) < 1.0 and float(r.audio.abs().max()) <= 0.95       # loudness stage ran
    texts = [f"Sentence number {i} with some more words in it" for i in range(6)]
    rs = p.generate(texts, output_path=str(tmp_path / "o"))
    assert len(rs) == 6 and all(x is not None and os.path.exists(x.path) for x in rs)
    again = p.generate(texts)
    for a, b in zip(rs, again):
        assert torch.equal(a.audio.cpu(), b.audio.cpu())                         # same seed, same texts -> same audio
    p.seed = 790
    other = p.generate(texts[:2])
    assert not torch.equal(other[0].audio.cpu(), rs[0].audio.cpu())
    p.seed = 789


def test_batched_equals_one_at_a_time(tiny_provider):
    p = tiny_provider
    texts = ["Alpha beta gamma delta", "One two", "A somewhat longer piece of text to speak here"]
    both = p.generate(texts)
    p.batch_size = 1
    try:
        single = p.generate(texts)
    finally:
        p.batch_size = 4
    for a, b in zip(both, single):
        assert torch.equal(a.audio.cpu(), b.audio.cpu())


def test_stream_cancel_and_config_errors(tiny_provider):
    p = tiny_provider
    p.max_chars_per_segment, p._max_chars_explicit, p.force_sentence_split = 30, True, True
    try:
        parts = list(p.stream("First sentence here. Second sentence there. Third one."))
    finally:
        p.max_chars_per_segment, p._max_chars_explicit, p.force_sentence_split = 1000, False, False
    assert len(parts) == 3 and all(x.segments_count == 1 and x.audio.numel() > 0 for x in parts)
    tok = api.CancellationToken()
    tok.cancel()
    assert p.generate("Anything", cancellation_token=tok) is None
    q = MI355XQwenTTS(model_path="tiny")                                          # Base model without reference audio
    with pytest.raises(ValueError):
        q.generate("Hello")
    q.close()
    c = MI355XQwenTTS(model_path="CustomVoice-tiny", speaker="Ryan")
    r = c.generate("Built in voice")
    assert r is not None and r.audio.numel() > 0
    c.close()
