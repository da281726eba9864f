#!/usr/bin/env python3
"""Generate tests/golden/*.npz|json from the REFERENCE implementation.

Runs only in the build container, where /root/reference exists.  It imports the
reference's own ``BaseTTS`` / ``QwenTTS`` (with an empty stub ``torchaudio``
module, the reference's own test convention, CLAUDE.md:39) and records
inputs + outputs as plain arrays.  Nothing of the reference's source text is
stored: the fixtures are data only.  The GPU box never runs this script.

    python tests/golden/make_golden.py
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF_SRC = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))

sys.dont_write_bytecode = True
sys.modules.setdefault("torchaudio", types.ModuleType("torchaudio"))
sys.path.insert(0, REF_SRC)

torch.set_num_threads(1)

from rho_tts import BaseTTS, CancellationToken, TTSFactory  # noqa: E402
from rho_tts.providers.qwen import QwenTTS  # noqa: E402

TTSFactory._default_providers_registered = True  # never run default registration (would reach pip)

SR = 24000


class Fake(BaseTTS):
    """Deterministic provider: tone whose pitch/length depend on the text only."""

    def __init__(self, **kw):
        super().__init__(device="cpu", **kw)
        self.calls = []
        self.fail_on = set()
        self.oom_on = set()

    def _generate_audio(self, text, **kwargs):
        self.calls.append(text)
        if text in self.fail_on:
            raise OSError("synthetic failure")          # generic error: retried, then item -> None
        if text in self.oom_on:
            raise RuntimeError("HIP out of memory")     # matched by the reference's OOM string test
        return torch.from_numpy(fake_wave(text))

    @property
    def sample_rate(self):
        return SR


def fake_wave(text: str) -> np.ndarray:
    n = 12000 + 480 * (len(text) % 50)
    f = 180.0 + 7.0 * (sum(map(ord, text)) % 40)
    i = np.arange(n, dtype=np.float64)
    lead = 1200 + 37 * (len(text) % 11)
    env = np.ones(n)
    env[:lead] = 0.0
    env[-lead:] = 0.0
    w = (0.25 * np.sin(2 * np.pi * f * i / SR) + 0.005) * env
    return w.astype(np.float32)


# Scripted validators shared with tests/test_pipeline_host.py (kept identical there): the drift of an attempt is looked up by
# the length of the WAV it is handed (fake_wave gives every text its own length) and the attempt number, the text similarity
# by text and attempt number.
VALIDATION_TEXTS = ["Accepted on the third try", "Fine at once", "Never good enough at all"]
DRIFT_SCRIPT = {VALIDATION_TEXTS[0]: [0.5, 0.1, 0.3], VALIDATION_TEXTS[1]: [0.05], VALIDATION_TEXTS[2]: [0.4, 0.6, 0.2]}
SIM_SCRIPT = {VALIDATION_TEXTS[0]: [0.5, 0.9], VALIDATION_TEXTS[1]: [0.95], VALIDATION_TEXTS[2]: []}


def install_scripted_validators(t, drift_threshold=0.35, sim_threshold=0.85):
    import wave
    by_len = {int(fake_wave(x).shape[0]): x for x in VALIDATION_TEXTS}
    assert len(by_len) == len(VALIDATION_TEXTS)
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
        s = SIM_SCRIPT[text][k]
        t.text_calls.append([text, s])
        return s >= sim_threshold, s, "transcribed " + text

    t._validate_accent_drift = drift
    t._validate_text_match = text_match
    t._auto_sort_audio = lambda path, drift_prob: None
    t._log_text_diff = lambda a, b: None


def text_similarity_fixtures():
    """Pairs scored by the reference's calculate_text_similarity (validation/stt/stt_validator.py:188-232).  The module
    imports its number normaliser at import time, which needs NeMo (absent): the normaliser module is stubbed with a function
    that raises, which the reference catches per call (stt_validator.py:27-31) - and the pairs hold no numerals anyway."""
    stub = types.ModuleType("rho_tts.validation.stt.number_normalizer")

    def _absent(text):
        raise RuntimeError("number normalisation unavailable offline")
    stub.normalize_numbers_to_digits = _absent
    sys.modules["rho_tts.validation.stt.number_normalizer"] = stub
    import logging
    logging.getLogger("rho_tts.validation.stt.stt_validator").setLevel(logging.ERROR)
    from rho_tts.validation.stt import stt_validator as V
    base = ["The quick brown fox jumps over the lazy dog.", "Hello there, General Kenobi!", "A well-known state-of-the-art method",
            "It's a long established fact that a reader will be distracted", "Supercalifragilisticexpialidocious is extraordinarily long",
            "to be or not to be", "An apple a day keeps the doctor away", "one", "I am", ""]
    variants = ["the quick brown fox jumps over the lazy dog", "quick brown fox jumped over lazy dogs", "The quick brown fax jumps over the hazy dog",
                "hello there general kenobi", "Hello their, general Kenobe", "a well known state of the art method", "well known state of art methods",
                "its a long established fact that a reader will be distracted", "It is a long-established fact the reader would be distracted",
                "supercalifragilisticexpialidocius is extraordinarly long", "super cali fragilistic is extra ordinarily long", "to be or not to bee",
                "be to not or be to", "an apple the day keeps doctors away", "apples a day keep the doctor away from me and you and everyone",
                "one", "won", "I am", "i'm", "", "the a an", "completely unrelated words here", "THE QUICK BROWN FOX", "dog lazy the over jumps fox brown quick the"]
    cases = []
    for o in base:
        for t in variants:
            cases.append({"original": o, "transcribed": t, "similarity": float(V.calculate_text_similarity(o, t))})
    words = [("cat", "cut"), ("cat", "dog"), ("ab", "ab"), ("ab", "ac"), ("abc", "abd"), ("extraordinary", "extraordinarly"), ("extraordinary", "extraordinaire"),
             ("hello", "help"), ("kitten", "sitting"), ("", "abc"), ("flaw", "lawn")]
    out = {"pairs": cases,
           "levenshtein": [{"a": a, "b": b, "distance": int(V._levenshtein_distance(a, b)), "fuzzy": bool(V._fuzzy_word_match(a, b))} for a, b in words],
           "normalize": [{"text": x, "normalized": V._normalize_text(x)} for x in base + variants + ["Wait -- what?!  Co-operate; don't   stop.", "\tTabs\nand newlines "]],
           "validate": [list(V.validate_audio_text_match.__defaults__)]}
    with open(os.path.join(HERE, "textsim_golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", len(cases), "text-similarity pairs")


def tone(n, f, amp=0.3, dc=0.0):
    i = np.arange(n, dtype=np.float64)
    return (amp * np.sin(2 * np.pi * f * i / SR) + dc).astype(np.float32)


def qwen_instance():
    q = QwenTTS.__new__(QwenTTS)
    BaseTTS.__init__(q, device="cpu")
    q.qwen3_sr = SR
    q.sound_decay_threshold = 0.3
    return q


def main():
    out = {}
    base = Fake()
    qw = qwen_instance()
    rng = np.random.default_rng(789)

    def T(a):
        return torch.from_numpy(np.array(a, dtype=np.float32, copy=True))

    # ---- leaf cases -------------------------------------------------------
    cases = {}
    z = np.zeros
    cases["k1"] = np.concatenate([z(12000, np.float32), tone(24000, 220, 0.3, 0.01), z(12000, np.float32)])
    cases["noise_burst"] = np.concatenate([
        (1e-4 * rng.standard_normal(5000)).astype(np.float32),
        (0.2 * rng.standard_normal(30000)).astype(np.float32),
        (1e-4 * rng.standard_normal(7777)).astype(np.float32)])
    cases["short_loud"] = (0.1 * rng.standard_normal(700)).astype(np.float32)       # < 2*fade
    cases["tiny"] = (0.1 * rng.standard_normal(100)).astype(np.float32)            # < window
    cases["all_zero"] = z(24000, np.float32)
    cases["quiet"] = (1e-3 * rng.standard_normal(20000)).astype(np.float32)         # below threshold
    cases["edge_thresh"] = (0.0031622776601683794 * np.ones(5000)).astype(np.float32)
    cases["ragged"] = (0.05 * rng.standard_normal(96001)).astype(np.float32)
    cases["loud_to_end"] = tone(48123, 330, 0.4, -0.02)

    for name, x in cases.items():
        out[f"leaf/{name}/x"] = x
        for fs, fe, tag in ((True, True, "both"), (True, False, "start"), (False, True, "end")):
            y = base._trim_silence(T(x), from_start=fs, from_end=fe)
            out[f"leaf/{name}/trim_{tag}_shape"] = np.array(y.shape, dtype=np.int64)
            out[f"leaf/{name}/trim_{tag}"] = y.reshape(-1).numpy().copy()
        out[f"leaf/{name}/dc"] = base._remove_dc_offset(T(x)).numpy().copy()
        out[f"leaf/{name}/fades"] = base._apply_fades(T(x)).numpy().copy()
        out[f"leaf/{name}/fade_in_only"] = base._apply_fades(T(x), fade_in=True, fade_out=False).numpy().copy()
        r, ok = base._validate_sound_decay(T(x))
        out[f"leaf/{name}/decay"] = np.array([r, float(ok)], dtype=np.float64)
        j = base._smooth_segment_join([T(x)])
        out[f"leaf/{name}/join1"] = j.reshape(-1).numpy().copy()
        out[f"leaf/{name}/join1_shape"] = np.array(j.shape, dtype=np.int64)
        out[f"leaf/{name}/post"] = qw._post_process_audio(T(x)).numpy().copy()
        e = torch.sqrt(torch.nn.functional.avg_pool1d((T(x) ** 2).unsqueeze(0), 240, 120, 120).mean(0))
        out[f"leaf/{name}/energy"] = e.numpy().copy()

    # ---- multi-segment joins (K3 + ragged variants) -------------------------
    def seg(k):
        return np.concatenate([z(2400, np.float32), tone(24000, 220 + 110 * k), z(2400, np.float32)])

    joins = {
        "k3_2": [seg(0), seg(1)],
        "k3_3": [seg(0), seg(1), seg(2)],
        "mixed_5": [seg(0), cases["noise_burst"], seg(2), cases["loud_to_end"], seg(1)],
        "short_mid": [seg(0), (0.2 * rng.standard_normal(900)).astype(np.float32), seg(1)],
        "tiny_overlap": [seg(0), (0.2 * rng.standard_normal(8)).astype(np.float32), seg(1)],
        "silent_mid": [seg(0), z(5000, np.float32), seg(1)],
    }
    for name, segs in joins.items():
        out[f"join/{name}/n"] = np.array([len(segs)], dtype=np.int64)
        for i, s in enumerate(segs):
            out[f"join/{name}/seg{i}"] = s
        try:
            y = base._smooth_segment_join([T(s) for s in segs])
            out[f"join/{name}/y"] = y.reshape(-1).numpy().copy()
            yp = qw._post_process_audio(y.reshape(-1).clone())
            out[f"join/{name}/y_post"] = yp.numpy().copy()
            r, ok = base._validate_sound_decay(yp)
            out[f"join/{name}/decay"] = np.array([r, float(ok)], dtype=np.float64)
        except Exception as e:  # the reference raises for 2-D/1-D mixes (all-silent middle segment)
            out[f"join/{name}/error"] = np.frombuffer(type(e).__name__.encode(), dtype=np.uint8)

    # ---- loudness (K4, K5 + more) -------------------------------------------
    i = np.arange(240000, dtype=np.float64)
    k4 = (np.sin(2 * np.pi * 440 * i / SR)).astype(np.float32) * torch.linspace(1, 0.2, 240000).numpy()
    k5 = tone(72000, 440, 0.5)
    loud = {
        "k4": k4.astype(np.float32),
        "k5": k5,
        "gap": np.concatenate([tone(60000, 300, 0.4), z(50000, np.float32), tone(100000, 300, 0.1)]),
        "rising": (tone(200000, 250, 1.0) * np.linspace(0.05, 0.9, 200000)).astype(np.float32),
        "exact_2w": tone(96000, 200, 0.2),
        "just_over": tone(96001, 200, 0.2),
        "hot": (2.5 * rng.standard_normal(150000)).astype(np.float32),
        "near_silent": (1e-9 * np.ones(1000)).astype(np.float32),
    }
    for name, x in loud.items():
        out[f"loud/{name}/x"] = x
        y = qw._post_process_audio(T(x))
        out[f"loud/{name}/y"] = y.numpy().copy()
        r0, _ = base._validate_sound_decay(T(x))
        r1, ok1 = base._validate_sound_decay(y)
        out[f"loud/{name}/decay"] = np.array([r0, r1, float(ok1)], dtype=np.float64)

    np.savez_compressed(os.path.join(HERE, "postprocess_golden.npz"), **out)

    # ---- pipeline behaviour with a deterministic fake provider ---------------
    pipe = {}

    def run(tts, texts, token=None, cb=None):
        token = token or CancellationToken()
        res = tts._run_pipeline(texts, token, cb)
        rec = []
        for r in res:
            if r is None:
                rec.append(None)
            else:
                a, nseg, meta = r
                a = a.reshape(-1).numpy()
                rec.append({"len": int(a.shape[0]), "segments": int(nseg),
                            "abs_sum": float(np.abs(a.astype(np.float64)).sum()),
                            "decay_ratio": float(meta["decay_ratio"]),
                            "meta_keys": sorted(meta.keys())})
                for k in ("drift_prob", "text_similarity"):
                    if k in meta:
                        rec[-1][k] = float(meta[k])
        return rec

    t = Fake()
    t._max_chars_explicit = True
    pipe["single_one_segment"] = {"texts": ["One sentence only"],
                                  "out": run(t, ["One sentence only"]), "calls": list(t.calls)}
    t = Fake()
    t._max_chars_explicit = True
    texts = ["Hello there. General test. Third sentence here."]
    pipe["three_segments_forced"] = {"texts": texts, "out": run(t, texts), "calls": list(t.calls)}
    t = Fake()
    t._max_chars_explicit = True
    t.force_sentence_split = False
    t.max_chars_per_segment = 30
    texts = ["Alpha beta gamma delta. Epsilon zeta eta theta iota. Kappa lambda mu.", "Short one"]
    pipe["max_chars_30"] = {"texts": texts, "out": run(t, texts), "calls": list(t.calls)}
    t = Fake()
    t._max_chars_explicit = True
    texts = [f"Item number {i} of the batch" for i in range(8)]
    t.fail_on = {texts[3]}
    t.oom_on = {texts[6]}
    msgs = []
    pipe["eight_one_fails"] = {"texts": texts, "out": run(t, texts, cb=msgs.append),
                               "calls": list(t.calls), "progress": msgs}
    t = Fake()
    t._max_chars_explicit = True
    t.phonetic_mapping = {"exocrine": "exo-crene"}
    texts = ["The exocrine gland"]
    pipe["phonetic"] = {"texts": texts, "out": run(t, texts), "calls": list(t.calls)}

    # validation with retries (max_iterations = 3) and scripted validators: which audio is kept and which scores are reported
    # (minimum drift over the attempts, last text similarity; base_tts.py:821-906)
    t = Fake()
    t._max_chars_explicit = True
    t.max_iterations = 3
    install_scripted_validators(t)
    texts = list(VALIDATION_TEXTS)
    pipe["validated_retries"] = {"texts": texts, "out": run(t, texts), "calls": list(t.calls), "drift_calls": list(t.drift_calls),
                                 "text_calls": list(t.text_calls)}

    # max_iterations = 1 with an auto-sort directory: drift detection + sorting still run, once per segment, and no score
    # reaches the metadata (base_tts.py:801-818)
    t = Fake()
    t._max_chars_explicit = True
    t.max_iterations = 1
    t.auto_sort_good_dir = "/nonexistent/good"
    install_scripted_validators(t)
    sorted_calls = []
    t._auto_sort_audio = lambda path, drift_prob: sorted_calls.append(float(drift_prob))
    texts = list(VALIDATION_TEXTS)
    pipe["auto_sort_single_pass"] = {"texts": texts, "out": run(t, texts), "calls": list(t.calls), "drift_calls": list(t.drift_calls),
                                     "text_calls": list(t.text_calls), "sorted": sorted_calls}

    seg_cases = {}
    t = Fake()
    for fs in (True, False):
        t.force_sentence_split = fs
        for mc in (10, 25, 60, 200):
            for txt in ["Hello there. General test.", "One sentence only",
                        "A. B. C. D.", "Supercalifragilisticexpialidocious is long. Ok.",
                        "word " * 30, "", "Trailing period. "]:
                seg_cases[f"{int(fs)}|{mc}|{txt}"] = t._split_text_into_segments(txt, mc)
    pipe["split_text"] = seg_cases

    with open(os.path.join(HERE, "pipeline_golden.json"), "w") as f:
        json.dump(pipe, f, indent=1, sort_keys=True)
    text_similarity_fixtures()
    print("wrote", len(out), "arrays and", len(pipe), "pipeline cases")


if __name__ == "__main__":
    main()
