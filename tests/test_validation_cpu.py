"""Text-match scoring (rho_tts_amd/validation.py) against pairs scored by the reference's own calculate_text_similarity
(validation/stt/stt_validator.py:188-232; fixtures: tests/golden/textsim_golden.json, made by tests/golden/make_golden.py),
and the tensor-level validation hooks of the batched pipeline (no temp WAV when both are set)."""
import json
import os

import numpy as np
import pytest
import torch

from rho_tts_amd import api, validation as V

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "textsim_golden.json")) as f:
        return json.load(f)


def test_similarity_pairs_equal_the_reference(golden):
    assert len(golden["pairs"]) >= 200
    for c in golden["pairs"]:
        got = V.calculate_text_similarity(c["original"], c["transcribed"])
        assert got == c["similarity"], (c, got)                 # same integers and the same SequenceMatcher: exact
    assert any(0.0 < c["similarity"] < 1.0 for c in golden["pairs"]) and any(c["similarity"] == 1.0 for c in golden["pairs"])


def test_levenshtein_fuzzy_and_normalisation_equal_the_reference(golden):
    for c in golden["levenshtein"]:
        assert V.levenshtein_distance(c["a"], c["b"]) == c["distance"] == V.levenshtein_distance(c["b"], c["a"])
        assert V.fuzzy_word_match(c["a"], c["b"]) == c["fuzzy"]
    for c in golden["normalize"]:
        assert V.normalize_text(c["text"]) == c["normalized"], c


def test_validate_text_match_contract(golden):
    assert golden["validate"] == [[0.85]]                         # the reference's default threshold
    assert V.validate_text_match(None, "anything") == (True, 0.0, None)     # failed transcription passes (stt_validator.py:249-252)
    ok, sim, tr = V.validate_text_match("hello there general kenobi", "Hello there, General Kenobi!")
    assert ok and sim == 1.0 and tr == "hello there general kenobi"
    ok, sim, _ = V.validate_text_match("completely unrelated words", "Hello there, General Kenobi!")
    assert not ok and sim < 0.5
    # a number normaliser that raises is ignored, one that works is applied first
    assert V.normalize_text("Two hundred", lambda t: (_ for _ in ()).throw(RuntimeError("x"))) == "two hundred"
    assert V.normalize_text("Two hundred", lambda t: t.replace("Two hundred", "200")) == "200"


def test_tensor_validators_skip_the_temp_wav(monkeypatch, tmp_path):
    from tests.test_pipeline_host import Fake
    import tempfile
    made = []
    real = tempfile.mkstemp
    monkeypatch.setattr(tempfile, "mkstemp", lambda *a, **k: (made.append(1), real(*a, **k))[1])
    t = Fake(batch_size=4); t._max_chars_explicit = True
    t.max_iterations = 2
    seen = []
    t.drift_scorer = lambda audio, sr: (seen.append((tuple(audio.shape), sr)), 0.05)[1]
    t.transcriber = lambda audio, sr: "hello general test"
    res = t._run_pipeline(["Hello general test", "Something else entirely"], api.CancellationToken(), None)
    assert not made                                              # no temp file was created
    assert len(seen) == 3 and all(sr == 24000 for _, sr in seen)  # item 0 accepted at once, item 1 tried twice
    (a0, n0, m0), (a1, n1, m1) = res
    assert m0["drift_prob"] == 0.05 and m0["text_similarity"] == 1.0
    assert m1["drift_prob"] == 0.05 and m1["text_similarity"] < 0.5 and a1.numel() > 0      # exhausted: best by drift is kept
