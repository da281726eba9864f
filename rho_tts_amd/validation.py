"""Text-match scoring of a transcription against the text that was to be spoken - the host-side half of the reference's
STT validation (validation/stt/stt_validator.py: ``_normalize_text`` :22-39, ``_levenshtein_distance`` :151-169,
``_fuzzy_word_match`` :172-185, ``calculate_text_similarity`` :188-232, ``validate_audio_text_match`` :235-260), restated so
that a provider can validate a waveform it still holds in HBM without the temp-WAV round trip of base_tts.py:821-830: the
transcriber is a callable on the tensor (``MI355XQwenTTS.transcriber``), the scoring below is pure string work.

Pinned by tests/golden/textsim_golden.json (pairs scored by the reference's own function).  Number normalisation
(number_normalizer.py, NeMo inverse text normalisation) is NOT restated: its dependency is absent offline, the reference
itself degrades to "warn and continue" when it fails (:27-31), and the fixtures therefore contain no numerals.
"""
from __future__ import annotations

import re
from difflib import SequenceMatcher
from typing import Callable, Optional, Tuple


def normalize_text(text: str, number_normalizer: Optional[Callable[[str], str]] = None) -> str:
    if number_normalizer is not None:
        try:
            text = number_normalizer(text)
        except Exception:  # noqa: BLE001  (the reference warns and goes on)
            pass
    text = text.lower()
    text = re.sub(r"\b(the|a|an)\b", " ", text)
    text = text.replace("-", " ")
    text = re.sub(r"[^\w\s']", " ", text)
    text = re.sub(r"\s+", " ", text)
    return text.strip()


def levenshtein_distance(s1: str, s2: str) -> int:
    if len(s1) < len(s2):
        s1, s2 = s2, s1
    if not s2:
        return len(s1)
    prev = list(range(len(s2) + 1))
    for i, c1 in enumerate(s1):
        cur = [i + 1]
        for j, c2 in enumerate(s2):
            cur.append(min(prev[j + 1] + 1, cur[j] + 1, prev[j] + (c1 != c2)))
        prev = cur
    return prev[-1]


def fuzzy_word_match(w1: str, w2: str, max_distance: int = 2) -> bool:
    if w1 == w2:
        return True
    if len(w1) < 3 or len(w2) < 3:
        return False
    limit = max_distance + 1 if (len(w1) > 8 or len(w2) > 8) else max_distance
    return levenshtein_distance(w1, w2) <= limit


def calculate_text_similarity(original_text: str, transcribed_text: str, number_normalizer=None) -> float:
    """max(Jaccard, matched / original words, character sequence ratio) over normalised texts, fuzzy word matches counted."""
    o, t = normalize_text(original_text, number_normalizer), normalize_text(transcribed_text, number_normalizer)
    ow, tw = set(o.split()), set(t.split())
    if not ow or not tw:
        return 0.0
    fuzzy = 0
    # (set iteration order only decides WHICH transcribed word a fuzzy match pairs with, never the count: every original word
    #  counts at most once and transcribed words are not consumed - exactly the reference's loop)
    for a in ow - tw:
        if any(fuzzy_word_match(a, b) for b in tw - ow):
            fuzzy += 1
    total = len(ow & tw) + fuzzy
    union = len(ow | tw)
    return max(total / union if union else 0.0, total / len(ow), SequenceMatcher(None, o, t).ratio())


def validate_text_match(transcribed: Optional[str], expected_text: str, threshold: float = 0.85) -> Tuple[bool, float, Optional[str]]:
    """(is_valid, similarity, transcription); a failed transcription passes with similarity 0.0 (:249-252)."""
    if transcribed is None:
        return True, 0.0, None
    sim = calculate_text_similarity(expected_text, transcribed)
    return sim >= threshold, sim, transcribed
