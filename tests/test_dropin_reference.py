"""The REAL drop-in configuration: with the reference package importable, ``rho_tts_amd.api`` must resolve to
``rho_tts``'s own BaseTTS / TTSFactory / result and error types, ``register()`` must land in the reference's factory
(factory.py:110-123), the provider must inherit the reference's ``generate`` (base_tts.py:960-1101) unchanged, and the whole
host-logic suite (tests/test_pipeline_host.py, including the reference-generated pipeline fixtures) must pass on top of it.

Runs in a child process (the parent stays on the mirror) and only where /root/reference exists - i.e. in the build container,
never on the GPU box.  torchaudio is absent here and stubbed exactly as the reference's own tests do (CLAUDE.md:39); default
provider registration is switched off so that nothing tries to create a venv or reach a package index."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = "/root/reference/src"

SCRIPT = r'''
import json, os, sys, types
sys.modules.setdefault("torchaudio", types.ModuleType("torchaudio"))
from rho_tts_amd import api
assert api.HOST == "rho_tts", api.HOST
import rho_tts, rho_tts.base_tts as B
from rho_tts import TTSFactory
TTSFactory._default_providers_registered = True              # never let default registration run (venv / pip path)
from rho_tts_amd.provider import MI355XQwenTTS, PROVIDER_NAME, register
assert register() == PROVIDER_NAME and TTSFactory._providers[PROVIDER_NAME] is MI355XQwenTTS
assert issubclass(MI355XQwenTTS, B.BaseTTS)
# generate() is the reference's own: the provider's thin override only steps in on the worker ranks of a data-parallel run
_o = MI355XQwenTTS.__new__(MI355XQwenTTS); _o.data_parallel = False
_seen = []
_orig = B.BaseTTS.generate
B.BaseTTS.generate = lambda self, *a, **k: _seen.append((a, k)) or "inherited"
assert MI355XQwenTTS.generate(_o, ["x"], None, None, "wav") == "inherited" and _seen and _seen[0][0][0] == ["x"]
B.BaseTTS.generate = _orig
assert MI355XQwenTTS.stream is not B.BaseTTS.stream        # same results per segment, two batched calls (tests/test_provider_gpu.py)
assert MI355XQwenTTS._run_pipeline is not B.BaseTTS._run_pipeline                                     # the batching seam
assert api.CancelledException is rho_tts.CancelledException and api.GenerationResult is rho_tts.GenerationResult
try:
    TTSFactory.register_provider("bad", dict)
    raise SystemExit("TypeError expected")
except TypeError:
    pass
# the pipeline fixtures through the reference's own generate()
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from tests.test_pipeline_host import Fake
g = json.load(open("tests/golden/pipeline_golden.json"))
for case in ("single_one_segment", "three_segments_forced"):
    t = Fake(batch_size=4); t._max_chars_explicit = True
    res = t.generate(list(g[case]["texts"]))
    for r, want in zip(res, g[case]["out"]):
        assert isinstance(r, rho_tts.GenerationResult) and r.audio.numel() == want["len"] and r.segments_count == want["segments"], case
import pytest
sys.exit(pytest.main(["-q", "-x", "-p", "no:cacheprovider", "tests/test_pipeline_host.py"]))
'''


@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="the reference tree is only present in the build container")
def test_provider_plugs_into_the_reference_package():
    env = dict(os.environ, PYTHONPATH=REF_SRC + os.pathsep + ROOT, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", SCRIPT], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-2000:]
