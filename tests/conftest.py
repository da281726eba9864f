import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# Parity tests run on seeded synthetic weights (no checkpoints exist offline): the engine only does that when asked to
# (rho_tts_amd/engine.py); tests that check the refusal clear the variable themselves.
os.environ.setdefault("RHO_TTS_AMD_SYNTHETIC", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_post():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "postprocess_golden.npz"))


@pytest.fixture(scope="session")
def golden_pipe():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "pipeline_golden.json")) as f:
        return json.load(f)
