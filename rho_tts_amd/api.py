"""Resolve the host API the provider plugs into.

If the reference package ``rho_tts`` is importable, its own ``BaseTTS`` / ``TTSFactory`` / result and
error types are used, so ``register()`` really registers with the reference's factory (the drop-in of
SURVEY.md section 8b).  Otherwise the mirror in ``hostapi.py`` supplies the same names.
"""
from __future__ import annotations

import importlib
import importlib.util
import sys
import types

HOST = "mirror"
try:  # the reference imports torchaudio at module level (base_tts.py:20); only speed/pitch needs the real thing
    try:
        importlib.import_module("torchaudio")
    except Exception:  # noqa: BLE001
        if importlib.util.find_spec("rho_tts") is not None:
            sys.modules.setdefault("torchaudio", types.ModuleType("torchaudio"))
    _ref = importlib.import_module("rho_tts")
    from rho_tts import (BaseTTS, CancellationToken, CancelledException, FormatConversionError, GenerationResult,  # noqa: F401
                         ProviderInfo, ProviderNotFoundError, RhoTTSError, TTSFactory, VoiceInfo)
    HOST = "rho_tts"
except Exception:  # noqa: BLE001
    from .hostapi import (BaseTTS, CancellationToken, CancelledException, FormatConversionError, GenerationResult,  # noqa: F401
                          ProviderInfo, ProviderNotFoundError, RhoTTSError, TTSFactory, VoiceInfo)
