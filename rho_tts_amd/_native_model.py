"""ctypes declarations of the model group of the C ABI (filled in as the group grows)."""
from __future__ import annotations

import ctypes as C


def declare(lib: C.CDLL) -> None:
    pass
