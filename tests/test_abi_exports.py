"""CPU checks of the C-ABI boundary: the library builds, loads, and exports every symbol include/rho_tts_amd.h (the drop-in
boundary) and include/rho_tts_amd_debug.h (measurement / test entry points) declare.  No compute call is made (no GPU here)."""
import ctypes
import os
import re

from rho_tts_amd import _build, _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="rho_tts_amd.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"RT_API\s+[^;(]*?\b(rt_\w+)\s*\(", src)))


def test_header_declares_something():
    syms = declared_symbols()
    assert "rt_create" in syms and "rt_post_process" in syms and "rt_generate" in syms and len(syms) >= 10


def test_boundary_header_carries_no_measurement_or_test_entry_points():
    """a host binding of the generation path sees no rt_debug_* / rt_bench_* / rt_profile_* declaration and no
    tune-code table; those live in their own header, which includes the boundary."""
    pub, dbg = declared_symbols(), declared_symbols("rho_tts_amd_debug.h")
    assert not [s for s in pub if s.startswith(("rt_debug_", "rt_bench_", "rt_profile_"))]
    assert dbg and all(s.startswith(("rt_debug_", "rt_bench_", "rt_profile_")) for s in dbg), dbg
    assert not set(pub) & set(dbg)
    assert "rt_debug_tune" not in open(os.path.join(ROOT, "include", "rho_tts_amd.h")).read()
    assert '#include "rho_tts_amd.h"' in open(os.path.join(ROOT, "include", "rho_tts_amd_debug.h")).read()


def test_library_builds_and_exports_every_declared_symbol():
    path = _build.build_native()
    lib = ctypes.CDLL(path)
    for header in _build.PUBLIC_HEADERS:
        missing = [s for s in declared_symbols(header) if not hasattr(lib, s)]
        assert not missing, f"declared in {header} but not exported: {missing}"


def test_abi_version_and_status_strings():
    lib = _native.load_library()
    assert lib.rt_abi_version() == 6
    assert lib.rt_status_string(0) == b"ok"
    assert b"memory" in lib.rt_status_string(_native.RT_ERR_OOM)


def test_struct_layouts_match_header_sizes():
    # rt_post_params: 8 x 4-byte + 4 doubles + 2 x u32 = 72 bytes; rt_post_stats: 3 x i64 + 2 doubles + 4 x i32 = 56
    assert ctypes.sizeof(_native.PostParams) == 72
    assert ctypes.sizeof(_native.PostStats) == 56


def test_capacity_is_pure_host_arithmetic():
    lib = _native.load_library()
    p = _native.make_post_params()
    assert (p.window, p.fade, p.crossfade, p.pause, p.loud_window) == (240, 480, 1200, 2400, 48000)
    lens = (ctypes.c_int64 * 3)(1000, 2000, 3000)
    assert lib.rt_post_capacity(ctypes.byref(p), 3, lens) == 6000 + 2400
    assert lib.rt_post_capacity(ctypes.byref(p), 1, lens) == 1000


def test_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        return
    try:
        _native.Context(0)
    except _native.NativeUnavailable as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("Context() must not succeed without a GPU")
