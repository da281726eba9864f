"""ctypes binding of ``librho_tts_amd.so`` (the C ABI in ``include/rho_tts_amd.h``).

There is no CPU fallback: if the HIP library is missing or no gfx950 device is
present, every entry point raises.  The oracle under ``oracle/`` is never
imported from here.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import List, Optional, Sequence

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librho_tts_amd.so")

RT_OK, RT_ERR_INVALID, RT_ERR_OOM, RT_ERR_LENGTH, RT_ERR_HIP, RT_ERR_CANCELLED, RT_ERR_STATE, RT_ERR_UNSUPPORTED = range(8)

POST_TRIM_START, POST_TRIM_END, POST_DC, POST_FADE_IN, POST_FADE_OUT, POST_JOIN, POST_LOUDNESS, POST_DECAY = (
    0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40, 0x80)
POST_PIPELINE = 0xFF
STREAM_MEASURE, STREAM_TRIM_START, STREAM_TRIM_END, STREAM_FADE_IN, STREAM_FADE_OUT = 0x01, 0x02, 0x04, 0x08, 0x10


class NativeUnavailable(RuntimeError):
    """The HIP library could not be loaded; the product path has no other implementation."""


class CancelledError(Exception):
    """Raised for RT_ERR_CANCELLED; the provider converts it to the host API's CancelledException."""


class PostParams(C.Structure):
    _fields_ = [("sample_rate", C.c_int32), ("silence_threshold", C.c_float), ("window", C.c_int32),
                ("fade", C.c_int32), ("crossfade", C.c_int32), ("pause", C.c_int32), ("loud_window", C.c_int32),
                ("trim_enabled", C.c_int32), ("target_rms_db", C.c_double), ("max_gain_db", C.c_double),
                ("max_amplitude", C.c_double), ("decay_threshold", C.c_double), ("stages", C.c_uint32),
                ("reserved", C.c_uint32)]


class PostStats(C.Structure):
    _fields_ = [("out_len", C.c_int64), ("first_trim_start", C.c_int64), ("first_trim_end", C.c_int64),
                ("decay_ratio", C.c_double), ("rms_out", C.c_double), ("decay_ok", C.c_int32),
                ("all_silent", C.c_int32), ("fallback_concat", C.c_int32), ("windowed_applied", C.c_int32)]


_lib = None
_lib_lock = threading.Lock()


def load_library(build_if_missing: bool = True) -> C.CDLL:
    """dlopen the in-tree library, building it with hipcc first if it is not there."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        # torch ships its own copies of the HIP/HSA runtime; whichever copy is mapped first owns the GPU, and a
        # second HSA runtime in the process then sees "No HIP GPUs".  Importing torch first makes this library
        # resolve libamdhip64.so.7 to the copy torch already mapped.
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            if not build_if_missing:
                raise NativeUnavailable(f"{LIB_PATH} not found; run `python -m rho_tts_amd._build`")
            from ._build import build_native
            try:
                build_native()
            except Exception as e:  # noqa: BLE001
                raise NativeUnavailable(f"cannot build {LIB_PATH}: {e}") from e
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as e:
            raise NativeUnavailable(f"cannot load {LIB_PATH}: {e}") from e
        _declare(lib)
        _lib = lib
        return lib


def _declare(lib: C.CDLL) -> None:
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    lib.rt_abi_version.restype = C.c_int
    lib.rt_status_string.restype = C.c_char_p
    lib.rt_status_string.argtypes = [C.c_int]
    lib.rt_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.rt_destroy.argtypes = [vp]
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_last_error.argtypes = [vp]
    lib.rt_set_stream.argtypes = [vp, vp]
    lib.rt_synchronize.argtypes = [vp]
    lib.rt_device_info.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(i64), C.POINTER(i64)]
    lib.rt_post_capacity.restype = i64
    lib.rt_post_capacity.argtypes = [C.POINTER(PostParams), i32, C.POINTER(i64)]
    post_args = [vp, C.POINTER(PostParams), i32, C.POINTER(i32), C.POINTER(vp), C.POINTER(i64),
                 C.POINTER(C.c_uint8), C.POINTER(vp), C.POINTER(i64), C.POINTER(PostStats)]
    lib.rt_post_process.argtypes = post_args
    lib.rt_post_process_host.argtypes = post_args
    lib.rt_pcm16.argtypes = [vp, vp, i64, vp]
    lib.rt_stream_chunk.argtypes = [vp, C.POINTER(PostParams), vp, i64, C.c_uint32, C.POINTER(C.c_double), vp, C.POINTER(i64)]
    from . import _native_model
    _native_model.declare(lib)


def raise_for_status(lib, ctx, rc: int, what: str) -> None:
    if rc == RT_OK:
        return
    msg = lib.rt_last_error(ctx).decode("utf-8", "replace") if ctx else ""
    text = f"{what}: {lib.rt_status_string(rc).decode()}" + (f" — {msg}" if msg else "")
    if rc == RT_ERR_INVALID:
        raise ValueError(text)
    if rc == RT_ERR_CANCELLED:
        raise CancelledError(text)
    if rc == RT_ERR_OOM and "out of memory" not in text.lower():
        text = "out of memory: " + text
    if rc == RT_ERR_LENGTH and "length" not in text.lower():
        text = "length: " + text
    raise RuntimeError(text)


class Context:
    """One ``rt_ctx`` (one per process and GPU)."""

    def __init__(self, device_ordinal: int = 0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.rt_create(int(device_ordinal), C.byref(h))
        if rc != RT_OK:
            raise NativeUnavailable(
                f"rt_create(device={device_ordinal}) failed: {self.lib.rt_status_string(rc).decode()} "
                "(a gfx950 / MI355X GPU is required; there is no CPU fallback)")
        self.handle = h
        self.device_ordinal = int(device_ordinal)

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.rt_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def check(self, rc: int, what: str) -> None:
        raise_for_status(self.lib, self.handle, rc, what)

    def set_stream(self, stream_ptr: Optional[int]) -> None:
        self.check(self.lib.rt_set_stream(self.handle, C.c_void_p(stream_ptr or 0)), "rt_set_stream")

    def synchronize(self) -> None:
        self.check(self.lib.rt_synchronize(self.handle), "rt_synchronize")

    def device_info(self) -> dict:
        arch = C.create_string_buffer(64)
        ncu = C.c_int()
        fr, tot = C.c_int64(), C.c_int64()
        self.check(self.lib.rt_device_info(self.handle, arch, 64, C.byref(ncu), C.byref(fr), C.byref(tot)), "rt_device_info")
        return {"arch": arch.value.decode(), "n_cu": ncu.value, "hbm_free": fr.value, "hbm_total": tot.value}

    # ------------------------------------------------------------------ post
    def post_capacity(self, params: PostParams, seg_lens: Sequence[int]) -> int:
        arr = (C.c_int64 * max(1, len(seg_lens)))(*seg_lens)
        return int(self.lib.rt_post_capacity(C.byref(params), len(seg_lens), arr))

    def post_process(self, params: PostParams, items: Sequence[Sequence["object"]],
                     seg_trim: Optional[Sequence[int]] = None):
        """Run the fused kernel over ``items`` (each a list of 1-D float32 torch tensors).

        All tensors must live on the same side: CPU tensors use the host-buffer
        entry point, GPU tensors the device-resident one.  Returns
        ``(outputs, stats)`` with one trimmed output tensor and one ``PostStats``
        per item, on the same device as the inputs.
        """
        import torch

        segs = [s for it in items for s in it]
        n_items, n_seg = len(items), len(segs)
        if n_items == 0:
            return [], []
        on_gpu = any(s.is_cuda for s in segs)
        if on_gpu and not all(s.is_cuda for s in segs):
            raise ValueError("post_process: mixed CPU/GPU segments")
        segs = [s.detach().reshape(-1).to(torch.float32).contiguous() for s in segs]
        first = (C.c_int32 * (n_items + 1))()
        acc = 0
        for i, it in enumerate(items):
            first[i] = acc
            acc += len(it)
        first[n_items] = acc
        lens = (C.c_int64 * max(1, n_seg))(*[int(s.numel()) for s in segs])
        ptrs = (C.c_void_p * max(1, n_seg))(*[C.c_void_p(s.data_ptr() if s.numel() else 0) for s in segs])
        trim = None
        if seg_trim is not None:
            trim = (C.c_uint8 * max(1, n_seg))(*[int(t) for t in seg_trim])
        caps, outs = [], []
        k = 0
        dev = segs[0].device if segs else torch.device("cpu")
        for it in items:
            ls = [int(s.numel()) for s in segs[k:k + len(it)]]
            k += len(it)
            cap = self.post_capacity(params, ls)
            caps.append(cap)
            outs.append(torch.empty(cap, dtype=torch.float32, device=dev))
        ccaps = (C.c_int64 * n_items)(*caps)
        optrs = (C.c_void_p * n_items)(*[C.c_void_p(o.data_ptr()) for o in outs])
        stats = (PostStats * n_items)()
        if on_gpu:
            torch.cuda.current_stream(dev).synchronize()
            fn, name = self.lib.rt_post_process, "rt_post_process"
        else:
            fn, name = self.lib.rt_post_process_host, "rt_post_process_host"
        rc = fn(self.handle, C.byref(params), n_items, first, ptrs, lens, trim, optrs, ccaps, stats)
        self.check(rc, name)
        return [o[: stats[i].out_len] for i, o in enumerate(outs)], [stats[i] for i in range(n_items)]

    def stream_chunk(self, params: PostParams, x, state, first: bool, last: bool):
        """rt_stream_chunk: the per-segment leaves for one chunk of a segment still being decoded.  ``state`` is the segment's
        ``[dc, gain]`` list: written by the first chunk that holds audible audio, applied by the later ones.  A ``first`` chunk
        without an audible frame leaves ``state[1] == 0``: drop its output and pass ``first=True`` again with the next chunk.
        Returns the processed chunk (GPU tensor)."""
        import torch
        x = x.detach().reshape(-1).to(torch.float32).contiguous()
        if not x.is_cuda:
            raise ValueError("stream_chunk expects a GPU tensor")
        flags = (STREAM_MEASURE | STREAM_TRIM_START | STREAM_FADE_IN if first else 0) | (STREAM_TRIM_END | STREAM_FADE_OUT if last else 0)
        out = torch.empty(max(1, x.numel()), dtype=torch.float32, device=x.device)
        st = (C.c_double * 2)(float(state[0]), float(state[1]))
        n_out = C.c_int64()
        torch.cuda.current_stream(x.device).synchronize()
        self.check(self.lib.rt_stream_chunk(self.handle, C.byref(params), C.c_void_p(x.data_ptr() if x.numel() else 0), x.numel(), flags, st,
                                            C.c_void_p(out.data_ptr()), C.byref(n_out)), "rt_stream_chunk")
        state[0], state[1] = float(st[0]), float(st[1])
        return out[: n_out.value]

    def pcm16(self, x):
        import torch
        x = x.detach().reshape(-1).to(torch.float32).contiguous()
        if not x.is_cuda:
            raise ValueError("pcm16 expects a GPU tensor")
        out = torch.empty(x.numel(), dtype=torch.int16, device=x.device)
        torch.cuda.current_stream(x.device).synchronize()
        self.check(self.lib.rt_pcm16(self.handle, C.c_void_p(x.data_ptr()), x.numel(), C.c_void_p(out.data_ptr())), "rt_pcm16")
        self.synchronize()
        return out


def make_post_params(sample_rate: int = 24000, silence_threshold_db: float = -50.0, fade_duration_sec: float = 0.02,
                     crossfade_duration_sec: float = 0.05, inter_sentence_pause_sec: float = 0.1,
                     trim_silence: bool = True, sound_decay_threshold: float = 0.3, stages: int = POST_PIPELINE,
                     target_rms_db: float = -23.0, window_sec: float = 2.0, max_gain_db: float = 18.0,
                     max_amplitude: float = 0.95) -> PostParams:
    """Derive the integer geometry exactly as the reference does (base_tts.py:366-367,420,455,519; qwen.py:293)."""
    p = PostParams()
    p.sample_rate = int(sample_rate)
    p.silence_threshold = 10 ** (silence_threshold_db / 20)
    p.window = int(sample_rate * 0.01)
    p.fade = int(sample_rate * fade_duration_sec)
    p.crossfade = int(sample_rate * crossfade_duration_sec)
    p.pause = int(sample_rate * inter_sentence_pause_sec) if inter_sentence_pause_sec > 0 else 0
    p.loud_window = int(sample_rate * window_sec)
    p.trim_enabled = 1 if trim_silence else 0
    p.target_rms_db = target_rms_db
    p.max_gain_db = max_gain_db
    p.max_amplitude = max_amplitude
    p.decay_threshold = sound_decay_threshold
    p.stages = int(stages)
    return p
