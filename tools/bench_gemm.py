#!/usr/bin/env python3
"""Microbenchmark of the decode GEMM variants on the GPU box (weights cycled through > 512 MB so they stream from HBM)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from rho_tts_amd import _native

ctx = _native.Context(0)
lib = ctx.lib
lib.rt_bench_gemm_skinny.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
shapes = [(32, 12288, 2048, "talker gate/up"), (32, 4096, 2048, "talker qkv"), (32, 2048, 2048, "talker o"), (32, 2048, 6144, "talker down"),
          (32, 3072, 2048, "talker head"), (32, 6144, 1024, "pred gate/up"), (32, 4096, 1024, "pred qkv"), (32, 1024, 2048, "pred o"),
          (32, 1024, 3072, "pred down"), (64, 4096, 1024, "pred qkv M=64")]
print("variant waves/cu | " + " | ".join(s[3] for s in shapes))
for variant in (0, 1, 2):
    for wpc in (4, 8, 16):
        lib.rt_debug_tune(variant, wpc)
        cells = []
        for M, N, K, name in shapes:
            mb = N * K * 2 / 1e6
            n_mats = max(2, int(600 / mb) + 1)
            us, sp = C.c_double(), C.c_int32()
            rc = lib.rt_bench_gemm_skinny(ctx.handle, M, N, K, 0, n_mats, 400, C.byref(us), C.byref(sp))
            cells.append(f"{us.value:6.2f}us S{sp.value:<2d} {mb / us.value / 1e3 * 1e3:5.2f}TB/s" if rc == 0 else "err")
        print(f"{variant} {wpc:2d} | " + " | ".join(cells), flush=True)
