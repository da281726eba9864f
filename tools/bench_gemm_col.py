#!/usr/bin/env python3
"""Microbenchmark of the column-owner decode GEMM (back-to-back launches, HBM-cold weights)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from rho_tts_amd import _native
ctx = _native.Context(0)
lib = ctx.lib
lib.rt_bench_gemm_col.argtypes = [C.c_void_p] + [C.c_int32] * 7 + [C.POINTER(C.c_double), C.POINTER(C.c_int64)]
shapes = [(32, 4096, 2048, 1, 0, "talker qkv  NORM STORE"), (32, 2048, 2048, 0, 1, "talker o    RESID"), (32, 12288, 2048, 1, 2, "talker gu   NORM SILU"),
          (32, 2048, 6144, 0, 1, "talker down RESID"), (32, 3072, 2048, 1, 0, "talker head NORM STORE"), (32, 4096, 1024, 1, 0, "pred qkv    NORM STORE"),
          (32, 1024, 2048, 0, 1, "pred o      RESID"), (32, 6144, 1024, 1, 2, "pred gu     NORM SILU"), (32, 1024, 3072, 0, 1, "pred down   RESID"),
          (32, 2048, 1024, 1, 0, "pred head   NORM STORE"), (32, 4096, 2048, 0, 0, "qkv plain STORE"), (8, 4096, 2048, 1, 0, "qkv M=8 NORM")]
if len(sys.argv) > 1 and sys.argv[1] == "hot":      # same matrix every launch, cacheable loads: L2 / Infinity-Cache resident weights
    for M, N, K, norm, epi, name in shapes:
        mb = N * K * 2 / 1e6
        us = C.c_double()
        st = (C.c_int64 * 8)()
        rc = lib.rt_bench_gemm_col(ctx.handle, M, N, K, norm | 2, epi, 1, 400, C.byref(us), st)
        ph = " ".join(f"{(st[i + 1] - st[i]) * 0.01:.2f}" for i in range(5))
        print(f"{name:26s} {mb:6.1f} MB  HOT {us.value:7.2f} us [{ph}]", flush=True)
    sys.exit(0)
for M, N, K, norm, epi, name in shapes:
    mb = N * K * 2 / 1e6
    n_mats = max(2, int(600 / mb) + 1)
    res = []
    for code in (501, 500):             # forced unsplit, automatic sub-tile split
        lib.rt_debug_tune(code, 0)
        us = C.c_double()
        st = (C.c_int64 * 8)()
        rc = lib.rt_bench_gemm_col(ctx.handle, M, N, K, norm, epi, n_mats, 400, C.byref(us), st)
        ph = " ".join(f"{(st[i + 1] - st[i]) * 0.01:.2f}" for i in range(5))
        res.append(f"{us.value:7.2f} us {mb / us.value:5.2f} TB/s (rc {rc}) [issue, first chunk, rest, reduce, epilogue: {ph}]")
    print(f"{name:26s} {mb:6.1f} MB  unsplit {res[0]}   auto split {res[1]}", flush=True)
