#!/usr/bin/env python3
"""Column-owner decode GEMM against the number of workgroups: gate/up (SILU) widths around the talker's 12288 (384 workgroups on
256 CUs) and the predictor's 6144 (192), HBM-cold weights, M = 32.  Shows what the uneven 1.5 workgroups per CU cost."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from rho_tts_amd import _native
ctx = _native.Context(0)
lib = ctx.lib
lib.rt_bench_gemm_col.argtypes = [C.c_void_p] + [C.c_int32] * 7 + [C.POINTER(C.c_double), C.POINTER(C.c_int64)]
for K in (2048, 1024):
    for N in (4096, 6144, 8192, 10240, 12288, 14336, 16384):
        mb = N * K * 2 / 1e6
        n_mats = max(2, int(600 / mb) + 1)
        res = []
        for code in (2400, 2401):           # one pair per workgroup / 1.5 pairs per workgroup where the pairs are 1.5x the CUs
            lib.rt_debug_tune(code, 0)
            us = C.c_double()
            st = (C.c_int64 * 8)()
            rc = lib.rt_bench_gemm_col(ctx.handle, 32, N, K, 1, 2, n_mats, 400, C.byref(us), st)
            res.append(f"{us.value:7.2f} us {mb / us.value:5.2f} TB/s (rc {rc})")
        print(f"SILU K={K} N={N:6d} pairs {N // 32:4d}  {mb:6.1f} MB  pairs: {res[0]}   1.5-pair workgroups where they apply: {res[1]}", flush=True)
