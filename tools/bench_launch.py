#!/usr/bin/env python3
"""Launch-floor microbenchmark (eager vs hipGraph) on the GPU box."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from rho_tts_amd import _native
ctx = _native.Context(0)
lib = ctx.lib
lib.rt_bench_launch.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
for wgs in (1, 32, 256, 1024):
    for graph in (0, 1):
        us = C.c_double()
        rc = lib.rt_bench_launch(ctx.handle, wgs, 1000, graph, 20, C.byref(us))
        print(f"grid {wgs:5d} WGs  {'graph' if graph else 'eager'}: {us.value:6.2f} us per dependent launch (rc {rc})", flush=True)
