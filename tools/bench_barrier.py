#!/usr/bin/env python3
"""Grid-barrier microbenchmark (persistent kernel, all workgroups co-resident) on the GPU box."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from rho_tts_amd import _native
ctx = _native.Context(0)
lib = ctx.lib
lib.rt_bench_grid_barrier.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
for wgs, threads in ((8, 512), (64, 512), (128, 512), (256, 256), (256, 512), (256, 1024), (512, 512)):
    for mode in (0, 1, 2):
        us, ab = C.c_double(), C.c_int32()
        rc = lib.rt_bench_grid_barrier(ctx.handle, wgs, threads, 2000, mode, C.byref(us), C.byref(ab))
        print(f"{wgs:4d} WGs x {threads:4d} threads mode {mode}: {us.value:6.2f} us per barrier (rc {rc}, flags {ab.value})", flush=True)

# pair hand-off ping-pong: 2 KB + a flag from one workgroup to ONE other, microseconds per hand-off (modes 3 / 4 of the same hook)
for wgs in (16, 256):
    for mode, what in ((3, "partner b ^ 1 (another XCD)"), (4, "partner b ^ 8 (the same XCD)")):
        us, ab = C.c_double(), C.c_int32()
        rc = lib.rt_bench_grid_barrier(ctx.handle, wgs, 512, 2000, mode, C.byref(us), C.byref(ab))
        print(f"{wgs:4d} WGs x  512 threads, {what}: {us.value:6.2f} us per hand-off (rc {rc}, flags {ab.value})", flush=True)
