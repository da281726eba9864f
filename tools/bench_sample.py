#!/usr/bin/env python3
"""Sampler microbenchmark on the GPU box: average launch time and the in-kernel phase stamps (100 MHz wall clock)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from rho_tts_amd import _native
from rho_tts_amd._native_model import RtSampling
ctx = _native.Context(0)
lib = ctx.lib
lib.rt_bench_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(RtSampling), C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
for V in (2048, 3072, 1024):
    x = (torch.randn(32, V, device="cuda") * 3).contiguous()
    for label, sp in (("greedy", RtSampling(0, 0.9, 50, 1.0, 1.0)), ("top-k 50", RtSampling(1, 0.9, 50, 1.0, 1.0)),
                      ("top-k 50 top-p .9", RtSampling(1, 0.9, 50, 0.9, 1.0)), ("top-k 8", RtSampling(1, 0.9, 8, 1.0, 1.0))):
        us = C.c_double()
        st = (C.c_int64 * 8)()
        rc = lib.rt_bench_sample(ctx.handle, x.data_ptr(), 32, V, C.byref(sp), 200, C.byref(us), st)
        d = [(st[i + 1] - st[i]) * 0.01 if st[i + 1] and st[i] else float("nan") for i in range(6)]
        d[1] = float("nan")
        print(f"V {V:5d} {label:18s}: {us.value:6.2f} us/launch (rc {rc})  phases us: " +
              "  ".join(f"{n} {t:.2f}" for n, t in zip(("load+prep", "-", "wave-topk", "final-topk", "sort"), d)) +
              f"  walk {(st[6]-st[5])*0.01 if st[5] else float('nan'):.2f}", flush=True)

# ---- rows per launch (VERDICT r3 #5a: could the LAST-ARRIVING workgroup of the head GEMM do the sampling itself?).  A launch over M
# rows runs M workgroups of 512 threads side by side; one workgroup sampling 32 rows would run them one after the other.  The
# time of a 1-row launch is what ONE row costs on one workgroup (launch + load + select + draw); 32 x (that - the launch floor) is
# the serial cost the fusion would put on the last arriver, against one 32-workgroup launch.
print("\nrows per launch, V 2048, top-k 50 (one workgroup per row):", flush=True)
x = (torch.randn(64, 2048, device="cuda") * 3).contiguous()
sp = RtSampling(1, 0.9, 50, 1.0, 1.0)
for M in (1, 2, 4, 8, 16, 32, 64):
    us = C.c_double()
    st = (C.c_int64 * 8)()
    rc = lib.rt_bench_sample(ctx.handle, x.data_ptr(), M, 2048, C.byref(sp), 400, C.byref(us), st)
    inside = (st[6] - st[0]) * 0.01 if st[6] and st[0] else float("nan")
    print(f"  M {M:3d}: {us.value:6.2f} us/launch (rc {rc}); first instruction -> last store of workgroup 0: {inside:.2f} us", flush=True)
