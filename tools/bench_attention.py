#!/usr/bin/env python3
"""Decode attention (attention.hip k_attention<128, 2, true, 16>) against its context: where the 12.9 us of a talker launch go.

  shared prefix of p rows + 13 .. 57 own rows, 32 rows x 8 kv heads x 2 query heads (the 1.7B talker's decode step), 28 layers
  cycled; the same total context WITHOUT a shared prefix slot; the predictor's form (4 waves, <= 16 positions).

The intercept (prefix 0, own 17) is launch + fused q/k/v prologue + cross-wave merge + store; the slope over the prefix length is
what a shared read of the prefix (one pass per kv head instead of one per row) could at best remove."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from rho_tts_amd import _native
ctx = _native.Context(0)
lib = ctx.lib
lib.rt_bench_attention_fused.argtypes = [C.c_void_p] + [C.c_int32] * 9 + [C.POINTER(C.c_double)]
for code in sys.argv[1:]:
    lib.rt_debug_tune(int(code), 0)


def run(M, heads, kvh, d, prefix, own, shared, layers=28, iters=560):
    us = C.c_double()
    rc = lib.rt_bench_attention_fused(ctx.handle, M, heads, kvh, d, prefix, own, shared, layers, iters, C.byref(us))
    return us.value if rc == 0 else float("nan")


print("tune codes:", sys.argv[1:] or "(defaults)", flush=True)
print("talker decode attention, 32 rows x 8 kv heads x 2 q heads x 128, 28 layers cycled: us per launch", flush=True)
print(f"{'prefix':>7} {'own':>4} {'shared us':>10} {'unshared us':>12}", flush=True)
for prefix in (0, 64, 128, 256, 460):
    for own in ((17, 35, 57) if prefix in (0, 460) else (35,)):
        a = run(32, 16, 8, 128, prefix, own, 1)
        b = run(32, 16, 8, 128, prefix, own, 0) if prefix else a
        print(f"{prefix:7d} {own:4d} {a:10.2f} {b:12.2f}", flush=True)
print("one row (a batch of 1), 460 + 35:", f"{run(1, 16, 8, 128, 460, 35, 1):.2f} us", flush=True)
print("predictor form, 32 rows x 8 kv heads, 5 layers cycled:", " ".join(f"own {o}: {run(32, 16, 8, 128, 0, o, 0, layers=5):.2f} us" for o in (2, 8, 16)), flush=True)
