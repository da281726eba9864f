#!/usr/bin/env python3
"""One launch of one kernel variant, for probing which launch shape trips `rocprofv3 --pmc` (round-1 abort in
gpurun_out/prof9/bench_pmc.log: host SIGSEGV inside the profiler's dispatch interception at the first 1024-thread
fused-attention launch).  Usage: rocprofv3 --pmc FETCH_SIZE -d DIR -- python3 tools/pmc_probe.py VARIANT"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rho_tts_amd import _native

variant = sys.argv[1]
ctx = _native.Context(0)
lib = ctx.lib
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None


def fused(max_pos, M=32, d=128, heads=16, kvh=8):
    slots = M + 1
    qkv = torch.randn(M, (heads + 2 * kvh) * d, device="cuda")
    w = torch.ones(d, device="cuda")
    cs = torch.ones(max_pos, d // 2, device="cuda")
    k = torch.zeros(slots, kvh, max_pos, d, dtype=torch.bfloat16, device="cuda")
    v = torch.zeros_like(k)
    slot = torch.arange(M, dtype=torch.int32, device="cuda")
    pos = torch.full((M,), min(max_pos - 1, 500), dtype=torch.int32, device="cuda")
    out = torch.zeros(M, heads * d, dtype=torch.bfloat16, device="cuda")
    torch.cuda.synchronize()
    ctx.check(lib.rt_debug_attention_fused(ctx.handle, p(qkv), M, heads, kvh, d, p(w), p(w), 1e-6, p(cs), p(cs), p(slot), p(pos), 0, p(k), p(v), slots,
                                           max_pos, -1, 0, p(out)), "fused")


def unfused(max_pos, M=11, d=128, heads=16, kvh=8):
    q = torch.randn(M, heads, d, device="cuda")
    k = torch.zeros(3, kvh, max_pos, d, dtype=torch.bfloat16, device="cuda")
    v = torch.zeros_like(k)
    slot = torch.zeros(M, dtype=torch.int32, device="cuda")
    pos = torch.full((M,), max_pos - 1, dtype=torch.int32, device="cuda")
    out = torch.zeros(M, heads * d, dtype=torch.bfloat16, device="cuda")
    torch.cuda.synchronize()
    ctx.check(lib.rt_debug_attention(ctx.handle, p(q), M, heads, kvh, d, p(slot), p(pos), 0, p(k), p(v), 3, max_pos, p(out)), "unfused")


if variant == "fused4":
    fused(17)
elif variant == "fused16":
    fused(1024)
elif variant == "unfused16":
    unfused(70)
elif variant == "unfused4":
    unfused(64)
elif variant == "post":
    x = torch.randn(48000, device="cuda") * 0.1
    ctx.post_process(_native.make_post_params(), [[x]])
elif variant.startswith("syncflood"):    # the same number of launches, but the host waits for the stream every 400
    n = int(variant[9:] or 20000)
    us = C.c_double()
    lib.rt_bench_launch.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
    for _ in range(n // 800):
        ctx.check(lib.rt_bench_launch(ctx.handle, 256, 400, 0, 1, C.byref(us)), "rt_bench_launch")     # (400 warm + 400 timed, each followed by a sync)
elif variant.startswith("flood"):        # N dependent trivial launches with no host sync in between: how many may be in flight?
    n = int(variant[5:] or 4000)
    us = C.c_double()
    lib.rt_bench_launch.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
    ctx.check(lib.rt_bench_launch(ctx.handle, 256, n, 0, 1, C.byref(us)), "rt_bench_launch")
else:
    raise SystemExit("unknown variant")
ctx.synchronize()
print("probe", variant, "ok", flush=True)
