#!/usr/bin/env python3
"""Codec decoder A/B on one box: wall time per call under each set of rt_debug_tune codes given as arguments (comma-separated sets),
and whether every set produces bit-identical waveforms (the switches compared here only change WHEN operands are requested).
usage: python tools/ab_vocoder.py 2600 2601 2602"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rho_tts_amd import config
from rho_tts_amd.engine import Engine

cfg = config.PRESETS["1.7b"]()
eng = Engine(cfg=cfg, model_path=cfg.name, device_ordinal=0, max_batch=32, synthetic=True)
g = torch.Generator().manual_seed(3)
codes = [torch.randint(0, cfg.codec.codebook_size, (44, cfg.n_groups), generator=g) for _ in range(32)]
sets = sys.argv[1:] or ["2601"]
ref = None
for rnd in range(2):
    for st in sets:
        for code in st.split(","):
            assert eng.ctx.lib.rt_debug_tune(int(code), 0) == 0
        ts = []
        for rep in range(4):
            torch.cuda.synchronize(); eng.ctx.synchronize()
            t0 = time.perf_counter()
            w = eng.vocode(codes)
            torch.cuda.synchronize(); eng.ctx.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        same = "" if ref is None else ("  bit-identical to the first set" if all(torch.equal(a, b) for a, b in zip(w, ref)) else "  DIFFERS from the first set")
        if ref is None:
            ref = [x.clone() for x in w]
        print(f"round {rnd} tune [{st}]: vocode {min(ts[1:]):.2f} ms (min of 3), {sum(ts[1:]) / 3:.2f} mean{same}", flush=True)
