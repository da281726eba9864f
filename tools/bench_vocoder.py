#!/usr/bin/env python3
"""Codec decoder alone at the bench shape (1.7B preset, 32 items x 44 frames): wall time per call; the target of
`rocprofv3 --pmc ... -- python3 tools/bench_vocoder.py` runs (few dispatches, no decode loop)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rho_tts_amd import config
from rho_tts_amd.engine import Engine

cfg = config.PRESETS["1.7b"]()
eng = Engine(cfg=cfg, model_path=cfg.name, device_ordinal=0, max_batch=32, synthetic=True)
for code in sys.argv[1:]:
    eng.ctx.lib.rt_debug_tune(int(code), 0)
g = torch.Generator().manual_seed(3)
codes = [torch.randint(0, cfg.codec.codebook_size, (44, cfg.n_groups), generator=g) for _ in range(32)]
reps = int(os.environ.get("REPS", "5"))
for rep in range(reps):
    torch.cuda.synchronize(); eng.ctx.synchronize()
    t0 = time.perf_counter()
    w = eng.vocode(codes)
    torch.cuda.synchronize(); eng.ctx.synchronize()
    print(f"rep {rep}: vocode {1e3 * (time.perf_counter() - t0):.2f} ms ({w[0].numel()} samples per item)", flush=True)
