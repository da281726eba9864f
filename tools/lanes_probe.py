#!/usr/bin/env python3
"""Do two PREDICTOR chains overlap?  The 1.7B preset with a ONE-layer talker (the frame is then ~95 % predictor) decoded as one
chain of 32 rows against two lanes of 16 rows on their own streams (rt_debug_tune 2301 + 402): what a predictor-only split of
the real frame could gain, without building it."""
import dataclasses, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from rho_tts_amd import config
from rho_tts_amd.engine import Engine
from rho_tts_amd.voice import synthetic_reference_clip

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = config.PRESETS["1.7b"]()
cfg = dataclasses.replace(cfg, talker=dataclasses.replace(cfg.talker, layers=layers))
eng = Engine(cfg=cfg, model_path=cfg.name, device_ordinal=0, max_batch=32, synthetic=True)
texts = bench.sentences(32, 10, seed=789)
clip = synthetic_reference_clip(30.0, cfg.sample_rate, 789)
ref_text = " ".join(bench.WORDS[i % len(bench.WORDS)] for i in range(75))
eng.set_voice_from_audio(clip, ref_text)
lib = eng.ctx.lib


def run(codes, name, n=6):
    for c in codes:
        lib.rt_debug_tune(c, 0)
    eng.generate_codes(texts, 789, list(range(32)))
    torch.cuda.synchronize(); eng.ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        eng.generate_codes(texts, 789, list(range(32)))
    eng.ctx.synchronize()
    print(f"talker layers {layers}: {name}: {1e3 * (time.perf_counter() - t0) / n:.1f} ms per batch of 32 x 44 frames", flush=True)


for rep in range(2):
    run((2300, 401), "one chain of 32 rows")
    run((2301, 402), "two lanes of 16 rows")
    run((2301, 404), "four lanes of 8 rows")
run((2300, 401), "one chain of 32 rows")
