#!/usr/bin/env python3
"""Time of the conditioning front-end (rt_voice_encode) on the benchmark's 30-s reference clip."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rho_tts_amd import config
from rho_tts_amd.engine import Engine
from rho_tts_amd.voice import synthetic_reference_clip
cfg = config.PRESETS[sys.argv[1] if len(sys.argv) > 1 else "1.7b"]()
eng = Engine(cfg=cfg, model_path=cfg.name, device_ordinal=0, max_batch=32, synthetic=True)
for secs in (30.0, 10.0, 3.0):
    clip = synthetic_reference_clip(secs, cfg.sample_rate, 789)
    eng.model.encode_voice(clip)
    torch.cuda.synchronize()
    t = []
    for _ in range(5):
        t0 = time.perf_counter()
        codes, spk = eng.model.encode_voice(clip)
        t.append((time.perf_counter() - t0) * 1e3)
    print(f"{secs:5.1f} s clip -> {codes.shape[0]} frames: {min(t):.2f} ms (median {sorted(t)[2]:.2f})", flush=True)
