#!/usr/bin/env python3
"""Does the codec decoder overlap with the decode chain on one GPU?  Two engines (own context, stream, pool) in one process,
one thread decoding batches, one thread vocoding: wall time of both together against each alone."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from rho_tts_amd import config
from rho_tts_amd.engine import Engine
from rho_tts_amd.voice import synthetic_reference_clip

cfg = config.PRESETS["1.7b"]()
# optional CU partition: overlap_probe.py 192  -> decode engine on CUs [0, 192), vocoder engine on [192, 256)
split = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if split:
    os.environ["RHO_TTS_AMD_CU_MASK"] = f"0:{split}"
A = Engine(cfg=cfg, model_path=cfg.name, device_ordinal=0, max_batch=32, synthetic=True)
if split:
    os.environ["RHO_TTS_AMD_CU_MASK"] = f"{split}:{256 - split}"
B = Engine(cfg=cfg, model_path=cfg.name, device_ordinal=0, max_batch=32, synthetic=True)
os.environ.pop("RHO_TTS_AMD_CU_MASK", None)
texts = bench.sentences(32, 10, seed=789)
clip = synthetic_reference_clip(30.0, cfg.sample_rate, 789)
ref_text = " ".join(bench.WORDS[i % len(bench.WORDS)] for i in range(75))
A.set_voice_from_audio(clip, ref_text)
codes = A.generate_codes(texts, 789, list(range(32)))
B.vocode(codes)
N = 6


def sync():
    torch.cuda.synchronize(); A.ctx.synchronize(); B.ctx.synchronize()


def dec():
    for _ in range(N):
        A.generate_codes(texts, 789, list(range(32)))
    A.ctx.synchronize()


def voc():
    for _ in range(N):
        B.vocode(codes)
    B.ctx.synchronize()


for fn, name in ((dec, "decode alone"), (voc, "vocode alone")):
    sync(); t0 = time.perf_counter(); fn(); sync()
    print(f"{name}: {1e3 * (time.perf_counter() - t0) / N:.1f} ms per batch", flush=True)
sync(); t0 = time.perf_counter()
t1, t2 = threading.Thread(target=dec), threading.Thread(target=voc)
t1.start(); t2.start(); t1.join(); t2.join(); sync()
print(f"both together: {1e3 * (time.perf_counter() - t0) / N:.1f} ms per (decode + vocode) pair", flush=True)
