#!/usr/bin/env python3
"""Python-side cost of one bench step: cProfile of step() (GPU work included as time spent inside the ctypes calls)."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from rho_tts_amd import _native, config
from rho_tts_amd.engine import Engine
from rho_tts_amd.voice import synthetic_reference_clip
cfg = config.PRESETS["1.7b"]()
eng = Engine(cfg=cfg, model_path=cfg.name, device_ordinal=0, max_batch=32, synthetic=True)
texts = bench.sentences(32, 10, seed=789)
clip = synthetic_reference_clip(30.0, cfg.sample_rate, 789)
ref_text = " ".join(bench.WORDS[i % len(bench.WORDS)] for i in range(75))
post = _native.make_post_params(sample_rate=cfg.sample_rate, stages=_native.POST_PIPELINE)


def step():
    eng.set_voice_from_audio(clip, ref_text)
    raw = eng.synthesize(texts, seed=789, item_ids=list(range(32)))
    outs, stats = eng.post_process([[w] for w in raw], post)
    return [o.cpu() for o in outs]


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(5):
    step()
pr.disable()
print(f"{(time.perf_counter() - t0) / 5 * 1e3:.1f} ms per step")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:6000])
