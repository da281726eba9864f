#!/usr/bin/env python3
"""Wall-clock breakdown of one bench step (synchronising after every phase): where the non-kernel time goes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from rho_tts_amd import _native, config
from rho_tts_amd.engine import Engine
from rho_tts_amd.voice import synthetic_reference_clip

cfg = config.PRESETS["1.7b"]()
eng = Engine(cfg=cfg, model_path=cfg.name, device_ordinal=0, max_batch=32, synthetic=True)
for code in sys.argv[1:]:
    eng.ctx.lib.rt_debug_tune(int(code), 0)
texts = bench.sentences(32, 10, seed=789)
clip = synthetic_reference_clip(30.0, cfg.sample_rate, 789)
ref_text = " ".join(bench.WORDS[i % len(bench.WORDS)] for i in range(75))
post = _native.make_post_params(sample_rate=cfg.sample_rate, stages=_native.POST_PIPELINE)


def sync():
    torch.cuda.synchronize()
    eng.ctx.synchronize()


def timed(label, fn, acc):
    sync(); t0 = time.perf_counter(); r = fn(); sync(); acc[label] = acc.get(label, 0.0) + (time.perf_counter() - t0) * 1e3
    return r


for rep in range(4):
    acc = {}
    cond = timed("encode", lambda: eng.conditioning_from_audio(clip, ref_text), acc)
    timed("set_voice", lambda: eng.set_voice(cond), acc)
    codes = timed("generate_codes", lambda: eng.generate_codes(texts, 789, list(range(32))), acc)
    raw = timed("vocode", lambda: eng.vocode(codes), acc)
    outs = timed("post_process", lambda: eng.post_process([[w] for w in raw], post), acc)
    timed("to_host", lambda: [o.cpu() for o in outs[0]], acc)
    print(f"rep {rep}: " + "  ".join(f"{k} {v:.1f} ms" for k, v in acc.items()) + f"  total {sum(acc.values()):.1f} ms", flush=True)
