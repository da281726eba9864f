#!/usr/bin/env python3
"""Diagnostic: teacher-forced logits of a preset on the GPU vs the oracle (f32 and bf16-activation modes), per frame / group."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle.model import OracleModel, Voice
from oracle.sampling import SamplingParams
from rho_tts_amd import _native, config, weights
from rho_tts_amd._native_model import NativeModel, RtSampling
from rho_tts_amd.tokenizer import HashTokenizer
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_model_shapes_gpu import clone_voice, sentences

preset, B, nf = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
secs = float(sys.argv[4]) if len(sys.argv) > 4 else 30.0
torch.set_num_threads(min(32, os.cpu_count() or 8))
cfg = config.PRESETS[preset]()
tok = HashTokenizer(cfg.text_vocab)
ctx = _native.Context(0)
state = weights.synthetic_state(cfg, 789, device="cuda")
nm = NativeModel(ctx, cfg, max_batch=B)
nm.load_state(state)
cpu_state = {k: v.cpu() for k, v in state.items()}
cond = clone_voice(cfg, tok, secs)
v = Voice(cond.language, None, cond.speaker_embed, cond.ref_text_ids, cond.ref_codes)
print("prefix", nm.set_voice(v.language, None, v.speaker_embed, v.ref_text_ids, v.ref_codes))
texts = [tok.encode(t) for t in sentences(B, 10, 789)]
frames = [nf] * B
V0, G1 = cfg.codec.codebook_size, cfg.n_groups - 1
free = None
keep = {}
for bf in (True, False):
    om = OracleModel(cfg, cpu_state, act_bf16=bf)
    tr_o = {}
    with torch.no_grad():
        out = om.generate(v, texts, frames, SamplingParams(), trace=tr_o, share_prefix=True, forced_codes=free)
    if free is None:
        free = out
    codes, tr = nm.generate(texts, frames, RtSampling(0, 1.0, 1, 1.0, 1.0), forced_codes=free, trace=True)
    t_o = torch.stack(tr_o["talker_logits"])[..., :V0]
    t_g = tr["talker"][:nf].cpu()[..., :V0]
    p_o = torch.stack(tr_o["pred_logits"]).view(nf, G1, B, -1)
    p_g = tr["predictor"][:nf].cpu()
    keep[bf] = (t_o, p_o)
    print(f"oracle act_bf16={bf}: talker sigma {float(t_o.std()):.4f} pred sigma {float(p_o.std()):.4f}")
    for t in range(nf):
        et = (t_g[t] - t_o[t]).abs()
        print(f"  frame {t}: talker max {float(et.max()) / float(t_o.std()):.5f} sigma (rms {float(et.pow(2).mean().sqrt()) / float(t_o.std()):.5f}); "
              "pred per group max: " + " ".join(f"{float((p_g[t, g] - p_o[t, g]).abs().max()) / float(p_o.std()):.4f}" for g in range(G1)))
dt, dp = keep[True][0] - keep[False][0], keep[True][1] - keep[False][1]
print(f"oracle bf16 vs oracle f32: talker max {float(dt.abs().max()) / float(t_o.std()):.5f} rms {float(dt.pow(2).mean().sqrt()) / float(t_o.std()):.5f}; "
      f"pred max {float(dp.abs().max()) / float(p_o.std()):.5f} rms {float(dp.pow(2).mean().sqrt()) / float(p_o.std()):.5f}")
for code in (100,):
    nm.lib.rt_debug_tune(code, 0)
    _, tr2 = nm.generate(texts, frames, RtSampling(0, 1.0, 1, 1.0, 1.0), forced_codes=free, trace=True)
    nm.lib.rt_debug_tune(101, 0)
    d = (tr2["talker"][:nf].cpu()[..., :V0] - t_g).abs().max()
    print(f"legacy vs column path talker max diff {float(d) / float(t_o.std()):.5f} sigma")
