#!/usr/bin/env python3
"""What would single-plane fp16 conv operands cost the codec decoder's waveform?  CPU only: the float32 oracle against itself with
the inputs of chosen convs rounded to fp16 (the GPU path multiplies hi + lo bf16 planes, ~16 mantissa bits, two MFMAs per weight
fragment; one fp16 plane would halve the matrix work and the plane traffic).  Test infrastructure - runs the oracle, not the product."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rho_tts_amd import config, weights
import oracle.model as om
cfg = config.PRESETS["0.6b"]()
state = weights.synthetic_state(cfg, 789, device="cpu")
state = {k: v for k, v in state.items() if k.startswith("codec.")}
M = om.OracleModel(cfg, state)
g = torch.Generator().manual_seed(9)
Q = cfg.codec.num_quantizers
codes = torch.randint(0, cfg.codec.codebook_size, (44, Q), generator=g)
torch.set_num_threads(8)
with torch.no_grad():
    ref = M.code2wav(codes.T[None])[0]
orig_c, orig_t = om.causal_conv1d, om.causal_trans_conv1d
h16 = lambda v: v.half().float()
def run(sel, name):
    # sel(kind, cin) -> bool: round this conv's input to fp16
    def c(x, w, b, dilation=1, groups=1):
        if groups == 1 and sel("k%d" % w.shape[-1], x.shape[1]): x = h16(x)
        return orig_c(x, w, b, dilation, groups)
    def t(x, w, b, stride):
        if sel("t", x.shape[1]): x = h16(x)
        return orig_t(x, w, b, stride)
    om.causal_conv1d, om.causal_trans_conv1d = c, t
    with torch.no_grad():
        w = M.code2wav(codes.T[None])[0]
    om.causal_conv1d, om.causal_trans_conv1d = orig_c, orig_t
    print(f"{name:44s}: rmse {float((w - ref).pow(2).mean().sqrt()):.3e}", flush=True)
run(lambda k, c: k == "t", "transposed convs only")
run(lambda k, c: k == "k7", "k7 convs only (incl. first / last conv)")
run(lambda k, c: k == "k1", "1x1 convs only")
run(lambda k, c: c == 96, "96-channel inputs only (stage 4 units + final)")
run(lambda k, c: c == 192, "192-channel inputs (stage 3 units + trans into 4)")
run(lambda k, c: c in (96, 192), "96 + 192")
run(lambda k, c: c >= 384, ">= 384")
run(lambda k, c: k == "k1" and c in (96, 192), "1x1 convs of stages 3, 4")
run(lambda k, c: k == "k7" and c in (96, 192), "k7 convs of stages 3, 4")
