#!/usr/bin/env python3
"""FETCH_SIZE of the decode GEMM (k_gemm_col) over one bench.py step, split by what is streamed.

Input: the counter_collection CSV of
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT -- python3 bench.py --steps 1 --warmup 0 \
        --no-cpu-baseline --no-roofline --tune 200,1301
(200 = eager launches, 1301 = a host wait per layer: rocprofv3's counter collection on this image faults once a few hundred
dispatches are queued un-waited - tools/pmc_probe.py flood / syncflood).

The k_gemm_col dispatches of a decode frame come in a fixed order (generate.hip enqueue_a / enqueue_b):
    part A: talker head, mtp projection, predictor pass 0 (5 layers x 4 GEMMs, 64-row launches),
            then 15 x [predictor head g, (5 layers x 4 GEMMs) for g < 14]
    part B: 28 talker layers x 4 GEMMs            (not run after the last frame)
so a dispatch's class follows from its index.  FETCH_SIZE is reported in KiB and, on gfx950, tallies the 128-B requests of
wide streaming reads at 64 B: bytes = value x 1024 x 2 (MI355X_MICROARCH.md, HBM section).  It counts L2 -> fabric requests,
Infinity-Cache hits included, so for the cache-resident predictor it is L2-miss traffic, not HBM traffic.
usage: pmc_step_split.py <counter_collection.csv> [preset=1.7b] [out.json]"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rho_tts_amd import config  # noqa: E402


def main():
    src = sys.argv[1]
    cfg = config.PRESETS[sys.argv[2] if len(sys.argv) > 2 else "1.7b"]()
    t, p, G = cfg.talker, cfg.predictor, cfg.n_groups
    rows = []
    with open(src) as f:
        for r in csv.DictReader(f):
            if "k_gemm_col" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
                rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"]) * 1024 * 2, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    rows.sort()
    per_pass = p.layers * 4
    a_labels = ["talker head+mtp"] * (2 if cfg.has_mtp_proj else 1) + ["predictor layers"] * per_pass
    for g in range(G - 1):
        a_labels.append("predictor heads")
        if g < G - 2:
            a_labels += ["predictor layers"] * per_pass
    b_labels = ["talker layers"] * (t.layers * 4)
    frame = a_labels + b_labels
    def layer_bytes(d):
        return 2 * (d.hidden * (d.q_dim + 2 * d.kv_dim) + d.q_dim * d.hidden + 3 * d.hidden * d.inter)
    algo = {"talker layers": t.layers * layer_bytes(t),
            "talker head+mtp": 2 * (cfg.codec_vocab * t.hidden + (p.hidden * t.hidden if cfg.has_mtp_proj else 0)),
            "predictor layers": (G - 1) * p.layers * layer_bytes(p),
            "predictor heads": 2 * (G - 1) * cfg.predictor_vocab * p.hidden}        # per frame
    acc = {k: [0, 0.0, 0] for k in algo}
    n_b = 0
    for i, (_, fetched, ns) in enumerate(rows):
        k = frame[i % len(frame)]
        acc[k][0] += 1
        acc[k][1] += fetched
        acc[k][2] += ns
        if k == "talker layers":
            n_b += 1
    n_frames_a = acc["predictor heads"][0] / (G - 1)
    n_frames_b = n_b / (t.layers * 4)
    from rho_tts_amd._build import source_hash
    # (build_sha256: bench.py reports these bytes only for this very build)
    out = {"source": os.path.basename(src), "build_sha256": source_hash(), "k_gemm_col_dispatches": len(rows), "frames_part_a": n_frames_a,
           "frames_part_b": n_frames_b, "classes": {}}
    tot_f = tot_a = 0.0
    for k, (n, fetched, ns) in acc.items():
        frames = n_frames_b if k == "talker layers" else n_frames_a
        a = algo[k] * frames
        tot_f += fetched
        tot_a += a
        out["classes"][k] = {"dispatches": n, "fetched_bytes": round(fetched), "algorithmic_bytes": round(a), "fetched_over_algorithmic": round(fetched / a, 4),
                             "fetched_bytes_per_dispatch": round(fetched / n), "profiled_us_per_dispatch": round(ns / n / 1e3, 2)}
    out["all"] = {"fetched_bytes": round(tot_f), "algorithmic_bytes": round(tot_a), "fetched_over_algorithmic": round(tot_f / tot_a, 4),
                  "fetched_bytes_per_dispatch": round(tot_f / len(rows)), "algorithmic_bytes_per_dispatch": round(tot_a / len(rows))}
    txt = json.dumps(out, indent=1)
    print(txt)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
