#!/usr/bin/env python3
"""HBM GB/s and MFMA-busy per kernel FAMILY of one bench.py step, against the chip's peaks (north_star: "rocprof HBM GB/s and
MFMA-busy reported against chip peak").  Three rocprofv3 runs of the same build feed it:

  time   rocprofv3 --kernel-trace --output-format csv -d T -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline
         (3 un-instrumented steps: graph replays, the durations the headline number is made of)
  bytes  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d F -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline \
             --no-roofline --tune 200,1301          (FETCH_SIZE is KiB; x 1024 x 2 on gfx950: MI355X_MICROARCH.md, HBM section)
  mfma   the same with --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE   (busy = MFMA cycles / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs))

usage: family_summary.py --trace T --trace-steps 3 --fetch F --mfma M [--json profiles/r03_pmc_fetch_bench_step.json]
Writes the table to stdout and, with --json, a "families" section into that file (bench.py prints it as extra.kernel_families when
the library's hash matches)."""
import argparse
import csv
import json
import os
import re
from collections import defaultdict

FAMILIES = [
    ("decode GEMM (k_gemm_col)", r"k_gemm_col"),
    ("decode attention", r"k_attention<\d+, \d+, true|k_attn_prefix_mfma<4"),
    ("sampler + embedding", r"k_sample|k_embed_rowsq|k_rowsq|k_frame_inc|k_gather|k_norm_tiled"),
    ("prompt prefill GEMM", r"k_gemm_mid|k_gemm_skinny|k_add_rmsnorm|k_silu_mul|k_reduce_slabs|k_qkv_post"),
    ("prompt prefill attention", r"k_attn_prefix_mfma<8|k_attention<\d+, \d+, false, \d+, false|k_tile_prefix"),
    ("codec decoder convs", r"k_conv_win|k_gemm_tiled<false, true|k_final_conv|k_dwconv|k_code_embed"),
    ("codec / encoder transformer + GEMMs", r"k_gemm_tiled<true|k_gemm_tiled<false, false|k_attention<\d+, \d+, false, \d+, true"),
    ("audio encoder", r"k_enc_|k_rvq|k_stats_pool|k_gemv"),
    ("post-processing", r"k_post_item|k_pcm16|k_stream_chunk"),
]


def family(name):
    if "at::native" in name or "rocclr" in name or "k_" not in name or "k_pack_weight" in name:
        return None
    for fam, pat in FAMILIES:
        if re.search(pat, name):
            return fam
    return "other"


def find(path, suffix):
    if os.path.isfile(path):
        return path
    for root, _, files in os.walk(path):
        for f in files:
            if f.endswith(suffix):
                return os.path.join(root, f)
    raise SystemExit(f"no {suffix} under {path}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace", required=True)
    ap.add_argument("--trace-steps", type=int, default=3)
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--mfma", required=True)
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    t_ns = defaultdict(float)
    with open(find(a.trace, "kernel_trace.csv")) as f:
        for r in csv.DictReader(f):
            fam = family(r["Kernel_Name"])
            if fam:
                t_ns[fam] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    fetched = defaultdict(float)
    with open(find(a.fetch, "counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            fam = family(r["Kernel_Name"])
            if fam and r["Counter_Name"] == "FETCH_SIZE":
                fetched[fam] += float(r["Counter_Value"]) * 1024 * 2
    mf = defaultdict(lambda: [0.0, 0.0])
    with open(find(a.mfma, "counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            fam = family(r["Kernel_Name"])
            if not fam:
                continue
            if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
                mf[fam][0] += float(r["Counter_Value"])
            elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                mf[fam][1] += float(r["Counter_Value"]) / 8.0 * 1024.0
    out = {}
    print(f"{'family':42s} {'ms/step':>8s} {'GB fetched':>10s} {'HBM GB/s':>9s} {'of 8 TB/s':>9s} {'MFMA busy':>9s}")
    for fam in [f for f, _ in FAMILIES] + ["other"]:
        if fam not in t_ns:
            continue
        ms = t_ns[fam] / 1e6 / a.trace_steps
        gb = fetched.get(fam, 0.0) / 1e9
        gbs = gb / (ms * 1e-3) if ms > 0 else 0.0
        busy = mf[fam][0] / mf[fam][1] if mf[fam][1] > 0 else 0.0
        out[fam] = {"ms_per_step": round(ms, 2), "fetched_gb_per_step": round(gb, 2), "hbm_gb_s": round(gbs, 1), "hbm_frac_of_peak": round(gbs / 8000.0, 4),
                    "mfma_busy": round(busy, 4)}
        print(f"{fam:42s} {ms:8.2f} {gb:10.2f} {gbs:9.1f} {gbs / 8000.0:9.3f} {busy:9.3f}")
    if a.json:
        with open(a.json) as f:
            js = json.load(f)
        js["families"] = out
        js["families_note"] = ("per bench step: time from an un-instrumented kernel trace (graph replays), bytes = FETCH_SIZE x 1024 x 2 of a counter "
                               "pass (L2 -> fabric requests, Infinity-Cache hits included), MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024)")
        with open(a.json, "w") as f:
            f.write(json.dumps(js, indent=1) + "\n")


if __name__ == "__main__":
    main()
