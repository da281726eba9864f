#!/usr/bin/env python3
"""MFMA-busy per kernel from a rocprofv3 counter pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE).

  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x n_simd)
GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back), so / 8 is the kernel's duration in shader
cycles; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over every SIMD (32 per v_mfma_f32_32x32x16_bf16, 16 per 16x16x32);
n_simd = 256 CUs x 4.  `clock_ghz` = GRBM_GUI_ACTIVE / 8 / duration.
usage: pmc_mfma_summary.py <dir-or-csv> [out.csv]"""
import csv
import os
import re
import sys
from collections import defaultdict


def find(path):
    if os.path.isfile(path):
        return path
    for root, _, files in os.walk(path):
        for f in files:
            if f.endswith("counter_collection.csv"):
                return os.path.join(root, f)
    raise SystemExit("no counter_collection.csv under " + path)


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:70]


def main():
    src = find(sys.argv[1])
    agg = defaultdict(lambda: defaultdict(float))
    seen = set()
    with open(src) as f:
        for r in csv.DictReader(f):
            n = r["Kernel_Name"]
            if "at::native" in n or "rocclr" in n or "k_" not in n:
                continue
            k = short(n)
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"] or 0)
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                agg[k]["_n"] += 1
                agg[k]["_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    w = csv.writer(out)
    w.writerow(["kernel", "dispatches", "profiled_ms_total", "avg_us", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES",
                "mfma_busy_frac", "clock_ghz"])
    for k, c in sorted(agg.items(), key=lambda kv: -kv[1]["_ns"]):
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        frac = busy / (gui / 8 * 1024) if gui else 0.0
        ghz = gui / 8 / c["_ns"] if c["_ns"] else 0.0
        w.writerow([k, int(c["_n"]), round(c["_ns"] / 1e6, 3), round(c["_ns"] / c["_n"] / 1e3, 2), int(busy), int(gui), int(c.get("SQ_BUSY_CU_CYCLES", 0)),
                    int(c.get("SQ_WAVE_CYCLES", 0)), round(frac, 4), round(ghz, 3)])


if __name__ == "__main__":
    main()
