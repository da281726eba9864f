#!/usr/bin/env python3
"""Aggregate a rocprofv3 kernel-trace CSV by (kernel, grid, workgroup): calls, average and total duration.
usage: summarize_trace.py <dir-or-csv> [out.csv]"""
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
            if f.endswith("kernel_trace.csv"):
                return os.path.join(root, f)
    raise SystemExit("no kernel_trace.csv under " + path)


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:60]


def main():
    src = find(sys.argv[1])
    agg = defaultdict(lambda: [0, 0])
    with open(src) as f:
        for r in csv.DictReader(f):
            n = r.get("Kernel_Name") or r.get("Name")
            if "at::native" in n or "rocclr" in n:
                continue
            dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            g = (r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
            wg = r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))
            k = (short(n), "x".join(x for x in g if x), wg)
            agg[k][0] += 1
            agg[k][1] += dur
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for _, v in rows)
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    w = csv.writer(out)
    w.writerow(["kernel", "grid(threads)", "wg", "calls", "avg_us", "total_ms", "pct"])
    for (n, g, wg), (c, d) in rows[:60]:
        w.writerow([n, g, wg, c, round(d / c / 1e3, 2), round(d / 1e6, 2), round(100.0 * d / tot, 2)])


if __name__ == "__main__":
    main()
