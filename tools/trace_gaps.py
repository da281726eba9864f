#!/usr/bin/env python3
"""Idle gaps between consecutive kernels of a rocprofv3 kernel trace: histogram plus the most frequent (previous kernel ->
next kernel) pairs among the larger gaps.  usage: trace_gaps.py <dir-or-csv> [min_gap_us]"""
import csv, os, re, sys
from collections import Counter, defaultdict


def find(path):
    if os.path.isfile(path):
        return path
    for root, _, files in os.walk(path):
        for f in files:
            if f.endswith("kernel_trace.csv"):
                return os.path.join(root, f)
    raise SystemExit("no kernel_trace.csv under " + path)


def short(n):
    n = re.sub(r"\(anonymous namespace\)::|^void ", "", n)
    m = re.match(r"([\w:]+(<[^(]*>)?)", n)
    return (m.group(1) if m else n)[:40]


rows = []
with open(find(sys.argv[1])) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r.get("Kernel_Name") or "")))
rows.sort()
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
hist = Counter()
pairs = defaultdict(lambda: [0, 0.0])
busy = 0
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    gap = (s1 - e0) / 1e3
    busy += e0 - s0
    b = "<0" if gap < 0 else ("0-1" if gap < 1 else ("1-2" if gap < 2 else ("2-4" if gap < 4 else ("4-10" if gap < 10 else ("10-50" if gap < 50 else ">50")))))
    hist[b] += 1
    if thr <= gap < 1000:
        pairs[(n0, n1)][0] += 1
        pairs[(n0, n1)][1] += gap
print("gap histogram (us):", dict(hist))
print("kernel busy ms:", round(busy / 1e6, 1), " span ms:", round((rows[-1][1] - rows[0][0]) / 1e6, 1))
for (a, b), (c, t) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:15]:
    print(f"{c:6d} x  avg {t / c:7.1f} us  total {t / 1e3:7.2f} ms   {a}  ->  {b}")

# positions of the larger gaps inside one decode frame (from one k_frame_inc to the next)
idx = [i for i, r in enumerate(rows) if r[2].startswith("k_frame_inc")]
if len(idx) > 12:
    a, b = idx[-10], idx[-9]
    pos = [(i - a, round((rows[i + 1][0] - rows[i][1]) / 1e3, 1), rows[i][2], rows[i + 1][2]) for i in range(a, b) if (rows[i + 1][0] - rows[i][1]) / 1e3 >= thr]
    print(f"one frame = {b - a} kernels, {round((rows[b][1] - rows[a][1]) / 1e3, 1)} us; gaps >= {thr} us at kernel index:")
    print("  " + "  ".join(f"{p}:{g}" for p, g, _, _ in pos))
    tot = [0.0, 0.0]
    for f0, f1 in zip(idx[-12:-1], idx[-11:]):
        g = sum(max(0.0, (rows[i + 1][0] - rows[i][1]) / 1e3) for i in range(f0, f1))
        tot[0] += g
        tot[1] += (rows[f1][1] - rows[f0][1]) / 1e3
    print(f"last 11 frames: idle between kernels {tot[0] / 11:.0f} us of {tot[1] / 11:.0f} us per frame")
