#!/usr/bin/env python3
"""Print the last N dispatches whose kernel name matches a regex, in start order, from a rocprofv3 kernel-trace CSV.
usage: list_trace.py <dir-or-csv> <regex> [N]"""
import csv, os, re, sys


def find(path):
    if os.path.isfile(path):
        return path
    for root, _, files in os.walk(path):
        for f in files:
            if f.endswith("kernel_trace.csv"):
                return os.path.join(root, f)
    raise SystemExit("no kernel_trace.csv under " + path)


rows = []
rx = re.compile(sys.argv[2])
with open(find(sys.argv[1])) as f:
    for r in csv.DictReader(f):
        n = r.get("Kernel_Name") or ""
        if rx.search(n):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Grid_Size_X"), r.get("Grid_Size_Y"), r.get("Grid_Size_Z")))
rows.sort()
N = int(sys.argv[3]) if len(sys.argv) > 3 else 40
t0 = rows[-N][0] if len(rows) >= N else (rows[0][0] if rows else 0)
for s, e, n, gx, gy, gz in rows[-N:]:
    short = re.sub(r"\(anonymous namespace\)::|^void ", "", n)[:48]
    print(f"{(s - t0) / 1e3:10.1f} us  {(e - s) / 1e3:9.1f} us  grid {gx}x{gy}x{gz}  {short}")
