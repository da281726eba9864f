#!/usr/bin/env python3
"""Reduce rocprofv3 counter_collection CSVs to one line per (kernel, grid): mean of every counter over the dispatches.
usage: pmc_by_kernel.py OUT_DIR [name-filter]"""
import csv, glob, os, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if flt and flt not in k:
            continue
        mm = re.search(r"(k_\w+(?:<[^>]*>)?)", k)
        key = (mm.group(1) if mm else k[:60], r.get("Grid_Size", ""))
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for v in acc.values() for c in v})
print("kernel,grid,calls," + ",".join(names))
for key, v in sorted(acc.items(), key=lambda kv: -max(len(x) for x in kv[1].values())):
    n = max(len(x) for x in v.values())
    print(f'"{key[0]}",{key[1]},{n},' + ",".join(f"{sum(v[c]) / len(v[c]):.4g}" if c in v else "" for c in names))
