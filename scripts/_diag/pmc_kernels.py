#!/usr/bin/env python3
"""Per-kernel means of a rocprofv3 --pmc counter_collection.csv (largest grid of each kernel name only).
usage: pmc_kernels.py <dir-with-counter_collection.csv> [name-filter]"""
import collections
import csv
import glob
import sys

path = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
flt = sys.argv[2] if len(sys.argv) > 2 else "nrm::"
rows = [r for r in csv.DictReader(open(path)) if flt in r["Kernel_Name"]]
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void nrm::", "").replace("nrm::", "")   # noqa: E731
big = collections.defaultdict(int)
for r in rows:
    big[name(r)] = max(big[name(r)], int(r["Grid_Size"]))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if int(r["Grid_Size"]) == big[name(r)]:
        agg[name(r)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0.0)
    print(k, "grid", big[k])
    for n, v in sorted(m.items()):
        print(f"    {n:28s} {v:14.4e}" + (f"   {v / wc:6.3f} of SQ_WAVE_CYCLES" if wc and n.startswith("SQ_") and n != "SQ_WAVE_CYCLES" else ""))
