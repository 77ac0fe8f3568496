#!/usr/bin/env python3
"""Print a per-step summary of a `rocprofv3 --kernel-trace --stats --output-format csv` kernel_stats.csv.
usage: scripts/rocprof_summary.py <kernel_stats.csv> <steps_profiled>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.2f} ms over {steps} steps = {tot / 1e6 / steps:.2f} ms/step")
groups = {"nrm:: (this repo's HIP kernels)": 0.0, "Cijk_ (rocBLAS/Tensile GEMM)": 0.0, "at::native / rocprim / other": 0.0}
for r in rows:
    n, t = r["Name"], float(r["TotalDurationNs"])
    key = "nrm:: (this repo's HIP kernels)" if "nrm::" in n else "Cijk_ (rocBLAS/Tensile GEMM)" if n.startswith("Cijk_") else "at::native / rocprim / other"
    groups[key] += t
for k, v in groups.items():
    print(f"  {k:40s} {v / 1e6 / steps:8.2f} ms/step  {100 * v / tot:5.1f}%")
for r in rows[:14]:
    print(f"{r['Name'][:90]:90s} calls/step={int(r['Calls']) / steps:6.1f} avg_us={float(r['AverageNs']) / 1e3:9.1f} {float(r['Percentage']):5.1f}%")
