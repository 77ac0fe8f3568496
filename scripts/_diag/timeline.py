#!/usr/bin/env python3
"""Timeline of ONE step from a rocprofv3 --kernel-trace CSV (the window between the last two Adam launches): GPU busy time (union of
kernel intervals), idle time, overlap, the largest gaps and the per-kernel-family sums.  usage: timeline.py <kernel_trace.csv> [n_gaps]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ngap = int(sys.argv[2]) if len(sys.argv) > 2 else 15
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows), key=lambda e: e[0])
adam = [i for i, e in enumerate(ev) if "adam_dev_kernel" in e[2]]
if len(adam) < 2:
    sys.exit("need two optimizer launches in the trace")
lo, hi = adam[-2] + 1, adam[-1] + 1
win = ev[lo:hi]
t0, t1 = ev[adam[-2]][1], win[-1][1]
short = lambda n: n.split("(")[0].replace("void ", "").replace("nrm::", "")[:60]      # noqa: E731
busy, cur_end, gaps = 0, t0, []
for s, e, n, q in win:
    if s > cur_end:
        gaps.append((s - cur_end, short(n)))
        busy += e - s
        cur_end = e
    elif e > cur_end:
        busy += e - cur_end
        cur_end = e
tot = sum(e - s for s, e, _, _ in win)
print(f"step window {1e-3 * (t1 - t0):.1f} us: {len(win)} launches, sum of durations {1e-3 * tot:.1f} us, GPU busy (union) {1e-3 * busy:.1f} us, "
      f"idle {1e-3 * (t1 - t0 - busy):.1f} us, overlapped {1e-3 * (tot - busy):.1f} us, queues {sorted({q for *_, q in win})}")
print("largest gaps (idle before kernel):")
for g, n in sorted(gaps, reverse=True)[:ngap]:
    print(f"  {1e-3 * g:7.1f} us  before {n}")
print(f"gaps: {len(gaps)} total {1e-3 * sum(g for g, _ in gaps):.1f} us; < 2us: {sum(1 for g, _ in gaps if g < 2000)}, 2-5us: {sum(1 for g, _ in gaps if 2000 <= g < 5000)}, >= 5us: {sum(1 for g, _ in gaps if g >= 5000)}")
fam = collections.defaultdict(lambda: [0, 0])
for s, e, n, _ in win:
    fam[short(n)][0] += 1
    fam[short(n)][1] += e - s
for n, (c, d) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {c:3d} x {n:60s} {1e-3 * d:8.1f} us")
if len(sys.argv) > 3:                       # full listing
    for s, e, n, q in win:
        print(f"{1e-3 * (s - t0):9.1f} +{1e-3 * (e - s):7.1f} q{q} {short(n)}")
