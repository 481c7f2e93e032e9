#!/usr/bin/env python3
"""Gaps in a rocprofv3 kernel trace: idle intervals of the device (no kernel running) longer than a threshold, with the kernels on either side."""
import csv
import glob
import sys

d, thr_us = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
rows.sort()
print("kernels", len(rows))
# keep the last 60 % of the trace (steady state of the last run)
t0 = rows[int(len(rows) * 0.55)][0]
rows = [r for r in rows if r[0] >= t0]
busy_end, gaps, busy = rows[0][1], [], 0
prev = rows[0]
for r in rows[1:]:
    if r[0] > busy_end:
        g = (r[0] - busy_end) / 1e3
        if g >= thr_us:
            gaps.append((g, prev[2], r[2]))
    busy_end = max(busy_end, r[1])
    prev = r if r[1] >= busy_end else prev
span = (rows[-1][1] - rows[0][0]) / 1e6
tot_gap = sum(g for g, _, _ in gaps) / 1e3
print(f"span {span:.1f} ms, gaps >= {thr_us} us: {len(gaps)} totalling {tot_gap:.1f} ms")
from collections import Counter
c = Counter((a, b) for _, a, b in gaps)
for (a, b), n in c.most_common(12):
    tot = sum(g for g, x, y in gaps if (x, y) == (a, b))
    print(f"{n:4d} x  avg {tot / n:8.1f} us   after [{a}]  before [{b}]")
