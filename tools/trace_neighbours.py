#!/usr/bin/env python3
"""Who launches it?  For every occurrence of a kernel (substring) in a rocprofv3 kernel-trace CSV: the kernels right before and after it,
counted over the whole trace.  usage: trace_neighbours.py <kernel_trace.csv> <substring> [context=1]"""
import csv
import sys
from collections import Counter

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
pat, ctx = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 1
name = lambda r: r["Kernel_Name"].replace("void ", "").split("(")[0][:60]
seen = Counter()
for i, r in enumerate(rows):
    if pat in r["Kernel_Name"]:
        before = " <- ".join(name(rows[j]) for j in range(max(i - ctx, 0), i))
        after = " -> ".join(name(rows[j]) for j in range(i + 1, min(i + 1 + ctx, len(rows))))
        grid = r.get("Grid_Size", r.get("Grid_Size_X", "?"))
        seen[(before, after, grid)] += 1
for (before, after, grid), n in seen.most_common(25):
    print(f"{n:5d} x  [{before}]  >>{pat} grid {grid}<<  [{after}]")
