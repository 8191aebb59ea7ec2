#!/usr/bin/env python3
"""Kernel time vs idle time between kernels in a rocprofv3 kernel-trace CSV, per repetition of a kernel sequence.
usage: trace_gaps.py <kernel_trace.csv> <kernels per repetition> [skip first repetitions=5]"""
import csv
import sys

import numpy as np

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
per, skip = int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 5
rows = rows[len(rows) % per:]                      # (set-up kernels in front of the repetitions)
reps = [rows[i:i + per] for i in range(0, len(rows), per)][skip:]
busy = [sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rep) / 1e3 for rep in reps]
span = [(int(rep[-1]["End_Timestamp"]) - int(rep[0]["Start_Timestamp"])) / 1e3 for rep in reps]
gaps = [[(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(rep, rep[1:])] for rep in reps]
print(f"{len(reps)} repetitions of {per} kernels: span {np.median(span):.1f} us, kernels {np.median(busy):.1f} us, "
      f"idle between kernels {np.median([sum(g) for g in gaps]):.1f} us (median gap {np.median([x for g in gaps for x in g]):.2f} us, "
      f"max {max(x for g in gaps for x in g):.2f} us)")
