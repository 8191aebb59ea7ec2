#!/usr/bin/env python3
"""kernel_stats csv -> per-step table.  usage: summarize_stats.py <csv> <steps>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); steps = float(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total GPU busy per step: {tot / steps / 1e6:.2f} ms")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 28]:
    name = r["Name"].replace("void ", "").split("(")[0][:44]
    print(f"{name:44s} calls/step {float(r['Calls']) / steps:6.1f}  avg {float(r['AverageNs']) / 1e3:9.1f} us  per-step {float(r['TotalDurationNs']) / steps / 1e6:7.3f} ms  {float(r['Percentage']):5.2f}%")
