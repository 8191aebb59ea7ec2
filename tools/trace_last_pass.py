#!/usr/bin/env python3
"""Per-launch listing of the LAST pass in a rocprofv3 kernel trace, a pass starting at the last launch whose name contains <marker>
(e.g. k_stem_pool_bf16 for the bf16 forward): start offset, duration, grid, kernel.
usage: trace_last_pass.py <kernel_trace.csv> <marker>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
last = rows[idx[-1]:]
t0 = int(last[0]["Start_Timestamp"])
busy = 0
for r in last:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    busy += d
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  {d / 1e3:8.1f} us  grid {r['Grid_Size_X']:>8}  {r['Kernel_Name'].replace('void ', '').split('(')[0][:70]}")
print(f"launches {len(last)}  busy {busy / 1e3:.1f} us  span {(int(last[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
