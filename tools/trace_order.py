#!/usr/bin/env python3
"""Per-launch listing of a rocprofv3 kernel trace: the launches of the LAST of N identical passes, in start order.
usage: trace_order.py <kernel_trace.csv> <passes>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = len(rows) // n
last = rows[len(rows) - per:]
t0 = int(last[0]["Start_Timestamp"])
tot = 0
for r in last:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot += d
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  {d / 1e3:8.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')):>4}  {r['Kernel_Name'][:90]}")
print(f"launches {len(last)}  busy {tot / 1e3:.1f} us  span {(int(last[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
