#!/usr/bin/env python3
"""Durations of the launches of one kernel in a rocprofv3 --kernel-trace output directory, largest first.
usage: surface_launches.py <dir> <kernel name substring> [how many]"""
import csv
import glob
import os
import sys

d, key = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 12
rows = []
for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"]:
            rows.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"][:60], int(r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", 0))))
rows.sort(reverse=True)
print(f"{key}: {len(rows)} launches, total {sum(r[0] for r in rows) / 1e6:.3f} ms")
for dur, name, grid in rows[:n]:
    print(f"  {dur / 1e6:9.3f} ms  grid {grid:>12}  {name}")
