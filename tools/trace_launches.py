#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel name the average duration of the 1st, 2nd, ... launch within a
repeating step pattern (staged plans launch the same kernel twice per traversal with different grids).
usage: tools/trace_launches.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size", 0) or 0),
                         int(r.get("VGPR_Count", 0) or 0), int(r.get("Scratch_Size", 0) or 0)))
rows.sort()
by = defaultdict(list)
for s, e, name, grid, vg, sc in rows:
    short = name.split("(")[0][-70:]
    by[(short, grid, vg, sc)].append((e - s) / 1e3)
print("%-72s %9s %5s %6s %6s %9s %9s" % ("kernel", "grid", "vgpr", "scr", "n", "avg_us", "min_us"))
for (name, grid, vg, sc), v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    tail = v[len(v) // 4:]  # skip warm-up launches
    print("%-72s %9d %5d %6d %6d %9.2f %9.2f" % (name, grid, vg, sc, len(v), sum(tail) / len(tail), min(v)))
# gaps between consecutive kernels of the stream (host / dispatch overhead inside a step)
gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
if gaps:
    gaps_us = sorted(g / 1e3 for g in gaps)
    print("inter-kernel gaps: median %.2f us, p90 %.2f us" % (gaps_us[len(gaps_us) // 2], gaps_us[int(len(gaps_us) * 0.9)]))
