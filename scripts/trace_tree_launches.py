#!/usr/bin/env python3
"""Per-launch durations of the executor's kernels from a rocprofv3 --kernel-trace output directory: the strand schedule is one launch per strand level, so
this is the time of every level of the LAST iteration (usage: trace_tree_launches.py <rocprof output dir> [launches per iteration to show])."""
import csv
import glob
import os
import sys

rows = []
for f in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        if "k_tree" in r.get("Kernel_Name", ""):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void rxhip::tree::", ""), r.get("Grid_Size", "?")))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
t_end = None
for s, e, name, grid in rows[-n:]:
    print(f"{name:48s} grid {grid:>10s}  {(e - s) / 1e3:9.1f} us   gap before {(s - t_end) / 1e3 if t_end else 0:7.1f} us")
    t_end = e
