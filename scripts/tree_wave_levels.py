#!/usr/bin/env python3
"""Where a sweep of the LDS-staged executor kernels goes, by op type.  Joins the per-launch durations of a rocprofv3 --kernel-trace of ONE schedule
(RXHIP_TREE_MODE=0: a launch per level) with the schedule dump (RXHIP_TREE_DUMP) of the same engine, and solves durations ≈ Σ count(op) · cost(op) in the
least-squares sense.

    RXHIP_TEST_HOOKS=1 RXHIP_TREE_MODE=0 RXHIP_TREE_DUMP=gpurun_out/lv.txt rocprofv3 --kernel-trace -d gpurun_out/lv -- python scripts/prof_tree_wave.py 16 64 4096 1
    python scripts/tree_wave_levels.py gpurun_out/lv gpurun_out/lv.txt
"""
import csv
import glob
import os
import sys

import numpy as np

NAMES = {1: "DERIVE_MUL", 2: "DERIVE_ADD", 3: "LEAF", 4: "NOISE", 5: "MUL_OUT", 6: "MUL_IN", 7: "ADD_OUT", 8: "ADD_IN", 9: "SHIFT", 10: "PRODUCT", 11: "MARGINAL",
         12: "FE_NOISE2", 13: "FE_NOISE1", 14: "FE_NOISE0", 15: "FE_ENT", 16: "FE_ADD2", 17: "SUM_TERMS", 18: "PREC_UPDATE", 19: "FE_NOISE2M", 20: "MARG_PUSH",
         21: "FE_NOISE_MF"}
MASK = 1 | 2 | 4 | 8 | 1024 | 2048   # forms of the inputs / output, image marginals
levels = []
for line in open(sys.argv[2]):
    parts = line.split()
    ops = [tuple(int(x) for x in q.split(":")) for q in parts[1:] if ":" in q]
    levels.append(ops)
levels = [l for l in levels if l and not all(o == 20 for o, _ in l)]   # (the level of stored image marginals runs only when marginals are read)
rows = []
for f in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        if "k_wave_ops" in r.get("Kernel_Name", "") or "k_tile_ops" in r.get("Kernel_Name", ""):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
n = len(levels)
assert len(rows) >= n and len(rows) % n == 0, (len(rows), n)
last = rows[-n:]
dur = np.array([(e - s) / 1e3 for s, e in last])
gaps = np.array([last[i + 1][0] - last[i][1] for i in range(n - 1)]) / 1e3
keys = sorted({(o, f & MASK) for l in levels for o, f in l})
col = {k: i for i, k in enumerate(keys)}
A = np.zeros((n, len(keys) + 1))
for i, l in enumerate(levels):
    for o, f in l:
        A[i, col[(o, f & MASK)]] += 1
    A[i, -1] = 1.0
x, *_ = np.linalg.lstsq(A, dur, rcond=None)
print(f"{n} launches, {dur.sum() / 1e3:.3f} ms in kernels, {gaps.sum() / 1e3:.3f} ms between them; fit residual {np.abs(A @ x - dur).sum() / dur.sum():.3f}")
tot = A.sum(0)
for k, i in sorted(col.items(), key=lambda kv: -x[kv[1]] * tot[kv[1]]):
    print(f"  {NAMES.get(k[0], k[0]):12s} flags {k[1]:5d}  × {int(tot[i]):5d}   {x[i]:8.2f} us each   {x[i] * tot[i] / 1e3:8.3f} ms")
print(f"  per launch {x[-1]:8.2f} us  × {n}  {x[-1] * n / 1e3:8.3f} ms")

# PMC passes of the same command in the same directory (rocprofv3 --pmc …, no trace): the counters per launch, fitted the same way → per ITEM (÷ replicas)
if len(sys.argv) > 3:
    R = float(sys.argv[3])
    per = {}
    for f in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if "k_wave_ops" in r.get("Kernel_Name", "") or "k_tile_ops" in r.get("Kernel_Name", ""):
                per.setdefault(r["Counter_Name"], {}).setdefault(int(r["Dispatch_Id"]), 0.0)
                per[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    fits = {}
    for name, by_id in sorted(per.items()):
        vals = [by_id[k] for k in sorted(by_id)][-n:]
        if len(vals) != n:
            continue
        fits[name], *_ = np.linalg.lstsq(A, np.array(vals), rcond=None)
    print("per item: " + " ".join(f"{c:>20s}" for c in fits))
    for k, i in sorted(col.items(), key=lambda kv: -x[kv[1]] * tot[kv[1]]):
        print(f"  {NAMES.get(k[0], k[0]):12s} flags {k[1]:5d} " + " ".join(f"{fits[c][i] / R:20.1f}" for c in fits))
