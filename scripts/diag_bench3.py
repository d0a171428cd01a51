import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench, rxhip
from rxhip import workloads
for i in range(4):
    r = bench.extra_c3(0)
    print("extra_c3 call", i, round(r["ms_per_step"], 3), r["kernels_ms_avg"], round(r["filter_ms_per_step"], 3), flush=True)
mdl = workloads.c3_model()
y = workloads.generate_batch(mdl, 10000, 1, seed0=6400)
for i in range(4):
    eng = rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=10000, n_chains=1)
    eng.set_data(y); eng.run(1, True)
    a = bench.timed_sweeps(eng, 20, 3)
    b = bench.timed_sweeps(eng, 10, 2, filter_run=True)
    print("no throwaway, smooth then filter", i, round(a[0], 3), round(b[0], 3), hex(eng.stream() or 0), flush=True)
    eng.close()
for i in range(4):
    eng = rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=10000, n_chains=1)
    eng.set_data(y); eng.run(1, True)
    a = bench.timed_sweeps(eng, 20, 3)
    print("no throwaway, smooth only", i, round(a[0], 3), hex(eng.stream() or 0), flush=True)
    eng.close()
