"""C2 with 10 % missing observations (d = 4, T = 1e5, 1024 chains): the sweep by number of time segments."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, rxhip
from rxhip import workloads
import bench
mdl = workloads.c1_model()
T, C = 100000, 1024
y = workloads.generate_batch(mdl, T, C, seed0=42, threads=32)
rng = np.random.default_rng(0)
y[rng.random((T, C)) < 0.1] = np.nan
for seg in (0, 64, 96, 128, 192, 256, 384):
    with rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=C, allow_missing=True, segments=seg) as eng:
        eng.set_data(y)
        eng.run(1, True)
        ms, kt = bench.timed_sweeps(eng, 5, 1)
        print(f"segments={seg:4d} -> {eng.schedule()}  {ms:.3f} ms {kt}", flush=True)
