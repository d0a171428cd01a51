"""Shared-model batches on the model / data split (d = 64 × 64 chains × T = 1000, d = 8 × 1024 × 1000, d = 32 × 256 × 1000): sweep time and
first-touch time (create + data + first sweep of a never-seen model) over the number of segments."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
import numpy as np, rxhip
from rxhip import workloads
for d, dy, C, T in ((64, 64, 64, 1000), (32, 32, 256, 1000), (8, 4, 1024, 1000)):
    m0 = workloads.random_model(d, dy, seed=d)
    y = workloads.generate_batch(m0, T, 8, seed0=1)
    y = np.tile(y, (1, C // 8, 1))
    for rep, seg in enumerate((0, 8, 16, 32, 64, 128)):
        m = dict(m0)
        m["P"] = m0["P"] * (1.0 + 0.003 * (rep + 1))   # never seen before
        t0 = time.perf_counter()
        eng = rxhip.LGSSMEngine(m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"], T=T, n_chains=C, segments=seg)
        eng.set_data(y)
        eng.run(1, True)
        first = (time.perf_counter() - t0) * 1e3
        for _ in range(3): eng.run_async(1, True)
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(10): eng.run_async(1, True)
        eng.sync()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        print(f"d={d} chains={C} T={T} segments asked {seg:4d} -> {eng.schedule()}  first touch {first:7.2f} ms  sweep {ms:6.3f} ms  stages {eng.create_stages()}", flush=True)
        eng.close()
