"""Shared-model batches on the MFMA path: sweep time with the model / data split (dense_split_kernels.hpp) and without
(RXHIP_DENSE_SPLIT=0), per-kernel averages from the engine's HIP events."""
import os, sys, time
os.environ["RXHIP_TEST_HOOKS"] = "1"   # the schedule switches below are test hooks (include/rxhip.h "Environment")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
import numpy as np
import rxhip
from rxhip import workloads

for d, dy, C, T in ((8, 4, 1024, 1000), (8, 8, 1024, 1000), (16, 16, 512, 1000), (32, 32, 128, 1000), (64, 64, 64, 1000), (64, 64, 8, 10000)):
    m = workloads.random_model(d, dy, seed=d)
    y = workloads.generate_batch(m, T, min(C, 8), seed0=1)
    y = np.tile(y, (1, C // min(C, 8), 1))
    line = f"d={d} dy={dy} chains={C} T={T}:"
    for mode in ("0", "1"):
        os.environ["RXHIP_DENSE_SPLIT"] = mode
        t0 = time.perf_counter()
        with rxhip.LGSSMEngine(m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"], T=T, n_chains=C) as eng:
            eng.set_data(y)
            eng.run(1, True); eng.free_energy()
            first = (time.perf_counter() - t0) * 1e3
            eng.set_profiling(True)
            t = time.perf_counter()
            for _ in range(5):
                eng.run(1, True)
            eng.free_energy()
            dt = (time.perf_counter() - t) / 5
            kt = {k: round(v["ms_avg"], 3) for k, v in eng.kernel_times().items() if v["launches"]}
        line += f"\n    split={mode}: {dt*1e3:.3f} ms/sweep = {T*C/dt:.3e} steps/s (create + data + first sweep {first:.1f} ms) {kt}"
    print(line, flush=True)
