"""d = 8 × 1024 chains × T = 1000 with 10 % of the observations missing (one segment per chain: the in-wave kernels of csrc/dense8_kernels.hpp) —
the driver the round-4 profile passes run under rocprofv3."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
import numpy as np, rxhip
from rxhip import workloads
d, dy, C, T = 8, 4, 1024, 1000
m = workloads.random_model(d, dy, seed=d)
y = np.tile(workloads.generate_batch(m, T, 8, seed0=1), (1, C // 8, 1))
y[np.random.default_rng(0).random((T, C)) < 0.1] = np.nan
with rxhip.LGSSMEngine(m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"], T=T, n_chains=C, allow_missing=True) as eng:
    eng.set_data(y)
    for _ in range(2): eng.run(1, True)
    t0 = time.perf_counter()
    for _ in range(5): eng.run_async(1, True)
    eng.sync()
    print({"workload": f"d={d} dy={dy} chains={C} T={T}, 10 % missing", "ms_per_sweep": (time.perf_counter() - t0) / 5 * 1e3, "schedule": eng.schedule()})
