"""One `missing`-data engine on the time-parallel MFMA schedule (dense_mseg_kernels.hpp), for rocprofv3 --kernel-trace --stats:
   python scripts/prof_mseg.py d chains T"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
import numpy as np
import rxhip
from rxhip import workloads

d, C, T = (int(v) for v in sys.argv[1:4])
mdl = workloads.random_model(d, d, seed=d)
y = workloads.generate_batch(mdl, T, C, seed0=1)
y[np.random.default_rng(0).random((T, C)) < 0.1] = np.nan
with rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=C, allow_missing=True) as eng:
    eng.set_data(y)
    eng.run(free_energy=True)
    t0 = time.perf_counter()
    for _ in range(3):
        eng.run(free_energy=True)
    print(f"d={d} chains={C} T={T}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per sweep; schedule {eng.schedule()}")
