"""One shared-model batch on the MFMA path for rocprofv3 --kernel-trace --stats:  python scripts/prof_mid.py d dy chains T"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
import numpy as np
import rxhip
from rxhip import workloads
d, dy, C, T = (int(v) for v in sys.argv[1:5])
m = workloads.random_model(d, dy, seed=d)
y = workloads.generate_batch(m, T, min(C, 8), seed0=1)
y = np.tile(y, (1, C // min(C, 8), 1))
with rxhip.LGSSMEngine(m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"], T=T, n_chains=C) as eng:
    eng.set_data(y)
    eng.run(2, True)
    t = time.perf_counter()
    for _ in range(10):
        eng.run(1, True)
    eng.free_energy()
    print(f"d={d} dy={dy} chains={C} T={T}: {(time.perf_counter() - t) / 10 * 1e3:.3f} ms per sweep; schedule {eng.schedule()}")
