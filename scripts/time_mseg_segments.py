"""Sweep time of the masked time-parallel schedule (dense_mseg_kernels.hpp) over the number of segments and the kind of boundary
recursion (RXHIP_MSEG_SCAN = sequential | log), 10 % of the observations missing:
   python scripts/time_mseg_segments.py d chains T S [S ...]        (S = 0: the cost model's choice)"""
import os, sys, time
os.environ["RXHIP_TEST_HOOKS"] = "1"   # the schedule switches below are test hooks (include/rxhip.h "Environment")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
import numpy as np
import rxhip
from rxhip import workloads

d, C, T = (int(v) for v in sys.argv[1:4])
mdl = workloads.random_model(d, d, seed=d)
y = workloads.generate_batch(mdl, T, C, seed0=1)
y[np.random.default_rng(0).random((T, C)) < 0.1] = np.nan
for S in (int(v) for v in sys.argv[4:]):
    row = []
    for mode in ("sequential", "log"):
        os.environ["RXHIP_MSEG_SCAN"] = mode
        with rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=C, allow_missing=True, segments=S) as eng:
            eng.set_data(y)
            eng.run(free_energy=True)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                eng.run(free_energy=True)
                ts.append((time.perf_counter() - t0) * 1e3)
            row.append(f"{mode} {sorted(ts)[2]:.2f} ms (S = {eng.schedule()['segments']}, L = {eng.schedule()['segment_len']})")
    print(f"d={d} chains={C} T={T} segments={S}: " + " | ".join(row), flush=True)
os.environ.pop("RXHIP_MSEG_SCAN", None)
