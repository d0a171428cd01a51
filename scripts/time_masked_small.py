"""Masked batches that fill the chip at small state dimensions: sweep + free energy with 10 % missing, in-wave kernels (d ≤ 8) and MFMA kernels,
next to the fully observed sweep of the same batch."""
import os, sys, time
os.environ["RXHIP_TEST_HOOKS"] = "1"   # the schedule switches below are test hooks (include/rxhip.h "Environment")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
import numpy as np, rxhip
from rxhip import workloads
def timed(eng, n=10):
    for _ in range(3): eng.run_async(1, True)
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(n): eng.run_async(1, True)
    eng.sync()
    return (time.perf_counter() - t0) / n * 1e3
for d, dy, C, T in ((8, 4, 1024, 1000), (8, 4, 4096, 1000), (6, 6, 1024, 1000), (16, 8, 512, 1000)):
    m = workloads.random_model(d, dy, seed=d)
    y = np.tile(workloads.generate_batch(m, T, 8, seed0=1), (1, C // 8, 1))
    ym = y.copy(); ym[np.random.default_rng(0).random((T, C)) < 0.1] = np.nan
    args = (m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"])
    with rxhip.LGSSMEngine(*args, T=T, n_chains=C) as eng:
        eng.set_data(y); full = timed(eng)
    row = [f"d={d} dy={dy} chains={C} T={T}: fully observed {full:.3f} ms"]
    for w8 in ("1", "0"):
        os.environ["RXHIP_WAVE8"] = w8
        with rxhip.LGSSMEngine(*args, T=T, n_chains=C, allow_missing=True) as eng:
            eng.set_data(ym); ms = timed(eng, 5)
            row.append(f"10% missing RXHIP_WAVE8={w8}: {ms:.3f} ms ({ms / full:.1f}x, schedule {eng.schedule()})")
    print(" | ".join(row), flush=True)
