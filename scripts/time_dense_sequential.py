"""Sweep time of the sequential schedule (csrc/gseq_kernels.hpp: `missing` observations / per-step constants at d > 4)
next to the time-parallel MFMA schedule on the same fully observed batch."""
import os, sys, time
os.environ["RXHIP_TEST_HOOKS"] = "1"   # the schedule switches below are test hooks (include/rxhip.h "Environment")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
import numpy as np
import rxhip
from rxhip import workloads


def timed(eng, n=5):
    """median of n individually timed sweeps (a process that has just released a large arena sees one-off stalls of 50–80 ms)"""
    eng.run(free_energy=True)
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        eng.run(free_energy=True)
        ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]


for d, C, T in ((8, 1024, 1000), (16, 512, 1000), (32, 256, 500), (64, 256, 200), (64, 1, 2000)):
    mdl = workloads.random_model(d, d, seed=d)
    y = workloads.generate_batch(mdl, T, C, seed0=1)
    args = (mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"])
    with rxhip.LGSSMEngine(*args, T=T, n_chains=C) as eng:
        eng.set_data(y)
        ms_par = timed(eng)
    ym = y.copy()
    ym[np.random.default_rng(0).random((T, C)) < 0.1] = np.nan
    os.environ.pop("RXHIP_GSEQ", None)
    with rxhip.LGSSMEngine(*args, T=T, n_chains=C, allow_missing=True) as eng:
        eng.set_data(ym)
        ms_mp = timed(eng)
        kt = {k: round(v["ms_avg"], 3) for k, v in eng.kernel_times().items() if v["launches"]}
    os.environ["RXHIP_GSEQ"] = "1"
    with rxhip.LGSSMEngine(*args, T=T, n_chains=C, allow_missing=True) as eng:
        eng.set_data(ym)
        ms_seq = timed(eng, 1)
    os.environ.pop("RXHIP_GSEQ", None)
    print(f"d=dy={d} chains={C} T={T}: fully observed {ms_par:.2f} ms | 10 % missing, parallel in time (dense_mseg_kernels) {ms_mp:.2f} ms = "
          f"{ms_mp / ms_par:.1f}x | 10 % missing, sequential (gseq) {ms_seq:.2f} ms = {ms_seq / ms_par:.0f}x", flush=True)
