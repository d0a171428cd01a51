"""Noise-VMP engine (d = 4, 1024 chains x T = 10^4): ms per VMP iteration over the number of time segments of the per-chain sweep."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
import numpy as np, rxhip
from rxhip import workloads
mdl = workloads.c1_model()
T, C, iters = 10000, 1024, 6
y = workloads.generate_batch(mdl, T, C, seed0=4242, threads=16)
for seg in (0, 4, 8, 16, 32, 64, 128, 256):
    with rxhip.LGSSMNoiseEngine(mdl["A"], mdl["B"], mdl["P"], mdl["m0"], mdl["V0"], T, 5.0, np.eye(4), n_chains=C, segments=seg) as eng:
        eng.set_data(y)
        eng.run(iters, True)
        t0 = time.perf_counter()
        eng.run(iters, True)
        ms = (time.perf_counter() - t0) / iters * 1e3
        eng.set_profiling(True); eng.reset_kernel_times(); eng.run(2, True); eng.sync()
        kt = {k: round(v["ms_avg"], 3) for k, v in eng.kernel_times().items() if v["launches"]}
        print(f"segments={seg:4d} -> {eng.schedule()}  {ms:.3f} ms per iteration  {kt}  fe {eng.free_energy()[-1]:.6f}")
