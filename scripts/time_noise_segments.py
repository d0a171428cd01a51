"""Noise VMP (bench extra_noise_vmp shape) by number of time segments: ms per VMP iteration."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
import rxhip  # noqa: E402
from rxhip import workloads  # noqa: E402

mdl = workloads.c1_model()
T, C, iters, dy = 10000, 1024, 10, 4
y = workloads.generate_batch(mdl, T, C, seed0=4242, threads=min(32, os.cpu_count() or 1))
for seg in (0, 32, 48, 64, 96, 127, 160, 200, 256, 320, 400):
    eng = rxhip.LGSSMNoiseEngine(mdl["A"], mdl["B"], mdl["P"], mdl["m0"], mdl["V0"], T, dy + 1.0, np.eye(dy), n_chains=C, segments=seg)
    eng.set_data(y)
    eng.run(iters, True)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        eng.run_async(iters, True)
        eng.sync()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    print(f"segments={seg:4d} -> {eng.schedule()}  {best / iters:.4f} ms per iteration  fe_last {eng.free_energy()[-1]:.6f}", flush=True)
    eng.close()
