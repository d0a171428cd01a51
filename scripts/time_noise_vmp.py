"""The chain with an unknown observation-noise precision (bench.py extra_noise_vmp): ms per VMP iteration and the per-kernel device times."""
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
eng = rxhip.LGSSMNoiseEngine(mdl["A"], mdl["B"], mdl["P"], mdl["m0"], mdl["V0"], T, dy + 1.0, np.eye(dy), n_chains=C)
eng.set_data(y)
eng.run(iters, True)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    eng.run_async(iters, True)
    eng.sync()
    best = min(best, (time.perf_counter() - t0) * 1e3)
print("ms per iteration", round(best / iters, 4), "schedule", eng.schedule())
eng.set_profiling(True)
eng.reset_kernel_times()
eng.run(iters, True)
kt = eng.kernel_times()
tot = 0.0
for k, v in kt.items():
    if v["launches"]:
        print(f"  {k:20s} launches {v['launches']:4d}  avg {v['ms_avg']:.4f} ms  per iteration {v['ms_avg'] * v['launches'] / iters:.4f} ms")
        tot += v["ms_avg"] * v["launches"] / iters
print("sum per iteration", round(tot, 4))
eng.close()
