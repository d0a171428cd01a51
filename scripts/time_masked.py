"""Sweep time of the masked (`missing` anywhere) and per-step-constant schedules at batch scale (d = dy = 4, 1024 chains):
in-lane segment elements (k_seg_elements) against one sequential segment per chain (RXHIP_ONE_SEGMENT=1)."""
import os
os.environ["RXHIP_TEST_HOOKS"] = "1"   # the schedule switches below are test hooks (include/rxhip.h "Environment")
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
import numpy as np  # noqa: E402

import rxhip  # noqa: E402
from rxhip import workloads  # noqa: E402

mdl = workloads.c1_model()
C = 1024
for T in (10000, 100000):
    y = workloads.generate_batch(mdl, T, C, seed0=1)          # [T][chain][dy]
    rng = np.random.default_rng(0)
    y[rng.random((T, C)) < 0.1] = np.nan
    for mode in ("masked", "per-step (2 regimes)", "masked, one segment"):
        if mode.endswith("one segment"):
            if T > 10000:
                continue
            os.environ["RXHIP_ONE_SEGMENT"] = "1"
        else:
            os.environ.pop("RXHIP_ONE_SEGMENT", None)
        kw = dict(allow_missing=True)
        args = [mdl[k] for k in ("A", "B", "P", "Q", "m0", "V0")]
        if mode.startswith("per-step"):
            args = [np.stack([a, a * (1.0 if k in (0, 1, 4) else 1.5)]) for k, a in enumerate(args)]
            kw = dict(step_model=(np.arange(T) // 50 % 2).astype(np.int32))
            yy = np.nan_to_num(y, nan=0.0)
        else:
            yy = y
        with rxhip.LGSSMEngine(*args, T=T, n_chains=C, **kw) as eng:
            eng.set_data(yy)
            eng.run(1, True)
            t0 = time.perf_counter()
            n = 3
            for _ in range(n):
                eng.run(1, True)
            dt = (time.perf_counter() - t0) / n
            eng.set_profiling(True)
            eng.reset_kernel_times()
            eng.run(1, True)
            kt = {k: round(v["ms_avg"], 3) for k, v in eng.kernel_times().items() if v["launches"]}
            print(f"T={T} {mode}: {dt * 1e3:.2f} ms per sweep, schedule {eng.schedule()}, kernels {kt}", flush=True)
