#!/usr/bin/env python3
"""Torch-free driver for rocprofv3: the same hot path as bench.py (C2: d=4, T=100000, 1024 chains,
one BP sweep + free energy per step) through the C ABI only, so that the trace holds nothing but
the engine's kernels.  Observations are i.i.d. normal draws (timing does not depend on values)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rxinfer.jl_amd"))
import numpy as np  # noqa: E402

import rxhip  # noqa: E402
from rxhip import workloads  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--T", type=int, default=100000)
ap.add_argument("--chains", type=int, default=1024)
ap.add_argument("--segments", type=int, default=0)
ap.add_argument("--config", default="c2")
a = ap.parse_args()
if a.config == "c3":
    mdl = workloads.c3_model()
    if a.T == 100000:
        a.T, a.chains = 10000, 1
else:
    mdl = workloads.c1_model()
rng = np.random.default_rng(0)
y = rng.standard_normal((a.T, a.chains, mdl["B"].shape[0])) * 3.0
eng = rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=a.T, n_chains=a.chains,
                        segments=a.segments)
eng.set_data(y)
eng.run(a.warmup, True)
eng.set_profiling(True)
t0 = time.perf_counter()
eng.run(a.steps, True)
dt = time.perf_counter() - t0
print({"ms_per_step": dt / a.steps * 1e3, "kernels": {k: round(v["ms_avg"], 4) for k, v in eng.kernel_times().items()},
       "schedule": eng.schedule()})
import os
if os.environ.get("RXHIP_ABLATE"): print("fe(debug)", eng.free_energy()[-1], "segments", eng.schedule())
eng.close()
