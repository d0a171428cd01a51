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
ap.add_argument("--filter", action="store_true", help="time the streaming / filtering driver (rxhip_run_filter) instead of the smoother")
a = ap.parse_args()
if a.config == "c4":
    # BASELINE config 4: 4096 independent HGF series, T = 2000, 10 VMP iterations / observation, GH-31
    S, T = (a.chains if a.chains != 1024 else 4096), (a.T if a.T != 100000 else 2000)
    from rxhip.workloads import generate_hgf_batch
    _, _, y = generate_hgf_batch(T, S, 42)
    eng = rxhip.HGFEngine(T, S, 1.0, 0.0, 0.04, 0.01)
    eng.set_data(y)
    eng.run(10, True)
    eng.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        eng.run(10, True)
    dt = (time.perf_counter() - t0) / a.steps
    fe = eng.free_energy()
    print({"config": "c4", "series": S, "T": T, "ms_per_run": dt * 1e3, "kernels": {k: round(v["ms_avg"], 4) for k, v in eng.kernel_times().items() if v["launches"]},
           "gh_evaluations_per_s": 31 * 10 * T * S / dt, "series_observations_per_s": T * S / dt, "rule_calls_per_s": eng.counters()["rule_calls"] / dt,
           "fe_mean_per_series": (fe / S).tolist()})
    eng.close()
    sys.exit(0)
if a.config == "c5":
    # BASELINE config 5: univariate GMM, K = 16, N = 1e7, 20 VMP iterations, component means k·10 − 80
    K, N = 16, (a.T if a.T != 100000 else 10_000_000)
    mus = np.arange(1, K + 1) * 10.0 - 80.0
    rng = np.random.default_rng(12345)
    z = rng.integers(0, K, size=N)
    y = mus[z] + rng.standard_normal(N)
    for mat in (False, True):
        eng = rxhip.GMMEngine(N, mus + 1.5, np.full(K, 1e3), np.full(K, 0.01), np.full(K, 0.01), np.ones(K), mus + 1.5,
                              np.full(K, 10.0), np.ones(K), np.ones(K), np.ones(K), materialize_responsibilities=mat)
        eng.set_data(y)
        eng.run(2, True)
        eng.set_profiling(True)
        t0 = time.perf_counter()
        eng.run(20, True)
        dt = time.perf_counter() - t0
        fe = eng.free_energy()
        kt = {k: round(v["ms_avg"], 4) for k, v in eng.kernel_times().items() if v["launches"]}
        print({"config": "c5", "materialize_last_q_z": mat, "ms_per_iteration": dt / 20 * 1e3, "kernels": kt,
               "point_iterations_per_s": N * 20 / dt, "GBps_read_y": N * 8 * 20 / dt / 1e9, "fe_last": fe[-1],
               "fe_monotone": bool(np.all(np.diff(fe) <= 1e-6 * abs(fe[-1]))), "means": np.round(eng.history()[-1, 0], 3).tolist()})
        eng.close()
    sys.exit(0)
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
if a.filter:
    for _ in range(a.warmup):
        eng.run_filter(True)
    eng.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        eng.run_filter_async(True)
    eng.sync()
else:
    eng.run(a.warmup, True)
    eng.set_profiling(True)
    t0 = time.perf_counter()
    eng.run(a.steps, True)
dt = time.perf_counter() - t0
print({"mode": "filter" if a.filter else "smooth", "ms_per_step": dt / a.steps * 1e3, "kernels": {k: round(v["ms_avg"], 4) for k, v in eng.kernel_times().items()},
       "schedule": eng.schedule()})
eng.close()
