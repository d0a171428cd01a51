#!/usr/bin/env python3
"""The reference's own benchmark (benchmarks notebook, cells 12 and 24): d = 2 LGSSM, one chain, smoothing and
filtering for T in 50 … 50 000 — here through the host mirror `rxhip.infer(...)`, END TO END per call: engine
construction (model tables), host → device copy of the observations, the sweep, device → host copy of the posteriors.
Prints the minimum over repetitions (the notebook reports BenchmarkTools' minimum) next to the published numbers
(Apple M4 Max, Julia 1.12) — different hardware, quoted only as the true-reference anchor."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rxinfer.jl_amd"))
import numpy as np  # noqa: E402

import rxhip  # noqa: E402
from rxhip import workloads  # noqa: E402

PUBLISHED_MS = {50: (3.362, 0.609625), 100: (6.664, 1.136), 500: (35.832, 5.275), 1000: (77.231, 10.464), 2000: (162.966, 20.855),
                5000: (439.728, 52.631), 10000: (901.667, 102.904), 25000: (2493.0, 267.599), 50000: (8326.0, 535.792)}
mdl = workloads.notebook_model()
spec = rxhip.linear_gaussian_ssm(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"])
spec_f = rxhip.linear_gaussian_ssm(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], prior_through_transition=True)
rxhip.infer(model=spec, data={"y": workloads.generate_chain(mdl, 10, 0)[1]})  # context creation is not part of a call
print("| T | smoothing, this engine (ms) | published RxInfer (ms) | filtering, this engine (ms) | published RxInfer (ms) |")
print("|---|---|---|---|---|")
for T, (ps, pf) in PUBLISHED_MS.items():
    _, y = workloads.generate_chain(mdl, T, 42)
    best_s = best_f = 1e9
    for _ in range(7):
        t0 = time.perf_counter()
        r = rxhip.infer(model=spec, data={"y": y}, options={"limit_stack_depth": 500})
        best_s = min(best_s, time.perf_counter() - t0)
        t0 = time.perf_counter()
        r = rxhip.infer(model=spec_f, data={"y": y}, autoupdates=True, keephistory=T)
        best_f = min(best_f, time.perf_counter() - t0)
    print(f"| {T} | {best_s * 1e3:.3f} | {ps:.3f} | {best_f * 1e3:.3f} | {pf:.3f} |")
