#!/usr/bin/env python3
"""What the detour over the node-array executor costs a model beyond the conditioning envelope of the information-form chain engines: `rxhip.infer` (smoothing + free
energy, one call: graph construction on the host, executor compile, sweep, read-back) on vague-prior models, next to the chain engine on the same shapes inside the
envelope.  Run on an MI355X: python scripts/time_envelope_detour.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rxinfer.jl_amd"))
import rxhip  # noqa: E402
from rxhip import workloads  # noqa: E402

for d, dy, T, C in ((16, 8, 200, 1), (32, 16, 200, 1), (32, 16, 1000, 1), (64, 32, 200, 1), (64, 32, 100, 8)):
    for v0, label in ((1.0, "inside"), (1e5, "beyond")):
        mdl = workloads.random_model(d, dy, seed=d)
        mdl["V0"] = v0 * np.eye(d)
        if v0 > 1.0:
            mdl["Q"] = 1e-2 * mdl["Q"]
        y = np.transpose(workloads.generate_batch(mdl, T, C, seed0=1, threads=1), (1, 0, 2))
        spec = rxhip.linear_gaussian_ssm(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"])
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            res = rxhip.infer(model=spec, data={"y": y if C > 1 else y[0]}, free_energy=True)
            ts.append(time.perf_counter() - t0)
        print(f"d={d:2d} dy={dy:2d} T={T:4d} chains={C}: {label} the envelope: infer() {1e3 * min(ts):8.2f} ms (first call {1e3 * ts[0]:8.2f})", flush=True)
