"""The composed graph of round 4 (unknown observation-noise precision, csrc/noise_kernels.hpp): d = dy = 4, 1024 chains × T = 10⁴, 10 VMP iterations —
the driver the round-4 profile pass runs under rocprofv3."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
import numpy as np, rxhip
from rxhip import workloads
mdl = workloads.c1_model()
T, C, iters = 10000, 1024, 10
y = workloads.generate_batch(mdl, T, C, seed0=4242, threads=16)
with rxhip.LGSSMNoiseEngine(mdl["A"], mdl["B"], mdl["P"], mdl["m0"], mdl["V0"], T, 5.0, np.eye(4), n_chains=C) as eng:
    eng.set_data(y)
    eng.run(iters, True)
    t0 = time.perf_counter()
    eng.run(iters, True)
    print({"workload": f"noise VMP d=4 chains={C} T={T}", "ms_per_iteration": (time.perf_counter() - t0) / iters * 1e3, "fe_first_last": [float(eng.free_energy()[0]), float(eng.free_energy()[-1])]})
