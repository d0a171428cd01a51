"""Sweep time of the MFMA path for mid-size state dimensions with many chains (not a BASELINE config): effect of the
number of time segments (workgroups per CU)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
import numpy as np
import rxhip
from rxhip import workloads

cases = [(8, 4, 1024, 1000, (0, 1, 2, 3, 6, 12)), (16, 16, 512, 1000, (0, 1, 3, 6, 12, 24)), (32, 32, 128, 1000, (0, 2, 4, 8, 16)),
         (48, 48, 64, 1000, (0, 4, 8)), (64, 64, 64, 1000, (0, 4, 8))]
if len(sys.argv) > 1:
    cases = [c[:4] + ((0,),) for c in cases]
for d, dy, C, T, segs in cases:
    m = workloads.random_model(d, dy, seed=d)
    y = workloads.generate_batch(m, T, min(C, 8), seed0=1)
    y = np.tile(y, (1, C // min(C, 8), 1))
    for sg in segs:
        with rxhip.LGSSMEngine(m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"], T=T, n_chains=C, segments=sg) as eng:
            eng.set_data(y)
            eng.run(1, True); eng.free_energy()
            t = time.time()
            for _ in range(3):
                eng.run(1, True)
            eng.free_energy()
            dt = (time.time() - t) / 3
            flops = 18.0 * d ** 3 * T * C
            print(f"d={d} dy={dy} chains={C} T={T} segments={sg}: {dt*1e3:.2f} ms/sweep, {T*C/dt:.3e} steps/s, {flops/dt/1e12:.2f} TF/s (18 d^3 per step), schedule {eng.schedule()}")
