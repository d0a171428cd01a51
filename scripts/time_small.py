"""Where the end-to-end time of a small infer() goes (d = 2, one chain): create / set_data / run+sync / read-back."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
import numpy as np
import rxhip
from rxhip import workloads

for T in (50, 1000, 10000):
    m = workloads.random_model(2, 2, seed=3)
    y = workloads.generate_batch(m, T, 1, seed0=1)
    best = None
    for rep in range(30):
        t0 = time.perf_counter()
        eng = rxhip.LGSSMEngine(m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"], T=T, n_chains=1)
        t1 = time.perf_counter(); eng.set_data(y)
        t2 = time.perf_counter(); eng.run(1, True)
        t3 = time.perf_counter(); fe = eng.free_energy(); mm, VV = eng.marginals()
        t4 = time.perf_counter(); eng.close()
        t5 = time.perf_counter()
        cur = (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0)
        if best is None or cur[-1] < best[-1]: best = cur
    print(f"T={T}: create {best[0]*1e6:.0f} us, set_data {best[1]*1e6:.0f}, run+sync {best[2]*1e6:.0f}, read-back {best[3]*1e6:.0f}, close {best[4]*1e6:.0f}, total {best[5]*1e6:.0f}; schedule {eng.schedule() if False else ''}")
