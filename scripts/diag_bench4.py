import sys, os, time
sys.path.insert(0, os.path.join(os.getcwd(), "rxinfer.jl_amd"))
import numpy as np, rxhip
from rxhip import workloads
mdl = workloads.c3_model()
y = workloads.generate_batch(mdl, 10000, 1, seed0=6400)
for rep in range(5):
    t0 = time.perf_counter()
    eng = rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"] * (1 + 0.1 * (rep % 2)), mdl["Q"], mdl["m0"], mdl["V0"], T=10000, n_chains=1)
    eng.set_data(y); eng.run(1, True); eng.sync()
    t1 = time.perf_counter()
    eng.set_profiling(True)
    ts = []
    for i in range(40):
        a = time.perf_counter(); eng.run_async(1, True); eng.sync(); ts.append(round(1e3 * (time.perf_counter() - a), 2))
    filt = []
    for i in range(12):
        a = time.perf_counter(); eng.run_filter_async(True); eng.sync(); filt.append(round(1e3 * (time.perf_counter() - a), 2))
    print(rep, "create+first %.1f ms" % (1e3 * (t1 - t0)), "smooth sweeps:", [t for t in ts if t > 1.5] or "all < 1.5", "max idx", int(np.argmax(ts)), "| filter:", [t for t in filt if t > 2.0] or "all < 2", flush=True)
    tc = time.perf_counter(); eng.close(); print("   close %.2f ms" % (1e3 * (time.perf_counter() - tc)), flush=True)
