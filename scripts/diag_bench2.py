import sys, os, time
sys.path.insert(0, os.path.join(os.getcwd(), "rxinfer.jl_amd"))
import numpy as np, rxhip
from rxhip import workloads
mdl = workloads.c3_model()
y = workloads.generate_batch(mdl, 10000, 1, seed0=6400)
def sweeps(tag):
    eng = rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=10000, n_chains=1)
    eng.set_data(y); eng.run(1, True)
    ts = []
    for i in range(5):
        t0 = time.perf_counter(); eng.run_async(1, True); t1 = time.perf_counter(); eng.sync(); t2 = time.perf_counter()
        ts.append((round(1e3*(t1-t0),3), round(1e3*(t2-t1),3)))
    t0 = time.perf_counter()
    for i in range(20): eng.run_async(1, True)
    t1 = time.perf_counter(); eng.sync(); t2 = time.perf_counter()
    print(tag, "single (launch, sync) ms:", ts, "| 20 back-to-back: launch %.2f ms, then sync %.2f ms" % (1e3*(t1-t0), 1e3*(t2-t1)), "stream", hex(eng.stream() or 0), flush=True)
    eng.close()
sweeps("fresh")
sweeps("again")
big = workloads.generate_batch(workloads.c1_model(), 20000, 256, seed0=42)
sweeps("after generate_batch")
sweeps("again")
z = np.random.default_rng(0).standard_normal((4000, 256, 64))
sweeps("after big numpy alloc")
del big, z
sweeps("after free")
