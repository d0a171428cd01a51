"""Engine creation for BASELINE config 3 (d = dy = 64, T = 10⁴, one chain): a model no engine of the process has seen (tables built
on the device: csrc/dense_tab_kernels.hpp), then the same model again (tables from the cache).  RXHIP_TRACE=1 prints the stages."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
import numpy as np, rxhip
from rxhip import workloads
mdl = workloads.c3_model()
y = workloads.generate_batch(mdl, 10000, 1, seed0=1)
for rep in range(6):
    m = dict(mdl)
    if rep < 4:
        m["P"] = mdl["P"] * (1.0 + 0.01 * rep)   # never seen before
    t = time.time()
    eng = rxhip.LGSSMEngine(m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"], T=10000, n_chains=1, segments=int(os.environ.get("C3_SEGMENTS", "0")))
    t1 = time.time(); eng.set_data(y); t2 = time.time(); eng.run(1, True); fe = eng.free_energy(); t3 = time.time()
    eng.close()
    print(f"{'new model' if rep < 4 else 'seen model'}: create {1e3*(t1-t):.2f} ms  set_data {1e3*(t2-t1):.2f}  first sweep + sync {1e3*(t3-t2):.2f}  total {1e3*(t3-t):.2f}", flush=True)
