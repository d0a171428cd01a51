import sys, os, time
sys.path.insert(0, os.path.join(os.getcwd(), "rxinfer.jl_amd"))
import numpy as np, rxhip
from rxhip import workloads
mdl = workloads.c3_model()
y = workloads.generate_batch(mdl, 10000, 1, seed0=1)
for rep in range(3):
    t = time.time()
    eng = rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=10000, n_chains=1)
    t1 = time.time(); eng.set_data(y); t2 = time.time(); eng.run(1, True); fe = eng.free_energy(); t3 = time.time()
    m, V = eng.marginals(); t4 = time.time()
    eng.close()
    print(f"create {1e3*(t1-t):.1f} ms  set_data {1e3*(t2-t1):.1f}  run+sync {1e3*(t3-t2):.1f}  marginals D2H {1e3*(t4-t3):.1f}")
