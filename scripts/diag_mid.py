import sys, os, time
sys.path.insert(0, os.path.join(os.getcwd(), "rxinfer.jl_amd"))
import numpy as np, rxhip
from rxhip import workloads
d, dy, C, T = 64, 64, 64, 1000
m = workloads.random_model(d, dy, seed=d)
y = workloads.generate_batch(m, T, 8, seed0=1)
y = np.tile(y, (1, C // 8, 1))
for mode in ("device", "host"):
    if mode == "host": os.environ["RXHIP_HOST_TABLES"] = "1"
    rxhip.lib().rxhip_release_cached_memory()
    eng = rxhip.LGSSMEngine(m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"], T=T, n_chains=C)
    eng.set_data(y)
    ts = []
    for i in range(8):
        t0 = time.perf_counter(); eng.run_async(1, True); t1 = time.perf_counter(); eng.sync(); t2 = time.perf_counter()
        ts.append((round(1e3*(t1-t0),3), round(1e3*(t2-t1),3)))
    print(mode, "launch/sync ms per sweep:", ts, eng.schedule(), flush=True)
    eng.close()
