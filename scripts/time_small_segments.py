"""One chain, short series: sweep + sync over the number of segments (one launch: k_small_sweep where chains·S ≤ 256, else five)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
import numpy as np, rxhip
from rxhip import workloads
def best(f, n=100):
    b = 1e9
    for _ in range(n):
        t0 = time.perf_counter(); f(); b = min(b, time.perf_counter() - t0)
    return b * 1e3
for name, mdl in (("d=4", workloads.c1_model()), ("d=2", workloads.notebook_model())):
    args = (mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"])
    for T in (100, 1000, 5000):
        _, y = workloads.generate_chain(mdl, T, 42)
        yb = np.ascontiguousarray(y[:, None, :])
        row = []
        for seg in (0, 16, 32, 57, 64, 96, 128, 192, 256, 400):
            if seg > T - 1: continue
            eng = rxhip.LGSSMEngine(*args, T=T, n_chains=1, segments=seg)
            eng.set_data(yb)
            def sweep():
                eng.run_async(1, True); eng.sync()
            sweep()
            row.append(f"{seg}:{eng.schedule()['segments']}x{eng.schedule()['segment_len']}={best(sweep):.4f}")
            eng.close()
        print(name, "T", T, " ".join(row), flush=True)
