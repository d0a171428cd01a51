"""d = dy = 64, T = 2000, one chain, 10 % missing: the sweep by number of time segments of the masked schedule."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, rxhip
from rxhip import workloads
import bench
mdl = workloads.c3_model()
T = 2000
y = workloads.generate_batch(mdl, T, 1, seed0=6400)
ym = y.copy()
ym[np.random.default_rng(0).random((T, 1)) < 0.1] = np.nan
for seg in (0, 125, 200, 250, 334, 400, 500, 667, 1000):
    try:
        with rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=1, allow_missing=True, segments=seg) as eng:
            eng.set_data(ym)
            eng.run(1, True)
            ms, kt = bench.timed_sweeps(eng, 10, 2)
            print(f"segments={seg:4d} -> {eng.schedule()}  {ms:.4f} ms  fe {eng.free_energy()[-1]:.9f}", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"segments={seg}: {e!r}", flush=True)
