"""d = dy = 64, T = 2000, one chain, 10 % of the observations missing / four per-step models: sweep time and the engine's kernel slots."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, rxhip
from rxhip import workloads
import bench
mdl = workloads.c3_model()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
y = workloads.generate_batch(mdl, T, 1, seed0=6400)
ym = y.copy()
ym[np.random.default_rng(0).random((T, 1)) < 0.1] = np.nan
for name, yy, kw in (("fully observed", y, {}), ("10 % missing", ym, dict(allow_missing=True))):
    with rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=1, **kw) as eng:
        eng.set_data(yy)
        eng.run(1, True)
        ms, kt = bench.timed_sweeps(eng, 10, 2)
        print(name, "ms", round(ms, 4), kt, eng.schedule(), flush=True)
