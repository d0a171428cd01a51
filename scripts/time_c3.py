"""C3 (d = dy = 64, T = 10^4, one chain) sweep time with the kernel split, for A/B runs of library variants (RXHIP_LIB).  Optional argv: T."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("rxinfer.jl_amd", "oracle", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import rxhip
from rxhip import workloads
import bench
T = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
SEG = int(sys.argv[2]) if len(sys.argv) > 2 else 0
mdl = workloads.c3_model()
y = workloads.generate_batch(mdl, T, 1, seed0=6400)
import time
rxhip.lib().rxhip_device_count()
t0 = time.perf_counter()
with rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=1, device=0, segments=SEG) as eng:
    eng.set_data(y)
    eng.run(1, True)
    print('create + set_data + first run ms', round((time.perf_counter() - t0) * 1e3, 1))
    ms, kt = bench.timed_sweeps(eng, 20, 3)
    fe = eng.free_energy_per_chain()[0]
    print("lib", os.environ.get("RXHIP_LIB", "default"), "T", T, "ms", round(ms, 4), {k: round(v, 4) for k, v in kt.items()}, "fe", repr(float(fe)), "sched", eng.schedule())
