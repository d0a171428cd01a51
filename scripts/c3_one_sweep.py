"""One sweep of BASELINE config 3 (debug builds that print from the kernel: RXHIP_LIB=…)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
import rxhip
from rxhip import workloads
mdl = workloads.c3_model()
T = 10000
y = workloads.generate_batch(mdl, T, 1, seed0=6400)
eng = rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=1, segments=int(os.environ.get("C3_SEGMENTS", "0")))
eng.set_data(y)
eng.run(1, True)
print("fe", eng.free_energy()[-1])
