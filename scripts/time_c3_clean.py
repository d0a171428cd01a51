"""BASELINE config 3 (d = dy = 64, T = 10⁴, one chain): the sweep timed WITHOUT per-kernel events (they cost ≈ 6 – 10 µs of queue gap each,
five per sweep), then the kernel breakdown from an instrumented pass.  RXHIP_LIB selects a library variant (scripts/ab_variants.sh)."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
import numpy as np, rxhip
MOCK = any(x in os.environ.get('RXHIP_LIB', '') for x in ('mock', 'nodiag', 'notrail', 'noinv'))   # timing experiments whose arithmetic is deliberately wrong
def sync(e):
    try: e.sync()
    except rxhip.RxHipError:
        if not MOCK: raise
from rxhip import workloads
mdl = workloads.c3_model()
T = 10000
y = workloads.generate_batch(mdl, T, 1, seed0=6400)
eng = rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=1, segments=int(os.environ.get("C3_SEGMENTS", "0")))
eng.set_data(y)
for _ in range(3): eng.run_async(1, True)
sync(eng)
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(30): eng.run_async(1, True)
    sync(eng)
    best = min(best, (time.perf_counter() - t0) / 30 * 1e3)
eng.set_profiling(True); eng.reset_kernel_times()
for _ in range(20): eng.run_async(1, True)
sync(eng)
kt = {k: round(v["ms_avg"], 4) for k, v in eng.kernel_times().items() if v["launches"]}
fe = [float('nan')] if MOCK else eng.free_energy()
print(json.dumps({"segments": eng.schedule() if hasattr(eng, "schedule") else None, "lib": os.environ.get("RXHIP_LIB", "default"), "ms_per_step_clean": round(best, 4), "kernels_ms_avg": kt, "fe": float(fe[-1])}))
