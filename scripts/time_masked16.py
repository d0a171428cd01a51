"""Masked batch at d = 16 (dy = 8, 512 chains, T = 1000, 10 % missing): sweep time and kernel breakdown over the number of segments."""
import os, sys, time
os.environ["RXHIP_TEST_HOOKS"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
import numpy as np, rxhip
from rxhip import workloads
d, dy, C, T = 16, 8, 512, 1000
m = workloads.random_model(d, dy, seed=d)
y = np.tile(workloads.generate_batch(m, T, 8, seed0=1), (1, C // 8, 1))
ym = y.copy(); ym[np.random.default_rng(0).random((T, C)) < 0.1] = np.nan
args = (m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"])
for seg in tuple(int(s) for s in os.environ.get("M16_SEGMENTS", "0,1,2,4,8").split(",")):
    with rxhip.LGSSMEngine(*args, T=T, n_chains=C, allow_missing=True, segments=seg) as eng:
        eng.set_data(ym)
        for _ in range(3): eng.run_async(1, True)
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(5): eng.run_async(1, True)
        eng.sync()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        eng.set_profiling(True); eng.reset_kernel_times(); eng.run(2, True); eng.sync()
        kt = {k: round(v["ms_avg"], 3) for k, v in eng.kernel_times().items() if v["launches"]}
        print(f"segments={seg}: {eng.schedule()} {ms:.3f} ms {kt} fe {eng.free_energy()[-1]:.4f}", flush=True)
