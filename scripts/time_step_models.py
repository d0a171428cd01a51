"""Per-step constants (desc.step_model) at d > 4: the masked MFMA schedule against the sequential one (RXHIP_STEPM_GSEQ=1)."""
import os, sys, time
os.environ["RXHIP_TEST_HOOKS"] = "1"   # the schedule switches below are test hooks (include/rxhip.h "Environment")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
import numpy as np
import rxhip
from rxhip import workloads


def med(eng, n=5):
    eng.run(free_energy=True)
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); eng.run(free_energy=True); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]


for d, C, T, M in ((64, 1, 2000, 4), (64, 256, 200, 4), (8, 1024, 1000, 4), (16, 1, 5000, 8), (32, 1, 1000, 1000), (64, 1, 600, 600)):   # the last two: every step its own model
    ms = [workloads.random_model(d, d, seed=d + 7 * m) for m in range(M)]
    mdl = tuple(np.stack([m[k] for m in ms]) for k in ("A", "B", "P", "Q", "m0", "V0"))
    sm = np.random.default_rng(0).integers(0, M, T).astype(np.int32) if M < T else np.arange(T, dtype=np.int32)
    y = np.random.default_rng(1).standard_normal((T, C, d))
    out = []
    for env in (None, "1"):
        if env:
            os.environ["RXHIP_STEPM_GSEQ"] = env
        else:
            os.environ.pop("RXHIP_STEPM_GSEQ", None)
        t0 = time.perf_counter()
        with rxhip.LGSSMEngine(*mdl, T=T, n_chains=C, step_model=sm) as eng:
            create_ms = (time.perf_counter() - t0) * 1e3
            eng.set_data(y)
            out.append((med(eng, 5 if env is None else 1), eng.free_energy()[-1], create_ms))
    os.environ.pop("RXHIP_STEPM_GSEQ", None)
    print(f"d=dy={d} chains={C} T={T} models={M}: masked MFMA schedule {out[0][0]:.2f} ms | sequential {out[1][0]:.2f} ms = {out[1][0] / out[0][0]:.0f}x | FE {out[0][1]:.6f} / {out[1][1]:.6f} | engine creation {out[0][2]:.1f} ms", flush=True)

# filtering runs (rxhip_run_filter) of a masked engine: masked schedule + km_filter_out against the sequential schedule
for d, C, T in ((64, 1, 2000), (16, 512, 1000)):
    m = workloads.random_model(d, d, seed=d)
    y = workloads.generate_batch(m, T, C, seed0=1)
    y[np.random.default_rng(0).random((T, C)) < 0.1] = np.nan
    out = []
    for env in (None, "1"):
        if env:
            os.environ["RXHIP_FILTER_GSEQ"] = env
        else:
            os.environ.pop("RXHIP_FILTER_GSEQ", None)
        with rxhip.LGSSMEngine(m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"], T=T, n_chains=C, allow_missing=True) as eng:
            eng.set_data(y)
            eng.run_filter(True)
            ts = []
            for _ in range(3 if env is None else 1):
                t0 = time.perf_counter(); eng.run_filter(True); ts.append((time.perf_counter() - t0) * 1e3)
            out.append(sorted(ts)[len(ts) // 2])
    os.environ.pop("RXHIP_FILTER_GSEQ", None)
    print(f"filtering run, d=dy={d} chains={C} T={T}, 10 % missing: masked MFMA schedule {out[0]:.2f} ms | sequential {out[1]:.2f} ms = {out[1] / out[0]:.0f}x", flush=True)

