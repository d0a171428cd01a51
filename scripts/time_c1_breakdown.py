"""BASELINE config 1 (d = 4, T = 1000, one chain), `infer(...)` end to end, and where the time goes: Python mirror, engine construction, the
one-round-trip call (rxhip_lgssm_infer: H2D + sweep + free energy + D2H + one synchronisation), destruction — with the sweep in one launch
(k_small_sweep) and as five (RXHIP_SMALL_SWEEP=0)."""
import os, sys, time
os.environ["RXHIP_TEST_HOOKS"] = "1"   # the schedule switches below are test hooks (include/rxhip.h "Environment")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
import numpy as np, rxhip
from rxhip import workloads
mdl = workloads.c1_model()
_, y = workloads.generate_chain(mdl, 1000, 42)
spec = rxhip.linear_gaussian_ssm(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"])
args = (mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"])
def best(f, n=200):
    b = 1e9
    for _ in range(n):
        t0 = time.perf_counter(); f(); b = min(b, time.perf_counter() - t0)
    return b * 1e3
for mode in ("1", "0"):
    os.environ["RXHIP_SMALL_SWEEP"] = mode
    rxhip.infer(model=spec, data={"y": y}, free_energy=True)
    total = best(lambda: rxhip.infer(model=spec, data={"y": y}, free_energy=True))
    create = best(lambda: rxhip.LGSSMEngine(*args, T=1000, n_chains=1).close())
    eng = rxhip.LGSSMEngine(*args, T=1000, n_chains=1)
    yb = np.ascontiguousarray(y[:, None, :])
    call = best(lambda: eng.infer(yb, iterations=1, free_energy=True))
    eng.set_data(yb)
    def sweep():
        eng.run_async(1, True); eng.sync()
    sw = best(sweep)
    eng.close()
    print(f"RXHIP_SMALL_SWEEP={mode}: infer(...) {total:.4f} ms | create + destroy {create:.4f} | rxhip_lgssm_infer {call:.4f} | sweep + sync alone {sw:.4f}", flush=True)
