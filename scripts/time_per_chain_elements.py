"""Per-chain models at the C2 size (d = 4, T = 10^5, 1024 chains, every chain its own model block): sweep and kernel times, with the frozen tail of
k_seg_elements (default), without it (RXHIP_ELEM_FULL=1) and with looser fixed-point tests (2: 1e-13, 3: 1e-11 — experiments only)."""
import os, sys, time
os.environ["RXHIP_TEST_HOOKS"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
import numpy as np, torch, rxhip
from rxhip import workloads
mdl = workloads.c1_model()
T, C = 100000, 1024
y = torch.from_numpy(workloads.generate_batch(mdl, T, C, seed0=42, threads=16)).cuda()
rep = lambda a: np.broadcast_to(a, (C,) + a.shape).copy()
for hook in (None, "1"):
    if hook: os.environ["RXHIP_ELEM_FULL"] = hook
    else: os.environ.pop("RXHIP_ELEM_FULL", None)
    eng = rxhip.LGSSMEngine(rep(mdl["A"]), rep(mdl["B"]), rep(mdl["P"]), rep(mdl["Q"]), rep(mdl["m0"]), rep(mdl["V0"]), T=T, n_chains=C, chain_model=np.arange(C, dtype=np.int32))
    eng.set_data_device(y.data_ptr(), y.numel(), keepalive=y)
    for _ in range(2): eng.run_async(1, True)
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(4): eng.run_async(1, True)
    eng.sync()
    ms = (time.perf_counter() - t0) / 4 * 1e3
    eng.set_profiling(True); eng.reset_kernel_times(); eng.run(2, True); eng.sync()
    kt = {k: round(v["ms_avg"], 3) for k, v in eng.kernel_times().items() if v["launches"]}
    print(f"RXHIP_ELEM_FULL={hook}: {ms:.3f} ms {kt} {eng.schedule()} fe {eng.free_energy()[-1]:.6f}", flush=True)
    eng.close()
