"""The headline sweep (C2: d = 4, T = 10^5, 1024 chains) is bimodal per PROCESS (k_backward 3.66 or 3.92 ms).  Inside one process: does a
different HIP stream (a fresh one, a high-priority one) or a second allocation change the mode?"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
import numpy as np, torch, rxhip
from rxhip import workloads
mdl = workloads.c1_model()
T, C = 100000, 1024
y = torch.from_numpy(workloads.generate_batch(mdl, T, C, seed0=42, threads=16)).cuda()
def sweep_ms(eng, n=8):
    for _ in range(3): eng.run_async(1, True)
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(n): eng.run_async(1, True)
    eng.sync()
    return (time.perf_counter() - t0) / n * 1e3
out = []
keep = []
labels = ("default stream of the pool", "fresh torch stream", "high-priority torch stream", "default again, after a 3 GB dummy allocation")
if os.environ.get("MODES_PLAIN"):
    labels = ("default stream of the pool", "default 2", "default 3", "default 4", "default 5")
for label in labels:
    stream = None
    if label.startswith("fresh"):
        stream = torch.cuda.Stream(); keep.append(stream)
    elif label.startswith("high"):
        stream = torch.cuda.Stream(priority=-1); keep.append(stream)
    elif label.startswith("default again"):
        keep.append(torch.empty(3 * 2**30, dtype=torch.uint8, device="cuda"))
    eng = rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=C, stream=(stream.cuda_stream if stream else None))
    eng.set_data_device(y.data_ptr(), y.numel(), keepalive=y)
    ms = sweep_ms(eng)
    eng.close()
    out.append(f"{label}: {ms:.3f}")
print(" | ".join(out), flush=True)
