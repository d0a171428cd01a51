"""Does the C3 line of bench.py depend on what the process did before?  extra_c3 in a fresh process; with torch's context alive; after the other extra lines
that precede it in bench.py (per-chain models: 23 GB engines created and destroyed; C1 with its host threads; the masked C2 batch)."""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rxinfer.jl_amd"))
import bench  # noqa: E402
from rxhip import workloads  # noqa: E402


def c3(tag):
    r = bench.extra_c3(0, parity=False)
    print(tag, round(r["ms_per_step"], 4), r["kernels_ms_avg"], flush=True)


c3("fresh process                 ")
import torch  # noqa: E402
mdl = workloads.c1_model()
T, C = 100000, 1024
y_host = workloads.generate_batch(mdl, T, C, seed0=42, threads=32)
y = torch.from_numpy(y_host).to("cuda:0")
torch.cuda.synchronize()
c3("torch context + 3.3 GB tensor ")
r = bench.extra_per_chain_models(mdl, T, C, y, 0, None)
print("per_chain_models", round(r["ms_per_step"], 3), flush=True)
c3("after per_chain_models        ")
r = bench.extra_c1(0, False)
c3("after c1                      ")
r = bench.extra_missing(mdl, T, C, y, 0, None)
print("c2_missing", round(r["ms_per_step"], 3), flush=True)
c3("after c2_missing              ")
