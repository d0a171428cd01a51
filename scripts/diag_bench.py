import sys, os, time, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench, rxhip
from rxhip import workloads
def c3():
    r = bench.extra_c3(0)
    return {k: r[k] for k in ("ms_per_step", "kernels_ms_avg", "create_set_data_first_run_ms", "create_stages_ms")}
print("1 alone:", c3(), flush=True)
mdl = workloads.c1_model()
y_host = workloads.generate_batch(mdl, 20000, 256, seed0=42)
print("2 after generate_batch:", c3()["ms_per_step"], flush=True)
base, _ = bench.cpu_baseline(mdl, y_host, 8)
print("3 after cpu_baseline (OpenMP all cores):", c3()["ms_per_step"], flush=True)
y = torch.from_numpy(y_host).cuda()
print("4 after torch H2D:", c3()["ms_per_step"], flush=True)
r = bench.extra_missing(mdl, 20000, 256, y, 0, y_host)
print("5 after extra_missing:", c3()["ms_per_step"], r["ms_per_step"], flush=True)
