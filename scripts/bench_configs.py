#!/usr/bin/env python3
"""BASELINE configs 4 and 5 with the same timing contract as bench.py (one JSON line, barrier + synchronize around
exactly K timed steps, max over ranks).  These are parity-test configurations, not the headline line; this script
exists so that their multi-GPU paths can be launched exactly like bench.py:

  python scripts/bench_configs.py --config c4 [--gpus N --steps K --warmup W]      (N > 1 under torch.distributed.run)

  c4: HGF, 512 series per GPU x T = 2000, 10 VMP iterations per observation, GH-31; series shard, the free energy
      (10 values) is all-reduced over RCCL.  step = one filtering pass over all observations.
  c5: univariate GMM, K = 16, 1.25e6 points per GPU, step = ONE VMP iteration: accumulate -> all-reduce of the
      3K+1 statistics (RCCL, in place on the engine's buffer) -> update.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rxinfer.jl_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import rxhip  # noqa: E402
from rxhip import distributed as rd  # noqa: E402
from rxhip import workloads  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=["c4", "c5"], required=True)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--per-gpu", type=int, default=0, help="series (c4) / points (c5) per GPU")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world == 1 and a.gpus > 1:
        sys.exit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        sys.exit("needs an MI355X: no HIP device visible (the product has no CPU path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    def timed(step, steps, finish):
        for _ in range(a.warmup):
            step()
        finish()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        finish()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    if a.config == "c4":
        S, T, iters = a.per_gpu or 512, 2000, 10
        steps = a.steps or 5
        _, _, y = workloads.generate_hgf_batch(T, S, seed=42 + rank)
        eng = rxhip.HGFEngine(T, S, 1.0, 0.0, 0.04, 0.01, device=local_rank)
        eng.set_data(y)
        fe = torch.zeros(iters, dtype=torch.float64, device=device)

        def step():
            eng.run_async(iters, True)
            if dist is not None:
                eng.sync()
                fe.copy_(torch.as_tensor(eng.free_energy(), device=device))
                dist.all_reduce(fe)

        dt = timed(step, steps, eng.sync)
        out = {"metric": "Gauss-Hermite evaluations/sec (HGF filtering, 10 VMP iterations per observation, GH-31)",
               "value": 31 * iters * T * S * world * steps / dt, "unit": "GH-evaluations/s",
               "series_observations_per_s": T * S * world * steps / dt, "ms_per_step": dt / steps * 1e3,
               "config": {"workload": f"HGF {S} series per GPU x T={T} (BASELINE config 4)", "parallelism": f"series sharded over {world} GPU(s), RCCL all-reduce of the free energy"},
               "free_energy_mean_per_series_rank0": (eng.free_energy() / S).tolist()}
        eng.close()
    else:
        K, N = 16, a.per_gpu or 1_250_000
        steps = a.steps or 20
        mus = np.arange(1, K + 1) * 10.0 - 80.0
        rng = np.random.default_rng(12345 + rank)
        y = mus[rng.integers(0, K, size=N)] + rng.standard_normal(N)
        stream = torch.cuda.Stream(device=device)  # engine kernels and the RCCL all-reduce share one stream: no host waits
        eng = rxhip.GMMEngine(N, mus + 1.5, np.full(K, 1e3), np.full(K, 0.01), np.full(K, 0.01), np.ones(K), mus + 1.5,
                              np.full(K, 10.0), np.ones(K), np.ones(K), np.ones(K), device=local_rank, stream=stream.cuda_stream)
        eng.set_data(y)
        shard = rd.DeviceMixtureShard(eng)
        with torch.cuda.stream(stream):
            shard.begin(a.warmup + steps)

        def step():
            with torch.cuda.stream(stream):
                stats = shard.accumulate()
                if dist is not None:
                    dist.all_reduce(stats)
                shard.update(True)

        dt = timed(step, steps, eng.sync)
        fe = eng.free_energy()
        out = {"metric": "VMP iterations/sec (univariate Gaussian mixture K=16)", "value": steps / dt, "unit": "VMP-iterations/s",
               "point_iterations_per_s": N * world * steps / dt, "ms_per_step": dt / steps * 1e3,
               "config": {"workload": f"GMM K=16, {N} points per GPU (BASELINE config 5)", "parallelism": f"points sharded over {world} GPU(s), RCCL all-reduce of the 3K+1 statistics per iteration"},
               "free_energy_last": float(fe[-1]), "free_energy_monotone": bool(np.all(np.diff(fe[a.warmup:]) <= 1e-6 * abs(fe[-1])))}
        eng.close()
    out.update({"n_gpus": world, "steps": steps, "warmup": a.warmup, "higher_is_better": True, "scaling": "weak", "dtype": "f64",
                "data": "synthetic", "vs_baseline": None})
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
