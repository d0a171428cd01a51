#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on BASELINE config 2.

A "step" is one full belief-propagation sweep (forward + backward messages, marginals and the
Bethe free energy) of the d=4 linear Gaussian state-space model over one batch of synthetic
observations: T = 100000 steps × 1024 independent chains PER GPU (weak scaling: chains shard
across ranks with no data-path collective; the only exchange is the RCCL all-reduce of the
scalar free energy).  Inputs are resident in HBM when the timed region starts.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

Prints ONE JSON line (rank 0).  `value` = reference-equivalent message-rule evaluations per
second over all ranks (6 per (chain, time step) per sweep, SURVEY Appendix C — what the
reference counts as after_message_rule_call events).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "rxinfer.jl_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import rxhip  # noqa: E402
from rxhip import workloads  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md:35); 6290 measured copy


def device_observations(mdl, T, C, seed, device):
    """Synthetic y [T][chain][dy] generated on the GPU from the model itself (x0 = 0, as the
    notebook's generate_data), one independent chain per column."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    f = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64, device=device)
    A, B = f(mdl["A"]), f(mdl["B"])
    Lp, Lq = f(np.linalg.cholesky(mdl["P"])), f(np.linalg.cholesky(mdl["Q"]))
    d, dy = A.shape[0], B.shape[0]
    # x_t = A x_{t-1} + w_t  (x_0 = 0)  ==  x_t = Σ_j A^{t-j} w_j : Hillis–Steele doubling over time,
    # 17 passes for T = 1e5 instead of 1e5 tiny launches
    x = torch.randn((T, C, d), generator=g, dtype=torch.float64, device=device) @ Lp.T
    Ak = A.clone()
    s = 1
    while s < T:
        x[s:] = x[s:] + x[:-s] @ Ak.T
        Ak = Ak @ Ak
        s *= 2
    y = x @ B.T + torch.randn((T, C, dy), generator=g, dtype=torch.float64, device=device) @ Lq.T
    del x
    return y


def cpu_baseline(mdl, T, sample_chains, seed):
    """CPU restatement oracle (reference message schedule, fp64, one thread — the reference is
    single-threaded) timed on a bounded sample of the same workload.  Checker/baseline only."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import rxoracle

    rxoracle.build()
    y = workloads.generate_batch(mdl, T, sample_chains, seed0=seed)
    t0 = time.perf_counter()
    *_, cnt = rxoracle.lgssm_bp_batch(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], y,
                                      free_energy=True, nthreads=1)
    dt = time.perf_counter() - t0
    return {"value": cnt.rule_calls / dt, "unit": "rule-calls/s", "cores": 1, "kind": "port",
            "sample": f"{sample_chains} chains x T={T} of the same model, 1 BP sweep with free energy, "
                      f"{dt:.1f} s on 1 of {os.cpu_count()} host cores (CPU restatement of the reference schedule, not RxInfer)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--T", type=int, default=100000)
    ap.add_argument("--chains", type=int, default=1024, help="chains per GPU")
    ap.add_argument("--segments", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-chains", type=int, default=40)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no HIP device visible (the product has no CPU path)")
    torch.cuda.set_device(local_rank)  # before the process group: every collective (and barrier) runs on THIS rank's GPU
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)  # RCCL on ROCm

    mdl = workloads.c1_model()
    T, C = args.T, args.chains
    y = device_observations(mdl, T, C, seed=42 + rank, device=device)
    stream = torch.cuda.Stream(device=device)
    fe_buf = torch.zeros(1, dtype=torch.float64, device=device)
    torch.cuda.synchronize()

    eng = rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=C,
                            segments=args.segments, device=local_rank, stream=stream.cuda_stream)
    eng.set_data_device(y.data_ptr(), y.numel(), keepalive=y)

    def step():
        with torch.cuda.stream(stream):
            eng.run_async(iterations=1, free_energy=True)
            if dist is not None:  # the path's only exchange: global Bethe free energy, 1 double
                eng.copy_free_energy_to_device(fe_buf.data_ptr())
                dist.all_reduce(fe_buf)

    for _ in range(args.warmup):
        step()
    eng.sync()
    torch.cuda.synchronize()
    eng.set_profiling(True)
    eng.reset_kernel_times()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    eng.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    eng.set_profiling(False)

    kt = eng.kernel_times()
    cnt = eng.counters()  # per run_async(1): rule calls of one sweep over this rank's chains
    fe_local = float(eng.free_energy()[0])
    sched = eng.schedule()
    rule_calls_per_step = cnt["rule_calls"] * world
    value = rule_calls_per_step * args.steps / dt
    units = T * C  # (chain, time-step) units per launch on this rank
    d, dy = 4, 4
    ns = d * (d + 1) // 2
    bytes_bwd = 8 * ((d + ns) + (d + d * d))  # read packed forward message, write posterior mean+cov
    bytes_fwd = 8 * (dy + (d + ns))           # read y, write packed forward message
    bytes_sweep = bytes_bwd + bytes_fwd       # = 416 B/U, SURVEY §8(d)
    dom_ms = kt["k_backward"]["ms_avg"]
    achieved = bytes_bwd * units / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("k_backward_hbm_bytes_per_launch")
        except Exception:
            traffic = None
    sweep_ms = dt / args.steps * 1e3
    out = {
        "metric": "node-message-updates/sec (d=4 LGSSM, T=100k, BP sweep with Bethe free energy)",
        "value": value,
        "unit": "rule-calls/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": sweep_ms,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"LGSSM d=4 dy=4 T={T}, {C} independent chains per GPU (BASELINE config 2), "
                               "1 BP sweep + Bethe free energy per step",
                   "chains_per_gpu": C, "T": T, "segments": sched["segments"], "segment_len": sched["segment_len"],
                   "parallelism": f"chains sharded over {world} GPU(s), RCCL all-reduce of the free-energy scalar"},
        "vmp_iters_per_sec": args.steps / dt,
        # `achieved`/`frac` follow the contract: ALGORITHMIC bytes (272 B per (chain, step) for this kernel)
        # ÷ measured kernel time.  The kernel physically moves fewer bytes (`traffic`, PMC) because the covariance
        # half of the forward message is stored once per model; `traffic_achieved` is what the HBM actually sustains.
        "roofline": {"bound": "hbm", "kernel": "k_backward", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_achieved": (traffic / (dom_ms * 1e-3) / 1e9) if (traffic and dom_ms > 0 and (T, C) == (100000, 1024)) else None,
                     "traffic_frac": (traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and dom_ms > 0 and (T, C) == (100000, 1024)) else None,
                     "algorithmic_bytes_per_launch": bytes_bwd * units, "kernel_ms_avg": dom_ms,
                     "sweep_achieved": bytes_sweep * units / (sweep_ms * 1e-3) / 1e9,
                     "sweep_frac": bytes_sweep * units / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        "kernels_ms_avg": {k: round(v["ms_avg"], 4) for k, v in kt.items()},
        "free_energy_rank0": fe_local,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(mdl, T, args.cpu_sample_chains, seed=42)
        out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
