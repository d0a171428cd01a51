"""bench.py's own main() at world size 2 over `gloo`, with a stub engine in place of the HIP engine (no GPU here).

What is under test is the host logic of the bench line that a 1-GPU box never reaches: which chains a rank draws (chain c of the job from
default_rng(42 + c)), weak / strong sharding of --chains, the free-energy exchange (all-gather + sum in rank order), the max-over-ranks
timing and the whole-job `value`.  The stub engine computes nothing: its "free energy" is the sum of the observations it was handed, so
the global value pins both the shard every rank generated and the exchange."""
import ctypes
import json
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class StubEngine:
    """The subset of rxhip.LGSSMEngine that bench.main() drives.  free energy := Σ y of the shard (a checksum of the data it was given)."""

    def __init__(self, A, B, P, Q, m0, V0, T, n_chains, segments=0, device=0, stream=None):
        self.T, self.C, self.fe, self.runs = int(T), int(n_chains), 0.0, 0

    def close(self):
        pass

    def model_tables_ms(self):
        return 0.0

    def set_data_device(self, ptr, n, keepalive=None):
        assert n == self.T * self.C * 4
        self._y = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_double)), shape=(n,))

    def run_async(self, iterations=1, free_energy=True):
        self.fe = float(np.sum(self._y))
        self.runs += 1

    def copy_free_energy_to_device(self, ptr):
        ctypes.cast(ptr, ctypes.POINTER(ctypes.c_double))[0] = self.fe

    def sync(self):
        pass

    def set_profiling(self, on):
        pass

    def reset_kernel_times(self):
        pass

    def kernel_times(self):
        return {"k_backward": {"ms_avg": 1.0, "launches": self.runs}}

    def counters(self):
        return {"rule_calls": (6 * self.T - 3) * self.C}

    def free_energy(self):
        return np.array([self.fe])

    def schedule(self):
        return {"segments": 1, "segment_len": self.T}

    def create_stages(self):
        return {}


class StubHgf:
    """free_energy() = a checksum of the series this rank generated, one value per VMP iteration"""

    def __init__(self, T, S, kappa, omega, zv, yv, device=0):
        self.T, self.S = T, S

    def set_data(self, y):
        self.sum = float(np.sum(y))

    def run_async(self, iters, fe):
        self.iters = iters

    def sync(self):
        pass

    def free_energy(self):
        return np.arange(1, self.iters + 1) * self.sum

    def close(self):
        pass


class StubGmm:
    def __init__(self, N, *a, device=0, stream=None):
        self.N, self.fe = N, []

    def set_data(self, y):
        self.sum = float(np.sum(y))

    def sync(self):
        pass

    def free_energy(self):
        return np.array(self.fe)

    def close(self):
        pass


class StubShard:
    """accumulate() -> this rank's statistics (here: its data checksum); update() books the all-reduced value as the free energy"""

    def __init__(self, eng):
        self.e = eng

    def begin(self, n):
        import torch
        self.stats = torch.zeros(3, dtype=torch.float64)

    def accumulate(self):
        self.stats[:] = self.e.sum
        return self.stats

    def update(self, fe):
        self.e.fe.append(-float(self.stats[0]) / (1 + len(self.e.fe)))   # decreasing in magnitude -> "monotone" has something to look at


def _worker_extras(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "rxinfer.jl_amd")):
        sys.path.insert(0, p)
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    import contextlib
    import io

    import bench

    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main(["--gpus", str(world), "--steps", "2", "--warmup", "1", "--T", "16", "--chains", "2", "--no-cpu-baseline", "--no-parity"],
                   engine_cls=StubEngine, gpu_cls=bench._HostOnly,
                   extra_cls=dict(hgf_cls=StubHgf, gmm_cls=StubGmm, shard_cls=StubShard, c4_series=3, c4_T=20, c5_points=50, steps4=2, steps5=4))
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write(buf.getvalue())


def test_bench_main_emits_the_sharded_configs_at_world_8(tmp_path):
    """BASELINE configs 4 and 5 are DEFINED on 8 GPUs: an N > 1 run carries their sharded lines next to the headline"""
    import torch.multiprocessing as mp

    from rxhip import workloads

    world = 8
    mp.spawn(_worker_extras, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    lines = (tmp_path / "rank0.txt").read_text().strip().splitlines()
    assert len(lines) == 1 and all((tmp_path / f"rank{r}.txt").read_text() == "" for r in range(1, world))
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and set(out["extra"]) == {"c4", "c5"}, out.get("extra")
    c4, c5 = out["extra"]["c4"], out["extra"]["c5"]
    # c4: the global free energy is the rank-ordered sum of what every rank's shard (seed 42 + rank) produced
    sums = [float(np.sum(workloads.generate_hgf_batch(20, 3, seed=42 + r)[2])) for r in range(world)]
    tot = 0.0
    for x in sums:
        tot += x
    assert c4["free_energy_mean_per_series_global"][0] == pytest.approx(tot / (3 * world), rel=1e-12) and len(c4["free_energy_mean_per_series_global"]) == 10
    assert c4["series_observations_per_s"] == pytest.approx(20 * 3 * world / (c4["ms"] * 1e-3), rel=1e-4)   # (the printed line rounds measurements to 6 digits)
    # c5: one all-reduce per iteration made every rank's statistics global
    K = 16
    mus = np.arange(1, K + 1) * 10.0 - 80.0
    tot5 = 0.0
    for r in range(world):
        rng = np.random.default_rng(12345 + r)
        tot5 += float(np.sum(mus[rng.integers(0, K, size=50)] + rng.standard_normal(50)))
    assert c5["free_energy_last"] == pytest.approx(-tot5 / 6, rel=1e-9)       # the 6th update of the run (2 warm-up + 4 timed)
    assert c5["point_iterations_per_s"] == pytest.approx(50 * world * c5["vmp_iters_per_sec"], rel=1e-4)
    # the line is the compact form of the record: below the 8 KB the driver keeps, every configuration in roofline.per_config as [ms, roofline fraction, parity ok]
    assert len(lines[0]) < 8192 and set(out["roofline"]["per_config"]) == {"c4", "c5"} and out["roofline"]["per_config"]["c4"][0] == c4["ms"]


def _worker(rank, world, port, out_dir, scaling, chains):
    for p in (ROOT, os.path.join(ROOT, "rxinfer.jl_amd")):
        sys.path.insert(0, p)
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    import contextlib
    import io

    import bench

    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--T", "40", "--chains", str(chains), "--scaling", scaling,
                    "--no-cpu-baseline", "--no-parity", "--no-extras"], engine_cls=StubEngine, gpu_cls=bench._HostOnly)
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write(buf.getvalue())


@pytest.mark.parametrize("scaling,chains,per_rank", [("weak", 3, 3), ("strong", 6, 3)])
def test_bench_main_two_ranks(tmp_path, scaling, chains, per_rank):
    import torch.multiprocessing as mp

    from rxhip import workloads

    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), scaling, chains), nprocs=2, join=True)
    assert (tmp_path / "rank1.txt").read_text() == ""          # rank 0 alone prints
    lines = (tmp_path / "rank0.txt").read_text().strip().splitlines()
    assert len(lines) == 1                                      # ONE JSON line
    out = json.loads(lines[0])
    T, world = 40, 2
    assert out["n_gpus"] == world and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == scaling
    assert out["config"]["chains_per_gpu"] == per_rank and out["timing"] == "single" and out["cpu_baseline"] is None
    # chain c of the job comes from default_rng(42 + c): rank r owns chains [r·per_rank, (r + 1)·per_rank)
    mdl = workloads.c1_model()
    shard = [float(np.sum(workloads.generate_batch(mdl, T, per_rank, seed0=42 + r * per_rank))) for r in range(world)]
    assert out["free_energy_rank0"] == shard[0]
    assert out["free_energy_global"] == shard[0] + shard[1]     # all-gather + sum in ascending rank order: bit-identical
    # whole-job throughput: every rank's rule calls over the slowest rank's time
    calls = (6 * T - 3) * per_rank * world
    assert out["value"] == pytest.approx(calls * out["steps"] / (out["ms_per_step"] * 1e-3 * out["steps"]), rel=1e-4)   # (ms_per_step is printed with 6 digits)
    assert out["vmp_iters_per_sec"] == pytest.approx(1e3 / out["ms_per_step"], rel=1e-4)


def test_strong_scaling_needs_divisible_chains():
    import bench

    os.environ.update({"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2"})
    try:
        with pytest.raises(SystemExit):
            bench.main(["--gpus", "2", "--chains", "5", "--scaling", "strong", "--T", "8", "--no-cpu-baseline"], engine_cls=StubEngine, gpu_cls=bench._HostOnly)
    finally:
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
            os.environ.pop(k, None)


def test_timed_sweeps_takes_the_better_repeat_for_time_and_for_every_kernel():
    """The extra lines of bench.py time `repeats` identical measurements and keep the better one — the sweep time AND, since a stalled queue sits
    inside one kernel's event pair, every kernel of the instrumented pass (no GPU: an engine stub that plays back scripted kernel times)."""
    sys.path.insert(0, ROOT)
    import bench

    class Eng:
        passes = [{"k_forward": 3.7, "k_backward": 0.23}, {"k_forward": 0.27, "k_backward": 0.25}]

        def __init__(self):
            self.n, self.prof, self.k = 0, False, -1

        def run_async(self, iterations, fe):
            self.n += 1

        def sync(self):
            pass

        def set_profiling(self, on):
            self.prof = on

        def reset_kernel_times(self):
            self.k += 1

        def kernel_times(self):
            out = {name: {"ms_avg": ms, "launches": 5} for name, ms in self.passes[self.k].items()}
            out["k_unused"] = {"ms_avg": 0.0, "launches": 0}
            return out

    e = Eng()
    ms, kt = bench.timed_sweeps(e, steps=5, warmup=2, repeats=2)
    assert kt == {"k_forward": 0.27, "k_backward": 0.23} and ms >= 0.0
    assert e.n == 2 + 2 * 5 + 2 * 5 and e.prof is False
