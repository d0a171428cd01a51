"""World-size-2 `gloo` test of the multi-GPU host logic (chains shard, free energy all-reduced).
No GPU here, so the per-shard compute is the CPU oracle standing in for the engine — allowed in
tests only; the sharding / all-reduce code under test is the product's (rxhip/distributed.py)."""
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


def _worker(rank, world, port, out_dir):
    for p in (os.path.join(ROOT, "rxinfer.jl_amd"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import torch.distributed as dist

    import rxoracle
    from rxhip import distributed as rd
    from rxhip import workloads

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mdl = workloads.c1_model()
    C, T = 7, 60  # odd chain count: uneven shards
    y = np.transpose(workloads.generate_batch(mdl, T, C), (1, 0, 2))  # [chain][T][dy]

    def run_shard(yb):
        m, V, fe, _ = rxoracle.lgssm_bp_batch(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"],
                                              np.ascontiguousarray(np.transpose(yb, (1, 0, 2))))
        return np.transpose(m, (1, 0, 2)), np.transpose(V, (1, 0, 2, 3)), fe

    mean, cov, fe, fe_total = rd.sharded_infer(run_shard, y, dist)
    fe_all = rd.gather_per_chain(fe, C, dist)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), fe_total=fe_total, fe_all=fe_all, lo_hi=rd.shard_bounds(C, rank, world),
             mean=mean)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_and_balance():
    from rxhip.distributed import shard_bounds

    for n in (1, 7, 8, 1024, 4096):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_sharded_free_energy(tmp_path):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    # every rank sees the same global free energy and the same gathered per-chain values
    assert r0["fe_total"] == r1["fe_total"]
    assert np.array_equal(r0["fe_all"], r1["fe_all"])
    assert tuple(r0["lo_hi"]) == (0, 4) and tuple(r1["lo_hi"]) == (4, 7)
    # and it equals the unsharded result
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import rxoracle
    from rxhip import workloads

    mdl = workloads.c1_model()
    y = workloads.generate_batch(mdl, 60, 7)
    _, _, fe, _ = rxoracle.lgssm_bp_batch(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], y)
    assert np.allclose(r0["fe_all"], fe, rtol=0, atol=0)
    assert abs(float(r0["fe_total"]) - fe.sum()) <= 1e-12 * abs(fe.sum())


# ---- mixture: sharded points, statistics all-reduced every iteration (C5's exchange step) -----------------
_GMM = dict(N=1001, mus=[-6.0, 0.0, 7.0], seed=4,
            priors=([-4.0, 1.0, 5.0], [1e2] * 3, [0.1] * 3, [0.1] * 3, [1.0] * 3),
            init=([-4.0, 1.0, 5.0], [1.0] * 3, [1.0] * 3, [1.0] * 3, [1.0] * 3), iters=6)


def _gmm_data():
    rng = np.random.default_rng(_GMM["seed"])
    z = rng.integers(0, 3, size=_GMM["N"])
    return np.asarray(_GMM["mus"])[z] + rng.standard_normal(_GMM["N"])


class _OracleMixtureShard:
    """CPU stand-in for DeviceMixtureShard (tests only): the oracle's split-phase functions on this rank's points."""

    def __init__(self, rxoracle, y):
        import torch

        self.o, self.y = rxoracle, y
        self.state = np.ascontiguousarray(np.array(_GMM["init"], dtype=np.float64))  # [5][K]
        self.stats = np.zeros(3 * 3 + 1)
        self.t = torch.from_numpy(self.stats)  # aliases the buffer: all_reduce happens in place
        self.fe, self.hist = [], []

    def begin(self, iterations):
        self.fe, self.hist = [], []

    def accumulate(self):
        self.o.gmm_accumulate(self.y, self.state, self.stats)
        return self.t

    def update(self, want_fe):
        self.fe.append(self.o.gmm_update(*_GMM["priors"], self.stats, self.state, want_fe))
        self.hist.append(self.state.copy())


def _gmm_worker(rank, world, port, out_dir):
    for p in (os.path.join(ROOT, "rxinfer.jl_amd"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import torch.distributed as dist

    import rxoracle
    from rxhip import distributed as rd

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    y = _gmm_data()
    lo, hi = rd.shard_bounds(y.size, rank, world)
    shard = _OracleMixtureShard(rxoracle, y[lo:hi])
    rd.sharded_mixture_vmp(shard, _GMM["iters"], True, dist)
    np.savez(os.path.join(out_dir, f"gmm{rank}.npz"), fe=np.array(shard.fe), hist=np.array(shard.hist), lo_hi=(lo, hi))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_mixture(tmp_path):
    import torch.multiprocessing as mp

    mp.spawn(_gmm_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "gmm0.npz"), np.load(tmp_path / "gmm1.npz")
    assert tuple(r0["lo_hi"]) == (0, 501) and tuple(r1["lo_hi"]) == (501, 1001)
    # every rank holds the same global posteriors and the same global free energy, bit for bit
    assert np.array_equal(r0["fe"], r1["fe"]) and np.array_equal(r0["hist"], r1["hist"])
    # and they are the unsharded run's (summation order differs: rounding only)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import rxoracle

    hist, fe, _, _ = rxoracle.gmm_vmp(_gmm_data(), *_GMM["priors"], *_GMM["init"], _GMM["iters"])
    assert np.max(np.abs(r0["fe"] - fe) / np.abs(fe)) < 1e-12
    # (the Gamma rate is a difference of sums, S2 − 2 m S1 + m² S0: its rounding is amplified ≈ 10³×)
    assert np.max(np.abs(r0["hist"].reshape(hist.shape) - hist) / np.abs(hist)) < 1e-9
    assert np.all(np.diff(fe) < 1e-9)


def _ordered_sum_worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "rxinfer.jl_amd"))
    import torch
    import torch.distributed as dist

    from rxhip import distributed as rd

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # contributions whose sum depends on the order of the additions (1e16 + 1 − 1e16 …): 3K + 1 = 49 statistics as C5 exchanges them
    rng = np.random.default_rng(100 + rank)
    mine = rng.standard_normal(49) * 10.0 ** rng.integers(-8, 17, 49)
    runs, scratch = [], None
    for _ in range(3):
        t = torch.from_numpy(mine.copy())
        scratch = rd.allgather_ordered_sum_(t, dist, scratch)
        runs.append(t.numpy().copy())
    np.savez(os.path.join(out_dir, f"sum{rank}.npz"), runs=np.array(runs), mine=mine)
    dist.barrier()
    dist.destroy_process_group()


def test_the_mixture_statistics_exchange_is_bit_identical_on_every_rank_and_every_run(tmp_path):
    """C5's one exchange (3K + 1 statistics per VMP iteration) is an all-gather + one local reduction over the rank axis — the same bytes through the same
    kernel on every rank — instead of an all-reduce, whose summation order is the backend's (ring or tree, chosen per message size and topology): every
    rank of a world-8 run holds the same bits, run after run (the 1e-8 run-to-run free-energy tolerance), and they are the reduction of the
    rank-ordered contributions."""
    import torch
    import torch.multiprocessing as mp

    world = 8
    mp.spawn(_ordered_sum_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [np.load(tmp_path / f"sum{r}.npz") for r in range(world)]
    want = torch.sum(torch.from_numpy(np.stack([r["mine"] for r in res])), dim=0).numpy()
    for r in res:
        assert all(np.array_equal(run, want) for run in r["runs"])
    exact = np.array([float(sum(map(lambda v: __import__("fractions").Fraction(float(v)), col))) for col in np.stack([r["mine"] for r in res]).T])
    assert np.max(np.abs(want - exact) / np.maximum(np.abs(exact), 1e-300)) < 1e-6   # (a sum, not something else: cancellation bounds the accuracy, the bits are what is pinned)
