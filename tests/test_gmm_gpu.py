"""GPU parity tests for the mean-field VMP families: the HIP path against the CPU
oracle's restatement on the same seeded inputs, every iteration (posteriors 1e-6 relative, free energy 1e-8).
The oracle's VMP schedule is an assumption (reference order undocumented): parity against the real
reference is unpinned for intermediate iterates; see DESIGN.md §5."""
import numpy as np
import pytest

import rxhip
import rxoracle

pytestmark = pytest.mark.gpu


def gmm_data(n, mus, ws, probs, seed):
    rng = np.random.default_rng(seed)
    z = rng.choice(len(mus), size=n, p=probs)
    return np.asarray(mus)[z] + rng.standard_normal(n) / np.sqrt(np.asarray(ws)[z])


def run_both(y, priors, init, iters, materialize=False):
    with rxhip.GMMEngine(y.size, *priors, *init, materialize_responsibilities=materialize) as eng:
        eng.set_data(y)
        eng.run(iters, True)
        hist, fe, cnt = eng.history(), eng.free_energy(), eng.counters()
        resp = eng.responsibilities() if materialize else None
    ohist, ofe, oresp, ocnt = rxoracle.gmm_vmp(y, *priors, *init, iters, want_resp=materialize)
    return hist, fe, resp, cnt, ohist, ofe, oresp, ocnt


def assert_parity(hist, fe, ohist, ofe):
    assert np.max(np.abs(hist - ohist) / np.maximum(np.abs(ohist), 1e-300)) < 1e-6
    assert np.max(np.abs(fe - ofe) / np.abs(ofe)) < 1e-8


def test_gmm_univariate_reference_test_shape():
    """test/models/mixtures/gmm_univariate_tests.jl: K = 2, n = 150, 10 iterations, same priors / initialisation."""
    y = gmm_data(150, [-10.0, 10.0], [3.777, 0.333], [1 / 3, 2 / 3], 12345)
    priors = ([-2.0, 2.0], [1e3, 1e3], [0.01, 0.01], [0.01, 0.01], [1.0, 1.0])
    init = ([-2.0, 2.0], [1e3, 1e3], [1.0, 1.0], [1e-12, 1e-12], [1.0, 1.0])  # vague(GammaShapeRate), vague(Beta)
    hist, fe, resp, cnt, ohist, ofe, oresp, ocnt = run_both(y, priors, init, 10, materialize=True)
    assert_parity(hist, fe, ohist, ofe)
    assert np.max(np.abs(resp - oresp)) < 1e-9
    assert np.all(np.diff(fe) <= 1e-10)  # FE non-increasing, gmm_univariate_tests.jl:96
    assert np.all(np.abs(np.sort(hist[-1, 0]) - np.array([-10.0, 10.0])) < 1.0)  # clusters found from the vague initialisation
    assert cnt["rule_calls"] == ocnt.rule_calls and cnt["products"] == ocnt.products


@pytest.mark.parametrize("K,n,seed", [(1, 1000, 1), (3, 5000, 2), (5, 20000, 3), (16, 100000, 4), (7, 333, 5)])
def test_gmm_k_components(K, n, seed):
    mus = np.linspace(-10 * K / 2, 10 * K / 2, K) if K > 1 else np.array([0.75])
    y = gmm_data(n, mus, np.ones(K), np.ones(K) / K, seed)
    priors = (mus + 1.0, np.full(K, 1e3), np.full(K, 0.01), np.full(K, 0.01), np.ones(K))
    init = (mus + 2.0, np.full(K, 10.0), np.ones(K), np.ones(K), np.ones(K))
    hist, fe, _, cnt, ohist, ofe, _, ocnt = run_both(y, priors, init, 8)
    assert_parity(hist, fe, ohist, ofe)
    assert np.all(np.abs(hist[-1, 0] - mus) < 1.0)  # component means recovered


def test_iid_gaussian_unknown_mean_and_precision():
    """K = 1: `y[i] ~ Normal(mean = μ, precision = τ)`, μ ~ Normal(4, 8), τ ~ Gamma(shape 4, scale 8) — the
    iid_gaussians_params model of test/models/models_tests.jl:114-128, init q(μ) = N(0,1), q(τ) = Gamma(1,1)."""
    y = 0.75 + 10.0 * np.random.default_rng(123).standard_normal(100)
    priors = ([4.0], [8.0], [4.0], [1.0 / 8.0], [1.0])
    init = ([0.0], [1.0], [1.0], [1.0], [1.0])
    hist, fe, _, _, ohist, ofe, _, _ = run_both(y, priors, init, 10)
    assert_parity(hist, fe, ohist, ofe)
    assert np.all(np.diff(fe) <= 1e-9)
    # fixed point: closed-form conjugate relations hold at convergence
    m, v, a, b = hist[-1, 0, 0], hist[-1, 1, 0], hist[-1, 2, 0], hist[-1, 3, 0]
    Ep = a / b
    assert abs(1.0 / v - (1 / 8.0 + Ep * 100)) < 1e-6 * (1.0 / v)
    assert abs(a - (4.0 + 50.0)) < 1e-12


def test_iid_gaussian_gamma_scale_spelling_equals_rate_spelling():
    """test/models/models_tests.jl:114-199: `iid_gaussians_priors` (τ ~ Gamma(4, 8), Distributions' shape/scale) and
    `iid_gaussians_params` (τ ~ Gamma(shape = 4, scale = 8)) give `mean.(posteriors[:μ]) ≈`, `mean.(posteriors[:τ]) ≈`,
    `free_energy ≈` of one another (models_tests.jl:183-195).  Both build a GammaShapeScale node; on the device that node and
    the rate spelling with β = 1/8 run the same engine, bit for bit, and match the oracle."""
    from rxhip import graph
    y = 0.75 + 10.0 * np.random.default_rng(123).standard_normal(100)
    init = dict(m=(0.0, 1.0), p=(1.0, 1.0))
    out = []
    for kw in (dict(scale=8.0), dict(rate=0.125)):
        gb, _ = graph.iid_normal_graph(100, 4.0, 8.0, 4.0, init=init, **kw)
        with graph.create_vmp_engine_from_graph(gb.tables()[0]) as eng:
            eng.set_data(y)
            eng.run(10, True)
            out.append((eng.history().copy(), eng.free_energy().copy()))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    ohist, ofe, _, _ = rxoracle.gmm_vmp(y, [4.0], [8.0], [4.0], [1 / 8.0], [1.0], [0.0], [1.0], [1.0], [1.0], [1.0], 10)
    assert_parity(out[0][0], out[0][1], ohist, ofe)
    res = rxhip.infer(model=rxhip.iid_normal_gamma(4.0, 8.0, 4.0, scale=8.0), data={"y": y}, iterations=10, free_energy=True,
                      initialization={"m": rxhip.NormalMeanVariance(0.0, 1.0), "p": rxhip.GammaShapeScale(1.0, 1.0)})
    assert np.allclose(res.free_energy, out[0][1], rtol=0, atol=0)


def test_split_phase_equals_run():
    """accumulate / update (the multi-GPU split with the statistics exposed for the RCCL all-reduce)."""
    y = gmm_data(4096, [-5.0, 0.0, 5.0], [1, 1, 1], [0.3, 0.3, 0.4], 9)
    priors = ([-4.0, 1.0, 4.0], [1e2] * 3, [0.1] * 3, [0.1] * 3, [1.0] * 3)
    init = ([-4.0, 1.0, 4.0], [1.0] * 3, [1.0] * 3, [1.0] * 3, [1.0] * 3)
    with rxhip.GMMEngine(y.size, *priors, *init) as eng:
        eng.set_data(y)
        eng.run(5, True)
        h1, f1 = eng.history(), eng.free_energy()
        eng.begin_run(5)
        for _ in range(5):
            eng.accumulate()
            ptr, n = eng.statistics_device()
            assert ptr and n == 3 * 4 + 1
            eng.update(True)
        eng.sync()
        assert np.array_equal(eng.history(), h1) and np.array_equal(eng.free_energy(), f1)  # deterministic


def test_infer_mirror_for_vmp_models():
    """`infer(model = …, data = (y = …,), constraints = MeanField(), initialization = …, iterations = 10, free_energy = true)`
    mirror for the mixture / iid Gaussian×Gamma / HGF models."""
    y = 0.75 + 10.0 * np.random.default_rng(123).standard_normal(100)
    res = rxhip.infer(model=rxhip.iid_normal_gamma(4.0, 8.0, 4.0, 1.0 / 8.0), data={"y": y}, iterations=10, free_energy=True,
                      initialization={"m": rxhip.NormalMeanVariance(0.0, 1.0), "p": rxhip.GammaShapeRate(1.0, 1.0)})
    ohist, ofe, _, _ = rxoracle.gmm_vmp(y, [4.0], [8.0], [4.0], [1 / 8.0], [1.0], [0.0], [1.0], [1.0], [1.0], [1.0], 10)
    assert res.posteriors["m"].mean.shape == (10, 1) and np.allclose(res.posteriors["m"].mean[:, 0], ohist[:, 0, 0], rtol=1e-9)
    assert np.allclose(res.posteriors["p"].rate[:, 0], ohist[:, 3, 0], rtol=1e-9) and np.allclose(res.free_energy, ofe, rtol=1e-9)
    with pytest.raises(ValueError):
        rxhip.infer(model=rxhip.iid_normal_gamma(4.0, 8.0, 4.0, 0.125), data={"y": y}, iterations=3)  # no initialization
    rng = np.random.default_rng(1)
    yh = np.cumsum(rng.standard_normal(300))
    r2 = rxhip.infer(model=rxhip.hierarchical_gaussian_filter(1.0, 0.0, 0.04, 0.01), data={"y": yh}, iterations=5, free_energy=True)
    o = rxoracle.hgf_filter(yh, 1.0, 0.0, 0.04, 0.01, vmp_iters=5)
    assert np.allclose(r2.posteriors["zt"].mean, o[0], rtol=1e-8, atol=1e-10) and np.allclose(r2.free_energy, o[4], rtol=1e-8)


def _mp_worker(rank, world, port, out_dir):
    """One process per shard on the SAME GPU (the box has one); gloo moves the device statistics — on the 8-GPU
    node the identical code runs with backend nccl (= RCCL)."""
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "rxinfer.jl_amd"))
    import torch
    import torch.distributed as dist

    import rxhip as rx
    from rxhip import distributed as rd

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    y = gmm_data(20001, [-5.0, 0.0, 5.0], [1, 1, 1], [0.3, 0.3, 0.4], 9)
    lo, hi = rd.shard_bounds(y.size, rank, world)
    priors = ([-4.0, 1.0, 4.0], [1e2] * 3, [0.1] * 3, [0.1] * 3, [1.0] * 3)
    init = ([-4.0, 1.0, 4.0], [1.0] * 3, [1.0] * 3, [1.0] * 3, [1.0] * 3)
    with rx.GMMEngine(hi - lo, *priors, *init, device=0) as eng:
        eng.set_data(y[lo:hi])
        rd.sharded_mixture_vmp(rd.DeviceMixtureShard(eng), 6, True, dist)
        eng.sync()
        np.savez(os.path.join(out_dir, f"gmm{rank}.npz"), hist=eng.history(), fe=eng.free_energy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_process_sharded_mixture(tmp_path):
    """C5's exchange step end to end: two processes, each with half of the points on its engine, the 3K'+1 device
    statistics all-reduced in place every iteration — equals the single-engine run."""
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_mp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "gmm0.npz"), np.load(tmp_path / "gmm1.npz")
    assert np.array_equal(r0["hist"], r1["hist"]) and np.array_equal(r0["fe"], r1["fe"])
    y = gmm_data(20001, [-5.0, 0.0, 5.0], [1, 1, 1], [0.3, 0.3, 0.4], 9)
    priors = ([-4.0, 1.0, 4.0], [1e2] * 3, [0.1] * 3, [0.1] * 3, [1.0] * 3)
    init = ([-4.0, 1.0, 4.0], [1.0] * 3, [1.0] * 3, [1.0] * 3, [1.0] * 3)
    with rxhip.GMMEngine(y.size, *priors, *init) as eng:
        eng.set_data(y)
        eng.run(6, True)
        h, f = eng.history(), eng.free_energy()
    assert np.max(np.abs(r0["fe"] - f) / np.abs(f)) < 1e-10 and np.max(np.abs(r0["hist"] - h) / np.abs(h)) < 1e-8
    ohist, ofe, _, _ = rxoracle.gmm_vmp(y, *priors, *init, 6)
    assert np.max(np.abs(r0["fe"] - ofe) / np.abs(ofe)) < 1e-8


def test_non_finite_observation_is_reported():
    """A NaN among the observations must not disappear in the softmax (the pass kernel's exp clamps its argument): the run
    fails with a non-finite free energy, as the reference's NaN check (src/score/diagnostics.jl:19-51) would."""
    y = gmm_data(500, [-10.0, 10.0], [3.777, 0.333], [1 / 3, 2 / 3], 7)
    y[123] = np.nan
    priors = ([-2.0, 2.0], [1e3, 1e3], [0.01, 0.01], [0.01, 0.01], [1.0, 1.0])
    init = ([-2.0, 2.0], [1e3, 1e3], [1.0, 1.0], [1e-12, 1e-12], [1.0, 1.0])
    with rxhip.GMMEngine(y.size, *priors, *init) as eng:
        eng.set_data(y)
        with pytest.raises(rxhip.RxHipError):
            eng.run(3, True)
            eng.free_energy()
