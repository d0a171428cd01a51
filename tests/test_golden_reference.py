"""Known-answer tests against the REAL reference: the RNG-dependent golden values asserted by RxInfer's own
tests, on data regenerated bit-faithfully (oracle/stable_rng.py; fixtures + generator under tests/golden/).
These pin the oracle — and through it the HIP path — to numbers produced by ReactiveMP itself."""
import math
import os

import numpy as np
import pytest

import rxoracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_fixtures_are_reproducible_from_the_committed_generator():
    import stable_rng

    g = np.load(os.path.join(GOLD, "mlgssm_stablerng1234.npz"))
    rng = stable_rng.StableRNG(1234)
    x0 = rng.mvnormal(g["A"] @ np.array([10.0, -10.0]), g["state_noise"])
    assert np.array_equal(x0, g["x"][0])
    assert stable_rng.StableRNG(1).rand_u64() == ((3 * 0x45A31EFC5A35D971261FD0407A968ADD) % (1 << 128)) >> 64


def test_mlgssm_golden_free_energy_cpu():
    """test/models/statespace/mlgssm_test.jl:86-135: FE = 6275.9015944677 (atol 0.01), posteriors PD and within
    mean ± 3·var of the truth, for the `x_prior; x[i] ~ MvNormal(A*x_prev, Q)` spelling."""
    g = np.load(os.path.join(GOLD, "mlgssm_stablerng1234.npz"))
    m, V, fe, _ = rxoracle.lgssm_bp(g["A"], g["B"], g["state_noise"], g["obs_noise"], g["prior_mean"], g["prior_cov"], g["y"],
                                    prior_through_transition=True)
    assert abs(fe - float(g["fe_reference"])) < 1e-6  # the reference prints 13 significant digits; we match them all
    var = np.stack([np.diag(v) for v in V])
    assert np.all((m - 3 * var < g["x"]) & (g["x"] < m + 3 * var))  # mlgssm_test.jl:119-125
    assert np.all(np.linalg.eigvalsh(V) > 0)


def test_ulgssm_golden_free_energy_closed_form():
    """test/models/statespace/ulgssm_tests.jl:27-48 (deterministic `x[i] ~ x_prev + c`, no process noise): the Bethe
    free energy is −log N(y − hidden; 0, prior_var·11' + P·I); golden 1854.297647.  Second pin of the RNG restatement."""
    g = np.load(os.path.join(GOLD, "ulgssm_stablerng123.npz"))
    r = g["y"] - g["hidden"]
    n = r.size
    S = float(g["prior_var"]) * np.ones((n, n)) + float(g["obs_var"]) * np.eye(n)
    fe = 0.5 * (n * math.log(2 * math.pi) + np.linalg.slogdet(S)[1] + r @ np.linalg.solve(S, r))
    assert abs(fe - float(g["fe_reference"])) < 1e-5


def test_hgf_reference_data_statistics_cpu():
    """test/models/statespace/hgf_tests.jl:119-133 on the reference's own data: ≥95 % of the truth within 3σ, all
    within 6σ, FE decreasing, and the golden FE 1.009879989585 at iteration 10 (reference tolerance 0.01; the oracle
    gives 1.00987052)."""
    g = np.load(os.path.join(GOLD, "hgf_stablerng42.npz"))
    zm, zv, xm, xv, fe, _ = rxoracle.hgf_filter(g["y"], float(g["kappa"]), float(g["omega"]), float(g["z_variance"]), float(g["y_variance"]))
    z, x = g["z"], g["x"]
    assert np.all(np.abs(zm - z) < 6 * np.sqrt(zv)) and np.all(np.abs(xm - x) < 6 * np.sqrt(xv))
    assert np.mean(np.abs(zm - z) < 3 * np.sqrt(zv)) > 0.95 and np.mean(np.abs(xm - x) < 3 * np.sqrt(xv)) > 0.95
    assert np.all(np.diff(fe) < 1e-9)
    assert abs(fe[-1] - float(g["fe_reference_it10"])) < 1e-4  # hgf_tests.jl:113 asserts 0.01


@pytest.mark.gpu
def test_mlgssm_golden_free_energy_gpu():
    """The HIP path on the reference's data reproduces the reference's golden free energy."""
    import rxhip

    g = np.load(os.path.join(GOLD, "mlgssm_stablerng1234.npz"))
    spec = rxhip.linear_gaussian_ssm(g["A"], g["B"], g["state_noise"], g["obs_noise"], g["prior_mean"], g["prior_cov"],
                                     prior_through_transition=True)
    res = rxhip.infer(model=spec, data={"y": g["y"]}, free_energy=True, options={"limit_stack_depth": 500})
    assert res.free_energy.shape == (1,)
    assert abs(res.free_energy[-1] - float(g["fe_reference"])) < 1e-6   # reference tolerance is 0.01
    m, V = res.posteriors["x"].mean, res.posteriors["x"].cov
    var = np.stack([np.diag(v) for v in V])
    assert np.all((m - 3 * var < g["x"]) & (g["x"] < m + 3 * var)) and np.all(np.linalg.eigvalsh(V) > 0)


@pytest.mark.gpu
def test_hgf_reference_data_gpu_matches_oracle():
    import rxhip

    g = np.load(os.path.join(GOLD, "hgf_stablerng42.npz"))
    with rxhip.HGFEngine(g["y"].size, 1, float(g["kappa"]), float(g["omega"]), float(g["z_variance"]), float(g["y_variance"])) as eng:
        eng.set_data(g["y"][:, None])
        eng.run(10, True)
        zm, zv, xm, xv = eng.history()
        fe = eng.free_energy()
    o = rxoracle.hgf_filter(g["y"], float(g["kappa"]), float(g["omega"]), float(g["z_variance"]), float(g["y_variance"]))
    assert np.max(np.abs(zm[:, 0] - o[0])) < 1e-6 * np.max(np.abs(o[0])) and np.max(np.abs(fe - o[4]) / np.abs(o[4])) < 1e-8
    assert abs(fe[-1] - float(g["fe_reference_it10"])) < 1e-4  # the reference's golden, hgf_tests.jl:113


def _mvgmm_fixture():
    g = np.load(os.path.join(GOLD, "mvgmm_stablerng43.npz"))
    K = g["prior_mean"].shape[0]
    S0 = np.tile(g["prior_cov"], (K, 1, 1))
    nu0 = np.full(K, float(g["wishart_nu"]))
    V0 = np.tile(g["wishart_scale"], (K, 1, 1))
    return g, K, S0, nu0, V0


def test_multivariate_mixture_golden_free_energy_cpu():
    """test/models/mixtures/gmm_multivariate_tests.jl:80-141: FE after 25 iterations = 3436.7 (atol 0.1), FE decreasing
    (differences above 1e-3), estimated mean directions within 0.1 of the true ones — on the regenerated data (fixture notes
    in tests/golden/make_golden.py: the alias-table layout of the label sampler is inferred)."""
    g, K, S0, nu0, V0 = _mvgmm_fixture()
    h, fe, _ = rxoracle.mvgmm_vmp(g["y"], g["prior_mean"], S0, nu0, V0, np.ones(K),
                                  rxoracle.mvgmm_pack(g["init_mean"], S0, nu0, V0, np.ones(K)), int(g["iterations"]))
    assert abs(fe[-1] - float(g["fe_reference_it25"])) < float(g["fe_atol"])
    d = np.diff(fe)
    assert np.all(d[np.abs(d) > 1e-3] < 0)
    em = rxoracle.mvgmm_unpack(h[-1], 2)["mean"]
    key = lambda v: math.atan(v[1] / v[0])
    for e, r in zip(sorted(em, key=key), sorted(g["true_means"], key=key)):
        assert np.linalg.norm(e / np.linalg.norm(e) - r / np.linalg.norm(r)) < 0.1


@pytest.mark.gpu
def test_multivariate_mixture_golden_free_energy_gpu():
    """The HIP path reproduces the reference's golden free energy of the multivariate mixture test."""
    import rxhip

    g, K, S0, nu0, V0 = _mvgmm_fixture()
    spec = rxhip.multivariate_gaussian_mixture(g["prior_mean"], S0, nu0, V0)
    res = rxhip.infer(model=spec, data={"y": g["y"]}, iterations=int(g["iterations"]), free_energy=True,
                      initialization={"m": rxhip.MvNormalMeanCovariance(g["init_mean"], S0), "w": rxhip.Wishart(nu0, V0),
                                      "s": rxhip.Dirichlet(np.ones(K))})
    assert res.free_energy.shape == (25,)
    assert abs(res.free_energy[-1] - float(g["fe_reference_it25"])) < float(g["fe_atol"])


def _uv_mixture_fe(y):
    _, fe, _, _ = rxoracle.gmm_vmp(y, [-2.0, 2.0], [1e3, 1e3], [0.01, 0.01], [0.01, 0.01], [1.0, 1.0], [-2.0, 2.0], [1e3, 1e3],
                                   [1.0, 1.0], [1e-12, 1e-12], [1.0, 1.0], 10)
    return fe


def test_univariate_mixture_golden_is_bounded_but_not_reproduced():
    """test/models/mixtures/gmm_univariate_tests.jl:42-60,97 asserts FE₁₀ ≈ 284.76 ± 0.1.  This is the one reference-held
    number of the path that the restated data stream does NOT reproduce, and the test documents what is known instead of
    hiding it:
    (1) three restatements of `rand(rng, Categorical([1/3, 2/3]), 150)` — AliasTables.jl (Distributions ≥ 0.25.109, what the
        Manifest pins and what reproduces the multivariate fixture's labels), the older two-draw StatsBase alias sampler, and
        the single-draw inverse CDF — put 42…57 of the 150 points into the tight cluster (as p = 1/3 must) and give
        FE₁₀ = 354…364;
    (2) a free energy is an upper bound of −log p(y) ≥ −log p(y | θ̂): with n₁ ≈ 50 points of entropy 0.755 nats, 100 of
        1.969 nats and the label entropy 150·H(1/3) = 95.5 that is ≥ 325 before any parameter cost, so NO correct inference
        run on data drawn with p = [1/3, 2/3] can report 284.76;
    (3) the same streams with the two cluster labels exchanged (2/3 of the points tight) give 274…314 — the golden lies
        inside that band.  The reference's label vector therefore has ≈ 100+ tight points, which a sampler of [1/3, 2/3]
        cannot produce: the discrepancy is in the reference's data (or its recorded constant), not in the mixture rules —
        those are pinned by the multivariate golden (3436.7 ± 0.1) and the d = 1 reduction (tests/test_mvgmm_gpu.py)."""
    import stable_rng

    g = np.load(os.path.join(GOLD, "uvgmm_stablerng12345.npz"))
    golden = float(g["fe_reference_it10"])
    mu, sig = g["mu"], 1.0 / np.sqrt(g["w"])
    expect = {"categorical_alias_table": (57, 357.686), "categorical_legacy_alias": (42, 354.473), "categorical_inverse_cdf": (46, 363.919)}
    swapped = []
    for name, (n_tight, fe_want) in expect.items():
        rng = stable_rng.StableRNG(12345)
        z = np.array([getattr(rng, name)([1 / 3, 2 / 3]) for _ in range(150)])
        eps = np.array([rng.randn() for _ in range(150)])
        y = mu[z] + sig[z] * eps
        fe = _uv_mixture_fe(y)
        assert int(np.sum(z == 0)) == n_tight and abs(fe[-1] - fe_want) < 5e-3 and np.all(np.diff(fe) <= 1e-10)
        if name == "categorical_alias_table":
            assert np.array_equal(y, g["y"])
        # (2) lower bound from the data alone: Gaussian entropies at the sample variances + label entropy, no parameter cost
        n1 = n_tight
        bound = 0.5 * n1 * (math.log(2 * math.pi * math.e * np.var(y[z == 0]))) + 0.5 * (150 - n1) * math.log(2 * math.pi * math.e * np.var(y[z == 1])) \
            - n1 * math.log(n1 / 150) - (150 - n1) * math.log(1 - n1 / 150)
        assert bound > golden + 30 and fe[-1] > bound
        ys = mu[1 - z] + sig[1 - z] * eps          # (3) labels exchanged: 2/3 of the points in the tight cluster
        swapped.append(_uv_mixture_fe(ys)[-1])
    assert min(swapped) < golden < max(swapped) and max(swapped) - min(swapped) < 45  # 274.3 … 314.0


@pytest.mark.gpu
def test_univariate_mixture_reference_data_gpu_matches_oracle():
    """The HIP path on the regenerated data of gmm_univariate_tests.jl (whatever the status of the golden constant)."""
    import rxhip

    g = np.load(os.path.join(GOLD, "uvgmm_stablerng12345.npz"))
    res = rxhip.infer(model=rxhip.gaussian_mixture([-2.0, 2.0], [1e3, 1e3], [0.01, 0.01], [0.01, 0.01]), data={"y": g["y"]}, iterations=10,
                      free_energy=True, initialization={"m": rxhip.NormalMeanVariance(np.array([-2.0, 2.0]), np.array([1e3, 1e3])),
                                                        "p": rxhip.GammaShapeRate(np.ones(2), np.full(2, 1e-12))})
    ofe = _uv_mixture_fe(g["y"])
    assert np.max(np.abs(res.free_energy - ofe) / np.abs(ofe)) < 1e-8 and np.all(np.diff(res.free_energy) <= 1e-10)
    # the reference's own structural assertions (:99-121): switch near 1/3 or 2/3, cluster means / precisions within 3 sd
    ms = res.posteriors["s"].alpha[-1] / res.posteriors["s"].alpha[-1].sum()
    assert min(abs(ms[0] - 1 / 3), abs(ms[0] - 2 / 3)) < 0.1
    m, v = res.posteriors["m"].mean[-1], res.posteriors["m"].var[-1]
    assert np.all(np.abs(np.sort(m) - np.array([-10.0, 10.0])) < 3 * np.sqrt(v[np.argsort(m)]))


_REF_DUMP = os.path.join(GOLD, "rxinfer_reference.json")


@pytest.mark.skipif(not os.path.exists(_REF_DUMP), reason="needs tests/golden/rxinfer_reference.json from a real RxInfer run "
                    "(tests/golden/dump_rxinfer_reference.jl; no Julia toolchain in the build image)")
def test_against_a_real_rxinfer_dump():
    """Hard versions of what DESIGN §5 can only bound or assume, for whoever has Julia: the RNG restatement draw by draw,
    the label vector of the univariate mixture test, and the per-iteration posteriors / free energies (the VMP update order)."""
    import json

    import stable_rng

    ref = json.load(open(_REF_DUMP))
    r = stable_rng.StableRNG(12345)
    assert [r.rand_u64() for _ in range(4)] == ref["draws"]["u64"]
    assert [r.rand() for _ in range(4)] == ref["draws"]["rand"] and [r.randn() for _ in range(4)] == ref["draws"]["randn"]
    r7 = stable_rng.StableRNG(7)
    assert [r7.categorical_alias_table([0.1, 0.2, 0.3, 0.25, 0.15]) + 1 for _ in range(32)] == ref["draws"]["categorical"]
    y = np.asarray(ref["y"])
    hist, fe, _, _ = rxoracle.gmm_vmp(y, [-2.0, 2.0], [1e3, 1e3], [0.01, 0.01], [0.01, 0.01], [1.0, 1.0], [-2.0, 2.0], [1e3, 1e3],
                                      [1.0, 1.0], [1e-12, 1e-12], [1.0, 1.0], 10)
    assert abs(fe[-1] - ref["free_energy"][-1]) < 1e-6 * abs(fe[-1])            # the fixed point
    assert np.allclose(fe, ref["free_energy"], rtol=1e-8)                        # … and the path to it: the update order
    assert np.allclose(hist[:, 0], ref["m_mean"], rtol=1e-6) and np.allclose(hist[:, 3], ref["p_rate"], rtol=1e-6)
