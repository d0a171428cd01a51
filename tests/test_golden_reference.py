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
