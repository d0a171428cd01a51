"""GPU parity tests of the multivariate Gaussian mixture engine (MvNormal components, Wishart precisions) against the
oracle, every iteration; model and driver shaped after test/models/mixtures/gmm_multivariate_tests.jl.
Tolerances: posteriors 1e-6 relative, free energy 1e-8 relative."""
import numpy as np
import pytest

import rxhip
import rxoracle

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def ring_data(N, K, d, L, seed):
    """clusters on a ring (the reference test's layout, gmm_multivariate_tests.jl:86-103), numpy default_rng"""
    rng = np.random.default_rng(seed)
    means, covs = [], []
    for k in range(K):
        v = rng.standard_normal(d)
        v *= L / np.linalg.norm(v)
        A = rng.standard_normal((d, d))
        means.append(v)
        covs.append(A @ A.T + np.diag(rng.uniform(5.0, 20.0, d)))
    z = rng.integers(0, K, N)
    y = np.stack([rng.multivariate_normal(means[k], covs[k]) for k in z])
    return y, np.array(means), np.array(covs)


def setup(K, d, means, seed):
    rng = np.random.default_rng(seed)
    mu0 = 0.5 * means + rng.uniform(0, 5, (K, d))
    S0 = np.tile(1e6 * np.eye(d), (K, 1, 1))
    nu0 = np.full(K, d + 1.0)
    V0 = np.tile(1e2 * np.eye(d), (K, 1, 1))
    al0 = np.ones(K)
    return mu0, S0, nu0, V0, al0


@pytest.mark.parametrize("d,K,N,iters", [(2, 3, 500, 25), (1, 3, 777, 8), (2, 16, 4000, 6), (3, 5, 1500, 10), (3, 8, 2000, 5),
                                          (4, 2, 900, 8), (4, 8, 3000, 4), (2, 1, 300, 5)])
def test_matches_oracle_every_iteration(d, K, N, iters):
    y, means, covs = ring_data(N, K, d, 50.0, seed=10 * d + K)
    mu0, S0, nu0, V0, al0 = setup(K, d, means, seed=K)
    init = (mu0, S0, nu0, V0, np.ones(K))
    with rxhip.MvGMMEngine(N, mu0, S0, nu0, V0, al0, *init, materialize_responsibilities=True) as eng:
        eng.set_data(y)
        eng.run(iters, True)
        h, fe, resp = eng.history(), eng.free_energy(), eng.responsibilities()
    ohist, ofe, oresp = rxoracle.mvgmm_vmp(y, mu0, S0, nu0, V0, al0, rxoracle.mvgmm_pack(*init), iters, want_resp=True)
    o = rxoracle.mvgmm_unpack(ohist, d)
    for key in ("mean", "cov", "nu", "V", "alpha"):
        assert rel(h[key], o[key]) < 1e-6, key
    assert np.max(np.abs(fe - ofe) / np.abs(ofe)) < 1e-8
    assert np.max(np.abs(resp - oresp)) < 1e-9
    assert np.all(np.diff(fe) < 1e-6 * np.abs(fe[-1]))  # free energy non-increasing (gmm_multivariate_tests.jl:139)


def test_reference_shaped_run_recovers_the_clusters():
    """K = 3, d = 2, N = 500, 25 iterations (gmm_multivariate_tests.jl:80-149): estimated means point at the true ones."""
    rng = np.random.default_rng(43)
    ang = 2 * np.pi / 3 * np.arange(3)
    means = 50.0 * np.stack([np.cos(ang), np.sin(ang)], axis=1)
    rot = [np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]) for a in ang]
    covs = [R @ np.diag([10.0, 20.0]) @ R.T for R in rot]
    y = np.stack([rng.multivariate_normal(means[k], covs[k]) for k in rng.integers(0, 3, 500)])
    mu0, S0, nu0, V0, al0 = setup(3, 2, means, seed=42)
    spec = rxhip.multivariate_gaussian_mixture(mu0, S0, nu0, V0)
    res = rxhip.infer(model=spec, data={"y": y}, iterations=25, free_energy=True,
                      initialization={"m": rxhip.MvNormalMeanCovariance(mu0, S0), "w": rxhip.Wishart(nu0, V0), "s": rxhip.Dirichlet(np.ones(3))})
    assert res.free_energy.shape == (25,) and np.all(np.diff(res.free_energy) < 1e-6 * abs(res.free_energy[-1]))
    em = res.posteriors["m"].mean[-1]
    for k in range(3):
        assert np.linalg.norm(em[k] / np.linalg.norm(em[k]) - means[k] / np.linalg.norm(means[k])) < 0.1
    ohist, ofe, _ = rxoracle.mvgmm_vmp(y, mu0, S0, nu0, V0, al0, rxoracle.mvgmm_pack(mu0, S0, nu0, V0, np.ones(3)), 25)
    assert abs(res.free_energy[-1] - ofe[-1]) < 1e-8 * abs(ofe[-1])


def test_univariate_engine_agrees_for_d1():
    """Wishart(ν, V) in one dimension is Gamma(ν/2, 1/(2V)): the two device engines give the same answers."""
    rng = np.random.default_rng(8)
    N, K = 5000, 4
    mus = np.array([-9.0, -2.0, 3.0, 11.0])
    y = mus[rng.integers(0, K, N)] + rng.standard_normal(N)
    mu0, v0, a0, b0, al0 = mus + 1.0, np.full(K, 1e2), np.full(K, 0.5), np.full(K, 0.25), np.ones(K)
    with rxhip.GMMEngine(N, mu0, v0, a0, b0, al0, mu0, np.ones(K), np.ones(K), np.ones(K), np.ones(K)) as e1:
        e1.set_data(y); e1.run(7, True)
        h1, f1 = e1.history(), e1.free_energy()
    with rxhip.MvGMMEngine(N, mu0[:, None], v0[:, None, None], 2 * a0, (1 / (2 * b0))[:, None, None], al0, mu0[:, None],
                           np.ones((K, 1, 1)), 2 * np.ones(K), 0.5 * np.ones((K, 1, 1)), np.ones(K)) as e2:
        e2.set_data(y[:, None]); e2.run(7, True)
        h2, f2 = e2.history(), e2.free_energy()
    assert np.max(np.abs(f1 - f2) / np.abs(f1)) < 1e-9
    assert rel(h2["mean"][..., 0], h1[:, 0]) < 1e-9 and rel(h2["nu"] / 2, h1[:, 2]) < 1e-9 and rel(1 / (2 * h2["V"][..., 0, 0]), h1[:, 3]) < 1e-8


def test_error_paths():
    I = np.eye(2)
    K = 2
    args = (np.zeros((K, 2)), np.tile(I, (K, 1, 1)), np.full(K, 3.0), np.tile(I, (K, 1, 1)), np.ones(K))
    with pytest.raises(rxhip.RxHipError) as ei:  # Wishart degrees of freedom must exceed d − 1
        rxhip.MvGMMEngine(10, args[0], args[1], np.full(K, 0.5), args[3], args[4], *args)
    assert ei.value.status == 3
    bad = np.tile(np.array([[1.0, 2.0], [2.0, 1.0]]), (K, 1, 1))
    with pytest.raises(rxhip.RxHipError) as ei:  # indefinite prior covariance
        rxhip.MvGMMEngine(10, args[0], bad, args[2], args[3], args[4], *args)
    assert ei.value.status == 3
    with pytest.raises(rxhip.RxHipError) as ei:  # d = 3 with more than 8 components has no device schedule
        rxhip.MvGMMEngine(10, np.zeros((9, 3)), np.tile(np.eye(3), (9, 1, 1)), np.full(9, 4.0), np.tile(np.eye(3), (9, 1, 1)), np.ones(9),
                          np.zeros((9, 3)), np.tile(np.eye(3), (9, 1, 1)), np.full(9, 4.0), np.tile(np.eye(3), (9, 1, 1)), np.ones(9))
    assert ei.value.status == 2
