"""Multivariate iid Gaussian with unknown mean and precision — `mv_iid_wishart` of test/models/iid/mv_iid_precision_tests.jl:11-15:
m ~ MvNormal(μ, Λ); P ~ Wishart(ν, S); y[i] ~ MvNormal(μ = m, Λ = P), q(m, P) = q(m)q(P).  It is the K = 1 form of the multivariate
mixture engine (MvNormalMeanPrecision × Wishart rules, SURVEY §8 a9 in d dimensions): lowering (CPU), device against the oracle
per iteration, and the reference test's own assertions on data drawn as the test draws it (GPU)."""
import numpy as np
import pytest

import rxhip  # noqa: F401
from rxhip import _lib, graph


def _graph(N, d, init):
    return graph.mv_iid_graph(N, np.zeros(d), 100.0 * np.eye(d), d + 1.0, np.eye(d), init=init)


def test_iid_wishart_graph_lowers_to_the_one_component_mixture():
    d, N = 2, 11
    init = dict(m=(np.zeros(d), 1e12 * np.eye(d)), w=(float(d), 1e12 * np.eye(d)))   # vague(MvNormalMeanPrecision, d), vague(Wishart, d)
    gb, ys = _graph(N, d, init)
    low = graph.lower_mvgmm(gb.tables(permute=np.random.default_rng(0).permutation(len(gb.ftype)))[0])
    assert low["N"] == N and low["K"] == 1 and low["d"] == d and sorted(low["data_var"]) == sorted(ys)
    assert np.allclose(low["S0"].reshape(d, d), 0.01 * np.eye(d))          # Λ = 100·I arrives as its covariance
    assert low["nu0"][0] == d + 1 and np.array_equal(low["V0"].reshape(d, d), np.eye(d)) and low["alpha0"][0] == 1.0
    # a known mean (constant μ on the observation nodes) is another rule family: rejected, not mis-lowered
    gb = graph.GraphBuilder()
    P = gb.randomvar(d)
    gb.node(_lib.NODE_WISHART, P, gb.constvar(3.0), gb.constvar(np.eye(d)))
    gb.node(_lib.NODE_MVNORMAL_MEAN_PRECISION, gb.datavar(d), gb.constvar(np.zeros(d)), P)
    with pytest.raises(rxhip.RxHipError) as ei:
        graph.lower_mvgmm(gb.tables()[0])
    assert ei.value.status == _lib.ERR_UNSUPPORTED


@pytest.mark.gpu
def test_device_matches_the_oracle_and_the_reference_tests_assertions():
    import rxoracle
    rng = np.random.default_rng(123)
    n, d, iters = 1500, 2, 10
    m = rng.random(d)
    Lm = rng.standard_normal((d, d))
    C = Lm @ Lm.T
    P = np.linalg.inv(C)
    y = rng.multivariate_normal(m, C, size=n)
    init = dict(m=(np.zeros(d), 1e12 * np.eye(d)), w=(float(d), 1e12 * np.eye(d)))
    gb, ys = _graph(n, d, init)
    eng = graph.create_vmp_engine_from_graph(gb.tables()[0])
    eng.set_data(y)
    eng.run(iters, True)
    hist = eng.history()
    fe = eng.free_energy()
    eng.close()
    qm, qnu, qW = hist["mean"][-1, 0], hist["nu"][-1, 0], hist["V"][-1, 0]
    # the assertions of mv_iid_precision_tests.jl:64-66 in a form that does not depend on the reference's StableRNG draw: the mean
    # is the conjugate combination of the prior N(0, (100 I)⁻¹) with n observations of precision E[P]; E[P] is the inverse sample
    # covariance up to O(1/n); the free energy decreases at every iteration
    EP = qnu * qW
    ybar = y.mean(axis=0)
    assert np.allclose(qm, np.linalg.solve(100.0 * np.eye(d) + n * EP, n * EP @ ybar), atol=1e-3)
    assert np.allclose(EP, np.linalg.inv(np.cov(y.T, bias=True) + np.outer(ybar - qm, ybar - qm)), rtol=0.02)
    assert np.allclose(qm, m, atol=0.15) and np.allclose(EP, P, rtol=0.15, atol=0.1)
    assert np.all(np.diff(fe)[np.abs(np.diff(fe)) > 1e-10] < 0)
    # the oracle's mean-field schedule, iteration by iteration
    init_o = rxoracle.mvgmm_pack(np.zeros((1, d)), 1e12 * np.eye(d)[None], np.array([float(d)]), 1e12 * np.eye(d)[None], np.array([1.0]))
    oh, ofe, _ = rxoracle.mvgmm_vmp(y, np.zeros((1, d)), 0.01 * np.eye(d)[None], np.array([d + 1.0]), np.eye(d)[None], np.array([1.0]),
                                    init_o, iters)
    assert np.allclose(fe, ofe, rtol=1e-8)
    assert np.allclose(hist["raw"], oh, rtol=1e-6, atol=1e-9)
