"""NormalMixture / Categorical / Dirichlet as ops of the node-array executor (VERDICT r5 "Next 8": "so a mixture layer can hang off a Gaussian tree"):
the executor on the reference's multivariate mixture model against the specialised engine and the pinned oracle, and on mixture layers whose means
share a Gaussian parent, whose `out` is a latent Gaussian variable, whose switch probabilities or precisions are constants — against oracle/tree_oracle.py
(pinned to oracle/rxoracle.c's mixture restatement in tests/test_tree_oracle.py), every schedule, several replicas, every iteration.

Tolerances: the contract is 1e-6 (posteriors) / 1e-8 (free energy); asserted at 1e-9."""
import numpy as np
import pytest

import tree_graphs as tg
from test_tree_engine_gpu import _check, _run

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("kw", [dict(N=12, K=2, d=2), dict(N=9, K=3, d=1), dict(N=8, K=2, d=2, latent_out=True), dict(N=10, K=3, d=3, const_switch=True),
                                dict(N=7, K=2, d=2, shared_parent=False, const_precision=True), dict(N=40, K=4, d=2, shared_parent=False),
                                dict(N=6, K=2, d=4, latent_out=True, const_switch=True), dict(N=5, K=2, d=6)])
def test_mixture_layers_against_the_oracle(kw, mode, monkeypatch):
    gb, ys, named = tg.mixture_on_tree(**kw)
    R = 3
    for its in (1, 3):
        eng, data = _run(gb, ys, R, iterations=its, mode=mode, monkeypatch=monkeypatch, seed=its)
        assert eng.info["kernels"] == 0
        ref = _check(gb, ys, eng, data, iterations=its, replicas=(0, R - 1), prec_vars=named["W"], tol=1e-9, tol_fe=1e-9, tol_nu=1e-10)
        assert eng.counters()["rule_calls"] == ref["counters"]["rule_calls"] * R * its
        eng.close()


def test_reference_mixture_model_equals_the_specialised_engine():
    """test/models/mixtures/gmm_multivariate_tests.jl:6-32 through rxhip_tree_create (a node per data point) and through the mixture engine (sufficient
    statistics over the data set): the same VMP, iteration by iteration"""
    import rxhip
    from rxhip import graph
    from rxhip.tree import TreeEngine
    rng = np.random.default_rng(5)
    K, d, N, iters = 3, 2, 120, 6
    cent = np.array([[6.0, 0.0], [-4.0, 5.0], [0.0, -6.0]])
    y = np.concatenate([cent[k] + rng.standard_normal((N // K, d)) for k in range(K)])
    rng.shuffle(y)
    mu0, S0 = cent + rng.standard_normal((K, d)), np.array([1e2 * np.eye(d)] * K)
    nu0, V0, al0 = np.array([3.0] * K), np.array([0.1 * np.eye(d)] * K), np.ones(K)
    init = dict(m=(mu0, S0), w=(nu0, V0), s=np.ones(K))
    gb, ys = graph.mv_mixture_graph(N, mu0, S0, nu0, V0, al0, init=init)
    with rxhip.MvGMMEngine(N, mu0, S0, nu0, V0, al0, mu0, S0, nu0, V0, np.ones(K)) as ref:
        ref.set_data(y)
        ref.run(iters, True)
        h, fe = ref.history(), ref.free_energy()
    with TreeEngine(gb, n_replicas=2) as eng:
        eng.set_data(ys, np.stack([y.ravel(), y.ravel()]))
        for it in range(iters):
            eng.run(it + 1, True)
            assert eng.free_energy_per_replica()[1] == pytest.approx(fe[it], rel=1e-10)
        import tree_oracle
        g = tree_oracle.TreeGraph(gb.to_dump())
        means = [mx for mx in g.mixtures][0]["m"]
        post = eng.marginals(means)
        for k, v in enumerate(means):
            assert np.allclose(post[v][0][0], h["mean"][-1][k], rtol=1e-9, atol=1e-10)
            assert np.allclose(post[v][1][0], h["cov"][-1][k], rtol=1e-9, atol=1e-12)
        for k, wv in enumerate([mx for mx in g.mixtures][0]["p"]):
            nu, V = eng.precision(wv)
            assert nu[0] == pytest.approx(h["nu"][-1][k], rel=1e-12) and np.allclose(V[0], h["V"][-1][k], rtol=1e-9)


def test_n_runs_of_one_iteration_equal_one_run_of_n_with_a_mixture():
    from rxhip.tree import TreeEngine
    gb, ys, named = tg.mixture_on_tree(N=9, K=2, d=2, latent_out=True)
    data = tg.random_data(gb, ys, 4, 3)
    with TreeEngine(gb, n_replicas=4) as a, TreeEngine(gb, n_replicas=4) as b:
        a.set_data(ys, data); b.set_data(ys, data)
        a.run(4, True)
        b.continue_runs(True)
        for _ in range(4):
            b.run(1, True)
        assert np.array_equal(a.free_energy_per_replica(), b.free_energy_per_replica())
        pa, pb = a.marginals(named["m"]), b.marginals(named["m"])
        for v in named["m"]:
            assert np.array_equal(pa[v][0], pb[v][0]) and np.array_equal(pa[v][1], pb[v][1])


def test_responsibilities_and_concentrations_are_readable():
    import tree_oracle
    from rxhip.tree import TreeEngine
    gb, ys, named = tg.mixture_on_tree(N=14, K=3, d=2)
    R = 2
    data = tg.random_data(gb, ys, R, 5)
    with TreeEngine(gb, n_replicas=R) as eng:
        eng.set_data(ys, data)
        eng.run(3, True)
        ref = tree_oracle.infer(gb.to_dump(), tg.data_dict(gb, ys, data[1]), iterations=3)
        for z in named["z"]:
            pz = eng.discrete(z)
            assert pz.shape == (R, 3) and np.allclose(pz.sum(axis=1), 1.0, atol=1e-14)
            assert np.allclose(pz[1], ref["q_cat"][z], atol=1e-11)
        assert np.allclose(eng.discrete(named["s"])[1], ref["q_dir"][named["s"]], rtol=1e-12)
        import rxhip
        with pytest.raises(rxhip.RxHipError):
            eng.discrete(named["m"][0])


def test_univariate_reference_model_equals_the_specialised_engine():
    """test/models/mixtures/gmm_univariate_tests.jl:7-20 (Beta / Bernoulli switch, Gamma precisions) through the executor and through GMMEngine"""
    import rxhip
    import tree_oracle
    from rxhip.tree import TreeEngine
    from test_tree_mixture_cpu import _univariate_reference_model
    y, priors, init, gb, ys = _univariate_reference_model(n=150)
    iters = 10
    with rxhip.GMMEngine(y.size, *priors, *init) as ref:
        ref.set_data(y)
        ref.run(iters, True)
        fe = ref.free_energy()
    with TreeEngine(gb, n_replicas=1) as eng:
        eng.set_data(ys, y[None, :])
        eng.run(iters, True)
        assert np.allclose(eng.free_energy(), fe, rtol=1e-10)
        assert np.all(np.diff(eng.free_energy()) < 1e-6 * abs(fe[-1]))   # gmm_univariate_tests.jl: the free energy does not increase
