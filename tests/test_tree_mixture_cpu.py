"""NormalMixture layers in the node-array executor, host side: oracle/tree_oracle.py's extension pinned to oracle/rxoracle.c's mixture restatement (itself pinned
to the reference's golden free energy of test/models/mixtures/gmm_multivariate_tests.jl, tests/test_golden_reference.py), the compiler's schedule and counts
(`rxhip_tree_plan`, no device), and what is refused by name."""
import numpy as np
import pytest

import rxhip
import rxoracle
import tree_graphs as tg
import tree_oracle
from rxhip import _lib, graph
from rxhip.tree import plan


def _mixture(K, d, N, seed):
    rng = np.random.default_rng(seed)
    cent = 5.0 * rng.standard_normal((K, d))
    y = np.concatenate([cent[k] + rng.standard_normal((N // K + 1, d)) for k in range(K)])[:N]
    rng.shuffle(y)
    mu0, S0 = cent + rng.standard_normal((K, d)), np.array([1e2 * np.eye(d)] * K)
    nu0, V0, al0 = np.array([d + 1.0] * K), np.array([0.1 * np.eye(d)] * K), 1.0 + rng.random(K)
    return y, mu0, S0, nu0, V0, al0


@pytest.mark.parametrize("K,d,N", [(3, 2, 40), (2, 1, 25), (4, 3, 30), (1, 2, 12)])
def test_tree_oracle_equals_the_pinned_mixture_restatement(K, d, N):
    y, mu0, S0, nu0, V0, al0 = _mixture(K, d, N, 10 * K + d)
    iters = 6
    h, fe, resp = rxoracle.mvgmm_vmp(y, mu0, S0, nu0, V0, al0, rxoracle.mvgmm_pack(mu0, S0, nu0, V0, np.ones(K)), iters, want_resp=True)
    gb, ys = graph.mv_mixture_graph(N, mu0, S0, nu0, V0, al0, init=dict(m=(mu0, S0), w=(nu0, V0), s=np.ones(K)))
    ref = tree_oracle.infer(gb.to_dump(), {ys[i]: y[i] for i in range(N)}, iterations=iters)
    assert np.max(np.abs((np.asarray(ref["fe"]) - fe) / fe)) < 1e-12
    o = rxoracle.mvgmm_unpack(h, d)
    g = tree_oracle.TreeGraph(gb.to_dump())
    mx = g.mixtures[0]
    for k in range(K):
        assert np.allclose(ref["mean"][mx["m"][k]], o["mean"][-1][k], rtol=1e-11, atol=1e-12)
        assert np.allclose(ref["cov"][mx["m"][k]], o["cov"][-1][k], rtol=1e-11, atol=1e-14)
        assert ref["q_prec"][mx["p"][k]][0] == pytest.approx(o["nu"][-1][k], rel=1e-12)
        assert np.allclose(ref["q_prec"][mx["p"][k]][1], o["V"][-1][k], rtol=1e-10)
    assert np.allclose(ref["q_dir"][g.cat[mx["z"]]], o["alpha"][-1], rtol=1e-12)
    assert np.allclose(np.stack([ref["q_cat"][m["z"]] for m in g.mixtures]), resp, atol=1e-12)
    assert np.all(np.diff(ref["fe"]) < 1e-9 * abs(ref["fe"][-1]))


@pytest.mark.parametrize("kw", [dict(N=12, K=2, d=2), dict(N=8, K=2, d=2, latent_out=True), dict(N=10, K=3, d=3, const_switch=True),
                                dict(N=7, K=2, d=2, shared_parent=False, const_precision=True)])
def test_free_energy_decreases_and_the_plan_counts_what_the_oracle_counts(kw):
    gb, ys, named = tg.mixture_on_tree(**kw)
    data = tg.random_data(gb, ys, 1, 0)
    ref = tree_oracle.infer(gb.to_dump(), tg.data_dict(gb, ys, data[0]), iterations=5)
    assert np.all(np.diff(ref["fe"]) < 1e-9 * abs(ref["fe"][-1]))   # coordinate ascent on the bound
    p = plan(gb)
    assert p["rule_calls"] == ref["counters"]["rule_calls"] and p["marginals"] == ref["counters"]["marginals"]
    assert p["n_precision_vars"] == len(named["W"])


def _refused(gb, status, *needles):
    with pytest.raises(rxhip.RxHipError) as ei:
        plan(gb)
    assert ei.value.status == status, ei.value
    for n in needles:
        assert n in str(ei.value), (n, str(ei.value))


def test_what_the_compiler_refuses():
    # a structured factor at the mixture node (the reference throws there too: gmm_univariate_tests.jl:117-124)
    gb, ys, named = tg.mixture_on_tree(N=3, K=2, d=2)
    f = gb.ftype.index(_lib.NODE_NORMAL_MIXTURE)
    cl = list(gb.clusters_of(f))
    cl[2] = cl[3]
    gb.set_clusters(f, cl)
    _refused(gb, _lib.ERR_UNSUPPORTED, "NormalMixture")
    # dimensions above 8: the mixture ops live in the lane-per-item kernels
    gb, _, _ = tg.mixture_on_tree(N=2, K=2, d=9)
    _refused(gb, _lib.ERR_UNSUPPORTED, "NormalMixture", "8")
    # a switch without a Categorical prior; a Bernoulli switch on a one-component node
    gb = graph.GraphBuilder()
    m, w, z, y = gb.randomvar(1), gb.randomvar(1), gb.randomvar(1), gb.datavar(1)
    gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, m, gb.constvar(0.0), gb.constvar(1.0))
    gb.node(_lib.NODE_GAMMA_SHAPE_RATE, w, gb.constvar(1.0), gb.constvar(1.0))
    gb.node(_lib.NODE_NORMAL_MIXTURE, y, z, m, w)
    _refused(gb, _lib.ERR_UNSUPPORTED, "switch")
    s = gb.randomvar(1)
    gb.node(_lib.NODE_BETA, s, gb.constvar(1.0), gb.constvar(1.0))
    gb.node(_lib.NODE_BERNOULLI, z, s)
    _refused(gb, _lib.ERR_BADARG, "components")   # a Bernoulli switch is the two-component spelling: this node has one
    # `missing` observations under a mixture node
    gb, ys, named = tg.mixture_on_tree(N=3, K=2, d=2)
    with pytest.raises(rxhip.RxHipError) as ei:
        plan(gb, allow_missing=True)
    assert ei.value.status == _lib.ERR_UNSUPPORTED and "missing" in str(ei.value)
    # components of another dimension than `out`
    gb, ys, named = tg.mixture_on_tree(N=2, K=2, d=2)
    bad = graph.GraphBuilder.from_dump(gb.to_dump())
    f = bad.ftype.index(_lib.NODE_NORMAL_MIXTURE)
    y3 = bad.datavar(3)
    bad.fiface[f] = (y3,) + tuple(bad.fiface[f][1:])
    _refused(bad, _lib.ERR_BADARG, "dimension")


def _univariate_reference_model(n=60, seed=12345):
    """test/models/mixtures/gmm_univariate_tests.jl:7-20 with its priors and @initialization (Beta / Bernoulli spelling of the two-component switch)"""
    rng = np.random.default_rng(seed)
    z = rng.choice(2, size=n, p=[1 / 3, 2 / 3])
    y = np.array([-10.0, 10.0])[z] + rng.standard_normal(n) / np.sqrt(np.array([3.777, 0.333])[z])
    priors = ([-2.0, 2.0], [1e3, 1e3], [0.01, 0.01], [0.01, 0.01], [1.0, 1.0])
    init = ([-2.0, 2.0], [1e3, 1e3], [1.0, 1.0], [1e-12, 1e-12], [1.0, 1.0])
    gb, ys = graph.mixture_graph(n, *priors, init=dict(m=(init[0], init[1]), p=(init[2], init[3]), s=init[4]), bernoulli=True)
    return y, priors, init, gb, ys


def test_univariate_reference_model_in_the_bernoulli_spelling():
    y, priors, init, gb, ys = _univariate_reference_model()
    iters = 8
    hist, fe, resp, _ = rxoracle.gmm_vmp(y, *priors, *init, iters, want_resp=True)
    ref = tree_oracle.infer(gb.to_dump(), {ys[i]: y[i:i + 1] for i in range(y.size)}, iterations=iters)
    assert np.max(np.abs((np.asarray(ref["fe"]) - fe) / fe)) < 1e-11
    g = tree_oracle.TreeGraph(gb.to_dump())
    assert np.allclose(np.stack([ref["q_cat"][m["z"]] for m in g.mixtures]), resp, atol=1e-11)
    assert plan(gb)["rule_calls"] == ref["counters"]["rule_calls"]
