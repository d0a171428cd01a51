"""The factorisation of q at the boundary (the descriptor carries "factorisation clusters"): the reference hands every node its
VariationalConstraintsFactorizationIndicesKey (src/model/plugins/reactivemp_inference.jl:499-506) and the rules dispatch on it.  rxhip_graph_desc.factor_cluster
carries the same table; every lowering pass and the node-array executor's compiler hold it against the ONE factorisation per node type their schedule implements.
A model whose @constraints ask for another family is refused with the node named (-> UnsupportedGraph -> stock plugin), never answered with the structured
posterior.  Host only: no device involved."""
import gzip
import json
import os

import numpy as np
import pytest

import rxhip
from rxhip import _lib, graph
from rxhip.graph import GraphBuilder
from rxhip.tree import plan

import tree_graphs as tg

DUMPS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "graph_dumps")


def chain(T=6, d=2):
    rng = np.random.default_rng(3)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    return graph.lgssm_graph(T, 0.9 * q, rng.standard_normal((d, d)), 0.1 * np.eye(d), np.eye(d), np.zeros(d), 4.0 * np.eye(d))[0]


def refused(fn, *needles):
    with pytest.raises(rxhip.RxHipError) as ei:
        fn()
    assert ei.value.status == _lib.ERR_UNSUPPORTED, ei.value
    for n in needles:
        assert n in str(ei.value), (n, str(ei.value))
    return str(ei.value)


def test_default_constraints_change_nothing():
    """GraphPPL's default (all random interfaces of a node joint) spelled out in the table: every family lowers exactly as with a NULL table"""
    gb = chain()
    a = graph.lower_lgssm(gb.tables()[0])
    b = graph.lower_lgssm(gb.bethe().tables()[0])
    assert a["T"] == b["T"] and np.array_equal(a["A"], b["A"]) and np.array_equal(a["state_var"], b["state_var"])
    gb, ys, _ = tg.two_branch_chain(T=5)
    p0 = plan(gb)
    p1 = plan(gb.bethe())
    assert p0["n_ops"] == p1["n_ops"] and p0["rule_calls"] == p1["rule_calls"]
    gb, ys, _ = tg.chain_state_noise_precision(T=5, d=2, dy=2, also_obs_noise=True)
    assert plan(gb.gaussian_joint())["n_precision_vars"] == plan(gb)["n_precision_vars"] == 2   # `q(x, W) = q(x)q(W)`: the precision interface a factor of its own
    refused(lambda: plan(gb.bethe()), "MvNormalMeanPrecision", "precision")   # GraphPPL's default on such a model is the joint q(out, μ, W): no rule, here as in the reference


def test_the_reference_models_dumps_carry_their_constraints_and_lower():
    for name, lower in (("mlgssm", graph.lower_lgssm), ("ulgssm", graph.lower_lgssm), ("gmm_univariate", graph.lower_gmm),
                        ("gmm_multivariate", graph.lower_mvgmm), ("hgf_step", graph.lower_hgf)):
        d = json.load(gzip.open(os.path.join(DUMPS, name + ".json.gz"), "rt"))
        assert all(len(f["clusters"]) == len(f["interfaces"]) for f in d["factors"]), name
        gb = GraphBuilder.from_dump(d)
        assert len(gb.fcluster) == len(gb.ftype)
        g, _ = gb.tables()
        assert bool(g.factor_cluster)
        lower(g)
        assert gb.to_dump(n_replicas=d["n_replicas"], n_observations=d["n_observations"]) == d   # the table survives the exchange format
    hgf = json.load(gzip.open(os.path.join(DUMPS, "hgf_step.json.gz"), "rt"))
    gcv = next(f for f in hgf["factors"] if f["type"] == "GCV")
    assert gcv["clusters"] == [0, 0, 1, 2, 3]            # hgf_tests.jl:33-35  q(xt, zt, xt_min) = q(xt, xt_min) q(zt)
    gmm = json.load(gzip.open(os.path.join(DUMPS, "gmm_univariate.json.gz"), "rt"))
    mix = next(f for f in gmm["factors"] if f["type"] == "NormalMixture")
    assert mix["clusters"] == list(range(6))             # MeanField(): gmm_univariate_tests.jl:63-72


def test_a_mean_field_chain_is_not_answered_with_the_structured_posterior():
    """`constraints = MeanField()` on the benchmark chain: q(x[t]) q(x[t-1]) around every transition.  The state-space lowering (parallel-in-time BP: the
    structured family) refuses it with the node named; rxhip_create then asks the node-array executor, whose compiler either runs the mean-field rules or
    refuses the same way."""
    gb = chain().mean_field()
    g, _ = gb.tables()
    msg = refused(lambda: graph.lower_lgssm(g), "MvNormalMeanCovariance", "q(out) q(μ)", "factor_cluster")
    assert "factor " in msg
    # the deterministic nodes of a MeanField() model keep their joint: only the Gaussian transitions are named
    first = int(msg.split("factor ")[1].split(" ")[0])
    assert gb.ftype[first] == _lib.NODE_MVNORMAL_MEAN_COV and gb.kind[gb.fiface[first][0]] == gb.kind[gb.fiface[first][1]] == _lib.VARKIND_RANDOM
    # the executor's compiler: the mean-field rules need the @initialization marginals of the variables they read (the reference refuses too) …
    with pytest.raises(rxhip.RxHipError) as ei:
        plan(gb)
    assert ei.value.status == _lib.ERR_BADARG and "@initialization" in str(ei.value)
    # … and with them the graph compiles: every transition node is two leaf rules (6T − 3 rule calls still: one per message), no message crosses it
    gb, ys, named = tg.mean_field_chain(T=6, d=2, dy=2)
    p_mf, p_bp = plan(gb), plan(tg.mean_field_chain(T=6, d=2, dy=2)[0].bethe())
    assert p_mf["rule_calls"] == p_bp["rule_calls"] == 6 * 6 - 3 and p_mf["n_levels"] < p_bp["n_levels"]
    # above 8 dimensions the same ops run on the LDS-staged kernels
    gb, _, _ = tg.mean_field_chain(T=3, d=9, dy=9)
    assert plan(gb)["dmax"] == 9 and plan(gb)["rule_calls"] == 6 * 3 - 3


def test_every_family_checks_its_own_factorisation():
    # a deterministic node factorised
    gb = chain().bethe()
    f = gb.ftype.index(_lib.NODE_MULTIPLY)
    gb.set_clusters(f, (0, 1, 2))
    refused(lambda: graph.lower_lgssm(gb.tables()[0]), "(*)", "deterministic")
    refused(lambda: plan(gb), "(*)", "deterministic")
    # a mixture asked for a structured factor (the reference itself throws there: gmm_univariate_tests.jl:117-124)
    gb = GraphBuilder.from_dump(os.path.join(DUMPS, "gmm_univariate.json.gz"))
    f = gb.ftype.index(_lib.NODE_NORMAL_MIXTURE)
    gb.set_clusters(f, (0, 1, 2, 2, 3, 4))   # q(m[1], m[2]) joint
    refused(lambda: graph.lower_gmm(gb.tables()[0]), "NormalMixture", "mean-field")
    gb = GraphBuilder.from_dump(os.path.join(DUMPS, "gmm_multivariate.json.gz"))
    f = gb.ftype.index(_lib.NODE_NORMAL_MIXTURE)
    cl = list(gb.clusters_of(f))
    cl[2] = cl[5]                            # q(m[1], w[1]) joint
    gb.set_clusters(f, cl)
    refused(lambda: graph.lower_mvgmm(gb.tables()[0]), "NormalMixture")
    # the HGF step under full mean-field (q(xt) q(xt_min)) or with z in the joint
    for cl, needle in (((0, 1, 2, 3, 4), "q(y) q(x)"), ((0, 0, 0, 1, 2), "volatility")):
        gb = GraphBuilder.from_dump(os.path.join(DUMPS, "hgf_step.json.gz"))
        gb.set_clusters(gb.ftype.index(_lib.NODE_GCV), cl)
        refused(lambda: graph.lower_hgf(gb.tables()[0]), "GCV", needle)
    # a Gaussian node with a random precision under a joint q(μ, W)
    gb, ys, nm = tg.chain_state_noise_precision(T=4, d=2, dy=2, also_obs_noise=True)
    gb.gaussian_joint()
    f = next(i for i, t in enumerate(gb.ftype) if t == _lib.NODE_MVNORMAL_MEAN_PRECISION and gb.kind[gb.fiface[i][2]] == _lib.VARKIND_RANDOM
             and gb.kind[gb.fiface[i][1]] == _lib.VARKIND_RANDOM)
    cl = list(gb.clusters_of(f))
    cl[2] = cl[1]
    gb.set_clusters(f, cl)
    refused(lambda: plan(gb), "MvNormalMeanPrecision", "precision")


def test_cluster_ids_of_clamped_interfaces_are_ignored():
    """GraphPPL gives every data / constant interface a cluster of its own; any numbering of those must do"""
    gb = chain().bethe()
    for f in range(len(gb.ftype)):
        ids = list(gb.clusters_of(f))
        for k, v in enumerate(gb.fiface[f]):
            if gb.kind[v] != _lib.VARKIND_RANDOM:
                ids[k] = 7 + k
        gb.set_clusters(f, ids)
    assert graph.lower_lgssm(gb.tables()[0])["T"] == 6
