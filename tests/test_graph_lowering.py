"""CPU tests of the graph → schedule lowering (host logic, no GPU): the SoA dump of the graph
RxInfer builds for the LGSSM model is recognised whatever the node order, its constants are recovered, and
graphs outside the supported family are rejected with RXHIP_ERR_UNSUPPORTED (→ fall back to the stock plugin)."""
import numpy as np
import pytest

import rxhip
from rxhip import _lib, graph, workloads


def test_benchmark_model_graph_is_recognised():
    mdl = workloads.c1_model()
    T = 50
    gb, xs, ys = graph.lgssm_graph(T, mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"])
    # Appendix C: per interior step 4 factor nodes, 3 random + 1 data + 4 constant variables
    assert len(gb.ftype) == 1 + 2 * T + 2 * (T - 1)
    g, keep = gb.tables(n_replicas=3)
    low = graph.lower_lgssm(g)
    assert (low["d"], low["dy"], low["T"], low["prior_through_transition"]) == (4, 4, T, False)
    for k in ("A", "B", "P", "Q", "m0", "V0"):
        assert np.array_equal(low[k], mdl[k])
    assert list(low["state_var"]) == xs and list(low["data_var"]) == ys  # time order recovered


def test_node_order_is_irrelevant_and_prior_variant():
    mdl = workloads.random_model(3, 2, seed=4)
    gb, xs, ys = graph.lgssm_graph(17, mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], prior_through_transition=True)
    perm = np.random.default_rng(0).permutation(len(gb.ftype))
    g, keep = gb.tables(permute=perm)
    low = graph.lower_lgssm(g)
    assert low["prior_through_transition"] and low["T"] == 17 and low["dy"] == 2
    assert np.array_equal(low["A"], mdl["A"]) and list(low["data_var"]) == ys


def test_single_observation_graph():
    I = np.eye(1)
    gb, xs, ys = graph.lgssm_graph(1, I, I, I, I, [3.0], I)
    low = graph.lower_lgssm(gb.tables()[0])
    assert low["T"] == 1 and low["m0"][0] == 3.0


def test_unsupported_graphs_are_rejected():
    mdl = workloads.c1_model()
    # a branching graph: two transitions out of one state
    gb, xs, ys = graph.lgssm_graph(3, mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"])
    a = gb.randomvar(4); gb.multiply(a, gb.constvar(mdl["A"]), xs[0]); xn = gb.randomvar(4); gb.mvnormal_mean_cov(xn, a, gb.constvar(mdl["P"]))
    with pytest.raises(rxhip.RxHipError) as ei:
        graph.lower_lgssm(gb.tables()[0])
    assert ei.value.status == _lib.ERR_UNSUPPORTED and "chain" in str(ei.value)
    # an unknown node type
    gb, xs, ys = graph.lgssm_graph(3, mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"])
    gb.ftype[2] = 99
    with pytest.raises(rxhip.RxHipError) as ei:
        graph.lower_lgssm(gb.tables()[0])
    assert ei.value.status == _lib.ERR_UNSUPPORTED
    # malformed tables
    g = _lib.GraphDesc()
    with pytest.raises(rxhip.RxHipError) as ei:
        graph.lower_lgssm(g)
    assert ei.value.status == _lib.ERR_BADARG


@pytest.mark.gpu
def test_engine_from_graph_matches_structured_descriptor():
    """rxhip_create(graph) == rxhip_lgssm_create(structured) on the device (T = 300, 5 replicas)."""
    mdl = workloads.c1_model()
    T, C = 300, 5
    y = workloads.generate_batch(mdl, T, C)
    gb, xs, ys = graph.lgssm_graph(T, mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"])
    g, keep = gb.tables(n_replicas=C)
    eng = graph.create_engine_from_graph(g)
    eng.set_data(y); eng.run(1, True)
    m1, V1 = eng.marginals(); f1 = eng.free_energy_per_chain(); eng.close()
    with rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=C) as e2:
        e2.set_data(y); e2.run(1, True)
        m2, V2 = e2.marginals(); f2 = e2.free_energy_per_chain()
    assert np.array_equal(m1, m2) and np.array_equal(V1, V2) and np.array_equal(f1, f2)


# ---- mean-field families (a9 / a10 / a11) ---------------------------------------------------------------------
_PRI = dict(mean=[-2.0, 2.0, 7.0], var=[1e3, 1e3, 5e2], shape=[0.01, 0.02, 0.03], rate=[0.01, 0.01, 0.04], alpha=[1.0, 2.0, 3.0])
_INIT = dict(m=([-2.0, 2.0, 6.0], [1e3, 1e2, 1e1]), p=([1.0, 1.5, 2.0], [1e-12, 1.0, 2.0]), s=[1.0, 1.0, 4.0])


def test_mixture_graph_is_recognised_whatever_the_node_order():
    N = 40
    gb, ys = graph.mixture_graph(N, _PRI["mean"], _PRI["var"], _PRI["shape"], _PRI["rate"], _PRI["alpha"], init=_INIT)
    assert len(gb.ftype) == 1 + 2 * 3 + 2 * N
    for perm in (None, np.random.default_rng(1).permutation(len(gb.ftype))):
        low = graph.lower_gmm(gb.tables(permute=perm)[0])
        assert (low["N"], low["K"]) == (N, 3)
        assert np.array_equal(low["mu0"], _PRI["mean"]) and np.array_equal(low["v0"], _PRI["var"])
        assert np.array_equal(low["a0"], _PRI["shape"]) and np.array_equal(low["b0"], _PRI["rate"]) and np.array_equal(low["alpha0"], _PRI["alpha"])
        assert np.array_equal(low["init_m_mean"], _INIT["m"][0]) and np.array_equal(low["init_m_var"], _INIT["m"][1])
        assert np.array_equal(low["init_p_shape"], _INIT["p"][0]) and np.array_equal(low["init_p_rate"], _INIT["p"][1])
        assert np.array_equal(low["init_s_alpha"], _INIT["s"])
        if perm is None:
            assert list(low["data_var"]) == ys


def test_reference_spelling_beta_bernoulli_and_iid_gaussian():
    """test/models/mixtures/gmm_univariate_tests.jl:7-26 (Beta / Bernoulli, K = 2) and models_tests.jl:114-128 (K = 1)."""
    init = dict(m=([-2.0, 2.0], [1e3, 1e3]), p=([1.0, 1.0], [1e-12, 1e-12]), s=[1.0, 1.0])
    gb, _ = graph.mixture_graph(15, [-2.0, 2.0], [1e3, 1e3], [0.01, 0.01], [0.01, 0.01], [1.0, 1.0], init=init, bernoulli=True)
    low = graph.lower_gmm(gb.tables()[0])
    assert (low["N"], low["K"]) == (15, 2) and np.array_equal(low["alpha0"], [1.0, 1.0]) and np.array_equal(low["init_p_rate"], [1e-12, 1e-12])
    gb, ys = graph.iid_normal_graph(9, 4.0, 8.0, 4.0, 0.125, init=dict(m=(0.0, 1.0), p=(1.0, 1.0)))
    low = graph.lower_gmm(gb.tables()[0])
    assert (low["N"], low["K"]) == (9, 1) and low["mu0"][0] == 4.0 and low["b0"][0] == 0.125 and low["alpha0"][0] == 1.0
    # the reference's own spelling of that prior: `τ ~ Gamma(shape = 4, scale = 8)` (models_tests.jl:121-127) is the same model
    gb, _ = graph.iid_normal_graph(9, 4.0, 8.0, 4.0, scale=8.0, init=dict(m=(0.0, 1.0), p=(1.0, 1.0)))
    low2 = graph.lower_gmm(gb.tables()[0])
    assert all(np.array_equal(low[k], low2[k]) for k in ("mu0", "v0", "a0", "b0", "alpha0", "init_p_shape", "init_p_rate"))
    gb, _ = graph.iid_normal_graph(9, 4.0, 8.0, 4.0, scale=-1.0, init=dict(m=(0.0, 1.0), p=(1.0, 1.0)))
    with pytest.raises(rxhip.RxHipError) as ei:
        graph.lower_gmm(gb.tables()[0])
    assert ei.value.status == _lib.ERR_BADARG and "scale" in str(ei.value)


def test_mixture_graphs_outside_the_family_are_rejected():
    # no @initialization: mean-field VMP cannot start (cf. test/inference/inference_tests.jl:361-363)
    gb, _ = graph.mixture_graph(5, _PRI["mean"], _PRI["var"], _PRI["shape"], _PRI["rate"], _PRI["alpha"])
    with pytest.raises(rxhip.RxHipError) as ei:
        graph.lower_gmm(gb.tables()[0])
    assert ei.value.status == _lib.ERR_BADARG and "initialization" in str(ei.value)
    # observation nodes that do not share the component variables
    gb, _ = graph.mixture_graph(5, _PRI["mean"], _PRI["var"], _PRI["shape"], _PRI["rate"], _PRI["alpha"], init=_INIT)
    it = list(gb.fiface[-1]); it[2], it[3] = it[3], it[2]; gb.fiface[-1] = tuple(it)
    with pytest.raises(rxhip.RxHipError) as ei:
        graph.lower_gmm(gb.tables()[0])
    assert ei.value.status == _lib.ERR_UNSUPPORTED and "share" in str(ei.value)
    # a stray node
    gb, _ = graph.mixture_graph(5, _PRI["mean"], _PRI["var"], _PRI["shape"], _PRI["rate"], _PRI["alpha"], init=_INIT)
    gb.node(_lib.NODE_GCV, 0, 1, 2, 3, 4)
    with pytest.raises(rxhip.RxHipError) as ei:
        graph.lower_gmm(gb.tables()[0])
    assert ei.value.status == _lib.ERR_UNSUPPORTED
    # an LGSSM graph is not a mixture and vice versa
    mdl = workloads.c1_model()
    gl, _, _ = graph.lgssm_graph(4, mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"])
    with pytest.raises(rxhip.RxHipError):
        graph.lower_gmm(gl.tables()[0])
    gb, _ = graph.mixture_graph(5, _PRI["mean"], _PRI["var"], _PRI["shape"], _PRI["rate"], _PRI["alpha"], init=_INIT)
    with pytest.raises(rxhip.RxHipError):
        graph.lower_lgssm(gb.tables()[0])


def test_hgf_step_graph_is_recognised():
    gb, ids = graph.hgf_step_graph(1.0, 0.0, 0.04, 0.01, q_zt=(0.1, 5.0), q_xt=(-0.2, 3.0), n_gh=21)
    for perm in (None, [4, 2, 0, 3, 1]):
        low = graph.lower_hgf(gb.tables(permute=perm, n_observations=100)[0])
        assert (low["kappa"], low["omega"], low["z_variance"], low["y_variance"]) == (1.0, 0.0, 0.04, 0.01)
        assert (low["z0_mean"], low["z0_var"], low["x0_mean"], low["x0_var"], low["n_gh"]) == (0.1, 5.0, -0.2, 3.0, 21)
        assert (low["zt_var"], low["xt_var"], low["y_var"]) == (ids["zt"], ids["xt"], ids["y"])
    # a constant instead of the @autoupdates data variables on a prior: not the streaming filter graph
    gb, _ = graph.hgf_step_graph(1.0, 0.0, 0.04, 0.01)
    it = list(gb.fiface[0]); it[1] = gb.constvar(0.0); gb.fiface[0] = tuple(it)
    with pytest.raises(rxhip.RxHipError) as ei:
        graph.lower_hgf(gb.tables()[0])
    assert ei.value.status == _lib.ERR_UNSUPPORTED


@pytest.mark.gpu
def test_vmp_engines_from_graphs_match_structured_descriptors():
    """rxhip_create(graph) == rxhip_gmm_create / rxhip_hgf_create (structured) on the device."""
    rng = np.random.default_rng(2)
    y = np.asarray(_PRI["mean"])[rng.integers(0, 3, 3000)] + rng.standard_normal(3000)
    init = dict(m=([-3.0, 1.0, 6.0], [1.0] * 3), p=([1.0] * 3, [1.0] * 3), s=[1.0] * 3)
    gb, _ = graph.mixture_graph(y.size, _PRI["mean"], [1e2] * 3, [0.1] * 3, [0.1] * 3, [1.0] * 3, init=init)
    g, keep = gb.tables()
    e1 = graph.create_vmp_engine_from_graph(g)
    e1.set_data(y); e1.run(5, True)
    h1, f1 = e1.history(), e1.free_energy(); e1.close()
    with rxhip.GMMEngine(y.size, _PRI["mean"], [1e2] * 3, [0.1] * 3, [0.1] * 3, [1.0] * 3, *init["m"], *init["p"], init["s"]) as e2:
        e2.set_data(y); e2.run(5, True)
        assert np.array_equal(h1, e2.history()) and np.array_equal(f1, e2.free_energy())
    T, S = 200, 3
    _, _, ys = workloads.generate_hgf_batch(T, S, seed=8)
    gb, _ = graph.hgf_step_graph(1.0, 0.0, 0.04, 0.01)
    g, keep = gb.tables(n_replicas=S, n_observations=T)
    e1 = graph.create_vmp_engine_from_graph(g)
    e1.set_data(ys); e1.run(10, True)
    z1, f1 = e1.history(), e1.free_energy(); e1.close()
    with rxhip.HGFEngine(T, S, 1.0, 0.0, 0.04, 0.01) as e2:
        e2.set_data(ys); e2.run(10, True)
        assert all(np.array_equal(a, b) for a, b in zip(z1, e2.history())) and np.array_equal(f1, e2.free_energy())


def _mv_graph(N=30, with_init=True):
    rng = np.random.default_rng(3)
    K, d = 3, 2
    pm = rng.standard_normal((K, d))
    pc = np.tile(1e6 * np.eye(d), (K, 1, 1))
    nu = np.array([3.0, 4.0, 5.0])
    sc = np.stack([(k + 1.0) * np.eye(d) + 0.1 for k in range(K)])
    init = dict(m=(pm + 1.0, pc * 0.5), w=(nu + 1.0, sc * 2.0), s=[1.0, 2.0, 3.0]) if with_init else None
    gb, ys = graph.mv_mixture_graph(N, pm, pc, nu, sc, [1.0, 1.0, 2.0], init=init)
    return gb, ys, pm, pc, nu, sc, init


def test_multivariate_mixture_graph_is_recognised():
    gb, ys, pm, pc, nu, sc, init = _mv_graph()
    for perm in (None, np.random.default_rng(5).permutation(len(gb.ftype))):
        low = graph.lower_mvgmm(gb.tables(permute=perm)[0])
        assert (low["N"], low["K"], low["d"]) == (30, 3, 2)
        assert np.array_equal(low["mu0"], pm) and np.array_equal(low["S0"], pc) and np.array_equal(low["nu0"], nu) and np.array_equal(low["V0"], sc)
        assert np.array_equal(low["alpha0"], [1.0, 1.0, 2.0]) and np.array_equal(low["init_s_alpha"], init["s"])
        assert np.array_equal(low["init_m_mean"], init["m"][0]) and np.array_equal(low["init_m_cov"], init["m"][1])
        assert np.array_equal(low["init_w_nu"], init["w"][0]) and np.array_equal(low["init_w_V"], init["w"][1])
    assert list(graph.lower_mvgmm(gb.tables()[0])["data_var"]) == ys
    with pytest.raises(rxhip.RxHipError) as ei:  # no @initialization
        graph.lower_mvgmm(_mv_graph(with_init=False)[0].tables()[0])
    assert ei.value.status == _lib.ERR_BADARG
    with pytest.raises(rxhip.RxHipError):  # a univariate mixture graph is not a multivariate one
        gu, _ = graph.mixture_graph(5, _PRI["mean"], _PRI["var"], _PRI["shape"], _PRI["rate"], _PRI["alpha"], init=_INIT)
        graph.lower_mvgmm(gu.tables()[0])


@pytest.mark.gpu
def test_multivariate_engine_from_graph_matches_structured_descriptor():
    gb, ys, pm, pc, nu, sc, init = _mv_graph(N=400)
    rng = np.random.default_rng(9)
    y = rng.standard_normal((400, 2)) * 3.0 + np.array([[4.0, -2.0]]) * rng.integers(-1, 2, (400, 1))
    e1 = graph.create_vmp_engine_from_graph(gb.tables()[0])
    e1.set_data(y); e1.run(4, True)
    h1, f1 = e1.history()["raw"], e1.free_energy(); e1.close()
    with rxhip.MvGMMEngine(400, pm, pc, nu, sc, [1.0, 1.0, 2.0], init["m"][0], init["m"][1], init["w"][0], init["w"][1], init["s"]) as e2:
        e2.set_data(y); e2.run(4, True)
        assert np.array_equal(h1, e2.history()["raw"]) and np.array_equal(f1, e2.free_energy())


# ---- generalised chain spellings, the `+` drift chain, malformed graphs (round 2) ------------------------------
def test_scalar_chain_spellings_are_recognised():
    """`x[t] ~ Normal(mean = x[t-1], var = p)`, `y[t] ~ Normal(mean = x[t], var = q)` and the `a * x` variants lower to the
    same structured descriptor (identity maps where no `*` node stands)."""
    for spell, a, b in (("normal", 1.0, 1.0), ("scaled", 0.9, 1.7), ("mixed", 0.8, 1.0)):
        for ptt in (False, True):
            gb, xs, ys = graph.scalar_chain_graph(9, a, b, 0.3, 2.0, -1.0, 25.0, prior_through_transition=ptt, spell=spell)
            perm = np.random.default_rng(3).permutation(len(gb.ftype))
            low = graph.lower_lgssm(gb.tables(permute=perm)[0])
            assert (low["d"], low["dy"], low["T"], low["prior_through_transition"], low["deterministic"]) == (1, 1, 9, ptt, False)
            assert low["A"][0, 0] == a and low["B"][0, 0] == b and low["P"][0, 0] == 0.3 and low["Q"][0, 0] == 2.0
            assert low["m0"][0] == -1.0 and low["V0"][0, 0] == 25.0 and list(low["data_var"]) == ys and list(low["state_var"]) == xs


def test_precision_parametrised_chains_are_the_covariance_form():
    """`x[i] ~ NormalMeanPrecision(z, 1.0)`, `y[i] ~ NormalMeanPrecision(x[i], 1.0)` (test/inference/prediction_tests.jl:197-213)
    and `MvNormal(μ = …, Λ = …)` transitions: constant precisions are inverted once, the chain lowers as before."""
    for ptt in (False, True):
        gb, xs, ys = graph.scalar_chain_graph(7, 1.0, 1.0, 0.25, 4.0, -1.0, 0.5, prior_through_transition=ptt, precision=True)
        low = graph.lower_lgssm(gb.tables(permute=np.random.default_rng(2).permutation(len(gb.ftype)))[0])
        ref = graph.lower_lgssm(graph.scalar_chain_graph(7, 1.0, 1.0, 0.25, 4.0, -1.0, 0.5, prior_through_transition=ptt)[0].tables()[0])
        for k in ("A", "B", "P", "Q", "m0", "V0"):
            assert np.allclose(low[k], ref[k], rtol=1e-15, atol=0), k
        assert (low["T"], low["prior_through_transition"]) == (7, ptt) and list(low["data_var"]) == ys and list(low["state_var"]) == xs
    # vector chain: Λ-parametrised transition and observation nodes
    mdl = workloads.random_model(3, 2, seed=4)
    gb, xs, ys = graph.lgssm_graph(6, mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"])
    for f, t in enumerate(gb.ftype):
        if t == _lib.NODE_MVNORMAL_MEAN_COV:
            it = list(gb.fiface[f])
            it[2] = gb.constvar(np.linalg.inv(np.asarray(gb.const_value(it[2]))))
            gb.fiface[f] = tuple(it)
            gb.ftype[f] = _lib.NODE_MVNORMAL_MEAN_PRECISION
    low = graph.lower_lgssm(gb.tables()[0])
    P, Q = np.reshape(low["P"], (-1, 3, 3))[0], np.reshape(low["Q"], (-1, 2, 2))[0]
    assert np.allclose(P, mdl["P"], rtol=1e-12) and np.allclose(Q, mdl["Q"], rtol=1e-12) and np.allclose(low["V0"], mdl["V0"], rtol=1e-12)
    assert np.array_equal(P, P.T)
    # a precision that is not positive definite is a malformed model, not an unsupported one
    gb, _, _ = graph.scalar_chain_graph(3, 1.0, 1.0, -0.25, 4.0, 0.0, 1.0, precision=True)
    with pytest.raises(rxhip.RxHipError) as ei:
        graph.lower_lgssm(gb.tables()[0])
    assert ei.value.status == _lib.ERR_BADARG and "positive definite" in str(ei.value)
    # ... and so is one that is not symmetric: the lowering does not repair it by symmetrising the inverse
    gb, xs, ys = graph.lgssm_graph(4, mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"])
    for f, t in enumerate(gb.ftype):
        if t == _lib.NODE_MVNORMAL_MEAN_COV and np.asarray(gb.const_value(gb.fiface[f][2])).shape == (3, 3):
            W = np.linalg.inv(np.asarray(gb.const_value(gb.fiface[f][2])))
            W[0, 1] += 0.05
            it = list(gb.fiface[f])
            it[2] = gb.constvar(W)
            gb.fiface[f] = tuple(it)
            gb.ftype[f] = _lib.NODE_MVNORMAL_MEAN_PRECISION
            break
    with pytest.raises(rxhip.RxHipError) as ei:
        graph.lower_lgssm(gb.tables()[0])
    assert ei.value.status == _lib.ERR_BADARG and "not symmetric" in str(ei.value)


def test_accepted_asymmetry_of_a_constant_precision_is_reported():
    """below the round-off bound (1e-8·max|W|) the symmetric part is what gets inverted — and rxhip_lowering_asymmetry() says how far the input was
    from symmetric, so that a caller can hold a tighter line"""
    mdl = workloads.random_model(3, 2, seed=4)

    def chain_with_precision(eps):
        gb, xs, ys = graph.lgssm_graph(4, mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"])
        for f, t in enumerate(gb.ftype):
            if t == _lib.NODE_MVNORMAL_MEAN_COV and np.asarray(gb.const_value(gb.fiface[f][2])).shape == (3, 3):
                W = np.linalg.inv(np.asarray(gb.const_value(gb.fiface[f][2])))
                W = 0.5 * (W + W.T)
                W[0, 1] += eps * np.max(np.abs(W))
                it = list(gb.fiface[f])
                it[2] = gb.constvar(W)
                gb.fiface[f] = tuple(it)
                gb.ftype[f] = _lib.NODE_MVNORMAL_MEAN_PRECISION
                break
        return gb
    low0 = graph.lower_lgssm(chain_with_precision(0.0).tables()[0])
    assert graph.lowering_asymmetry() == 0.0
    low = graph.lower_lgssm(chain_with_precision(3e-10).tables()[0])
    assert graph.lowering_asymmetry() == pytest.approx(3e-10, rel=1e-3)
    assert np.allclose(low["P"], low0["P"], rtol=1e-8) and np.array_equal(low["P"][0], low["P"][0].T)
    graph.lower_lgssm(chain_with_precision(0.0).tables()[0])
    assert graph.lowering_asymmetry() == 0.0          # per call, not sticky
    with pytest.raises(rxhip.RxHipError):
        graph.lower_lgssm(chain_with_precision(1e-6).tables()[0])


def test_identity_observation_in_a_vector_chain():
    """`y[t] ~ MvNormal(μ = x[t], Σ = Q)` without a `*` node: B = I."""
    mdl = workloads.random_model(3, 3, seed=2)
    gb = graph.GraphBuilder()
    x = gb.randomvar(3)
    gb.mvnormal_mean_cov(x, gb.constvar(mdl["m0"]), gb.constvar(mdl["V0"]))
    ys = []
    for t in range(6):
        if t:
            a = gb.randomvar(3); gb.multiply(a, gb.constvar(mdl["A"]), x)
            xn = gb.randomvar(3); gb.mvnormal_mean_cov(xn, a, gb.constvar(mdl["P"])); x = xn
        y = gb.datavar(3); gb.mvnormal_mean_cov(y, x, gb.constvar(mdl["Q"])); ys.append(y)
    low = graph.lower_lgssm(gb.tables()[0])
    assert np.array_equal(low["B"], np.eye(3)) and np.array_equal(low["A"], mdl["A"]) and low["T"] == 6


def test_drift_chain_of_the_reference_test_is_recognised():
    """test/models/statespace/ulgssm_tests.jl:8-15: `x[i] ~ x_prev + c` (typeof(+) with a constant), either argument order."""
    for const_first in (False, True):
        gb, xs, ys = graph.drift_chain_graph(12, 0.0, 1e4, 1.0, 100.0, const_first=const_first)
        assert len(gb.ftype) == 1 + 2 * 12
        low = graph.lower_lgssm(gb.tables(permute=np.random.default_rng(5).permutation(len(gb.ftype)))[0])
        assert low["deterministic"] and low["prior_through_transition"] and low["T"] == 12
        assert low["c"][0] == 1.0 and low["Q"][0, 0] == 100.0 and low["V0"][0, 0] == 1e4 and low["P"][0, 0] == 0.0
        assert list(low["data_var"]) == ys
    # mixing noisy and noise-free transitions has no schedule
    gb, xs, ys = graph.drift_chain_graph(4, 0.0, 1.0, 1.0, 1.0)
    xn = gb.randomvar(1); gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, xn, xs[-1], gb.constvar(0.5))
    y = gb.datavar(1); gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, y, xn, gb.constvar(1.0))
    with pytest.raises(rxhip.RxHipError) as ei:
        graph.lower_lgssm(gb.tables()[0])
    assert ei.value.status == _lib.ERR_UNSUPPORTED


def test_cyclic_and_merging_graphs_are_rejected_not_followed():
    """a chain whose last transition writes back into x[1] used to be walked forever."""
    mdl = workloads.notebook_model()
    gb, xs, ys = graph.lgssm_graph(4, mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"])
    a = gb.randomvar(2); gb.multiply(a, gb.constvar(mdl["A"]), xs[-1])
    gb.mvnormal_mean_cov(xs[0], a, gb.constvar(mdl["P"]))  # x[1] now has two writers: the prior and this transition
    with pytest.raises(rxhip.RxHipError) as ei:
        graph.lower_lgssm(gb.tables()[0])
    assert ei.value.status == _lib.ERR_UNSUPPORTED and "two nodes" in str(ei.value)
    # pure cycle without a second writer on the prior's variable: x[2] written by the last transition
    gb, xs, ys = graph.lgssm_graph(4, mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"])
    a = gb.randomvar(2); gb.multiply(a, gb.constvar(mdl["A"]), xs[-1])
    gb.mvnormal_mean_cov(xs[1], a, gb.constvar(mdl["P"]))
    with pytest.raises(rxhip.RxHipError) as ei:
        graph.lower_lgssm(gb.tables()[0])
    assert ei.value.status == _lib.ERR_UNSUPPORTED


def test_malformed_constant_tables_are_bad_arguments():
    """constants without a value (offset −1), offsets past the pool, CSR offsets that go backwards."""
    mdl = workloads.notebook_model()
    gb, xs, ys = graph.lgssm_graph(3, mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"])
    g, keep = gb.tables()
    consts = [i for i, k in enumerate(gb.kind) if k == _lib.VARKIND_CONST]
    keep["coff"][consts[-1]] = -1  # a constant that carries no value
    with pytest.raises(rxhip.RxHipError) as ei:
        graph.lower_lgssm(g)
    assert ei.value.status in (_lib.ERR_UNSUPPORTED, _lib.ERR_BADARG)
    g, keep = gb.tables()
    keep["coff"][consts[-1]] = keep["pool"].size - 1  # 2×2 matrix starting at the last pool entry
    with pytest.raises(rxhip.RxHipError) as ei:
        graph.lower_lgssm(g)
    assert ei.value.status == _lib.ERR_BADARG
    gbm, _ = graph.mixture_graph(4, [0.0, 1.0], [1.0, 1.0], [1.0, 1.0], [1.0, 1.0], [1.0, 1.0],
                                 init=dict(m=([0.0, 1.0], [1.0, 1.0]), p=([1.0, 1.0], [1.0, 1.0])))
    g, keep = gbm.tables()
    keep["ptr"][2] = keep["ptr"][1] - 1
    with pytest.raises(rxhip.RxHipError) as ei:
        graph.lower_gmm(g)
    assert ei.value.status == _lib.ERR_BADARG


def test_per_step_constants_are_grouped_into_models():
    """`A[t] * x[t-1]`, `Σ = Q[t]` in the @model loop: equal constants share a model, whatever variable carries them."""
    mdl = workloads.c1_model()
    T = 9
    A_of_t = lambda t: mdl["A"] * (1.0 + 0.01 * (t % 3 == 0))          # two transition regimes
    Q_of_t = lambda t: mdl["Q"] * (2.0 if t >= 6 else 1.0)             # the noise doubles at t = 6
    for ptt in (False, True):
        gb, xs, ys = graph.lgssm_graph(T, mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], prior_through_transition=ptt,
                                       A_of_t=A_of_t, Q_of_t=Q_of_t)
        perm = np.random.default_rng(0).permutation(len(gb.ftype))
        low = graph.lower_lgssm(gb.tables(permute=perm)[0])
        sm = low["step_model"]
        assert low["T"] == T and low["n_models"] == len(set(sm)) and low["A"].shape == (low["n_models"], 4, 4)
        for t in range(T):
            if t > 0 or ptt:   # without a transition into the first state its A is never used (the next step's is stored)
                assert np.array_equal(low["A"][sm[t]], A_of_t(t))
            assert np.array_equal(low["Q"][sm[t]], Q_of_t(t)) and np.array_equal(low["B"][sm[t]], mdl["B"])
        assert low["n_models"] == (4 if ptt else 4)
    # time-invariant graphs keep plain matrices
    gb, xs, ys = graph.lgssm_graph(T, mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"])
    low = graph.lower_lgssm(gb.tables()[0])
    assert low["n_models"] == 1 and low["step_model"] is None and low["A"].shape == (4, 4)
