"""oracle/tree_oracle.py (generic sum-product / mean-field VMP on acyclic Gaussian graphs) pinned three ways: brute-force conditioning of the
joint Gaussian, oracle/rxoracle.c on the state-space graphs (itself pinned to the reference's goldens), the reference's RNG-free known
answers.  CPU only."""
import numpy as np
import pytest

import tree_graphs as tg
import tree_oracle


def _check_bruteforce(gb, ys, seed=0, tol=1e-9):
    row = tg.random_data(gb, ys, 1, seed)[0]
    data = tg.data_dict(gb, ys, row)
    res = tree_oracle.infer(gb.to_dump(), data, iterations=1, free_energy=True)
    post, nle = tg.brute_force(gb, data)
    for v, (m, V) in post.items():
        sd = np.sqrt(np.diag(V)) + 1e-300
        if np.all(np.diag(V) < 1e-14):   # a deterministic image of the data (no posterior spread)
            continue
        assert np.max(np.abs(res["mean"][v] - m) / sd) < tol, v
        assert np.max(np.abs(res["cov"][v] - V) / np.outer(sd, sd)) < tol, v
    return res, nle


@pytest.mark.parametrize("builder,kw", [(tg.two_branch_chain, dict(T=6)), (tg.two_branch_chain, dict(T=5, d=2, dy1=2, dy2=2, precision_spelling=True)),
                                        (tg.branching_tree, dict(depth=2, fanout=2)), (tg.branching_tree, dict(depth=3, fanout=2, d=3, observe_leaves_only=True)),
                                        (tg.scalar_tree, dict(n_leaves=4)), (tg.chain_with_prediction, dict(T=5, H=3)), (tg.star, dict(n_leaves=40, d=2))])
def test_marginals_and_free_energy_equal_the_joint_gaussian(builder, kw):
    gb, ys, _ = builder(**kw)
    res, nle = _check_bruteforce(gb, ys)
    if builder is not tg.branching_tree:   # (B x with more rows than columns: H[q(Bx)] = −∞ in the Bethe sum, as in the reference)
        assert res["fe"][0] == pytest.approx(nle, rel=1e-9, abs=1e-9)


def test_branching_tree_free_energy_with_full_rank_maps():
    gb, ys, _ = tg.branching_tree(depth=2, fanout=3, d=1, seed=5)
    res, nle = _check_bruteforce(gb, ys, seed=2)
    assert res["fe"][0] == pytest.approx(nle, rel=1e-9, abs=1e-9)


def test_known_answers_of_the_reference():
    """/root/reference/test/models/models_tests.jl:242-256 (x ~ N(a + b, 1), y ~ N(x, 1); a = 2, b = 1, y = 0 -> 3.51551, mean 1.5) and :294-308"""
    from rxhip import _lib
    from rxhip.graph import GraphBuilder
    gb = GraphBuilder()
    a, b, y = gb.datavar(1), gb.datavar(1), gb.datavar(1)
    s = gb.randomvar(1)
    gb.node(_lib.NODE_ADD, s, a, b)
    x = gb.randomvar(1)
    gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, x, s, gb.constvar(1.0))
    gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, y, x, gb.constvar(1.0))
    res = tree_oracle.infer(gb.to_dump(), {a: [2.0], b: [1.0], y: [0.0]})
    assert res["fe"][0] == pytest.approx(3.51551, abs=1e-5)
    assert res["mean"][x][0] == pytest.approx(1.5, abs=1e-12)


def test_state_space_chain_equals_the_c_oracle():
    import rxoracle as rxo
    from rxhip import workloads
    from rxhip.graph import lgssm_graph
    m = workloads.random_model(3, 3, seed=4)   # (dy < d: the reference schedule inverts a rank-deficient message at the last state, rxoracle.c returns NOT_POSDEF)
    T = 40
    y = workloads.generate_batch(m, T, 1, seed0=1)[:, 0]
    gb, xs, ys = lgssm_graph(T, m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"])
    res = tree_oracle.infer(gb.to_dump(), {v: y[t] for t, v in enumerate(ys)})
    om, oc, fe, cnt = rxo.lgssm_bp(m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"], y)
    for t, v in enumerate(xs):
        assert np.allclose(res["mean"][v], om[t], rtol=1e-10, atol=1e-12)
        assert np.allclose(res["cov"][v], oc[t], rtol=1e-10, atol=1e-12)
    assert res["fe"][0] == pytest.approx(fe, rel=1e-11)
    assert res["counters"]["rule_calls"] == cnt.rule_calls


@pytest.mark.parametrize("gamma", [None, "rate"])
def test_observation_noise_vmp_equals_the_c_oracle(gamma):
    import rxoracle as rxo
    from rxhip import workloads
    from rxhip.graph import lgssm_noise_graph
    d, dy = (2, 2) if not gamma else (2, 1)
    m = workloads.random_model(d, dy, seed=9)
    T, its = 30, 5
    y = workloads.generate_batch(m, T, 1, seed0=3)[:, 0]
    if gamma:
        gb, xs, ys, W = lgssm_noise_graph(T, m["A"], m["B"], m["P"], m["m0"], m["V0"], 2.0, 0.7, init=(2.0, 1.5), gamma=gamma)
        nu0, S0, inu, iV = 4.0, np.array([[1.0 / 1.4]]), 4.0, np.array([[1.0 / 3.0]])
    else:
        S0 = np.eye(dy) * 0.5
        gb, xs, ys, W = lgssm_noise_graph(T, m["A"], m["B"], m["P"], m["m0"], m["V0"], dy + 2.0, S0, init=(dy + 1.0, np.eye(dy)))
        nu0, inu, iV = dy + 2.0, dy + 1.0, np.eye(dy)
    res = tree_oracle.infer(gb.to_dump(), {v: y[t] for t, v in enumerate(ys)}, iterations=its)
    pm, pc, wh, fe = rxo.lgssm_noise_vmp(m["A"], m["B"], m["P"], m["m0"], m["V0"], y, nu0, S0, inu, iV, its)
    for t, v in enumerate(xs):
        assert np.allclose(res["mean"][v], pm[t], rtol=1e-9, atol=1e-11)
        assert np.allclose(res["cov"][v], pc[t], rtol=1e-9, atol=1e-11)
    assert np.allclose(res["fe"], fe, rtol=1e-10)
    nu, V = res["q_prec"][W]
    assert nu == pytest.approx(wh[-1][0]) and np.allclose(V.ravel(), wh[-1][1:], rtol=1e-10)


def test_state_noise_vmp_free_energy_decreases():
    gb, ys, named = tg.chain_state_noise_precision(T=25, d=2, dy=2)
    row = tg.random_data(gb, ys, 1, 3)[0]
    res = tree_oracle.infer(gb.to_dump(), tg.data_dict(gb, ys, row), iterations=12)
    fe = np.array(res["fe"])
    assert np.all(np.diff(fe) <= 1e-9 * np.abs(fe[:-1])), fe
    assert np.isfinite(fe).all()


@pytest.mark.parametrize("seed", range(24))
def test_random_forests_against_brute_force(seed):
    """random acyclic graphs with constant noise parameters (tests/tree_graphs.py::random_forest): the generic restatement against the joint Gaussian
    conditioned on the data — every named posterior and the Bethe free energy (= −log evidence on a tree)"""
    gb, ys, named = tg.random_forest(seed, n_steps=12, dmax=5 if seed % 3 else 12)
    data = tg.data_dict(gb, ys, tg.random_data(gb, ys, 1, seed)[0])
    ref = tree_oracle.infer(gb.to_dump(), data)
    post, nle = tg.brute_force(gb, data)
    for v in named["x"]:
        sd = np.sqrt(np.diag(post[v][1]))
        assert np.max(np.abs(ref["mean"][v] - post[v][0]) / sd) < 1e-8, v
        assert np.max(np.abs(ref["cov"][v] - post[v][1]) / np.outer(sd, sd)) < 1e-8, v
    assert ref["fe"][0] == pytest.approx(nle, rel=1e-9, abs=1e-9)


@pytest.mark.parametrize("kw", [dict(n=40, d=2), dict(n=25, d=3), dict(n=30, d=1, gamma=True)])
def test_known_mean_precision_model_is_the_conjugate_closed_form(kw):
    """test/models/iid/mv_iid_precision_known_mean_tests.jl: no Gaussian random variable at all — q(P) is the conjugate Wishart after one iteration and the free
    energy is −log evidence at EVERY iteration (the reference asserts `all(==(first(fe)), fe)`); both in closed form, no RNG of the reference needed"""
    gb, ys, named = tg.known_mean_precision(**kw)
    d = gb.rows[ys[0]]
    y = np.random.default_rng(3).standard_normal((len(ys), d)) * 1.7 + named["m"]
    ref = tree_oracle.infer(gb.to_dump(), {v: y[i] for i, v in enumerate(ys)}, iterations=4)
    nu, V, nle = tg.known_mean_closed_form(y, named["m"], *named["prior"])
    qn, qV = ref["q_prec"][named["W"][0]]
    assert qn == pytest.approx(nu, rel=1e-14) and np.allclose(qV, V, rtol=1e-11)
    assert np.allclose(ref["fe"], nle, rtol=1e-12)


@pytest.mark.parametrize("kw", [dict(T=5, d=2, dy=2), dict(T=4, d=3, dy=2, branches=2), dict(T=6, d=1, dy=1), dict(T=6, d=2, dy=1, partial=True)])
def test_mean_field_between_gaussian_interfaces_converges_to_the_closed_form(kw):
    """`constraints = MeanField()` on a Gaussian chain: q(out) q(μ) around every Gaussian node (factor_cluster of the graph tables = the reference's
    VariationalConstraintsFactorizationIndicesKey).  The rules read marginals — MvNormalMeanCovariance(:out)(q_μ, q_Σ) = N(mean(q_μ), Σ) — so the posterior
    is a fixed point of iterations; its closed form is known without any schedule: the exact posterior means and, per cluster of variables that still share a
    factor of q, the inverse of the cluster's block of the joint precision; the free energy there is E_q[−log p] − Σ H[q_c].  (Mixed graphs too: `partial`.)"""
    gb, ys, named = tg.mean_field_chain(**kw)
    data = tg.data_dict(gb, ys, tg.random_data(gb, ys, 1, 3)[0])
    out = tree_oracle.infer(gb.to_dump(), data, iterations=1500 if kw.get("partial") else 400)   # (the simultaneous update converges linearly: the dy < d mixed chain is the slow one)
    post, fe = tg.mean_field_fixed_point(gb, data)
    for v in named["x"]:
        sd = np.sqrt(np.diag(post[v][1]))
        assert np.max(np.abs(out["mean"][v] - post[v][0]) / sd) < 1e-9, v
        assert np.allclose(out["cov"][v], post[v][1], rtol=1e-9, atol=1e-12)
    assert out["fe"][-1] == pytest.approx(fe, rel=1e-10)
    # the bound tightens along the iterations (late: monotone) and stays above the evidence of the structured (exact) posterior
    exact = tree_oracle.infer(gb.bethe().to_dump(), data)["fe"][-1]
    assert out["fe"][-1] > exact and np.all(np.diff(out["fe"][50:]) <= 1e-9 * abs(fe))
    # a missing @initialization is an error, as in the reference
    gb2, ys2, _ = tg.mean_field_chain(**kw)
    gb2.init_family.clear()
    gb2.init_off.clear()
    with pytest.raises(ValueError):
        tree_oracle.infer(gb2.to_dump(), data, iterations=2)


@pytest.mark.parametrize("seed", range(6))
def test_missing_observations_are_dropped_observations(seed):
    """a `missing` (NaN) observation sends no message and contributes nothing to the free energy: posterior and −log evidence equal those of brute-force
    conditioning on the remaining observations (the joint Gaussian with the missing nodes removed)"""
    gb, ys, nm = tg.two_branch_chain(T=7, d=3, dy1=2, dy2=1, seed=seed)
    data = tg.data_dict(gb, ys, tg.random_data(gb, ys, 1, seed)[0])
    rng = np.random.default_rng(seed)
    gone = [v for v in ys if rng.random() < 0.35] or [ys[1]]
    with_nan = {v: (np.full_like(x, np.nan) if v in gone else x) for v, x in data.items()}
    out = tree_oracle.infer(gb.to_dump(), with_nan)
    # the same model without those observation nodes
    from rxhip.graph import GraphBuilder
    keep = [f for f, ifs in enumerate(gb.fiface) if ifs[0] not in gone]
    gb2 = GraphBuilder.from_dump(gb.to_dump())
    gb2.ftype = [gb.ftype[f] for f in keep]
    gb2.fiface = [gb.fiface[f] for f in keep]
    bf, nle = tg.brute_force(gb2, data)
    for v in nm["x"]:
        sd = np.sqrt(np.diag(bf[v][1]))
        assert np.max(np.abs(out["mean"][v] - bf[v][0]) / sd) < 1e-9 and np.allclose(out["cov"][v], bf[v][1], rtol=1e-9, atol=1e-12)
    assert out["fe"][-1] == pytest.approx(nle, rel=1e-10)
