"""Per-step constants A_t, P_t, B_t, Q_t (`x[t] ~ MvNormal(μ = A[t] * x[t-1], Σ = P[t])` in the @model loop; the reference
builds one `*` / MvNormal node per step, so every step may carry its own constant): oracle against brute-force conditioning of
the joint Gaussian (CPU), device (rxhip_lgssm_desc.step_model) against the oracle (GPU)."""
import numpy as np
import pytest

from oracle import rxoracle as rxo


def _models(rng, d, dy, M):
    A = np.stack([0.9 * np.linalg.qr(rng.standard_normal((d, d)))[0] for _ in range(M)])
    B = rng.standard_normal((M, dy, d))
    P = np.stack([np.eye(d) * (0.1 + 0.4 * rng.random()) + 0.03 for _ in range(M)])
    Q = np.stack([np.eye(dy) * (0.2 + rng.random()) + 0.05 for _ in range(M)])
    m0 = np.tile(rng.standard_normal(d), (M, 1))
    V0 = np.tile(np.eye(d) * 2.0, (M, 1, 1))
    return A, B, P, Q, m0, V0


def _simulate(rng, mdl, sm, C, ptt):
    A, B, P, Q, m0, V0 = mdl
    d, dy, T = A.shape[-1], B.shape[-2], len(sm)
    y = np.empty((C, T, dy))
    for c in range(C):
        x = rng.multivariate_normal(m0[0], V0[0])
        for t in range(T):
            if t or ptt:
                x = A[sm[t]] @ x + rng.multivariate_normal(np.zeros(d), P[sm[t]])
            y[c, t] = B[sm[t]] @ x + rng.multivariate_normal(np.zeros(dy), Q[sm[t]])
    return y


def _joint_posterior(mdl, sm, y, ptt):
    """Brute force: joint Gaussian of (x_1..x_T, y_1..y_T), conditioned on the observed rows of y."""
    A, B, P, Q, m0, V0 = mdl
    d, dy, T = A.shape[-1], B.shape[-2], len(sm)
    mx, Vx = np.zeros((T, d)), np.zeros((T, d, T, d))
    if ptt:
        mx[0], Vx[0, :, 0, :] = A[sm[0]] @ m0[0], A[sm[0]] @ V0[0] @ A[sm[0]].T + P[sm[0]]
    else:
        mx[0], Vx[0, :, 0, :] = m0[0], V0[0]
    for t in range(1, T):
        At = A[sm[t]]
        mx[t] = At @ mx[t - 1]
        Vx[t, :, t, :] = At @ Vx[t - 1, :, t - 1, :] @ At.T + P[sm[t]]
        for s in range(t):
            Vx[t, :, s, :] = At @ Vx[t - 1, :, s, :]
            Vx[s, :, t, :] = Vx[t, :, s, :].T
    Vx = Vx.reshape(T * d, T * d)
    Bb = np.zeros((T * dy, T * d))
    Qb = np.zeros((T * dy, T * dy))
    for t in range(T):
        Bb[t * dy:(t + 1) * dy, t * d:(t + 1) * d] = B[sm[t]]
        Qb[t * dy:(t + 1) * dy, t * dy:(t + 1) * dy] = Q[sm[t]]
    keep = np.flatnonzero(~np.isnan(y).any(axis=1))
    idx = (keep[:, None] * dy + np.arange(dy)).ravel()
    Syy = (Bb @ Vx @ Bb.T + Qb)[np.ix_(idx, idx)]
    Vxy = (Vx @ Bb.T)[:, idx]
    r = y[keep].ravel() - (Bb @ mx.ravel())[idx]
    K = np.linalg.solve(Syy, Vxy.T).T
    pm = (mx.ravel() + K @ r).reshape(T, d)
    pV = Vx - K @ Vxy.T
    nll = 0.5 * (idx.size * np.log(2 * np.pi) + np.linalg.slogdet(Syy)[1] + r @ np.linalg.solve(Syy, r))
    return pm, np.stack([pV[t * d:(t + 1) * d, t * d:(t + 1) * d] for t in range(T)]), nll


@pytest.mark.parametrize("d,dy,ptt", [(1, 1, False), (2, 1, True), (3, 2, False), (4, 3, True)])
def test_oracle_with_per_step_constants_is_the_conditional_of_the_joint(d, dy, ptt):
    rng = np.random.default_rng(7 * d + dy)
    T = 10
    mdl = _models(rng, d, dy, T)
    sm = rng.permutation(T)
    y = _simulate(rng, mdl, sm, 1, ptt)[0]
    y[[2, 6]] = np.nan
    mean, cov, nll = rxo.lgssm_kalman_rts_tv(*mdl, sm, y, prior_through_transition=ptt)
    pm, pV, ref = _joint_posterior(mdl, sm, y, ptt)
    assert np.allclose(mean, pm, rtol=1e-9, atol=1e-11) and np.allclose(cov, pV, rtol=1e-9, atol=1e-11)
    assert nll == pytest.approx(ref, rel=1e-10)


def test_one_model_schedule_is_the_time_invariant_oracle():
    rng = np.random.default_rng(2)
    mdl = _models(rng, 2, 2, 1)
    sm = np.zeros(30, dtype=np.int32)
    y = _simulate(rng, mdl, sm, 1, False)[0]
    a = rxo.lgssm_kalman_rts_tv(*mdl, sm, y)
    b = rxo.lgssm_kalman_rts(*(x[0] for x in mdl), y)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]


# ---------------------------------------------------------------------------------------------------------------- device
@pytest.mark.gpu
@pytest.mark.parametrize("d,dy,ptt,M", [(1, 1, True, 3), (2, 1, False, 2), (2, 2, True, 40), (3, 2, False, 5), (4, 4, True, 40),
                                        (4, 1, False, 7)])
@pytest.mark.parametrize("segments", [0, 1, 6], ids=lambda s: f"seg{s}")
def test_per_step_constants_match_the_oracle(d, dy, ptt, M, segments):
    import rxhip
    rng = np.random.default_rng(31 * d + dy + M)
    C, T = 21, 40
    mdl = _models(rng, d, dy, M)
    sm = (rng.permutation(T) % M).astype(np.int32)
    y = _simulate(rng, mdl, sm, C, ptt)
    with rxhip.LGSSMEngine(*mdl, T=T, n_chains=C, prior_through_transition=ptt, step_model=sm, segments=segments) as eng:
        eng.set_data(y, layout="chain_time")
        eng.run(free_energy=True)
        mean, cov = eng.marginals(layout="chain_time")
        fe = eng.free_energy_per_chain()
        pm, pc = eng.predictions(layout="chain_time")
        eng.run_filter(free_energy=False)
        fm, fc = eng.marginals(layout="chain_time")
    for c in range(C):
        om, oc, nll = rxo.lgssm_kalman_rts_tv(*mdl, sm, y[c], prior_through_transition=ptt)
        assert np.allclose(mean[c], om, rtol=1e-6, atol=1e-9) and np.allclose(cov[c], oc, rtol=1e-6, atol=1e-9)
        assert fe[c] == pytest.approx(nll, rel=1e-8, abs=1e-9)
        for t in (0, T // 2, T - 1):   # leave-one-out prediction of y[t] = the smoother WITHOUT y[t], through B_t, Q_t
            yl = y[c].copy()
            yl[t] = np.nan
            lm, lc, _ = rxo.lgssm_kalman_rts_tv(*mdl, sm, yl, prior_through_transition=ptt)
            Bt, Qt = mdl[1][sm[t]], mdl[3][sm[t]]
            assert np.allclose(pm[c, t], Bt @ lm[t], rtol=1e-6, atol=1e-8)
            assert np.allclose(pc[c, t], Bt @ lc[t] @ Bt.T + Qt, rtol=1e-6, atol=1e-8)
        # filtering: the smoother of the first t+1 observations ends in the filtered belief of t
        for t in (0, 5, T - 1):
            qm, qc, _ = rxo.lgssm_kalman_rts_tv(*mdl, sm[:t + 1], y[c, :t + 1], prior_through_transition=ptt)
            assert np.allclose(fm[c, t], qm[-1], rtol=1e-6, atol=1e-9) and np.allclose(fc[c, t], qc[-1], rtol=1e-6, atol=1e-9)


@pytest.mark.gpu
def test_per_step_constants_with_missing_values_and_a_forecast_horizon():
    import rxhip
    rng = np.random.default_rng(77)
    d, dy, T, H, C = 3, 2, 50, 6, 5
    mdl = _models(rng, d, dy, T + H)
    sm = np.arange(T + H, dtype=np.int32)
    y = _simulate(rng, mdl, sm, C, False)
    y[:, T:] = np.nan
    y[1, [4, 5, 20]] = np.nan
    with rxhip.LGSSMEngine(*mdl, T=T, n_chains=C, step_model=sm, horizon=H, allow_missing=True) as eng:
        eng.set_data(y[:, :T], layout="chain_time")
        eng.run(free_energy=True)
        mean, cov = eng.marginals(layout="chain_time")
        pm, pc = eng.predictions(layout="chain_time")
        fe = eng.free_energy_per_chain()
    assert mean.shape == (C, T + H, d) and pm.shape == (C, T + H, dy)
    for c in range(C):
        om, oc, nll = rxo.lgssm_kalman_rts_tv(*mdl, sm, y[c])
        assert np.allclose(mean[c], om, rtol=1e-6, atol=1e-9) and np.allclose(cov[c], oc, rtol=1e-6, atol=1e-9)
        assert fe[c] == pytest.approx(nll, rel=1e-8)
        for t in range(T, T + H):
            Bt, Qt = mdl[1][sm[t]], mdl[3][sm[t]]
            assert np.allclose(pm[c, t], Bt @ om[t], rtol=1e-6, atol=1e-9)
            assert np.allclose(pc[c, t], Bt @ oc[t] @ Bt.T + Qt, rtol=1e-6, atol=1e-9)


@pytest.mark.gpu
def test_infer_mirror_with_arrays_of_matrices():
    import rxhip
    rng = np.random.default_rng(5)
    T = 64
    th = np.linspace(0.05, 0.4, T)
    A = np.stack([[[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]] for a in th])     # a rotation that speeds up
    Q = np.stack([np.eye(1) * (0.5 if t % 2 else 2.0) for t in range(T)])              # two observation regimes
    B, P = np.array([[1.0, 0.0]]), np.eye(2) * 0.1
    spec = rxhip.time_varying_gaussian_ssm(A, B, P, Q, np.zeros(2), np.eye(2) * 10)
    assert spec.A.shape[0] == T and spec.step_model.shape == (T,)
    y = rng.standard_normal((T, 1))
    res = rxhip.infer(model=spec, data={"y": y}, free_energy=True)
    M = spec.A.shape[0]
    om, oc, nll = rxo.lgssm_kalman_rts_tv(spec.A, spec.B, spec.P, spec.Q, np.tile(spec.prior_mean, (M, 1)),
                                          np.tile(spec.prior_cov, (M, 1, 1)), spec.step_model, y)
    assert np.allclose(res.posteriors["x"].mean, om, rtol=1e-6, atol=1e-9)
    assert np.allclose(res.posteriors["x"].cov, oc, rtol=1e-6, atol=1e-9)
    assert res.free_energy[-1] == pytest.approx(nll, rel=1e-8)
    # only Q varies: two models
    spec2 = rxhip.time_varying_gaussian_ssm(A[0], B, P, Q, np.zeros(2), np.eye(2) * 10)
    assert spec2.A.shape[0] == 2 and set(spec2.step_model) == {0, 1}


@pytest.mark.gpu
def test_step_model_is_validated():
    import rxhip
    rng = np.random.default_rng(1)
    mdl = _models(rng, 2, 1, 2)
    with pytest.raises(Exception):
        rxhip.LGSSMEngine(*mdl, T=4, n_chains=1, step_model=np.array([0, 1, 2, 0], dtype=np.int32))
    with pytest.raises(Exception):
        rxhip.LGSSMEngine(*mdl, T=4, n_chains=2, step_model=np.zeros(4, dtype=np.int32), chain_model=np.zeros(2, dtype=np.int32))
    big = _models(rng, 6, 2, 2)   # larger states run the sequential schedule (tests/test_dense_sequential.py)
    rxhip.LGSSMEngine(*big, T=4, n_chains=2, step_model=np.zeros(4, dtype=np.int32)).close()


@pytest.mark.gpu
def test_graph_with_per_step_constants_and_missing_data_runs_on_the_device():
    """The route of the Julia plugin: tables of the graph GraphPPL builds (a new constant variable per use) -> rxhip_create."""
    import rxhip
    from rxhip import graph
    rng = np.random.default_rng(12)
    d, dy, T, C = 2, 2, 30, 4
    A, B, P, Q, m0, V0 = _models(rng, d, dy, 3)
    sm = (np.arange(T) // 4) % 3
    y = _simulate(rng, (A, B, P, Q, m0, V0), sm, C, True)
    y[2, [3, 9, 10]] = np.nan
    gb, xs, ys = graph.lgssm_graph(T, A[0], B[0], P[0], Q[0], m0[0], V0[0], prior_through_transition=True,
                                   A_of_t=lambda t: A[sm[t]], P_of_t=lambda t: P[sm[t]], B_of_t=lambda t: B[sm[t]],
                                   Q_of_t=lambda t: Q[sm[t]])
    perm = rng.permutation(len(gb.ftype))
    g, keep = gb.tables(n_replicas=C, permute=perm, allow_missing=True)
    eng = graph.create_engine_from_graph(g)
    eng.set_data(y, layout="chain_time")
    eng.run(1, True)
    mean, cov = eng.marginals(layout="chain_time")
    fe = eng.free_energy_per_chain()
    eng.close()
    for c in range(C):
        om, oc, nll = rxo.lgssm_kalman_rts_tv(A, B, P, Q, m0, V0, sm, y[c], prior_through_transition=True)
        assert np.allclose(mean[c], om, rtol=1e-6, atol=1e-9) and np.allclose(cov[c], oc, rtol=1e-6, atol=1e-9)
        assert fe[c] == pytest.approx(nll, rel=1e-8)
