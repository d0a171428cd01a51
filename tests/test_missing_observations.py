"""`missing` observations anywhere in the data (reference: docs/src/manuals/inference/static.md:98-123): the oracle's smoother
with skipped updates against brute-force conditioning of the joint Gaussian of the whole chain (CPU), and the device's masked
schedule against the oracle (GPU)."""
import numpy as np
import pytest

from oracle import rxoracle as rxo


def _model(rng, d, dy):
    A = 0.9 * np.linalg.qr(rng.standard_normal((d, d)))[0]
    B = rng.standard_normal((dy, d))
    P = np.eye(d) * 0.3 + 0.05
    Q = np.eye(dy) * 0.5 + 0.1
    m0 = rng.standard_normal(d)
    V0 = np.eye(d) * 2.0
    return A, B, P, Q, m0, V0


def _simulate(rng, A, B, P, Q, m0, V0, T, C):
    d, dy = A.shape[0], B.shape[0]
    y = np.empty((C, T, dy))
    for c in range(C):
        x = rng.multivariate_normal(m0, V0)
        for t in range(T):
            if t:
                x = A @ x + rng.multivariate_normal(np.zeros(d), P)
            y[c, t] = B @ x + rng.multivariate_normal(np.zeros(dy), Q)
    return y


def _joint(A, B, P, Q, m0, V0, T):
    """Mean and covariance of (x_1..x_T, y_1..y_T) with x_1 ~ N(m0, V0)."""
    d, dy = A.shape[0], B.shape[0]
    mx = np.zeros((T, d))
    Vx = np.zeros((T, d, T, d))
    mx[0], Vx[0, :, 0, :] = m0, V0
    for t in range(1, T):
        mx[t] = A @ mx[t - 1]
        Vx[t, :, t, :] = A @ Vx[t - 1, :, t - 1, :] @ A.T + P
        for s in range(t):
            Vx[t, :, s, :] = A @ Vx[t - 1, :, s, :]
            Vx[s, :, t, :] = Vx[t, :, s, :].T
    Vx = Vx.reshape(T * d, T * d)
    Bb = np.kron(np.eye(T), B)
    my = Bb @ mx.ravel()
    Vy = Bb @ Vx @ Bb.T + np.kron(np.eye(T), Q)
    return mx.ravel(), Vx, my, Vy, Vx @ Bb.T


@pytest.mark.parametrize("d,dy", [(1, 1), (2, 1), (3, 2), (4, 4)])
def test_oracle_smoother_with_missing_rows_is_the_conditional_of_the_joint(d, dy):
    rng = np.random.default_rng(10 * d + dy)
    A, B, P, Q, m0, V0 = _model(rng, d, dy)
    T = 9
    y = _simulate(rng, A, B, P, Q, m0, V0, T, 1)[0]
    gone = np.array([0, 3, 4, 8] if d > 1 else [2, 3, 8])
    y[gone] = np.nan
    mean, cov, nll = rxo.lgssm_kalman_rts(A, B, P, Q, m0, V0, y, prior_through_transition=False)
    mx, Vx, my, Vy, Vxy = _joint(A, B, P, Q, m0, V0, T)
    keep = np.setdiff1d(np.arange(T), gone)
    idx = (keep[:, None] * dy + np.arange(dy)).ravel()
    Syy = Vy[np.ix_(idx, idx)]
    r = y[keep].ravel() - my[idx]
    K = np.linalg.solve(Syy, Vxy[:, idx].T).T
    pm = (mx + K @ r).reshape(T, d)
    pV = Vx - K @ Vxy[:, idx].T
    assert np.allclose(mean, pm, rtol=1e-9, atol=1e-11)
    for t in range(T):
        assert np.allclose(cov[t], pV[t * d:(t + 1) * d, t * d:(t + 1) * d], rtol=1e-9, atol=1e-11)
    ref = 0.5 * (idx.size * np.log(2 * np.pi) + np.linalg.slogdet(Syy)[1] + r @ np.linalg.solve(Syy, r))
    assert nll == pytest.approx(ref, rel=1e-10)


def test_a_partly_missing_vector_observation_is_missing():
    rng = np.random.default_rng(3)
    A, B, P, Q, m0, V0 = _model(rng, 2, 2)
    y = _simulate(rng, A, B, P, Q, m0, V0, 6, 1)[0]
    y1, y2 = y.copy(), y.copy()
    y1[2, 0] = np.nan
    y2[2] = np.nan
    a = rxo.lgssm_kalman_rts(A, B, P, Q, m0, V0, y1)
    b = rxo.lgssm_kalman_rts(A, B, P, Q, m0, V0, y2)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]


# ---------------------------------------------------------------------------------------------------------------- device
def _holes(rng, y, frac):
    y = y.copy()
    C, T, _ = y.shape
    mask = rng.random((C, T)) < frac
    mask[0, 0] = True          # the prior step itself
    mask[-1, T - 1] = True     # the last step of a chain
    mask[0, 5:9] = True        # a run
    y[mask] = np.nan
    return y, mask


@pytest.mark.gpu
@pytest.mark.parametrize("segments", [0, 1, 7, 59], ids=lambda s: f"seg{s}")
@pytest.mark.parametrize("d,dy,ptt", [(1, 1, True), (2, 1, False), (2, 2, True), (3, 2, False), (4, 1, True), (4, 4, False),
                                      (2, 3, True)])
def test_masked_schedule_matches_the_oracle(d, dy, ptt, segments):
    """segments = 0: the engine's own choice (several segments: elements computed in the lane); 1: one segment, sequential."""
    import rxhip
    rng = np.random.default_rng(100 * d + dy)
    A, B, P, Q, m0, V0 = _model(rng, d, dy)
    C, T = 37, 60
    y, mask = _holes(rng, _simulate(rng, A, B, P, Q, m0, V0, T, C), 0.25)
    y[3, 16:40] = np.nan       # whole segments without a single observation
    mask[3, 16:40] = True
    with rxhip.LGSSMEngine(A, B, P, Q, m0, V0, T=T, n_chains=C, prior_through_transition=ptt, allow_missing=True,
                           segments=segments) as eng:
        assert segments == 0 or eng.schedule()["segments"] == segments
        eng.set_data(y, layout="chain_time")
        eng.run(free_energy=True)
        mean, cov = eng.marginals(layout="chain_time")
        fe = eng.free_energy_per_chain()
        pm, pc = eng.predictions(layout="chain_time")
    for c in range(C):
        om, oc, nll = rxo.lgssm_kalman_rts(A, B, P, Q, m0, V0, y[c], prior_through_transition=ptt)
        assert np.allclose(mean[c], om, rtol=1e-6, atol=1e-9)
        assert np.allclose(cov[c], oc, rtol=1e-6, atol=1e-9)
        assert fe[c] == pytest.approx(nll, rel=1e-8, abs=1e-9)
        # predictions: the smoothed predictive where y[t] is missing, leave-one-out (= the smoother WITHOUT y[t]) where observed
        for t in np.flatnonzero(mask[c])[:4]:
            assert np.allclose(pm[c, t], B @ om[t], rtol=1e-6, atol=1e-9)
            assert np.allclose(pc[c, t], B @ oc[t] @ B.T + Q, rtol=1e-6, atol=1e-9)
        for t in np.flatnonzero(~mask[c])[:3]:
            yl = y[c].copy()
            yl[t] = np.nan
            lm, lc, _ = rxo.lgssm_kalman_rts(A, B, P, Q, m0, V0, yl, prior_through_transition=ptt)
            assert np.allclose(pm[c, t], B @ lm[t], rtol=1e-6, atol=1e-8)
            assert np.allclose(pc[c, t], B @ lc[t] @ B.T + Q, rtol=1e-6, atol=1e-8)


@pytest.mark.gpu
def test_masked_schedule_without_missing_values_is_the_plain_sweep():
    import rxhip
    rng = np.random.default_rng(5)
    A, B, P, Q, m0, V0 = _model(rng, 3, 2)
    C, T = 16, 200
    y = _simulate(rng, A, B, P, Q, m0, V0, T, C)
    out = []
    for flag in (False, True):
        with rxhip.LGSSMEngine(A, B, P, Q, m0, V0, T=T, n_chains=C, allow_missing=flag) as eng:
            eng.set_data(y, layout="chain_time")
            eng.run(free_energy=True)
            out.append((*eng.marginals(layout="chain_time"), eng.free_energy_per_chain()))
    for a, b in zip(*out):
        assert np.allclose(a, b, rtol=1e-9, atol=1e-11)


@pytest.mark.gpu
def test_infer_routes_interior_missing_values_to_the_masked_schedule():
    import rxhip
    rng = np.random.default_rng(8)
    A, B, P, Q, m0, V0 = _model(rng, 2, 1)
    T = 40
    y = _simulate(rng, A, B, P, Q, m0, V0, T, 1)[0]
    y[[3, 4, 17]] = np.nan
    y[-5:] = np.nan               # …and a tail: in this mode it is just five more missing observations
    spec = rxhip.linear_gaussian_ssm(A, B, P, Q, m0, V0)
    res = rxhip.infer(model=spec, data={"y": y}, free_energy=True, predictvars=("y",))
    om, oc, nll = rxo.lgssm_kalman_rts(A, B, P, Q, m0, V0, y, prior_through_transition=spec.prior_through_transition)
    assert np.allclose(res.posteriors["x"].mean, om, rtol=1e-6, atol=1e-9)
    assert np.allclose(res.posteriors["x"].cov, oc, rtol=1e-6, atol=1e-9)
    assert res.free_energy[-1] == pytest.approx(nll, rel=1e-8)
    for t in (3, 17, T - 1):
        assert np.allclose(res.predictions["y"].mean[t], B @ om[t], rtol=1e-6, atol=1e-9)
        assert np.allclose(res.predictions["y"].cov[t], B @ oc[t] @ B.T + Q, rtol=1e-6, atol=1e-9)


@pytest.mark.gpu
def test_masked_filtering_skips_the_update():
    import rxhip
    rng = np.random.default_rng(9)
    A, B, P, Q, m0, V0 = _model(rng, 2, 2)
    C, T = 8, 50
    y, mask = _holes(rng, _simulate(rng, A, B, P, Q, m0, V0, T, C), 0.3)
    with rxhip.LGSSMEngine(A, B, P, Q, m0, V0, T=T, n_chains=C, prior_through_transition=True, allow_missing=True) as eng:
        eng.set_data(y, layout="chain_time")
        eng.run_filter(free_energy=False)
        mean, cov = eng.marginals(layout="chain_time")
    for c in range(C):
        m, V = m0, V0
        for t in range(T):
            m, V = A @ m, A @ V @ A.T + P
            if not mask[c, t]:
                S = B @ V @ B.T + Q
                K = np.linalg.solve(S, B @ V).T
                m, V = m + K @ (y[c, t] - B @ m), V - K @ B @ V
            assert np.allclose(mean[c, t], m, rtol=1e-6, atol=1e-9)
            assert np.allclose(cov[c, t], V, rtol=1e-6, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("data", [[1.0, -500.0, np.nan, 100.0], [1.0, -500.0, np.nan, 100.0, np.nan, np.nan]], ids=["one_missing", "missing_tail"])
@pytest.mark.parametrize("iterations", [10, 20])
def test_reference_prediction_test_with_missing_values_in_an_array(data, iterations):
    """test/inference/prediction_tests.jl:197-250 ("test #1: array with missing + KeepLast predictvars"): the random walk
    x_0 ~ NormalMeanPrecision(0, 1); x[i] ~ NormalMeanPrecision(x[i-1], 1); y[i] ~ NormalMeanPrecision(x[i], 1) with two further
    states observed by o[1], o[2], for which no data is given.  The graph the plugin would emit (precision-parametrised nodes,
    `missing` entries inside y, o never observed) runs on the device: a prediction for every y[i] and both o, posteriors and
    predictions equal to the oracle's smoother that skips the missing rows — in every iteration (a tree: the iterations agree)."""
    import rxhip
    from rxhip import graph
    n = len(data)
    gb, xs, ys = graph.scalar_chain_graph(n + 2, 1.0, 1.0, 1.0, 1.0, 0.0, 1.0, prior_through_transition=True, precision=True)
    yfull = np.array(data + [np.nan, np.nan])          # o[1], o[2]: predictvars without data
    with graph.create_engine_from_graph(gb.tables(allow_missing=True)[0]) as eng:
        eng.set_data(yfull.reshape(n + 2, 1, 1))
        eng.run(iterations, True)
        mean, cov = eng.marginals()
        pm, pc = eng.predictions()
        fe = eng.free_energy()
    assert pm.shape == (n + 2, 1, 1) and np.all(np.isfinite(pm)) and np.all(pc > 0)   # length(predictions[:y]) == length(data), 2 for o
    assert fe.shape == (iterations,) and np.all(fe == fe[0])
    one = (np.eye(1), np.eye(1), np.eye(1), np.eye(1), np.zeros(1), np.eye(1))
    om, oc, nll = rxo.lgssm_kalman_rts(*one, yfull.reshape(-1, 1), prior_through_transition=True)
    assert np.allclose(mean[:, 0], om, rtol=1e-6, atol=1e-9) and np.allclose(cov[:, 0], oc, rtol=1e-6, atol=1e-9)
    assert fe[-1] == pytest.approx(nll, rel=1e-8)
    for t in range(n + 2):
        yl = yfull.copy()
        yl[t] = np.nan                                   # the message toward y[t] never contains y[t]
        lm, lc, _ = rxo.lgssm_kalman_rts(*one, yl.reshape(-1, 1), prior_through_transition=True)
        assert pm[t, 0, 0] == pytest.approx(lm[t, 0], rel=1e-6, abs=1e-8) and pc[t, 0, 0, 0] == pytest.approx(lc[t, 0, 0] + 1.0, rel=1e-6)
