"""Known inputs / offsets: x[t] ~ MvNormal(μ = A*x[t-1] + c[t], Σ = P), y[t] ~ MvNormal(μ = B*x[t] + d[t], Σ = Q) — a `+` node with
a constant behind the `*` node (control inputs B_u*u[t], drifts, biases).  The oracle's affine smoother against brute-force
conditioning of the joint Gaussian (CPU); the device — which runs the homogeneous sweep on shifted data — against the oracle."""
import numpy as np
import pytest

from oracle import rxoracle as rxo
from test_time_varying import _models


def _simulate(rng, mdl, cx, cy, C, ptt):
    A, B, P, Q, m0, V0 = (x[0] for x in mdl)
    d, dy, T = A.shape[0], B.shape[0], cx.shape[0]
    y = np.empty((C, T, dy))
    for c in range(C):
        x = rng.multivariate_normal(m0, V0)
        for t in range(T):
            if t or ptt:
                x = A @ x + cx[t] + rng.multivariate_normal(np.zeros(d), P)
            y[c, t] = B @ x + cy[t] + rng.multivariate_normal(np.zeros(dy), Q)
    return y


def _brute(mdl, cx, cy, y, ptt):
    A, B, P, Q, m0, V0 = (x[0] for x in mdl)
    d, dy, T = A.shape[0], B.shape[0], y.shape[0]
    mx, Vx = np.zeros((T, d)), np.zeros((T, d, T, d))
    if ptt:
        mx[0], Vx[0, :, 0, :] = A @ m0 + cx[0], A @ V0 @ A.T + P
    else:
        mx[0], Vx[0, :, 0, :] = m0, V0
    for t in range(1, T):
        mx[t] = A @ mx[t - 1] + cx[t]
        Vx[t, :, t, :] = A @ Vx[t - 1, :, t - 1, :] @ A.T + P
        for s in range(t):
            Vx[t, :, s, :] = A @ Vx[t - 1, :, s, :]
            Vx[s, :, t, :] = Vx[t, :, s, :].T
    Vx = Vx.reshape(T * d, T * d)
    Bb, Qb = np.kron(np.eye(T), B), np.kron(np.eye(T), Q)
    my = Bb @ mx.ravel() + cy.ravel()
    Syy = Bb @ Vx @ Bb.T + Qb
    K = np.linalg.solve(Syy, Bb @ Vx).T
    r = y.ravel() - my
    pm = (mx.ravel() + K @ r).reshape(T, d)
    pV = Vx - K @ Bb @ Vx
    nll = 0.5 * (T * dy * np.log(2 * np.pi) + np.linalg.slogdet(Syy)[1] + r @ np.linalg.solve(Syy, r))
    return pm, np.stack([pV[t * d:(t + 1) * d, t * d:(t + 1) * d] for t in range(T)]), nll


@pytest.mark.parametrize("d,dy,ptt", [(1, 1, False), (2, 2, True), (3, 2, False), (4, 3, True)])
def test_affine_oracle_is_the_conditional_of_the_joint(d, dy, ptt):
    rng = np.random.default_rng(3 * d + dy)
    mdl, T = _models(rng, d, dy, 1), 9
    cx, cy = rng.standard_normal((T, d)), rng.standard_normal((T, dy))
    y = _simulate(rng, mdl, cx, cy, 1, ptt)[0]
    m, V, nll = rxo.lgssm_kalman_rts_affine(*(x[0] for x in mdl), y, cx, cy, prior_through_transition=ptt)
    pm, pV, ref = _brute(mdl, cx, cy, y, ptt)
    assert np.allclose(m, pm, rtol=1e-9, atol=1e-11) and np.allclose(V, pV, rtol=1e-9, atol=1e-11)
    assert nll == pytest.approx(ref, rel=1e-10)


# ---------------------------------------------------------------------------------------------------------------- device
@pytest.mark.gpu
@pytest.mark.parametrize("d,dy,ptt,C,T,H", [(1, 1, True, 3, 50, 0), (2, 2, False, 64, 300, 5), (4, 4, True, 70, 120, 0), (3, 1, False, 5, 40, 3),
                                            (6, 6, False, 4, 60, 4), (16, 12, True, 2, 40, 0), (64, 64, False, 1, 30, 2)])
def test_device_with_known_inputs_matches_the_oracle(d, dy, ptt, C, T, H, monkeypatch):
    import rxhip
    rng = np.random.default_rng(17 * d + T)
    mdl = _models(rng, d, dy, 1)
    one = tuple(x[0] for x in mdl)
    cx, cy = rng.standard_normal((T + H, d)), rng.standard_normal((T + H, dy))
    y = _simulate(rng, mdl, cx[:T], cy[:T], C, ptt)
    if C % 64 == 0:
        monkeypatch.setenv("RXHIP_ONE_PASS", "1")
    with rxhip.LGSSMEngine(*one, T=T, n_chains=C, prior_through_transition=ptt, horizon=H, state_offset=cx, obs_offset=cy) as eng:
        eng.set_data(y, layout="chain_time")
        eng.run(2, True)
        mean, cov = eng.marginals(layout="chain_time")
        fe = eng.free_energy_per_chain()
        pm, pc = eng.predictions(layout="chain_time")
        jm = eng.node_marginals(layout="chain_time")[0] if d <= 4 else None
        eng.run_filter(False)
        fm, _ = eng.marginals(layout="chain_time")
    A, B, P, Q = one[:4]
    for c in sorted({0, C - 1}):
        yy = np.vstack([y[c], np.full((H, dy), np.nan)])
        om, oc, nll = rxo.lgssm_kalman_rts_affine(*one, yy, cx, cy, prior_through_transition=ptt)
        assert np.allclose(mean[c], om, rtol=1e-6, atol=1e-8) and np.allclose(cov[c], oc, rtol=1e-6, atol=1e-8)
        assert fe[c] == pytest.approx(nll, rel=1e-8)
        for t in (0, T // 2, T - 1):   # leave-one-out prediction of y[t]
            yl = yy.copy()
            yl[t] = np.nan
            lm, lc, _ = rxo.lgssm_kalman_rts_affine(*one, yl, cx, cy, prior_through_transition=ptt)
            assert np.allclose(pm[c, t], B @ lm[t] + cy[t], rtol=1e-6, atol=1e-7)
            assert np.allclose(pc[c, t], B @ lc[t] @ B.T + Q, rtol=1e-6, atol=1e-7)
        for t in range(T, T + H):      # forecasts carry the inputs of their time index
            assert np.allclose(pm[c, t], B @ om[t] + cy[t], rtol=1e-6, atol=1e-8)
        if jm is not None:
            for k in (0, T - 2):       # node-local joint of the transition into x[k+1]: mean [m(x[k+1]); A m(x[k]) + c[k+1]]
                assert np.allclose(jm[c, k], np.concatenate([om[k + 1], A @ om[k] + cx[k + 1]]), rtol=1e-6, atol=1e-8)
        qm, _, _ = rxo.lgssm_kalman_rts_affine(*one, y[c, :T // 2 + 1], cx[:T // 2 + 1], cy[:T // 2 + 1], prior_through_transition=ptt)
        assert np.allclose(fm[c, T // 2], qm[-1], rtol=1e-6, atol=1e-8)   # filtering: the smoother of the first half ends there


@pytest.mark.gpu
def test_known_inputs_with_missing_values_per_step_constants_and_device_data():
    import ctypes

    import rxhip
    rng = np.random.default_rng(4)
    d, dy, T, C = 2, 2, 30, 3
    mdl = _models(rng, d, dy, T)
    sm = np.arange(T, dtype=np.int32)
    cx, cy = rng.standard_normal((T, d)), rng.standard_normal((T, dy))
    y = rng.standard_normal((T, C, dy))
    y[[4, 9], 1] = np.nan
    hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")   # the runtime librxhip uses (torch, if loaded, carries its own copy)
    yd, back = ctypes.c_void_p(), np.empty_like(y)
    assert hip.hipMalloc(ctypes.byref(yd), ctypes.c_size_t(y.nbytes)) == 0
    assert hip.hipMemcpy(yd, y.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(y.nbytes), 1) == 0
    with rxhip.LGSSMEngine(*mdl, T=T, n_chains=C, step_model=sm, allow_missing=True, state_offset=cx, obs_offset=cy) as eng:
        eng.set_data_device(yd.value, y.size)   # the engine shifts its own copy: the caller's buffer stays as it is
        eng.run(1, True)
        mean, cov = eng.marginals(layout="chain_time")
        fe = eng.free_energy_per_chain()
    assert hip.hipMemcpy(back.ctypes.data_as(ctypes.c_void_p), yd, ctypes.c_size_t(y.nbytes), 2) == 0 and hip.hipFree(yd) == 0
    assert np.array_equal(back, y, equal_nan=True)
    for c in range(C):
        om, oc, nll = rxo.lgssm_kalman_rts_affine(*mdl, y[:, c], cx, cy, step_model=sm)
        assert np.allclose(mean[c], om, rtol=1e-6, atol=1e-8) and np.allclose(cov[c], oc, rtol=1e-6, atol=1e-8)
        assert fe[c] == pytest.approx(nll, rel=1e-8)


@pytest.mark.gpu
def test_infer_mirror_with_a_constant_drift():
    import rxhip
    rng = np.random.default_rng(6)
    A, B, P, Q = np.eye(2), np.eye(2), np.eye(2) * 0.1, np.eye(2)
    spec = rxhip.linear_gaussian_ssm(A, B, P, Q, np.zeros(2), np.eye(2) * 10, state_offset=[0.5, -0.25])
    y = np.cumsum(np.tile([0.5, -0.25], (80, 1)), axis=0) + rng.standard_normal((80, 2))
    res = rxhip.infer(model=spec, data={"y": y}, free_energy=True)
    om, oc, nll = rxo.lgssm_kalman_rts_affine(A, B, P, Q, np.zeros(2), np.eye(2) * 10, y, np.tile([0.5, -0.25], (80, 1)), None)
    assert np.allclose(res.posteriors["x"].mean, om, rtol=1e-6, atol=1e-8) and res.free_energy[-1] == pytest.approx(nll, rel=1e-8)


# ---------------------------------------------------------------------------------------------------------------- graphs
def test_plus_nodes_in_front_of_gaussian_means_lower_to_known_inputs():
    import rxhip  # noqa: F401
    from rxhip import _lib, graph
    rng = np.random.default_rng(2)
    d, dy, T = 3, 2, 7
    mdl = _models(rng, d, dy, 1)
    A, B, P, Q, m0, V0 = (x[0] for x in mdl)
    cx, cy = rng.standard_normal((T, d)), rng.standard_normal((T, dy))
    for ptt, const_first in ((False, False), (True, True)):
        gb, xs, ys = graph.lgssm_graph(T, A, B, P, Q, m0, V0, prior_through_transition=ptt, const_first=const_first,
                                       c_of_t=lambda t: cx[t] if t % 2 == 0 else None,   # every other transition has an input
                                       d_of_t=lambda t: cy[t])
        low = graph.lower_lgssm(gb.tables(permute=rng.permutation(len(gb.ftype)))[0])
        assert low["has_offsets"] and low["n_models"] == 1 and low["T"] == T
        for t in range(T):
            want = cx[t] if (t % 2 == 0 and (t > 0 or ptt)) else np.zeros(d)
            assert np.array_equal(low["state_offset"][t], want) and np.array_equal(low["obs_offset"][t], cy[t])
        assert np.array_equal(low["A"], A) and list(low["data_var"]) == ys
    # the scalar random walk with drift and state noise, no `*` node: x[t] ~ Normal(mean = x[t-1] + c, var = p)
    gb = graph.GraphBuilder()
    x = gb.randomvar(1)
    gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, x, gb.constvar(0.0), gb.constvar(4.0))
    for t in range(5):
        if t:
            w, xn = gb.randomvar(1), gb.randomvar(1)
            gb.node(_lib.NODE_ADD, w, x, gb.constvar(0.3))
            gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, xn, w, gb.constvar(0.1))
            x = xn
        gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, gb.datavar(1), x, gb.constvar(1.0))
    low = graph.lower_lgssm(gb.tables()[0])
    assert not low["deterministic"] and low["has_offsets"] and np.allclose(low["state_offset"][1:, 0], 0.3) and low["state_offset"][0, 0] == 0.0
    assert low["A"][0, 0] == 1.0 and low["P"][0, 0] == 0.1
    # …and the noise-free drift chain of ulgssm_tests.jl stays what it was
    gb, xs, ys = graph.drift_chain_graph(6, 0.0, 100.0, 1.0, 1.0)
    low = graph.lower_lgssm(gb.tables()[0])
    assert low["deterministic"] and not low["has_offsets"]


@pytest.mark.gpu
def test_graph_with_known_inputs_runs_on_the_device():
    import rxhip  # noqa: F401
    from rxhip import graph
    rng = np.random.default_rng(11)
    d, dy, T, C = 2, 2, 40, 3
    mdl = _models(rng, d, dy, 1)
    A, B, P, Q, m0, V0 = (x[0] for x in mdl)
    cx, cy = rng.standard_normal((T, d)), rng.standard_normal((T, dy))
    y = _simulate(rng, mdl, cx, cy, C, True)
    gb, xs, ys = graph.lgssm_graph(T, A, B, P, Q, m0, V0, prior_through_transition=True, c_of_t=lambda t: cx[t], d_of_t=lambda t: cy[t])
    g, keep = gb.tables(n_replicas=C, permute=rng.permutation(len(gb.ftype)))
    eng = graph.create_engine_from_graph(g)
    eng.set_data(y, layout="chain_time")
    eng.run(1, True)
    mean, cov = eng.marginals(layout="chain_time")
    fe = eng.free_energy_per_chain()
    eng.close()
    for c in range(C):
        om, oc, nll = rxo.lgssm_kalman_rts_affine(A, B, P, Q, m0, V0, y[c], cx, cy, prior_through_transition=True)
        assert np.allclose(mean[c], om, rtol=1e-6, atol=1e-8) and np.allclose(cov[c], oc, rtol=1e-6, atol=1e-8)
        assert fe[c] == pytest.approx(nll, rel=1e-8)


@pytest.mark.gpu
def test_new_inputs_for_a_live_engine():
    """A control loop re-plans its inputs: same engine, same tables, same observations, new c[t] / d[t] (rxhip_lgssm_set_offsets)."""
    import rxhip
    rng = np.random.default_rng(21)
    d, dy, T, C = 3, 2, 60, 5
    mdl = _models(rng, d, dy, 1)
    one = tuple(x[0] for x in mdl)
    y = rng.standard_normal((C, T, dy))
    with rxhip.LGSSMEngine(*one, T=T, n_chains=C, state_offset=np.zeros(d), obs_offset=np.zeros(dy)) as eng:
        eng.set_data(y, layout="chain_time")
        for trial in range(3):
            cx, cy = rng.standard_normal((T, d)), (rng.standard_normal((T, dy)) if trial != 1 else None)
            eng.set_offsets(cx, cy)
            eng.run(1, True)
            mean, cov = eng.marginals(layout="chain_time")
            fe = eng.free_energy_per_chain()
            for c in (0, C - 1):
                om, oc, nll = rxo.lgssm_kalman_rts_affine(*one, y[c], cx, cy)
                assert np.allclose(mean[c], om, rtol=1e-6, atol=1e-8) and fe[c] == pytest.approx(nll, rel=1e-8)
    with rxhip.LGSSMEngine(*one, T=T, n_chains=C) as eng, pytest.raises(rxhip.RxHipError):
        eng.set_offsets(np.zeros(d))


@pytest.mark.gpu
@pytest.mark.parametrize("d,dy,C,T,H,ptt,lay", [(2, 2, 5, 60, 0, False, "chain_time"), (4, 3, 64, 130, 4, True, "time_chain"), (8, 6, 4, 50, 3, False, "chain_time")])
def test_inputs_that_are_data_of_every_chain(d, dy, C, T, H, ptt, lay, monkeypatch):
    """`x[t] ~ MvNormal(μ = A*x[t-1] + B_u*u[t], Σ = P)` with u a datavar: every chain has its own inputs (rxhip_lgssm_set_chain_offsets)."""
    import rxhip
    if C % 64 == 0:
        monkeypatch.setenv("RXHIP_ONE_PASS", "1")
    rng = np.random.default_rng(d + T)
    mdl = _models(rng, d, dy, 1)
    one = tuple(x[0] for x in mdl)
    A, B, P, Q = one[:4]
    cx, cy = rng.standard_normal((C, T + H, d)), rng.standard_normal((C, T + H, dy))
    y = rng.standard_normal((C, T, dy))
    put = (lambda a: np.ascontiguousarray(np.transpose(a, (1, 0, 2)))) if lay == "time_chain" else (lambda a: a)
    with rxhip.LGSSMEngine(*one, T=T, n_chains=C, horizon=H, prior_through_transition=ptt, state_offset=np.zeros(d)) as eng:
        eng.set_data(y, layout="chain_time")
        eng.set_chain_offsets(put(cx), put(cy), layout=lay)
        eng.run(1, True)
        mean, cov = eng.marginals(layout="chain_time")
        fe = eng.free_energy_per_chain()
        pm, _ = eng.predictions(layout="chain_time")
        with pytest.raises(rxhip.RxHipError):
            eng.set_offsets(np.zeros(d))
        eng.set_chain_offsets(put(2 * cx), None, layout=lay)      # re-planned inputs: same engine, same observations
        eng.run(1, True)
        mean2, _ = eng.marginals(layout="chain_time")
        fe2 = eng.free_energy_per_chain()
    for c in sorted({0, C // 2, C - 1}):
        yy = np.vstack([y[c], np.full((H, dy), np.nan)])
        om, oc, nll = rxo.lgssm_kalman_rts_affine(*one, yy, cx[c], cy[c], prior_through_transition=ptt)
        assert np.allclose(mean[c], om, rtol=1e-6, atol=1e-8) and np.allclose(cov[c], oc, rtol=1e-6, atol=1e-8)
        assert fe[c] == pytest.approx(nll, rel=1e-8)
        for t in range(T, T + H):
            assert np.allclose(pm[c, t], B @ om[t] + cy[c, t], rtol=1e-6, atol=1e-8)
        yl = yy.copy(); yl[3] = np.nan
        lm, _, _ = rxo.lgssm_kalman_rts_affine(*one, yl, cx[c], cy[c], prior_through_transition=ptt)
        assert np.allclose(pm[c, 3], B @ lm[3] + cy[c, 3], rtol=1e-6, atol=1e-7)
        om2, _, nll2 = rxo.lgssm_kalman_rts_affine(*one, yy, 2 * cx[c], None, prior_through_transition=ptt)
        assert np.allclose(mean2[c], om2, rtol=1e-6, atol=1e-8) and fe2[c] == pytest.approx(nll2, rel=1e-8)


@pytest.mark.gpu
def test_infer_mirror_with_control_inputs_as_data():
    import rxhip
    rng = np.random.default_rng(33)
    A, B, P, Q = np.array([[1.0, 0.1], [0.0, 1.0]]), np.array([[1.0, 0.0]]), np.eye(2) * 0.01, np.eye(1) * 0.25
    Bu = np.array([[0.005], [0.1]])
    spec = rxhip.linear_gaussian_ssm(A, B, P, Q, np.zeros(2), np.eye(2), input_matrix=Bu)
    T = 70
    u = rng.standard_normal((T, 1))
    y = rng.standard_normal((T, 1))
    res = rxhip.infer(model=spec, data={"y": y, "u": u}, free_energy=True)
    om, oc, nll = rxo.lgssm_kalman_rts_affine(A, B, P, Q, np.zeros(2), np.eye(2), y, u @ Bu.T, None)
    assert np.allclose(res.posteriors["x"].mean, om, rtol=1e-6, atol=1e-8) and res.free_energy[-1] == pytest.approx(nll, rel=1e-8)


@pytest.mark.gpu
def test_control_inputs_as_data_keep_a_constant_observation_offset():
    """`input_matrix` (u as data) together with `obs_offset` — the per-chain inputs must not wipe the constant d[t]
    (NULL in rxhip_lgssm_set_chain_offsets keeps the offsets of creation).  Checked against the oracle with both offsets and
    against the same model spelt with the inputs folded into a constant `state_offset`."""
    import rxhip
    rng = np.random.default_rng(34)
    A, B, P, Q = np.array([[1.0, 0.1], [0.0, 1.0]]), np.array([[1.0, 0.0]]), np.eye(2) * 0.01, np.eye(1) * 0.25
    Bu = np.array([[0.005], [0.1]])
    T = 60
    u, y = rng.standard_normal((T, 1)), rng.standard_normal((T, 1))
    dofs = 1.5 + 0.1 * np.arange(T)[:, None]                      # d[t], (T, dy)
    spec = rxhip.linear_gaussian_ssm(A, B, P, Q, np.zeros(2), np.eye(2), input_matrix=Bu, obs_offset=dofs)
    res = rxhip.infer(model=spec, data={"y": y, "u": u}, free_energy=True)
    om, oc, nll = rxo.lgssm_kalman_rts_affine(A, B, P, Q, np.zeros(2), np.eye(2), y, u @ Bu.T, dofs)
    assert np.allclose(res.posteriors["x"].mean, om, rtol=1e-6, atol=1e-8)
    assert res.free_energy[-1] == pytest.approx(nll, rel=1e-8)
    folded = rxhip.linear_gaussian_ssm(A, B, P, Q, np.zeros(2), np.eye(2), state_offset=u @ Bu.T, obs_offset=dofs)
    ref = rxhip.infer(model=folded, data={"y": y}, free_energy=True)
    assert np.allclose(res.posteriors["x"].mean, ref.posteriors["x"].mean, rtol=1e-9, atol=1e-10)
    assert res.free_energy[-1] == pytest.approx(ref.free_energy[-1], rel=1e-10)
    # several chains through the same path
    yc, uc = rng.standard_normal((3, T, 1)), rng.standard_normal((3, T, 1))
    resc = rxhip.infer(model=spec, data={"y": yc, "u": uc}, free_energy=True)
    for c in range(3):
        om, _, nll = rxo.lgssm_kalman_rts_affine(A, B, P, Q, np.zeros(2), np.eye(2), yc[c], uc[c] @ Bu.T, dofs)
        assert np.allclose(resc.posteriors["x"].mean[c], om, rtol=1e-6, atol=1e-8)


def test_data_inputs_in_a_graph_are_recognised():
    """`x[t] ~ MvNormal(μ = A * x[t-1] + B_u * u[t], Σ = P)` with u[t] a data variable: `*`(B_u, u) feeding a `+` without a constant."""
    import rxhip  # noqa: F401
    from rxhip import graph
    rng = np.random.default_rng(8)
    d, dy, T, du = 3, 2, 6, 2
    mdl = _models(rng, d, dy, 1)
    A, B, P, Q, m0, V0 = (x[0] for x in mdl)
    Bu = rng.standard_normal((d, du))
    for ptt in (False, True):
        gb, xs, ys, us = graph.lgssm_graph(T, A, B, P, Q, m0, V0, prior_through_transition=ptt, Bu=Bu, du=du)
        low = graph.lower_lgssm(gb.tables(permute=rng.permutation(len(gb.ftype)))[0])
        assert low["du"] == du and np.array_equal(low["input_matrix"], Bu) and low["has_offsets"] and low["T"] == T
        want = [-1] * (0 if ptt else 1) + us   # no transition into the first state when the prior sits on x[1]
        assert list(low["input_var"]) == want and list(low["data_var"]) == ys
    gb, xs, ys, us = graph.lgssm_graph(T, A, B, P, Q, m0, V0, du=d)       # u[t] added directly: B_u = I
    low = graph.lower_lgssm(gb.tables()[0])
    assert low["du"] == d and np.array_equal(low["input_matrix"], np.eye(d))


@pytest.mark.gpu
def test_graph_engine_with_data_inputs():
    import rxhip
    from rxhip import graph
    rng = np.random.default_rng(13)
    d, dy, T, du, C = 2, 2, 45, 1, 3
    mdl = _models(rng, d, dy, 1)
    A, B, P, Q, m0, V0 = (x[0] for x in mdl)
    Bu = rng.standard_normal((d, du))
    cy = rng.standard_normal((T, dy))
    gb, xs, ys, us = graph.lgssm_graph(T, A, B, P, Q, m0, V0, prior_through_transition=True, Bu=Bu, du=du, d_of_t=lambda t: cy[t])
    eng = graph.create_engine_from_graph(gb.tables(n_replicas=C)[0])
    y, u = rng.standard_normal((C, T, dy)), rng.standard_normal((C, T, du))
    eng.set_data(y, layout="chain_time")
    # a datavar without a value: the reference refuses to run (batch.jl:387-407), and so does the engine
    with pytest.raises(rxhip.RxHipError) as ei:
        eng.run(1, True)
    assert ei.value.status == rxhip._lib.ERR_STATE and "RXHIP_VAR_U" in str(ei.value)
    eng.set_inputs(u, layout="chain_time")
    eng.run(1, True)
    mean, cov = eng.marginals(layout="chain_time")
    fe = eng.free_energy_per_chain()
    eng.close()
    for c in range(C):
        om, oc, nll = rxo.lgssm_kalman_rts_affine(A, B, P, Q, m0, V0, y[c], u[c] @ Bu.T, cy, prior_through_transition=True)
        assert np.allclose(mean[c], om, rtol=1e-6, atol=1e-8) and np.allclose(cov[c], oc, rtol=1e-6, atol=1e-8)
        assert fe[c] == pytest.approx(nll, rel=1e-8)
