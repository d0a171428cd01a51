"""Node-local joint marginals of the transition nodes (`@marginalrule MvNormalMeanCovariance(:out_μ)`): the oracle's
joints — formed inside its Bethe sum exactly as the reference's rule does — against brute-force conditioning of the joint
Gaussian of the whole chain (CPU), and rxhip_get_node_marginals against the oracle (GPU)."""
import numpy as np
import pytest

from oracle import rxoracle as rxo
from test_time_varying import _models, _simulate


def _full_posterior(mdl, y, ptt):
    """(mean, covariance) of (x_1 … x_T | y) by brute force."""
    A, B, P, Q, m0, V0 = mdl
    d, dy, T = A.shape[-1], B.shape[-2], y.shape[0]
    sm = np.zeros(T, dtype=int)
    mx, Vx = np.zeros((T, d)), np.zeros((T, d, T, d))
    if ptt:
        mx[0], Vx[0, :, 0, :] = A[0] @ m0[0], A[0] @ V0[0] @ A[0].T + P[0]
    else:
        mx[0], Vx[0, :, 0, :] = m0[0], V0[0]
    for t in range(1, T):
        mx[t] = A[0] @ mx[t - 1]
        Vx[t, :, t, :] = A[0] @ Vx[t - 1, :, t - 1, :] @ A[0].T + P[0]
        for s in range(t):
            Vx[t, :, s, :] = A[0] @ Vx[t - 1, :, s, :]
            Vx[s, :, t, :] = Vx[t, :, s, :].T
    Vx = Vx.reshape(T * d, T * d)
    Bb, Qb = np.kron(np.eye(T), B[0]), np.kron(np.eye(T), Q[0])
    Syy = Bb @ Vx @ Bb.T + Qb
    K = np.linalg.solve(Syy, Bb @ Vx).T
    return (mx.ravel() + K @ (y.ravel() - Bb @ mx.ravel())).reshape(T, d), Vx - K @ Bb @ Vx


@pytest.mark.parametrize("d,ptt", [(1, False), (2, True), (3, False), (4, True)])
def test_oracle_joints_are_blocks_of_the_full_posterior(d, ptt):
    rng = np.random.default_rng(40 + d)
    mdl = _models(rng, d, d, 1)
    T = 8
    y = _simulate(rng, mdl, np.zeros(T, dtype=int), 1, ptt)[0]
    A = mdl[0][0]
    jm, jc = rxo.lgssm_joints(*(x[0] for x in mdl), y, prior_through_transition=ptt)
    pm, pV = _full_posterior(mdl, y, ptt)
    assert jm.shape == (T - 1, 2 * d)
    for k in range(T - 1):
        V0 = pV[k * d:(k + 1) * d, k * d:(k + 1) * d]
        V1 = pV[(k + 1) * d:(k + 2) * d, (k + 1) * d:(k + 2) * d]
        X = pV[k * d:(k + 1) * d, (k + 1) * d:(k + 2) * d]          # Cov(x[k], x[k+1])
        ref = np.block([[V1, (A @ X).T], [A @ X, A @ V0 @ A.T]])
        assert np.allclose(jm[k], np.concatenate([pm[k + 1], A @ pm[k]]), rtol=1e-8, atol=1e-10)
        assert np.allclose(jc[k], ref, rtol=1e-7, atol=1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("d,dy,ptt,C,T,one_pass", [(1, 1, True, 3, 40, "0"), (2, 2, False, 64, 70, "1"), (3, 3, True, 5, 33, "1"),
                                                    (4, 4, False, 128, 50, "1"), (4, 4, True, 2, 300, "0"), (2, 2, True, 1, 2, "0")])
def test_device_joints_match_the_oracle(d, dy, ptt, C, T, one_pass, monkeypatch):
    import rxhip
    monkeypatch.setenv("RXHIP_ONE_PASS", one_pass)
    rng = np.random.default_rng(d * 100 + T)
    mdl = _models(rng, d, dy, 1)
    one = tuple(x[0] for x in mdl)
    y = _simulate(rng, mdl, np.zeros(T, dtype=int), C, ptt)
    with rxhip.LGSSMEngine(*one, T=T, n_chains=C, prior_through_transition=ptt) as eng:
        eng.set_data(y, layout="chain_time")
        eng.run(free_energy=True)
        jm, jc = eng.node_marginals(layout="chain_time")
    assert jm.shape == (C, T - 1, 2 * d) and jc.shape == (C, T - 1, 2 * d, 2 * d)
    for c in ([0] if C == 1 else [0, C // 2, C - 1]):
        om, oc = rxo.lgssm_joints(*one, y[c], prior_through_transition=ptt)
        assert np.allclose(jm[c], om, rtol=1e-6, atol=1e-9)
        assert np.allclose(jc[c], oc, rtol=1e-6, atol=1e-9)


@pytest.mark.gpu
def test_joints_with_per_chain_models_missing_values_and_per_step_constants():
    import rxhip
    rng = np.random.default_rng(9)
    d, dy, T, C = 2, 1, 12, 3
    mdl = _models(rng, d, dy, T)
    sm = np.arange(T, dtype=np.int32)
    y = _simulate(rng, mdl, sm, C, False)
    y[1, [3, 4]] = np.nan
    with rxhip.LGSSMEngine(*mdl, T=T, n_chains=C, step_model=sm, allow_missing=True) as eng:
        eng.set_data(y, layout="chain_time")
        eng.run(free_energy=False)
        jm, jc = eng.node_marginals(layout="chain_time")
    A = mdl[0]
    for c in range(C):
        full_m, full_V = _cross(mdl, sm, y[c])   # brute force, keeping the cross blocks
        for k in range(T - 1):
            Ak = A[sm[k + 1]]
            X = full_V[k * d:(k + 1) * d, (k + 1) * d:(k + 2) * d]
            V0 = full_V[k * d:(k + 1) * d, k * d:(k + 1) * d]
            V1 = full_V[(k + 1) * d:(k + 2) * d, (k + 1) * d:(k + 2) * d]
            ref = np.block([[V1, (Ak @ X).T], [Ak @ X, Ak @ V0 @ Ak.T]])
            assert np.allclose(jm[c, k], np.concatenate([full_m[k + 1], Ak @ full_m[k]]), rtol=1e-6, atol=1e-9)
            assert np.allclose(jc[c, k], ref, rtol=1e-6, atol=1e-9)


def _cross(mdl, sm, y):
    """Full posterior (mean [T,d], covariance [Td,Td]) with per-step constants and missing rows, by brute force."""
    A, B, P, Q, m0, V0 = mdl
    d, dy, T = A.shape[-1], B.shape[-2], len(sm)
    mx, Vx = np.zeros((T, d)), np.zeros((T, d, T, d))
    mx[0], Vx[0, :, 0, :] = m0[0], V0[0]
    for t in range(1, T):
        At = A[sm[t]]
        mx[t] = At @ mx[t - 1]
        Vx[t, :, t, :] = At @ Vx[t - 1, :, t - 1, :] @ At.T + P[sm[t]]
        for s in range(t):
            Vx[t, :, s, :] = At @ Vx[t - 1, :, s, :]
            Vx[s, :, t, :] = Vx[t, :, s, :].T
    Vx = Vx.reshape(T * d, T * d)
    Bb, Qb = np.zeros((T * dy, T * d)), np.zeros((T * dy, T * dy))
    for t in range(T):
        Bb[t * dy:(t + 1) * dy, t * d:(t + 1) * d] = B[sm[t]]
        Qb[t * dy:(t + 1) * dy, t * dy:(t + 1) * dy] = Q[sm[t]]
    keep = np.flatnonzero(~np.isnan(y).any(axis=1))
    idx = (keep[:, None] * dy + np.arange(dy)).ravel()
    Syy = (Bb @ Vx @ Bb.T + Qb)[np.ix_(idx, idx)]
    Vxy = (Vx @ Bb.T)[:, idx]
    K = np.linalg.solve(Syy, Vxy.T).T
    return (mx.ravel() + K @ (y[keep].ravel() - (Bb @ mx.ravel())[idx])).reshape(T, d), Vx - K @ Vxy.T
