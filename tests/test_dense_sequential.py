"""`missing` observations anywhere in the data and per-step constants A[t], P[t], B[t], Q[t] at ANY state dimension (d, dy ≤ 64):
the sequential schedule of csrc/gseq_kernels.hpp.

CPU: the kernel source itself, compiled as plain C++ with a one-thread workgroup (tests/emu/gseq_emu.cpp), against the oracle's
smoother with missing rows and per-step models — index arithmetic, LDS layout and formulas without a GPU.
GPU: the same cases through the C ABI (rxhip_lgssm_desc.allow_missing / step_model / chain_model), plus predictions, the forecast
horizon, filtering runs and known inputs on that schedule."""
import ctypes
import os
import subprocess
import tempfile

import numpy as np
import pytest

from oracle import rxoracle as rxo

HERE = os.path.dirname(os.path.abspath(__file__))


def _models(rng, d, dy, M):
    A = np.stack([0.9 * np.linalg.qr(rng.standard_normal((d, d)))[0] for _ in range(M)])
    B = rng.standard_normal((M, dy, d)) / np.sqrt(d)
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
        x = m0[0] + np.linalg.cholesky(V0[0]) @ rng.standard_normal(d)
        for t in range(T):
            if t or ptt:
                x = A[sm[t]] @ x + np.linalg.cholesky(P[sm[t]]) @ rng.standard_normal(d)
            y[c, t] = B[sm[t]] @ x + np.linalg.cholesky(Q[sm[t]]) @ rng.standard_normal(dy)
    return y


def _punch(rng, y, frac=0.25):
    """whole observations and single entries go missing (any NaN entry makes y[t] missing)"""
    y = y.copy()
    C, T, dy = y.shape
    hit = rng.random((C, T)) < frac
    y[hit] = np.nan
    y[0, 1, dy - 1] = np.nan
    return y


# ------------------------------------------------------------------------------------------------ host emulation (CPU)
@pytest.fixture(scope="module")
def emu():
    out = os.path.join(tempfile.mkdtemp(prefix="gseq_emu_"), "gseq_emu.so")
    subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-o", out, os.path.join(HERE, "emu", "gseq_emu.cpp")], check=True)
    lib = ctypes.CDLL(out)
    lib.gseq_emu_run.restype = ctypes.c_int
    return lib


def _emu_run(lib, mdl, y, ptt, chain_model=None, step_model=None, smooth=True):
    A, B, P, Q, m0, V0 = (np.ascontiguousarray(x, dtype=np.float64) for x in mdl)
    M, d, dy = A.shape[0], A.shape[-1], B.shape[-2]
    C, T, _ = y.shape
    user = np.concatenate([np.concatenate([A[m].ravel(), P[m].ravel(), B[m].ravel(), Q[m].ravel(), np.linalg.inv(Q[m]).ravel()]) for m in range(M)])
    prior = np.concatenate([np.concatenate([m0[m].ravel(), V0[m].ravel()]) for m in range(M)])
    yt = np.ascontiguousarray(np.transpose(y, (1, 0, 2)))   # [T][chain][dy]
    mean, cov, logev = np.empty((T, C, d)), np.empty((T, C, d, d)), np.zeros(C)
    dp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    ip = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int32).ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    cm = None if chain_model is None else np.ascontiguousarray(chain_model, dtype=np.int32)
    sm = None if step_model is None else np.ascontiguousarray(step_model, dtype=np.int32)
    st = lib.gseq_emu_run(ctypes.c_longlong(T), ctypes.c_longlong(C), d, dy, int(ptt), M, dp(user), dp(prior), ip(cm), ip(sm), dp(yt),
                          int(smooth), dp(mean), dp(cov), dp(logev))
    assert st == 0
    return np.transpose(mean, (1, 0, 2)), np.transpose(cov, (1, 0, 2, 3)), -logev


@pytest.mark.parametrize("d,dy,ptt", [(5, 3, False), (7, 7, True), (3, 9, False), (16, 4, True), (13, 17, False)])
def test_kernel_source_on_the_host_matches_the_oracle(emu, d, dy, ptt):
    rng = np.random.default_rng(100 * d + dy)
    T, C, M = 14, 2, 5
    mdl = _models(rng, d, dy, M)
    sm = (rng.permutation(T) % M).astype(np.int32)
    y = _punch(rng, _simulate(rng, mdl, sm, C, ptt))
    mean, cov, nll = _emu_run(emu, mdl, y, ptt, step_model=sm)
    fmean, fcov, _ = _emu_run(emu, mdl, y, ptt, step_model=sm, smooth=False)
    for c in range(C):
        om, oc, onll = rxo.lgssm_kalman_rts_tv(*mdl, sm, y[c], prior_through_transition=ptt)
        assert np.allclose(mean[c], om, rtol=1e-9, atol=1e-11) and np.allclose(cov[c], oc, rtol=1e-9, atol=1e-11)
        assert nll[c] == pytest.approx(onll, rel=1e-11)
        for t in (0, 6, T - 1):  # filtering: the smoother of the first t+1 observations ends in the filtered belief of t
            qm, qc, _ = rxo.lgssm_kalman_rts_tv(*mdl, sm[:t + 1], y[c, :t + 1], prior_through_transition=ptt)
            assert np.allclose(fmean[c, t], qm[-1], rtol=1e-9, atol=1e-11) and np.allclose(fcov[c, t], qc[-1], rtol=1e-9, atol=1e-11)


def test_kernel_source_on_the_host_with_one_model_per_chain(emu):
    rng = np.random.default_rng(8)
    d, dy, T, C = 6, 2, 11, 3
    mdl = _models(rng, d, dy, C)
    y = np.concatenate([_simulate(rng, tuple(x[c:c + 1] for x in mdl), np.zeros(T, dtype=np.int32), 1, False) for c in range(C)])
    y = _punch(rng, y)
    mean, cov, nll = _emu_run(emu, mdl, y, False, chain_model=np.arange(C))
    for c in range(C):
        one = tuple(x[c] for x in mdl)
        om, oc, onll = rxo.lgssm_kalman_rts(*one, y[c])
        assert np.allclose(mean[c], om, rtol=1e-9, atol=1e-11) and np.allclose(cov[c], oc, rtol=1e-9, atol=1e-11)
        assert nll[c] == pytest.approx(onll, rel=1e-11)


@pytest.mark.parametrize("T,ptt", [(1, False), (1, True), (5, True)])
def test_kernel_source_on_the_host_edge_cases(emu, T, ptt):
    """A single time index, and a chain that never observes anything (the posterior is the prior pushed through the
    transitions, the evidence is 1)."""
    rng = np.random.default_rng(40 + T)
    d, dy, C = 6, 4, 2
    mdl = _models(rng, d, dy, 1)
    one = tuple(x[0] for x in mdl)
    y = _simulate(rng, mdl, np.zeros(T, dtype=np.int32), C, ptt)
    y[1] = np.nan                                       # chain 1 observes nothing
    mean, cov, nll = _emu_run(emu, mdl, y, ptt)
    for c in range(C):
        om, oc, onll = rxo.lgssm_kalman_rts(*one, y[c], prior_through_transition=ptt)
        assert np.allclose(mean[c], om, rtol=1e-10, atol=1e-12) and np.allclose(cov[c], oc, rtol=1e-10, atol=1e-12)
        assert nll[c] == pytest.approx(onll, rel=1e-11, abs=1e-13)
    assert nll[1] == 0.0
    m, V = one[4], one[5]
    for t in range(T):
        if t or ptt:
            m, V = one[0] @ m, one[0] @ V @ one[0].T + one[2]
        assert np.allclose(mean[1, t], m, rtol=1e-12) and np.allclose(cov[1, t], V, rtol=1e-12)


@pytest.mark.parametrize("d,dy,ptt", [(5, 5, False), (9, 9, True), (7, 7, False), (16, 16, True)])   # the oracle's reference schedule needs a square, full-rank B
def test_joint_kernel_source_on_the_host_matches_the_oracle(emu, d, dy, ptt):
    """rxhip_get_node_marginals at d > 4: the backward kernel keeps Cov(x[t], x[t+1] | y), k_joint_generic assembles q(out, μ) of
    every transition node — against the joints the oracle forms inside its Bethe sum."""
    rng = np.random.default_rng(7 * d + dy)
    T, C = 9, 2
    mdl = _models(rng, d, dy, 1)
    one = tuple(x[0] for x in mdl)
    y = _simulate(rng, mdl, np.zeros(T, dtype=np.int32), C, ptt)
    A, B, P, Q, m0, V0 = (np.ascontiguousarray(x, dtype=np.float64) for x in mdl)
    user = np.concatenate([A[0].ravel(), P[0].ravel(), B[0].ravel(), Q[0].ravel(), np.linalg.inv(Q[0]).ravel()])
    prior = np.concatenate([m0[0].ravel(), V0[0].ravel()])
    yt = np.ascontiguousarray(np.transpose(y, (1, 0, 2)))
    jm, jc = np.empty((T - 1, C, 2 * d)), np.empty((T - 1, C, 2 * d, 2 * d))
    dp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    st = emu.gseq_emu_joints(ctypes.c_longlong(T), ctypes.c_longlong(C), d, dy, int(ptt), dp(user), dp(prior), None, dp(yt), dp(jm), dp(jc))
    assert st == 0
    for c in range(C):
        om, oc = rxo.lgssm_joints(*one, y[c], prior_through_transition=ptt)
        assert np.allclose(jm[:, c], om, rtol=1e-9, atol=1e-11) and np.allclose(jc[:, c], oc, rtol=1e-8, atol=1e-11)


def test_stream_step_source_on_the_host_matches_the_filtering_oracle(emu):
    """k_gseq_stream_step (rxhip_filter_step at d > 4) one observation at a time, with per-step constants, known inputs and
    missing observations, against the oracle's smoother of the observations seen so far (its last belief is the filtered one)."""
    rng = np.random.default_rng(19)
    d, dy, T, C, M = 6, 3, 12, 2, 4
    mdl = _models(rng, d, dy, M)
    sm = (rng.permutation(T) % M).astype(np.int32)
    cx, cy = rng.standard_normal((T, d)), rng.standard_normal((T, dy))
    y = _punch(rng, _simulate(rng, mdl, sm, C, True) + cy[None])
    A, B, P, Q, m0, V0 = (np.ascontiguousarray(x, dtype=np.float64) for x in mdl)
    user = np.concatenate([np.concatenate([A[m].ravel(), P[m].ravel(), B[m].ravel(), Q[m].ravel(), np.linalg.inv(Q[m]).ravel()]) for m in range(M)])
    prior = np.concatenate([np.concatenate([m0[m].ravel(), V0[m].ravel()]) for m in range(M)])
    yt = np.ascontiguousarray(np.transpose(y, (1, 0, 2)))
    mean, cov, fe = np.empty((T, C, d)), np.empty((T, C, d, d)), np.empty((T, C))
    dp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    st = emu.gseq_emu_stream(ctypes.c_longlong(T), ctypes.c_longlong(C), d, dy, 1, dp(user), dp(prior), None,
                             sm.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), dp(cx), dp(cy), dp(yt), dp(mean), dp(cov), dp(fe))
    assert st == 0
    for c in range(C):
        prev = 0.0
        for t in range(T):
            qm, qc, nll = rxo.lgssm_kalman_rts_affine(*mdl, y[c, :t + 1], state_offset=cx[:t + 1], obs_offset=cy[:t + 1], step_model=sm[:t + 1],
                                                      prior_through_transition=True)
            assert np.allclose(mean[t, c], qm[-1], rtol=1e-9, atol=1e-11) and np.allclose(cov[t, c], qc[-1], rtol=1e-9, atol=1e-11)
            assert fe[t, c] == pytest.approx(nll - prev, rel=1e-9, abs=1e-11)
            prev = nll


# ---------------------------------------------------------------------------------------------------------------- device
@pytest.mark.gpu
@pytest.mark.parametrize("d,dy,ptt,C", [(5, 3, False, 3), (8, 8, True, 20), (16, 4, False, 5), (33, 7, True, 2), (64, 64, False, 2), (6, 40, True, 3)])
def test_missing_observations_at_any_dimension(d, dy, ptt, C):
    import rxhip
    rng = np.random.default_rng(3 * d + dy)
    T = 25 if d < 64 else 12
    mdl = _models(rng, d, dy, 1)
    one = tuple(x[0] for x in mdl)
    sm = np.zeros(T, dtype=np.int32)
    y = _punch(rng, _simulate(rng, mdl, sm, C, ptt))
    with rxhip.LGSSMEngine(*one, T=T, n_chains=C, prior_through_transition=ptt, allow_missing=True) as eng:
        eng.set_data(y, layout="chain_time")
        eng.run(free_energy=True)
        mean, cov = eng.marginals(layout="chain_time")
        fe, total = eng.free_energy_per_chain(), eng.free_energy()[-1]
        pm, pc = eng.predictions(layout="chain_time")
        eng.run_filter(free_energy=False)
        fm, fc = eng.marginals(layout="chain_time")
    nlls = []
    for c in range(C):
        om, oc, nll = rxo.lgssm_kalman_rts(*one, y[c], prior_through_transition=ptt)
        nlls.append(nll)
        assert np.allclose(mean[c], om, rtol=1e-6, atol=1e-9) and np.allclose(cov[c], oc, rtol=1e-6, atol=1e-9)
        assert fe[c] == pytest.approx(nll, rel=1e-8, abs=1e-9)
        for t in (0, T // 2, T - 1):  # prediction of y[t]: the smoother without y[t], pushed through B, Q
            yl = y[c].copy()
            yl[t] = np.nan
            lm, lc, _ = rxo.lgssm_kalman_rts(*one, yl, prior_through_transition=ptt)
            assert np.allclose(pm[c, t], one[1] @ lm[t], rtol=1e-6, atol=1e-8)
            assert np.allclose(pc[c, t], one[1] @ lc[t] @ one[1].T + one[3], rtol=1e-6, atol=1e-8)
        for t in (0, 5, T - 1):
            qm, qc, _ = rxo.lgssm_kalman_rts(*one, y[c, :t + 1], prior_through_transition=ptt)
            assert np.allclose(fm[c, t], qm[-1], rtol=1e-6, atol=1e-9) and np.allclose(fc[c, t], qc[-1], rtol=1e-6, atol=1e-9)
    assert total == pytest.approx(sum(nlls), rel=1e-8)


@pytest.mark.gpu
@pytest.mark.parametrize("d,dy,ptt,M", [(5, 2, True, 3), (12, 12, False, 30), (32, 5, True, 4)])
def test_per_step_constants_at_any_dimension(d, dy, ptt, M):
    import rxhip
    rng = np.random.default_rng(17 * d + M)
    C, T, H = 4, 30, 3
    mdl = _models(rng, d, dy, M)
    sm = (rng.permutation(T + H) % M).astype(np.int32)
    y = _simulate(rng, mdl, sm, C, ptt)
    y[:, T:] = np.nan
    y[2, [3, 4, 17]] = np.nan
    with rxhip.LGSSMEngine(*mdl, T=T, n_chains=C, prior_through_transition=ptt, step_model=sm, horizon=H, allow_missing=True) as eng:
        eng.set_data(y[:, :T], layout="chain_time")
        eng.run(free_energy=True)
        mean, cov = eng.marginals(layout="chain_time")
        pm, pc = eng.predictions(layout="chain_time")
        fe = eng.free_energy_per_chain()
    assert mean.shape == (C, T + H, d) and pm.shape == (C, T + H, dy)
    for c in range(C):
        om, oc, nll = rxo.lgssm_kalman_rts_tv(*mdl, sm, y[c], prior_through_transition=ptt)
        assert np.allclose(mean[c], om, rtol=1e-6, atol=1e-9) and np.allclose(cov[c], oc, rtol=1e-6, atol=1e-9)
        assert fe[c] == pytest.approx(nll, rel=1e-8)
        for t in range(T, T + H):
            Bt, Qt = mdl[1][sm[t]], mdl[3][sm[t]]
            assert np.allclose(pm[c, t], Bt @ om[t], rtol=1e-6, atol=1e-9)
            assert np.allclose(pc[c, t], Bt @ oc[t] @ Bt.T + Qt, rtol=1e-6, atol=1e-9)


@pytest.mark.gpu
def test_one_model_per_chain_with_missing_observations_at_d_9():
    import rxhip
    rng = np.random.default_rng(21)
    d, dy, T, C = 9, 3, 18, 5
    mdl = _models(rng, d, dy, C)
    y = np.concatenate([_simulate(rng, tuple(x[c:c + 1] for x in mdl), np.zeros(T, dtype=np.int32), 1, False) for c in range(C)])
    y = _punch(rng, y)
    with rxhip.LGSSMEngine(*mdl, T=T, n_chains=C, chain_model=np.arange(C, dtype=np.int32), allow_missing=True) as eng:
        eng.set_data(y, layout="chain_time")
        eng.run(free_energy=True)
        mean, cov = eng.marginals(layout="chain_time")
        fe = eng.free_energy_per_chain()
    for c in range(C):
        om, oc, nll = rxo.lgssm_kalman_rts(*(x[c] for x in mdl), y[c])
        assert np.allclose(mean[c], om, rtol=1e-6, atol=1e-9) and np.allclose(cov[c], oc, rtol=1e-6, atol=1e-9)
        assert fe[c] == pytest.approx(nll, rel=1e-8)


@pytest.mark.gpu
def test_known_inputs_with_missing_observations_at_d_6():
    import rxhip
    rng = np.random.default_rng(4)
    d, dy, T, C = 6, 3, 20, 3
    mdl = _models(rng, d, dy, 1)
    one = tuple(x[0] for x in mdl)
    cx, cy = rng.standard_normal((T, d)), rng.standard_normal((T, dy))
    y = _punch(rng, _simulate(rng, mdl, np.zeros(T, dtype=np.int32), C, False) + cy[None])
    with rxhip.LGSSMEngine(*one, T=T, n_chains=C, allow_missing=True, state_offset=cx, obs_offset=cy) as eng:
        eng.set_data(y, layout="chain_time")
        eng.run(free_energy=True)
        mean, cov = eng.marginals(layout="chain_time")
        fe = eng.free_energy_per_chain()
    for c in range(C):
        om, oc, nll = rxo.lgssm_kalman_rts_affine(*one, y[c], state_offset=cx, obs_offset=cy)
        assert np.allclose(mean[c], om, rtol=1e-6, atol=1e-9) and np.allclose(cov[c], oc, rtol=1e-6, atol=1e-9)
        assert fe[c] == pytest.approx(nll, rel=1e-8)


@pytest.mark.gpu
def test_infer_mirror_routes_interior_missing_values_at_d_10():
    import rxhip
    rng = np.random.default_rng(12)
    d, dy, T = 10, 4, 40
    mdl = _models(rng, d, dy, 1)
    one = tuple(x[0] for x in mdl)
    y = _simulate(rng, mdl, np.zeros(T, dtype=np.int32), 1, False)[0]
    y[[5, 6, 30]] = np.nan
    res = rxhip.infer(model=rxhip.linear_gaussian_ssm(*one), data={"y": y}, free_energy=True)
    om, oc, nll = rxo.lgssm_kalman_rts(*one, y)
    assert np.allclose(res.posteriors["x"].mean, om, rtol=1e-6, atol=1e-9) and np.allclose(res.posteriors["x"].cov, oc, rtol=1e-6, atol=1e-9)
    assert res.free_energy[-1] == pytest.approx(nll, rel=1e-8)


@pytest.mark.gpu
@pytest.mark.parametrize("d,dy,C,masked", [(6, 3, 4, False), (16, 16, 3, True), (64, 8, 2, False)])
def test_filter_step_at_any_dimension(d, dy, C, masked):
    """rxhip_filter_step on the MFMA-path engines (time-parallel or sequential schedule): one observation at a time ≡ the
    filtering run of the whole series ≡ the oracle."""
    import rxhip
    rng = np.random.default_rng(d + dy)
    T = 15
    mdl = _models(rng, d, dy, 1)
    one = tuple(x[0] for x in mdl)
    y = _simulate(rng, mdl, np.zeros(T, dtype=np.int32), C, True)
    if masked:
        y = _punch(rng, y)
    with rxhip.LGSSMEngine(*one, T=T, n_chains=C, prior_through_transition=True, allow_missing=masked) as eng:
        eng.set_data(y, layout="chain_time")
        eng.run_filter(free_energy=False)
        fm, fc = eng.marginals(layout="chain_time")
        for rep in range(2):
            eng.filter_reset()
            nll = np.zeros(C)
            for t in range(T):
                m, v, f = eng.filter_step(y[:, t])
                assert np.allclose(m, fm[:, t], rtol=1e-8, atol=1e-10) and np.allclose(v, fc[:, t], rtol=1e-8, atol=1e-10)
                nll += f
    for c in range(C):
        _, _, onll = rxo.lgssm_kalman_rts(*one, y[c], prior_through_transition=True)
        assert nll[c] == pytest.approx(onll, rel=1e-8)


@pytest.mark.gpu
@pytest.mark.parametrize("d,ptt,C,T,masked", [(5, False, 3, 20, False), (8, True, 16, 30, False), (16, False, 4, 25, True), (64, True, 2, 10, False)])
def test_node_local_joints_at_any_dimension(d, ptt, C, T, masked):
    """rxhip_get_node_marginals on the MFMA-path engines (time-parallel, shared-model split, sequential): q(x[t+1], A x[t]) of every
    transition node against the joints the oracle forms inside its Bethe sum (square, full-rank B: what its reference schedule needs),
    and — with missing observations — against brute-force conditioning of the whole chain."""
    import rxhip
    from test_node_marginals import _full_posterior
    rng = np.random.default_rng(5 * d + T)
    mdl = _models(rng, d, d, 1)
    one = tuple(x[0] for x in mdl)
    y = _simulate(rng, mdl, np.zeros(T, dtype=np.int32), C, ptt)
    with rxhip.LGSSMEngine(*one, T=T, n_chains=C, prior_through_transition=ptt, allow_missing=masked) as eng:
        eng.set_data(y, layout="chain_time")
        eng.run(free_energy=True)
        mean, cov = eng.marginals(layout="chain_time")
        jm, jc = eng.node_marginals(layout="chain_time")
        mean2, cov2 = eng.marginals(layout="chain_time")
    assert np.array_equal(mean, mean2) and np.array_equal(cov, cov2)     # the getter leaves the posteriors alone
    assert jm.shape == (C, T - 1, 2 * d) and jc.shape == (C, T - 1, 2 * d, 2 * d)
    for c in range(C):
        om, oc = rxo.lgssm_joints(*one, y[c], prior_through_transition=ptt)
        assert np.allclose(jm[c], om, rtol=1e-6, atol=1e-9) and np.allclose(jc[c], oc, rtol=1e-6, atol=1e-9)
        assert np.allclose(jc[c][:, :d, :d], cov[c, 1:], rtol=1e-9, atol=1e-12)   # the (out, out) block is the posterior of x[t+1]
    if d <= 8:   # independent of the oracle's schedule: blocks of the full posterior of the chain
        pm, pV = _full_posterior(mdl, y[0], ptt)
        A = one[0]
        for k in (0, T // 2, T - 2):
            X = pV[k * d:(k + 1) * d, (k + 1) * d:(k + 2) * d]
            assert np.allclose(jc[0, k][d:, :d], A @ X, rtol=1e-6, atol=1e-9)
