"""ctypes binding of the CPU oracle (oracle/librxoracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product (rxinfer.jl_amd) never imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Counters(ctypes.Structure):
    _fields_ = [("rule_calls", ctypes.c_uint64), ("products", ctypes.c_uint64), ("marginals", ctypes.c_uint64)]


def build(force=False):
    so = os.path.join(_HERE, "librxoracle.so")
    src = [os.path.join(_HERE, f) for f in ("rxoracle.c", "rxoracle.h", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "librxoracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "librxoracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        dp = ctypes.POINTER(ctypes.c_double)
        _LIB.rxo_lgssm_bp.restype = ctypes.c_int
        _LIB.rxo_lgssm_bp.argtypes = [ctypes.c_int] * 3 + [dp] * 6 + [ctypes.c_int, dp, dp, dp, dp,
                                                                    ctypes.POINTER(Counters)]
        _LIB.rxo_lgssm_kalman_rts.restype = ctypes.c_int
        _LIB.rxo_lgssm_kalman_rts.argtypes = [ctypes.c_int] * 3 + [dp] * 6 + [ctypes.c_int, dp, dp, dp, dp]
        _LIB.rxo_lgssm_kalman_rts_tv.restype = ctypes.c_int
        _LIB.rxo_lgssm_kalman_rts_tv.argtypes = ([ctypes.c_int] * 4 + [dp] * 6 +
                                                 [ctypes.POINTER(ctypes.c_int), ctypes.c_int, dp, dp, dp, dp])
        _LIB.rxo_lgssm_bp_batch.restype = ctypes.c_int
        _LIB.rxo_lgssm_bp_batch.argtypes = [ctypes.c_int] * 4 + [dp] * 6 + [ctypes.c_int, dp, dp, dp, dp,
                                                                          ctypes.c_int, ctypes.POINTER(Counters)]
        _LIB.rxo_gmm_vmp.restype = ctypes.c_int
        _LIB.rxo_gmm_vmp.argtypes = [ctypes.c_longlong, ctypes.c_int] + [dp] * 11 + [ctypes.c_int, dp, dp, dp,
                                                                                        ctypes.POINTER(Counters)]
        _LIB.rxo_hgf_filter.restype = ctypes.c_int
        _LIB.rxo_hgf_filter.argtypes = [ctypes.c_longlong, dp] + [ctypes.c_double] * 8 + [ctypes.c_int, ctypes.c_int] + \
            [dp] * 5 + [ctypes.POINTER(Counters)]
        _LIB.rxo_gmm_accumulate.restype = ctypes.c_int
        _LIB.rxo_gmm_accumulate.argtypes = [ctypes.c_longlong, ctypes.c_int, dp, dp, dp, dp, ctypes.POINTER(Counters)]
        _LIB.rxo_gmm_update.restype = ctypes.c_int
        _LIB.rxo_gmm_update.argtypes = [ctypes.c_int] + [dp] * 8 + [ctypes.POINTER(Counters)]
        _LIB.rxo_mvgmm_vmp.restype = ctypes.c_int
        _LIB.rxo_mvgmm_vmp.argtypes = [ctypes.c_longlong, ctypes.c_int, ctypes.c_int] + [dp] * 7 + [ctypes.c_int, dp, dp, dp]
        _LIB.rxo_lgssm_filter.restype = ctypes.c_int
        _LIB.rxo_lgssm_filter.argtypes = [ctypes.c_int] * 3 + [dp] * 6 + [ctypes.c_int] + [dp] * 4 + [ctypes.POINTER(Counters)]
        _LIB.rxo_lgssm_predict.restype = ctypes.c_int
        _LIB.rxo_lgssm_predict.argtypes = [ctypes.c_int] * 4 + [dp] * 6 + [ctypes.c_int] + [dp] * 5
        _LIB.rxo_drift_chain_bp.restype = ctypes.c_int
        _LIB.rxo_drift_chain_bp.argtypes = [ctypes.c_longlong, dp] + [ctypes.c_double] * 4 + [ctypes.c_int, dp, dp, dp,
                                                                                                 ctypes.POINTER(Counters)]
        _LIB.rxo_gauss_hermite.restype = ctypes.c_int
        _LIB.rxo_gauss_hermite.argtypes = [ctypes.c_int, dp, dp]
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def lgssm_bp(A, B, P, Q, m0, V0, y, prior_through_transition=False, free_energy=True):
    """One chain, reference schedule.  y: [T][dy].  Returns (mean [T,d], cov [T,d,d], fe|None, Counters)."""
    A, B, P, Q, m0, V0, y = map(_c, (A, B, P, Q, m0, V0, y))
    d, dy, T = A.shape[0], B.shape[0], y.shape[0]
    mean = np.empty((T, d))
    cov = np.empty((T, d, d))
    fe = ctypes.c_double(0.0)
    cnt = Counters()
    rc = lib().rxo_lgssm_bp(d, dy, T, _p(A), _p(B), _p(P), _p(Q), _p(m0), _p(V0), int(prior_through_transition),
                            _p(y), _p(mean), _p(cov), ctypes.byref(fe) if free_energy else None, ctypes.byref(cnt))
    if rc:
        raise RuntimeError(f"rxo_lgssm_bp failed with status {rc}")
    return mean, cov, (fe.value if free_energy else None), cnt


def lgssm_joints(A, B, P, Q, m0, V0, y, prior_through_transition=False):
    """Node-local joints q(x[t], A x[t-1]) of the transition nodes, reference schedule: mean [T-1, 2d], cov [T-1, 2d, 2d]."""
    A, B, P, Q, m0, V0, y = map(_c, (A, B, P, Q, m0, V0, y))
    d, dy, T = A.shape[0], B.shape[0], y.shape[0]
    jm, jc = np.empty((max(T - 1, 0), 2 * d)), np.empty((max(T - 1, 0), 2 * d, 2 * d))
    L = lib()
    L.rxo_lgssm_bp_joints.restype = ctypes.c_int
    rc = L.rxo_lgssm_bp_joints(ctypes.c_int(d), ctypes.c_int(dy), ctypes.c_int(T), _p(A), _p(B), _p(P), _p(Q), _p(m0), _p(V0),
                               ctypes.c_int(int(prior_through_transition)), _p(y), _p(jm), _p(jc))
    if rc:
        raise RuntimeError(f"rxo_lgssm_bp_joints failed with status {rc}")
    return jm, jc


def lgssm_predict(A, B, P, Q, m0, V0, y, horizon=0, prior_through_transition=False):
    """Predictions of y[1..T+H] (leave-one-out for the observed part, forecasts for the H unobserved steps) and the
    x-posteriors of the unobserved steps.  Returns pred_mean [T+H,dy], pred_cov [T+H,dy,dy], post_mean [H,d], post_cov [H,d,d]."""
    A, B, P, Q, m0, V0, y = map(_c, (A, B, P, Q, m0, V0, y))
    d, dy, T, H = A.shape[0], B.shape[0], y.shape[0], int(horizon)
    pm, pc = np.empty((T + H, dy)), np.empty((T + H, dy, dy))
    xm, xc = np.empty((H, d)), np.empty((H, d, d))
    rc = lib().rxo_lgssm_predict(d, dy, T, H, _p(A), _p(B), _p(P), _p(Q), _p(m0), _p(V0), int(prior_through_transition), _p(y),
                                 _p(pm), _p(pc), _p(xm), _p(xc))
    if rc:
        raise RuntimeError(f"rxo_lgssm_predict failed with status {rc}")
    return pm, pc, xm, xc


def drift_chain_bp(y, m0, v0, c, obs_var, prior_through_transition=True, free_energy=True):
    """Noise-free drift chain, one chain (rxo_drift_chain_bp).  Returns mean [T], var [T], fe | None, Counters."""
    y = _c(y).ravel()
    mean, var = np.empty(y.size), np.empty(y.size)
    fe = np.zeros(1)
    cnt = Counters()
    rc = lib().rxo_drift_chain_bp(y.size, _p(y), float(m0), float(v0), float(c), float(obs_var), int(prior_through_transition),
                                  _p(mean), _p(var), _p(fe) if free_energy else None, ctypes.byref(cnt))
    if rc:
        raise RuntimeError(f"rxo_drift_chain_bp failed with status {rc}")
    return mean, var, (float(fe[0]) if free_energy else None), cnt


def lgssm_kalman_rts(A, B, P, Q, m0, V0, y, prior_through_transition=False):
    A, B, P, Q, m0, V0, y = map(_c, (A, B, P, Q, m0, V0, y))
    d, dy, T = A.shape[0], B.shape[0], y.shape[0]
    mean = np.empty((T, d))
    cov = np.empty((T, d, d))
    nll = ctypes.c_double(0.0)
    rc = lib().rxo_lgssm_kalman_rts(d, dy, T, _p(A), _p(B), _p(P), _p(Q), _p(m0), _p(V0),
                                    int(prior_through_transition), _p(y), _p(mean), _p(cov), ctypes.byref(nll))
    if rc:
        raise RuntimeError(f"rxo_lgssm_kalman_rts failed with status {rc}")
    return mean, cov, nll.value


def lgssm_kalman_rts_tv(A, B, P, Q, m0, V0, step_model, y, prior_through_transition=False):
    """Textbook smoother with time-varying constants: A … V0 carry a leading model axis, step_model[t] names the model of t."""
    A, B, P, Q, m0, V0, y = map(_c, (A, B, P, Q, m0, V0, y))
    sm = np.ascontiguousarray(step_model, dtype=np.int32)
    M, d, dy, T = A.shape[0], A.shape[-1], B.shape[-2], y.shape[0]
    mean = np.empty((T, d))
    cov = np.empty((T, d, d))
    nll = ctypes.c_double(0.0)
    rc = lib().rxo_lgssm_kalman_rts_tv(d, dy, T, M, _p(A), _p(B), _p(P), _p(Q), _p(m0), _p(V0),
                                       sm.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), int(prior_through_transition),
                                       _p(y), _p(mean), _p(cov), ctypes.byref(nll))
    if rc:
        raise RuntimeError(f"rxo_lgssm_kalman_rts_tv failed with status {rc}")
    return mean, cov, nll.value


def lgssm_kalman_rts_affine(A, B, P, Q, m0, V0, y, state_offset=None, obs_offset=None, step_model=None, prior_through_transition=False):
    """Textbook smoother with known inputs: x[t] ~ N(A x[t-1] + cx[t], P), y[t] ~ N(B x[t] + cy[t], Q).  With `step_model`
    A … V0 carry a leading model axis."""
    A, B, P, Q, m0, V0, y = map(_c, (A, B, P, Q, m0, V0, y))
    M = A.shape[0] if step_model is not None else 1
    d, dy, T = A.shape[-1], B.shape[-2], y.shape[0]
    cx = None if state_offset is None else _c(np.broadcast_to(state_offset, (T, d)))
    cy = None if obs_offset is None else _c(np.broadcast_to(obs_offset, (T, dy)))
    sm = None if step_model is None else np.ascontiguousarray(step_model, dtype=np.int32)
    mean, cov, nll = np.empty((T, d)), np.empty((T, d, d)), ctypes.c_double(0.0)
    L = lib()
    L.rxo_lgssm_kalman_rts_affine.restype = ctypes.c_int
    ip = ctypes.POINTER(ctypes.c_int)
    rc = L.rxo_lgssm_kalman_rts_affine(ctypes.c_int(d), ctypes.c_int(dy), ctypes.c_int(T), ctypes.c_int(M), _p(A), _p(B), _p(P), _p(Q),
                                       _p(m0), _p(V0), sm.ctypes.data_as(ip) if sm is not None else None,
                                       ctypes.c_int(int(prior_through_transition)), _p(cx) if cx is not None else None,
                                       _p(cy) if cy is not None else None, _p(y), _p(mean), _p(cov), ctypes.byref(nll))
    if rc:
        raise RuntimeError(f"rxo_lgssm_kalman_rts_affine failed with status {rc}")
    return mean, cov, nll.value


def lgssm_filter(A, B, P, Q, m0, V0, y, prior_through_transition=True, free_energy=True):
    """Streaming / filtering run of one chain (rxo_lgssm_filter).  Returns history mean [T,d], cov [T,d,d],
    fe (mean over observations) | None, Counters."""
    A, B, P, Q, m0, V0, y = map(_c, (A, B, P, Q, m0, V0, y))
    d, dy, T = A.shape[0], B.shape[0], y.shape[0]
    mean = np.empty((T, d))
    cov = np.empty((T, d, d))
    fe = np.zeros(1)
    cnt = Counters()
    rc = lib().rxo_lgssm_filter(d, dy, T, _p(A), _p(B), _p(P), _p(Q), _p(m0), _p(V0), int(prior_through_transition),
                                _p(y), _p(mean), _p(cov), _p(fe) if free_energy else None, ctypes.byref(cnt))
    if rc:
        raise RuntimeError(f"rxo_lgssm_filter failed with status {rc}")
    return mean, cov, (float(fe[0]) if free_energy else None), cnt


def lgssm_bp_batch(A, B, P, Q, m0, V0, y, prior_through_transition=False, free_energy=True, nthreads=1, out=None):
    """Batch of chains sharing one model.  y: [T][chain][dy].  Returns mean [T,C,d], cov [T,C,d,d], fe[C]|None.
    out = (mean, cov): result arrays of the caller (bench.py's timed baseline hands over arrays whose pages are already mapped — the
    first touch of 8 GB of fresh pages by 256 threads of one process would otherwise be most of the measurement)."""
    A, B, P, Q, m0, V0, y = map(_c, (A, B, P, Q, m0, V0, y))
    d, dy = A.shape[0], B.shape[0]
    T, C = y.shape[0], y.shape[1]
    if out is not None:
        mean, cov = out
        assert mean.shape == (T, C, d) and cov.shape == (T, C, d, d) and mean.flags.c_contiguous and cov.flags.c_contiguous
    else:
        mean = np.empty((T, C, d))
        cov = np.empty((T, C, d, d))
    fe = np.empty(C) if free_energy else None
    cnt = Counters()
    rc = lib().rxo_lgssm_bp_batch(d, dy, T, C, _p(A), _p(B), _p(P), _p(Q), _p(m0), _p(V0),
                                  int(prior_through_transition), _p(y), _p(mean), _p(cov),
                                  _p(fe) if free_energy else None, int(nthreads), ctypes.byref(cnt))
    if rc:
        raise RuntimeError(f"rxo_lgssm_bp_batch failed with status {rc}")
    return mean, cov, fe, cnt


def lgssm_noise_vmp(A, B, P, m0, V0, y, nu0, S0, init_nu, init_V, iterations, prior_through_transition=False):
    """LGSSM with unknown observation-noise precision W ~ Wishart(nu0, S0), q(x, W) = q(x) q(W) (rxo_lgssm_noise_vmp), one chain.
    y: [T][dy].  Returns mean [T,d], cov [T,d,d] (last iteration), w_hist [it, 1 + dy²] (ν | V), fe [it]."""
    A, B, P, m0, V0, y, S0, init_V = map(_c, (A, B, P, m0, V0, y, S0, init_V))
    d, dy, T = A.shape[0], B.shape[0], y.shape[0]
    mean, cov = np.empty((T, d)), np.empty((T, d, d))
    wh, fe = np.empty((iterations, 1 + dy * dy)), np.empty(iterations)
    L = lib()
    L.rxo_lgssm_noise_vmp.restype = ctypes.c_int
    dp = ctypes.POINTER(ctypes.c_double)
    L.rxo_lgssm_noise_vmp.argtypes = [ctypes.c_int] * 3 + [dp] * 5 + [ctypes.c_int, dp, ctypes.c_double, dp, ctypes.c_double, dp, ctypes.c_int, dp, dp, dp, dp]
    rc = L.rxo_lgssm_noise_vmp(d, dy, T, _p(A), _p(B), _p(P), _p(m0), _p(V0), int(prior_through_transition), _p(y), float(nu0), _p(S0),
                               float(init_nu), _p(init_V), int(iterations), _p(mean), _p(cov), _p(wh), _p(fe))
    if rc:
        raise RuntimeError(f"rxo_lgssm_noise_vmp failed with status {rc}")
    return mean, cov, wh, fe


def gmm_vmp(y, mu0, v0, a0, b0, alpha0, init_m_mean, init_m_var, init_p_shape, init_p_rate, init_s_alpha, iterations,
            want_resp=False):
    """Univariate GMM mean-field VMP (see rxoracle.h).  Returns hist [it,5,K], fe [it], resp [N,K]|None, Counters."""
    y = _c(y)
    args = [_c(a) for a in (mu0, v0, a0, b0, alpha0, init_m_mean, init_m_var, init_p_shape, init_p_rate, init_s_alpha)]
    K, N = args[0].size, y.size
    hist = np.empty((iterations, 5, K))
    fe = np.empty(iterations)
    resp = np.empty((N, K)) if want_resp else None
    cnt = Counters()
    rc = lib().rxo_gmm_vmp(N, K, _p(y), *[_p(a) for a in args], int(iterations), _p(hist), _p(fe),
                           _p(resp) if want_resp else None, ctypes.byref(cnt))
    if rc:
        raise RuntimeError(f"rxo_gmm_vmp failed with status {rc}")
    return hist, fe, resp, cnt


def gauss_hermite(n):
    x, w = np.empty(n), np.empty(n)
    rc = lib().rxo_gauss_hermite(n, _p(x), _p(w))
    if rc:
        raise RuntimeError(f"rxo_gauss_hermite failed with status {rc}")
    return x, w


def gmm_accumulate(y, state, stats=None):
    """One shard's pass of one VMP iteration: state [5][K] (mean m, var m, shape p, rate p, alpha s) -> stats [3K+1]."""
    y = _c(y).ravel()
    state = _c(state)
    K = state.size // 5
    if stats is None:
        stats = np.empty(3 * K + 1)
    rc = lib().rxo_gmm_accumulate(y.size, K, _p(y), _p(state), _p(stats), None, None)
    if rc:
        raise RuntimeError(f"rxo_gmm_accumulate failed with status {rc}")
    return stats


def gmm_update(mu0, v0, a0, b0, alpha0, stats, state, want_fe=True):
    """Update from the (global) statistics; `state` [5][K] is modified in place.  Returns the free energy or None."""
    pri = [_c(a) for a in (mu0, v0, a0, b0, alpha0)]
    K = pri[0].size
    fe = np.zeros(1)
    assert state.flags.c_contiguous and state.dtype == np.float64 and state.size == 5 * K
    rc = lib().rxo_gmm_update(K, *[_p(a) for a in pri], _p(_c(stats)), _p(state), _p(fe) if want_fe else None, None)
    if rc:
        raise RuntimeError(f"rxo_gmm_update failed with status {rc}")
    return float(fe[0]) if want_fe else None


def mvgmm_pack(mean, cov, nu, V, alpha):
    """[K][SZ] state block (mean[d] | cov[d][d] | nu | V[d][d] | alpha) from per-component arrays."""
    mean, cov, V = (np.asarray(a, dtype=np.float64) for a in (mean, cov, V))
    K, d = mean.shape
    out = np.empty((K, 2 + d + 2 * d * d))
    out[:, :d] = mean
    out[:, d:d + d * d] = cov.reshape(K, -1)
    out[:, d + d * d] = nu
    out[:, d + d * d + 1:d + 2 * d * d + 1] = V.reshape(K, -1)
    out[:, -1] = alpha
    return out


def mvgmm_unpack(state, d):
    """inverse of mvgmm_pack for an array [..., K, SZ]: dict(mean, cov, nu, V, alpha)"""
    s = np.asarray(state)
    dd = d * d
    return dict(mean=s[..., :d], cov=s[..., d:d + dd].reshape(s.shape[:-1] + (d, d)), nu=s[..., d + dd],
                V=s[..., d + dd + 1:d + 2 * dd + 1].reshape(s.shape[:-1] + (d, d)), alpha=s[..., -1])


def mvgmm_vmp(y, mu0, S0, nu0, V0, alpha0, init, iterations, want_resp=False):
    """Multivariate mixture VMP (see rxoracle.h).  y: [N][d]; init: [K][SZ] (mvgmm_pack).  Returns hist [it][K][SZ], fe [it], resp|None."""
    y = _c(y)
    N, d = y.shape
    mu0, S0, nu0, V0, alpha0, init = (_c(a) for a in (mu0, S0, nu0, V0, alpha0, init))
    K = mu0.shape[0]
    SZ = 2 + d + 2 * d * d
    hist = np.empty((iterations, K, SZ))
    fe = np.empty(iterations)
    resp = np.empty((N, K)) if want_resp else None
    rc = lib().rxo_mvgmm_vmp(N, K, d, _p(y), _p(mu0), _p(S0), _p(nu0), _p(V0), _p(alpha0), _p(init), int(iterations),
                             _p(hist), _p(fe), _p(resp) if want_resp else None)
    if rc:
        raise RuntimeError(f"rxo_mvgmm_vmp failed with status {rc}")
    return hist, fe, resp


def hgf_filter(y, kappa, omega, z_variance, y_variance, z0=(0.0, 5.0), x0=(0.0, 5.0), vmp_iters=10, n_gh=31, want_fe=True):
    """HGF online filtering of one series (see rxoracle.h).  Returns zm, zv, xm, xv [T], fe [vmp_iters], Counters."""
    y = _c(y)
    T = y.size
    zm, zv, xm, xv, fe = (np.empty(T) for _ in range(4)) + (np.empty(vmp_iters),) if False else \
        (np.empty(T), np.empty(T), np.empty(T), np.empty(T), np.empty(vmp_iters))
    cnt = Counters()
    rc = lib().rxo_hgf_filter(T, _p(y), kappa, omega, z_variance, y_variance, z0[0], z0[1], x0[0], x0[1], vmp_iters, n_gh,
                              _p(zm), _p(zv), _p(xm), _p(xv), _p(fe) if want_fe else None, ctypes.byref(cnt))
    if rc:
        raise RuntimeError(f"rxo_hgf_filter failed with status {rc}")
    return zm, zv, xm, xv, fe, cnt
