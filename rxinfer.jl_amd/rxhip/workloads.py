"""Synthetic workloads of BASELINE.json / SURVEY.md §8(d) (shared by tests and bench.py)."""
import math

import numpy as np


def rot(theta):
    return np.array([[np.cos(theta), -np.sin(theta)], [np.sin(theta), np.cos(theta)]])


def c1_model():
    """C1/C2 model: d = dy = 4.  A = blockdiag(R(π/15), R(π/35)) (benchmarks notebook cell 6,
    test/models/statespace/mlgssm_test.jl:90-91), B = diag(1.3,0.7,1.3,0.7), state noise 0.05·I,
    observation noise 10·I, prior N(0, 100·I)."""
    A = np.zeros((4, 4))
    A[:2, :2] = rot(np.pi / 15)
    A[2:, 2:] = rot(np.pi / 35)
    B = np.diag([1.3, 0.7, 1.3, 0.7])
    P = 0.05 * np.eye(4)
    Q = 10.0 * np.eye(4)
    return dict(A=A, B=B, P=P, Q=Q, m0=np.zeros(4), V0=100.0 * np.eye(4))


def notebook_model():
    """d = 2 model of the benchmark notebook (cell 6)."""
    return dict(A=rot(np.pi / 15), B=np.diag([1.3, 0.7]), P=0.05 * np.eye(2), Q=10.0 * np.eye(2), m0=np.zeros(2),
                V0=100.0 * np.eye(2))


def generate_chain(model, T, seed):
    """Generative loop of the notebook (cell 5) with numpy's default_rng(seed): returns x [T][d], y [T][dy]."""
    A, B, P, Q = model["A"], model["B"], model["P"], model["Q"]
    d, dy = A.shape[0], B.shape[0]
    rng = np.random.default_rng(seed)
    Lp = np.linalg.cholesky(P)
    Lq = np.linalg.cholesky(Q)
    wx = rng.standard_normal((T, d)) @ Lp.T
    wy = rng.standard_normal((T, dy)) @ Lq.T
    x = np.zeros((T, d))
    xp = np.zeros(d)
    for t in range(T):
        xp = A @ xp + wx[t]
        x[t] = xp
    y = x @ B.T + wy
    return x, y


def generate_batch(model, T, n_chains, seed0=42, threads=None):
    """y [T][chain][dy]; chain c is the generative loop of `generate_chain(model, T, seed0 + c)` (same draws from
    default_rng(seed0 + c), same recursion; SURVEY §8d C2 prescribes exactly this data).  The noise of the chains is drawn
    on a thread pool (numpy's generators release the GIL) and the time recursion runs over all chains at once, so the
    full C2 batch (T = 1e5 x 1024 chains, 3.3 GB) takes seconds instead of 1e8 interpreted steps."""
    import os
    from concurrent.futures import ThreadPoolExecutor

    A, B, P, Q = model["A"], model["B"], model["P"], model["Q"]
    d, dy = A.shape[0], B.shape[0]
    Lp = np.linalg.cholesky(P)
    Lq = np.linalg.cholesky(Q)
    x = np.empty((T, n_chains, d))
    wy = np.empty((T, n_chains, dy))

    def draw(c):  # standard normals only: no BLAS call inside the pool (concurrent matmuls from Python threads returned
        rng = np.random.default_rng(seed0 + c)  # different data from run to run with this numpy build)
        x[:, c, :] = rng.standard_normal((T, d))
        wy[:, c, :] = rng.standard_normal((T, dy))

    nthr = threads or min(32, os.cpu_count() or 1, n_chains)
    if nthr > 1:
        with ThreadPoolExecutor(nthr) as ex:
            list(ex.map(draw, range(n_chains)))
    else:
        for c in range(n_chains):
            draw(c)
    x = (x.reshape(-1, d) @ Lp.T).reshape(T, n_chains, d)      # w_t = L_P ε,  v_t = L_Q ε'
    wy = (wy.reshape(-1, dy) @ Lq.T).reshape(T, n_chains, dy)
    At = np.ascontiguousarray(A.T)
    for t in range(1, T):  # x_t = A x_{t-1} + w_t for every chain (x_0 = 0, as the notebook's generate_data)
        x[t] += x[t - 1] @ At
    y = x @ B.T
    y += wy
    return y


def random_model(d, dy, seed, stable=0.95):
    """Random dense, well-conditioned model for parity tests."""
    rng = np.random.default_rng(seed)
    Qm, _ = np.linalg.qr(rng.standard_normal((d, d)))
    A = stable * Qm
    B = rng.standard_normal((dy, d)) / np.sqrt(d) + (np.eye(dy, d) if dy <= d else 0.0)
    Wp = rng.standard_normal((d, d)) * 0.3
    P = Wp @ Wp.T + 0.1 * np.eye(d)
    Wq = rng.standard_normal((dy, dy)) * 0.5
    Q = Wq @ Wq.T + 0.5 * np.eye(dy)
    Wv = rng.standard_normal((d, d))
    V0 = Wv @ Wv.T + np.eye(d)
    m0 = rng.standard_normal(d)
    return dict(A=A, B=B, P=P, Q=Q, m0=m0, V0=V0)


def c3_model(d=64):
    """BASELINE config 3: d = dy = 64, dense A = 0.98·Q-factor of default_rng(64) normals,
    dense full-rank B = I + 0.1·G/8 (G from the same generator), state noise 0.05·I, obs noise 10·I,
    prior N(0, 100·I)."""
    rng = np.random.default_rng(64)
    Qm, _ = np.linalg.qr(rng.standard_normal((d, d)))
    G = rng.standard_normal((d, d))
    return dict(A=0.98 * Qm, B=np.eye(d) + 0.1 * G / 8.0, P=0.05 * np.eye(d), Q=10.0 * np.eye(d), m0=np.zeros(d),
                V0=100.0 * np.eye(d))


def generate_hgf_batch(T=2000, n_series=4096, seed=42, kappa=1.0, omega=0.0, z_variance=0.04, y_variance=0.01,
                       z_bound=5.0):
    """BASELINE config 4 data: the generative loop of test/models/statespace/hgf_tests.jl:72-92 for `n_series`
    independent series (numpy default_rng), returned as (z, x, y) arrays [T][series].

    The log-volatility random walk is reflected at ±z_bound.  The reference test has ONE series of 2000 steps; over
    thousands of series an unbounded walk (σ = 0.2·√2000 ≈ 9) takes some of them beyond the range the reference's
    31-point Gauss–Hermite rule can represent (nodes up to ±9.9 for the N(0,1)-based `mean_var` of the z-message),
    where the reference's free energy is NaN — and so is ours, by the same arithmetic (status
    RXHIP_ERR_NONFINITE_FE).  z_bound=None gives the unbounded walk.
    """
    rng = np.random.default_rng(seed)
    dz = math.sqrt(z_variance) * rng.standard_normal((T, n_series))
    ex = rng.standard_normal((T, n_series))
    ey = math.sqrt(y_variance) * rng.standard_normal((T, n_series))
    z = np.empty((T, n_series))
    x = np.empty((T, n_series))
    zp = np.zeros(n_series)
    xp = np.zeros(n_series)
    for t in range(T):
        zt = zp + dz[t]
        if z_bound is not None:
            zt = np.where(zt > z_bound, 2 * z_bound - zt, zt)
            zt = np.where(zt < -z_bound, -2 * z_bound - zt, zt)
        xt = xp + np.exp(0.5 * (kappa * zt + omega)) * ex[t]
        z[t], x[t] = zt, xt
        zp, xp = zt, xt
    return z, x, x + ey
