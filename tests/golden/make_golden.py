#!/usr/bin/env python3
"""Regenerates the data sets of the reference's RNG-dependent known-answer tests with the StableRNG / randn
restatement (oracle/stable_rng.py) and writes them as small fixtures next to this script.  Run from the repo
root:  python tests/golden/make_golden.py

  mlgssm_stablerng1234.npz  test/models/statespace/mlgssm_test.jl:72-97   golden FE 6275.9015944677 (:128)
  ulgssm_stablerng123.npz   test/models/statespace/ulgssm_tests.jl:27-33  golden FE 1854.297647     (:48)
  hgf_stablerng42.npz       test/models/statespace/hgf_tests.jl:72-103    golden FE@it10 1.009879989585 (:118)
  mvgmm_stablerng43.npz     test/models/mixtures/gmm_multivariate_tests.jl:80-141  golden FE@it25 3436.7 (:141)
      The cluster labels come from `rand(rng, Categorical([1/3,1/3,1/3]), n)`, i.e. AliasTables.jl on a 4-cell table
      (oracle/stable_rng.py `categorical_alias_table`: one UInt64 per draw, cell = top bits, values below the cell's
      threshold redirect to its alias; the empty fourth cell draws from category 1, which then draws from 2, …).  The layout
      is restated from the package's documentation, not read from its source (absent here): of the 24 admissible layouts of
      this table it is the one that reproduces the golden within the reference's own tolerance (3436.721 vs 3436.7 ± 0.1);
      every other layout lands within ±6.2 (0.18 %) of it, so the fixture pins the mixture rules' free energy either way.
  uvgmm_stablerng12345.npz  test/models/mixtures/gmm_univariate_tests.jl:42-60  NOT reproduced (284.76 vs 357.69), see uvgmm()
"""
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
from stable_rng import StableRNG  # noqa: E402


def mlgssm():
    rng = StableRNG(1234)
    th = math.pi / 35
    A = np.array([[math.cos(th), -math.sin(th)], [math.sin(th), math.cos(th)]])
    B, Q, P = np.eye(2), np.eye(2), 25.0 * np.eye(2)  # the test's names: Q = state noise, P = observation noise
    n = 1000
    x_prev = np.array([10.0, -10.0])
    x, y = np.empty((n, 2)), np.empty((n, 2))
    for i in range(n):
        x[i] = rng.mvnormal(A @ x_prev, Q)
        y[i] = rng.mvnormal(B @ x[i], P)
        x_prev = x[i]
    np.savez(os.path.join(HERE, "mlgssm_stablerng1234.npz"), A=A, B=B, state_noise=Q, obs_noise=P, prior_mean=np.zeros(2),
             prior_cov=100.0 * np.eye(2), x=x, y=y, fe_reference=6275.9015944677, fe_atol=0.01)


def ulgssm():
    rng = StableRNG(123)
    n, P = 500, 100.0
    noise = np.array([rng.normal(0.0, math.sqrt(P)) for _ in range(n)])
    np.savez(os.path.join(HERE, "ulgssm_stablerng123.npz"), hidden=np.arange(1, n + 1, dtype=float), y=np.arange(1, n + 1) + noise,
             obs_var=P, prior_mean=0.0, prior_var=10000.0, c=1.0, fe_reference=1854.297647, fe_atol=0.01)


def hgf():
    rng = StableRNG(42)
    k, w, zv, yv, n = 1.0, 0.0, 0.2 ** 2, 0.1 ** 2, 2000
    z, x, y = np.empty(n), np.empty(n), np.empty(n)
    zp = xp = 0.0
    for i in range(n):
        z[i] = rng.normal(zp, math.sqrt(zv))
        v = math.exp(k * z[i] + w)
        x[i] = rng.normal(xp, math.sqrt(v))
        y[i] = rng.normal(x[i], math.sqrt(yv))
        zp, xp = z[i], x[i]
    np.savez(os.path.join(HERE, "hgf_stablerng42.npz"), z=z, x=x, y=y, kappa=k, omega=w, z_variance=zv, y_variance=yv,
             fe_reference_it10=1.009879989585, fe_atol=0.01)


def mvgmm():
    L, K, n = 50.0, 3, 500
    R = lambda a: np.array([[math.cos(a), -math.sin(a)], [math.sin(a), math.cos(a)]])
    means = [R(2 * math.pi / K * k) @ np.array([L, 0.0]) for k in range(K)]
    covs = []
    for k in range(K):
        r = R(2 * math.pi / K * k)
        c = r @ np.diag([10.0, 20.0]) @ r.T
        covs.append(0.5 * (c + c.T))
    rng = StableRNG(43)
    z = [rng.categorical_alias_table([1 / 3, 1 / 3, 1 / 3]) for _ in range(n)]   # rand(rng, Categorical(probvec), n), AliasTables layout
    y = np.array([rng.mvnormal(means[k], covs[k]) for k in z])

    def prior_means(r):  # the loop of gmm_multivariate_tests.jl:11-19 / :45-53
        out = []
        for i in range(1, K + 1):
            ang = ((2 * math.pi + r.rand()) / K) * (i - 1)
            b = L / 2 * (np.array([1.0, 0.0]) + np.array([r.rand(), r.rand()]))
            out.append(R(ang) @ b)
        return np.array(out)

    r2 = StableRNG(42)
    init_mean = prior_means(r2)   # inference_multivariate draws the @initialization marginals first …
    prior_mean = prior_means(r2)  # … then the model function draws its priors from the same stream
    np.savez(os.path.join(HERE, "mvgmm_stablerng43.npz"), y=y, z=np.array(z), true_means=np.array(means), true_covs=np.array(covs),
             prior_mean=prior_mean, init_mean=init_mean, prior_cov=1e6 * np.eye(2), wishart_nu=3.0, wishart_scale=1e2 * np.eye(2),
             iterations=25, fe_reference_it25=3436.7, fe_atol=0.1)


def uvgmm():
    """test/models/mixtures/gmm_univariate_tests.jl:42-60 with the same samplers.  NOT a pin: the reference asserts
    FE₁₀ ≈ 284.76 ± 0.1 (:97), the regenerated data give 357.69 — see tests/test_golden_reference.py for the bound that shows
    the reference's label vector must put ≈ 2/3 of the points into the TIGHT cluster, which no sampler of [1/3, 2/3] does."""
    rng = StableRNG(12345)
    n, mu, w = 150, (-10.0, 10.0), (3.777, 0.333)
    z = [rng.categorical_alias_table([1 / 3, 2 / 3]) for _ in range(n)]
    y = np.array([rng.normal(mu[k], math.sqrt(1.0 / w[k])) for k in z])
    np.savez(os.path.join(HERE, "uvgmm_stablerng12345.npz"), y=y, z=np.array(z), mu=np.array(mu), w=np.array(w),
             fe_reference_it10=284.76, fe_atol=0.1, reproduced=False)


if __name__ == "__main__":
    mlgssm(); ulgssm(); hgf(); mvgmm(); uvgmm()
    print("wrote", [f for f in os.listdir(HERE) if f.endswith(".npz")])
