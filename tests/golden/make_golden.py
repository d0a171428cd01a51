#!/usr/bin/env python3
"""Regenerates the data sets of the reference's RNG-dependent known-answer tests with the StableRNG / randn
restatement (oracle/stable_rng.py) and writes them as small fixtures next to this script.  Run from the repo
root:  python tests/golden/make_golden.py

  mlgssm_stablerng1234.npz  test/models/statespace/mlgssm_test.jl:72-97   golden FE 6275.9015944677 (:128)
  ulgssm_stablerng123.npz   test/models/statespace/ulgssm_tests.jl:27-33  golden FE 1854.297647     (:48)
  hgf_stablerng42.npz       test/models/statespace/hgf_tests.jl:72-103    golden FE@it10 1.009879989585 (:118)
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


if __name__ == "__main__":
    mlgssm(); ulgssm(); hgf()
    print("wrote", [f for f in os.listdir(HERE) if f.endswith(".npz")])
