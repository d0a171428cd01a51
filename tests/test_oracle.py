"""CPU tests: pin the oracle (the restatement of the reference schedule) before trusting it.

Pins available without a Julia toolchain:
  * RNG-free known answers of the reference tests: test/models/models_tests.jl:255,286 (FE 3.51551,
    mean 1.5) and :308,335 (FE 2.26551, mean 1.0)
  * identities: BP on a tree == Kalman/RTS smoother, Bethe FE == -log p(y)
    (docs/src/manuals/variational/bethe-free-energy.md:70)
  * operation counts per step of the LGSSM graph (the reference counts rule
    calls the same way in test/callbacks/trace_tests.jl:93-104)
  * three spellings of the model give identical results (test/models/statespace/mlgssm_test.jl:131-135):
    here the two prior conventions must agree when expressed through each other.
"""
import numpy as np
import pytest

import rxoracle
from rxhip import workloads


def test_known_answer_sum_model():
    # x ~ N(a + b, 1); y ~ N(x, 1); a = 2, b = 1, y = 0   (models_tests.jl:242-256)
    m, V, fe, cnt = rxoracle.lgssm_bp(np.eye(1), np.eye(1), np.eye(1), np.eye(1), [3.0], [[1.0]], [[0.0]])
    assert abs(fe - 3.51551) < 1e-5
    assert abs(fe - (0.5 * np.log(4 * np.pi) + 9.0 / 4)) < 1e-12
    assert abs(m[0, 0] - 1.5) < 1e-12 and abs(V[0, 0, 0] - 0.5) < 1e-12


def test_known_answer_ratio_model():
    # x ~ N(a / b, 1); y ~ N(x, 1); a = 2, b = 1, y = 0   (models_tests.jl:294-308)
    m, V, fe, cnt = rxoracle.lgssm_bp(np.eye(1), np.eye(1), np.eye(1), np.eye(1), [2.0], [[1.0]], [[0.0]])
    assert abs(fe - 2.26551) < 1e-5
    assert abs(m[0, 0] - 1.0) < 1e-12


@pytest.mark.parametrize("d,dy,T,seed", [(4, 4, 1000, 0), (2, 2, 300, 1), (3, 3, 200, 2), (4, 2, 150, 3), (2, 1, 150, 4),
                                         (1, 1, 500, 5), (6, 3, 60, 6)])
def test_bp_equals_rts_and_bethe_equals_evidence(d, dy, T, seed):
    mdl = workloads.c1_model() if (d, dy, seed) == (4, 4, 0) else workloads.random_model(d, dy, seed)
    _, y = workloads.generate_chain(mdl, T, 42 + seed)
    args = (mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], y)
    m2, V2, nll = rxoracle.lgssm_kalman_rts(*args)
    if dy < d:
        # partial observations make the `*`_B(:in) precision B'Q⁻¹B singular; the reference schedule
        # converts it with cholinv (mean_cov) on the backward edge and fails (SURVEY §7 hard part (e));
        # the restatement reproduces that failure.  The textbook smoother is the oracle for dy < d.
        with pytest.raises(RuntimeError):
            rxoracle.lgssm_bp(*args)
        return
    m, V, fe, _ = rxoracle.lgssm_bp(*args)
    assert np.max(np.abs(m - m2)) < 1e-9 * max(1.0, np.max(np.abs(m2)))
    assert np.max(np.abs(V - V2)) < 1e-9 * max(1.0, np.max(np.abs(V2)))
    assert abs(fe - nll) < 1e-10 * abs(nll)
    # posterior covariances symmetric positive definite (mlgssm_test.jl:126 isposdef)
    assert np.all(np.linalg.eigvalsh(V) > 0)


def test_operation_counts_match_appendix_c():
    mdl = workloads.c1_model()
    for T in (1, 2, 7, 100):
        _, y = workloads.generate_chain(mdl, T, 1)
        *_, cnt = rxoracle.lgssm_bp(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], y, free_energy=False)
        assert cnt.rule_calls == 6 * T - 3          # 6 per interior step
        assert cnt.products == (1 if T == 1 else 4 * T - 4)
        assert cnt.marginals == T


def test_prior_conventions_agree():
    # x0 ~ N(m0,V0); x1 ~ N(A x0, P)  ==  x1 ~ N(A m0, A V0 A' + P)
    mdl = workloads.random_model(3, 3, 11)
    _, y = workloads.generate_chain(mdl, 80, 5)
    A, P = mdl["A"], mdl["P"]
    m1 = A @ mdl["m0"]
    V1 = A @ mdl["V0"] @ A.T + P
    a = rxoracle.lgssm_bp(A, mdl["B"], P, mdl["Q"], mdl["m0"], mdl["V0"], y, prior_through_transition=True)
    b = rxoracle.lgssm_bp(A, mdl["B"], P, mdl["Q"], m1, V1, y)
    assert np.max(np.abs(a[0] - b[0])) < 1e-10 and np.max(np.abs(a[1] - b[1])) < 1e-10
    assert abs(a[2] - b[2]) < 1e-9 * abs(b[2])


def test_non_posdef_is_an_error():
    mdl = workloads.c1_model()
    _, y = workloads.generate_chain(mdl, 5, 1)
    bad = -np.eye(4)
    with pytest.raises(RuntimeError):
        rxoracle.lgssm_bp(mdl["A"], mdl["B"], bad, mdl["Q"], mdl["m0"], mdl["V0"], y)


def test_batch_driver_matches_single_chain():
    mdl = workloads.c1_model()
    y = workloads.generate_batch(mdl, 50, 5)
    bm, bV, bfe, cnt = rxoracle.lgssm_bp_batch(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], y, nthreads=2)
    for c in range(5):
        m, V, fe, _ = rxoracle.lgssm_bp(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], y[:, c])
        assert np.array_equal(m, bm[:, c]) and np.array_equal(V, bV[:, c]) and fe == bfe[c]
    assert cnt.rule_calls == 5 * (6 * 50 - 3)


# ---- multivariate mixture oracle --------------------------------------------------------------------------------
def test_multivariate_mixture_oracle_reduces_to_the_univariate_one():
    """d = 1: Wishart(ν, V) = Gamma(shape ν/2, rate 1/(2V)) — the two independent restatements agree to rounding,
    marginals and free energy, at every iteration."""
    rng = np.random.default_rng(5)
    K, N = 3, 400
    mus = np.array([-5.0, 0.0, 6.0])
    z = rng.integers(0, K, N)
    y = mus[z] + rng.standard_normal(N) * np.array([0.5, 1.0, 2.0])[z]
    mu0, v0, a0, b0, al0 = np.array([-4.0, 1.0, 5.0]), np.array([1e2, 50.0, 1e3]), np.array([0.1, 0.5, 1.5]), np.array([0.2, 0.1, 0.7]), np.array([1.0, 2.0, 0.5])
    im, iv, ia, ib, isa = np.array([-4.0, 1.0, 5.0]), np.array([1.0, 2.0, 3.0]), np.array([1.0, 2.0, 1.5]), np.array([1.0, 0.5, 2.0]), np.array([1.0, 1.0, 2.0])
    h1, f1, _, _ = rxoracle.gmm_vmp(y, mu0, v0, a0, b0, al0, im, iv, ia, ib, isa, 8)
    init = rxoracle.mvgmm_pack(im[:, None], iv[:, None, None], 2 * ia, (1 / (2 * ib))[:, None, None], isa)
    h2, f2, _ = rxoracle.mvgmm_vmp(y[:, None], mu0[:, None], v0[:, None, None], 2 * a0, (1 / (2 * b0))[:, None, None], al0, init, 8)
    u = rxoracle.mvgmm_unpack(h2, 1)
    assert np.max(np.abs(f1 - f2) / np.abs(f1)) < 1e-12
    assert np.max(np.abs(u["mean"][..., 0] - h1[:, 0])) < 1e-12 and np.max(np.abs(u["cov"][..., 0, 0] - h1[:, 1]) / h1[:, 1]) < 1e-12
    assert np.max(np.abs(u["nu"] / 2 - h1[:, 2])) < 1e-12 and np.max(np.abs(1 / (2 * u["V"][..., 0, 0]) - h1[:, 3]) / h1[:, 3]) < 1e-12
    assert np.max(np.abs(u["alpha"] - h1[:, 4])) < 1e-12


def test_multivariate_mixture_oracle_on_the_reference_layout():
    """K = 3 clusters on a ring of radius 50, covariances diag(10, 20) rotated (gmm_multivariate_tests.jl:86-103), vague
    priors: the free energy decreases monotonically, the means are found, E[W]⁻¹ is close to the true covariances."""
    rng = np.random.default_rng(11)
    K, N = 3, 600
    ang = 2 * np.pi / K * np.arange(K)
    means = 50.0 * np.stack([np.cos(ang), np.sin(ang)], axis=1)
    covs = [np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]) @ np.diag([10.0, 20.0]) @ np.array([[np.cos(a), np.sin(a)], [-np.sin(a), np.cos(a)]]) for a in ang]
    zs = rng.integers(0, K, N)
    y = np.stack([rng.multivariate_normal(means[k], covs[k]) for k in zs])
    mu0 = 0.5 * means + rng.uniform(0, 10, (K, 2))
    S0, nu0, V0 = np.tile(1e6 * np.eye(2), (K, 1, 1)), np.full(K, 3.0), np.tile(1e2 * np.eye(2), (K, 1, 1))
    h, fe, resp = rxoracle.mvgmm_vmp(y, mu0, S0, nu0, V0, np.ones(K), rxoracle.mvgmm_pack(mu0, S0, nu0, V0, np.ones(K)), 25, want_resp=True)
    u = rxoracle.mvgmm_unpack(h[-1], 2)
    assert np.all(np.diff(fe) < 1e-9 * abs(fe[-1]))
    assert np.max(np.abs(u["mean"] - means)) < 1.5
    for k in range(K):
        assert np.max(np.abs(np.linalg.inv(u["nu"][k] * u["V"][k]) - covs[k])) < 6.0
    assert np.mean(np.argmax(resp, axis=1) == zs) > 0.99 and abs(u["alpha"].sum() - (N + K)) < 1e-9


def test_batch_generator_is_the_per_chain_generator_and_reproducible():
    """chain c of the C2 batch is the notebook's generative loop on default_rng(42 + c) — whatever the thread pool does."""
    from rxhip import workloads
    mdl = workloads.c1_model()
    a = workloads.generate_batch(mdl, 3000, 96, seed0=42)
    b = workloads.generate_batch(mdl, 3000, 96, seed0=42)
    assert np.array_equal(a, b)
    for c in (0, 17, 95):
        assert np.allclose(a[:, c], workloads.generate_chain(mdl, 3000, 42 + c)[1], rtol=0, atol=1e-11)
