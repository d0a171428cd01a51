"""The oracle of the first composed graph — a state-space chain whose observation-noise precision W has a Wishart prior, q(x, W) = q(x) q(W)
(oracle/rxoracle.c rxo_lgssm_noise_vmp; chain: test/models/statespace/mlgssm_test.jl:9-14, node pair: test/models/iid/mv_iid_precision_tests.jl:11-15).
No reference-held golden exists for this composition, so the oracle is pinned two ways: in the limit A = I, P → 0, B = I the chain IS the iid
model with unknown mean and precision, whose restatement (rxo_mvgmm_vmp, K = 1) is pinned to the reference's mixture golden; and the free
energy of a coordinate-ascent schedule must not increase."""
import numpy as np
import pytest

import rxoracle as rxo
from rxhip import workloads


@pytest.mark.parametrize("d", [1, 2, 3])
def test_constant_state_limit_is_the_iid_model(d):
    rng = np.random.default_rng(10 + d)
    N, it = 60, 6
    L = rng.standard_normal((d, d)) * 0.4 + np.eye(d)
    y = rng.standard_normal((N, d)) @ L.T + rng.standard_normal(d)
    mu0, S0m = rng.standard_normal(d), np.eye(d) * 4.0              # prior of the mean / of x[1]
    nu0, V0 = d + 1.0, np.eye(d)
    init = rxo.mvgmm_pack(mean=[mu0], cov=[S0m], nu=[d + 2.0], V=[np.eye(d) * 0.5], alpha=[1.0])
    hist, fe_iid, _ = rxo.mvgmm_vmp(y, [mu0], [S0m], [nu0], [V0], [1.0], init, it)
    m, c, wh, fe = rxo.lgssm_noise_vmp(np.eye(d), np.eye(d), np.eye(d) * 1e-11, mu0, S0m, y, nu0, V0, d + 2.0, np.eye(d) * 0.5, it)
    iid = rxo.mvgmm_unpack(hist, d)
    assert np.allclose(wh[:, 0], iid["nu"][:, 0], rtol=1e-12)
    assert np.allclose(wh[:, 1:].reshape(it, d, d), iid["V"][:, 0], rtol=1e-6, atol=1e-9)
    assert np.allclose(m[N // 2], iid["mean"][-1, 0], rtol=1e-6, atol=1e-8) and np.allclose(c[N // 2], iid["cov"][-1, 0], rtol=1e-5, atol=1e-9)
    assert np.allclose(fe, fe_iid, rtol=1e-6)


@pytest.mark.parametrize("d,dy,T,ptt", [(4, 4, 300, False), (3, 2, 120, True), (2, 4, 80, False), (1, 1, 50, False)])
def test_free_energy_does_not_increase_and_the_noise_is_recovered(d, dy, T, ptt):
    mdl = workloads.random_model(d, dy, seed=700 + d + dy)
    _, y = workloads.generate_chain(mdl, T, 3)
    m, c, wh, fe = rxo.lgssm_noise_vmp(mdl["A"], mdl["B"], mdl["P"], mdl["m0"], mdl["V0"], y, dy + 1.0, np.eye(dy), dy + 1.0, np.eye(dy), 12,
                                       prior_through_transition=ptt)
    assert np.all(np.diff(fe) <= 1e-9 * np.abs(fe[:-1]))
    assert fe[0] - fe[-1] > 0.0
    # with the converged E[W] as a KNOWN noise precision the smoother reproduces the posterior of the last iteration's q(x) update
    W = wh[-2, 0] * wh[-2, 1:].reshape(dy, dy)
    om, oc, _ = rxo.lgssm_kalman_rts(mdl["A"], mdl["B"], mdl["P"], np.linalg.inv(W), mdl["m0"], mdl["V0"], y, prior_through_transition=ptt)
    assert np.allclose(m, om, rtol=1e-9, atol=1e-11) and np.allclose(c, oc, rtol=1e-9, atol=1e-12)
    if T >= 120:   # E[W]⁻¹ lands near the generating covariance
        assert np.max(np.abs(np.linalg.inv(wh[-1, 0] * wh[-1, 1:].reshape(dy, dy)) - mdl["Q"])) < 0.6 * np.max(np.abs(mdl["Q"]))
