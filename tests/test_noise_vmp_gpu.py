"""The first composed graph on the device: state-space chains with an unknown observation-noise precision W ~ Wishart, q(x, W) = q(x) q(W)
(csrc/noise_kernels.hpp, rxhip_lgssm_noise_create) — per iteration one BP sweep of every chain with its own E[W], then every chain's Wishart
update, against the oracle (oracle/rxoracle.c rxo_lgssm_noise_vmp, pinned in tests/test_noise_vmp_oracle.py) iteration by iteration.
Reference models: test/models/statespace/mlgssm_test.jl:9-14 (chain), test/models/iid/mv_iid_precision_tests.jl:11-15 (node pair)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rxinfer.jl_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("d,dy,T,C,ptt,iters", [(4, 4, 400, 5, False, 6), (2, 2, 150, 3, True, 8), (3, 2, 90, 1, False, 5), (1, 1, 60, 70, False, 4),
                                                 (4, 1, 2000, 2, False, 3), (2, 4, 33, 4, True, 5)])
def test_every_iteration_against_the_oracle(d, dy, T, C, ptt, iters):
    import rxhip
    import rxoracle as rxo
    from rxhip import workloads
    mdl = workloads.random_model(d, dy, seed=800 + 10 * d + dy)
    y = workloads.generate_batch(mdl, T, C, seed0=17)
    nu0, S0 = dy + 1.0, np.eye(dy) * 0.7
    init_nu, init_V = dy + 2.0, np.eye(dy) * 0.3
    with rxhip.LGSSMNoiseEngine(mdl["A"], mdl["B"], mdl["P"], mdl["m0"], mdl["V0"], T, nu0, S0, init_nu, init_V, n_chains=C,
                                prior_through_transition=ptt) as eng:
        eng.set_data(y)
        eng.run(iters, True)
        mean, cov = eng.marginals()
        fe = eng.free_energy()
        fec = eng.free_energy_per_chain()
        nu, V = eng.noise_posterior()
        eng.run(iters, True)      # a run starts again from the initial marginal: same result
        assert np.array_equal(eng.free_energy(), fe) and np.array_equal(eng.marginals()[0], mean)
    fe_sum = np.zeros(iters)
    for c in range(C):
        om, oc, wh, ofe = rxo.lgssm_noise_vmp(mdl["A"], mdl["B"], mdl["P"], mdl["m0"], mdl["V0"], np.ascontiguousarray(y[:, c]), nu0, S0, init_nu,
                                              init_V, iters, prior_through_transition=ptt)
        fe_sum += ofe
        if c < 6:
            sd = np.sqrt(np.einsum("tii->ti", oc))
            assert np.max(np.abs(mean[:, c] - om) / sd) < 1e-6 and np.max(np.abs(cov[:, c] - oc) / (sd[:, :, None] * sd[:, None, :])) < 1e-6, c
            assert nu[c] == wh[-1, 0] and np.allclose(V[c], wh[-1, 1:].reshape(dy, dy), rtol=1e-9, atol=1e-12), c
            assert fec[c] == pytest.approx(ofe[-1], rel=1e-8), c
    assert np.allclose(fe, fe_sum, rtol=1e-8)
    assert np.all(np.diff(fe) <= 1e-9 * np.abs(fe[:-1]))      # coordinate ascent: the free energy does not increase


def test_what_the_engine_refuses():
    import rxhip
    from rxhip import _lib, workloads
    mdl = workloads.random_model(8, 4, seed=1)     # the MFMA path has no Wishart schedule yet
    with pytest.raises(rxhip.RxHipError) as ei:
        rxhip.LGSSMNoiseEngine(mdl["A"], mdl["B"], mdl["P"], mdl["m0"], mdl["V0"], 50, 5.0, np.eye(4))
    assert ei.value.status == _lib.ERR_UNSUPPORTED
    mdl = workloads.random_model(2, 2, seed=1)
    with pytest.raises(rxhip.RxHipError) as ei:     # Wishart(ν, ·) needs ν > dy − 1
        rxhip.LGSSMNoiseEngine(mdl["A"], mdl["B"], mdl["P"], mdl["m0"], mdl["V0"], 50, 0.5, np.eye(2))
    assert ei.value.status == _lib.ERR_BADARG
