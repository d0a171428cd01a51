"""rxhip_filter_step: the streaming driver one observation at a time (src/inference/streaming.jl:349-407) against the batch
filter of the same engine and against the oracle's one-step-graph filter."""
import numpy as np
import pytest

import rxhip
import rxoracle
from rxhip import workloads

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("d,dy,C,ptt", [(1, 1, 1, True), (2, 2, 5, False), (3, 2, 70, True), (4, 4, 3, False)])
def test_stepwise_filter_equals_the_batch_filter_and_the_oracle(d, dy, C, ptt):
    mdl = workloads.random_model(d, dy, seed=3 * d + dy)
    T = 40
    y = workloads.generate_batch(mdl, T, C, seed0=5)
    y[7, 0] = np.nan                                         # a missing observation in the stream
    args = [mdl[k] for k in ("A", "B", "P", "Q", "m0", "V0")]
    with rxhip.LGSSMEngine(*args, T=T, n_chains=C, prior_through_transition=ptt, allow_missing=True) as eng:
        eng.set_data(y)
        eng.run_filter(False)
        bm, bc = eng.marginals()
        means, covs, fes = [], [], []
        for t in range(T):
            m, V, fe = eng.filter_step(y[t])
            means.append(m); covs.append(V); fes.append(fe)
        eng.filter_reset()
        m0, _, _ = eng.filter_step(y[0])                     # after a reset the stream starts from the prior again
    means, covs, fes = np.stack(means), np.stack(covs), np.stack(fes)
    assert np.allclose(means, bm, rtol=1e-9, atol=1e-12) and np.allclose(covs, bc, rtol=1e-9, atol=1e-12)
    assert np.array_equal(m0, means[0])
    for c in sorted({0, C - 1}):
        # evidence of the whole stream = the textbook filter's −log p(y) (missing steps contribute nothing)
        nll = rxoracle.lgssm_kalman_rts(*args, y[:, c], prior_through_transition=ptt)[2]
        assert fes[:, c].sum() == pytest.approx(nll, rel=1e-9)
    if ptt and C > 1:                                        # the reference's one-step graph (oracle's filter, no missing values)
        om, oc, _ = rxoracle.lgssm_filter(*args, y[:, 1], prior_through_transition=True, free_energy=False)[:3]
        assert np.allclose(means[:, 1], om, rtol=1e-6, atol=1e-9)


def test_stepwise_filter_with_known_inputs_and_per_step_constants():
    rng = np.random.default_rng(1)
    d, dy, T, C = 2, 2, 25, 3
    from test_time_varying import _models
    mdl = _models(rng, d, dy, T)
    sm = np.arange(T, dtype=np.int32)
    cx, cy = rng.standard_normal((T, d)), rng.standard_normal((T, dy))
    y = rng.standard_normal((T, C, dy))
    from oracle import rxoracle as rxo
    with rxhip.LGSSMEngine(*mdl, T=T, n_chains=C, step_model=sm, state_offset=cx, obs_offset=cy) as eng:
        out = [eng.filter_step(y[t]) for t in range(T)]
        with pytest.raises(rxhip.RxHipError):                # the per-step tables end here
            eng.filter_step(y[0])
    for c in range(C):
        for t in (0, 9, T - 1):                              # the filtered belief of t = the last state of the smoother of y[:t+1]
            qm, qc, _ = rxo.lgssm_kalman_rts_affine(*mdl, y[:t + 1, c], cx[:t + 1], cy[:t + 1], step_model=sm[:t + 1])
            assert np.allclose(out[t][0][c], qm[-1], rtol=1e-6, atol=1e-9) and np.allclose(out[t][1][c], qc[-1], rtol=1e-6, atol=1e-9)
