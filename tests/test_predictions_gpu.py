"""Predictions of the data variables and unobserved (`missing`) trailing time steps (`obtain_prediction`,
src/model/plugins/reactivemp_inference.jl:619-624): the HIP path against the oracle's restatement of the reference's
message schedule (forward ⊗ backward into `*`_B(:out), MvN_y(:out)), for d, dy ≤ 4 (state-space form, predict_kernels.hpp) and
on the MFMA path up to d = dy = 64 (observation-space form, generic_kernels.hpp), with dy ≥ d (for dy < d the reference's own
backward conversion fails)."""
import numpy as np
import pytest

import rxhip
import rxoracle
from rxhip import workloads

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


@pytest.mark.parametrize("d,dy,T,H,C,ptt", [(1, 1, 50, 7, 3, False), (2, 2, 400, 0, 5, True), (2, 4, 333, 12, 2, False),
                                          (3, 3, 1200, 30, 66, False), (4, 4, 5000, 100, 130, True),
                                          # the MFMA path (generic_kernels.hpp): packed pairs, padded and full tiles
                                          (6, 6, 120, 9, 4, False), (8, 8, 90, 0, 6, True), (5, 7, 77, 3, 3, False),
                                          (16, 16, 64, 5, 3, False), (20, 24, 50, 4, 2, True), (64, 64, 40, 6, 1, False)])
def test_predictions_and_forecast_match_oracle(d, dy, T, H, C, ptt):
    mdl = workloads.random_model(d, dy, seed=10 * d + dy)
    y = workloads.generate_batch(mdl, T, C, seed0=T)
    with rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=C, horizon=H,
                           prior_through_transition=ptt) as eng:
        eng.set_data(y)
        eng.run(1, True)
        mean, cov = eng.marginals()
        pm, pc = eng.predictions()
        fe = eng.free_energy_per_chain()
        sub_m, _ = eng.marginals_of_chains([C - 1])
    assert mean.shape == (T + H, C, d) and pm.shape == (T + H, C, dy) and pc.shape == (T + H, C, dy, dy)
    assert np.array_equal(sub_m[0], mean[:, C - 1])
    for c in sorted({0, C // 2, C - 1}):
        # dy > d: the reference's Bethe sum needs the (singular) marginal of b = B x — posteriors only, evidence from the textbook filter
        om, oc, ofe, _ = rxoracle.lgssm_bp(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], y[:, c], prior_through_transition=ptt,
                                           free_energy=dy <= d)
        if ofe is None:
            ofe = rxoracle.lgssm_kalman_rts(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], y[:, c], prior_through_transition=ptt)[2]
        opm, opc, oxm, oxc = rxoracle.lgssm_predict(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], y[:, c], horizon=H,
                                                    prior_through_transition=ptt)
        assert rel(mean[:T, c], om) < 1e-6 and rel(cov[:T, c], oc) < 1e-6 and abs(fe[c] - ofe) < 1e-8 * abs(ofe)  # the horizon changes nothing observed
        assert rel(pm[:, c], opm) < 1e-6 and rel(pc[:, c], opc) < 1e-6
        if H:
            assert rel(mean[T:, c], oxm) < 1e-6 and rel(cov[T:, c], oxc) < 1e-6


def test_infer_mirror_with_predictvars_and_missing_tail():
    """`infer(model = …, data = (y = [obs…, missing, missing],), predictvars = (y = KeepLast(),))`"""
    mdl = workloads.notebook_model()
    _, y = workloads.generate_chain(mdl, 120, 4)
    ym = np.vstack([y, np.full((5, 2), np.nan)])
    spec = rxhip.linear_gaussian_ssm(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"])
    res = rxhip.infer(model=spec, data={"y": ym}, predictvars=("y",), free_energy=True)
    opm, opc, oxm, oxc = rxoracle.lgssm_predict(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], y, horizon=5)
    assert res.predictions["y"].mean.shape == (125, 2) and rel(res.predictions["y"].mean, opm) < 1e-6 and rel(res.predictions["y"].cov, opc) < 1e-6
    assert res.posteriors["x"].mean.shape == (125, 2) and rel(res.posteriors["x"].mean[120:], oxm) < 1e-6
    # a `missing` value inside the data selects the masked schedule; the tail is then five more missing observations
    holes = ym.copy(); holes[10] = np.nan
    res2 = rxhip.infer(model=spec, data={"y": holes}, predictvars=("y",))
    om, oc, _ = rxoracle.lgssm_kalman_rts(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], holes,
                                          prior_through_transition=spec.prior_through_transition)
    assert res2.posteriors["x"].mean.shape == (125, 2) and rel(res2.posteriors["x"].mean, om) < 1e-6 and rel(res2.posteriors["x"].cov, oc) < 1e-6
    assert rel(res2.predictions["y"].mean[120:], opm[120:]) < 1e-3   # one hole 110 steps earlier: the forecasts barely move


def test_predictions_call_order_and_unsupported_shapes():
    mdl = workloads.notebook_model()
    with rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=20) as eng:
        with pytest.raises(rxhip.RxHipError) as ei:
            eng.predictions()
        assert ei.value.status == 7
        eng.set_data(np.zeros((20, 1, 2)))
        eng.run_filter(True)
        with pytest.raises(rxhip.RxHipError) as ei:   # a filtering run has no backward messages
            eng.predictions()
        assert ei.value.status == 7
    big = workloads.random_model(8, 8, seed=1)
    # `missing` inside the data at d > 4: the sequential schedule; node-local joints at any d (tests/test_dense_sequential.py)
    with rxhip.LGSSMEngine(big["A"], big["B"], big["P"], big["Q"], big["m0"], big["V0"], T=20, allow_missing=True) as eng:
        eng.set_data(np.zeros((20, 1, 8)))
        eng.run()
        jm, jc = eng.node_marginals()
        assert jm.shape == (19, 1, 16) and jc.shape == (19, 1, 16, 16) and np.all(np.isfinite(jc))


@pytest.mark.parametrize("C", [64, 70])
def test_one_pass_schedule_with_horizon_predictions_and_joints(C, monkeypatch):
    """The one-pass / table-driven schedule under everything that reads its results afterwards: forecast tail,
    predictions, node-local joints, chain gather — for a multiple of 64 chains (k_backward_sh) and not (k_backward<FUSED>)."""
    monkeypatch.setenv("RXHIP_ONE_PASS", "1")
    d, dy, T, H = 3, 3, 257, 11
    mdl = workloads.random_model(d, dy, seed=5)
    y = workloads.generate_batch(mdl, T, C, seed0=9)
    with rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=C, horizon=H,
                           prior_through_transition=True) as eng:
        eng.set_data(y)
        eng.run(2, True)
        mean, cov = eng.marginals()
        pm, pc = eng.predictions()
        jm, jc = eng.node_marginals()
        fe = eng.free_energy_per_chain()
    for c in (0, 63, C - 1):
        args = (mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], y[:, c])
        om, oc, ofe, _ = rxoracle.lgssm_bp(*args, prior_through_transition=True)
        opm, opc, oxm, oxc = rxoracle.lgssm_predict(*args, horizon=H, prior_through_transition=True)
        ojm, ojc = rxoracle.lgssm_joints(*args, prior_through_transition=True)
        assert rel(mean[:T, c], om) < 1e-6 and rel(cov[:T, c], oc) < 1e-6 and abs(fe[c] - ofe) < 1e-8 * abs(ofe)
        assert rel(mean[T:, c], oxm) < 1e-6 and rel(cov[T:, c], oxc) < 1e-6
        assert rel(pm[:, c], opm) < 1e-6 and rel(pc[:, c], opc) < 1e-6
        assert rel(jm[:, c], ojm) < 1e-6 and rel(jc[:, c], ojc) < 1e-6
