"""GPU parity tests of the streaming / filtering driver (rxhip_run_filter) against the oracle's one-step-graph loop
(rxo_lgssm_filter).  Reference: benchmark notebook cells 4 and 7 (`linear_gaussian_ssm_filtering`,
`rxinfer_inference_filtering`), driver src/inference/streaming.jl:349-407 with `@autoupdates`.
Tolerances as for the smoother: posteriors 1e-6 relative, free energy 1e-8 relative."""
import numpy as np
import pytest

import rxhip
import rxoracle
from rxhip import workloads

pytestmark = pytest.mark.gpu

RTOL_POST = 1e-6
RTOL_FE = 1e-8


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def check_filter(mdl, y, ptt=True, **kw):
    T, C = y.shape[:2]
    with rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=C,
                           prior_through_transition=ptt, **kw) as eng:
        eng.set_data(y)
        eng.run_filter(free_energy=True)
        mean, cov = eng.marginals()
        fe, fetot, cnt = eng.free_energy_per_chain(), eng.free_energy(), eng.counters()
        assert fetot.shape == (1,)
        # a smoothing run on the same handle afterwards is unaffected by the filtering run, and vice versa
        eng.run(1, True)
        sm, sc = eng.marginals()
        fes = eng.free_energy_per_chain()
        eng.run_filter(free_energy=False)
        mean2, cov2 = eng.marginals()
    assert np.array_equal(mean, mean2) and np.array_equal(cov, cov2)
    rules = prods = 0
    for c in range(C):
        om, oc, ofe, ocnt = rxoracle.lgssm_filter(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], y[:, c], ptt)
        assert rel(mean[:, c], om) < RTOL_POST and rel(cov[:, c], oc) < RTOL_POST
        assert abs(fe[c] - ofe) < RTOL_FE * abs(ofe)
        # mean over observations of −log p(y_t | y_<t) = the chain's smoothing free energy / T
        assert abs(fe[c] - fes[c] / T) < 1e-10 * abs(fe[c])
        rules += ocnt.rule_calls
        prods += ocnt.products
    assert abs(fetot[0] - fe.sum()) < RTOL_FE * abs(fe.sum())
    assert cnt["rule_calls"] == rules and cnt["products"] == prods
    # the last filtered belief is the last smoothed belief
    assert rel(mean[-1], sm[-1]) < RTOL_POST and rel(cov[-1], sc[-1]) < RTOL_POST
    return mean, cov


def test_notebook_filtering_model():
    """d = 2 notebook model, T = 1000, one chain — `rxinfer_inference_filtering(real_y, A, B, P, Q)`."""
    mdl = workloads.notebook_model()
    y = workloads.generate_batch(mdl, 1000, 1)
    check_filter(mdl, y)


@pytest.mark.parametrize("C,T,segments,ptt", [(64, 300, 0, True), (70, 257, 9, False), (1, 50, 49, True), (3, 2, 0, True),
                                              (5, 1, 0, True), (5, 1, 0, False), (128, 64, 1, True)])
def test_filter_shapes_and_segmentations(C, T, segments, ptt):
    mdl = workloads.c1_model()
    y = workloads.generate_batch(mdl, T, C, seed0=11)
    check_filter(mdl, y, ptt=ptt, segments=segments)


@pytest.mark.parametrize("d,dy", [(1, 1), (2, 1), (2, 2), (3, 3), (4, 2), (4, 4)])
def test_filter_dimensions(d, dy):
    mdl = workloads.random_model(d, dy, seed=100 + 10 * d + dy)
    y = workloads.generate_batch(mdl, 120, 66, seed0=5)
    check_filter(mdl, y, segments=6)


@pytest.mark.parametrize("d,T,C,segments", [(16, 90, 3, 4), (64, 61, 2, 5), (32, 1, 2, 0)])
def test_filter_dense_state_dimensions(d, T, C, segments):
    mdl = workloads.random_model(d, d, seed=7 + d)
    y = workloads.generate_batch(mdl, T, C, seed0=3)
    check_filter(mdl, y, segments=segments)


def test_filter_per_chain_models():
    """chains with different constants in one batch (non-uniform constant path)."""
    mdls = [workloads.random_model(4, 4, seed=s) for s in (1, 2, 3)]
    C, T = 11, 120
    cm = np.arange(C) % 3
    y = np.empty((T, C, 4))
    for c in range(C):
        y[:, c] = workloads.generate_chain(mdls[cm[c]], T, 50 + c)[1]
    stack = lambda k: np.stack([m[k] for m in mdls])
    with rxhip.LGSSMEngine(stack("A"), stack("B"), stack("P"), stack("Q"), stack("m0"), stack("V0"), T=T, n_chains=C,
                           chain_model=cm, segments=4, prior_through_transition=True) as eng:
        eng.set_data(y)
        eng.run_filter(True)
        mean, cov = eng.marginals()
        fe = eng.free_energy_per_chain()
    for c in range(C):
        m = mdls[cm[c]]
        om, oc, ofe, _ = rxoracle.lgssm_filter(m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"], y[:, c], True)
        assert rel(mean[:, c], om) < RTOL_POST and rel(cov[:, c], oc) < RTOL_POST
        assert abs(fe[c] - ofe) < RTOL_FE * abs(ofe)


def test_infer_mirror_filtering():
    """`infer(model, data, autoupdates, initialization, historyvars, keephistory)` as in the notebook."""
    mdl = workloads.notebook_model()
    _, y = workloads.generate_chain(mdl, 400, 21)
    spec = rxhip.linear_gaussian_ssm(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], np.ones(2), np.eye(2), prior_through_transition=True)
    res = rxhip.infer(model=spec, data={"y": y}, autoupdates=True, keephistory=len(y), historyvars={"x": "KeepLast"},
                      initialization={"x": rxhip.MvNormalMeanCovariance(mdl["m0"], mdl["V0"])}, free_energy=True)
    om, oc, ofe, _ = rxoracle.lgssm_filter(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], y, True)
    h = res.history["x"]
    assert h.mean.shape == (400, 2) and rel(h.mean, om) < RTOL_POST and rel(h.cov, oc) < RTOL_POST
    assert res.free_energy_history.shape == (1,) and abs(res.free_energy_history[0] - ofe) < RTOL_FE * abs(ofe)
    with pytest.raises(ValueError):
        rxhip.infer(model=spec, data={"y": y}, autoupdates=True, iterations=3)


def test_run_filter_rejected_for_other_engines():
    with rxhip.HGFEngine(10, 1, 1.0, 0.0, 0.04, 0.01) as eng:
        eng.set_data(np.zeros((10, 1)))
        assert rxhip._lib.lib().rxhip_run_filter(eng._h, 1) == 1  # RXHIP_ERR_BADARG
