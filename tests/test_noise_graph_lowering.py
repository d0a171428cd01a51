"""rxhip_create / rxhip_graph_lower_lgssm_noise on the first COMPOSED graph: a state-space chain whose observation nodes are precision-parametrised
on ONE random W with a Wishart prior (dy = 1: a Gamma prior on τ) — lowered to the chain + (ν₀, S₀, q(W)) of rxhip_lgssm_noise_create.  Host only."""
import numpy as np
import pytest

import rxhip
from rxhip import _lib, graph, workloads


def _model(d, dy, seed=3):
    m = workloads.random_model(d, dy, seed=seed)
    return m


@pytest.mark.parametrize("d,dy,T,ptt", [(4, 4, 12, False), (3, 2, 7, True), (2, 1, 5, False), (1, 1, 9, False)])
def test_wishart_chain_is_recognised(d, dy, T, ptt):
    m = _model(d, dy)
    S0 = np.eye(dy) * 0.3 + 0.05
    gb, xs, ys, W = graph.lgssm_noise_graph(T, m["A"], m["B"], m["P"], m["m0"], m["V0"], dy + 2.5, S0, init=(dy + 4.0, 2.0 * S0), prior_through_transition=ptt)
    rng = np.random.default_rng(0)
    low = graph.lower_lgssm_noise(gb.tables(n_replicas=3, permute=rng.permutation(len(gb.ftype)))[0])   # node order does not matter
    assert (low["d"], low["dy"], low["T"], low["prior_through_transition"]) == (d, dy, T, ptt)
    for k in ("A", "B", "P", "m0", "V0"):
        np.testing.assert_array_equal(low[k], np.asarray(m[k], float).reshape(low[k].shape))
    assert low["nu0"] == dy + 2.5 and low["init_nu"] == dy + 4.0 and low["precision_var"] == W
    np.testing.assert_array_equal(low["S0"], S0)
    np.testing.assert_array_equal(low["init_V"], 2.0 * S0)
    assert list(low["state_var"]) == xs and list(low["data_var"]) == ys


@pytest.mark.parametrize("form", ["rate", "scale"])
def test_gamma_prior_on_a_scalar_precision(form):
    m = _model(1, 1)      # (a scalar chain: `Normal(mean = …, precision = τ)` nodes observe scalar states)
    a, b = 3.0, 0.5       # shape, rate
    gb, xs, ys, W = graph.lgssm_noise_graph(6, m["A"], m["B"], m["P"], m["m0"], m["V0"], a, b if form == "rate" else 1.0 / b, init=(4.0, 2.0), gamma=form)
    low = graph.lower_lgssm_noise(gb.tables()[0])
    # Gamma(a, b) = Wishart_1(2a, 1/(2b))
    assert low["nu0"] == 2 * a and low["S0"][0, 0] == pytest.approx(1.0 / (2 * b), rel=1e-15)
    assert low["init_nu"] == 8.0 and low["init_V"][0, 0] == 0.25


def _refused(gb, match, **kw):
    with pytest.raises(rxhip.RxHipError, match=match) as ei:
        graph.lower_lgssm_noise(gb.tables(**kw)[0])
    assert ei.value.status in (_lib.ERR_UNSUPPORTED, _lib.ERR_BADARG)


def test_what_is_refused():
    m = _model(2, 2)
    S0 = np.eye(2)
    args = (5, m["A"], m["B"], m["P"], m["m0"], m["V0"], 4.0, S0)
    gb, *_ = graph.lgssm_noise_graph(*args)                               # no @initialization marginal for W
    _refused(gb, "initialization")
    gb, xs, ys, W = graph.lgssm_noise_graph(*args, init=(4.0, S0))         # one observation with a constant covariance next to the random ones
    b = gb.randomvar(2)
    gb.multiply(b, gb.constvar(m["B"]), xs[-1])
    gb.mvnormal_mean_cov(gb.datavar(2), b, gb.constvar(np.eye(2)))
    _refused(gb, "two observation|constant covariance")
    gb, xs, ys, W = graph.lgssm_noise_graph(*args, init=(4.0, S0))         # a second prior node
    gb.node(_lib.NODE_WISHART, gb.randomvar(2), gb.constvar(4.0), gb.constvar(S0))
    _refused(gb, "more than one precision prior")
    gb, xs, ys, W = graph.lgssm_noise_graph(*args, init=(4.0, S0))         # W as a TRANSITION precision
    xn = gb.randomvar(2)
    gb.node(_lib.NODE_MVNORMAL_MEAN_PRECISION, xn, xs[-1], W)
    _refused(gb, "observation nodes only")
    gb, *_ = graph.lgssm_noise_graph(*args, init=(4.0, S0))
    _refused(gb, "missing", allow_missing=True)
    m5 = workloads.random_model(5, 2, seed=1)                              # d > 4: no device schedule for this family
    gb, *_ = graph.lgssm_noise_graph(5, m5["A"], m5["B"], m5["P"], m5["m0"], m5["V0"], 4.0, S0, init=(4.0, S0))
    _refused(gb, "d, dy <= 4")


def test_plain_chains_and_mixtures_are_not_claimed():
    """the dispatch of rxhip_create: a graph without a precision prior, and the Wishart graphs of the mixture family, go where they went"""
    m = _model(2, 2)
    gb, xs, ys = graph.lgssm_graph(4, m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"])
    with pytest.raises(rxhip.RxHipError, match="no Wishart / Gamma prior"):
        graph.lower_lgssm_noise(gb.tables()[0])
    gb, ys = graph.mv_iid_graph(6, np.zeros(2), np.eye(2), 4.0, np.eye(2), init=dict(m=(np.zeros(2), np.eye(2)), w=(4.0, np.eye(2))))
    with pytest.raises(rxhip.RxHipError):
        graph.lower_lgssm_noise(gb.tables()[0])
    assert graph.lower_mvgmm(gb.tables()[0])["K"] == 1
