"""The univariate state-space graphs of the reference on the HIP path (SURVEY §8 a3/a5/a6/a7 with the scalar node family):
the noise-free drift chain of test/models/statespace/ulgssm_tests.jl (typeof(+) transitions) incl. its golden free energy,
and the `Normal(mean = …, var = …)` spellings of scalar random-walk / AR(1) chains, each built from its factor graph through
rxhip_create and compared with the oracle's restatement of the reference schedule."""
import os

import numpy as np
import pytest

import rxhip
import rxoracle
from rxhip import graph

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel(a, b):
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


def test_ulgssm_golden_free_energy_gpu():
    """ulgssm_tests.jl:27-48 on the reference's own data (StableRNG(123), regenerated): FE = 1854.297647 (atol 0.01),
    hidden signal within mean ± 3 std, variances positive, one free-energy value."""
    g = np.load(os.path.join(GOLD, "ulgssm_stablerng123.npz"))
    spec = rxhip.univariate_drift_chain(float(g["prior_mean"]), float(g["prior_var"]), float(g["c"]), float(g["obs_var"]))
    res = rxhip.infer(model=spec, data={"y": g["y"]}, free_energy=True)
    x = res.posteriors["x"]
    assert x.mean.shape == (500,) and res.free_energy.shape == (1,)
    assert abs(res.free_energy[-1] - float(g["fe_reference"])) < 1e-5  # the reference asserts 0.01
    sd = np.sqrt(x.var)
    assert np.all((x.mean - 3 * sd < g["hidden"]) & (g["hidden"] < x.mean + 3 * sd)) and np.all(x.var > 0)
    om, ov, ofe, _ = rxoracle.drift_chain_bp(g["y"], float(g["prior_mean"]), float(g["prior_var"]), float(g["c"]), float(g["obs_var"]))
    assert rel(x.mean, om) < 1e-6 and rel(x.var, ov) < 1e-6 and abs(res.free_energy[-1] - ofe) < 1e-8 * abs(ofe)


@pytest.mark.parametrize("T,C,ptt,const_first", [(1, 1, True, False), (7, 3, False, True), (500, 5, True, False), (3000, 130, True, True)])
def test_drift_chain_engine_from_graph_matches_oracle(T, C, ptt, const_first):
    rng = np.random.default_rng(T + C)
    m0, v0, c, ov = 0.7, 30.0, -0.25, 2.5
    y = (rng.normal(m0, 3.0, size=C)[None, :] + c * (np.arange(T)[:, None] + 1) + rng.normal(0, np.sqrt(ov), size=(T, C)))
    if ptt:
        gb, xs, ys = graph.drift_chain_graph(T, m0, v0, c, ov, const_first=const_first)
        eng = graph.create_engine_from_graph(gb.tables(n_replicas=C)[0])
    else:
        eng = rxhip.DriftChainEngine(T, m0, v0, c, ov, n_chains=C, prior_through_transition=False)
    with eng:
        eng.set_data(y[..., None])
        eng.run(2, True)
        mean, var = eng.marginals()
        fe, fe_it, cnt = eng.free_energy_per_chain(), eng.free_energy(), eng.counters()
        sub_m, sub_v = eng.marginals_of_chains([C - 1])
    assert fe_it.shape == (2,) and fe_it[0] == fe_it[1] and abs(fe_it[0] - fe.sum()) < 1e-12 * abs(fe_it[0])
    rc = pr = mg = 0
    for ch in range(C):
        om, ovr, ofe, oc = rxoracle.drift_chain_bp(y[:, ch], m0, v0, c, ov, prior_through_transition=ptt)
        assert rel(mean[:, ch, 0], om) < 1e-6 and rel(var[:, ch, 0, 0], ovr) < 1e-6
        assert abs(fe[ch] - ofe) < 1e-8 * abs(ofe)
        rc, pr, mg = rc + oc.rule_calls, pr + oc.products, mg + oc.marginals
    assert (cnt["rule_calls"], cnt["products"], cnt["marginals"]) == (2 * rc, 2 * pr, 2 * mg)
    assert np.array_equal(sub_m[0, :, 0], mean[:, C - 1, 0]) and np.array_equal(sub_v[0, :, 0, 0], var[:, C - 1, 0, 0])


@pytest.mark.parametrize("spell,a,b", [("normal", 1.0, 1.0), ("scaled", 0.9, 1.7), ("mixed", 0.8, 1.0)])
def test_scalar_chain_spellings_run_on_the_device(spell, a, b):
    """`x[t] ~ Normal(mean = a*x[t-1], var = p)`, `y[t] ~ Normal(mean = b*x[t], var = q)` with and without `*` nodes:
    graph → rxhip_create → d = dy = 1 schedule ≡ oracle (reference rule order) on the equivalent 1×1 matrices."""
    T, C, p, q, m0, v0 = 400, 4, 0.3, 2.0, -1.0, 25.0
    rng = np.random.default_rng(11)
    y = rng.standard_normal((T, C, 1)) * 2.0
    gb, xs, ys = graph.scalar_chain_graph(T, a, b, p, q, m0, v0, prior_through_transition=True, spell=spell)
    with graph.create_engine_from_graph(gb.tables(n_replicas=C)[0]) as eng:
        eng.set_data(y)
        eng.run(1, True)
        mean, cov = eng.marginals()
        fe = eng.free_energy_per_chain()
    M = lambda v: np.array([[v]])
    for ch in range(C):
        om, oc, ofe, _ = rxoracle.lgssm_bp(M(a), M(b), M(p), M(q), np.array([m0]), M(v0), y[:, ch], prior_through_transition=True)
        assert rel(mean[:, ch], om) < 1e-6 and rel(cov[:, ch], oc) < 1e-6 and abs(fe[ch] - ofe) < 1e-8 * abs(ofe)


def test_drift_chain_rejects_bad_arguments():
    with pytest.raises(rxhip.RxHipError) as ei:
        rxhip.DriftChainEngine(10, 0.0, -1.0, 1.0, 1.0)
    assert ei.value.status == 3
    with rxhip.DriftChainEngine(10, 0.0, 1.0, 1.0, 1.0) as eng:
        with pytest.raises(rxhip.RxHipError) as ei:
            eng.run(1, True)
        assert ei.value.status == 7
        y = np.zeros((10, 1, 1)); y[4] = np.nan
        eng.set_data(y)
        with pytest.raises(rxhip.RxHipError):
            eng.run(1, True)
