"""GCV as an op of the node-array executor (VERDICT r5 "Next 8": "so a … HGF layer can hang off a Gaussian tree"): the one-step graph of
test/models/statespace/hgf_tests.jl:9-31 — priors whose mean AND variance arrive as data (@autoupdates), the GCV node under q(y, x) q(z), 31-point
Gauss–Hermite cubature — through rxhip_tree_create against oracle/tree_oracle.py (pinned to oracle/rxoracle.c's HGF restatement, which reproduces the
reference's golden free energy: tests/test_tree_gcv_cpu.py, tests/test_golden_reference.py), against the specialised HGF engine, as an online filter
(rxhip_tree_continue keeps γ(z) from observation to observation, the host feeds the posteriors back as the next priors), and unrolled over several time
steps as a volatility layer on top of a Gaussian chain.  Every schedule, several replicas."""
import numpy as np
import pytest

import rxoracle
import tree_graphs as tg
from test_tree_engine_gpu import _check

pytestmark = pytest.mark.gpu


def _step_data(gb, names, rng, R):
    """[replica][z_prev_mean, z_prev_var, x_prev_mean, x_prev_var, y] in the graph's data-variable order"""
    return np.stack([[rng.normal(), 0.5 + 2.0 * rng.random(), rng.normal(), 0.5 + 2.0 * rng.random(), rng.normal()] for _ in range(R)])


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("kappa,omega", [(1.0, 0.0), (0.6, -1.2)])
def test_reference_step_graph_against_the_oracle(kappa, omega, mode, monkeypatch):
    from rxhip import graph
    from rxhip.tree import TreeEngine
    gb, names = graph.hgf_step_graph(kappa, omega, 0.04, 0.01, q_zt=(0.2, 1.5), q_xt=(0.0, 5.0), n_gh=31)
    ys = [v for v in range(len(gb.kind)) if gb.kind[v] == 1]
    R = 5
    data = _step_data(gb, names, np.random.default_rng(3), R)
    monkeypatch.setenv("RXHIP_TREE_MODE", str(mode))
    for its in (1, 4):
        with TreeEngine(gb, n_replicas=R) as eng:
            assert eng.info["kernels"] == 0
            eng.set_data(ys, data)
            eng.run(its, True)
            ref = _check(gb, ys, eng, data, iterations=its, replicas=(0, R - 1), tol=1e-9, tol_fe=1e-9)
            assert eng.counters()["rule_calls"] == ref["counters"]["rule_calls"] * R * its


def test_online_filter_equals_the_hgf_restatement_and_the_hgf_engine():
    """T observations one at a time: the engine keeps γ(z) between calls (rxhip_tree_continue), the host feeds q(zt), q(xt) back as the next priors — the loop of
    src/inference/streaming.jl:349-407 with the @autoupdates of hgf_tests.jl:42-45"""
    import rxhip
    from rxhip import graph
    from rxhip.tree import TreeEngine
    kappa, omega, zvar, yvar, iters, T = 1.0, 0.0, 0.04, 0.01, 5, 12
    rng = np.random.default_rng(7)
    y = np.cumsum(rng.standard_normal(T)) * 0.3
    z0, x0 = (0.0, 5.0), (0.0, 5.0)
    zm, zv, xm, xv, fe, _ = rxoracle.hgf_filter(y, kappa, omega, zvar, yvar, z0=z0, x0=x0, vmp_iters=iters, n_gh=31)
    gb, names = graph.hgf_step_graph(kappa, omega, zvar, yvar, q_zt=z0, q_xt=x0, n_gh=31)
    ys = [v for v in range(len(gb.kind)) if gb.kind[v] == 1]
    qz, qx, fes = z0, x0, []
    with TreeEngine(gb, n_replicas=1) as eng:
        eng.continue_runs(True)
        for t in range(T):
            eng.set_data(ys, np.array([[qz[0], qz[1], qx[0], qx[1], y[t]]]))
            eng.run(iters, True)
            post = eng.marginals([names["zt"], names["xt"]])
            qz = (float(post[names["zt"]][0][0, 0]), float(post[names["zt"]][1][0, 0, 0]))
            qx = (float(post[names["xt"]][0][0, 0]), float(post[names["xt"]][1][0, 0, 0]))
            fes.append(eng.free_energy())
            assert qz[0] == pytest.approx(zm[t], rel=1e-9, abs=1e-11) and qz[1] == pytest.approx(zv[t], rel=1e-9)
            assert qx[0] == pytest.approx(xm[t], rel=1e-9, abs=1e-11) and qx[1] == pytest.approx(xv[t], rel=1e-9)
    assert np.allclose(np.mean(fes, axis=0), fe, rtol=1e-9)   # (the restatement reports the per-iteration free energy averaged over the observations)
    with rxhip.HGFEngine(T, 1, kappa, omega, zvar, yvar, z0=z0, x0=x0, n_gh=31) as ref:   # the specialised engine: the whole series in one call
        ref.set_data(y[:, None])
        ref.run(iters, True)
        hzm, hzv, hxm, hxv = ref.history()
    assert qz[0] == pytest.approx(hzm[-1, 0], rel=1e-8, abs=1e-10) and qz[1] == pytest.approx(hzv[-1, 0], rel=1e-8)
    assert qx[0] == pytest.approx(hxm[-1, 0], rel=1e-8, abs=1e-10) and qx[1] == pytest.approx(hxv[-1, 0], rel=1e-8)


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_a_volatility_layer_on_top_of_a_gaussian_chain(mode, monkeypatch):
    """the filter's graph unrolled: z[t] ~ N(z[t−1], σz²), x[t] ~ GCV(x[t−1], z[t], κ, ω), y[t] ~ N(x[t], σy²) over several steps — every z[t] between two transition
    nodes and its GCV node, every x[t] the `y` of one GCV node and the `x` of the next"""
    from rxhip.tree import TreeEngine
    gb, ys, named = tg.volatility_chain(T=5)
    R = 3
    data = tg.random_data(gb, ys, R, 2)
    monkeypatch.setenv("RXHIP_TREE_MODE", str(mode))
    for its in (1, 3):
        with TreeEngine(gb, n_replicas=R) as eng:
            eng.set_data(ys, data)
            eng.run(its, True)
            _check(gb, ys, eng, data, iterations=its, replicas=(0, R - 1), tol=1e-9, tol_fe=1e-9)


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_data_valued_variances_and_precisions(mode, monkeypatch):
    """scalar Gaussian nodes whose variance (`Normal(mean = m_prev, var = v_prev)`) or precision arrives with the data — the prior nodes of an @autoupdates model —
    a different value in every replica"""
    from rxhip import _lib, graph
    from rxhip.tree import TreeEngine
    gb = graph.GraphBuilder()
    xp, x, w = gb.randomvar(1), gb.randomvar(1), gb.randomvar(1)
    m, v, y, tau, y2 = gb.datavar(1), gb.datavar(1), gb.datavar(1), gb.datavar(1), gb.datavar(1)
    gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, xp, m, v)
    gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, x, xp, gb.constvar(0.3))
    gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, y, x, gb.constvar(0.5))
    gb.node(_lib.NODE_NORMAL_MEAN_PRECISION, w, x, tau)
    gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, y2, w, gb.constvar(0.2))
    ys = [m, v, y, tau, y2]
    R = 6
    rng = np.random.default_rng(1)
    data = np.stack([[rng.normal(), 0.5 + rng.random(), rng.normal(), 0.5 + 2.0 * rng.random(), rng.normal()] for _ in range(R)])
    monkeypatch.setenv("RXHIP_TREE_MODE", str(mode))
    with TreeEngine(gb, n_replicas=R) as eng:
        eng.set_data(ys, data)
        eng.run(1, True)
        _check(gb, ys, eng, data, replicas=(0, 3, R - 1), tol=1e-10, tol_fe=1e-10)


def test_rxhip_create_hands_generic_gcv_and_mixture_graphs_to_the_executor():
    """the pattern matcher takes the flat HGF filter and the flat mixture models; a volatility chain unrolled in time and a mixture whose means share a Gaussian parent
    reach the executor through the same rxhip_create"""
    from rxhip.tree import TreeEngine
    for gb, ys, its in ((tg.volatility_chain(T=4)[:2] + (2,)), (tg.mixture_on_tree(N=6, K=2, d=2)[:2] + (2,))):
        data = tg.random_data(gb, ys, 2, 4)
        with TreeEngine(gb, n_replicas=2, force_executor=False) as eng:
            eng.set_data(ys, data)
            eng.run(its, True)
            _check(gb, ys, eng, data, iterations=its, replicas=(1,), tol=1e-9, tol_fe=1e-9)
