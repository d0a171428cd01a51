"""GCV in the node-array executor, host side: oracle/tree_oracle.py's extension pinned to oracle/rxoracle.c's HGF restatement (which reproduces the reference's
golden free energy of test/models/statespace/hgf_tests.jl to 1e-5: tests/test_golden_reference.py), data-valued prior variances (@autoupdates), the compiler's
schedule and counts, what is refused by name."""
import numpy as np
import pytest

import rxhip
import rxoracle
import tree_graphs as tg
import tree_oracle
from rxhip import _lib, graph
from rxhip.tree import plan


def _step(kappa, omega, zvar, yvar, z0, x0, y0, its, n_gh=31):
    gb, names = graph.hgf_step_graph(kappa, omega, zvar, yvar, q_zt=z0, q_xt=x0, n_gh=n_gh)
    dv = [v for v in range(len(gb.kind)) if gb.kind[v] == _lib.VARKIND_DATA]
    data = {dv[0]: [z0[0]], dv[1]: [z0[1]], dv[2]: [x0[0]], dv[3]: [x0[1]], dv[4]: [y0]}
    return gb, names, tree_oracle.infer(gb.to_dump(), data, iterations=its)


@pytest.mark.parametrize("kappa,omega,n_gh", [(1.0, 0.0, 31), (0.5, -2.0, 31), (1.4, 0.7, 15)])
def test_one_step_of_the_reference_model_equals_the_pinned_restatement(kappa, omega, n_gh):
    z0, x0, y0, its = (0.3, 2.0), (-0.5, 1.5), 0.7, 7
    gb, names, ref = _step(kappa, omega, 0.04, 0.01, z0, x0, y0, its, n_gh)
    zm, zv, xm, xv, fe, _ = rxoracle.hgf_filter(np.array([y0]), kappa, omega, 0.04, 0.01, z0=z0, x0=x0, vmp_iters=its, n_gh=n_gh)
    assert np.max(np.abs(np.asarray(ref["fe"]) - fe) / np.abs(fe)) < 1e-12       # every iteration
    assert ref["mean"][names["zt"]][0] == pytest.approx(zm[0], rel=1e-12, abs=1e-14) and ref["cov"][names["zt"]][0, 0] == pytest.approx(zv[0], rel=1e-12)
    assert ref["mean"][names["xt"]][0] == pytest.approx(xm[0], rel=1e-12, abs=1e-14) and ref["cov"][names["xt"]][0, 0] == pytest.approx(xv[0], rel=1e-12)
    p = plan(gb)
    assert p["rule_calls"] == ref["counters"]["rule_calls"] == 8 and p["marginals"] == ref["counters"]["marginals"] == 4


def test_a_filter_run_step_by_step_equals_the_restatement():
    """the streaming loop by hand on the oracle: posteriors fed back as the next priors AND as the `@initialization` of the next step (the marginal q(zt) persists
    between observations in the reference's engine)"""
    kappa, omega, zvar, yvar, its, T = 1.0, 0.0, 0.04, 0.01, 5, 8
    y = np.cumsum(np.random.default_rng(4).standard_normal(T)) * 0.3
    zm, zv, xm, xv, fe, _ = rxoracle.hgf_filter(y, kappa, omega, zvar, yvar, vmp_iters=its, n_gh=31)
    qz, qx = (0.0, 5.0), (0.0, 5.0)
    for t in range(T):
        _, names, ref = _step(kappa, omega, zvar, yvar, qz, qx, y[t], its)
        qz = (float(ref["mean"][names["zt"]][0]), float(ref["cov"][names["zt"]][0, 0]))
        qx = (float(ref["mean"][names["xt"]][0]), float(ref["cov"][names["xt"]][0, 0]))
        assert qz == pytest.approx((zm[t], zv[t]), rel=1e-11) and qx == pytest.approx((xm[t], xv[t]), rel=1e-11)


def test_a_data_valued_variance_is_the_constant_one():
    gb = graph.GraphBuilder()
    xp, x = gb.randomvar(1), gb.randomvar(1)
    m, v, y = gb.datavar(1), gb.datavar(1), gb.datavar(1)
    gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, xp, m, v)
    gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, x, xp, gb.constvar(0.3))
    gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, y, x, gb.constvar(0.5))
    ref = tree_oracle.infer(gb.to_dump(), {m: [1.0], v: [2.0], y: [0.2]})
    pv = 1.0 / (1.0 / 2.3 + 1.0 / 0.5)
    assert ref["cov"][x][0, 0] == pytest.approx(pv, rel=1e-13) and ref["mean"][x][0] == pytest.approx(pv * (1.0 / 2.3 + 0.2 / 0.5), rel=1e-13)
    assert ref["fe"][0] == pytest.approx(0.5 * np.log(2 * np.pi * 2.8) + 0.5 * 0.8 ** 2 / 2.8, rel=1e-13)   # −log N(y; 1, 2 + 0.3 + 0.5)
    assert plan(gb)["rule_calls"] == ref["counters"]["rule_calls"]


def test_unrolled_volatility_chain_compiles_and_counts_like_the_oracle():
    gb, ys, named = tg.volatility_chain(T=6)
    data = tg.random_data(gb, ys, 1, 1)
    ref = tree_oracle.infer(gb.to_dump(), tg.data_dict(gb, ys, data[0]), iterations=10)
    assert abs(ref["fe"][-1] - ref["fe"][-2]) < 1e-6 * abs(ref["fe"][-1])   # (a fixed point; the cubature steps are not exact coordinate ascent: no monotonicity claim)
    p = plan(gb)
    assert p["rule_calls"] == ref["counters"]["rule_calls"] and p["products"] == ref["counters"]["products"] and p["marginals"] == ref["counters"]["marginals"]


def _refused(gb, status, *needles):
    with pytest.raises(rxhip.RxHipError) as ei:
        plan(gb)
    assert ei.value.status == status, ei.value
    for n in needles:
        assert n in str(ei.value), (n, str(ei.value))


def test_what_the_compiler_refuses():
    # q(y) q(x) at the GCV node (hgf_tests.jl:33-35 asks for q(y, x) q(z))
    gb, names = graph.hgf_step_graph(1.0, 0.0, 0.04, 0.01)
    f = gb.ftype.index(_lib.NODE_GCV)
    gb.set_clusters(f, (0, 1, 2, 3, 4))
    _refused(gb, _lib.ERR_UNSUPPORTED, "GCV")
    # no @initialization on the volatility input
    gb, names = graph.hgf_step_graph(1.0, 0.0, 0.04, 0.01)
    d = gb.to_dump()
    d["variables"][names["zt"]].pop("init")
    _refused(graph.GraphBuilder.from_dump(d), _lib.ERR_BADARG, "@initialization")
    # an observed y (the node's joint q(y, x) needs two random interfaces here)
    gb = graph.GraphBuilder()
    x, z, y = gb.randomvar(1), gb.randomvar(1), gb.datavar(1)
    gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, x, gb.constvar(0.0), gb.constvar(1.0))
    gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, z, gb.constvar(0.0), gb.constvar(1.0))
    gb.node(_lib.NODE_GCV, y, x, z, gb.constvar(1.0), gb.constvar(0.0))
    gb.initialize(z, _lib.INIT_NORMAL, (0.0, 1.0))
    _refused(gb, _lib.ERR_UNSUPPORTED, "GCV")
    # the volatility input behind a deterministic node
    gb = graph.GraphBuilder()
    x0, x1, z0, z1 = gb.randomvar(1), gb.randomvar(1), gb.randomvar(1), gb.randomvar(1)
    gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, x0, gb.constvar(0.0), gb.constvar(1.0))
    gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, z0, gb.constvar(0.0), gb.constvar(1.0))
    gb.node(_lib.NODE_MULTIPLY, z1, gb.constvar(0.5), z0)
    gb.node(_lib.NODE_GCV, x1, x0, z1, gb.constvar(1.0), gb.constvar(0.0))
    yv = gb.datavar(1)
    gb.node(_lib.NODE_NORMAL_MEAN_VARIANCE, yv, x1, gb.constvar(0.1))
    gb.initialize(z1, _lib.INIT_NORMAL, (0.0, 1.0))
    _refused(gb, _lib.ERR_UNSUPPORTED, "GCV", "volatility")
