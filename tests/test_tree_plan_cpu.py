"""The executor's graph compiler on the CPU (rxhip_tree_plan: no device involved): the schedule's reference-equivalent counts — rule calls, message products,
marginals per replica and iteration — against the counters of oracle/tree_oracle.py on every test graph and on random forests; what the compiler refuses and
why; static figures that the layout implies.  (The GPU tests compare the same counts after a run; this is the `not gpu` half.)"""
import numpy as np
import pytest

import rxhip
from rxhip import _lib
from rxhip.tree import plan

import tree_graphs as tg
import tree_oracle

CASES = [(tg.two_branch_chain, dict(T=12)), (tg.two_branch_chain, dict(T=5, d=8, dy1=8, dy2=5)), (tg.two_branch_chain, dict(T=4, d=16, dy1=9, dy2=16)),
         (tg.branching_tree, dict(depth=3, fanout=2, d=1, seed=5)), (tg.scalar_tree, dict(n_leaves=6)), (tg.chain_with_prediction, dict(T=8, H=3)),
         (tg.star, dict(n_leaves=70, d=3)), (tg.chain_state_noise_precision, dict(T=8, d=3, dy=2, also_obs_noise=True)), (tg.known_mean_precision, dict(n=9, d=2))]


def _oracle_counts(gb, ys, seed=0):
    data = tg.data_dict(gb, ys, tg.random_data(gb, ys, 1, seed)[0])
    return tree_oracle.infer(gb.to_dump(), data)["counters"]


@pytest.mark.parametrize("builder,kw", CASES)
def test_counts_equal_the_oracles(builder, kw):
    gb, ys, _ = builder(**kw)
    p = plan(gb)
    c = _oracle_counts(gb, ys)
    assert p["rule_calls"] == c["rule_calls"]
    dmx = max(gb.rows[v] for v in range(len(gb.kind)) if gb.kind[v] != _lib.VARKIND_CONST)
    assert p["dmax"] == (dmx if dmx > 8 else (1 if dmx <= 1 else 2 if dmx <= 2 else 4 if dmx <= 4 else 8))
    assert p["n_ops"] > 0 and p["n_levels"] > 0 and p["mode"] == -1
    assert p["bytes_per_sweep"] % 8 == 0 and p["doubles_per_replica"] > 0


@pytest.mark.parametrize("seed", range(30))
def test_counts_on_random_forests(seed):
    gb, ys, named = tg.random_forest(seed, n_steps=14, dmax=(4, 5, 8, 12, 20)[seed % 5], precision_vars=seed % 2 == 1)
    p = plan(gb)
    assert p["rule_calls"] == _oracle_counts(gb, ys, seed)["rule_calls"]
    assert p["n_precision_vars"] == len(named["W"])


def test_what_the_compiler_refuses():
    # a cycle among the Gaussian variables
    gb, ys, nm = tg.two_branch_chain(T=3)
    gb.mvnormal_mean_cov(nm["x"][2], nm["x"][0], gb.constvar(np.eye(3)))
    with pytest.raises(rxhip.RxHipError) as ei:
        plan(gb)
    assert ei.value.status == _lib.ERR_UNSUPPORTED and "cycle" in str(ei.value)
    # a dimension above 64
    gb, _, _ = tg.two_branch_chain(T=2, d=65, dy1=3, dy2=3)
    with pytest.raises(rxhip.RxHipError) as ei:
        plan(gb)
    assert ei.value.status == _lib.ERR_UNSUPPORTED and "64" in str(ei.value)
    # `*` with a random (non-constant) matrix
    from rxhip.graph import GraphBuilder
    gb = GraphBuilder()
    x, a = gb.randomvar(2), gb.randomvar(2)
    gb.mvnormal_mean_cov(x, gb.constvar(np.zeros(2)), gb.constvar(np.eye(2)))
    gb.mvnormal_mean_cov(a, gb.constvar(np.zeros(2)), gb.constvar(np.eye(2)))
    o = gb.randomvar(2)
    gb.node(_lib.NODE_MULTIPLY, o, a, x)
    gb.mvnormal_mean_cov(gb.datavar(2), o, gb.constvar(np.eye(2)))
    with pytest.raises(rxhip.RxHipError) as ei:
        plan(gb)
    assert ei.value.status == _lib.ERR_UNSUPPORTED
    # a noise parameter that is not positive definite
    gb = GraphBuilder()
    x = gb.randomvar(2)
    gb.mvnormal_mean_cov(x, gb.constvar(np.zeros(2)), gb.constvar(np.array([[1.0, 2.0], [2.0, 1.0]])))
    gb.mvnormal_mean_cov(gb.datavar(2), x, gb.constvar(np.eye(2)))
    with pytest.raises(rxhip.RxHipError) as ei:
        plan(gb)
    assert ei.value.status == _lib.ERR_NOT_POSDEF


def test_a_hub_costs_a_linear_number_of_ops():
    """a star whose every third leaf sits behind a map needs n / 3 products of n − 1 messages each: shared partial-product trees keep the schedule linear in n
    (one by one it was 440 000 ops at n = 3 000)"""
    sizes = {}
    for n in (300, 3000):
        gb, ys, _ = tg.star(n_leaves=n, d=2)
        p = plan(gb)
        sizes[n] = p["n_ops"]
        if n == 300:   # (the oracle forms every product on its own: quadratic)
            assert p["rule_calls"] == _oracle_counts(gb, ys)["rule_calls"]
    assert sizes[3000] < 12 * 3000 and sizes[3000] < 11 * sizes[300]
