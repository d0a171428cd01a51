"""CPU tests of the graph → schedule lowering (host logic, no GPU): the SoA dump of the graph
RxInfer builds for the LGSSM model is recognised whatever the node order, its constants are recovered, and
graphs outside the supported family are rejected with RXHIP_ERR_UNSUPPORTED (→ fall back to the stock plugin)."""
import numpy as np
import pytest

import rxhip
from rxhip import _lib, graph, workloads


def test_benchmark_model_graph_is_recognised():
    mdl = workloads.c1_model()
    T = 50
    gb, xs, ys = graph.lgssm_graph(T, mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"])
    # Appendix C: per interior step 4 factor nodes, 3 random + 1 data + 4 constant variables
    assert len(gb.ftype) == 1 + 2 * T + 2 * (T - 1)
    g, keep = gb.tables(n_replicas=3)
    low = graph.lower_lgssm(g)
    assert (low["d"], low["dy"], low["T"], low["prior_through_transition"]) == (4, 4, T, False)
    for k in ("A", "B", "P", "Q", "m0", "V0"):
        assert np.array_equal(low[k], mdl[k])
    assert list(low["state_var"]) == xs and list(low["data_var"]) == ys  # time order recovered


def test_node_order_is_irrelevant_and_prior_variant():
    mdl = workloads.random_model(3, 2, seed=4)
    gb, xs, ys = graph.lgssm_graph(17, mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], prior_through_transition=True)
    perm = np.random.default_rng(0).permutation(len(gb.ftype))
    g, keep = gb.tables(permute=perm)
    low = graph.lower_lgssm(g)
    assert low["prior_through_transition"] and low["T"] == 17 and low["dy"] == 2
    assert np.array_equal(low["A"], mdl["A"]) and list(low["data_var"]) == ys


def test_single_observation_graph():
    I = np.eye(1)
    gb, xs, ys = graph.lgssm_graph(1, I, I, I, I, [3.0], I)
    low = graph.lower_lgssm(gb.tables()[0])
    assert low["T"] == 1 and low["m0"][0] == 3.0


def test_unsupported_graphs_are_rejected():
    mdl = workloads.c1_model()
    # time-varying transition matrix
    gb, xs, ys = graph.lgssm_graph(5, mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"],
                                   A_of_t=lambda t: mdl["A"] * (1.0 + 0.01 * (t == 3)))
    with pytest.raises(rxhip.RxHipError) as ei:
        graph.lower_lgssm(gb.tables()[0])
    assert ei.value.status == _lib.ERR_UNSUPPORTED
    # a branching graph: two transitions out of one state
    gb, xs, ys = graph.lgssm_graph(3, mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"])
    a = gb.randomvar(4); gb.multiply(a, gb.constvar(mdl["A"]), xs[0]); xn = gb.randomvar(4); gb.mvnormal_mean_cov(xn, a, gb.constvar(mdl["P"]))
    with pytest.raises(rxhip.RxHipError) as ei:
        graph.lower_lgssm(gb.tables()[0])
    assert ei.value.status == _lib.ERR_UNSUPPORTED and "chain" in str(ei.value)
    # an unknown node type
    gb, xs, ys = graph.lgssm_graph(3, mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"])
    gb.ftype[2] = 99
    with pytest.raises(rxhip.RxHipError) as ei:
        graph.lower_lgssm(gb.tables()[0])
    assert ei.value.status == _lib.ERR_UNSUPPORTED
    # malformed tables
    g = _lib.GraphDesc()
    with pytest.raises(rxhip.RxHipError) as ei:
        graph.lower_lgssm(g)
    assert ei.value.status == _lib.ERR_BADARG


@pytest.mark.gpu
def test_engine_from_graph_matches_structured_descriptor():
    """rxhip_create(graph) == rxhip_lgssm_create(structured) on the device (T = 300, 5 replicas)."""
    mdl = workloads.c1_model()
    T, C = 300, 5
    y = workloads.generate_batch(mdl, T, C)
    gb, xs, ys = graph.lgssm_graph(T, mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"])
    g, keep = gb.tables(n_replicas=C)
    eng = graph.create_engine_from_graph(g)
    eng.set_data(y); eng.run(1, True)
    m1, V1 = eng.marginals(); f1 = eng.free_energy_per_chain(); eng.close()
    with rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=C) as e2:
        e2.set_data(y); e2.run(1, True)
        m2, V2 = e2.marginals(); f2 = e2.free_energy_per_chain()
    assert np.array_equal(m1, m2) and np.array_equal(V1, V2) and np.array_equal(f1, f2)
