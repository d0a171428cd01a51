"""The Julia plugin's output format end to end: tests/golden/graph_dumps/ holds the factor graphs of the
reference's own test models in the exchange format `dump_graph` (rxinfer.jl_amd/julia/HIPInferencePlugin.jl) writes after
walking a GraphPPL model.  CPU: every dump loads, lowers to the structured descriptor the reference model implies, and is
byte-reproducible from the committed generator.  GPU: rxhip_create on the dump + the regenerated reference data reproduce
the reference's golden free energies."""
import gzip
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import rxhip
import rxoracle
from rxhip import _lib, graph

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DUMPS = os.path.join(GOLD, "graph_dumps")


def load(name):
    return graph.GraphBuilder.from_dump(os.path.join(DUMPS, name + ".json.gz"))


def test_dumps_are_reproducible_from_the_committed_generator(tmp_path):
    before = {f: open(os.path.join(DUMPS, f), "rb").read() for f in sorted(os.listdir(DUMPS))}
    assert set(before) == {"mlgssm.json.gz", "ulgssm.json.gz", "gmm_univariate.json.gz", "gmm_multivariate.json.gz", "hgf_step.json.gz"}
    subprocess.check_call([sys.executable, os.path.join(GOLD, "make_graph_dumps.py")], stdout=subprocess.DEVNULL)
    for f, b in before.items():
        assert open(os.path.join(DUMPS, f), "rb").read() == b, f


def test_dump_format_and_node_vocabulary():
    d = json.load(gzip.open(os.path.join(DUMPS, "mlgssm.json.gz"), "rt"))
    assert d["format"] == "rxhip-graph-1" and len(d["factors"]) == 1 + 4 * 1000 and len(d["variables"]) == 3 + 8 * 1000
    assert d["factors"][1] == {"type": "*", "interfaces": [["out", 3], ["A", 4], ["in", 0]], "clusters": [0, 1, 0]}   # ((1, 3), (2,)): the constant on its own
    assert [i[0] for i in d["factors"][2]["interfaces"]] == ["out", "μ", "Σ"] and d["variables"][0]["name"] == "x_prior"
    names = {v[0] for v in graph.NODE_VOCABULARY.values()}
    for f in os.listdir(DUMPS):
        dd = json.load(gzip.open(os.path.join(DUMPS, f), "rt"))
        assert {x["type"] for x in dd["factors"]} <= names
    with pytest.raises(rxhip.RxHipError) as ei:
        graph.GraphBuilder.from_dump({"format": "rxhip-graph-1", "variables": [], "factors": [{"type": "Probit", "interfaces": []}]})
    assert ei.value.status == _lib.ERR_UNSUPPORTED


def test_dumps_lower_to_the_reference_models():
    g = np.load(os.path.join(GOLD, "mlgssm_stablerng1234.npz"))
    low = graph.lower_lgssm(load("mlgssm").tables()[0])
    assert (low["d"], low["dy"], low["T"], low["prior_through_transition"], low["deterministic"]) == (2, 2, 1000, True, False)
    assert np.array_equal(low["A"], g["A"]) and np.array_equal(low["P"], g["state_noise"]) and np.array_equal(low["Q"], g["obs_noise"])
    low = graph.lower_lgssm(load("ulgssm").tables()[0])
    assert low["deterministic"] and low["T"] == 500 and low["c"][0] == 1.0 and low["V0"][0, 0] == 1e4 and low["Q"][0, 0] == 100.0
    low = graph.lower_gmm(load("gmm_univariate").tables()[0])
    assert (low["N"], low["K"]) == (150, 2) and list(low["mu0"]) == [-2.0, 2.0] and list(low["init_p_rate"]) == [1e-12, 1e-12]
    low = graph.lower_mvgmm(load("gmm_multivariate").tables()[0])
    assert (low["N"], low["K"], low["d"]) == (500, 3, 2) and np.all(low["nu0"] == 3.0)
    low = graph.lower_hgf(load("hgf_step").tables()[0])
    assert low["kappa"] == 1.0 and low["omega"] == 0.0 and abs(low["z_variance"] - 0.04) < 1e-15 and low["n_gh"] == 31


@pytest.mark.gpu
def test_mlgssm_dump_reproduces_the_reference_golden():
    g = np.load(os.path.join(GOLD, "mlgssm_stablerng1234.npz"))
    with graph.create_engine_from_graph(load("mlgssm").tables()[0]) as eng:
        eng.set_data(g["y"][:, None, :])
        eng.run(1, True)
        assert abs(eng.free_energy()[0] - float(g["fe_reference"])) < 1e-6   # mlgssm_test.jl:128 asserts 0.01


@pytest.mark.gpu
def test_ulgssm_dump_reproduces_the_reference_golden():
    g = np.load(os.path.join(GOLD, "ulgssm_stablerng123.npz"))
    with graph.create_engine_from_graph(load("ulgssm").tables()[0]) as eng:
        eng.set_data(g["y"][:, None, None])
        eng.run(1, True)
        assert abs(eng.free_energy()[0] - float(g["fe_reference"])) < 1e-5   # ulgssm_tests.jl:48 asserts 0.01


@pytest.mark.gpu
def test_mixture_dumps_run_and_match():
    g = np.load(os.path.join(GOLD, "mvgmm_stablerng43.npz"))
    with graph.create_vmp_engine_from_graph(load("gmm_multivariate").tables()[0]) as eng:
        eng.set_data(g["y"])
        eng.run(int(g["iterations"]), True)
        assert abs(eng.free_energy()[-1] - float(g["fe_reference_it25"])) < float(g["fe_atol"])   # gmm_multivariate_tests.jl:141
    rng = np.random.default_rng(5)
    z = rng.random(150) < 1 / 3
    y = np.where(z, -10 + rng.standard_normal(150) / np.sqrt(3.777), 10 + rng.standard_normal(150) / np.sqrt(0.333))
    with graph.create_vmp_engine_from_graph(load("gmm_univariate").tables()[0]) as eng:
        eng.set_data(y)
        eng.run(10, True)
        fe, hist = eng.free_energy(), eng.history()
    oh, ofe, _, _ = rxoracle.gmm_vmp(y, [-2.0, 2.0], [1e3, 1e3], [0.01, 0.01], [0.01, 0.01], [1.0, 1.0], [-2.0, 2.0], [1e3, 1e3],
                                     [1.0, 1.0], [1e-12, 1e-12], [1.0, 1.0], 10)
    assert np.max(np.abs(fe - ofe) / np.abs(ofe)) < 1e-8 and np.max(np.abs(hist - oh) / np.maximum(np.abs(oh), 1e-300)) < 1e-6


@pytest.mark.gpu
def test_hgf_dump_reproduces_the_reference_golden():
    g = np.load(os.path.join(GOLD, "hgf_stablerng42.npz"))
    gb = load("hgf_step")
    with graph.create_vmp_engine_from_graph(gb.tables(n_observations=gb.n_observations)[0]) as eng:
        eng.set_data(g["y"][:, None])
        eng.run(10, True)
        assert abs(eng.free_energy()[-1] - float(g["fe_reference_it10"])) < 1e-4   # hgf_tests.jl:118 asserts 0.01


@pytest.mark.gpu
def test_reference_mixture_dump_through_the_generic_executor_reproduces_the_golden():
    """the multivariate mixture model as GraphPPL builds it (500 NormalMixture + 500 Categorical nodes, Dirichlet, Wishart and MvNormal priors) through rxhip_tree_create — a
    node per data point on the node-array executor — on the reference's own data: the golden free energy of gmm_multivariate_tests.jl:141, and the mixture engine's"""
    from rxhip.tree import TreeEngine
    g = np.load(os.path.join(GOLD, "mvgmm_stablerng43.npz"))
    gb = load("gmm_multivariate")
    its = int(g["iterations"])
    ys = [v for v in range(len(gb.kind)) if gb.kind[v] == 1]
    with TreeEngine(gb, n_replicas=1) as eng:
        eng.set_data(ys, np.asarray(g["y"], float).reshape(1, -1))
        eng.run(its, True)
        fe = eng.free_energy()
    assert abs(fe[-1] - float(g["fe_reference_it25"])) < float(g["fe_atol"])
    with graph.create_vmp_engine_from_graph(gb.tables()[0]) as ref:
        ref.set_data(g["y"])
        ref.run(its, True)
        assert np.max(np.abs(fe - ref.free_energy()) / np.abs(fe)) < 1e-9   # every iteration


@pytest.mark.gpu
def test_reference_hgf_dump_through_the_generic_executor_reproduces_the_golden():
    """the HGF one-step graph as GraphPPL builds it (GCV under q(y, x) q(z), priors whose mean and variance are data, @initialization, 31-point cubature) through
    rxhip_tree_create as the streaming driver runs it — an observation per call, rxhip_tree_continue, the posteriors fed back by the @autoupdates — on the reference's own
    series: the golden free energy of hgf_tests.jl:113-118 (the reference asserts it to 0.01)"""
    from rxhip.tree import TreeEngine
    g = np.load(os.path.join(GOLD, "hgf_stablerng42.npz"))
    gb = load("hgf_step")
    y = np.asarray(g["y"], float).ravel()
    dv = [v for v in range(len(gb.kind)) if gb.kind[v] == 1]      # z_prev_mean, z_prev_var, x_prev_mean, x_prev_var, y (creation order)
    d = gb.to_dump()
    zt = next(i for i, v in enumerate(d["variables"]) if v.get("name") == "zt")
    xt = next(i for i, v in enumerate(d["variables"]) if v.get("name") == "xt")
    qz, qx, fes = (0.0, 5.0), (0.0, 5.0), []
    with TreeEngine(gb, n_replicas=1) as eng:
        eng.continue_runs(True)
        for t in range(y.size):
            eng.set_data(dv, np.array([[qz[0], qz[1], qx[0], qx[1], y[t]]]))
            eng.run(10, True)
            post = eng.marginals([zt, xt])
            qz = (float(post[zt][0][0, 0]), float(post[zt][1][0, 0, 0]))
            qx = (float(post[xt][0][0, 0]), float(post[xt][1][0, 0, 0]))
            fes.append(eng.free_energy()[-1])
    assert abs(np.mean(fes) - float(g["fe_reference_it10"])) < 1e-4


@pytest.mark.gpu
def test_reference_univariate_mixture_dump_through_the_generic_executor():
    """gmm_univariate_tests.jl:7-26 as GraphPPL builds it (Beta / Bernoulli switch, Gamma precisions, NormalMeanVariance priors) through rxhip_tree_create against the
    restatement the mixture engine is held to"""
    from rxhip.tree import TreeEngine
    rng = np.random.default_rng(5)
    z = rng.random(150) < 1 / 3
    y = np.where(z, -10 + rng.standard_normal(150) / np.sqrt(3.777), 10 + rng.standard_normal(150) / np.sqrt(0.333))
    gb = load("gmm_univariate")
    ys = [v for v in range(len(gb.kind)) if gb.kind[v] == 1]
    with TreeEngine(gb, n_replicas=1) as eng:
        eng.set_data(ys, y[None, :])
        eng.run(10, True)
        fe = eng.free_energy()
    _, ofe, _, _ = rxoracle.gmm_vmp(y, [-2.0, 2.0], [1e3, 1e3], [0.01, 0.01], [0.01, 0.01], [1.0, 1.0], [-2.0, 2.0], [1e3, 1e3],
                                    [1.0, 1.0], [1e-12, 1e-12], [1.0, 1.0], 10)
    assert np.max(np.abs(fe - ofe) / np.abs(ofe)) < 1e-8
