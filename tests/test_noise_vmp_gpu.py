"""The first composed graph on the device: state-space chains with an unknown observation-noise precision W ~ Wishart, q(x, W) = q(x) q(W)
(csrc/noise_kernels.hpp, rxhip_lgssm_noise_create) — per iteration one BP sweep of every chain with its own E[W], then every chain's Wishart
update, against the oracle (oracle/rxoracle.c rxo_lgssm_noise_vmp, pinned in tests/test_noise_vmp_oracle.py) iteration by iteration.
Reference models: test/models/statespace/mlgssm_test.jl:9-14 (chain), test/models/iid/mv_iid_precision_tests.jl:11-15 (node pair)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rxinfer.jl_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("d,dy,T,C,ptt,iters", [(4, 4, 400, 5, False, 6), (2, 2, 150, 3, True, 8), (3, 2, 90, 1, False, 5), (1, 1, 60, 70, False, 4),
                                                 (4, 1, 2000, 2, False, 3), (2, 4, 33, 4, True, 5)])
def test_every_iteration_against_the_oracle(d, dy, T, C, ptt, iters):
    import rxhip
    import rxoracle as rxo
    from rxhip import workloads
    mdl = workloads.random_model(d, dy, seed=800 + 10 * d + dy)
    y = workloads.generate_batch(mdl, T, C, seed0=17)
    nu0, S0 = dy + 1.0, np.eye(dy) * 0.7
    init_nu, init_V = dy + 2.0, np.eye(dy) * 0.3
    with rxhip.LGSSMNoiseEngine(mdl["A"], mdl["B"], mdl["P"], mdl["m0"], mdl["V0"], T, nu0, S0, init_nu, init_V, n_chains=C,
                                prior_through_transition=ptt) as eng:
        eng.set_data(y)
        eng.run(iters, True)
        mean, cov = eng.marginals()
        fe = eng.free_energy()
        fec = eng.free_energy_per_chain()
        nu, V = eng.noise_posterior()
        eng.run(iters, True)      # a run starts again from the initial marginal: same result
        assert np.array_equal(eng.free_energy(), fe) and np.array_equal(eng.marginals()[0], mean)
    fe_sum = np.zeros(iters)
    for c in range(C):
        om, oc, wh, ofe = rxo.lgssm_noise_vmp(mdl["A"], mdl["B"], mdl["P"], mdl["m0"], mdl["V0"], np.ascontiguousarray(y[:, c]), nu0, S0, init_nu,
                                              init_V, iters, prior_through_transition=ptt)
        fe_sum += ofe
        if c < 6:
            sd = np.sqrt(np.einsum("tii->ti", oc))
            assert np.max(np.abs(mean[:, c] - om) / sd) < 1e-6 and np.max(np.abs(cov[:, c] - oc) / (sd[:, :, None] * sd[:, None, :])) < 1e-6, c
            assert nu[c] == wh[-1, 0] and np.allclose(V[c], wh[-1, 1:].reshape(dy, dy), rtol=1e-9, atol=1e-12), c
            assert fec[c] == pytest.approx(ofe[-1], rel=1e-8), c
    assert np.allclose(fe, fe_sum, rtol=1e-8)
    assert np.all(np.diff(fe) <= 1e-9 * np.abs(fe[:-1]))      # coordinate ascent: the free energy does not increase


def test_what_the_engine_refuses():
    import rxhip
    from rxhip import _lib, workloads
    mdl = workloads.random_model(8, 4, seed=1)     # the MFMA path has no Wishart schedule yet
    with pytest.raises(rxhip.RxHipError) as ei:
        rxhip.LGSSMNoiseEngine(mdl["A"], mdl["B"], mdl["P"], mdl["m0"], mdl["V0"], 50, 5.0, np.eye(4))
    assert ei.value.status == _lib.ERR_UNSUPPORTED
    mdl = workloads.random_model(2, 2, seed=1)
    with pytest.raises(rxhip.RxHipError) as ei:     # Wishart(ν, ·) needs ν > dy − 1
        rxhip.LGSSMNoiseEngine(mdl["A"], mdl["B"], mdl["P"], mdl["m0"], mdl["V0"], 50, 0.5, np.eye(2))
    assert ei.value.status == _lib.ERR_BADARG


@pytest.mark.parametrize("d,dy,T,C,ptt,gamma", [(4, 4, 60, 3, False, None), (2, 2, 40, 1, True, None), (1, 1, 50, 4, False, "rate"), (1, 1, 30, 2, False, "scale")])
def test_engine_from_the_graph_is_the_structured_engine(d, dy, T, C, ptt, gamma):
    """rxhip_create on the GraphPPL spelling (Wishart / Gamma node on a random precision, precision-parametrised observation nodes, the
    `@initialization` marginal) builds the engine rxhip_lgssm_noise_create builds from the same numbers: identical results, bit for bit."""
    import rxhip
    from rxhip import graph, workloads
    mdl = workloads.random_model(d, dy, seed=40 + d)
    y = workloads.generate_batch(mdl, T, C, seed0=3)
    if gamma:
        a, b, ia, ib = 2.5, 0.8, 3.0, 1.5
        nu0, S0, init_nu, init_V = 2 * a, np.array([[1 / (2 * b)]]), 2 * ia, np.array([[1 / (2 * ib)]])
        gb, xs, ys, W = graph.lgssm_noise_graph(T, mdl["A"], mdl["B"], mdl["P"], mdl["m0"], mdl["V0"], a, b if gamma == "rate" else 1 / b, init=(ia, ib), gamma=gamma)
    else:
        nu0, S0, init_nu, init_V = dy + 1.5, np.eye(dy) * 0.4 + 0.1, dy + 3.0, np.eye(dy) * 0.2
        gb, xs, ys, W = graph.lgssm_noise_graph(T, mdl["A"], mdl["B"], mdl["P"], mdl["m0"], mdl["V0"], nu0, S0, init=(init_nu, init_V), prior_through_transition=ptt)
    g, keep = gb.tables(n_replicas=C, permute=np.random.default_rng(1).permutation(len(gb.ftype)))
    out = []
    for eng in (graph.create_noise_engine_from_graph(g),
                rxhip.LGSSMNoiseEngine(mdl["A"], mdl["B"], mdl["P"], mdl["m0"], mdl["V0"], T, nu0, S0, init_nu, init_V, n_chains=C, prior_through_transition=ptt)):
        with eng:
            eng.set_data(y)
            eng.run(5, True)
            out.append((eng.marginals(), eng.free_energy(), eng.noise_posterior()))
    (m1, f1, w1), (m2, f2, w2) = out
    assert np.array_equal(m1[0], m2[0]) and np.array_equal(m1[1], m2[1]) and np.array_equal(f1, f2)
    assert np.array_equal(w1[0], w2[0]) and np.array_equal(w1[1], w2[1])


def test_one_iteration_per_call_equals_one_call():
    """rxhip_lgssm_noise_continue: the iteration-at-a-time driver of the plugin (one `fire!` per iteration of batch.jl:391-430)."""
    import rxhip
    from rxhip import workloads
    d, dy, T, C, iters = 3, 2, 120, 4, 6
    mdl = workloads.random_model(d, dy, seed=9)
    y = workloads.generate_batch(mdl, T, C, seed0=5)
    args = (mdl["A"], mdl["B"], mdl["P"], mdl["m0"], mdl["V0"], T, dy + 1.0, np.eye(dy) * 0.6, dy + 2.0, np.eye(dy) * 0.3)
    with rxhip.LGSSMNoiseEngine(*args, n_chains=C) as eng:
        eng.set_data(y)
        eng.run(iters, True)
        want = (eng.marginals(), eng.free_energy(), eng.noise_posterior())
    with rxhip.LGSSMNoiseEngine(*args, n_chains=C) as eng:
        eng.continue_runs()
        fes = []
        for _ in range(iters):
            eng.set_data(y)           # the reference re-pushes the data every iteration
            eng.run(1, True)
            fes.append(eng.free_energy()[-1])
        got = (eng.marginals(), np.array(fes), eng.noise_posterior())
        eng.continue_runs(False)
        eng.run(iters, True)          # switched off: from the initial marginal again
        again = eng.free_energy()
    for a, b in zip(want, got):
        for x, z in zip(a, b) if isinstance(a, tuple) else [(a, b)]:
            assert np.array_equal(x, z)
    assert np.array_equal(again, want[1])


def test_infer_with_an_unknown_observation_precision():
    """the `infer(...)` mirror: linear_gaussian_ssm(..., Q = None, noise_precision_prior = Wishart(ν, S)) with initialization = {"W": …}"""
    import rxhip
    import rxoracle as rxo
    from rxhip import workloads
    d, dy, T, iters = 2, 2, 200, 7
    mdl = workloads.random_model(d, dy, seed=21)
    y = workloads.generate_batch(mdl, T, 1, seed0=8)[:, 0]
    S0 = np.eye(dy) * 0.5
    spec = rxhip.linear_gaussian_ssm(mdl["A"], mdl["B"], mdl["P"], None, mdl["m0"], mdl["V0"], noise_precision_prior=rxhip.Wishart(dy + 1.0, S0))
    res = rxhip.infer(model=spec, data={"y": y}, iterations=iters, free_energy=True, initialization={"W": rxhip.Wishart(dy + 2.0, 0.4 * np.eye(dy))})
    om, oc, wh, ofe = rxo.lgssm_noise_vmp(mdl["A"], mdl["B"], mdl["P"], mdl["m0"], mdl["V0"], y, dy + 1.0, S0, dy + 2.0, 0.4 * np.eye(dy), iters)
    assert np.allclose(res.posteriors["x"].mean, om, rtol=1e-7, atol=1e-9) and np.allclose(res.posteriors["x"].cov, oc, rtol=1e-7, atol=1e-10)
    assert res.posteriors["W"].nu == wh[-1, 0] and np.allclose(res.posteriors["W"].S, wh[-1, 1:].reshape(dy, dy), rtol=1e-9)
    assert np.allclose(res.free_energy, ofe, rtol=1e-9)
    with pytest.raises(ValueError, match="initialization"):
        rxhip.infer(model=spec, data={"y": y}, iterations=2)


def test_moments_inside_the_sweep_equal_the_separate_pass(monkeypatch):
    """the residual second moments accumulated by the backward sweep (per segment) against the pass over the posteriors (per time slice)"""
    import rxhip
    from rxhip import workloads
    d, dy, T, C, iters = 4, 3, 700, 70, 5
    mdl = workloads.random_model(d, dy, seed=31)
    y = workloads.generate_batch(mdl, T, C, seed0=6)
    out = []
    for hook in (None, "1"):
        if hook:
            monkeypatch.setenv("RXHIP_NOISE_MOMENTS_PASS", hook)
        else:
            monkeypatch.delenv("RXHIP_NOISE_MOMENTS_PASS", raising=False)
        with rxhip.LGSSMNoiseEngine(mdl["A"], mdl["B"], mdl["P"], mdl["m0"], mdl["V0"], T, dy + 1.0, np.eye(dy) * 0.5, dy + 2.0, np.eye(dy) * 0.3, n_chains=C) as eng:
            eng.set_data(y)
            eng.run(iters, True)
            out.append((eng.marginals()[0], eng.free_energy(), eng.noise_posterior()[1], eng.schedule()))
    (m1, f1, v1, s1), (m2, f2, v2, s2) = out
    assert s1["segments"] > 1
    assert np.allclose(m1, m2, rtol=1e-10, atol=1e-12) and np.allclose(f1, f2, rtol=1e-12) and np.allclose(v1, v2, rtol=1e-11)
