"""Small problems in one launch (k_small_sweep, csrc/lgssm_kernels.hpp): the four phases of the sweep and the free-energy reduction of a few
short chains run back to back in one workgroup.  Same bodies as the separate kernels, so the results must be BIT-identical to the five-launch
schedule (RXHIP_SMALL_SWEEP=0), and both must match the oracle.  Reference shape: benchmarks/Linear Multivariate Gaussian State Space Model
Benchmark.ipynb cells 12 / 24 (one chain, T = 50 … 5000) and BASELINE config 1 (d = 4, T = 1000)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rxinfer.jl_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

pytestmark = pytest.mark.gpu


def _run(mdl, y, ptt, fused, monkeypatch, fe=True, iterations=1, segments=0):
    import rxhip
    monkeypatch.setenv("RXHIP_SMALL_SWEEP", "1" if fused else "0")
    T, C = y.shape[0], y.shape[1]
    with rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=C, prior_through_transition=ptt, segments=segments) as eng:
        eng.set_data(y)
        eng.run(iterations, fe)
        mean, cov = eng.marginals()
        return mean, cov, (eng.free_energy_per_chain() if fe else None), (eng.free_energy() if fe else None), eng.schedule()


@pytest.mark.parametrize("d,dy,T,C,ptt", [(4, 4, 1000, 1, False), (2, 2, 50, 1, False), (2, 2, 5000, 1, True), (4, 4, 257, 3, False), (3, 2, 400, 8, True),
                                           (1, 1, 90, 2, False), (4, 1, 130, 5, False), (1, 4, 77, 4, True), (2, 3, 10000, 1, False), (4, 4, 9, 16, False),
                                           (3, 3, 3, 2, False), (4, 4, 64, 64, True)])
def test_one_launch_equals_five_and_the_oracle(d, dy, T, C, ptt, monkeypatch):
    import rxoracle as rxo
    from rxhip import workloads
    mdl = workloads.random_model(d, dy, seed=900 + 10 * d + dy) if (d, dy) != (4, 4) else workloads.c1_model()
    y = workloads.generate_batch(mdl, T, C, seed0=21)
    m1, c1, f1, t1, sched = _run(mdl, y, ptt, True, monkeypatch, iterations=2)
    # the five launches on the SAME segmentation (the one-launch schedule picks short segments of its own: its boundary recursion is log-depth)
    m0, c0, f0, t0, _ = _run(mdl, y, ptt, False, monkeypatch, iterations=2, segments=sched["segments"])
    assert C > 16 or C * sched["segments"] <= 256, sched          # the shape the one-launch schedule takes (rxhip.hip caps S for it at these sizes)
    if C > 16 or sched["segments"] < 24:      # same kernels bodies, sequential boundary recursion: bit for bit
        assert np.array_equal(m1, m0) and np.array_equal(c1, c0) and np.array_equal(f1, f0) and np.array_equal(t1, t0)
    else:                                      # log-depth boundary recursion: the same maps composed in another order
        sd0 = np.sqrt(np.einsum("tcii->tci", c0))
        assert np.max(np.abs(m1 - m0) / sd0) < 1e-10 and np.array_equal(c1, c0) and np.max(np.abs(f1 - f0) / np.abs(f0)) < 1e-12
    for c in range(C):
        yc = np.ascontiguousarray(y[:, c])
        if d == dy:    # the reference message schedule (its observation message in moment form needs dy = d), else the smoother it is pinned to
            om, oc, ofe, _ = rxo.lgssm_bp(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], yc, prior_through_transition=ptt)
        else:
            om, oc, ofe = rxo.lgssm_kalman_rts(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], yc, prior_through_transition=ptt)
        sd = np.sqrt(np.einsum("tii->ti", oc))
        assert np.max(np.abs(m1[:, c] - om) / sd) < 1e-6 and np.max(np.abs(c1[:, c] - oc) / (sd[:, :, None] * sd[:, None, :])) < 1e-6
        assert f1[c] == pytest.approx(ofe, rel=1e-8)


def test_without_free_energy_and_through_infer(monkeypatch):
    import rxhip
    from rxhip import workloads
    mdl = workloads.c1_model()
    _, y = workloads.generate_chain(mdl, 1000, 42)
    yb = y[:, None, :]
    m1, c1, _, _, sched = _run(mdl, yb, False, True, monkeypatch, fe=False)
    m0, c0, _, _, _ = _run(mdl, yb, False, False, monkeypatch, fe=False, segments=sched["segments"])
    assert sched["segments"] >= 200     # one chain, T = 1000: as many short segments as the workgroup has lanes for
    assert np.max(np.abs(m1 - m0)) < 1e-10 * np.max(np.abs(m0)) and np.array_equal(c1, c0)
    spec = rxhip.linear_gaussian_ssm(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"])
    res = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("RXHIP_SMALL_SWEEP", fused)
        r = rxhip.infer(model=spec, data={"y": y}, free_energy=True)
        res[fused] = (r.posteriors["x"].mean, r.posteriors["x"].cov, r.free_energy[-1])
    scale = np.max(np.abs(res["0"][0]))
    assert np.max(np.abs(res["1"][0] - res["0"][0])) < 1e-9 * scale and np.max(np.abs(res["1"][1] - res["0"][1])) < 1e-9 * np.max(np.abs(res["0"][1]))
    assert abs(res["1"][2] - res["0"][2]) < 1e-11 * abs(res["0"][2])
    assert np.array_equal(res["1"][0], m1[:, 0])


@pytest.mark.parametrize("d,dy,T,C,ptt", [(4, 4, 1000, 1, True), (2, 2, 300, 4, True), (3, 3, 40, 2, False), (4, 2, 700, 1, True)])
def test_filtering_runs_in_one_launch(d, dy, T, C, ptt, monkeypatch):
    """rxhip_run_filter of a small problem: the same kernel without the suffix direction and the backward phase (FILT), against the launches it
    replaces on the same segmentation and — where the reference schedule exists (dy = d) — the oracle's streaming run."""
    import rxhip
    import rxoracle as rxo
    from rxhip import workloads
    mdl = workloads.random_model(d, dy, seed=950 + 10 * d + dy) if (d, dy) != (4, 4) else workloads.c1_model()
    y = workloads.generate_batch(mdl, T, C, seed0=23)
    res = {}
    seg = 0
    for fused in (True, False):
        monkeypatch.setenv("RXHIP_SMALL_SWEEP", "1" if fused else "0")
        with rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=C, prior_through_transition=ptt, segments=seg) as eng:
            eng.set_data(y)
            eng.run_filter(True)
            res[fused] = (*eng.marginals(), eng.free_energy_per_chain())
            seg = eng.schedule()["segments"]
    (m1, c1, f1), (m0, c0, f0) = res[True], res[False]
    sd0 = np.sqrt(np.einsum("tcii->tci", c0))
    assert np.max(np.abs(m1 - m0) / sd0) < 1e-10 and np.array_equal(c1, c0) and np.max(np.abs(f1 - f0) / np.abs(f0)) < 1e-12
    if d == dy:
        for c in range(C):
            om, oc, ofe, _ = rxo.lgssm_filter(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], np.ascontiguousarray(y[:, c]), prior_through_transition=ptt)
            sd = np.sqrt(np.einsum("tii->ti", oc))
            assert np.max(np.abs(m1[:, c] - om) / sd) < 1e-6 and np.max(np.abs(c1[:, c] - oc) / (sd[:, :, None] * sd[:, None, :])) < 1e-6
            assert f1[c] == pytest.approx(ofe, rel=1e-8)
