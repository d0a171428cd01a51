"""`missing` observations on the MFMA path, parallel in time (csrc/dense_mseg_kernels.hpp): per-(chain, segment) elements and a
boundary recursion built on the device with the observation mask applied per step, against the sequential schedule it replaces
for smoothing runs (RXHIP_GSEQ=1: csrc/gseq_kernels.hpp, kept as the checker) and against the oracle — the smoother with skipped
updates, itself pinned to brute-force conditioning of the joint Gaussian (tests/test_missing_observations.py).
Reference behaviour: docs/src/manuals/inference/static.md:98-123, test/inference/prediction_tests.jl:197-213."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rxinfer.jl_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

pytestmark = pytest.mark.gpu


def _run(mdl, y, ptt, sequential, monkeypatch, segments=0):
    import rxhip
    if sequential:
        monkeypatch.setenv("RXHIP_GSEQ", "1")
    else:
        monkeypatch.delenv("RXHIP_GSEQ", raising=False)
    T, C = y.shape[0], y.shape[1]
    with rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=C, prior_through_transition=ptt,
                           allow_missing=True, segments=segments) as eng:
        eng.set_data(y)
        eng.run(1, True)
        mean, cov = eng.marginals()
        fe = eng.free_energy_per_chain()
        eng.run(2, True)                      # iterations re-push the data: same result
        assert np.array_equal(eng.marginals()[0], mean)
        return mean, cov, fe


@pytest.mark.parametrize("d,dy,T,C,ptt,segments", [(64, 64, 300, 1, False, 0), (6, 40, 130, 2, False, 0), (20, 33, 70, 1, True, 4), (64, 20, 150, 2, True, 5), (33, 7, 200, 3, False, 0), (16, 16, 120, 4, True, 0),
                                                  (8, 4, 260, 6, False, 0), (5, 3, 40, 3, True, 1), (48, 48, 90, 2, False, 89), (12, 12, 2, 2, False, 0)])
def test_missing_observations_parallel_in_time(d, dy, T, C, ptt, segments, monkeypatch):
    import rxoracle as rxo
    from rxhip import workloads
    mdl = workloads.random_model(d, dy, seed=300 + d + dy)
    y = workloads.generate_batch(mdl, T, C, seed0=11)
    rng = np.random.default_rng(d + T)
    y[rng.random((T, C)) < 0.15] = np.nan
    y[0, 0] = np.nan                                   # the first observation of a chain
    y[-1, C - 1] = np.nan                              # the last one
    if T > 60:
        y[20:55, 0] = np.nan                           # whole segments without a single observation
    y[T // 2, C - 1, 0] = np.nan                       # a partly missing vector counts as missing
    mp, cp, fp = _run(mdl, y, ptt, False, monkeypatch, segments)
    ms, cs, fs = _run(mdl, y, ptt, True, monkeypatch)
    for c in range(C):
        om, oc, nll = rxo.lgssm_kalman_rts(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], np.ascontiguousarray(y[:, c]), prior_through_transition=ptt)
        sd = np.sqrt(np.einsum("tii->ti", oc))
        for name, mean, cov, fe in (("parallel", mp, cp, fp), ("sequential", ms, cs, fs)):
            em = np.max(np.abs(mean[:, c] - om) / sd)
            ec = np.max(np.abs(cov[:, c] - oc) / (sd[:, :, None] * sd[:, None, :]))
            assert em < 1e-6 and ec < 1e-6, (name, c, em, ec)
            assert fe[c] == pytest.approx(nll, rel=1e-8, abs=1e-9), (name, c)


def test_the_rest_of_the_engine_is_unchanged(monkeypatch):
    """filtering runs (masked schedule + km_filter_out), predictions and the step-wise filter (sequential kernels) of such an engine"""
    import rxhip
    import rxoracle as rxo
    from rxhip import workloads
    monkeypatch.delenv("RXHIP_GSEQ", raising=False)
    mdl = workloads.random_model(24, 6, seed=9)
    T, C = 60, 2
    y = workloads.generate_batch(mdl, T, C, seed0=3)
    y[np.random.default_rng(1).random((T, C)) < 0.2] = np.nan
    one = (mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"])
    with rxhip.LGSSMEngine(*one, T=T, n_chains=C, allow_missing=True) as eng:
        eng.set_data(y)
        eng.run(1, True)
        mean, cov = eng.marginals()
        pm, pc = eng.predictions()
        eng.run_filter(True)
        fm, fc = eng.marginals()
    for c in range(C):
        om, oc, _ = rxo.lgssm_kalman_rts(*one, np.ascontiguousarray(y[:, c]))
        assert np.allclose(mean[:, c], om, rtol=1e-6, atol=1e-9)
        t = T // 3
        yl = y[:, c].copy()
        yl[t] = np.nan
        lm, lc, _ = rxo.lgssm_kalman_rts(*one, yl)
        assert np.allclose(pm[t, c], one[1] @ lm[t], rtol=1e-6, atol=1e-8)
        qm, qc, _ = rxo.lgssm_kalman_rts(*one, np.ascontiguousarray(y[:t + 1, c]))
        assert np.allclose(fm[t, c], qm[-1], rtol=1e-6, atol=1e-9) and np.allclose(fc[t, c], qc[-1], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("d,dy,T,C,segments", [(24, 6, 400, 2, 0), (64, 64, 260, 1, 50), (16, 16, 700, 1, 0)])
def test_two_level_boundary_recursion_equals_one_level(d, dy, T, C, segments, monkeypatch):
    """from 16 segments on the boundary recursion runs over group elements first (km_group, km_scan levels 2 / 3): the same posteriors and
    free energy as the plain recursion over all segments (RXHIP_MSEG_ONE_LEVEL), ragged last group included"""
    from rxhip import workloads
    mdl = workloads.random_model(d, dy, seed=41 + d)
    y = workloads.generate_batch(mdl, T, C, seed0=4)
    y[np.random.default_rng(T).random((T, C)) < 0.2] = np.nan
    out = []
    monkeypatch.setenv("RXHIP_MSEG_SCAN", "sequential")       # (not the log-depth recursion, which the cost model may prefer)
    for one_level in (False, True):
        if one_level:
            monkeypatch.setenv("RXHIP_MSEG_ONE_LEVEL", "1")
        else:
            monkeypatch.delenv("RXHIP_MSEG_ONE_LEVEL", raising=False)
        out.append(_run(mdl, y, False, False, monkeypatch, segments))
    (m2, c2, f2), (m1, c1, f1) = out
    sd = np.sqrt(np.einsum("tcii->tci", c1))
    assert np.max(np.abs(m2 - m1) / sd) < 1e-8
    assert np.max(np.abs(c2 - c1) / (sd[..., :, None] * sd[..., None, :])) < 1e-8
    assert np.allclose(f2, f1, rtol=1e-10, atol=1e-9)


@pytest.mark.parametrize("d,dy,T,C,segments,ptt", [(24, 6, 400, 2, 0, False), (64, 64, 260, 1, 50, False), (16, 16, 700, 1, 0, True), (8, 3, 90, 3, 3, False),
                                                      (40, 12, 130, 2, 33, True), (64, 20, 1200, 1, 0, False), (12, 12, 64, 5, 63, False), (32, 32, 50, 1, 4, True)])
def test_log_depth_boundary_recursion_equals_the_sequential_one(d, dy, T, C, segments, ptt, monkeypatch):
    """km_compose / km_apply (all prefix and suffix compositions of the segment elements in ⌈log₂ S⌉ rounds, RXHIP_MSEG_SCAN=log) against the
    plain recursion over all segments (RXHIP_MSEG_ONE_LEVEL): S = 3 … T − 1 segments, powers of two and their neighbours, several chains"""
    from rxhip import workloads
    mdl = workloads.random_model(d, dy, seed=61 + d)
    y = workloads.generate_batch(mdl, T, C, seed0=5)
    y[np.random.default_rng(T + d).random((T, C)) < 0.2] = np.nan
    y[0, 0] = np.nan
    monkeypatch.delenv("RXHIP_MSEG_ONE_LEVEL", raising=False)
    monkeypatch.setenv("RXHIP_MSEG_SCAN", "log")
    ml, cl, fl = _run(mdl, y, ptt, False, monkeypatch, segments)
    monkeypatch.setenv("RXHIP_MSEG_SCAN", "sequential")
    monkeypatch.setenv("RXHIP_MSEG_ONE_LEVEL", "1")
    m1, c1, f1 = _run(mdl, y, ptt, False, monkeypatch, segments)
    sd = np.sqrt(np.einsum("tcii->tci", c1))
    assert np.max(np.abs(ml - m1) / sd) < 1e-8
    assert np.max(np.abs(cl - c1) / (sd[..., :, None] * sd[..., None, :])) < 1e-8
    assert np.allclose(fl, f1, rtol=1e-10, atol=1e-9)


@pytest.mark.parametrize("d,dy,T,C,segments,group,ptt", [(64, 64, 260, 1, 50, 2, False), (24, 6, 400, 2, 37, 3, True), (16, 16, 700, 1, 129, 4, False), (8, 3, 90, 3, 9, 4, False),
                                                            (40, 12, 130, 2, 33, 8, True), (64, 20, 1200, 1, 0, 2, False), (12, 12, 64, 5, 6, 4, False), (32, 32, 50, 1, 4, 2, True)])
def test_log_depth_recursion_over_groups_of_segments(d, dy, T, C, segments, group, ptt, monkeypatch):
    """segments finer than the entries of the rounds (RXHIP_MSEG_GROUP: km_fold composes every group first, km_inner carries the boundary
    states from the group edges inwards): ragged last groups, a last group of one segment, two entries, one entry"""
    from rxhip import workloads
    mdl = workloads.random_model(d, dy, seed=71 + d)
    y = workloads.generate_batch(mdl, T, C, seed0=6)
    y[np.random.default_rng(T + d).random((T, C)) < 0.2] = np.nan
    y[-1, 0] = np.nan
    monkeypatch.delenv("RXHIP_MSEG_ONE_LEVEL", raising=False)
    monkeypatch.setenv("RXHIP_MSEG_SCAN", "log")
    monkeypatch.setenv("RXHIP_MSEG_GROUP", str(group))
    mg, cg, fg = _run(mdl, y, ptt, False, monkeypatch, segments)
    monkeypatch.delenv("RXHIP_MSEG_GROUP", raising=False)
    monkeypatch.setenv("RXHIP_MSEG_SCAN", "sequential")
    monkeypatch.setenv("RXHIP_MSEG_ONE_LEVEL", "1")
    m1, c1, f1 = _run(mdl, y, ptt, False, monkeypatch, segments)
    sd = np.sqrt(np.einsum("tcii->tci", c1))
    assert np.max(np.abs(mg - m1) / sd) < 1e-8
    assert np.max(np.abs(cg - c1) / (sd[..., :, None] * sd[..., None, :])) < 1e-8
    assert np.allclose(fg, f1, rtol=1e-10, atol=1e-9)


def _random_cases(n, seed):
    rng = np.random.default_rng(seed)
    for i in range(n):
        d = int(rng.choice([5, 8, 13, 16, 24, 32, 40, 48, 64]))
        dy = int(rng.integers(1, min(d, 64) + 1))
        T = int(rng.integers(2, 420))
        C = int(rng.choice([1, 1, 2, 3, 9]))
        seg = int(rng.choice([0, 0, 0, 1, 2, 17, 33])) if T > 40 else 0
        yield dict(i=i, d=d, dy=dy, T=T, C=C, segments=min(seg, T - 1), ptt=bool(rng.integers(0, 2)), rate=float(rng.choice([0.05, 0.3, 0.8])))


@pytest.mark.parametrize("case", list(_random_cases(12 + int(os.environ.get("RXHIP_STRESS", "0")), 777)),
                         ids=lambda c: f"{c['i']}-d{c['d']}x{c['dy']}-T{c['T']}-C{c['C']}-s{c['segments']}-{'ptt' if c['ptt'] else 'x1'}-{c['rate']}")
def test_random_masked_case_against_the_sequential_schedule(case, monkeypatch):
    """random shapes, segment counts (one level, two levels, one segment per chain) and missing rates up to 80 %: the masked MFMA schedule
    against the sequential one (itself checked against the oracle above)"""
    from rxhip import workloads
    mdl = workloads.random_model(case["d"], case["dy"], seed=900 + case["i"])
    y = workloads.generate_batch(mdl, case["T"], case["C"], seed0=case["i"])
    y[np.random.default_rng(case["i"]).random((case["T"], case["C"])) < case["rate"]] = np.nan
    mp, cp, fp = _run(mdl, y, case["ptt"], False, monkeypatch, case["segments"])
    ms, cs, fs = _run(mdl, y, case["ptt"], True, monkeypatch)
    sd = np.sqrt(np.einsum("tcii->tci", cs))
    assert np.max(np.abs(mp - ms) / sd) < 1e-6
    assert np.max(np.abs(cp - cs) / (sd[..., :, None] * sd[..., None, :])) < 1e-6
    assert np.allclose(fp, fs, rtol=1e-8, atol=1e-9)


def _step_models(d, dy, M, seed):
    from rxhip import workloads
    ms = [workloads.random_model(d, dy, seed=seed + 7 * m) for m in range(M)]
    return tuple(np.stack([m[k] for m in ms]) for k in ("A", "B", "P", "Q", "m0", "V0"))


@pytest.mark.parametrize("d,dy,T,C,M,ptt,rate,segments", [(24, 6, 300, 2, 3, False, 0.15, 0), (64, 64, 120, 1, 2, True, 0.0, 0), (16, 16, 500, 1, 5, False, 0.3, 0),
                                                            (40, 12, 90, 300, 4, False, 0.1, 0), (8, 8, 64, 3, 64, True, 0.2, 7)])
def test_per_step_constants_on_the_masked_schedule(d, dy, T, C, M, ptt, rate, segments, monkeypatch):
    """A[t], P[t], B[t], Q[t] (desc.step_model) at d > 4: the transition into step t and the observation at t use the constants of model
    step_model[t] — in the segment elements, the sweep kernel (kd_forward_info<…, STEPM>), the residual forms (one pass per model) and the
    free-energy constant (km_feconst).  Against the oracle and against the sequential schedule (RXHIP_STEPM_GSEQ)."""
    import rxhip
    import rxoracle as rxo
    mdl = _step_models(d, dy, M, seed=500 + d)
    rng = np.random.default_rng(d + T)
    sm = rng.integers(0, M, T).astype(np.int32)
    sm[:3] = [M - 1, 0, M - 1]
    y = rng.standard_normal((T, C, dy)) * 2.0
    if rate > 0:
        y[rng.random((T, C)) < rate] = np.nan
    out = {}
    for name, env in (("masked", None), ("sequential", "1")):
        if env:
            monkeypatch.setenv("RXHIP_STEPM_GSEQ", env)
        else:
            monkeypatch.delenv("RXHIP_STEPM_GSEQ", raising=False)
        with rxhip.LGSSMEngine(*mdl, T=T, n_chains=C, prior_through_transition=ptt, step_model=sm, allow_missing=rate > 0, segments=segments) as eng:
            eng.set_data(y)
            eng.run(1, True)
            out[name] = eng.marginals() + (eng.free_energy_per_chain(),)
    mm, cm, fm = out["masked"]
    ms, cs, fs = out["sequential"]
    sd = np.sqrt(np.einsum("tcii->tci", cs))
    assert np.max(np.abs(mm - ms) / sd) < 1e-6
    assert np.max(np.abs(cm - cs) / (sd[..., :, None] * sd[..., None, :])) < 1e-6
    assert np.allclose(fm, fs, rtol=1e-8, atol=1e-9)
    for c in (0, C - 1):
        om, oc, nll = rxo.lgssm_kalman_rts_affine(*mdl, np.ascontiguousarray(y[:, c]), step_model=sm, prior_through_transition=ptt)
        sdo = np.sqrt(np.einsum("tii->ti", oc))
        assert np.max(np.abs(mm[:, c] - om) / sdo) < 1e-6
        assert fm[c] == pytest.approx(nll, rel=1e-8, abs=1e-9)


@pytest.mark.parametrize("d,dy,T,C,M,segments", [(9, 4, 150, 6, 6, 0), (64, 20, 80, 3, 2, 5), (16, 16, 300, 300, 7, 0)])
def test_one_model_per_chain_with_missing_observations_on_the_masked_schedule(d, dy, T, C, M, segments, monkeypatch):
    """desc.chain_model with `missing` values at d > 4: every chain's elements, prior and sweep use ITS model's constant blocks"""
    import rxhip
    import rxoracle as rxo
    mdl = _step_models(d, dy, M, seed=700 + d)
    rng = np.random.default_rng(d * T)
    cm = rng.integers(0, M, C).astype(np.int32)
    cm[0], cm[-1] = M - 1, 0
    y = rng.standard_normal((T, C, dy)) * 2.0
    y[rng.random((T, C)) < 0.2] = np.nan
    out = {}
    for name, env in (("masked", None), ("sequential", "1")):
        if env:
            monkeypatch.setenv("RXHIP_STEPM_GSEQ", env)
        else:
            monkeypatch.delenv("RXHIP_STEPM_GSEQ", raising=False)
        with rxhip.LGSSMEngine(*mdl, T=T, n_chains=C, chain_model=cm, allow_missing=True, segments=segments) as eng:
            eng.set_data(y)
            eng.run(1, True)
            out[name] = eng.marginals() + (eng.free_energy_per_chain(),)
    mm, cmv, fm = out["masked"]
    ms, cs, fs = out["sequential"]
    sd = np.sqrt(np.einsum("tcii->tci", cs))
    assert np.max(np.abs(mm - ms) / sd) < 1e-6
    assert np.max(np.abs(cmv - cs) / (sd[..., :, None] * sd[..., None, :])) < 1e-6
    assert np.allclose(fm, fs, rtol=1e-8, atol=1e-9)
    for c in (0, C - 1):
        one = tuple(a[cm[c]] for a in mdl)
        om, oc, nll = rxo.lgssm_kalman_rts(*one, np.ascontiguousarray(y[:, c]))
        sdo = np.sqrt(np.einsum("tii->ti", oc))
        assert np.max(np.abs(mm[:, c] - om) / sdo) < 1e-6
        assert fm[c] == pytest.approx(nll, rel=1e-8, abs=1e-9)


def _random_step_cases(n, seed):
    rng = np.random.default_rng(seed)
    for i in range(n):
        d = int(rng.choice([5, 8, 16, 24, 32, 48, 64]))
        dy = int(rng.integers(1, d + 1))
        T = int(rng.integers(2, 360))
        yield dict(i=i, d=d, dy=dy, T=T, C=int(rng.choice([1, 1, 2, 5])), M=int(rng.choice([2, 3, 7, max(2, T // 2)])),
                   segments=int(rng.choice([0, 0, 1, 3, 19])) if T > 40 else 0, ptt=bool(rng.integers(0, 2)), rate=float(rng.choice([0.0, 0.1, 0.5])))


@pytest.mark.parametrize("case", list(_random_step_cases(10 + int(os.environ.get("RXHIP_STRESS", "0")), 4242)),
                         ids=lambda c: f"{c['i']}-d{c['d']}x{c['dy']}-T{c['T']}-C{c['C']}-M{c['M']}-s{c['segments']}-{'ptt' if c['ptt'] else 'x1'}-{c['rate']}")
def test_random_per_step_constants_case_against_the_sequential_schedule(case, monkeypatch):
    import rxhip
    d, dy, T, C, M = case["d"], case["dy"], case["T"], case["C"], case["M"]
    mdl = _step_models(d, dy, M, seed=3000 + case["i"])
    rng = np.random.default_rng(case["i"])
    sm = rng.integers(0, M, T).astype(np.int32)
    y = rng.standard_normal((T, C, dy)) * 2.0
    if case["rate"] > 0:
        y[rng.random((T, C)) < case["rate"]] = np.nan
    out = []
    for env in (None, "1"):
        if env:
            monkeypatch.setenv("RXHIP_STEPM_GSEQ", env)
        else:
            monkeypatch.delenv("RXHIP_STEPM_GSEQ", raising=False)
        with rxhip.LGSSMEngine(*mdl, T=T, n_chains=C, prior_through_transition=case["ptt"], step_model=sm, allow_missing=case["rate"] > 0,
                               segments=min(case["segments"], T - 1)) as eng:
            eng.set_data(y)
            eng.run(1, True)
            out.append(eng.marginals() + (eng.free_energy_per_chain(),))
    (mm, cm, fm), (ms, cs, fs) = out
    sd = np.sqrt(np.einsum("tcii->tci", cs))
    assert np.max(np.abs(mm - ms) / sd) < 1e-6
    assert np.max(np.abs(cm - cs) / (sd[..., :, None] * sd[..., None, :])) < 1e-6
    assert np.allclose(fm, fs, rtol=1e-8, atol=1e-9)


@pytest.mark.parametrize("d,dy,T,C,M", [(12, 5, 140, 3, 4), (64, 16, 90, 1, 3)])
def test_per_step_constants_with_known_inputs_missing_values_and_a_forecast_horizon(d, dy, T, C, M, monkeypatch):
    """everything at once on the masked schedule: A[t] … Q[t], `+ c[t]` / `+ d[t]` offsets, missing observations, an unobserved tail"""
    import rxhip
    import rxoracle as rxo
    monkeypatch.delenv("RXHIP_STEPM_GSEQ", raising=False)
    H = 7
    mdl = _step_models(d, dy, M, seed=80 + d)
    rng = np.random.default_rng(T + d)
    sm = rng.integers(0, M, T + H).astype(np.int32)
    cx, cy = rng.standard_normal((T + H, d)), rng.standard_normal((T + H, dy))
    y = rng.standard_normal((T, C, dy)) * 2.0
    y[rng.random((T, C)) < 0.2] = np.nan
    with rxhip.LGSSMEngine(*mdl, T=T, n_chains=C, step_model=sm, horizon=H, allow_missing=True, state_offset=cx, obs_offset=cy) as eng:
        eng.set_data(y)
        eng.run(1, True)
        mean, cov = eng.marginals()
        fe = eng.free_energy_per_chain()
    for c in range(C):
        yy = np.concatenate([y[:, c], np.full((H, dy), np.nan)])
        om, oc, nll = rxo.lgssm_kalman_rts_affine(*mdl, np.ascontiguousarray(yy), state_offset=cx, obs_offset=cy, step_model=sm)
        sd = np.sqrt(np.einsum("tii->ti", oc))
        assert np.max(np.abs(mean[:, c] - om) / sd) < 1e-6
        assert np.max(np.abs(cov[:, c] - oc) / (sd[:, :, None] * sd[:, None, :])) < 1e-6
        assert fe[c] == pytest.approx(nll, rel=1e-8, abs=1e-9)


@pytest.mark.parametrize("d,dy,T,C,M,rate,segments", [(24, 6, 200, 2, 1, 0.2, 0), (64, 64, 130, 1, 1, 0.1, 0), (16, 16, 300, 300, 1, 0.3, 0), (40, 12, 150, 2, 3, 0.15, 4), (9, 20, 90, 3, 5, 0.0, 0)])
def test_filtering_runs_on_the_masked_schedule(d, dy, T, C, M, rate, segments, monkeypatch):
    """rxhip_run_filter of masked / per-step engines: the filtered moments come out of the forward sweep's records (km_filter_out), the free
    energy is −log p(y) / T — against the sequential schedule (RXHIP_FILTER_GSEQ) and the oracle's filter"""
    import rxhip
    import rxoracle as rxo
    mdl = _step_models(d, dy, M, seed=60 + d)
    rng = np.random.default_rng(T + d)
    sm = rng.integers(0, M, T).astype(np.int32) if M > 1 else None
    y = rng.standard_normal((T, C, dy)) * 2.0
    if rate > 0:
        y[rng.random((T, C)) < rate] = np.nan
    args = mdl if M > 1 else tuple(a[0] for a in mdl)
    out = []
    for env in (None, "1"):
        if env:
            monkeypatch.setenv("RXHIP_FILTER_GSEQ", env)
        else:
            monkeypatch.delenv("RXHIP_FILTER_GSEQ", raising=False)
        with rxhip.LGSSMEngine(*args, T=T, n_chains=C, step_model=sm, allow_missing=rate > 0, segments=segments) as eng:
            eng.set_data(y)
            eng.run_filter(True)
            out.append(eng.marginals() + (eng.free_energy_per_chain(),))
            eng.run(1, True)                       # and a smoothing run afterwards is the smoothing run
            sm_mean = eng.marginals()[0]
            eng.run_filter(False)                  # no free energy: forward sweep only
            assert np.allclose(eng.marginals()[0], out[-1][0], rtol=0, atol=0)
            assert not np.allclose(sm_mean[: T // 2], out[-1][0][: T // 2])
    (mm, cm, fm), (ms, cs, fs) = out
    sd = np.sqrt(np.einsum("tcii->tci", cs))
    assert np.max(np.abs(mm - ms) / sd) < 1e-6
    assert np.max(np.abs(cm - cs) / (sd[..., :, None] * sd[..., None, :])) < 1e-6
    assert np.allclose(fm, fs, rtol=1e-8, atol=1e-9)


@pytest.mark.parametrize("d,dy,T,C,M,rate", [(64, 64, 150, 1, 1, 0.0), (24, 6, 200, 2, 1, 0.2), (40, 12, 120, 2, 3, 0.1), (20, 20, 300, 1, 1, 0.0)])
def test_node_local_joints_from_the_sweep_records(d, dy, T, C, M, rate, monkeypatch):
    """q(x[t], x[t+1] | y) of the transition nodes after a sweep of the information-form kernels: Cov(x[t], x[t+1] | y) = G_t V_s(t+1) from
    the gains the forward sweep left in its records (kd_cross_from_records) against the sequential re-run (RXHIP_JOINTS_GSEQ)"""
    import rxhip
    mdl = _step_models(d, dy, M, seed=20 + d)
    rng = np.random.default_rng(T)
    sm = rng.integers(0, M, T).astype(np.int32) if M > 1 else None
    y = rng.standard_normal((T, C, dy)) * 2.0
    if rate > 0:
        y[rng.random((T, C)) < rate] = np.nan
    args = mdl if M > 1 else tuple(a[0] for a in mdl)
    out = []
    for env in (None, "1"):
        if env:
            monkeypatch.setenv("RXHIP_JOINTS_GSEQ", env)
        else:
            monkeypatch.delenv("RXHIP_JOINTS_GSEQ", raising=False)
        with rxhip.LGSSMEngine(*args, T=T, n_chains=C, step_model=sm, allow_missing=rate > 0) as eng:
            eng.set_data(y)
            eng.run(1, True)
            out.append(eng.node_marginals())
    (jm, jc), (sm_, sc) = out
    sd = np.sqrt(np.einsum("tcii->tci", sc))
    assert np.max(np.abs(jm - sm_) / sd) < 1e-6
    assert np.max(np.abs(jc - sc) / (sd[..., :, None] * sd[..., None, :])) < 1e-6


def test_records_that_do_not_fit_keep_the_sequential_schedule(monkeypatch):
    """The masked schedule keeps per-step records for every chain (4.5 KB per chain-step at d ≤ 16): an engine whose block does not fit
    the device stays on the sequential kernels instead of failing at creation.  RXHIP_MSEG_MAX_BYTES caps the block."""
    import rxhip
    import rxoracle as rxo
    from rxhip import workloads
    d, dy, T, C = 8, 4, 90, 3
    mdl = workloads.random_model(d, dy, seed=77)
    y = workloads.generate_batch(mdl, T, C, seed0=3)
    y[np.random.default_rng(5).random((T, C)) < 0.2] = np.nan
    monkeypatch.delenv("RXHIP_GSEQ", raising=False)
    res = {}
    for cap in (None, "4096"):
        if cap:
            monkeypatch.setenv("RXHIP_MSEG_MAX_BYTES", cap)
        with rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=C, allow_missing=True) as eng:
            eng.set_data(y)
            eng.run(1, True)
            res[cap] = (*eng.marginals(), eng.free_energy_per_chain(), eng.schedule())
    assert res[None][3]["segments"] >= 1 and res["4096"][3]["segments"] == 0       # time-parallel / sequential (no segments)
    for c in range(C):
        om, oc, nll = rxo.lgssm_kalman_rts(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], np.ascontiguousarray(y[:, c]))
        sd = np.sqrt(np.einsum("tii->ti", oc))
        for cap in res:
            mean, cov, fe, _ = res[cap]
            assert np.max(np.abs(mean[:, c] - om) / sd) < 1e-6 and np.max(np.abs(cov[:, c] - oc) / (sd[:, :, None] * sd[:, None, :])) < 1e-6, cap
            assert fe[c] == pytest.approx(nll, rel=1e-8, abs=1e-9), cap


def test_more_chains_than_a_grid_dimension_on_the_masked_schedule(monkeypatch):
    """grid.y holds 65 535 blocks: a masked batch beyond that is launched in slices of 32 768 chains.  70 000 chains of a
    tiny problem: a handful of them against the oracle, and chains that carry the same data must agree bit for bit across slices."""
    import rxhip
    import rxoracle as rxo
    from rxhip import workloads
    d, dy, T, C = 6, 3, 5, 70000
    mdl = workloads.random_model(d, dy, seed=78)
    base = workloads.generate_batch(mdl, T, 7, seed0=9)
    base[1, 2] = np.nan
    base[3, 5] = np.nan
    y = np.ascontiguousarray(np.tile(base, (1, C // 7, 1)))
    monkeypatch.delenv("RXHIP_GSEQ", raising=False)
    with rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=C, allow_missing=True) as eng:
        eng.set_data(y)
        eng.run(1, True)
        chains = [0, 2, 5, 32767, 32768, 65535, 65536, C - 1]
        mean, cov = eng.marginals_of_chains(chains)
        fe = eng.free_energy_per_chain()
    for i, c in enumerate(chains):
        om, oc, nll = rxo.lgssm_kalman_rts(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], np.ascontiguousarray(y[:, c]))
        sd = np.sqrt(np.einsum("tii->ti", oc))
        assert np.max(np.abs(mean[i] - om) / sd) < 1e-6 and np.max(np.abs(cov[i] - oc) / (sd[:, :, None] * sd[:, None, :])) < 1e-6, c
        assert fe[c] == pytest.approx(nll, rel=1e-8, abs=1e-9), c
    assert np.array_equal(fe[:7], fe[C - 7:]) and np.array_equal(fe[32767 - 32767 % 7:32767 - 32767 % 7 + 7], fe[:7])
