"""Parity at the FULL sizes of BASELINE.json's configs (north_star: "... with matching posteriors"): the HIP path on the
default schedule of each headline workload against the CPU oracle on the same observations.  The batch results stay on the
device (20 GB at C2); the chains / series that are compared are gathered there (rxhip_get_marginals_chains) and checked
over their whole length: posterior means / covariances 1e-6 relative, free energy 1e-8 relative (BASELINE.json)."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import rxhip
import rxoracle
from rxhip import workloads

pytestmark = pytest.mark.gpu

RTOL_POST, RTOL_FE = 1e-6, 1e-8


def rel(a, b):
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


def step_rel(mean, cov, om, oc):
    """element-wise in time: every posterior on the scale of ITS OWN step — a mean error in posterior standard deviations, a
    covariance error relative to the largest entry of that step's covariance (not the max norm over the whole chain)"""
    sd = np.sqrt(np.einsum("tii->ti", oc))
    return (float(np.max(np.abs(mean - om) / sd)), float(np.max(np.abs(cov - oc) / np.max(np.abs(oc), axis=(1, 2), keepdims=True))))


def test_c2_full_size():
    """C2 exactly as bench.py runs it: d = dy = 4, T = 100 000, 1024 chains (chain c from default_rng(42 + c)), automatic
    schedule (S = 128 segments of 782 steps, two waves per SIMD, table-driven boundary scan with 64-segment LDS chunks).
    Checked chains: first / last lane of the first and last wavefront, the two lanes either side of a wavefront boundary in
    the middle, and two more; every segment boundary of each is inside the comparison."""
    mdl = workloads.c1_model()
    T, C = 100000, 1024
    y = workloads.generate_batch(mdl, T, C, seed0=42)
    chains = [0, 63, 64, 511, 512, 777, 960, 1023]
    with rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=C) as eng:
        sched = eng.schedule()
        assert sched["segments"] == 128 and sched["segment_len"] == 782  # the schedule the headline number is measured on
        eng.set_data(y)
        eng.run(1, True)
        mean, cov = eng.marginals_of_chains(chains)
        fe = eng.free_energy_per_chain()
        fe_total = eng.free_energy()[0]
        cnt = eng.counters()
    assert cnt["rule_calls"] == C * (6 * T - 3)
    with ThreadPoolExecutor(8) as ex:
        ref = list(ex.map(lambda c: rxoracle.lgssm_bp(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], y[:, c]), chains))
    for i, c in enumerate(chains):
        om, oc, ofe, _ = ref[i]
        assert rel(mean[i], om) < RTOL_POST, (c, rel(mean[i], om))
        assert rel(cov[i], oc) < RTOL_POST, (c, rel(cov[i], oc))
        # per step, not only in the max norm of the whole chain: every posterior on the scale of its own step
        em, ec = step_rel(mean[i], cov[i], om, oc)
        assert em < RTOL_POST and ec < RTOL_POST, (c, em, ec)
        assert abs(fe[c] - ofe) < RTOL_FE * abs(ofe), (c, fe[c], ofe)
    # the batch free energy is the fixed-order sum of the per-chain values
    assert abs(fe_total - np.sum(fe)) < 1e-12 * abs(fe_total)
    # all chains of the shared model carry the same covariances; spot the whole batch through one more gather
    m2, c2 = None, None
    with rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=C) as eng:
        eng.set_data(y)
        eng.run(1, True)
        m2, c2 = eng.marginals_of_chains(chains[:2])
    assert np.array_equal(m2, mean[:2]) and np.array_equal(c2, cov[:2])  # bit-identical from run to run / engine to engine


def test_c2_full_size_per_chain_models():
    """The batch bench.py reports as `roofline_per_chain_models`: the C2 data with one constant set PER CHAIN (n_models = n_chains;
    every chain its own model — here perturbed per chain, so that nothing can be shared by accident), on the schedule that
    workload runs on (segment elements computed in the lane, per-chain records).  SURVEY's 416 B/U applies to it unmodified."""
    mdl = workloads.c1_model()
    T, C = 100000, 1024
    y = workloads.generate_batch(mdl, T, C, seed0=42)
    rng = np.random.default_rng(5)
    scale = 1.0 + 0.2 * rng.random(C)
    tile = lambda a: np.broadcast_to(np.asarray(a, dtype=np.float64), (C,) + np.shape(a)).copy()
    A, B, P, Q, m0, V0 = (tile(mdl[k]) for k in ("A", "B", "P", "Q", "m0", "V0"))
    P *= scale[:, None, None]
    Q /= scale[:, None, None]
    chains = [0, 63, 64, 700, 1023]
    with rxhip.LGSSMEngine(A, B, P, Q, m0, V0, T=T, n_chains=C, chain_model=np.arange(C, dtype=np.int32)) as eng:
        sched = eng.schedule()
        assert sched["segments"] == 128
        eng.set_data(y)
        eng.run(1, True)
        mean, cov = eng.marginals_of_chains(chains)
        fe = eng.free_energy_per_chain()
    with ThreadPoolExecutor(8) as ex:
        ref = list(ex.map(lambda c: rxoracle.lgssm_bp(A[c], B[c], P[c], Q[c], m0[c], V0[c], y[:, c]), chains))
    for i, c in enumerate(chains):
        om, oc, ofe, _ = ref[i]
        em, ec = step_rel(mean[i], cov[i], om, oc)
        assert em < RTOL_POST and ec < RTOL_POST, (c, em, ec)
        assert abs(fe[c] - ofe) < RTOL_FE * abs(ofe), (c, fe[c], ofe)


def test_c2_full_size_with_missing_observations():
    """The batch bench.py reports as `extra.c2_missing`: 10 % of the observations of the C2 batch `missing` (NaN), the masked
    time-parallel schedule at S = 128 segments — boundary scan over elements computed in the lane included.  Checker: the
    smoother with skipped updates (docs/src/manuals/inference/static.md:98-123; pinned to brute-force conditioning of the joint
    Gaussian in tests/test_missing_observations.py)."""
    mdl = workloads.c1_model()
    T, C = 100000, 1024
    y = workloads.generate_batch(mdl, T, C, seed0=42)
    rng = np.random.default_rng(0)
    gone = rng.random((T, C)) < 0.1
    gone[5000:5900, 64] = True      # more than a whole segment (782 steps) without a single observation
    y[gone] = np.nan
    chains = [0, 64, 511, 1023]
    with rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=C, allow_missing=True) as eng:
        assert eng.schedule()["segments"] == 128
        eng.set_data(y)
        eng.run(1, True)
        mean, cov = eng.marginals_of_chains(chains)
        fe = eng.free_energy_per_chain()
    with ThreadPoolExecutor(4) as ex:
        ref = list(ex.map(lambda c: rxoracle.lgssm_kalman_rts(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], np.ascontiguousarray(y[:, c])), chains))
    for i, c in enumerate(chains):
        om, oc, nll = ref[i]
        em, ec = step_rel(mean[i], cov[i], om, oc)
        assert em < RTOL_POST and ec < RTOL_POST, (c, em, ec)
        assert abs(fe[c] - nll) < RTOL_FE * abs(nll), (c, fe[c], nll)


def test_c2_full_size_filtering():
    """The streaming twin at the same size (rxhip_run_filter): q(x_t | y_1..t) and the mean-over-observations free energy."""
    mdl = workloads.c1_model()
    T, C = 100000, 1024
    y = workloads.generate_batch(mdl, T, C, seed0=42)
    chains = [0, 64, 1023]
    with rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=C) as eng:
        eng.set_data(y)
        eng.run_filter(True)
        mean, cov = eng.marginals_of_chains(chains)
        fe = eng.free_energy_per_chain()
    for i, c in enumerate(chains):
        om, oc, ofe, _ = rxoracle.lgssm_filter(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], y[:, c],
                                               prior_through_transition=False)
        assert rel(mean[i], om) < RTOL_POST and rel(cov[i], oc) < RTOL_POST
        assert abs(fe[c] - ofe) < RTOL_FE * abs(ofe)


def test_c4_full_size():
    """C4 on one GPU: 4096 HGF series × T = 2000, 10 VMP iterations per observation, GH-31."""
    S, T, iters = 4096, 2000, 10
    _, _, y = workloads.generate_hgf_batch(T, S, seed=42)
    with rxhip.HGFEngine(T, S, 1.0, 0.0, 0.04, 0.01) as eng:
        eng.set_data(y)
        eng.run(iters, True)
        zm, zv, xm, xv = eng.history()
        fe_series = eng.free_energy_per_chain()
        fe = eng.free_energy()
    for s in (0, 3, 4, 2047, 4095):  # first 16-lane row, a row boundary, middle, last
        o = rxoracle.hgf_filter(y[:, s], 1.0, 0.0, 0.04, 0.01, vmp_iters=iters)
        for got, want in ((zm[:, s], o[0]), (zv[:, s], o[1]), (xm[:, s], o[2]), (xv[:, s], o[3])):
            assert rel(got, want) < RTOL_POST, s
        assert abs(fe_series[s] - o[4][-1]) < RTOL_FE * abs(o[4][-1]), s
    assert abs(fe[-1] - np.sum(fe_series)) < 1e-11 * abs(fe[-1])


def test_c5_full_size():
    """C5 on one GPU: univariate mixture, K = 16, N = 10^7, 20 VMP iterations.  The oracle runs in its split-phase form
    (rxo_gmm_accumulate over 64 shards on host threads, statistics summed in shard order, rxo_gmm_update) — the same
    arithmetic as rxo_gmm_vmp, which would take minutes on one core."""
    K, N, iters = 16, 10_000_000, 20
    mus = np.arange(1, K + 1) * 10.0 - 80.0
    rng = np.random.default_rng(12345)
    y = mus[rng.integers(0, K, size=N)] + rng.standard_normal(N)
    priors = (mus + 1.5, np.full(K, 1e3), np.full(K, 0.01), np.full(K, 0.01), np.ones(K))
    init = (mus + 1.5, np.full(K, 10.0), np.ones(K), np.ones(K), np.ones(K))
    with rxhip.GMMEngine(N, *priors, *init) as eng:
        eng.set_data(y)
        eng.run(iters, True)
        hist, fe = eng.history(), eng.free_energy()
    state = np.ascontiguousarray(np.stack([np.asarray(a, dtype=np.float64) for a in init]))  # [5][K]
    shards = np.array_split(y, 64)
    ohist, ofe = np.empty((iters, 5, K)), np.empty(iters)
    with ThreadPoolExecutor(16) as ex:
        for it in range(iters):
            parts = list(ex.map(lambda sh: rxoracle.gmm_accumulate(sh, state.copy()), shards))
            stats = np.sum(np.stack(parts), axis=0)
            ofe[it] = rxoracle.gmm_update(*priors, stats, state)
            ohist[it] = state
    assert np.max(np.abs(hist - ohist) / np.maximum(np.abs(ohist), 1e-300)) < RTOL_POST
    assert np.max(np.abs(fe - ofe) / np.abs(ofe)) < RTOL_FE
    assert np.all(np.diff(fe) <= 1e-9 * abs(fe[-1]))  # non-increasing, gmm_univariate_tests.jl:96
    assert np.all(np.abs(hist[-1, 0] - mus) < 0.1)
