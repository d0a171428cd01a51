"""The masked sweep of d ≤ 8 chains inside one wavefront per chain (csrc/dense8_kernels.hpp): one segment per chain, an 8×8 matrix one
element per lane, inverse by the symmetric sweep operator.  Checked against the MFMA kernels it replaces on this schedule (RXHIP_WAVE8=0)
and against the oracle's smoother with skipped updates (pinned to brute-force conditioning, tests/test_missing_observations.py).
Reference behaviour: `missing` observations, docs/src/manuals/inference/static.md:98-123, test/inference/prediction_tests.jl:197-213."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rxinfer.jl_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

pytestmark = pytest.mark.gpu


def _run(mdl, y, ptt, wave8, monkeypatch, fe=True, segments=1):
    import rxhip
    monkeypatch.setenv("RXHIP_WAVE8", "1" if wave8 else "0")
    monkeypatch.delenv("RXHIP_GSEQ", raising=False)
    T, C = y.shape[0], y.shape[1]
    with rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=C, prior_through_transition=ptt,
                           allow_missing=True, segments=segments) as eng:
        eng.set_data(y)
        eng.run(1, fe)
        mean, cov = eng.marginals()
        f = eng.free_energy_per_chain() if fe else None
        eng.run(2, fe)                        # iterations re-push the data: same result
        assert np.array_equal(eng.marginals()[0], mean)
        return mean, cov, f, eng.schedule()


@pytest.mark.parametrize("d,dy,T,C,ptt,rate", [(8, 4, 300, 5, False, 0.1), (8, 8, 120, 3, True, 0.3), (5, 3, 90, 4, False, 0.2), (6, 12, 70, 2, True, 0.15),
                                                (7, 1, 200, 3, False, 0.5), (8, 40, 50, 2, False, 0.1), (8, 4, 2, 3, False, 0.0), (5, 5, 64, 130, True, 0.25)])
def test_in_wave_sweep_against_the_mfma_kernels_and_the_oracle(d, dy, T, C, ptt, rate, monkeypatch):
    import rxoracle as rxo
    from rxhip import workloads
    mdl = workloads.random_model(d, dy, seed=500 + 7 * d + dy)
    y = workloads.generate_batch(mdl, T, C, seed0=31)
    rng = np.random.default_rng(d * T + C)
    y[rng.random((T, C)) < rate] = np.nan
    if T > 4:
        y[0, 0] = np.nan                       # the first observation of a chain
        y[-1, C - 1] = np.nan                  # the last one
        y[T // 3:T // 3 + 7, 1 % C] = np.nan   # a run of missing steps
    m8, c8, f8, sched = _run(mdl, y, ptt, True, monkeypatch)
    mm, cm, fm, _ = _run(mdl, y, ptt, False, monkeypatch)
    assert sched["segments"] == 1
    for c in range(min(C, 6)):
        om, oc, nll = rxo.lgssm_kalman_rts(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], np.ascontiguousarray(y[:, c]), prior_through_transition=ptt)
        sd = np.sqrt(np.einsum("tii->ti", oc))
        for name, mean, cov, fe in (("in-wave", m8, c8, f8), ("mfma", mm, cm, fm)):
            assert np.max(np.abs(mean[:, c] - om) / sd) < 1e-6 and np.max(np.abs(cov[:, c] - oc) / (sd[:, :, None] * sd[:, None, :])) < 1e-6, (name, c)
            assert fe[c] == pytest.approx(nll, rel=1e-8, abs=1e-9), (name, c)
    sdm = np.sqrt(np.einsum("tcii->tci", cm))
    assert np.max(np.abs(m8 - mm) / sdm) < 1e-9 and np.max(np.abs(f8 - fm) / np.abs(fm)) < 1e-11     # every chain against the MFMA kernels


def test_without_free_energy_and_the_default_schedule_of_a_full_batch(monkeypatch):
    """1024 chains fill the chip: the cost model takes one segment per chain (and with it the in-wave kernels) on its own."""
    from rxhip import workloads
    mdl = workloads.random_model(8, 4, seed=8)
    y = np.tile(workloads.generate_batch(mdl, 200, 8, seed0=1), (1, 128, 1))
    y[np.random.default_rng(0).random((200, 1024)) < 0.1] = np.nan
    m8, c8, _, sched = _run(mdl, y, False, True, monkeypatch, fe=False, segments=0)
    mm, cm, _, _ = _run(mdl, y, False, False, monkeypatch, fe=False, segments=1)
    assert sched["segments"] == 1
    sdm = np.sqrt(np.einsum("tcii->tci", cm))
    assert np.max(np.abs(m8 - mm) / sdm) < 1e-9 and np.max(np.abs(c8 - cm)) < 1e-9 * np.max(np.abs(cm))


def test_badly_scaled_states(monkeypatch):
    """State components six decades apart (the case the rank-4 sweep of rounds 1–2 lost every digit on, tests/test_badly_scaled_models_gpu.py)."""
    import rxoracle as rxo
    from rxhip import workloads
    d, dy, T = 8, 4, 150
    mdl = workloads.random_model(d, dy, seed=41)
    sc = np.logspace(-3, 3, d)
    S, Si = np.diag(sc), np.diag(1.0 / sc)
    bad = dict(A=S @ mdl["A"] @ Si, B=mdl["B"] @ Si, P=S @ mdl["P"] @ S, Q=mdl["Q"], m0=S @ mdl["m0"], V0=S @ mdl["V0"] @ S)
    y = workloads.generate_batch(mdl, T, 2, seed0=5)
    y[np.random.default_rng(2).random((T, 2)) < 0.2] = np.nan
    m8, c8, f8, _ = _run(bad, y, False, True, monkeypatch)
    for c in range(2):
        om, oc, nll = rxo.lgssm_kalman_rts(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], np.ascontiguousarray(y[:, c]))
        sd = np.sqrt(np.einsum("tii->ti", oc))
        assert np.max(np.abs(m8[:, c] / sc - om) / sd) < 1e-6
        assert np.max(np.abs(c8[:, c] / (sc[:, None] * sc[None, :]) - oc) / (sd[:, :, None] * sd[:, None, :])) < 1e-6
        assert f8[c] == pytest.approx(nll, rel=1e-8, abs=1e-9)
