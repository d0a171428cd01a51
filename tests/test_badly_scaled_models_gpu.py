"""State components that live on very different scales (metres next to micro-radians): S A S⁻¹, S P S, S V0 S, B S⁻¹ with
S = diag(10^u), u up to ±3 — the matrices the MFMA path inverts then carry twelve decades between their diagonal entries.  The
panel inverse equilibrates by exact powers of two (dense_kernels.hpp, blk_inverse); the round-1/2 sweep inverse lost every digit on
such input (profiles/r03/inv_micro.txt) and no parity test noticed, because every test model was well scaled.  Compared component by
component on the scale of each posterior standard deviation."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scaled(d, dy, decades, seed):
    from rxhip import workloads
    m = workloads.random_model(d, dy, seed=seed)
    s = 10.0 ** np.random.default_rng(seed + 1).uniform(-decades, decades, d)
    S, Si = np.diag(s), np.diag(1.0 / s)
    return dict(A=S @ m["A"] @ Si, B=m["B"] @ Si, P=S @ m["P"] @ S, Q=m["Q"], m0=s * m["m0"], V0=S @ m["V0"] @ S), m, s


@pytest.mark.parametrize("d,dy,T,C,decades", [(4, 4, 400, 64, 3.0), (2, 2, 300, 3, 3.0), (3, 1, 200, 70, 2.0), (8, 8, 200, 64, 3.0), (16, 16, 150, 2, 3.0), (24, 7, 90, 3, 2.0), (48, 48, 60, 1, 3.0), (64, 64, 120, 1, 3.0), (64, 20, 70, 5, 1.5)])
def test_scaled_state_components(d, dy, T, C, decades):
    import rxhip
    import rxoracle as rxo
    from rxhip import workloads
    ms, m, s = _scaled(d, dy, decades, seed=7 * d + dy)
    y = workloads.generate_batch(m, T, C, seed0=2)            # the observations do not change with the state scaling
    with rxhip.LGSSMEngine(ms["A"], ms["B"], ms["P"], ms["Q"], ms["m0"], ms["V0"], T=T, n_chains=C) as eng:
        eng.set_data(y)
        eng.run(1, True)
        mean, cov = eng.marginals()
        fe = eng.free_energy_per_chain()
    for c in range(C):
        om, oc, nll = rxo.lgssm_kalman_rts(m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"], np.ascontiguousarray(y[:, c]))
        sd = np.sqrt(np.einsum("tii->ti", oc))
        em = np.max(np.abs(mean[:, c] / s - om) / sd)                                     # back on the well-scaled problem's axes
        ec = np.max(np.abs(cov[:, c] / (s[:, None] * s[None, :]) - oc) / (sd[:, :, None] * sd[:, None, :]))
        assert em < 1e-6 and ec < 1e-6, (c, em, ec)
        assert fe[c] == pytest.approx(nll, rel=1e-8, abs=1e-9), c                          # −log p(y) does not depend on the state's units


@pytest.mark.parametrize("d,dy,T,C,gseq", [(24, 7, 150, 1, False), (64, 64, 90, 1, False), (16, 16, 120, 300, False), (24, 7, 80, 2, True), (4, 4, 300, 64, False)])
def test_scaled_state_components_with_missing_observations(d, dy, T, C, gseq, monkeypatch):
    """the same on the masked schedules: parallel in time with device-built elements (one chain), one segment per chain (chain-rich batch),
    the sequential coverage schedule (RXHIP_GSEQ), and the d ≤ 4 masked sweep"""
    import rxhip
    import rxoracle as rxo
    from rxhip import workloads
    if gseq:
        monkeypatch.setenv("RXHIP_GSEQ", "1")
    else:
        monkeypatch.delenv("RXHIP_GSEQ", raising=False)
    ms, m, s = _scaled(d, dy, 2.5, seed=11 * d + dy)
    y = workloads.generate_batch(m, T, min(C, 4), seed0=5)
    y[np.random.default_rng(3).random((T, min(C, 4))) < 0.15] = np.nan
    y = np.tile(y, (1, (C + 3) // 4, 1))[:, :C]
    with rxhip.LGSSMEngine(ms["A"], ms["B"], ms["P"], ms["Q"], ms["m0"], ms["V0"], T=T, n_chains=C, allow_missing=True) as eng:
        eng.set_data(y)
        eng.run(1, True)
        mean, cov = eng.marginals()
        fe = eng.free_energy_per_chain()
    for c in (0, C - 1):
        om, oc, nll = rxo.lgssm_kalman_rts(m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"], np.ascontiguousarray(y[:, c]))
        sd = np.sqrt(np.einsum("tii->ti", oc))
        em = np.max(np.abs(mean[:, c] / s - om) / sd)
        ec = np.max(np.abs(cov[:, c] / (s[:, None] * s[None, :]) - oc) / (sd[:, :, None] * sd[:, None, :]))
        assert em < 1e-6 and ec < 1e-6, (c, em, ec)
        assert fe[c] == pytest.approx(nll, rel=1e-8, abs=1e-9), c
