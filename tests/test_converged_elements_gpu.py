"""k_seg_elements (per-chain models at d, dy ≤ 4) stops its matrix recursion where the known-start filter's covariance has reached its fixed point and
runs the frozen recursion over the rest of the segment (lgssm_kernels.hpp).  Against the full recursion (RXHIP_ELEM_FULL=1) and against the oracle,
with models whose filters settle at very different speeds inside one wavefront, segments shorter and much longer than the settling time."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _models(d, dy, C, seed):
    from rxhip import workloads
    rng = np.random.default_rng(seed)
    out = []
    for c in range(C):
        m = workloads.random_model(d, dy, seed=seed + c)
        # spread the mixing times: scale the state noise over four decades and the spectral radius between 0.5 and 0.999
        rho = 0.5 + 0.499 * rng.random()
        ev = np.max(np.abs(np.linalg.eigvals(m["A"])))
        out.append(dict(m, A=m["A"] * (rho / ev), P=m["P"] * 10.0 ** rng.uniform(-3, 1)))
    return out


@pytest.mark.parametrize("d,dy,T,C,segments", [(4, 4, 6000, 64, 4), (4, 4, 3000, 70, 0), (3, 2, 5000, 64, 2), (2, 1, 4000, 128, 3), (1, 1, 9000, 64, 2), (4, 2, 900, 64, 30)])
def test_frozen_tail_against_the_full_recursion_and_the_oracle(d, dy, T, C, segments, monkeypatch):
    import rxhip
    import rxoracle as rxo
    from rxhip import workloads
    ms = _models(d, dy, C, seed=1000 * d + dy)
    y = np.stack([workloads.generate_batch(ms[c], T, 1, seed0=c)[:, 0] for c in range(C)], axis=1)
    stack = lambda k: np.stack([m[k] for m in ms])
    res = []
    for full in (None, "1"):
        if full:
            monkeypatch.setenv("RXHIP_ELEM_FULL", full)
        else:
            monkeypatch.delenv("RXHIP_ELEM_FULL", raising=False)
        with rxhip.LGSSMEngine(stack("A"), stack("B"), stack("P"), stack("Q"), stack("m0"), stack("V0"), T=T, n_chains=C, chain_model=np.arange(C, dtype=np.int32),
                               segments=segments) as eng:
            eng.set_data(y)
            eng.run(1, True)
            res.append((eng.marginals(), eng.free_energy_per_chain(), eng.schedule()))
    (m1, c1), f1, sched = res[0]
    (m2, c2), f2, _ = res[1]
    assert sched["segments"] > 1
    sd = np.sqrt(np.einsum("tcii->tci", c2))
    assert np.max(np.abs(m1 - m2) / sd) < 1e-11 and np.max(np.abs(c1 - c2) / (sd[..., :, None] * sd[..., None, :])) < 1e-11
    assert np.max(np.abs(f1 - f2) / np.abs(f2)) < 1e-12
    for c in (0, C // 2, C - 1):
        om, oc, nll = rxo.lgssm_kalman_rts(ms[c]["A"], ms[c]["B"], ms[c]["P"], ms[c]["Q"], ms[c]["m0"], ms[c]["V0"], np.ascontiguousarray(y[:, c]))
        s = np.sqrt(np.einsum("tii->ti", oc))
        assert np.max(np.abs(m1[:, c] - om) / s) < 1e-8 and abs(f1[c] - nll) < 1e-9 * abs(nll), c
