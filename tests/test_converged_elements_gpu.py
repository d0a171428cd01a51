"""Per-chain, time-invariant models at d, dy ≤ 4 on long segments (lgssm_kernels.hpp): k_seg_elements stops its matrix recursion where the known-start
filter's covariance has reached its fixed point and k_seg_elements_tail runs the frozen recursion; k_forward_tinv writes mean-only records behind the
fixed point of V_f and k_backward_tinv reads them (the covariance from the segment's last record).  Against the full recursions (RXHIP_ELEM_FULL=1) and
the oracle, with models whose filters settle at very different speeds inside one wavefront, segments shorter and much longer than the settling time,
and the node-local joints (which read the forward records) on top."""
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


@pytest.mark.parametrize("d,dy,T,C,segments", [(4, 4, 6000, 64, 4), (4, 4, 3000, 70, 0), (3, 2, 5000, 64, 2), (2, 1, 4000, 128, 3), (1, 1, 9000, 64, 2), (4, 2, 900, 64, 30),
                                                 (3, 3, 1500, 64, 3), (1, 1, 1200, 128, 2), (3, 1, 2100, 64, 7)])
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
            res.append((eng.marginals(), eng.free_energy_per_chain(), eng.schedule(), eng.node_marginals() if T <= 4000 else None))
    (m1, c1), f1, sched, j1 = res[0]
    (m2, c2), f2, _, j2 = res[1]
    if j1 is not None:   # the node-local joints read V_f(t) from the forward records: mean-only behind the fixed point, the segment's last record has it
        assert np.allclose(j1[0], j2[0], rtol=1e-9, atol=1e-11) and np.allclose(j1[1], j2[1], rtol=1e-8, atol=1e-11)
    assert sched["segments"] > 1
    sd = np.sqrt(np.einsum("tcii->tci", c2))
    assert np.max(np.abs(m1 - m2) / sd) < 1e-11 and np.max(np.abs(c1 - c2) / (sd[..., :, None] * sd[..., None, :])) < 1e-11
    assert np.max(np.abs(f1 - f2) / np.abs(f2)) < 1e-12
    for c in (0, C // 2, C - 1):
        om, oc, nll = rxo.lgssm_kalman_rts(ms[c]["A"], ms[c]["B"], ms[c]["P"], ms[c]["Q"], ms[c]["m0"], ms[c]["V0"], np.ascontiguousarray(y[:, c]))
        s = np.sqrt(np.einsum("tii->ti", oc))
        assert np.max(np.abs(m1[:, c] - om) / s) < 1e-8 and abs(f1[c] - nll) < 1e-9 * abs(nll), c


def test_runs_of_every_kind_on_one_engine(monkeypatch):
    """mean-only records are a property of the LAST smoothing run: filtering runs, repeated iterations and the node-local joints in between"""
    import rxhip
    from rxhip import workloads
    d, dy, T, C = 4, 2, 2400, 64
    ms = _models(d, dy, C, seed=77)
    y = np.stack([workloads.generate_batch(ms[c], T, 1, seed0=c)[:, 0] for c in range(C)], axis=1)
    stack = lambda k: np.stack([m[k] for m in ms])
    out = []
    for full in (None, "1"):
        if full:
            monkeypatch.setenv("RXHIP_ELEM_FULL", full)
        else:
            monkeypatch.delenv("RXHIP_ELEM_FULL", raising=False)
        with rxhip.LGSSMEngine(stack("A"), stack("B"), stack("P"), stack("Q"), stack("m0"), stack("V0"), T=T, n_chains=C, chain_model=np.arange(C, dtype=np.int32), segments=4) as eng:
            eng.set_data(y)
            eng.run(2, True)
            a = (eng.marginals()[0], eng.free_energy(), eng.node_marginals()[1])
            eng.run_filter(True)
            b = (eng.marginals()[0], eng.free_energy_per_chain())
            eng.run(1, False)
            c = (eng.marginals()[1], eng.node_marginals()[0])
            out.append((a, b, c))
    for x, z in zip(out[0], out[1]):
        for u, v in zip(x, z):
            assert np.allclose(u, v, rtol=1e-9, atol=1e-11)
