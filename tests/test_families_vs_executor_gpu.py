"""The pattern-matched state-space engines against the node-array executor on the SAME graph descriptors: random chains with everything the chain
lowering accepts — prior on the first state or through a transition, per-step constants A[t], P[t], B[t], Q[t], known inputs `A x + c[t]`, `B x + d[t]`
(constant on either side of `+`), data inputs `+ B_u u[t]` — built by rxhip.graph.lgssm_graph, run once through `rxhip_create` (lowering + a fused
parallel-in-time engine) and once through `rxhip_tree_create` (graph compiler + one rule per op).  Two implementations that share no kernel and no
host algebra: posteriors and free energies must agree to rounding (asserted at 1e-9; the contract is 1e-6 / 1e-8)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _spd(rng, d, s=1.0):
    a = rng.standard_normal((d, d))
    return s * (a @ a.T / d + 0.5 * np.eye(d))


@pytest.mark.parametrize("seed", range(24))
def test_chain_families_agree_with_the_executor(seed):
    from rxhip import graph
    from rxhip.tree import TreeEngine
    rng = np.random.default_rng(7000 + seed)
    d = int(rng.choice([1, 2, 3, 4, 4, 6]))
    dy = int(rng.integers(1, d + 1))
    T = int(rng.integers(4, 14))
    C = 3
    ptt = bool(rng.integers(0, 2))
    per_step = seed % 4 == 1
    nm = 3
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    As = [q @ np.diag(rng.uniform(0.4, 0.95, d)) @ q.T * rng.uniform(0.8, 1.0) for _ in range(nm)]
    Bs = [rng.standard_normal((dy, d)) for _ in range(nm)]
    Ps = [_spd(rng, d, 0.2) for _ in range(nm)]
    Qs = [_spd(rng, dy, 1.0) for _ in range(nm)]
    m0, V0 = rng.standard_normal(d), _spd(rng, d, 3.0)
    pick = rng.integers(0, nm, size=T + 1)
    kw = {}
    if per_step:
        kw.update(A_of_t=lambda t: As[pick[t]], P_of_t=lambda t: Ps[pick[t]], B_of_t=lambda t: Bs[pick[t]], Q_of_t=lambda t: Qs[pick[t]])
    if seed % 3 == 0:
        cx = rng.standard_normal((T + 1, d))
        kw.update(c_of_t=lambda t: cx[t], const_first=bool(seed % 2))
    if seed % 5 in (1, 2):
        cy = rng.standard_normal((T + 1, dy))
        kw.update(d_of_t=lambda t: cy[t])
    du = 0
    if seed % 6 == 2:
        du = int(rng.integers(1, 3))
        kw.update(Bu=rng.standard_normal((d, du)), du=du)
    elif seed % 6 == 5:
        du = d
        kw.update(du=d)
    out = graph.lgssm_graph(T, As[0], Bs[0], Ps[0], Qs[0], m0, V0, prior_through_transition=ptt, **kw)
    gb, xs, ys = out[0], out[1], out[2]
    us = out[3] if du else []
    y = rng.standard_normal((C, T, dy)) * 2.0
    u = rng.standard_normal((C, len(us), du)) if du else None
    # the specialised engine the pattern matcher picks
    eng = graph.create_engine_from_graph(gb.tables(n_replicas=C)[0])
    eng.set_data(y, layout="chain_time")
    if du:
        un = np.zeros((C, T, du))
        un[:, T - len(us):] = u           # (no transition into the first state when the prior sits on x[1]: that step has no input)
        eng.set_inputs(un, layout="chain_time")
    eng.run(1, True)
    mean, cov = eng.marginals(layout="chain_time")
    fe = eng.free_energy_per_chain()
    eng.close()
    # the executor on the same descriptor
    with TreeEngine(gb, n_replicas=C) as te:
        te.set_data(ys, y.reshape(C, T * dy))
        if du:
            te.set_data(us, u.reshape(C, len(us) * du))
        te.run(1, True)
        post = te.marginals(xs)
        tfe = te.free_energy_per_replica()
    tm = np.stack([post[v][0] for v in xs], axis=1)      # [C][T][d]
    tc = np.stack([post[v][1] for v in xs], axis=1)
    sd = np.sqrt(np.einsum("ctii->cti", cov))
    assert np.max(np.abs(tm - mean) / sd) < 1e-9
    assert np.max(np.abs(tc - cov) / (sd[..., :, None] * sd[..., None, :])) < 1e-9
    assert np.max(np.abs(tfe - fe) / np.abs(fe)) < 1e-10
