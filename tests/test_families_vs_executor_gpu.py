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


def test_iid_gaussian_with_unknown_mean_and_precision_agrees_at_the_fixed_point():
    """`iid_gaussians_params` (m ~ Normal, p ~ Gamma, y[i] ~ Normal(m, p⁻¹), q(m)q(p)) and `mv_iid_wishart`: the mixture engine at K = 1 against the
    executor (a hub of N leaves, one precision variable shared by N nodes).  The two schedules order the updates of an iteration differently, so what must
    agree is the fixed point: q(m), E[p] and the free energy after 80 iterations."""
    from rxhip import _lib, graph
    from rxhip.tree import TreeEngine
    rng = np.random.default_rng(31)
    # scalar
    N = 120
    y = 0.75 + 3.0 * rng.standard_normal(N)
    gb, ys = graph.iid_normal_graph(N, 4.0, 8.0, 4.0, 0.125, init=dict(m=(0.0, 1.0), p=(1.0, 1.0)))
    with graph.create_vmp_engine_from_graph(gb.tables()[0]) as eng:
        eng.set_data(y)
        eng.run(80, True)
        h, fe = eng.history()[-1], eng.free_energy()[-1]
    with TreeEngine(gb, n_replicas=1) as te:
        te.set_data(ys, y.reshape(1, N))
        te.run(80, True)
        m, p = [v for v in range(len(gb.kind)) if gb.kind[v] == _lib.VARKIND_RANDOM][:2]   # the two random variables, in creation order: m, p
        post = te.marginals([m])
        nu, V = te.precision(p)
        tfe = te.free_energy()[-1]
    assert post[m][0][0, 0] == pytest.approx(h[0, 0], rel=1e-9) and post[m][1][0, 0, 0] == pytest.approx(h[1, 0], rel=1e-9)
    assert nu[0] * V[0, 0, 0] == pytest.approx(h[2, 0] / h[3, 0], rel=1e-9)      # E[p] = shape / rate
    assert tfe == pytest.approx(fe, rel=1e-9)
    # multivariate
    d, N = 3, 200
    Lm = rng.standard_normal((d, d))
    yv = rng.multivariate_normal(rng.random(d), Lm @ Lm.T + 0.3 * np.eye(d), size=N)
    gb, ys = graph.mv_iid_graph(N, np.zeros(d), 100.0 * np.eye(d), d + 1.0, np.eye(d), init=dict(m=(np.zeros(d), np.eye(d)), w=(float(d + 1), np.eye(d))))
    with graph.create_vmp_engine_from_graph(gb.tables()[0]) as eng:
        eng.set_data(yv)
        eng.run(80, True)
        h, fe = eng.history(), eng.free_energy()[-1]
    with TreeEngine(gb, n_replicas=1) as te:
        te.set_data(ys, yv.reshape(1, N * d))
        te.run(80, True)
        m, p = [v for v in range(len(gb.kind)) if gb.kind[v] == _lib.VARKIND_RANDOM][:2]
        post = te.marginals([m])
        nu, V = te.precision(p)
        tfe = te.free_energy()[-1]
    assert np.allclose(post[m][0][0], h["mean"][-1, 0], rtol=1e-8, atol=1e-10)
    assert np.allclose(nu[0] * V[0], h["nu"][-1, 0] * h["V"][-1, 0], rtol=1e-8)
    assert tfe == pytest.approx(fe, rel=1e-9)
