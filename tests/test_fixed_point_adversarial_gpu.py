"""The fixed-point exits of the time-invariant sweeps (d ≤ 4: k_seg_elements / k_boundary_scan / k_forward_tinv and the table builder;
d ≥ 48: FROZEN / BFROZEN of kd_forward_info / kd_backward_info) stop recomputing a covariance recursion once it repeats.  The inputs
below are the ones such a test has the hardest time with (include/rxhip.h "Fixed-point exits" states the bound they are held to):

  * block-diagonal models whose blocks live six decades apart, the SMALL block mixing slowly (spectral radius 0.9999) — whatever
    summarises the matrix by a few numbers is dominated by the large, quickly converged block;
  * a near-unit-root state with a process noise eight decades below the observation noise (local level): the recursion approaches its
    fixed point like 1/t first and geometrically with a rate next to one afterwards;

each against the CPU oracle's smoother per time step (1e-6 on the scale of every component's own posterior standard deviation, the
contract of include/rxhip.h) and against the same engine with the exits switched off (RXHIP_ELEM_FULL / RXHIP_NO_FROZEN)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(m, y, monkeypatch, full, per_chain=False, segments=0):
    import rxhip
    for k in ("RXHIP_ELEM_FULL", "RXHIP_NO_FROZEN"):
        (monkeypatch.setenv(k, "1") if full else monkeypatch.delenv(k, raising=False))
    T, C = y.shape[0], y.shape[1]
    if per_chain:   # one model per chain (identical values): the per-chain kernels with their own exits
        mm = {k: np.repeat(np.asarray(v)[None], C, 0) for k, v in m.items()}
        kw = dict(chain_model=np.arange(C, dtype=np.int32))
    else:
        mm, kw = m, {}
    with rxhip.LGSSMEngine(mm["A"], mm["B"], mm["P"], mm["Q"], mm["m0"], mm["V0"], T=T, n_chains=C, segments=segments, **kw) as eng:
        eng.set_data(y)
        eng.run(1, True)
        mean, cov = eng.marginals()
        return mean, cov, eng.free_energy_per_chain(), eng.schedule()


def _against_oracle(m, y, mean, cov, fe, tol=1e-6, tol_fe=1e-8, chains=None):
    import rxoracle as rxo
    for c in (range(y.shape[1]) if chains is None else chains):
        om, oc, nll = rxo.lgssm_kalman_rts(m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"], np.ascontiguousarray(y[:, c]))
        sd = np.sqrt(np.einsum("tii->ti", oc))
        em = np.max(np.abs(mean[:, c] - om) / sd)
        ec = np.max(np.abs(cov[:, c] - oc) / (sd[:, :, None] * sd[:, None, :]))
        assert em < tol and ec < tol, (c, em, ec)
        assert abs(fe[c] - nll) <= tol_fe * abs(nll), (c, fe[c], nll)


def _two_scale_model(d, rho_slow, decades, seed):
    """Two decoupled halves: a well-mixed one at unit scale and a slowly mixing one `decades` below it.  Both are observed."""
    rng = np.random.default_rng(seed)
    h = d // 2
    q1, _ = np.linalg.qr(rng.standard_normal((h, h)))
    q2, _ = np.linalg.qr(rng.standard_normal((d - h, d - h)))
    A = np.zeros((d, d))
    A[:h, :h] = q1 @ np.diag(np.linspace(0.3, 0.8, h)) @ q1.T
    A[h:, h:] = q2 @ np.diag(np.linspace(0.9, rho_slow, d - h)) @ q2.T
    s = np.concatenate([np.ones(h), np.full(d - h, 10.0 ** -decades)])
    P = np.diag(np.concatenate([np.full(h, 0.05), np.full(d - h, 1e-4)]) * s * s)
    V0 = np.diag(25.0 * s * s)
    B = np.zeros((d, d))
    B[:h, :h] = np.eye(h) + 0.05 * rng.standard_normal((h, h))
    B[h:, h:] = (np.eye(d - h) + 0.05 * rng.standard_normal((d - h, d - h))) * 10.0 ** decades   # the small half is observed at unit scale
    Q = np.diag(np.concatenate([np.full(h, 10.0), np.full(d - h, 1.0)]))
    return dict(A=A, B=B, P=P, Q=Q, m0=np.zeros(d), V0=V0)


def _generate(m, T, C, seed):
    rng = np.random.default_rng(seed)
    d, dy = m["A"].shape[0], m["B"].shape[0]
    Lp, Lq, L0 = (np.linalg.cholesky(m[k]) for k in ("P", "Q", "V0"))
    y = np.empty((T, C, dy))
    for c in range(C):
        x = m["m0"] + L0 @ rng.standard_normal(d)
        for t in range(T):
            if t:
                x = m["A"] @ x + Lp @ rng.standard_normal(d)
            y[t, c] = m["B"] @ x + Lq @ rng.standard_normal(dy)
    return y


@pytest.mark.parametrize("per_chain,C", [(False, 64), (True, 64), (False, 3), (True, 5)])
def test_two_scales_slow_small_block_d4(per_chain, C, monkeypatch):
    m = _two_scale_model(4, 0.9999, 6.0, seed=3)
    y = _generate(m, 6000, min(C, 4), seed=4)
    y = np.ascontiguousarray(np.tile(y, (1, (C + 3) // 4, 1))[:, :C])
    mean, cov, fe, sched = _run(m, y, monkeypatch, full=False, per_chain=per_chain)
    _against_oracle(m, y, mean, cov, fe, chains=(0, C - 1))
    mean_f, cov_f, fe_f, _ = _run(m, y, monkeypatch, full=True, per_chain=per_chain)
    sd = np.sqrt(np.einsum("tcii->tci", cov_f))
    assert np.max(np.abs(mean - mean_f) / sd) < 1e-7, sched
    assert np.max(np.abs(cov - cov_f) / (sd[..., :, None] * sd[..., None, :])) < 1e-7, sched
    assert np.max(np.abs(fe - fe_f) / np.abs(fe_f)) < 1e-9, sched


@pytest.mark.parametrize("per_chain", [False, True])
def test_near_unit_root_tiny_process_noise_d4(per_chain, monkeypatch):
    """local level in every component: A = (1 − 1e-6) I (rotated), P = 1e-8 Q — the filter covariance falls like 1/t for 10⁴ steps, then
    settles at a rate of 1 − 2e-4 per step.  (At P = 1e-12 Q the smoother's own V_s(t+1) − V_p(t+1) cancels eight digits in every implementation
    that forms it, the CPU oracle included: such a model has no fp64 reference to be held to.)"""
    d = 4
    rng = np.random.default_rng(12)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    m = dict(A=(1.0 - 1e-6) * (q @ np.diag([1.0, 0.999, 0.99, 0.9]) @ q.T), B=np.eye(d), P=1e-8 * np.eye(d), Q=np.eye(d), m0=np.zeros(d),
             V0=4.0 * np.eye(d))
    C = 64
    y = _generate(m, 20000, 2, seed=5)
    y = np.ascontiguousarray(np.tile(y, (1, C // 2, 1)))
    mean, cov, fe, sched = _run(m, y, monkeypatch, full=False, per_chain=per_chain)
    _against_oracle(m, y, mean, cov, fe, chains=(0, 1))
    mean_f, cov_f, fe_f, _ = _run(m, y, monkeypatch, full=True, per_chain=per_chain)
    sd = np.sqrt(np.einsum("tcii->tci", cov_f))
    assert np.max(np.abs(mean - mean_f) / sd) < 1e-7, sched
    assert np.max(np.abs(cov - cov_f) / (sd[..., :, None] * sd[..., None, :])) < 1e-7, sched


@pytest.mark.parametrize("d,T,segments", [(64, 4000, 0), (64, 4000, 16), (48, 3000, 0)])
def test_two_scales_slow_small_block_mfma(d, T, segments, monkeypatch):
    m = _two_scale_model(d, 0.9999, 6.0, seed=d)
    y = _generate(m, T, 1, seed=d + 1)
    mean, cov, fe, sched = _run(m, y, monkeypatch, full=False, segments=segments)
    _against_oracle(m, y, mean, cov, fe)
    mean_f, cov_f, fe_f, _ = _run(m, y, monkeypatch, full=True, segments=segments)
    sd = np.sqrt(np.einsum("tcii->tci", cov_f))
    assert np.max(np.abs(mean - mean_f) / sd) < 1e-7, sched
    assert np.max(np.abs(cov - cov_f) / (sd[..., :, None] * sd[..., None, :])) < 1e-7, sched
    assert np.max(np.abs(fe - fe_f) / np.abs(fe_f)) < 1e-9, sched


@pytest.mark.parametrize("d", [48, 64])
def test_identity_like_transition_small_noise_mfma(d, monkeypatch):
    """the advisor's case: A close to the identity and a small P at d = 48 / 64"""
    rng = np.random.default_rng(d)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    A = q @ np.diag(np.linspace(0.999, 0.99999, d)) @ q.T
    m = dict(A=A, B=np.eye(d) + 0.02 * rng.standard_normal((d, d)), P=1e-6 * np.eye(d), Q=np.eye(d), m0=np.zeros(d), V0=10.0 * np.eye(d))
    y = _generate(m, 5000, 1, seed=2 * d)
    mean, cov, fe, sched = _run(m, y, monkeypatch, full=False)
    _against_oracle(m, y, mean, cov, fe)
    mean_f, cov_f, fe_f, _ = _run(m, y, monkeypatch, full=True)
    sd = np.sqrt(np.einsum("tcii->tci", cov_f))
    assert np.max(np.abs(mean - mean_f) / sd) < 1e-7, sched
    assert np.max(np.abs(cov - cov_f) / (sd[..., :, None] * sd[..., None, :])) < 1e-7, sched


@pytest.mark.parametrize("d,per_chain", [(64, False), (4, True)])
def test_the_exits_can_be_switched_off_per_engine(d, per_chain, monkeypatch):
    """rxhip_set_fixed_point_exits(engine, 0) is the test hooks' RXHIP_NO_FROZEN / RXHIP_ELEM_FULL as an API: bit-identical results, and back on again"""
    import rxhip
    from rxhip import workloads
    m = workloads.random_model(d, d, seed=5 + d)
    T, C = (600, 1) if d == 64 else (4000, 64)
    y = workloads.generate_batch(m, T, C, seed0=3)
    hook = _run(m, y, monkeypatch, full=True, per_chain=per_chain)
    for k in ("RXHIP_ELEM_FULL", "RXHIP_NO_FROZEN"):
        monkeypatch.delenv(k, raising=False)
    if per_chain:
        mm = {k: np.repeat(np.asarray(v)[None], C, 0) for k, v in m.items()}
        kw = dict(chain_model=np.arange(C, dtype=np.int32))
    else:
        mm, kw = m, {}
    with rxhip.LGSSMEngine(mm["A"], mm["B"], mm["P"], mm["Q"], mm["m0"], mm["V0"], T=T, n_chains=C, **kw) as eng:
        eng.set_data(y)
        eng.run(1, True)
        on = (eng.marginals(), eng.free_energy_per_chain())
        eng.set_fixed_point_exits(False)
        eng.run(1, True)
        off = (eng.marginals(), eng.free_energy_per_chain())
        eng.set_fixed_point_exits(True)
        eng.run(1, True)
        again = (eng.marginals(), eng.free_energy_per_chain())
    assert np.array_equal(off[0][0], hook[0]) and np.array_equal(off[0][1], hook[1]) and np.array_equal(off[1], hook[2])
    assert np.array_equal(again[0][0], on[0][0]) and np.array_equal(again[0][1], on[0][1]) and np.array_equal(again[1], on[1])
    sd = np.sqrt(np.einsum("tcii->tci", off[0][1]))
    assert np.max(np.abs(on[0][0] - off[0][0]) / sd) < 1e-7
    with pytest.raises(rxhip.RxHipError):
        eng2 = rxhip.LGSSMEngine(m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"], T=8, n_chains=1)
        try:
            eng2._chk(rxhip._lib.lib().rxhip_set_fixed_point_exits(eng2._h, 7))
        finally:
            eng2.close()
