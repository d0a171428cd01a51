"""The node-array executor on its wavefront-per-item kernels (csrc/tree_tile_kernels.hpp: register tiles, dimensions 5 … 32; csrc/tree_wave_kernels.hpp:
LDS-staged, 33 … 64) through the C ABI
against oracle/tree_oracle.py and against the specialised engines — both schedules (a launch per level / a wavefront per replica walks the schedule),
dimensions 9 … 64, several replicas, VMP over precision variables, the single-rule entry point.

Tolerances as in tests/test_tree_engine_gpu.py: the contract is 1e-6 (posteriors) and 1e-8 (free energy); asserted at 1e-9 / 1e-10 here."""
import numpy as np
import pytest

import tree_graphs as tg
from test_tree_engine_gpu import _check, _run

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", [0, 2])
@pytest.mark.parametrize("builder,kw,R", [(tg.two_branch_chain, dict(T=5, d=12, dy1=12, dy2=7), 5), (tg.two_branch_chain, dict(T=4, d=16, dy1=9, dy2=16), 70),
                                          (tg.two_branch_chain, dict(T=3, d=33, dy1=20, dy2=33), 3), (tg.two_branch_chain, dict(T=2, d=64, dy1=64, dy2=30), 2),
                                          (tg.two_branch_chain, dict(T=4, d=9, dy1=9, dy2=9, precision_spelling=True), 4),
                                          (tg.chain_with_prediction, dict(T=6, H=2, d=10, dy=10), 3), (tg.star, dict(n_leaves=40, d=9), 3)])
def test_dimensions_above_8_against_the_oracle(builder, kw, R, mode, monkeypatch):
    gb, ys, _ = builder(**kw)
    eng, data = _run(gb, ys, R, mode=mode, monkeypatch=monkeypatch)
    assert eng.info["mode"] == mode and eng.info["dmax"] == kw["d"]
    ref = _check(gb, ys, eng, data, replicas=(0, R - 1), tol=1e-9, tol_fe=1e-10)
    assert eng.counters()["rule_calls"] == ref["counters"]["rule_calls"] * R
    eng.close()


def test_default_schedule_and_workgroup_mode_request(monkeypatch):
    """without the test hook: a launch per level until the replicas alone fill the device; the workgroup-resident schedule (mode 1) does not exist for
    these kernels and maps to the walk.  info.kernels: 1 = register tiles (a wavefront per item, 5 … 32), 2 = LDS-staged (33 … 64), 0 = a lane per item"""
    from rxhip.tree import TreeEngine
    gb, ys, _ = tg.two_branch_chain(T=2, d=10, dy1=10, dy2=4)
    monkeypatch.delenv("RXHIP_TREE_MODE", raising=False)
    monkeypatch.delenv("RXHIP_TREE_TILE", raising=False)
    with TreeEngine(gb, n_replicas=1) as eng:
        assert eng.info["mode"] == 0 and eng.info["kernels"] == 1
    with TreeEngine(gb, n_replicas=1024) as eng:
        assert eng.info["mode"] == 0
    with TreeEngine(gb, n_replicas=2048) as eng:
        assert eng.info["mode"] == 2          # a wavefront per replica walks the schedule once the replicas fill the device (d ≤ 16: from 2 048)
    gb32, _, _ = tg.two_branch_chain(T=2, d=20, dy1=20, dy2=4)
    with TreeEngine(gb32, n_replicas=512) as eng:
        assert eng.info["mode"] == 0 and eng.info["kernels"] == 1
    with TreeEngine(gb32, n_replicas=1024) as eng:
        assert eng.info["mode"] == 2          # (17 … 32: from 1 024)
    gb64, _, _ = tg.two_branch_chain(T=2, d=40, dy1=40, dy2=4)
    with TreeEngine(gb64, n_replicas=8) as eng:
        assert eng.info["mode"] == 0 and eng.info["kernels"] == 2
    with TreeEngine(gb64, n_replicas=256) as eng:
        assert eng.info["mode"] == 2          # (a workgroup per item: 256 of them are one per CU)
    monkeypatch.setenv("RXHIP_TREE_MODE", "1")
    with TreeEngine(gb, n_replicas=3) as eng:
        assert eng.info["mode"] == 2
    # dimensions 5 … 8: register tiles up to 1 024 replicas, a lane per item above
    monkeypatch.delenv("RXHIP_TREE_MODE", raising=False)
    gb6, _, _ = tg.two_branch_chain(T=2, d=6, dy1=6, dy2=4)
    with TreeEngine(gb6, n_replicas=1024) as eng:
        assert eng.info["kernels"] == 1 and eng.info["mode"] == 0 and eng.info["dmax"] == 8   # (dmax: the instance class — 1, 2, 4, 8 — up to 8)
    with TreeEngine(gb6, n_replicas=1025) as eng:
        assert eng.info["kernels"] == 0 and eng.info["dmax"] == 8
    gb4, _, _ = tg.two_branch_chain(T=2, d=4, dy1=4, dy2=4)
    with TreeEngine(gb4, n_replicas=1) as eng:
        assert eng.info["kernels"] == 0 and eng.info["mode"] == 3


@pytest.mark.parametrize("tile", [0, 1])
@pytest.mark.parametrize("mode", [0, 2])
@pytest.mark.parametrize("builder,kw,R", [(tg.two_branch_chain, dict(T=5, d=8, dy1=8, dy2=5), 5), (tg.two_branch_chain, dict(T=6, d=6, dy1=3, dy2=5), 70),
                                          (tg.two_branch_chain, dict(T=4, d=5, dy1=5, dy2=5, precision_spelling=True), 3), (tg.star, dict(n_leaves=40, d=7), 3)])
def test_dimensions_5_to_8_on_either_kernel_family(builder, kw, R, mode, tile, monkeypatch):
    """both kernel families on the same graphs against the oracle, whichever the batch would pick"""
    monkeypatch.setenv("RXHIP_TREE_TILE", str(tile))
    gb, ys, _ = builder(**kw)
    eng, data = _run(gb, ys, R, mode=mode, monkeypatch=monkeypatch)
    assert eng.info["mode"] == mode and eng.info["kernels"] == tile
    ref = _check(gb, ys, eng, data, replicas=(0, R - 1), tol=1e-9, tol_fe=1e-10)
    assert eng.counters()["rule_calls"] == ref["counters"]["rule_calls"] * R
    eng.close()


@pytest.mark.parametrize("d,dy,T,R,mode", [(12, 12, 20, 3, 0), (16, 8, 15, 70, 2), (24, 24, 6, 1, 2)])
def test_state_space_chain_equals_the_specialised_engine(d, dy, T, R, mode, monkeypatch):
    """the LGSSM chain at d > 8 through the generic path against LGSSMEngine (the MFMA schedule)"""
    import rxhip
    from rxhip import workloads
    from rxhip.graph import lgssm_graph
    from rxhip.tree import TreeEngine
    monkeypatch.setenv("RXHIP_TREE_MODE", str(mode))
    m = workloads.random_model(d, dy, seed=10 * d + dy)
    y = workloads.generate_batch(m, T, R, seed0=7)                     # [T][R][dy]
    gb, xs, ys = lgssm_graph(T, m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"])
    with TreeEngine(gb, n_replicas=R) as eng:
        eng.set_data(ys, np.ascontiguousarray(np.transpose(y, (1, 0, 2))).reshape(R, T * dy))
        eng.run(1, True)
        post = eng.marginals(xs)
        fe = eng.free_energy_per_replica()
    with rxhip.LGSSMEngine(m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"], T=T, n_chains=R) as ref:
        ref.set_data(y)
        ref.run(1, True)
        mean, cov = ref.marginals()
        rfe = ref.free_energy_per_chain()
    tm = np.stack([post[v][0] for v in xs])
    tc = np.stack([post[v][1] for v in xs])
    sd = np.sqrt(np.einsum("trii->tri", cov))
    assert np.max(np.abs(tm - mean) / sd) < 1e-9
    assert np.max(np.abs(tc - cov) / (sd[..., :, None] * sd[..., None, :])) < 1e-9
    assert np.max(np.abs(fe - rfe) / np.abs(rfe)) < 1e-10


@pytest.mark.parametrize("mode", [0, 2])
@pytest.mark.parametrize("kw", [dict(T=6, d=10, dy=9, also_obs_noise=True), dict(T=5, d=17, dy=17)])
def test_unknown_state_noise_precision_vmp(kw, mode, monkeypatch):
    import tree_oracle
    gb, ys, named = tg.chain_state_noise_precision(**kw)
    R, its = 3, 5
    eng, data = _run(gb, ys, R, iterations=its, mode=mode, monkeypatch=monkeypatch)
    _check(gb, ys, eng, data, iterations=its, replicas=(0, R - 1), prec_vars=named["W"])
    fe_it = eng.free_energy()
    tot = np.zeros(its)
    for r in range(R):
        tot += np.array(tree_oracle.infer(gb.to_dump(), tg.data_dict(gb, ys, data[r]), iterations=its)["fe"])
    assert np.allclose(fe_it, tot, rtol=1e-10)
    eng.close()


def test_rule_eval_above_8():
    from rxhip import _lib
    from rxhip.tree import rule_eval
    rng = np.random.default_rng(8)
    n = 19

    def spd(k):
        a = rng.standard_normal((n, k, k + 2))
        return a @ np.transpose(a, (0, 2, 1)) / k + 0.5 * np.eye(k)
    for d, dy in ((12, 9), (40, 33), (64, 20)):   # (dy <= d: the moment form of A V Aᵀ is asked back through a precision)
        m, V, S = rng.standard_normal((n, d)), spd(d), spd(d)[0]
        a, B = rule_eval(_lib.NODE_MVNORMAL_MEAN_COV, 0, S, (m, V))
        assert np.allclose(a, m, rtol=1e-13) and np.allclose(B, V + S, rtol=1e-13)
        L = np.linalg.inv(V)
        a, B = rule_eval(_lib.NODE_MVNORMAL_MEAN_COV, 1, S, (np.einsum("nij,nj->ni", L, m), L), in_form="wp", out_form="mv")
        assert np.allclose(a, m, rtol=1e-8, atol=1e-9) and np.allclose(B, V + S, rtol=1e-8, atol=1e-9)
        A = rng.standard_normal((dy, d))
        a, B = rule_eval(_lib.NODE_MULTIPLY, 0, A, (m, V))
        assert np.allclose(a, m @ A.T, rtol=1e-8, atol=1e-9) and np.allclose(B, A @ V @ A.T, rtol=1e-8, atol=1e-9)
        xi, Ly = rng.standard_normal((n, dy)), spd(dy)
        a, B = rule_eval(_lib.NODE_MULTIPLY, 2, A, (xi, Ly), in_form="wp", out_form="wp")
        assert np.allclose(a, xi @ A, rtol=1e-12, atol=1e-12) and np.allclose(B, A.T @ Ly @ A, rtol=1e-12, atol=1e-12)
        m2, V2 = rng.standard_normal((n, d)), spd(d)
        a, B = rule_eval(_lib.NODE_ADD, 1, None, (m, V), (m2, V2))
        assert np.allclose(a, m - m2) and np.allclose(B, V + V2)


def test_dimension_65_is_refused_with_a_reason():
    import rxhip
    from rxhip import _lib
    from rxhip.tree import TreeEngine
    gb, ys, _ = tg.two_branch_chain(T=2, d=65, dy1=3, dy2=3)
    with pytest.raises(rxhip.RxHipError) as ei:
        TreeEngine(gb, n_replicas=1)
    assert ei.value.status == _lib.ERR_UNSUPPORTED and "64" in str(ei.value)


@pytest.mark.parametrize("seed,dmax,mode", [(100, 33, 0), (101, 48, 2), (102, 64, 0), (103, 64, 2), (104, 33, 2), (105, 48, 0)])
def test_random_forests_at_large_dimensions(seed, dmax, mode, monkeypatch):
    """tests/tree_graphs.py::random_forest with dimensions up to 64 (odd seeds: shared precision variables, 2 VMP iterations) against the oracle"""
    prec = seed % 2 == 1
    its = 2 if prec else 1
    gb, ys, named = tg.random_forest(seed, n_steps=8, dmax=dmax, precision_vars=prec)
    R = 2
    eng, data = _run(gb, ys, R, iterations=its, mode=mode, monkeypatch=monkeypatch, seed=seed)
    assert eng.info["mode"] == mode
    _check(gb, ys, eng, data, iterations=its, replicas=(0, R - 1), prec_vars=named["W"], tol=1e-8, tol_fe=1e-9)
    eng.close()


@pytest.mark.parametrize("seed,dmax,mode", [(200 + i, dm, i % 2 * 2) for i, dm in enumerate((9, 12, 16, 16, 17, 20, 24, 24, 29, 32, 32, 32, 6, 7, 8, 5))])
def test_random_forests_on_the_register_tiles(seed, dmax, mode, monkeypatch):
    """random forests at the dimensions of the register-tile kernels (one tile: 9 … 16 and, forced, 5 … 8; four tiles: 17 … 32) — every construct of the family,
    odd seeds with shared precision variables and two VMP iterations — against the oracle, in both schedules"""
    prec = seed % 2 == 1
    its = 2 if prec else 1
    gb, ys, named = tg.random_forest(seed, n_steps=8, dmax=dmax, precision_vars=prec, dim_set=(1, 3, 5, 6, 7, 8, 9, 12, 15, 16, 17, 20, 24, 29, 31, 32))
    if dmax <= 8:
        monkeypatch.setenv("RXHIP_TREE_TILE", "1")
    R = 3
    eng, data = _run(gb, ys, R, iterations=its, mode=mode, monkeypatch=monkeypatch, seed=seed)
    assert eng.info["mode"] == mode and eng.info["kernels"] == (1 if eng.info["dmax"] > 4 else 0)   # (a forest's largest dimension is drawn: some stay at 4 and below)
    _check(gb, ys, eng, data, iterations=its, replicas=(0, R - 1), prec_vars=named["W"], tol=1e-8, tol_fe=1e-9)
    eng.close()


def test_a_batch_whose_message_array_passes_four_gigabytes():
    """64-bit addressing on the register tiles: 100 000 replicas × ≈ 90 KB of state each — the last replica's slots lie beyond 2³² bytes"""
    from rxhip.tree import TreeEngine
    gb, ys, _ = tg.two_branch_chain(T=8, d=12, dy1=12, dy2=7)
    R = 100000
    data = tg.random_data(gb, ys, R, 9)
    with TreeEngine(gb, n_replicas=R) as eng:
        assert eng.info["kernels"] == 1 and eng.info["doubles_per_replica"] * 8 * R > 2 ** 32
        eng.set_data(ys, data)
        eng.run(1, True)
        _check(gb, ys, eng, data, replicas=(0, R // 2, R - 1), tol=1e-9, tol_fe=1e-10)


@pytest.mark.parametrize("seed", range(48))
def test_rule_eval_random_calls(seed):
    """rxhip_rule_eval with random node types, interfaces, dimensions (1 … 64: register and LDS-staged kernels) and message forms on both sides,
    against the rules of SURVEY Appendix A.2 evaluated in numpy"""
    from rxhip import _lib
    from rxhip.tree import rule_eval
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.integers(1, 40))
    d = int(rng.choice([1, 2, 3, 4, 5, 8, 9, 16, 20, 33, 64]))

    def spd(k, s=1.0):
        a = rng.standard_normal((n, k, k + 2))
        return s * (a @ np.transpose(a, (0, 2, 1)) / (k + 2) + 0.4 * np.eye(k))

    def give(m, V, form):     # a Gaussian (m, V) as the arrays of the requested message form
        if form == "mv":
            return m, V
        L = np.linalg.inv(V)
        return np.einsum("nij,nj->ni", L, m), L

    def take(a, B, form):     # … and back to (m, V)
        if form == "mv":
            return a, B
        V = np.linalg.inv(B)
        return np.einsum("nij,nj->ni", V, a), V
    kind = seed % 4
    in_form, out_form = ("mv", "wp")[int(rng.integers(0, 2))], ("mv", "wp")[int(rng.integers(0, 2))]
    m, V = rng.standard_normal((n, d)), spd(d)
    if kind == 0:      # Gaussian node, covariance or precision parametrised, scalar spellings at d = 1
        S = spd(d)[0]
        prec = bool(rng.integers(0, 2))
        t = {(False, False): _lib.NODE_MVNORMAL_MEAN_COV, (True, False): _lib.NODE_MVNORMAL_MEAN_PRECISION,
             (False, True): _lib.NODE_NORMAL_MEAN_VARIANCE, (True, True): _lib.NODE_NORMAL_MEAN_PRECISION}[(prec, d == 1 and bool(rng.integers(0, 2)))]
        a, B = rule_eval(t, int(rng.integers(0, 2)), np.linalg.inv(S) if prec else S, give(m, V, in_form), in_form=in_form, out_form=out_form)
        rm, rV = m, V + S
    elif kind == 1:    # typeof(*)(:out): rows <= columns so that either form of the result exists
        r = int(rng.integers(1, d + 1))
        A = rng.standard_normal((r, d))
        a, B = rule_eval(_lib.NODE_MULTIPLY, 0, A, give(m, V, in_form), in_form=in_form, out_form=out_form)
        rm, rV = m @ A.T, A @ V @ A.T
    elif kind == 2:    # typeof(*)(:in): the message from `out` has the rows' dimension; the result is a precision of rank <= rows
        r = int(rng.integers(1, d + 1))
        A = rng.standard_normal((r, d))
        my, Vy = rng.standard_normal((n, r)), spd(r)
        out_form = "wp" if r < d else out_form
        a, B = rule_eval(_lib.NODE_MULTIPLY, 2, A, give(my, Vy, in_form), in_form=in_form, out_form=out_form)
        Ly = np.linalg.inv(Vy)
        xi, L = np.einsum("ij,nik,nk->nj", A, Ly, my), np.einsum("ij,nik,kl->njl", A, Ly, A)
        if out_form == "wp":
            scale = np.max(np.abs(L))
            assert np.allclose(a, xi, rtol=1e-8, atol=1e-9 * scale) and np.allclose(B, L, rtol=1e-8, atol=1e-9 * scale)
            return
        rV = np.linalg.inv(L)
        rm = np.einsum("nij,nj->ni", rV, xi)
    else:              # typeof(+): (:out), (:in1), (:in2)
        iface = int(rng.integers(0, 3))
        m2, V2 = rng.standard_normal((n, d)), spd(d)
        a, B = rule_eval(_lib.NODE_ADD, iface, None, give(m, V, in_form), give(m2, V2, in_form), in_form=in_form, out_form=out_form)
        rm, rV = (m + m2, V + V2) if iface == 0 else (m - m2, V + V2)
    gm, gV = take(a, B, out_form)
    sd = np.sqrt(np.einsum("nii->ni", rV))
    assert np.max(np.abs(gm - rm) / sd) < 1e-7 and np.max(np.abs(gV - rV) / (sd[:, :, None] * sd[:, None, :])) < 1e-7
