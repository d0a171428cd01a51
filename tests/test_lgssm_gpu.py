"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs.  Tolerances are BASELINE.json's: posterior means / covariances 1e-6 relative, free energy
1e-8 relative.  Shaped after test/models/statespace/mlgssm_test.jl."""
import numpy as np
import pytest

import rxhip
import rxoracle
from rxhip import workloads

pytestmark = pytest.mark.gpu

RTOL_POST = 1e-6
RTOL_FE = 1e-8


def rel(a, b):
    """relative error PER LEADING INDEX (time step): max |Δ| over the trailing axes on the scale of that step's reference (floored
    at 1e-3 of the global scale, so that a mean crossing zero does not divide by nothing) — element-wise in time, not a norm
    over the whole array (a norm-wise 1e-6 is not what "1e-6 relative" says)"""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if b.ndim < 2:
        return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
    ax = tuple(range(1, b.ndim))
    scale = np.maximum(np.max(np.abs(b), axis=ax), 1e-3 * np.max(np.abs(b)))
    return float(np.max(np.max(np.abs(a - b), axis=ax) / scale))


def run_engine(mdl, y, **kw):
    T, C = y.shape[0], y.shape[1]
    with rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=C, **kw) as eng:
        eng.set_data(y)
        eng.run(iterations=1, free_energy=True)
        mean, cov = eng.marginals()
        return mean, cov, eng.free_energy_per_chain(), eng.free_energy()[0], eng.counters(), eng.schedule()


def oracle_batch(mdl, y, ptt=False):
    """Reference-schedule oracle; for dy < d (where the reference schedule itself fails on a singular
    precision, see tests/test_oracle.py) the textbook Kalman/RTS oracle."""
    d, dy = mdl["A"].shape[0], mdl["B"].shape[0]
    if dy >= d:
        return rxoracle.lgssm_bp_batch(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], y,
                                       prior_through_transition=ptt)
    T, C = y.shape[:2]
    om, oc, ofe = np.empty((T, C, d)), np.empty((T, C, d, d)), np.empty(C)
    for c in range(C):
        om[:, c], oc[:, c], ofe[c] = rxoracle.lgssm_kalman_rts(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"],
                                                              mdl["V0"], y[:, c], prior_through_transition=ptt)
    cnt = rxoracle.Counters(C * (6 * T + 1 if ptt else 6 * T - 3), C * (4 * T - 2 if ptt else max(1, 4 * T - 4)), C * T)
    return om, oc, ofe, cnt


def check_against_oracle(mdl, y, **kw):
    mean, cov, fe, fetot, cnt, sched = run_engine(mdl, y, **kw)
    om, oc, ofe, ocnt = oracle_batch(mdl, y, kw.get("prior_through_transition", False))
    assert rel(mean, om) < RTOL_POST, rel(mean, om)
    assert rel(cov, oc) < RTOL_POST, rel(cov, oc)
    assert np.max(np.abs(fe - ofe) / np.abs(ofe)) < RTOL_FE
    assert abs(fetot - ofe.sum()) < RTOL_FE * abs(ofe.sum())
    assert cnt["rule_calls"] == ocnt.rule_calls
    assert cnt["products"] == ocnt.products
    return mean, cov, fe, sched


def test_c1_single_chain_t1000():
    """BASELINE config 1: d=4, T=1000, one chain."""
    mdl = workloads.c1_model()
    y = workloads.generate_batch(mdl, 1000, 1)
    mean, cov, fe, sched = check_against_oracle(mdl, y)
    assert sched["segments"] > 1  # the parallel-in-time schedule is exercised
    assert np.all(np.linalg.eigvalsh(cov[:, 0]) > 0)  # mlgssm_test.jl:126


@pytest.mark.parametrize("C,T,segments", [(70, 257, 0), (64, 100, 7), (1, 50, 49), (3, 2, 0), (5, 1, 0), (130, 33, 1),
                                          (2, 1000, 3)])
def test_ragged_shapes_and_segmentations(C, T, segments):
    """chains not a multiple of the wave size, T=1/T=2, one segment, one step per segment, a short last segment."""
    mdl = workloads.c1_model()
    y = workloads.generate_batch(mdl, T, C, seed0=7)
    check_against_oracle(mdl, y, segments=segments)


@pytest.mark.parametrize("d,dy", [(1, 1), (2, 1), (2, 2), (3, 3), (4, 2), (4, 4)])
def test_dimensions_dense_models(d, dy):
    mdl = workloads.random_model(d, dy, seed=100 + 10 * d + dy)
    y = workloads.generate_batch(mdl, 300, 9, seed0=3)
    check_against_oracle(mdl, y, segments=5)


@pytest.mark.parametrize("d,dy,T,C,segments", [(16, 16, 90, 3, 4), (32, 32, 70, 2, 0), (48, 48, 40, 1, 3), (64, 64, 61, 2, 5),
                                                 (32, 8, 50, 2, 3), (64, 64, 1, 1, 0), (16, 16, 2, 2, 0)])
def test_dense_state_dimensions(d, dy, T, C, segments):
    """d = 16·NT path (BASELINE config 3 is d = dy = 64): one workgroup per (chain, segment), MFMA f64
    contractions, Gauss–Jordan SPD inverses."""
    mdl = workloads.random_model(d, dy, seed=7 + d + dy, stable=0.9)
    y = workloads.generate_batch(mdl, T, C, seed0=11)
    check_against_oracle(mdl, y, segments=segments)


def test_c3_model_short():
    """BASELINE config 3's model (dense A, dense full-rank B, d = 64) on a short chain."""
    mdl = workloads.c3_model()
    y = workloads.generate_batch(mdl, 200, 1, seed0=6400)
    mean, cov, fe, sched = check_against_oracle(mdl, y)
    assert sched["segments"] > 1


def test_c3_full_size():
    """BASELINE config 3 at its full size (d = dy = 64, T = 10⁴, one chain): 250 segments, two-level boundary scan,
    aggregation as a matrix product — against the oracle's reference schedule over the whole chain (≈30 s of CPU)."""
    mdl = workloads.c3_model()
    y = workloads.generate_batch(mdl, 10000, 1, seed0=6400)   # BASELINE config 3 as bench.py draws it
    mean, cov, fe, sched = check_against_oracle(mdl, y)
    assert sched["segments"] >= 200


def test_prior_through_transition_variant():
    """test/models/statespace/mlgssm_test.jl:9-17 spelling: x0 ~ prior; x[1] ~ MvNormal(A*x0, .)"""
    mdl = workloads.random_model(2, 2, seed=5)
    y = workloads.generate_batch(mdl, 200, 4)
    check_against_oracle(mdl, y, prior_through_transition=True)


def test_per_chain_models():
    """chains with different constants in one batch (non-uniform constant path)."""
    mdls = [workloads.random_model(4, 4, seed=s) for s in (1, 2, 3)]
    C, T = 11, 120
    cm = np.arange(C) % 3
    y = np.empty((T, C, 4))
    for c in range(C):
        y[:, c] = workloads.generate_chain(mdls[cm[c]], T, 50 + c)[1]
    stack = lambda k: np.stack([m[k] for m in mdls])
    with rxhip.LGSSMEngine(stack("A"), stack("B"), stack("P"), stack("Q"), stack("m0"), stack("V0"), T=T, n_chains=C,
                           chain_model=cm, segments=4) as eng:
        eng.set_data(y)
        eng.run(1, True)
        mean, cov = eng.marginals()
        fe = eng.free_energy_per_chain()
    for c in range(C):
        m = mdls[cm[c]]
        om, oc, ofe, _ = rxoracle.lgssm_bp(m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"], y[:, c])
        assert rel(mean[:, c], om) < RTOL_POST and rel(cov[:, c], oc) < RTOL_POST
        assert abs(fe[c] - ofe) < RTOL_FE * abs(ofe)


def test_known_answers_on_device():
    """RNG-free known answers of test/models/models_tests.jl:255,286,308,335 through the HIP path."""
    I = np.eye(1)
    for mu, fe_ref, mean_ref in [(3.0, 3.51551, 1.5), (2.0, 2.26551, 1.0)]:
        mdl = dict(A=I, B=I, P=I, Q=I, m0=np.array([mu]), V0=I)
        mean, cov, fe, *_ = run_engine(mdl, np.zeros((1, 1, 1)))
        assert abs(fe[0] - fe_ref) < 1e-5 and abs(mean[0, 0, 0] - mean_ref) < 1e-12 and abs(cov[0, 0, 0, 0] - 0.5) < 1e-12


def test_layouts_and_infer_mirror():
    """`infer(model = ..., data = (y = ...,), free_energy = true)` mirror, chain-major host layout."""
    mdl = workloads.c1_model()
    y = workloads.generate_batch(mdl, 200, 6)  # [T][C][dy]
    spec = rxhip.linear_gaussian_ssm(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"])
    res = rxhip.infer(model=spec, data={"y": np.transpose(y, (1, 0, 2))}, free_energy=True, iterations=2)
    om, oc, ofe, _ = rxoracle.lgssm_bp_batch(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], y)
    assert res.error is None
    assert rel(res.posteriors["x"].mean, np.transpose(om, (1, 0, 2))) < RTOL_POST
    assert rel(res.posteriors["x"].cov, np.transpose(oc, (1, 0, 2, 3))) < RTOL_POST
    assert res.free_energy.shape == (6, 2)
    assert np.max(np.abs(res.free_energy[:, -1] - ofe) / np.abs(ofe)) < RTOL_FE
    # single chain: data [T][dy]
    r1 = rxhip.infer(model=spec, data={"y": y[:, 0]}, free_energy=True)
    assert rel(r1.posteriors["x"].mean, om[:, 0]) < RTOL_POST and r1.free_energy.shape == (1,)


def test_iterations_repush_data():
    """every iteration re-pushes the data and recomputes (src/inference/batch.jl:391-430): FE per
    iteration identical, counters scale with the iteration count."""
    mdl = workloads.c1_model()
    y = workloads.generate_batch(mdl, 64, 3)
    with rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=64, n_chains=3) as eng:
        eng.set_data(y)
        eng.run(3, True)
        fe = eng.free_energy()
        assert fe.shape == (3,) and fe[0] == fe[1] == fe[2]
        assert eng.counters()["rule_calls"] == 3 * 3 * (6 * 64 - 3)


def test_error_paths():
    mdl = workloads.c1_model()
    # non-SPD constant -> RXHIP_ERR_NOT_POSDEF at create (mirrors FastCholesky PosDefException)
    with pytest.raises(rxhip.RxHipError) as ei:
        rxhip.LGSSMEngine(mdl["A"], mdl["B"], -mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=10)
    assert ei.value.status == 3
    with rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=10) as eng:
        with pytest.raises(rxhip.RxHipError) as ei:  # run before data
            eng.run()
        assert ei.value.status == 7
        with pytest.raises(rxhip.RxHipError):  # wrong size
            eng.set_data(np.zeros((9, 1, 4)))
        # NaN observation -> non-finite free energy is reported (src/score/diagnostics.jl:19-51)
        y = np.zeros((10, 1, 4))
        y[3, 0, 1] = np.nan
        eng.set_data(y)
        with pytest.raises(rxhip.RxHipError) as ei:
            eng.run(1, True)
        assert ei.value.status in (3, 4)
    res = rxhip.infer(model=rxhip.linear_gaussian_ssm(mdl["A"], mdl["B"], -mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"]),
                      data={"y": np.zeros((5, 4))}, catch_exception=True)
    assert res.error is not None  # catch_exception semantics, src/inference/batch.jl:440-446


def test_properties_at_scale():
    """Size-independent properties on a larger batch (the full C2 size runs in bench.py):
    (i) posterior means are affine in y: m(y1 + y2) = m(y1) + m(y2) − m(0);
    (ii) covariances do not depend on the data and are identical across chains of one model;
    (iii) a sample of chains matches the oracle at full length."""
    mdl = workloads.c1_model()
    T, C = 20000, 256
    rng = np.random.default_rng(0)
    y1 = rng.standard_normal((T, C, 4)) * 3
    y2 = rng.standard_normal((T, C, 4)) * 3
    with rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=C) as eng:
        out = []
        for yy in (y1, y2, y1 + y2, np.zeros_like(y1)):
            eng.set_data(yy)
            eng.run(1, True)
            m, V = eng.marginals()
            out.append((m, V, eng.free_energy_per_chain()))
    m1, m2, m12, m0 = (o[0] for o in out)
    assert np.max(np.abs(m12 - (m1 + m2 - m0))) < 1e-9 * np.max(np.abs(m12))
    V = out[0][1]
    assert np.max(np.abs(V - V[:, :1])) < 1e-12 and np.max(np.abs(out[1][1] - V)) < 1e-12
    for c in (0, 77, 255):
        om, oc, ofe, _ = rxoracle.lgssm_bp(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], y1[:, c])
        assert rel(m1[:, c], om) < RTOL_POST and rel(V[:, c], oc) < RTOL_POST
        assert abs(out[0][2][c] - ofe) < RTOL_FE * abs(ofe)


def test_independent_handles_from_several_host_threads():
    """One host thread per handle, handles independent (include/rxhip.h): four threads create / run / destroy engines
    concurrently (the stream and block pools are the only shared state) and every result matches the oracle."""
    import threading

    mdl = workloads.notebook_model()
    errs, lock = [], threading.Lock()

    def worker(seed):
        try:
            rng = np.random.default_rng(seed)
            for _ in range(25):
                T, C = int(rng.choice([3, 40, 300])), int(rng.choice([1, 2, 64]))
                y = workloads.generate_batch(mdl, T, C, seed0=int(rng.integers(1000)))
                with rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=C) as eng:
                    eng.set_data(y)
                    eng.run(1, True)
                    mean, _ = eng.marginals()
                    fe = eng.free_energy_per_chain()
                om, _, ofe, _ = rxoracle.lgssm_bp(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], y[:, 0])
                assert rel(mean[:, 0], om) < RTOL_POST and abs(fe[0] - ofe) < RTOL_FE * abs(ofe)
        except Exception as e:  # noqa: BLE001
            with lock:
                errs.append(repr(e))

    ts = [threading.Thread(target=worker, args=(s,)) for s in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


def test_one_round_trip_inference_equals_the_plain_sequence():
    """rxhip_lgssm_infer (set_data + run + marginals + free energy, one synchronisation) against the four separate calls, incl. a
    problem large enough to take the plain sequence internally and an engine with a forecast horizon."""
    for T, C, H in ((50, 1, 0), (300, 7, 3), (40000, 16, 0)):
        mdl = workloads.c1_model()
        y = workloads.generate_batch(mdl, T, C, seed0=T)
        with rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=C, horizon=H) as eng:
            m1, c1, f1 = eng.infer(y, iterations=2, free_energy=True)
            eng.set_data(y)
            eng.run(2, True)
            m2, c2 = eng.marginals()
            f2 = eng.free_energy_per_chain()
            m3, _, _ = eng.infer(y, free_energy=False, want_cov=False)
            eng.run_filter(True)
            fm, fc = eng.marginals()
            ff = eng.free_energy_per_chain()
            m4, c4, f4 = eng.infer(y, free_energy=True, filtering=True)
        assert np.array_equal(m4, fm) and np.array_equal(c4, fc) and np.array_equal(f4, ff)
        assert np.array_equal(m1, m2) and np.array_equal(c1, c2) and np.array_equal(f1, f2) and np.array_equal(m3, m2)


def test_one_round_trip_inference_on_engines_whose_results_are_not_one_span():
    """rxhip_lgssm_infer reads status | mean | cov | free energy back with ONE copy where the arena holds them next to each other (d ≤ 4:
    the test above); the MFMA path's engines keep them apart and take the four copies — same numbers as the plain sequence there too."""
    for d, dy, T, C in ((8, 3, 60, 2), (20, 5, 40, 1)):
        mdl = workloads.random_model(d, dy, seed=3 * d)
        y = workloads.generate_batch(mdl, T, C, seed0=d)
        with rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=C) as eng:
            m1, c1, f1 = eng.infer(y, free_energy=True)
            eng.set_data(y)
            eng.run(1, True)
            m2, c2 = eng.marginals()
            f2 = eng.free_energy_per_chain()
            m3, c3, _ = eng.infer(y, free_energy=False)
        assert np.array_equal(m1, m2) and np.array_equal(c1, c2) and np.array_equal(f1, f2)
        assert np.array_equal(m3, m2) and np.array_equal(c3, c2)


def test_model_tables_timing_entry_point(monkeypatch):
    """rxhip_get_model_tables_ms: device time of what an engine computes once because it depends on the model only — positive for
    a shared-model batch on the one-pass schedule, zero for an engine without such tables; the sweep results do not depend on
    when it is asked."""
    mdl = workloads.random_model(3, 2, seed=8)
    args = (mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"])
    y = workloads.generate_batch(mdl, 300, 64, seed0=2)
    monkeypatch.setenv("RXHIP_ONE_PASS", "1")
    with rxhip.LGSSMEngine(*args, T=300, n_chains=64) as eng:
        ms = eng.model_tables_ms()
        assert 0.0 < ms < 1e3
        eng.set_data(y)
        eng.run(1, True)
        m1, fe1 = eng.marginals()[0].copy(), eng.free_energy()[-1]
        assert eng.model_tables_ms() == ms      # the events of creation, not of the sweep
    monkeypatch.setenv("RXHIP_ONE_PASS", "0")
    with rxhip.LGSSMEngine(*args, T=300, n_chains=64) as eng:
        assert eng.model_tables_ms() == 0.0
        eng.set_data(y)
        eng.run(1, True)
        assert rel(eng.marginals()[0], m1) < 1e-9 and abs(eng.free_energy()[-1] - fe1) < 1e-9 * abs(fe1)
