"""The engine pool (csrc/rxhip.hip): rxhip_destroy parks a small engine, rxhip_lgssm_create hands it out again for a byte-identical descriptor.
A revived engine must be indistinguishable from a new one — whatever its previous owner did with it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(eng, y, filtering=False):
    eng.set_data(y)
    eng.run_filter(True) if filtering else eng.run(1, True)
    m, c = eng.marginals()
    return m, c, eng.free_energy_per_chain()


@pytest.mark.parametrize("d,dy,T,C", [(4, 4, 1000, 1), (2, 2, 50, 1), (3, 1, 700, 8), (1, 1, 4000, 16)])
def test_a_revived_engine_is_a_new_engine(d, dy, T, C, monkeypatch):
    import rxhip
    from rxhip import workloads
    m = workloads.random_model(d, dy, seed=100 + d)
    ya, yb = workloads.generate_batch(m, T, C, seed0=1), workloads.generate_batch(m, T, C, seed0=2)
    mk = lambda: rxhip.LGSSMEngine(m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"], T=T, n_chains=C)
    monkeypatch.setenv("RXHIP_ENGINE_POOL", "0")
    with mk() as e:
        want_b = _run(e, yb)
    with mk() as e:
        want_f = _run(e, yb, filtering=True)
    monkeypatch.delenv("RXHIP_ENGINE_POOL")
    rxhip._lib.lib().rxhip_release_cached_memory()
    with mk() as e:                         # first life: another data set, a filtering run, profiling, the covariance option
        _run(e, ya)
        _run(e, ya, filtering=True)
        e.set_profiling(True); e.run(1, True); e.sync(); e.set_profiling(False)
        stages_new = e.create_stages()
    with mk() as e:                         # second life
        assert sum(e.create_stages().values()) == 0.0 and sum(stages_new.values()) > 0.0      # nothing was built: it IS the parked engine
        with pytest.raises(rxhip.RxHipError):
            e.run(1, True)                  # no data yet, as in a new engine
        got_b = _run(e, yb)
        assert all(v["launches"] <= 2 for v in e.kernel_times().values())                      # counters of this life only
    with mk() as e:                         # third life: a filtering run first
        got_f = _run(e, yb, filtering=True)
    for a, b in zip(want_b + want_f, got_b + got_f):
        assert np.array_equal(a, b)


def test_other_descriptors_do_not_hit_and_errors_do_not_park():
    import rxhip
    from rxhip import workloads
    m = workloads.random_model(2, 2, seed=5)
    T = 300
    y = workloads.generate_batch(m, T, 1, seed0=3)
    rxhip._lib.lib().rxhip_release_cached_memory()
    with rxhip.LGSSMEngine(m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"], T=T, n_chains=1) as e:
        base = _run(e, y)
    for kw in (dict(P=m["P"] * (1 + 1e-15)), dict(T=T - 1), dict(segments=3)):   # (the pool keeps four engines: the first one must survive these)
        args = dict(A=m["A"], B=m["B"], P=m["P"], Q=m["Q"], m0=m["m0"], V0=m["V0"], T=T, n_chains=1)
        args.update(kw)
        with rxhip.LGSSMEngine(**args) as e:
            assert sum(e.create_stages().values()) > 0.0, kw          # built, not revived
    with rxhip.LGSSMEngine(m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"], T=T, n_chains=1) as e:
        assert sum(e.create_stages().values()) == 0.0
        bad = y.copy(); bad[10] = np.inf
        e.set_data(bad)
        with pytest.raises(rxhip.RxHipError):
            e.run(1, True)
    with rxhip.LGSSMEngine(m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"], T=T, n_chains=1) as e:
        assert sum(e.create_stages().values()) > 0.0                   # the engine that failed was destroyed, not parked
        got = _run(e, y)
    for a, b in zip(base, got):
        assert np.array_equal(a, b)


def test_caching_off_means_no_process_wide_state():
    """rxhip_set_caching(0): the ABI's "handles are independent, no global state" (threading) to the letter — a destroyed engine is not parked, an
    engine of a model another live engine has is built from scratch (its own tables), and the results are the same bits; rxhip_set_caching(1) restores the pools."""
    import rxhip
    from rxhip import workloads
    L = rxhip._lib.lib()
    m4 = workloads.random_model(4, 4, seed=31)
    m24 = workloads.random_model(24, 6, seed=32)
    y4, y24 = workloads.generate_batch(m4, 500, 1, seed0=1), workloads.generate_batch(m24, 200, 1, seed0=2)
    mk4 = lambda: rxhip.LGSSMEngine(m4["A"], m4["B"], m4["P"], m4["Q"], m4["m0"], m4["V0"], T=500, n_chains=1)
    mk24 = lambda: rxhip.LGSSMEngine(m24["A"], m24["B"], m24["P"], m24["Q"], m24["m0"], m24["V0"], T=200, n_chains=1)
    with mk4() as e:
        want4 = _run(e, y4)
    with mk24() as e:
        want24 = _run(e, y24)
    try:
        assert L.rxhip_set_caching(0) == 0
        for _ in range(2):
            with mk4() as e:
                assert sum(e.create_stages().values()) > 0.0           # built every time: nothing was parked
                got4 = _run(e, y4)
        with mk24() as a, mk24() as b:                                 # two live engines of one model on the MFMA path: each with tables of its own
            assert sum(a.create_stages().values()) > 0.0 and sum(b.create_stages().values()) > 0.0
            assert b.create_stages()["tables_device_ms"] + b.create_stages()["tables_host_ms"] > 0.0
            got24 = _run(b, y24)
    finally:
        assert L.rxhip_set_caching(1) == 0
    for a, b in zip(want4 + want24, got4 + got24):
        assert np.array_equal(a, b)
    with mk4() as e:
        pass
    with mk4() as e:
        assert sum(e.create_stages().values()) == 0.0                  # the pools are back
