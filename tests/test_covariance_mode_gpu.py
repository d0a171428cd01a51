"""rxhip_set_covariance_mode(1): shared-model batches on the MFMA path keep ONE covariance table per model and write the per-chain
posterior array on request.  Whatever is asked for, in whatever order, must equal what mode 0 (every sweep writes every chain) returns."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _engine(d, dy, T, C, seed=5):
    import rxhip
    from rxhip import workloads
    m = workloads.random_model(d, dy, seed=seed)
    y = workloads.generate_batch(m, T, C, seed0=3)
    eng = rxhip.LGSSMEngine(m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"], T=T, n_chains=C)
    eng.set_data(y)
    return eng, m, y


@pytest.mark.parametrize("d,dy,T,C", [(8, 8, 120, 64), (16, 5, 90, 16), (40, 12, 64, 8), (64, 64, 48, 4)])
def test_on_request_equals_every_sweep(d, dy, T, C):
    eng, m, y = _engine(d, dy, T, C)
    with eng:
        eng.run(1, True)
        mean0, cov0 = eng.marginals()
        fe0 = eng.free_energy_per_chain()
        pm0, pc0 = eng.predictions()
        eng.set_covariance_mode(1)
        eng.run(1, True)                                   # the array is current: nothing to write
        mean1, cov1 = eng.marginals()
        assert np.array_equal(mean1, mean0) and np.array_equal(cov1, cov0)
        eng.run_filter(True)                               # rewrites the arrays with filtered beliefs
        fm, fc = eng.marginals()
        assert not np.array_equal(fc, cov0)
        eng.run(2, True)                                   # smoothing again: the per-chain covariances are pending …
        assert np.array_equal(eng.free_energy_per_chain(), fe0)
        pm1, pc1 = eng.predictions()                       # … and written when the predictions need them
        assert np.array_equal(pm1, pm0) and np.array_equal(pc1, pc0)
        mean2, cov2 = eng.marginals()
        assert np.array_equal(mean2, mean0) and np.array_equal(cov2, cov0)
        eng.run_filter(True)
        eng.run(1, True)
        sub_mean, sub_cov = eng.marginals_of_chains([C - 1, 0])   # chain-major: [2][T][d][d]
        assert np.array_equal(sub_cov[0], cov0[:, C - 1]) and np.array_equal(sub_cov[1], cov0[:, 0])
        eng.set_covariance_mode(0)
        eng.run(1, True)
        assert np.array_equal(eng.marginals()[1], cov0)


def test_engines_without_a_shared_table_ignore_the_mode():
    import rxhip
    from rxhip import workloads
    m = workloads.c1_model()
    y = workloads.generate_batch(m, 200, 8, seed0=1)
    with rxhip.LGSSMEngine(m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"], T=200, n_chains=8) as eng:
        eng.set_data(y)
        eng.run(1, True)
        ref = eng.marginals()
        eng.set_covariance_mode(1)
        eng.run(1, True)
        out = eng.marginals()
        assert np.array_equal(out[0], ref[0]) and np.array_equal(out[1], ref[1])
        with pytest.raises(Exception):
            eng.set_covariance_mode(2)
