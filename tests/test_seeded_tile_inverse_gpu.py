"""kd_forward_info at d ≥ 48 seeds the in-wave inverse of each 16×16 pivot tile with the one of the previous time step (one Newton – Schulz
step on the accumulator registers, determinant through the trace of the residual — dense_kernels.hpp, DiagSeed) wherever the residual is
below 10⁻⁸, and runs the exact rank-4 rounds elsewhere.  Chains long enough for the Riccati recursion to converge (almost every step
seeded), chains that never settle (slow modes: the exact rounds most of the time), badly scaled states (the equilibration exponents move),
and different segmentations of the same chain, all against the CPU oracle's smoother and −log p(y)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(m, y, segments=0, tol_m=1e-9, tol_c=1e-10, tol_fe=1e-11):
    import rxhip
    import rxoracle as rxo
    T = y.shape[0]
    with rxhip.LGSSMEngine(m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"], T=T, n_chains=y.shape[1], segments=segments) as eng:
        eng.set_data(y)
        eng.run(1, True)
        mean, cov = eng.marginals()
        fe = eng.free_energy_per_chain()
        sched = eng.schedule()
    for c in range(y.shape[1]):
        om, oc, nll = rxo.lgssm_kalman_rts(m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"], np.ascontiguousarray(y[:, c]))
        sd = np.sqrt(np.einsum("tii->ti", oc))
        assert np.max(np.abs(mean[:, c] - om) / sd) < tol_m, (c, sched)
        assert np.max(np.abs(cov[:, c] - oc) / (sd[:, :, None] * sd[:, None, :])) < tol_c, (c, sched)
        assert abs(fe[c] - nll) <= tol_fe * abs(nll), (c, sched, fe[c], nll)
    return sched


@pytest.mark.parametrize("d,dy,T,segments", [(64, 64, 2400, 0), (64, 64, 2400, 30), (64, 64, 900, 256), (48, 48, 1500, 0), (64, 20, 1200, 0), (56, 56, 800, 16)])
def test_converged_chains_against_the_oracle(d, dy, T, segments):
    from rxhip import workloads
    m = workloads.c3_model() if (d, dy) == (64, 64) else workloads.random_model(d, dy, seed=3 * d + dy)
    y = workloads.generate_batch(m, T, 1, seed0=64 + d)
    _check(m, y, segments)


def test_slow_modes_never_settle():
    """A spectral radius of 0.9995 and a small process noise: the filter covariance is still moving at the tenth digit after thousands of steps."""
    from rxhip import workloads
    d = 64
    m = workloads.random_model(d, d, seed=5)
    rng = np.random.default_rng(8)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    A = q @ np.diag(np.linspace(0.5, 0.9995, d)) @ q.T
    m = dict(m, A=A, P=1e-4 * np.eye(d), V0=25.0 * np.eye(d))
    y = workloads.generate_batch(m, 1500, 1, seed0=9)
    _check(m, y, tol_m=1e-8, tol_c=1e-9, tol_fe=1e-10)


def test_scaled_states_and_several_chains():
    from rxhip import workloads
    d = 64
    m0 = workloads.random_model(d, d, seed=77)
    s = 10.0 ** np.random.default_rng(78).uniform(-2.0, 2.0, d)
    S, Si = np.diag(s), np.diag(1.0 / s)
    m = dict(A=S @ m0["A"] @ Si, B=m0["B"] @ Si, P=S @ m0["P"] @ S, Q=m0["Q"], m0=s * m0["m0"], V0=S @ m0["V0"] @ S)
    y = workloads.generate_batch(m0, 700, 3, seed0=4)
    _check(m, y, tol_m=1e-7, tol_c=1e-7, tol_fe=1e-10)


def test_free_energy_is_the_same_on_every_segmentation():
    """The seeded steps book log det through a trace, the exact ones through pivots: which steps are which depends on where the segments start."""
    import rxhip
    from rxhip import workloads
    m = workloads.c3_model()
    y = workloads.generate_batch(m, 3000, 1, seed0=11)
    fes = []
    for segments in (0, 7, 64, 300):
        with rxhip.LGSSMEngine(m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"], T=3000, n_chains=1, segments=segments) as eng:
            eng.set_data(y)
            eng.run(1, True)
            fes.append(eng.free_energy_per_chain()[0])
    assert np.max(np.abs(np.array(fes) - fes[0])) <= 1e-12 * abs(fes[0]), fes
