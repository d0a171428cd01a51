"""Replays of the randomized differential campaign (scripts/fuzz_executor.py; tests/fuzz_cases.py holds the case generator): the seeds that found defects in
round 6, and a short run of fresh ones on every suite run.

  1301, 2084  a `missing` observation behind `*`(:out) / `+`(:out): the zero of the precision form reached a rule that works on moments and was inverted
              (NOT_POSDEF where the reference drops the message) — now the moment form of "no information" (T_ABSENT_VARIANCE, csrc/tree_kernels.hpp load_msg)
  3902        the unobserved end of a `*` / `+` chain at d = 48: the marginal of its ONE moment-form message went through the precision and back, and the two
              sweep inverses squared the condition number of A V Aᵀ in the error (7e-6 sd against the oracle's 4e-10) — now the message itself; and the
              LDS-staged kernels read one triangle of an inverse whose sweep is not symmetric — now the mean of the two

Chain cases (the pattern-matched state-space engines against the executor on the same descriptor, run_chain_case):
  101839, 119783  `y ~ N(B x + c, Q)` behind a square B of condition 3e5 / 2e4 at d = 48 / 33: the four-pivot MFMA sweep of the LDS-staged kernels lost q(B x + c)
              (0.3 sd; free energy 1e-3) where the one-pivot sweep is exact to 1e-10 — now guarded by the largest variance inflation a_kk (A⁻¹)_kk, redone one pivot
              at a time above 1e3 (csrc/tree_wave_kernels.hpp spd_inv_blocked)
  101987      the same B with `missing` observations: the marginal of one moment-form message next to zeros of the precision form is that message"""
import numpy as np
import pytest

import tree_graphs as tg
from fuzz_cases import run_case

pytestmark = pytest.mark.gpu


def _replay(seed, monkeypatch):
    monkeypatch.setenv("RXHIP_TEST_HOOKS", "1")
    monkeypatch.setenv("RXHIP_TREE_MODE", "0")   # (registered with monkeypatch so that what run_case sets is undone)
    monkeypatch.setenv("RXHIP_TREE_TILE", "0")
    return run_case(seed)


@pytest.mark.parametrize("seed", [1301, 2084, 3902])
def test_seeds_that_found_defects(seed, monkeypatch):
    assert _replay(seed, monkeypatch) is None


@pytest.mark.parametrize("seed", [101839, 101987, 119783])
def test_chain_seeds_that_found_defects(seed, monkeypatch):
    monkeypatch.setenv("RXHIP_TREE_MODE", "0")
    monkeypatch.setenv("RXHIP_TREE_TILE", "0")
    from fuzz_cases import run_chain_case
    assert run_chain_case(seed) is None


@pytest.mark.parametrize("first", [9000, 9060])
def test_sixty_fresh_chain_cases(first, monkeypatch):
    monkeypatch.setenv("RXHIP_TREE_MODE", "0")
    monkeypatch.setenv("RXHIP_TREE_TILE", "0")
    from fuzz_cases import run_chain_case
    findings = [f for f in (run_chain_case(s) for s in range(first, first + 60)) if f]
    assert not findings, findings


@pytest.mark.parametrize("seed", [2592, 2851, 2881, 2962, 21574])
def test_engine_seeds_that_found_the_conditioning_defect(seed):
    """run_engine_case: models on which the information-form MFMA path was wrong by 0.04 … 2.7 sd (smoothing; 2962, 21574: filtering) — refused at creation now
    (csrc/model_envelope.hpp), which the case counts as a refusal by name"""
    from fuzz_cases import run_engine_case
    assert run_engine_case(seed) is None


@pytest.mark.parametrize("first", [11000, 11100])
def test_a_hundred_fresh_engine_cases(first):
    from fuzz_cases import run_engine_case
    findings = [f for f in (run_engine_case(s) for s in range(first, first + 100)) if f]
    assert not findings, findings


def test_two_hundred_fresh_cases_of_the_engine_options():
    """run_option_case: known inputs per engine and per chain, per-step constants, forecast horizons with predictions, node-local joints, per-chain models, the step-wise
    filter, rxhip_lgssm_infer, the unknown-noise chain — against the oracle's restatements and, for partly observed states, a covariance-form numpy smoother"""
    from fuzz_cases import run_option_case
    findings = [f for f in (run_option_case(s) for s in range(60000, 60200)) if f]
    assert not findings, findings


def test_two_hundred_fresh_cases_of_the_variational_engines():
    """run_vmp_case: univariate / multivariate mixtures and the hierarchical Gaussian filter at random sizes, priors and iteration counts against the oracle's
    restatements, every iteration (20 000 cases on the GPU without a finding, profiles/r06/fuzz_campaign.txt)"""
    from fuzz_cases import run_vmp_case
    findings = [f for f in (run_vmp_case(s) for s in range(50000, 50200)) if f]
    assert not findings, findings


@pytest.mark.parametrize("first", [7000, 7040, 7080])
def test_forty_fresh_cases(first, monkeypatch):
    findings = [f for f in (_replay(s, monkeypatch) for s in range(first, first + 40)) if f]
    assert not findings, findings


@pytest.mark.parametrize("d", [4, 12, 24, 48, 64])
def test_unobserved_end_of_a_deterministic_chain_is_the_forward_message(d, monkeypatch):
    """x ~ N(m, V), y ~ N(x, Σ) observed, s = A x + c with a square random A, leaf ~ N(s, W⁻¹) never observed: q(s) is the one message that reaches s — exact to
    the conditioning of ONE pass over A V Aᵀ, not two inversions of it.  Against brute-force conditioning of the joint Gaussian."""
    from rxhip import _lib
    from rxhip.tree import TreeEngine
    rng = np.random.default_rng(d)
    gb = tg.GraphBuilder()
    x = gb.randomvar(d)
    gb.mvnormal_mean_cov(x, gb.constvar(rng.standard_normal(d)), gb.constvar(tg._spd(rng, d, 3.0)))
    y = gb.datavar(d)
    gb.mvnormal_mean_cov(y, x, gb.constvar(tg._spd(rng, d, 1.0)))
    a, s, leaf = gb.randomvar(d), gb.randomvar(d), gb.randomvar(d)
    gb.multiply(a, gb.constvar(rng.standard_normal((d, d))), x)
    gb.node(_lib.NODE_ADD, s, a, gb.constvar(rng.standard_normal(d)))
    gb.node(_lib.NODE_MVNORMAL_MEAN_PRECISION, leaf, s, gb.constvar(np.linalg.inv(tg._spd(rng, d, 1.0))))
    data = tg.random_data(gb, [y], 2, d)
    monkeypatch.delenv("RXHIP_TREE_MODE", raising=False)
    monkeypatch.delenv("RXHIP_TREE_TILE", raising=False)
    with TreeEngine(gb, n_replicas=2) as eng:
        eng.set_data([y], data)
        eng.run(1, True)
        post = eng.marginals([x, s, leaf])
    bf, _ = tg.brute_force(gb, tg.data_dict(gb, [y], data[1]))
    for v in (x, s, leaf):
        sd = np.sqrt(np.diag(bf[v][1]))
        assert np.max(np.abs(post[v][0][1] - bf[v][0]) / sd) < 1e-9
        assert np.max(np.abs(post[v][1][1] - bf[v][1]) / np.outer(sd, sd)) < 1e-9
