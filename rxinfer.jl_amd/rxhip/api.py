"""`infer(model=..., data=..., ...)` — Python mirror of RxInfer's front door for the hot path.

Mirrors src/inference/inference.jl:577-609 (keyword names, `free_energy`, `iterations`,
`options`, `catch_exception`) and `InferenceResult` (src/inference/batch.jl:18-24) for the model
family the device schedule covers.  The Julia host shim (rxinfer.jl_amd/julia/RxHip.jl) binds the
same C ABI; this mirror exists because no Julia toolchain is present in the build image, so the
parity tests are written against it in the reference tests' own shape
(test/models/statespace/mlgssm_test.jl)."""
from dataclasses import dataclass
from typing import Any, Optional

import numpy as np

from .engine import LGSSMEngine


@dataclass
class MvNormalMeanCovariance:
    """ExponentialFamily.MvNormalMeanCovariance(μ, Σ) — the parametrisation posteriors are returned in."""
    mean: np.ndarray
    cov: np.ndarray


@dataclass
class LinearGaussianSSM:
    """Model specification produced by `linear_gaussian_ssm(...)` — the lowered form of
        x[1] ~ MvNormal(μ = m0, Σ = V0); y[t] ~ MvNormal(μ = B*x[t], Σ = Q);
        x[t] ~ MvNormal(μ = A*x[t-1], Σ = P)            (benchmarks notebook, cell 4)."""
    A: np.ndarray
    B: np.ndarray
    P: np.ndarray
    Q: np.ndarray
    prior_mean: np.ndarray
    prior_cov: np.ndarray
    prior_through_transition: bool = False


def linear_gaussian_ssm(A, B, P, Q, prior_mean, prior_cov, prior_through_transition=False):
    f = lambda a: np.asarray(a, dtype=np.float64)
    return LinearGaussianSSM(f(A), f(B), f(P), f(Q), f(prior_mean), f(prior_cov), bool(prior_through_transition))


@dataclass
class InferenceResult:
    """src/inference/batch.jl:18-24"""
    posteriors: dict
    predictions: Optional[dict]
    free_energy: Optional[np.ndarray]
    model: Any
    error: Optional[BaseException] = None


_OPTION_KEYS = {"limit_stack_depth", "warn", "device", "segments", "backend"}


def infer(*, model, data, iterations=None, free_energy=False, options=None, returnvars=None,
          catch_exception=False):
    """Static (batch) inference on the device engine.

    data = {"y": array}: [T][dy] for one chain (as `data = (y = observations,)`), or
    [chain][T][dy] for a batch of independent chains sharing the model.
    Returns posteriors["x"] as MvNormalMeanCovariance with mean [T][d] / [chain][T][d]."""
    if not isinstance(model, LinearGaussianSSM):
        raise TypeError("infer: no device schedule for this model type")
    options = dict(options or {})
    unknown = set(options) - _OPTION_KEYS
    if unknown:  # closed key set, as reactivemp_inference.jl:129-143
        raise ValueError(f"Unknown option keys {sorted(unknown)}; available: {sorted(_OPTION_KEYS)}")
    if "y" not in data:
        raise ValueError("data must provide `y`")
    y = np.asarray(data["y"], dtype=np.float64)
    single = y.ndim == 2
    if single:
        y = y[None]
    C, T, dy = y.shape
    iters = 1 if iterations is None else int(iterations)
    eng = None
    try:
        eng = LGSSMEngine(model.A, model.B, model.P, model.Q, model.prior_mean, model.prior_cov, T=T, n_chains=C,
                          prior_through_transition=model.prior_through_transition,
                          segments=int(options.get("segments", 0)), device=int(options.get("device", -1)))
        eng.set_data(y, layout="chain_time")
        eng.run(iterations=iters, free_energy=free_energy)
        mean, cov = eng.marginals(layout="chain_time")
        if free_energy:
            # one value per iteration, per chain-graph: what `infer` returns for each chain
            fe = eng.free_energy_per_chain()
            fe = np.repeat(fe[:, None], iters, axis=1)
        else:
            fe = None
        if single:
            mean, cov = mean[0], cov[0]
            fe = fe[0] if fe is not None else None
        post = {"x": MvNormalMeanCovariance(mean, cov)}
        if returnvars is not None:
            post = {k: v for k, v in post.items() if k in returnvars}
        return InferenceResult(post, None, fe, model, None)
    except Exception as err:  # catch_exception semantics of batch.jl:440-446
        if not catch_exception:
            raise
        return InferenceResult({}, None, None, model, err)
    finally:
        if eng is not None:
            eng.close()
