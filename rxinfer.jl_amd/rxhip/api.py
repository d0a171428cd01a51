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

from . import _lib
from ._lib import RxHipError
from .engine import DriftChainEngine, GMMEngine, HGFEngine, LGSSMEngine, MvGMMEngine


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
    step_model: Optional[np.ndarray] = None   # time-varying constants: A, B, P, Q are [n_models, …], step_model[t] picks
    state_offset: Optional[np.ndarray] = None  # known inputs: x[t] ~ MvNormal(μ = A*x[t-1] + c[t], Σ = P); [d] or [T, d]
    obs_offset: Optional[np.ndarray] = None    # y[t] ~ MvNormal(μ = B*x[t] + d[t], Σ = Q); [dy] or [T, dy]
    input_matrix: Optional[np.ndarray] = None  # B_u of x[t] ~ MvNormal(μ = A*x[t-1] + B_u*u[t], Σ = P), u a data variable
    noise_precision_prior: Any = None          # Wishart(ν, S): y[t] ~ MvNormal(μ = B*x[t], Λ = W), W ~ Wishart(ν, S), q(x, W) = q(x)q(W); Q is unused


def linear_gaussian_ssm(A, B, P, Q, prior_mean, prior_cov, prior_through_transition=False, state_offset=None, obs_offset=None,
                        input_matrix=None, noise_precision_prior=None):
    """`state_offset` / `obs_offset`: known inputs added to the means (`A * x[t-1] + c`, `B * x[t] + d`), one vector or one
    per time index.  `input_matrix` = B_u: the inputs are DATA, `infer(..., data = {"y": …, "u": …})` with u [T][du] per chain.
    `noise_precision_prior = Wishart(ν, S)` (Q = None): the observation noise is UNKNOWN, `y[t] ~ MvNormal(μ = B * x[t], Λ = W)` with
    `W ~ Wishart(ν, S)` and `@constraints q(x, W) = q(x)q(W)` — `infer(..., iterations = n, initialization = {"W": Wishart(ν, V)})` runs mean-field
    VMP on the device (d, dy ≤ 4) and returns posteriors["W"] next to posteriors["x"]."""
    f = lambda a: np.asarray(a, dtype=np.float64)
    g = lambda a: None if a is None else f(a)
    if noise_precision_prior is not None:
        if Q is not None or state_offset is not None or obs_offset is not None or input_matrix is not None:
            raise ValueError("an unknown observation precision takes Q = None and no offsets / inputs")
        Q = np.eye(np.asarray(B).shape[0])
    return LinearGaussianSSM(f(A), f(B), f(P), f(Q), f(prior_mean), f(prior_cov), bool(prior_through_transition),
                             state_offset=g(state_offset), obs_offset=g(obs_offset), input_matrix=g(input_matrix),
                             noise_precision_prior=noise_precision_prior)


def time_varying_gaussian_ssm(A, B, P, Q, prior_mean, prior_cov, prior_through_transition=False):
    """The state-space model with per-step constants, as written in a @model loop over arrays of matrices:
        x[t] ~ MvNormal(μ = A[t] * x[t-1], Σ = P[t]);  y[t] ~ MvNormal(μ = B[t] * x[t], Σ = Q[t]).
    Any of A, B, P, Q may carry a leading time axis [T, …] (the others are held constant); equal steps share one model."""
    f = lambda a: np.asarray(a, dtype=np.float64)
    A, B, P, Q = f(A), f(B), f(P), f(Q)
    lens = {a.shape[0] for a in (A, B, P, Q) if a.ndim == 3}
    if len(lens) != 1:
        raise ValueError("time-varying constants: give at least one of A, B, P, Q as [T, …], all with the same T")
    T = lens.pop()
    full = [a if a.ndim == 3 else np.broadcast_to(a, (T,) + a.shape) for a in (A, B, P, Q)]
    keys = np.concatenate([a.reshape(T, -1) for a in full], axis=1)
    _, first, step_model = np.unique(keys, axis=0, return_index=True, return_inverse=True)
    A, B, P, Q = (np.ascontiguousarray(a[first]) for a in full)
    return LinearGaussianSSM(A, B, P, Q, f(prior_mean), f(prior_cov), bool(prior_through_transition),
                             np.ascontiguousarray(step_model.ravel(), dtype=np.int32))


@dataclass
class NormalMeanVariance:
    mean: np.ndarray
    var: np.ndarray


@dataclass
class GammaShapeRate:
    shape: np.ndarray
    rate: np.ndarray


@dataclass
class GammaShapeScale:
    """ExponentialFamily.GammaShapeScale (= Distributions.Gamma(α, θ)); `.rate` is 1/θ"""
    shape: np.ndarray
    scale: np.ndarray

    @property
    def rate(self):
        return 1.0 / np.asarray(self.scale, dtype=np.float64)


@dataclass
class Dirichlet:
    alpha: np.ndarray


@dataclass
class UnivariateGaussianMixture:
    """`s ~ Dirichlet(alpha0); m[k] ~ Normal(mean, variance); p[k] ~ Gamma(shape, rate); z[i] ~ Categorical(s);
    y[i] ~ NormalMixture(switch = z[i], m = m, p = p)` with MeanField() constraints
    (test/models/mixtures/gmm_univariate_tests.jl:7-26).  K = 1: the iid Gaussian with unknown mean and precision
    (test/models/models_tests.jl:114-128)."""
    prior_mean: np.ndarray
    prior_var: np.ndarray
    prior_shape: np.ndarray
    prior_rate: np.ndarray
    prior_alpha: np.ndarray


def gaussian_mixture(prior_mean, prior_var, prior_shape, prior_rate, prior_alpha=None):
    f = lambda a: np.atleast_1d(np.asarray(a, dtype=np.float64))
    pm = f(prior_mean)
    return UnivariateGaussianMixture(pm, f(prior_var), f(prior_shape), f(prior_rate),
                                     np.ones_like(pm) if prior_alpha is None else f(prior_alpha))


def iid_normal_gamma(mean, variance, shape, rate=None, *, scale=None):
    """`μ ~ Normal(mean, variance); τ ~ Gamma(shape, rate); y[i] ~ Normal(mean = μ, precision = τ)`; `scale = θ` is the
    `Gamma(shape = …, scale = …)` spelling of test/models/models_tests.jl:121-127 (rate = 1/θ)."""
    if (rate is None) == (scale is None):
        raise ValueError("give exactly one of rate and scale")
    return gaussian_mixture([mean], [variance], [shape], [rate if scale is None else 1.0 / scale], [1.0])


@dataclass
class Wishart:
    """ExponentialFamily.Wishart(ν, S): degrees of freedom and scale matrix (arrays over components allowed)"""
    nu: Any
    S: Any


@dataclass
class MultivariateGaussianMixture:
    """test/models/mixtures/gmm_multivariate_tests.jl:6-32: m[k] ~ MvNormal(mean, cov), w[k] ~ Wishart(ν, V), s ~ Dirichlet"""
    prior_mean: np.ndarray   # [K][d]
    prior_cov: np.ndarray    # [K][d][d]
    prior_nu: np.ndarray     # [K]
    prior_scale: np.ndarray  # [K][d][d]
    prior_alpha: np.ndarray  # [K]


def multivariate_gaussian_mixture(prior_mean, prior_cov, prior_nu, prior_scale, prior_alpha=None):
    m = np.asarray(prior_mean, dtype=np.float64)
    K = m.shape[0]
    return MultivariateGaussianMixture(m, np.asarray(prior_cov, dtype=np.float64), np.asarray(prior_nu, dtype=np.float64),
                                       np.asarray(prior_scale, dtype=np.float64),
                                       np.ones(K) if prior_alpha is None else np.asarray(prior_alpha, dtype=np.float64))


@dataclass
class HierarchicalGaussianFilter:
    """One-step HGF graph streamed over the observations with posterior→prior autoupdates
    (test/models/statespace/hgf_tests.jl:9-70)."""
    kappa: float
    omega: float
    z_variance: float
    y_variance: float
    n_gh: int = 31


def hierarchical_gaussian_filter(kappa, omega, z_variance, y_variance, n_gh=31):
    return HierarchicalGaussianFilter(float(kappa), float(omega), float(z_variance), float(y_variance), int(n_gh))


@dataclass
class UnivariateDriftChain:
    """`x_prior ~ Normal(μ = mean(x0), v = var(x0)); x[i] ~ x_prev + c; y[i] ~ Normal(μ = x[i], v = P)`
    (test/models/statespace/ulgssm_tests.jl:8-15): noise-free transitions through `typeof(+)` nodes."""
    prior_mean: float
    prior_var: float
    c: float
    obs_var: float
    prior_through_transition: bool = True


def univariate_drift_chain(prior_mean, prior_var, c, obs_var, prior_through_transition=True):
    return UnivariateDriftChain(float(prior_mean), float(prior_var), float(c), float(obs_var), bool(prior_through_transition))


def _infer_drift_chain(model, data, iterations, free_energy, options, catch_exception):
    """posteriors["x"] = NormalMeanVariance with mean / var [T] (or [chain][T]); free_energy one value per iteration."""
    options = _check_options(options)
    y = np.asarray(data["y"], dtype=np.float64)
    single = y.ndim == 1
    if single:
        y = y[None]
    C, T = y.shape
    iters = 1 if iterations is None else int(iterations)
    eng = None
    try:
        eng = DriftChainEngine(T, model.prior_mean, model.prior_var, model.c, model.obs_var, n_chains=C,
                               prior_through_transition=model.prior_through_transition, device=int(options.get("device", -1)))
        eng.set_data(y[..., None], layout="chain_time")
        eng.run(iterations=iters, free_energy=free_energy)
        mean, var = eng.marginals(layout="chain_time")
        mean, var = mean[..., 0], var[..., 0, 0]
        fe = np.repeat(eng.free_energy_per_chain()[:, None], iters, axis=1) if free_energy else None
        if single:
            mean, var = mean[0], var[0]
            fe = fe[0] if fe is not None else None
        return InferenceResult({"x": NormalMeanVariance(mean, var)}, None, fe, model, None)
    except Exception as err:
        if not catch_exception:
            raise
        return InferenceResult({}, None, None, model, err)
    finally:
        if eng is not None:
            eng.close()


@dataclass
class InferenceResult:
    """src/inference/batch.jl:18-24"""
    posteriors: dict
    predictions: Optional[dict]
    free_energy: Optional[np.ndarray]
    model: Any
    error: Optional[BaseException] = None
    # streaming runs (src/inference/streaming.jl:18-30): history of the `historyvars`, free_energy_history
    history: Optional[dict] = None
    free_energy_history: Optional[np.ndarray] = None


_OPTION_KEYS = {"limit_stack_depth", "warn", "device", "segments", "backend", "materialize_z"}


def _check_options(options):
    options = dict(options or {})
    unknown = set(options) - _OPTION_KEYS
    if unknown:
        raise ValueError(f"Unknown option keys {sorted(unknown)}; available: {sorted(_OPTION_KEYS)}")
    return options


def _infer_mixture(model, data, iterations, free_energy, options, initialization, catch_exception):
    """posteriors (KeepEach, one entry per iteration as `returnvars = KeepEach()`): m -> NormalMeanVariance [it][K],
    p -> GammaShapeRate [it][K], s -> Dirichlet [it][K]; z (last iteration) if options['materialize_z'].
    Only the fixed point is pinned to the reference: the update order inside an iteration is an assumption, so per-iteration
    entries are drop-in for RxInfer's `KeepEach` results up to that order (include/rxhip.h)."""
    options = _check_options(options)
    if initialization is None:
        raise ValueError("mean-field VMP needs `initialization` (q(m), q(p), q(s)); cf. test/inference/inference_tests.jl:361-363")
    y = np.asarray(data["y"], dtype=np.float64).ravel()
    iters = 1 if iterations is None else int(iterations)
    K = model.prior_mean.size
    qm, qp = initialization["m"], initialization["p"]
    qs = initialization.get("s", Dirichlet(np.ones(K)))
    eng = None
    try:
        eng = GMMEngine(y.size, model.prior_mean, model.prior_var, model.prior_shape, model.prior_rate, model.prior_alpha,
                        qm.mean, qm.var, qp.shape, qp.rate, qs.alpha,
                        materialize_responsibilities=bool(options.get("materialize_z", False)), device=int(options.get("device", -1)))
        eng.set_data(y)
        eng.run(iters, free_energy)
        h = eng.history()
        post = {"m": NormalMeanVariance(h[:, 0], h[:, 1]), "p": GammaShapeRate(h[:, 2], h[:, 3]), "s": Dirichlet(h[:, 4])}
        if options.get("materialize_z", False):
            post["z"] = eng.responsibilities()
        return InferenceResult(post, None, eng.free_energy() if free_energy else None, model, None)
    except Exception as err:
        if not catch_exception:
            raise
        return InferenceResult({}, None, None, model, err)
    finally:
        if eng is not None:
            eng.close()


def _infer_mv_mixture(model, data, iterations, free_energy, options, initialization, catch_exception):
    """posteriors (KeepEach): m -> MvNormalMeanCovariance [it][K], w -> Wishart [it][K], s -> Dirichlet [it][K];
    z (last iteration) if options['materialize_z'].  initialization = {"m": MvNormalMeanCovariance, "w": Wishart, "s": Dirichlet}."""
    options = _check_options(options)
    if initialization is None or "m" not in initialization or "w" not in initialization:
        raise ValueError("mean-field VMP needs `initialization` (q(m), q(w)[, q(s)]); cf. test/models/mixtures/gmm_multivariate_tests.jl:66-70")
    y = np.asarray(data["y"], dtype=np.float64)
    iters = 1 if iterations is None else int(iterations)
    K = model.prior_mean.shape[0]
    qm, qw = initialization["m"], initialization["w"]
    qs = initialization.get("s", Dirichlet(np.ones(K)))
    eng = None
    try:
        eng = MvGMMEngine(y.shape[0], model.prior_mean, model.prior_cov, model.prior_nu, model.prior_scale, model.prior_alpha,
                          qm.mean, qm.cov, np.broadcast_to(np.asarray(qw.nu, dtype=np.float64), (K,)).copy(), qw.S, qs.alpha,
                          materialize_responsibilities=bool(options.get("materialize_z", False)), device=int(options.get("device", -1)))
        eng.set_data(y)
        eng.run(iters, free_energy)
        h = eng.history()
        post = {"m": MvNormalMeanCovariance(h["mean"], h["cov"]), "w": Wishart(h["nu"], h["V"]), "s": Dirichlet(h["alpha"])}
        if options.get("materialize_z", False):
            post["z"] = eng.responsibilities()
        return InferenceResult(post, None, eng.free_energy() if free_energy else None, model, None)
    except Exception as err:
        if not catch_exception:
            raise
        return InferenceResult({}, None, None, model, err)
    finally:
        if eng is not None:
            eng.close()


def _infer_hgf(model, data, iterations, free_energy, options, initialization, catch_exception):
    """history[:zt], history[:xt] (KeepLast per observation) as NormalMeanVariance [T] (or [series][T]);
    free_energy = free_energy_history: mean over observations per VMP iteration (per series)."""
    options = _check_options(options)
    y = np.asarray(data["y"], dtype=np.float64)
    single = y.ndim == 1
    if single:
        y = y[None]
    S, T = y.shape
    iters = 1 if iterations is None else int(iterations)
    init = initialization or {}
    z0 = init.get("zt", NormalMeanVariance(0.0, 5.0))
    x0 = init.get("xt", NormalMeanVariance(0.0, 5.0))
    eng = None
    try:
        eng = HGFEngine(T, S, model.kappa, model.omega, model.z_variance, model.y_variance, (float(z0.mean), float(z0.var)),
                        (float(x0.mean), float(x0.var)), n_gh=model.n_gh, device=int(options.get("device", -1)))
        eng.set_data(y, layout="chain_time")
        eng.run(iters, free_energy)
        zm, zv, xm, xv = eng.history("chain_time")
        fe = None
        if free_energy:
            fe = eng.free_energy() / S if single else None
            if not single:
                fe = eng.free_energy_per_chain()
        if single:
            zm, zv, xm, xv = zm[0], zv[0], xm[0], xv[0]
        return InferenceResult({"zt": NormalMeanVariance(zm, zv), "xt": NormalMeanVariance(xm, xv)}, None, fe, model, None)
    except Exception as err:
        if not catch_exception:
            raise
        return InferenceResult({}, None, None, model, err)
    finally:
        if eng is not None:
            eng.close()


def _infer_lgssm_filtering(model, data, free_energy, options, initialization, catch_exception):
    """Streaming inference with posterior -> prior feedback (`autoupdates`), the notebook's
    `rxinfer_inference_filtering` (benchmarks notebook cell 7): history["x"] = q(x_t) after every observation
    (`historyvars = (x_t = KeepLast(),)`, `keephistory = T`); free_energy_history = mean over observations of the
    per-observation free energy, one value (iterations = 1) per chain.  `initialization = {"x": MvNormalMeanCovariance}`
    overrides the model's prior as the initial q(x_t)."""
    options = _check_options(options)
    y = np.asarray(data["y"], dtype=np.float64)
    single = y.ndim == 2
    if single:
        y = y[None]
    C, T, dy = y.shape
    m0, V0 = model.prior_mean, model.prior_cov
    if initialization and "x" in initialization:
        m0, V0 = initialization["x"].mean, initialization["x"].cov
    eng = None
    try:
        eng = LGSSMEngine(model.A, model.B, model.P, model.Q, m0, V0, T=T, n_chains=C,
                          prior_through_transition=model.prior_through_transition,
                          state_offset=model.state_offset, obs_offset=model.obs_offset,
                          segments=int(options.get("segments", 0)), device=int(options.get("device", -1)))
        if single:   # one series: the whole call in one round trip
            m1, c1, f1 = eng.infer(y[0][:, None, :], free_energy=free_energy, filtering=True)
            mean, cov = np.transpose(m1, (1, 0, 2)), np.transpose(c1, (1, 0, 2, 3))
            fe = f1[:, None] if free_energy else None
        else:
            eng.set_data(y, layout="chain_time")
            eng.run_filter(free_energy=free_energy)
            mean, cov = eng.marginals(layout="chain_time")
            fe = eng.free_energy_per_chain()[:, None] if free_energy else None
        if single:
            mean, cov = mean[0], cov[0]
            fe = fe[0] if fe is not None else None
        return InferenceResult({}, None, None, model, None, history={"x": MvNormalMeanCovariance(mean, cov)},
                               free_energy_history=fe)
    except Exception as err:
        if not catch_exception:
            raise
        return InferenceResult({}, None, None, model, err)
    finally:
        if eng is not None:
            eng.close()


def _infer_lgssm_noise(model, data, iterations, free_energy, options, initialization, returnvars, catch_exception):
    """mean-field VMP of the chain with an unknown observation-noise precision (LGSSMNoiseEngine)"""
    from .engine import LGSSMNoiseEngine
    options = dict(options or {})
    unknown = set(options) - _OPTION_KEYS
    if unknown:
        raise ValueError(f"Unknown option keys {sorted(unknown)}; available: {sorted(_OPTION_KEYS)}")
    eng = None
    try:
        y = np.asarray(data["y"], dtype=np.float64)
        single = y.ndim == 2
        if single:
            y = y[None]
        if np.isnan(y).any():
            raise ValueError("unknown observation precision: missing observations have no device schedule")
        C, T, dy = y.shape
        pri = model.noise_precision_prior
        init = (initialization or {}).get("W")
        if init is None:   # the reference refuses mean-field VMP without an @initialization marginal
            raise ValueError("mean-field VMP needs initialization = {\"W\": Wishart(ν, V)}")
        iters = 1 if iterations is None else int(iterations)
        eng = LGSSMNoiseEngine(model.A, model.B, model.P, model.prior_mean, model.prior_cov, T, float(pri.nu), np.asarray(pri.S, float).reshape(dy, dy),
                               float(init.nu), np.asarray(init.S, float).reshape(dy, dy), n_chains=C,
                               prior_through_transition=model.prior_through_transition, segments=int(options.get("segments", 0)),
                               device=int(options.get("device", -1)))
        eng.set_data(y, layout="chain_time")
        eng.run(iterations=iters, free_energy=free_energy)
        mean, cov = eng.marginals(layout="chain_time")
        nu, V = eng.noise_posterior()
        fe = None
        if free_energy:
            fe = eng.free_energy()                      # per iteration, summed over the chains (every chain is its own graph)
            if single:
                fe = np.asarray(fe)
        if single:
            mean, cov, nu, V = mean[0], cov[0], nu[0], V[0]
        post = {"x": MvNormalMeanCovariance(mean, cov), "W": Wishart(nu, V)}
        if returnvars is not None:
            post = {k: v for k, v in post.items() if k in returnvars}
        return InferenceResult(post, None, fe, model, None)
    except Exception as err:
        if not catch_exception:
            raise
        return InferenceResult({}, None, None, model, err)
    finally:
        if eng is not None:
            eng.close()


def _smooth_on_executor(model, y, iters, free_energy, allow_missing, device):
    """y [chain][T][dy] through the node-array executor on the graph GraphPPL would build for the model (graph.lgssm_graph): mean [chain][T][d], cov, fe [chain][iters] | None"""
    from . import graph
    from .tree import TreeEngine
    C, T, dy = y.shape
    so, oo = model.state_offset, model.obs_offset
    kw = {}
    if so is not None:
        so = np.broadcast_to(np.asarray(so, float), (T, model.A.shape[-1]))
        kw["c_of_t"] = lambda t: so[t]
    if oo is not None:
        oo = np.broadcast_to(np.asarray(oo, float), (T, dy))
        kw["d_of_t"] = lambda t: oo[t]
    gb, xs, ys = graph.lgssm_graph(T, model.A, model.B, model.P, model.Q, model.prior_mean, model.prior_cov,
                                   prior_through_transition=model.prior_through_transition, **kw)[:3]
    with TreeEngine(gb, n_replicas=C, allow_missing=allow_missing, device=device) as te:
        te.set_data(ys, np.ascontiguousarray(y.reshape(C, T * dy)))
        te.run(1, bool(free_energy))
        post = te.marginals(xs)
        fe = np.repeat(te.free_energy_per_replica()[:, None], iters, axis=1) if free_energy else None
    return np.stack([post[v][0] for v in xs], axis=1), np.stack([post[v][1] for v in xs], axis=1), fe


def infer(*, model, data, iterations=None, free_energy=False, options=None, returnvars=None, predictvars=None,
          catch_exception=False, initialization=None, autoupdates=None, keephistory=None, historyvars=None):
    """Static (batch) inference on the device engine; with `autoupdates` (any truthy value: the state-space spec has
    exactly one feedback, `mean_cov(q(x_t))` -> prior of the next step) the streaming / filtering twin.

    data = {"y": array}: [T][dy] for one chain (as `data = (y = observations,)`), or
    [chain][T][dy] for a batch of independent chains sharing the model.
    Returns posteriors["x"] as MvNormalMeanCovariance with mean [T][d] / [chain][T][d].
    `predictvars = ("y",)` (reference: `predictvars = (y = KeepLast(),)`): result.predictions["y"] holds the message toward
    every y[t] — leave-one-out predictive for observed steps.  NaN rows of `y` are `missing` observations: a tail that no
    chain observed is a forecast horizon (posteriors and predictions there are forward predictions, the sweep stays
    time-parallel); `missing` values anywhere else select the masked schedule (any d, dy ≤ 64), where a missing y[t] sends no
    message and its prediction is the smoothed predictive."""
    if isinstance(model, UnivariateGaussianMixture):
        return _infer_mixture(model, data, iterations, free_energy, options, initialization, catch_exception)
    if isinstance(model, MultivariateGaussianMixture):
        return _infer_mv_mixture(model, data, iterations, free_energy, options, initialization, catch_exception)
    if isinstance(model, HierarchicalGaussianFilter):
        return _infer_hgf(model, data, iterations, free_energy, options, initialization, catch_exception)
    if isinstance(model, UnivariateDriftChain):
        return _infer_drift_chain(model, data, iterations, free_energy, options, catch_exception)
    if not isinstance(model, LinearGaussianSSM):
        raise TypeError("infer: no device schedule for this model type")
    if model.noise_precision_prior is not None:
        if autoupdates or predictvars:
            raise ValueError("unknown observation precision: no streaming twin and no predictions on the device")
        return _infer_lgssm_noise(model, data, iterations, free_energy, options, initialization, returnvars, catch_exception)
    if autoupdates:
        if iterations not in (None, 1):
            raise ValueError("filtering: the one-step graph is a tree, iterations must be 1")
        return _infer_lgssm_filtering(model, data, free_energy, options, initialization, catch_exception)
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
    horizon, allow_missing = 0, False
    nans = np.isnan(y)
    if nans.any():
        if not predictvars:   # data with `missing` values is predicted without being asked (src/inference/batch.jl:221-227)
            predictvars = ("y",)
        # `missing` observations (docs/src/manuals/inference/static.md:98-123).  A tail that no chain observed is a forecast
        # horizon and keeps the time-parallel schedule; anything else runs the per-chain masked schedule.
        missing = np.all(nans, axis=(0, 2))
        while horizon < T - 1 and missing[T - 1 - horizon]:
            horizon += 1
        if nans[:, :T - horizon].any():
            horizon, allow_missing = 0, True
        else:
            y, T = y[:, :T - horizon], T - horizon
    eng = None
    try:
        m0, V0, step_model = model.prior_mean, model.prior_cov, model.step_model
        if step_model is not None:
            if step_model.shape[0] != T + horizon:
                raise ValueError(f"time-varying model has {step_model.shape[0]} steps, the data {T + horizon}")
            M = model.A.shape[0]
            m0, V0 = np.broadcast_to(m0, (M,) + m0.shape[-1:]), np.broadcast_to(V0, (M,) + V0.shape[-2:])
        try:
            eng = LGSSMEngine(model.A, model.B, model.P, model.Q, m0, V0, T=T, n_chains=C,
                              prior_through_transition=model.prior_through_transition, horizon=horizon,
                              allow_missing=allow_missing, step_model=step_model,
                              state_offset=(model.state_offset if model.input_matrix is None else np.zeros(model.A.shape[-1])),
                              obs_offset=model.obs_offset,
                              segments=int(options.get("segments", 0)), device=int(options.get("device", -1)))
        except RxHipError as err:
            # a model beyond the conditioning envelope of the information-form chain engines (include/rxhip.h rxhip_set_conditioning_guard): the same graph on the
            # node-array executor, as rxhip_create does for a host that hands over the graph — plain smoothing only (the executor has no forecast / prediction entry)
            if err.status != _lib.ERR_UNSUPPORTED or "kappa" not in str(err) or horizon or predictvars or step_model is not None or model.input_matrix is not None:
                raise
            mean, cov, fe = _smooth_on_executor(model, y, iters, free_energy, allow_missing, int(options.get("device", -1)))
            if single:
                mean, cov, fe = mean[0], cov[0], (fe[0] if fe is not None else None)
            post = {"x": MvNormalMeanCovariance(mean, cov)}
            if returnvars is not None:
                post = {k: v for k, v in post.items() if k in returnvars}
            return InferenceResult(post, None, fe, model, None)
        if model.input_matrix is not None:   # control inputs as data: c[t] = B_u u[t] (+ the constant part) for every chain
            if "u" not in data:
                raise ValueError("this model has data inputs: data must provide `u`")
            u = np.asarray(data["u"], dtype=np.float64)
            u = u[None] if single else u
            c = u @ model.input_matrix.T
            if model.state_offset is not None:
                c = c + model.state_offset
            if c.shape[1] != T + horizon:
                raise ValueError(f"`u` has {c.shape[1]} steps, the data {T + horizon}")
            eng.set_chain_offsets(c, None, layout="chain_time")
        fe1 = None
        if single:   # one chain: the whole call in one round trip (rxhip_lgssm_infer)
            m1, c1, fe1 = eng.infer(y[0][:, None, :], iterations=iters, free_energy=free_energy)
            mean, cov = np.transpose(m1, (1, 0, 2)), np.transpose(c1, (1, 0, 2, 3))
        else:
            eng.set_data(y, layout="chain_time")
            eng.run(iterations=iters, free_energy=free_energy)
            mean, cov = eng.marginals(layout="chain_time")
        pred = None
        if predictvars:
            pm, pc = eng.predictions(layout="chain_time")
            pred = {"y": MvNormalMeanCovariance(pm[0], pc[0]) if single else MvNormalMeanCovariance(pm, pc)}
        if free_energy:
            # one value per iteration, per chain-graph: what `infer` returns for each chain
            fe = fe1 if fe1 is not None else eng.free_energy_per_chain()
            fe = np.repeat(fe[:, None], iters, axis=1)
        else:
            fe = None
        if single:
            mean, cov = mean[0], cov[0]
            fe = fe[0] if fe is not None else None
        post = {"x": MvNormalMeanCovariance(mean, cov)}
        if returnvars is not None:
            post = {k: v for k, v in post.items() if k in returnvars}
        return InferenceResult(post, pred, fe, model, None)
    except Exception as err:  # catch_exception semantics of batch.jl:440-446
        if not catch_exception:
            raise
        return InferenceResult({}, None, None, model, err)
    finally:
        if eng is not None:
            eng.close()
