/*
 * rxoracle.h — CPU restatement ("oracle") of the RxInfer / ReactiveMP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / the reported CPU baseline.  The shipped engine is librxhip (rxinfer.jl_amd/csrc).
 *
 * Provenance / pinning status
 * ---------------------------
 * The arithmetic of the reference hot path does not live in /root/reference: it lives in the
 * un-vendored Julia packages ReactiveMP.jl (~6.0.0), ExponentialFamily.jl (2.1.0),
 * BayesBase.jl (1.5.0) and FastCholesky.jl (1.3.0) (reference Project.toml:43-73).  There is no
 * Julia toolchain in this image, so the reference itself cannot be executed.  This file restates
 * the published algorithm of those rules and follows the reference's own call sites:
 *   - which node each `MvNormal(μ, Σ)` / `A * x` spelling selects  src/model/graphppl.jl:372-376,
 *     docs/src/manuals/model-specification.md:217-240
 *   - message product order (left-to-right over a variable's neighbours)
 *     src/model/plugins/reactivemp_inference.jl:365-374
 *   - iteration semantics (same data re-pushed each iteration)      src/inference/batch.jl:391-430
 *   - Bethe free energy: which terms, how summed, point-mass bookkeeping
 *     src/model/plugins/reactivemp_free_energy.jl:51-126, src/helpers.jl:21
 * It is pinned against (a) the RNG-free known answers of the reference tests
 * (test/models/models_tests.jl:255,286,308,335: FE 3.51551 / 2.26551, means 1.5 / 1.0) and
 * (b) the identities BP == Kalman/RTS smoother and Bethe FE == -log p(y) on trees
 * (docs/src/manuals/variational/bethe-free-energy.md:70), and (c) the RNG-dependent golden values of the
 * reference tests, on data regenerated bit-exactly by oracle/stable_rng.py (StableRNGs + Julia's randn restated):
 * mlgssm_test.jl:128 FE 6275.9015944677 (13 digits), ulgssm_tests.jl:48 FE 1854.297647, hgf_tests.jl:113 FE
 * 1.009879989585 (to 1e-5), gmm_multivariate_tests.jl:141 FE 3436.7 (3436.721; the label sampler's alias-table layout
 * is inferred — DESIGN.md §5); fixtures under tests/golden/.  Not reproduced: gmm_univariate_tests.jl:97.
 */
#ifndef RXORACLE_H
#define RXORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes */
#define RXO_OK 0
#define RXO_ERR_NOT_POSDEF 3 /* mirrors FastCholesky / PosDefException */
#define RXO_ERR_BADARG 1
#define RXO_ERR_NONFINITE_FE 4 /* mirrors src/score/diagnostics.jl:19-51 */

/* Operation counters, the oracle's equivalent of the reference's callback events
 * (src/callbacks/events.jl; counted in test/callbacks/trace_tests.jl:93-104). */
typedef struct {
    uint64_t rule_calls;   /* after_message_rule_call */
    uint64_t products;     /* after_product_of_two_messages */
    uint64_t marginals;    /* after_marginal_computation */
} rxo_counters;

/*
 * Linear Gaussian state-space model, one chain, one BP sweep in the reference's message
 * schedule.  Model (benchmarks notebook cell 4):
 *     x[1] ~ MvNormal(μ = m0, Σ = V0)                      (prior_through_transition = 0)
 *     x[t] ~ MvNormal(μ = A * x[t-1], Σ = P)    t = 2..T    (P: state noise)
 *     y[t] ~ MvNormal(μ = B * x[t],   Σ = Q)    t = 1..T    (Q: observation noise)
 * With prior_through_transition = 1 (test/models/statespace/mlgssm_test.jl:9-17):
 *     x0 ~ MvNormal(m0, V0);  x[1] ~ MvNormal(A * x0, P);  ...
 * All matrices row-major.  A: d×d, B: dy×d, P: d×d, Q: dy×dy, y: [T][dy].
 * Outputs: post_mean [T][d], post_cov [T][d][d] (posteriors of x[1..T]);
 * free_energy (nullable): Bethe free energy, full node/variable sum with CountingReal
 * bookkeeping; counters (nullable) counts the posterior-only sweep (6 rule calls per step).
 */
int rxo_lgssm_bp(int d, int dy, int T, const double* A, const double* B, const double* P,
                 const double* Q, const double* m0, const double* V0,
                 int prior_through_transition, const double* y, double* post_mean,
                 double* post_cov, double* free_energy, rxo_counters* counters);

/*
 * Batch driver used for the CPU baseline: n_chains independent chains sharing one model,
 * y laid out [T][chain][dy], outputs [T][chain][d] and [T][chain][d][d], fe[chain] (nullable).
 * nthreads > 1 uses OpenMP over chains (the reference itself is single-threaded).
 */
/* node-local joints q(out = x[t], μ = A x[t-1]) of the transition nodes between observed states (mean [T-1][2d], cov [T-1][2d][2d]) */
int rxo_lgssm_bp_joints(int d, int dy, int T, const double* A, const double* B, const double* P, const double* Q,
                        const double* m0, const double* V0, int ptt, const double* y, double* joint_mean, double* joint_cov);
int rxo_lgssm_bp_batch(int d, int dy, int T, int n_chains, const double* A, const double* B,
                       const double* P, const double* Q, const double* m0, const double* V0,
                       int prior_through_transition, const double* y, double* post_mean,
                       double* post_cov, double* fe, int nthreads, rxo_counters* counters);

/*
 * Streaming / filtering twin (src/inference/streaming.jl:349-407 with `@autoupdates`, benchmark notebook cells 4 and 7,
 * `linear_gaussian_ssm_filtering`): for every observation the one-step graph
 *     x_min_t ~ MvNormal(μ = m, Σ = V);  x_t ~ MvNormal(μ = A * x_min_t, Σ = P);  y_t ~ MvNormal(μ = B * x_t, Σ = Q)
 * is evaluated in the reference's rule order (`*`_A(:out), MvN_x(:out), MvN_y(:μ), `*`_B(:in), product, mean_cov) and
 * (m, V) <- mean_cov(q(x_t)).  prior_through_transition = 0: the first observation sees the prior on x_1 directly.
 * hist_mean [T][d], hist_cov [T][d][d]: q(x_t) after each observation; fe (nullable): mean over observations of the
 * per-observation Bethe free energy (= −log p(y_t | y_<t) on the one-step tree), the value the reference's
 * free_energy_history holds for a streaming run (src/score/actor.jl:98-104).
 */
int rxo_lgssm_filter(int d, int dy, int T, const double* A, const double* B, const double* P, const double* Q,
                     const double* m0, const double* V0, int prior_through_transition, const double* y,
                     double* hist_mean, double* hist_cov, double* fe, rxo_counters* counters);

/* Predictions of the data variables (`predictvars`, src/model/plugins/reactivemp_inference.jl:619-624): the message
 * MvN_y(:out) toward y[t] = N(B m, B V B' + Q) with (m, V) = product of the forward and backward messages into x[t]
 * (its own observation excluded).  H ≥ 0 further time steps carry no observation (`missing`): their x-posteriors
 * (post_mean / post_cov, [H][d] / [H][d][d], nullable) are the forward predictions.  pred_mean [T+H][dy], pred_cov
 * [T+H][dy][dy]. */
int rxo_lgssm_predict(int d, int dy, int T, int H, const double* A, const double* B, const double* P, const double* Q,
                      const double* m0, const double* V0, int prior_through_transition, const double* y,
                      double* pred_mean, double* pred_cov, double* post_mean, double* post_cov);

/* Noise-free drift chain (test/models/statespace/ulgssm_tests.jl:8-15) in the reference's message schedule:
 *     x_prior ~ Normal(μ = m0, v = v0);  x[t] ~ x[t-1] + c;  y[t] ~ Normal(μ = x[t], v = obs_var),  t = 1..T
 * (prior_through_transition = 0: the prior sits on x[1]).  post_mean / post_var [T]: q(x[t]); free_energy (nullable):
 * full Bethe sum with CountingReal bookkeeping — reproduces the reference's golden 1854.297647 (ulgssm_tests.jl:48)
 * on the regenerated data (tests/golden/ulgssm_stablerng123.npz). */
int rxo_drift_chain_bp(long long T, const double* y, double m0, double v0, double c, double obs_var,
                       int prior_through_transition, double* post_mean, double* post_var, double* free_energy,
                       rxo_counters* counters);

/* Independent cross-check used only to validate the oracle itself: textbook Kalman filter +
 * RTS smoother and -log p(y) via innovations.  Same argument layout as rxo_lgssm_bp. */
int rxo_lgssm_kalman_rts(int d, int dy, int T, const double* A, const double* B, const double* P,
                         const double* Q, const double* m0, const double* V0,
                         int prior_through_transition, const double* y, double* post_mean,
                         double* post_cov, double* neg_loglik);
/* the same with known inputs: x[t] ~ N(A x[t-1] + cx[t], P), y[t] ~ N(B x[t] + cy[t], Q); cx [T][d], cy [T][dy], either may be NULL */
int rxo_lgssm_kalman_rts_affine(int d, int dy, int T, int n_models, const double* A, const double* B, const double* P,
                                const double* Q, const double* m0, const double* V0, const int* step_model, int ptt,
                                const double* cx, const double* cy, const double* y, double* post_mean, double* post_cov,
                                double* neg_loglik);
/* the same with time-varying constants: n_models models, step_model[t] = model of time index t */
int rxo_lgssm_kalman_rts_tv(int d, int dy, int T, int n_models, const double* A, const double* B, const double* P,
                            const double* Q, const double* m0, const double* V0, const int* step_model, int ptt,
                            const double* y, double* post_mean, double* post_cov, double* neg_loglik);

/*
 * Univariate Gaussian mixture, mean-field VMP (test/models/mixtures/gmm_univariate_tests.jl:7-26,
 * generalised to K components: Beta(1,1)/Bernoulli -> Dirichlet/Categorical as in
 * gmm_multivariate_tests.jl:22-31):
 *     s ~ Dirichlet(alpha0);  m[k] ~ Normal(mean = mu0[k], variance = v0[k]);
 *     p[k] ~ Gamma(shape = a0[k], rate = b0[k]);
 *     z[i] ~ Categorical(s);  y[i] ~ NormalMixture(switch = z[i], m = m, p = p)
 * with q(z, s, m, p) = q(z)q(s)Πq(m[k])Πq(p[k]).  K = 1 is the iid Gaussian with unknown mean and
 * precision (test/models/models_tests.jl:114-128, `iid_gaussians_params`).
 * Rules restated from SURVEY.md Appendix A.4/A.5.  Schedule (ASSUMED — the reactive update order is
 * not documented in the reference tree): per iteration, q(z[i]) from the marginals of the
 * previous iteration; then q(s) and q(m[k]) from the new q(z) (q(m) with E[p] of the previous iteration); then
 * q(p[k]) from the new q(z) AND the new q(m[k]).  (With q(p) computed from the previous q(m) the reference
 * test's own initialisation — vague Gamma, E[p] = 1e12 — drives the iteration into a poor optimum, FE ≈ 571 on
 * the reference's data, whereas the reference reaches 284.76; the sequential order converges at once.)
 * init_*: the `@initialization` marginals.  hist: [iterations][5][K] = (mean m, var m, shape p, rate p,
 * alpha s) after each iteration; fe: [iterations] Bethe free energy; resp (nullable): [N][K] final q(z).
 */
int rxo_gmm_vmp(long long N, int K, const double* y, const double* mu0, const double* v0, const double* a0,
                const double* b0, const double* alpha0, const double* init_m_mean, const double* init_m_var,
                const double* init_p_shape, const double* init_p_rate, const double* init_s_alpha, int iterations,
                double* hist, double* fe, double* resp, rxo_counters* counters);

/* Split-phase form of one iteration (rxo_gmm_vmp is the loop over these): what several GPUs do — every rank
 * accumulates its shard of the points, the 3K+1 statistics (Σπ, Σπy, Σπy² per component, Σ_i H[q(z_i)]) are summed
 * over ranks, every rank applies the same update.  state: 5K doubles (mean m, var m, shape p, rate p, alpha s),
 * stats: 3K+1 doubles.  resp (nullable): [N][K] q(z) of this call. */
int rxo_gmm_accumulate(long long N, int K, const double* y, const double* state, double* stats, double* resp,
                       rxo_counters* counters);
int rxo_gmm_update(int K, const double* mu0, const double* v0, const double* a0, const double* b0, const double* alpha0,
                   const double* stats, double* state, double* fe, rxo_counters* counters);

/* LGSSM with an unknown observation-noise precision, W ~ Wishart(nu0, S0), y[t] ~ MvNormal(μ = B x[t], Λ = W), q(x, W) = q(x) q(W):
 * mean-field VMP, see rxoracle.c.  post_mean [T][d], post_cov [T][d][d] of the last iteration; w_hist [iterations][1 + dy·dy] (ν | V,
 * nullable); fe [iterations] (nullable).  Checker of rxhip_lgssm_noise_create. */
int rxo_lgssm_noise_vmp(int d, int dy, int T, const double* A, const double* B, const double* P, const double* m0, const double* V0,
                        int prior_through_transition, const double* y, double nu0, const double* S0, double init_nu, const double* init_V,
                        int iterations, double* post_mean, double* post_cov, double* w_hist, double* fe);

/*
 * Multivariate Gaussian mixture, mean-field VMP (test/models/mixtures/gmm_multivariate_tests.jl:6-32):
 *     m[k] ~ MvNormal(mean = mu0[k], cov = S0[k]);  w[k] ~ Wishart(nu0[k], V0[k]);  s ~ Dirichlet(alpha0);
 *     z[i] ~ Categorical(s);  y[i] ~ NormalMixture(switch = z[i], m = m, p = w)
 * y: [N][d].  init / hist layout per component: mean[d] | cov[d][d] | nu | V[d][d] | alpha  (SZ = 2 + d + 2d² doubles);
 * init: [K][SZ] (`@initialization`), hist: [iterations][K][SZ], fe: [iterations], resp (nullable): [N][K] final q(z).
 * Schedule as rxo_gmm_vmp.  d = 1 reproduces rxo_gmm_vmp (Wishart(ν, V) = Gamma(ν/2, 1/(2V))).
 */
int rxo_mvgmm_vmp(long long N, int K, int d, const double* y, const double* mu0, const double* S0, const double* nu0,
                  const double* V0, const double* alpha0, const double* init, int iterations, double* hist, double* fe,
                  double* resp);

/*
 * Hierarchical Gaussian filter, one series, online (test/models/statespace/hgf_tests.jl:9-70):
 *     zt_min ~ Normal(zm, zv); xt_min ~ Normal(xm, xv); zt ~ Normal(mean = zt_min, var = z_variance);
 *     xt ~ GCV(xt_min, zt, kappa, omega);  y ~ Normal(mean = xt, var = y_variance)
 *     q(xt, zt, xt_min) = q(xt, xt_min) q(zt);  GCVMetadata(GaussHermiteCubature(n_gh))
 * streamed over the observations with `@autoupdates` zt_min ← mean_var(q(zt)), xt_min ← mean_var(q(xt)) and
 * `vmp_iters` iterations per observation; initial q(zt) = N(z0m, z0v), q(xt) = N(x0m, x0v).
 * GCV rules restated from ReactiveMP (average energy verbatim from
 * test/inference/inference_tests.jl:594-606); the z-message is the ExponentialLinearQuadratic
 * exp(−½(κz + ψA·exp(−κz))) and its product with the Gaussian forward message is moment-matched with the
 * Gauss–Hermite rule (approximate_meancov).  Order inside an iteration (ASSUMED): joint q(xt, xt_min)
 * from the previous q(zt), then q(zt).  The joint q(zt, zt_min) entering the Bethe free energy is the Gaussian
 * marginalrule of the transition node fed with mean_var(z-message), i.e. the Gauss–Hermite moments of
 * pdf(z)·exp(z²/2) against N(0,1) — the reading that reproduces the golden free energy of hgf_tests.jl:113 to 1e-5.
 * Outputs: zm/zv/xm/xv [T] final marginals per observation (historyvars KeepLast), fe [vmp_iters] = mean over
 * observations of the per-iteration free energy (free_energy_history, src/score/actor.jl:98-104).
 */
int rxo_hgf_filter(long long T, const double* y, double kappa, double omega, double z_variance, double y_variance,
                   double z0m, double z0v, double x0m, double x0v, int vmp_iters, int n_gh, double* zm, double* zv,
                   double* xm, double* xv, double* fe, rxo_counters* counters);
/* Gauss–Hermite nodes / weights (physicists' convention, Σ w f(x) ≈ ∫ e^{−x²} f), n ≤ 64 */
int rxo_gauss_hermite(int n, double* x, double* w);

const char* rxo_version(void);

#ifdef __cplusplus
}
#endif
#endif
