/*
 * rxhip.h — C ABI of librxhip, the MI355X-native message-passing engine that replaces the
 * ReactiveMP hot path behind RxInfer's `infer(...)`.
 *
 * The reference has no FFI for this path (it is pure Julia): the boundary it
 * sits behind is a GraphPPL plugin.  Each entry point below names the reference interface it
 * stands in for (paths relative to the RxInfer.jl checkout).  A Julia `ccall` shim binding
 * exactly these symbols is in rxinfer.jl_amd/julia/RxHip.jl; see INTEGRATION.md.
 *
 * Conventions: plain C, no exceptions cross the ABI, every function returns an rxhip_status
 * (0 = ok).  All matrices are dense row-major IEEE fp64.  The caller owns every host buffer it
 * passes; the engine copies what it needs and owns its device memory until rxhip_destroy.
 * One host thread per handle; handles are independent.  The only process-wide state is mutex-protected and exists because
 * `infer(...)` builds an engine per call while driver calls cost more than a sweep of the reference's own benchmark sizes:
 * a pool of idle engine-owned HIP streams (creation / destruction costs milliseconds on this runtime), up to 4 parked device
 * blocks / 4 GB per process (hipFree ≈0.15 ms for megabytes, tens of ms for a gigabyte), and the read-only per-model tables of
 * the MFMA path, shared by reference count between engines of the same model (≈50 ms of host recursions + a 100 MB upload at
 * d = 64 otherwise), and up to 4 parked small ENGINES: rxhip_destroy of an engine of the d, dy <= 4 family (one model, no masks / per-step
 * constants / offsets / horizon / caller's stream, T * chains <= 65536, no error ever reported) keeps it, and the next rxhip_lgssm_create
 * whose descriptor is the same byte for byte (shapes, schedule options, device, every model matrix) returns it reset to the state of a new
 * engine — an `infer(...)` per call costs a sweep and two copies then, not a construction.  rxhip_release_cached_memory() returns
 * everything idle to the driver, and rxhip_set_caching(0) switches ALL of it off for the process: nothing is parked, nothing is shared between
 * handles — "handles are independent; no global state" (threading) to the letter, at the price of the driver calls above per engine.
 */
#ifndef RXHIP_H
#define RXHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------
 * Environment.  The library reads the process environment in three places only:
 *   RXHIP_TRACE=1        stage timings of engine creation on stderr (measurement aid; no effect on results or schedules)
 *   RXHIP_RCCL_LIB=path  the librccl.so to open for rxhip_comm_* / rxhip_allreduce_* (default: the one of the HIP runtime in the process)
 *   RXHIP_TEST_HOOKS=1   enables the schedule switches below.  Without it NONE of them is read: a host's environment cannot change the
 *                        schedule behind a result.  All of them select between schedules that compute the same posteriors and free energy
 *                        (the parity tests run both sides); they exist so that one schedule can check another and for A/B timings.
 *     RXHIP_ONE_PASS=0|1        d, dy <= 4 shared-model batches: force the four-phase / the one-pass table-driven schedule (default: by size)
 *     RXHIP_ONE_SEGMENT=1       d, dy <= 4 masked / per-step engines: one segment per chain
 *     RXHIP_BACKWARD_LANES=1    one-pass schedule: the backward sweep of the four-phase schedule instead of the table-driven one
 *     RXHIP_SMALL_SWEEP=0       few short chains: the five launches of the four-phase schedule instead of k_small_sweep (one launch)
 *     RXHIP_ELEM_FULL=1         per-chain models at d, dy <= 4: every recursion of the sweep in full to the end of every segment (no frozen tails, full records)
 *     RXHIP_NOISE_MOMENTS_PASS=1  unknown-noise engines: the residual second moments by a separate pass over the posteriors instead of inside the backward sweep
 *     RXHIP_ENGINE_POOL=0       rxhip_destroy frees small engines instead of parking them for the next rxhip_lgssm_create of the same descriptor
 *     RXHIP_NO_PACK=1           d <= 8 on the MFMA path: one chain per 16x16 tile instead of two
 *     RXHIP_DENSE_SPLIT=0|1     MFMA path, one model: without / with the model-data split (default: from four workgroups' worth of chains)
 *     RXHIP_HOST_TABLES=1       MFMA path: per-model tables by the host builder instead of the device builder (d >= 32)
 *     RXHIP_FE_RESID_VALU=1     MFMA path: the round-2 residual kernel (vector FMAs) instead of the MFMA one
 *     RXHIP_GSEQ=1, RXHIP_STEPM_GSEQ=1, RXHIP_FILTER_GSEQ=1, RXHIP_JOINTS_GSEQ=1
 *                               `missing` / per-step-constant engines at d > 4 (all runs / per-step engines only / filtering runs only /
 *                               node-local joints only): the sequential schedule (one workgroup per chain, the reference's message order)
 *                               instead of the time-parallel masked MFMA schedule
 *     RXHIP_MSEG_SCAN=sequential|log, RXHIP_MSEG_ONE_LEVEL=1, RXHIP_MSEG_GROUP=g
 *                               masked schedule: kind of the boundary recursion (default: the cheaper one by a cost model)
 *     RXHIP_MSEG_MAX_BYTES=n    masked schedule: cap on its record block (an engine that exceeds it stays on the sequential schedule)
 *     RXHIP_WAVE8=0             masked schedule, one segment per chain, d <= 8: the MFMA sweep kernels instead of the in-wave ones
 *     RXHIP_NO_FROZEN=1         MFMA path, time-invariant models at d >= 48: every step of the sweeps in full (no FROZEN / BFROZEN stretches, below)
 *     RXHIP_TREE_MODE=0|1|2     node-array executor: a launch per level / workgroup-resident levels / a lane (a wavefront above d = 8) per replica walks the
 *                               schedule, for both phases (default: by batch and graph shape, per phase)
 *     RXHIP_TREE_RB=n, RXHIP_TREE_WG=256|512
 *                               node-array executor, workgroup-resident levels: replicas per workgroup (multiple of 16) and threads per workgroup of the sweep phase
 *
 * Fixed-point exits (time-invariant models only: one set of constants per chain, no `missing`, no per-step constants).  The covariance
 * recursions of such a chain — forward Riccati, backward smoother — converge geometrically, and the sweeps stop RECOMPUTING a recursion's
 * matrices once they repeat; the means are always computed step by step.  "Repeat" is decided on the device, per chain or per segment:
 *   d, dy <= 4 (k_seg_elements, k_boundary_scan, k_forward_tinv): two weighted sums of the matrix's entries, each entry first scaled by the exact
 *     power of two 2^-floor((e_i + e_j)/2) (e_i: binary exponent of the diagonal entry i, so every scaled entry is O(1) whatever the units of
 *     the state's components), unchanged to 2 ulp on two consecutive steps in every lane of the wavefront;
 *   d >= 48 forward (kd_forward_info): in EVERY lane of the workgroup two weighted sums of the lane's 16 entries of the information matrix as
 *     equilibrated for the inverse (the same power-of-two scaling: entries of order one) unchanged to 1.4e-14, and two sums over all tiles to 2 ulp;
 *   d >= 48 backward (kd_backward_info): two sums over the tiles of V_s unchanged to 2 ulp, then V_s(t) against V_s(t+1) entry by entry,
 *     |dV_ij| <= 1.5e-14 sqrt(V_ii V_jj);
 *   the per-model boundary tables (host and device builders): two consecutive boundary matrices entry by entry, |d_ij| <= 5e-15 sqrt(a_ii a_jj).
 * Bound: a per-step change of an entry above ~3e-14 sqrt(M_ii M_jj) keeps the full recursion running (d <= 4 and the forward test: unless the
 * other entries of the same sum cancel it in both sums — two linear conditions on the direction of a converging iteration); with a contraction
 * rate rho of the recursion (its closed-loop spectral radius squared) what is frozen is within ~3e-14 / (1 - rho) sqrt(M_ii M_jj) of the fixed
 * point, entry by entry — 1e-8 of the entry's scale at rho = 1 - 3e-6, a mixing time of 3e5 steps.  Recursions that slow do not repeat that closely
 * inside a supported chain length and are simply computed in full.  tests/test_fixed_point_adversarial_gpu.py holds the sweeps to the contract
 * (1e-6 / 1e-8 against the CPU oracle, 1e-7 against the full recursion) on block models six decades apart with a slowly mixing small block and on
 * near-unit-root states; rxhip_set_fixed_point_exits(engine, 0) switches the exits of an engine's sweeps off (RXHIP_ELEM_FULL / RXHIP_NO_FROZEN: the same for
 * every engine of a process, test hooks).
 * ------------------------------------------------------------------------------------------ */
typedef struct rxhip_engine rxhip_engine;
typedef int32_t rxhip_status;

enum {
    RXHIP_OK = 0,
    RXHIP_ERR_BADARG = 1,           /* malformed descriptor / argument */
    RXHIP_ERR_UNSUPPORTED = 2,      /* node type / graph shape without a device schedule
                                       (reference: RuleMethodError, docs/src/manuals/inference/undefinedrules.md) */
    RXHIP_ERR_NOT_POSDEF = 3,       /* mirrors FastCholesky PosDefException (.github/workflows/CI.yml:72) */
    RXHIP_ERR_NONFINITE_FE = 4,     /* mirrors ObjectiveDiagnosticCheckNaNs/Infs, src/score/diagnostics.jl:19-51 */
    RXHIP_ERR_HIP = 5,              /* HIP runtime failure (see rxhip_last_error) */
    RXHIP_ERR_NO_DEVICE = 6,        /* no MI355X visible: the product path never falls back to the CPU */
    RXHIP_ERR_STATE = 7,            /* call order violated (e.g. run before set_data) */
    RXHIP_ERR_RCCL = 8
};

/* host/device layouts of per-(time, chain) arrays */
enum {
    RXHIP_LAYOUT_TIME_CHAIN = 0, /* [T][chain][k]  (device-native) */
    RXHIP_LAYOUT_CHAIN_TIME = 1  /* [chain][T][k]  (one Vector{Vector{Float64}} per chain, as `data = (y = ...,)`) */
};

/* variable ids of the LGSSM schedule */
enum {
    RXHIP_VAR_Y = 0, /* data variable y[t]   (src/inference/batch.jl:405-407 new_observation!) */
    RXHIP_VAR_X = 1, /* random variable x[t] (posteriors[:x], src/inference/batch.jl:325-332) */
    RXHIP_VAR_U = 2  /* data variable u[t] of `A * x[t-1] + B_u * u[t]`: engines built by rxhip_create from a graph with such
                        inputs take them through rxhip_set_data, (T·n_chains·du doubles, same layouts as y) */
};

/* ------------------------------------------------------------------------------------------
 * Structured descriptor of a batch of linear Gaussian state-space factor graphs — what the
 * lowering of the materialised graph produces (replaces the per-node objects built by
 * GraphPPL.postprocess_plugin(::ReactiveMPInferencePlugin, model),
 * src/model/plugins/reactivemp_inference.jl:272-326, for this model family):
 *     x[1] ~ MvNormal(μ = m0, Σ = V0)                   (prior_through_transition = 0;
 *                                                        benchmarks notebook cell 4)
 *     x0 ~ MvNormal(m0, V0); x[1] ~ MvNormal(A*x0, P)   (prior_through_transition = 1;
 *                                                        test/models/statespace/mlgssm_test.jl:9-17)
 *     x[t] ~ MvNormal(μ = A * x[t-1], Σ = P)     P: state noise        d×d
 *     y[t] ~ MvNormal(μ = B * x[t],   Σ = Q)     Q: observation noise  dy×dy, B: dy×d
 * `n_models` distinct constant sets; chain c uses model chain_model[c] (NULL: all chains use
 * model 0).  The schedule is compiled per (d, dy); supported: see rxhip_lgssm_supported().
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t d;
    int32_t dy;
    int64_t T;
    int64_t n_chains;
    int32_t n_models;
    int32_t prior_through_transition;
    const double* A;  /* [n_models][d][d]   */
    const double* B;  /* [n_models][dy][d]  */
    const double* P;  /* [n_models][d][d]   */
    const double* Q;  /* [n_models][dy][dy] */
    const double* m0; /* [n_models][d]      */
    const double* V0; /* [n_models][d][d]   */
    const int32_t* chain_model; /* [n_chains] or NULL */
    int32_t segments; /* time segments per chain for the parallel-in-time schedule; 0 = auto */
    int32_t device;   /* HIP device ordinal; -1 = current device */
    void* stream;     /* hipStream_t to run on; NULL = engine-owned stream */
    int64_t horizon;  /* further time indices T+1 … T+horizon WITHOUT an observation (`missing` at the end of the data,
                         test/inference/inference_tests.jl `predictvars`): their posteriors are forward predictions; the
                         posterior arrays then hold T + horizon rows.  Any d, dy ≤ 64; 0 = none */
    int32_t allow_missing; /* 1: an observation y[t] of a chain whose entries are NaN is `missing` ANYWHERE in the data
                         (docs/src/manuals/inference/static.md:98-123): no message from its observation branch, no evidence
                         term; its prediction (rxhip_get_predictions) is the plain predictive.  The covariances then differ
                         per chain and time index: the engine keeps per-chain records and computes the segment elements of
                         the time-parallel schedule in the lane (no per-model tables) at d, dy ≤ 4; larger states (any d, dy ≤ 64)
                         run the masked MFMA schedule, also parallel in time (csrc/dense_mseg_kernels.hpp: per-(chain, segment)
                         elements in information form, a log-depth boundary recursion over them, then the fully observed sweep
                         kernels with a mask).  An engine whose per-step records do not fit the device's free memory stays on
                         the sequential schedule — the reference's own message order, one workgroup per chain
                         (csrc/gseq_kernels.hpp) — which is also what the step-wise filter and the checker runs use.
                         rxhip_counters keeps reporting the all-observed schedule */
    const int32_t* step_model; /* NULL, or [T + horizon]: time-varying constants.  step_model[t] names the model (of n_models)
                         whose A, P make the transition INTO x[t] and whose B, Q observe y[t] (`A[t] * x[t-1]`,
                         `MvNormal(μ = …, Σ = P[t])` with per-step constants in the @model loop); the prior (m0, V0) is that of
                         model step_model[0].  All chains share the schedule (chain_model must be NULL); runs on the
                         table-free schedules of allow_missing (any d, dy ≤ 64) */
    const double* state_offset; /* NULL, or [T + horizon][d]: known inputs.  x[t] ~ MvNormal(μ = A * x[t-1] + c[t], Σ = P) — the
                         `+` node with a constant (or a `*` of a constant with data, e.g. B_u * u[t]) behind the transition's `*`
                         node; row 0 enters only through the prior's transition (prior_through_transition) */
    const double* obs_offset;   /* NULL, or [T + horizon][dy]: y[t] ~ MvNormal(μ = B * x[t] + d[t], Σ = Q).
                         Offsets are shared by all chains and need one model per time index (chain_model must be NULL).  They cost
                         no kernel anything: with μ[t] = A μ[t-1] + c[t] the chain x − μ is the homogeneous model observed through
                         y − B μ − d, so the data are shifted on the way in and the means on the way out (any d, dy) */
} rxhip_lgssm_desc;

/* New known inputs for an engine created WITH offsets (pass zero arrays at creation to reserve them): same shapes as
 * rxhip_lgssm_desc.state_offset / obs_offset, either may be NULL (= zeros).  A control loop that re-plans its inputs keeps the
 * engine, its tables and its observations; the next run / filter step uses the new inputs. */
rxhip_status rxhip_lgssm_set_offsets(rxhip_engine* e, const double* state_offset, const double* obs_offset);
/* Known inputs that are DATA of every chain — `x[t] ~ MvNormal(μ = A * x[t-1] + B_u * u[t], Σ = P)` with `u` a data variable: the
 * host passes c = B_u u per chain, (T + horizon)·n_chains·d doubles in `layout` ([t][chain][d] or [chain][t][d]); obs_offset
 * likewise with dy.  Either may be NULL = the offsets the engine was CREATED with (rxhip_lgssm_desc — NOT the latest
 * rxhip_lgssm_set_offsets values), replicated over the chains (a model with data inputs on the transitions and a constant observation
 * offset passes NULL for the latter); zeros if it was created without.  On an engine whose graph declares data inputs u[t]
 * (rxhip_create with a `*`(const, data) node behind a transition) a non-NULL state_offset counts as those inputs having been
 * supplied, exactly like rxhip_set_data(RXHIP_VAR_U).
 * The engine runs μ[t] = A μ[t-1] + c[t] per chain on the device and shifts data and means per chain from then on.  Same
 * precondition as rxhip_lgssm_set_offsets. */
rxhip_status rxhip_lgssm_set_chain_offsets(rxhip_engine* e, const double* state_offset, const double* obs_offset, int32_t layout);

/* replaces: create_model(...) + postprocess_plugin (src/inference/batch.jl:252,
 * src/model/plugins/reactivemp_inference.jl:272-326) for the LGSSM family */
rxhip_status rxhip_lgssm_create(const rxhip_lgssm_desc* desc, rxhip_engine** out);

/* The first COMPOSED graph: the state-space chain above with an UNKNOWN observation-noise precision,
 *     W ~ Wishart(nu0, S0);   y[t] ~ MvNormal(μ = B * x[t], Λ = W);   q(x[1..T], W) = q(x[1..T]) q(W)      (dy = 1: Gamma(ν/2, 1/(2 S0)))
 * — the chain of test/models/statespace/mlgssm_test.jl:9-14 with the observation nodes precision-parametrised and the node pair of
 * test/models/iid/mv_iid_precision_tests.jl:11-15 on W; mean-field factorisation as `@constraints q(x, W) = q(x)q(W)` asks for
 * (reactivemp_inference.jl:499-501).  replaces: the VMP iteration of src/inference/batch.jl:391-430 over that graph — per iteration ONE
 * belief-propagation sweep of every chain with its own Q⁻¹ = E[W] (MvNormalMeanPrecision(:μ) with q(Λ)), then the Wishart update of every
 * chain from the sweep's residual second moments (MvNormalMeanPrecision(:Λ), product with the prior), both on the device; the Bethe free
 * energy of the iteration's marginals per iteration (csrc/noise_kernels.hpp).  Every chain of the batch is its own graph with its own W.
 * desc: as for rxhip_lgssm_create with ONE model, d, dy ≤ 4, no horizon / allow_missing / step_model / offsets (RXHIP_ERR_UNSUPPORTED);
 * desc->Q is ignored (may be NULL).  init_nu, init_V: the `@initialization` marginal q(W) = Wishart(init_nu, init_V).
 * rxhip_run(iterations, want_fe), rxhip_get_marginals (q(x[t]) of the last iteration), rxhip_get_free_energy (per iteration, summed over
 * the chains), rxhip_get_free_energy_per_chain (last iteration) as for every engine; rxhip_lgssm_noise_get: q(W) of every chain after the
 * last iteration — nu [n_chains], V [n_chains][dy][dy] (either may be NULL). */
typedef struct {
    double nu0;        /* prior Wishart(nu0, S0) */
    const double* S0;  /* [dy][dy] */
    double init_nu;    /* initial marginal q(W) = Wishart(init_nu, init_V) */
    const double* init_V;
} rxhip_noise_prior;
rxhip_status rxhip_lgssm_noise_create(const rxhip_lgssm_desc* desc, const rxhip_noise_prior* prior, rxhip_engine** out);
rxhip_status rxhip_lgssm_noise_get(rxhip_engine* e, double* nu, double* V);
/* on != 0: every later rxhip_run CONTINUES from the q(W) the previous run ended with instead of the `@initialization` marginal (the first run
 * still starts there) — for drivers that take one VMP iteration per call, as the loop of src/inference/batch.jl:391-430 does: k calls of
 * rxhip_run(1) then equal one rxhip_run(k), bit for bit.  rxhip_get_free_energy covers the last call's iterations. */
rxhip_status rxhip_lgssm_noise_continue(rxhip_engine* e, int32_t on);

/* ------------------------------------------------------------------------------------------
 * Generic factor-graph descriptor — the struct-of-arrays dump of a materialised GraphPPL model, i.e. what
 * GraphPPL.postprocess_plugin(::ReactiveMPInferencePlugin, model) iterates over
 * (src/model/plugins/reactivemp_inference.jl:272-326: variable_nodes / factor_nodes, GraphPPL.fform,
 * GraphPPL.neighbors + getname(edge), is_random / is_data / is_constant, GraphPPL.value).
 * One descriptor holds ONE graph; `n_replicas` independent copies (different data, same constants) form the
 * chain batch.  The lowering pass (host C++) recognises the graph shapes that have a device schedule and
 * produces the structured descriptor above; anything else is RXHIP_ERR_UNSUPPORTED (the caller falls back to
 * the stock ReactiveMP plugin, cf. options.rulefallback, docs/src/manuals/inference/undefinedrules.md:101-108).
 * ------------------------------------------------------------------------------------------ */
enum {
    RXHIP_VARKIND_RANDOM = 0, /* randomvar  (reactivemp_inference.jl:337-341) */
    RXHIP_VARKIND_DATA = 1,   /* datavar    (:349-353) */
    RXHIP_VARKIND_CONST = 2   /* constvar   (:343-348) */
};
enum {
    /* ExponentialFamily.MvNormalMeanCovariance, interfaces (out, μ, Σ) — `MvNormal(μ = …, Σ = …)`, src/model/graphppl.jl:372-376 */
    RXHIP_NODE_MVNORMAL_MEAN_COV = 1,
    /* typeof(*), interfaces (out, A, in) — `A * x`, docs/src/manuals/model-specification.md:217-240 */
    RXHIP_NODE_MULTIPLY = 2,
    /* the mean-field families (aliases src/model/graphppl.jl:340-370 Normal, :399-423 Gamma; interface orders as in
     * ReactiveMP's @node declarations) */
    RXHIP_NODE_NORMAL_MEAN_VARIANCE = 3,  /* (out, μ, v)  `Normal(mean = …, var = …)` */
    RXHIP_NODE_NORMAL_MEAN_PRECISION = 4, /* (out, μ, τ)  `Normal(mean = …, precision = …)`: random τ -> the iid Gaussian×Gamma family; constant τ
                                             -> a Gaussian chain in precision form (test/inference/prediction_tests.jl:197-213), lowered as 1/τ */
    RXHIP_NODE_GAMMA_SHAPE_RATE = 5,      /* (out, α, β)  `Gamma(shape = …, rate = …)` */
    RXHIP_NODE_DIRICHLET = 6,             /* (out, a) */
    RXHIP_NODE_BETA = 7,                  /* (out, a, b) */
    RXHIP_NODE_CATEGORICAL = 8,           /* (out, p) */
    RXHIP_NODE_BERNOULLI = 9,             /* (out, p) */
    RXHIP_NODE_NORMAL_MIXTURE = 10,       /* (out, switch, m[1..K], p[1..K])  test/models/mixtures/gmm_univariate_tests.jl:16-19 */
    RXHIP_NODE_GCV = 11,                  /* (y, x, z, κ, ω)  test/models/statespace/hgf_tests.jl:28 */
    RXHIP_NODE_WISHART = 12,              /* (out, ν, S)  `Wishart(ν, S)`, test/models/mixtures/gmm_multivariate_tests.jl:23 */
    RXHIP_NODE_ADD = 13,                  /* typeof(+), interfaces (out, in1, in2) — `x_prev + c`, test/models/statespace/ulgssm_tests.jl:12 */
    RXHIP_NODE_MVNORMAL_MEAN_PRECISION = 14, /* (out, μ, Λ)  `MvNormal(μ = …, Λ = …)`, test/models/iid/mv_iid_precision_tests.jl:11-15 */
    RXHIP_NODE_GAMMA_SHAPE_SCALE = 15     /* (out, α, θ)  `Gamma(shape = …, scale = …)` and Distributions' `Gamma(α, θ)` (src/model/graphppl.jl:399-423,
                                             test/models/models_tests.jl:121-127): lowered as the rate form with β = 1/θ */
};
enum { /* family of an `@initialization` marginal (InitMarExtraKey, src/model/plugins/initialization_plugin.jl:201-202) */
    RXHIP_INIT_NONE = 0,
    RXHIP_INIT_NORMAL = 1,   /* (mean, variance) */
    RXHIP_INIT_GAMMA = 2,    /* (shape, rate) */
    RXHIP_INIT_DIRICHLET = 3, /* alpha[rows]; Beta(a, b) = (a, b) */
    RXHIP_INIT_MVNORMAL = 4,  /* mean[d], cov[d][d] */
    RXHIP_INIT_WISHART = 5    /* nu, S[d][d] */
};
typedef struct {
    int64_t n_variables;
    const int32_t* var_kind;     /* [n_variables] RXHIP_VARKIND_* */
    const int32_t* var_rows;     /* [n_variables] rows of the value (vector length; matrix rows) */
    const int32_t* var_cols;     /* [n_variables] 1 for vectors, columns for matrix-valued constants */
    const int64_t* var_const;    /* [n_variables] offset of a constant's value (row-major) in const_pool, −1 otherwise */
    int64_t n_factors;
    const int32_t* factor_type;  /* [n_factors] RXHIP_NODE_* */
    const int64_t* factor_iface; /* [n_factors][3] variable id per interface, in the node's interface order */
    const double* const_pool;
    int64_t n_const;
    int64_t n_replicas;
    /* ---- optional extensions (all-zero = the 3-interface Gaussian graphs above) ---- */
    const int64_t* factor_iface_ptr; /* NULL: factor_iface is [n_factors][3]; else CSR offsets [n_factors+1] into factor_iface */
    const int32_t* var_init_family;  /* NULL or [n_variables] RXHIP_INIT_*: the `@initialization` marginal of a random variable */
    const int64_t* var_init;         /* [n_variables] offset of its parameters in const_pool (−1: none) */
    int32_t gh_points;               /* GCVMetadata(GaussHermiteCubature(n)) of the GCV nodes; 0 = 31 */
    int64_t n_observations;          /* streaming (one-step) graphs: observations that will be pushed per replica */
    int32_t allow_missing;           /* some data variable holds `missing` (the data is known when the model is created,
                                        src/inference/batch.jl:252): state-space graphs — as rxhip_lgssm_desc.allow_missing (the masked schedule);
                                        graphs of the node-array executor — the data leaves stay in precision form, where NaN is the zero message */
    /* The factorisation of q around every node — what the stock plugin hands to `factornode(fform, interfaces, factorization)` as
     * GraphPPL.VariationalConstraintsFactorizationIndicesKey (src/model/plugins/reactivemp_inference.jl:499-506): NULL, or one cluster id per entry
     * of factor_iface (same indexing, [n_factors][3] or CSR): interfaces of ONE node with equal ids share a factor of q — the reference's
     * ((1, 2), (3,)) on (out, μ, Σ) is 0, 0, 1.  Ids are local to their node; ids of clamped (data / constant) interfaces are ignored.
     * Every schedule implements ONE factorisation per node type:
     *   Gaussian nodes — q(out, μ) joint, a random third interface (precision) in a factor of its own;  `*`, `+` — all random interfaces joint;
     *   GCV — q(y, x) q(z);  NormalMixture, Categorical, Bernoulli and the priors — mean-field (every random interface its own factor);
     * the node-array executor ALSO runs Gaussian nodes under q(out) q(μ) (mean-field between the two Gaussian interfaces: `MeanField()` on a chain; any dimension ≤ 64).
     * A table that asks a node for anything else is RXHIP_ERR_UNSUPPORTED with the node named (→ stock plugin) from every lowering pass, rxhip_create,
     * rxhip_tree_create and rxhip_tree_plan — never silently the other variational family's posterior.  NULL = the factorisation above (joint Gaussian
     * interfaces), which is what GraphPPL's default constraints (BetheFactorization) produce for the BP families. */
    const int32_t* factor_cluster;
} rxhip_graph_desc;

/* result of the lowering pass for the LGSSM family; matrices are written into caller buffers of the sizes below
 * (n_models = 1 unless the constants differ from step to step) */
typedef struct {
    int32_t d, dy;
    int64_t T;
    int32_t prior_through_transition;
    double* A;  /* [n_models][d][d]   */
    double* B;  /* [n_models][dy][d]  */
    double* P;  /* [n_models][d][d]   */
    double* Q;  /* [n_models][dy][dy] */
    double* m0; /* [d]      */
    double* V0; /* [d][d]   */
    int64_t* state_var; /* [T] variable id of x[t] in time order (nullable) */
    int64_t* data_var;  /* [T] variable id of y[t] in time order (nullable) */
    int32_t deterministic; /* 1: noise-free drift chain `x[t] ~ x[t-1] + c` (P is zero, A the identity) */
    double* c;             /* [d] drift (nullable); zero unless deterministic */
    int32_t n_models;      /* distinct (A, P, B, Q) along the chain: per-step constants `A[t] * x[t-1]`, `Σ = P[t]`, … */
    int32_t* step_model;   /* [T] model of time index t (nullable; all zero when n_models = 1) */
    int32_t has_offsets;   /* 1: some mean carries a `+` with a constant (known inputs) */
    double* state_offset;  /* [T][d]  c[t] of `A * x[t-1] + c[t]` (nullable; zeros where a step has none) */
    double* obs_offset;    /* [T][dy] d[t] of `B * x[t] + d[t]` (nullable) */
    int32_t du;            /* > 0: the transitions carry DATA inputs `+ B_u * u[t]` of this dimension */
    double* input_matrix;  /* [d][du] B_u (nullable) */
    int64_t* input_var;    /* [T] variable id of u[t], −1 where a transition has none (nullable) */
} rxhip_lgssm_lowered;

/* Host-only (no device needed): recognise a linear Gaussian state-space chain in `g` — MvNormalMeanCovariance or (scalar
 * chains) NormalMeanVariance nodes, each mean either `A * x` through a `*` node or the state itself (identity map), in any
 * mixture of the spellings of mlgssm_test.jl:9-66 — or the noise-free drift chain of ulgssm_tests.jl:8-15.  First call with all
 * pointer members of `out` NULL to learn d, dy, T; then with buffers to receive the constants.
 * RXHIP_ERR_UNSUPPORTED if the graph is not such a chain (message via rxhip_lowering_error()). */
rxhip_status rxhip_graph_lower_lgssm(const rxhip_graph_desc* g, rxhip_lgssm_lowered* out);
const char* rxhip_lowering_error(void); /* thread-local text of the last lowering failure */
/* A constant precision / covariance parameter must be symmetric within round-off: max|W − W′| ≤ 1e-8·max|W| (a precision computed as inv(Σ) on the host
 * carries ≈ eps·cond(Σ)·max|W|), else RXHIP_ERR_BADARG; below the bound the symmetric part ½(W + W′) is what the engines run on.  This returns the largest
 * max|W − W′| / max|W| the last lowering call of this thread (rxhip_graph_lower_*, rxhip_create, rxhip_tree_create) accepted that way — 0 when every
 * parameter was exactly symmetric — so that a caller who wants a tighter bound can enforce it. */
double rxhip_lowering_asymmetry(void);

/* Mean-field mixture (a9/a10): recognises  s ~ Dirichlet|Beta(const); m[k] ~ Normal(mean, var const); p[k] ~ Gamma(shape,
 * rate const); z[i] ~ Categorical|Bernoulli(s); y[i] (data) ~ NormalMixture(switch = z[i], m, p)  in any node order, and
 * the K = 1 form  y[i] ~ Normal(mean = m, precision = p)  (test/models/models_tests.jl:114-128).  Two-call protocol as
 * above (N, K first; then the [K] arrays and data_var [N]).  init_*: the `@initialization` marginals of m[k], p[k]
 * (required) and s (default Dirichlet(1)). */
typedef struct {
    int64_t N;
    int32_t K;
    double *mu0, *v0, *a0, *b0, *alpha0;                                             /* [K] priors  */
    double *init_m_mean, *init_m_var, *init_p_shape, *init_p_rate, *init_s_alpha;    /* [K] q init  */
    int64_t* data_var;                                                               /* [N] variable id of y[i] (nullable) */
} rxhip_gmm_lowered;
rxhip_status rxhip_graph_lower_gmm(const rxhip_graph_desc* g, rxhip_gmm_lowered* out);

/* Multivariate mixture (test/models/mixtures/gmm_multivariate_tests.jl:6-32): m[k] ~ MvNormal(mean, cov const);
 * w[k] ~ Wishart(ν, S const); s ~ Dirichlet; z[i] ~ Categorical(s); y[i] (data, d-vector) ~ NormalMixture(z[i], m, w).
 * Two-call protocol (N, K, d first).  Arrays as in rxhip_mvgmm_desc. */
/* Also the K = 1 form without a switch (test/models/iid/mv_iid_precision_tests.jl:11-15):  m ~ MvNormal(μ, Λ | Σ const);
 * P ~ Wishart(ν, S const); y[i] (data) ~ MvNormal(μ = m, Λ = P)  with q(m, P) = q(m)q(P) — a precision-parametrised prior is
 * returned as its covariance. */
typedef struct {
    int64_t N;
    int32_t K, d;
    double *mu0, *S0, *nu0, *V0, *alpha0;
    double *init_m_mean, *init_m_cov, *init_w_nu, *init_w_V, *init_s_alpha;
    int64_t* data_var; /* [N] (nullable) */
} rxhip_mvgmm_lowered;
rxhip_status rxhip_graph_lower_mvgmm(const rxhip_graph_desc* g, rxhip_mvgmm_lowered* out);

/* The state-space chain with an unknown observation-noise precision (rxhip_lgssm_noise_create above):
 *     W ~ Wishart(ν, S)   [dy = 1 also: τ ~ Gamma(shape, rate | scale), lowered as Wishart₁(2·shape, 1/(2·rate))]
 *     x-chain as for rxhip_graph_lower_lgssm;   y[t] ~ MvNormal(μ = B * x[t], Λ = W)   (`Normal(mean = …, precision = τ)`)
 * with the `@initialization` marginal q(W) (Wishart, or Gamma for dy = 1).  `chain` is filled as rxhip_graph_lower_lgssm fills it, except
 * Q (not written: the noise is the unknown).  One model, no offsets / inputs / missing observations, d, dy ≤ 4 — else RXHIP_ERR_UNSUPPORTED. */
typedef struct {
    rxhip_lgssm_lowered chain;
    int64_t precision_var; /* variable id of W */
    double nu0, init_nu;
    double* S0;     /* [dy][dy] (nullable) */
    double* init_V; /* [dy][dy] (nullable) */
} rxhip_lgssm_noise_lowered;
rxhip_status rxhip_graph_lower_lgssm_noise(const rxhip_graph_desc* g, rxhip_lgssm_noise_lowered* out);

/* Hierarchical Gaussian filter one-step graph (a11; test/models/statespace/hgf_tests.jl:9-31):
 *     zt_min ~ Normal(data, data); xt_min ~ Normal(data, data); zt ~ Normal(mean = zt_min, var = const);
 *     xt ~ GCV(xt_min, zt, κ const, ω const); y (data) ~ Normal(mean = xt, var = const)
 * with `@initialization` q(zt), q(xt) and the cubature order of the GCV meta. */
typedef struct {
    double kappa, omega, z_variance, y_variance, z0_mean, z0_var, x0_mean, x0_var;
    int32_t n_gh;
    int64_t zt_var, xt_var, y_var; /* variable ids of zt, xt, y */
} rxhip_hgf_lowered;
rxhip_status rxhip_graph_lower_hgf(const rxhip_graph_desc* g, rxhip_hgf_lowered* out);

/* replaces: create_model + postprocess_plugin for ANY supported graph (src/inference/batch.jl:252): lowers `g`,
 * then builds the engine exactly as rxhip_lgssm_create / rxhip_gmm_create / rxhip_hgf_create would (family chosen by
 * the node types present; HGF: T = n_observations, series = n_replicas).  segments/device/stream as in rxhip_lgssm_desc. */
rxhip_status rxhip_create(const rxhip_graph_desc* g, int32_t segments, int32_t device, void* stream, rxhip_engine** out);

/* ------------------------------------------------------------------------------------------
 * The level-scheduled node-array executor: ANY acyclic Gaussian factor graph (SURVEY §7's design stance: "one kernel over all nodes of the same
 * (factor type, interface)").  Replaces what `factornode(fform, interfaces, factorization)` + `activate!` build for an arbitrary model
 * (src/model/plugins/reactivemp_inference.jl:490-540), the message products (:432-447), the marginals (:440-447) and the Bethe sum
 * (src/model/plugins/reactivemp_free_energy.jl:51-126) for graphs of
 *     MvNormalMeanCovariance / NormalMeanVariance / MvNormalMeanPrecision / NormalMeanPrecision   (out, μ: random, data or constant;
 *         third interface: a constant, or — precision nodes — a Wishart / Gamma variable under q(out, μ) q(W));
 *     typeof(*) with a constant matrix;  typeof(+) (random + random, random + data / constant);
 *     Wishart / GammaShapeRate / GammaShapeScale priors with constant parameters;
 *     NormalMixture (out, switch, m[1..K], p[1..K]) under mean field with  switch ~ Categorical(s),  s ~ Dirichlet(a) | a constant — or, K = 2, switch ~ Bernoulli(s),
 *         s ~ Beta(a, b)  (round 6; dimensions ≤ 8):
 *         `out` data or a Gaussian variable, the means Gaussian variables of the forest, the precisions Wishart / Gamma variables or constants.  The node acts on
 *         (out, m[k], p[k]) as K Gaussian precision nodes weighted by q(switch = k); q(switch) is formed from the marginals of the previous iteration before
 *         the sweep, q(s) behind it (the schedule of the mixture engines; `@initialization` marginals on m, p, s — and `out` — as in the reference)
 *     GCV (y, x, z, κ, ω) under q(y, x) q(z), κ and ω constants, everything scalar (round 6; test/models/statespace/hgf_tests.jl:28-35): toward (y, x) a Gaussian node
 *         whose precision exp(−(κ z + ω)) is taken under q(z) of the previous iteration (`@initialization` on z required); toward z the ExponentialLinearQuadratic
 *         message — q(z) is its product with all other messages into z, moment-matched by Gauss–Hermite cubature (rxhip_graph_desc.gh_points, default 31) against
 *         that product, and z's neighbours (structured Gaussian nodes only) see the message through its Gaussian moments, as the reference does;
 *     scalar Gaussian nodes whose variance / precision is a DATA variable (`x ~ Normal(mean = m_prev, var = v_prev)` of a model driven by @autoupdates)
 * whose Gaussian variables form a forest (no cycles; precision variables may touch any number of nodes — the mean-field factorisation cuts those
 * loops; no Gaussian variable at all is a forest too: `P ~ Wishart; y[i] ~ MvNormal(μ = m, Λ = P)` with a known mean,
 * test/models/iid/mv_iid_precision_known_mean_tests.jl), every dimension ≤ 64.  Factorisation (rxhip_graph_desc.factor_cluster): a Gaussian node under
 * q(out, μ) — structured, the default — or under q(out) q(μ) (`constraints = MeanField()`): its rules then read MARGINALS,
 * MvNormalMeanCovariance(:out)(q_μ, q_Σ) = N(mean(q_μ), Σ), the node cuts the graph (cycles through it are fine), the marginals of its two variables are state
 * that starts from their `@initialization` marginals (RXHIP_INIT_NORMAL / MVNORMAL; an anonymous `A * x` output starts as the image of x's; none:
 * RXHIP_ERR_BADARG), every rule of an iteration reads the marginals of the previous one, and the free energy books the average energy with both marginals.
 * The host compiles the graph into ops sorted by dependency level (csrc/tree_compiler.hpp; the engine: csrc/tree_engine.hip); kernels evaluate (op, replica) items.
 * Dimensions ≤ 8: a LANE per item, matrices in registers (csrc/tree_kernels.hpp), replica-fastest storage; schedules — a launch per level, workgroup-resident
 * levels, a lane per replica over the whole schedule, and (dimensions ≤ 4: the default) STRANDS: the sweep cut into paths of dependent ops
 * that a lane walks with the message in registers, a message going to HBM only when somebody outside its strand reads it.  Marginals of `A * x` outputs are
 * images of x's marginal: the Bethe terms form them on the fly, they are stored when a caller asks.
 * Dimensions 9 … 32 (and 5 … 8 up to 1 024 replicas): a WAVEFRONT per (op, replica), matrices in registers in the accumulator layout of v_mfma_f64_16x16x4_f64
 * (csrc/tree_tile_kernels.hpp: products straight from registers, the inverse a symmetric sweep), a replica's slots contiguous in HBM; 33 … 64: a workgroup of
 * four wavefronts per (op, replica), matrices staged in LDS (csrc/tree_wave_kernels.hpp: products on the matrix cores, the inverse a 4-pivot block sweep with
 * the matrix in accumulator registers); a launch per level, or an item per replica over the whole schedule (rxhip_tree_info.kernels / .mode).
 * Data variables, derived clamped values (`a + b` of two data variables), unobserved leaves (predictions) are part of the family; so is `missing` anywhere in
 * the data when the engine is created with rxhip_graph_desc.allow_missing (a NaN observation sends no message, its node's Bethe terms cancel; not under a
 * random precision: RXHIP_ERR_UNSUPPORTED) — without it rxhip_tree_set_data refuses NaN / Inf with RXHIP_ERR_BADARG.
 * rxhip_create falls through to this executor for every graph the pattern matcher rejects; rxhip_tree_create asks for it directly (the tests
 * compare it with the specialised engines on the graphs both can run).
 * Message forms: a rule keeps the form its inbound message has wherever the algebra allows — the additive rule and the backward rule of `+` (two random
 * inputs) on a weighted-mean / precision message are Λ' = Λ (Λ + W)⁻¹ W, ξ' = W (Λ + W)⁻¹ (ξ + ξ2) − ξ2 (W: the noise precision, or the other input's,
 * ξ2 its weighted mean, 0 for noise): the messages of the reference's rules wherever those exist, and defined for the rank-deficient backward message of an
 * observation map with fewer rows than columns (where `mean_cov` of the reference throws).
 * Iterations (VMP): per iteration one sum-product sweep with E[W] of the current q(W), all marginals, then every q(W) update, then the free energy —
 * the order of rxhip_lgssm_noise_create above.  A run starts from the `@initialization` marginals (default: the priors).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int64_t n_ops, n_levels, n_messages;  /* ops of one iteration, dependency levels, stored messages */
    int64_t doubles_per_replica;          /* device state per replica */
    int64_t bytes_per_sweep;              /* message traffic of one iteration per replica in the engine's schedule: 8·(d + d(d+1)/2) per message a rule reads from or writes to HBM */
    int32_t dmax;                         /* kernel instance: 1, 2, 4 or 8 (a lane per item); above 8 the graph's largest dimension (a wavefront or workgroup per item: `kernels`) */
    int32_t mode;                         /* schedule of the sweep phase — 0: one launch per level; 1: one launch per phase, workgroup-resident levels (dmax ≤ 8); 2: a lane (dmax ≤ 8)
                                             or a wavefront per replica walks the schedule; 3 (dmax ≤ 4: the default): strands — a lane per (strand, replica) walks a path of
                                             dependent ops with the message in registers, one launch per strand level; bytes_per_sweep then counts what THAT schedule moves.
                                             (The Bethe / q(W) phase keeps 1 or 2.) */
    int32_t replicas_per_workgroup;       /* mode 1 */
    int32_t n_precision_vars;
    double last_iteration_ms;             /* device time of the last rxhip_run ÷ its iterations (HIP events around the launches) */
    int64_t io_bytes_per_sweep;           /* the floor of ANY schedule, per replica: the data in, the posteriors (mean, packed covariance, log-determinant) of the named variables out */
    int64_t n_strands, n_strand_levels;   /* mode 3: the sweep cut into strands of dependent ops (a lane walks a strand, messages handed over in registers), their dependency levels */
    int32_t longest_strand;               /* ops of the longest strand */
    int32_t kernels;                      /* which kernels run this engine — 0: a lane per (op, replica), matrices in the lane's registers (dmax ≤ 8; from 5 on only above
                                             1 024 replicas); 1: a wavefront per (op, replica), matrices in registers in the matrix cores' accumulator layout (5 … 32);
                                             2: a workgroup of four wavefronts per (op, replica), matrices staged in LDS (33 … 64); −1 from rxhip_tree_plan (chosen with the batch) */
    int64_t strand_bytes_per_sweep;       /* bytes_per_sweep of the strand schedule, whichever mode runs */
    int64_t fe_bytes_per_sweep;           /* what the second phase (Bethe terms, q(W) updates, the sum) reads and writes per replica: messages, marginals, data values, terms, statistics */
} rxhip_tree_info;
rxhip_status rxhip_tree_create(const rxhip_graph_desc* g, int32_t device, void* stream, rxhip_engine** out);
/* The graph compiler alone — host only, no device: would rxhip_tree_create take this graph, and with what schedule?  Fills the static fields of `out`
 * (mode = −1: chosen with the batch) and the reference-equivalent counts of ONE replica and iteration (what after_message_rule_call / the products / the
 * marginals would count: src/inference/batch.jl:495-496); failures as rxhip_tree_create (text: rxhip_lowering_error()).  The plugin can ask before it builds
 * an engine; the CPU tests hold these counts to the oracle's. */
rxhip_status rxhip_tree_plan(const rxhip_graph_desc* g, rxhip_tree_info* out, uint64_t* rule_calls, uint64_t* products, uint64_t* marginals);
/* data of the listed data variables, host [replica][rows of vars[0] | rows of vars[1] | …] (src/inference/batch.jl:405-407 new_observation!) */
rxhip_status rxhip_tree_set_data(rxhip_engine* e, const int64_t* vars, int64_t n_vars, const double* host);
/* posteriors of the listed random (Gaussian) variables: mean [var][replica][d], cov [var][replica][d][d], concatenated in list order.  A random
 * variable of the model that the compiler found clamped (the output of `a + b` of two data variables) and a data variable are reported as point
 * masses: mean = the value, zero covariance. */
rxhip_status rxhip_tree_get_marginals(rxhip_engine* e, const int64_t* vars, int64_t n_vars, double* mean, double* cov);
/* q(W) of a precision variable: nu [replica], V [replica][d][d] (a Gamma(a, b) variable is reported as Wishart_1(2a, 1/(2b))) */
rxhip_status rxhip_tree_get_precision(rxhip_engine* e, int64_t var, double* nu, double* V);
/* the discrete side of a NormalMixture layer: for the switch z[i] of a mixture node its responsibilities q(z[i] = k), for the probability vector s of
 * `z ~ Categorical(s)` the concentrations of q(s) = Dirichlet(α) — out [replica][K]; *n_components (nullable) receives K.  Any other variable: RXHIP_ERR_BADARG. */
rxhip_status rxhip_tree_get_discrete(rxhip_engine* e, int64_t var, double* out, int32_t* n_components);
rxhip_status rxhip_tree_get_info(rxhip_engine* e, rxhip_tree_info* out);
/* on != 0: every later rxhip_run CONTINUES from the q(W) the previous run ended with instead of the `@initialization` marginals (the first run still
 * starts there) — for drivers that take one VMP iteration per call, as the loop of src/inference/batch.jl:391-430 does (the plugin's `fire!`):
 * k calls of rxhip_run(1) then equal one rxhip_run(k), bit for bit.  The twin of rxhip_lgssm_noise_continue. */
rxhip_status rxhip_tree_continue(rxhip_engine* e, int32_t on);
/* rxhip_run, rxhip_get_free_energy (sum over the replicas, per iteration), rxhip_get_free_energy_per_chain (per replica, last iteration),
 * rxhip_counters, rxhip_sync, rxhip_get_stream, rxhip_last_error, rxhip_destroy apply as to every engine. */

/* One rule, evaluated on the device for a batch of inputs — the fine-grained A/B hook of SURVEY §8(b): what
 * `@rule NodeType(:iface, Marginalisation) (m_… , q_…)` returns for the given inbound message(s) and constants
 * (test/inference/inference_tests.jl:2049-2066 redirects a node to custom rule code the same way).  Runs the executor's own op on a one-node
 * schedule: the number a test compares with the reference's rule output. */
typedef struct {
    int32_t node_type;      /* RXHIP_NODE_MVNORMAL_MEAN_COV | NORMAL_MEAN_VARIANCE | MVNORMAL_MEAN_PRECISION | NORMAL_MEAN_PRECISION | MULTIPLY | ADD */
    int32_t iface;          /* the interface the message leaves through, index in the node's interface order (0 = out) */
    int32_t d_out, d_in;    /* dimension of `out`; of the input of MULTIPLY (A is d_out × d_in), else = d_out */
    int64_t n;              /* independent evaluations */
    const double* constant; /* Gaussian nodes: Σ or Λ [d][d]; MULTIPLY: A [d_out][d_in]; ADD: NULL */
    int32_t in_form;        /* inbound message(s): 0 = (mean, covariance), 1 = (weighted mean, precision) */
    const double* in_a;     /* [n][d]: Gaussian nodes — the message on the other interface; MULTIPLY — on `in` (iface 0) or `out` (iface 2);
                               ADD — on in1 (iface 0) or on out (iface 1, 2) */
    const double* in_B;     /* [n][d][d] */
    const double* in2_a;    /* ADD: the message on in2 (iface 0) / on the other input (iface 1, 2); else NULL */
    const double* in2_B;
    int32_t out_form;       /* form of the result: 0 = (mean, covariance), 1 = (weighted mean, precision) */
    double* out_a;          /* [n][d'] */
    double* out_B;          /* [n][d'][d'] */
} rxhip_rule_call;
rxhip_status rxhip_rule_eval(const rxhip_rule_call* call, int32_t device);

/* 1 if a device schedule exists for state dimension d and observation dimension dy: every d, dy ≤ 4 has a dedicated
 * one-lane-per-chain schedule; any other d ≤ 64 with dy ≤ 64 runs on the MFMA path (state dimension rounded up to a
 * multiple of 16 with decoupled padding dimensions — exact, but priced as the padded size; d ≤ 8 with an even batch of one
 * model: two chains per 16×16 tile, and a batch of one model computes the matrices of the sweep once per engine; several
 * models per engine allowed; any number of chains — batches beyond a grid dimension are launched in slices) */
int32_t rxhip_lgssm_supported(int32_t d, int32_t dy);

/* replaces: new_observation!(datavar, value) (src/inference/batch.jl:405-407).  Copies n doubles
 * from the host; n must equal T*n_chains*dy.  layout: RXHIP_LAYOUT_*. */
rxhip_status rxhip_set_data(rxhip_engine* e, int32_t var_id, const double* host, size_t n, int32_t layout);

/* same, but the observations already live in device memory (layout RXHIP_LAYOUT_TIME_CHAIN is
 * used in place, zero-copy: the buffer must stay valid until the next set_data / destroy). */
rxhip_status rxhip_set_data_device(rxhip_engine* e, int32_t var_id, const double* dev, size_t n, int32_t layout);

/* replaces: the iteration loop `for iteration in 1:iterations` with its synchronous reactive
 * cascade (src/inference/batch.jl:391-430).  Every iteration re-pushes the data and recomputes
 * every message, marginal and (if want_free_energy) the Bethe free energy, as the reference
 * does.  Synchronous: returns after the stream has drained (drivers read results immediately,
 * batch.jl:415).  Non-SPD matrices / non-finite free energy are reported as status codes. */
rxhip_status rxhip_run(rxhip_engine* e, int32_t iterations, int32_t want_free_energy);

/* asynchronous variant used for measurement: enqueues the same work, does not wait. */
rxhip_status rxhip_run_async(rxhip_engine* e, int32_t iterations, int32_t want_free_energy);
/* replaces: the streaming driver with `@autoupdates` posterior -> prior feedback kept on the device
 * (src/inference/streaming.jl:349-407, src/inference/autoupdates.jl:640-659) for the one-step state-space graph of
 * the benchmark notebook (`linear_gaussian_ssm_filtering`, cell 4; driver `rxinfer_inference_filtering`, cell 7):
 *     x_min_t ~ MvNormal(μ = x_min_t_mean, Σ = x_min_t_cov);  x_t ~ MvNormal(μ = A * x_min_t, Σ = P);
 *     y_t ~ MvNormal(μ = B * x_t, Σ = Q);   x_min_t_mean, x_min_t_cov = mean_cov(q(x_t))
 * with `initialization q(x_t) = MvNormalMeanCovariance(m0, V0)` (engine created with prior_through_transition = 1;
 * with 0 the first observation sees the prior on x_1 directly).  One observation per chain and time index is
 * pushed through the one-step graph; the history of q(x_t) (`historyvars = (x_t = KeepLast(),)`, `keephistory = T`)
 * is what rxhip_get_marginals returns afterwards.  The time loop is evaluated parallel-in-time (segment aggregation
 * + prefix scan + forward kernel), exact to rounding.  Free energy (if requested): rxhip_get_free_energy returns ONE
 * value, Σ_chains mean_t F_t — the mean over observations of the per-observation Bethe free energy that the
 * reference's `free_energy_history` holds for a streaming run (src/score/actor.jl:98-104);
 * rxhip_get_free_energy_per_chain the per-chain means.  LGSSM engines only. */
rxhip_status rxhip_run_filter(rxhip_engine* e, int32_t want_free_energy);

/* The same driver ONE OBSERVATION AT A TIME — what `RxInferenceEngine` does with an unbounded datastream
 * (src/inference/streaming.jl:349-407: `on_next!` pushes the datum, the one-step graph fires, `@autoupdates` turns the posterior
 * into the next prior): y holds the new observation of every chain, [chains][dy] on the host (NaN = `missing`: the belief is
 * propagated only); mean [chains][d], cov [chains][d][d] and free_energy [chains] (−log p(y_k | y_<k); any of the three may be
 * NULL) receive the posterior after this observation.  The belief stays on the device between calls; the first call after
 * creation or rxhip_filter_reset starts from the prior.  Per-step constants (step_model) and known inputs are indexed by the
 * number of observations seen (streams longer than the engine's T + horizon need a time-invariant model).  Any d, dy ≤ 64
 * (one thread per chain at d, dy ≤ 4, one workgroup per chain above). */
rxhip_status rxhip_filter_step(rxhip_engine* e, const double* y, double* mean, double* cov, double* free_energy);
rxhip_status rxhip_filter_reset(rxhip_engine* e);
rxhip_status rxhip_run_filter_async(rxhip_engine* e, int32_t want_free_energy);
/* waits for the engine's stream and collects device-side diagnostics (status as rxhip_run) */
rxhip_status rxhip_sync(rxhip_engine* e);

/* replaces: obtain_marginal(var) |> subscribe! + KeepLast actor (reactivemp_inference.jl:626-629,
 * batch.jl:325-340) followed by mean_cov of the posterior (src/inference/postprocess.jl:32-38).
 * mean: T*n_chains*d doubles, cov: T*n_chains*d*d doubles (either may be NULL), in `layout`. */
rxhip_status rxhip_get_marginals(rxhip_engine* e, int32_t var_id, double* mean, double* cov, int32_t layout);

/* The whole body of a static `infer(...)` call on an existing engine in ONE round trip — new_observation!, the iteration loop,
 * the marginal and free-energy actors (src/inference/batch.jl:387-430): y [T][chain][dy] in, mean [T+horizon][chain][d], cov
 * […][d][d] and the per-chain free energy out (any of the three may be NULL), one pinned copy each way and a single
 * synchronisation.  At the reference's own benchmark sizes the four blocking calls of the plain sequence (≈12 µs each) are most
 * of the run time; problems above 512 KB of traffic take the plain sequence internally.  Layout: RXHIP_LAYOUT_TIME_CHAIN. */
rxhip_status rxhip_lgssm_infer(rxhip_engine* e, const double* y, size_t n, int32_t iterations, int32_t want_free_energy,
                               int32_t filtering /* 1: the streaming twin, rxhip_run_filter */, double* mean, double* cov,
                               double* free_energy_per_chain);

/* replaces: obtain_prediction(var) |> subscribe! (reactivemp_inference.jl:619-624; `predictvars = (y = KeepLast(),)`):
 * the message toward every data variable y[t], N(B m, B V B' + Q) with (m, V) the product of the forward and backward
 * messages into x[t] — its own observation excluded — and, for the `horizon` unobserved steps, of the forward prediction.
 * mean: (T+horizon)*n_chains*dy doubles, cov: …*dy*dy (either may be NULL), in `layout`.  After rxhip_run; any d, dy ≤ 64
 * (d, dy ≤ 4: state-space form, predict_kernels.hpp; the MFMA path: observation-space form with one dy×dy inverse per step,
 * generic_kernels.hpp).
 * (The reference refuses free_energy together with predictions, src/inference/batch.jl:337-341; here both are available.) */
rxhip_status rxhip_get_predictions(rxhip_engine* e, int32_t var_id, double* mean, double* cov, int32_t layout);

/* replaces: ReactiveMP.get_node_local_marginals / `@marginalrule MvNormalMeanCovariance(:out_μ)` (the joint the
 * node's Bethe energy is taken over, reactivemp_free_energy.jl:57-66; deterministic neighbours:
 * reactivemp_force_marginal_computation_plugin.jl:52-98): the node-local joint marginal q(out, μ) of every transition node
 * MvNormalMeanCovariance(out = x[t], μ = A x[t-1], Σ = P) between observed states, t = 2 … T, in (out, μ) order:
 * mean (T-1)*n_chains*2d doubles = [m(x[t]); A m(x[t-1])], cov …*2d*2d = [[V(x[t]), (A X)′], [A X, A V(x[t-1]) A′]] with
 * X = Cov(x[t-1], x[t] | y) (either may be NULL), in `layout` ([T-1][chain][·] or [chain][T-1][·]).
 * node_type: RXHIP_NODE_MVNORMAL_MEAN_COV.  After rxhip_run of a state-space engine; any d, dy ≤ 64 (d, dy ≤ 4: from the sweep's own
 * records; above: Cov(x[t], x[t+1] | y) = G_t V_s(t+1) from the smoother gains the information-form sweep leaves in its records, one product
 * per time index (kd_cross_from_records); batches on the model / data split and packed pairs re-run the sequential kernels into scratch
 * arrays instead — a getter, not a hot path). */
rxhip_status rxhip_get_node_marginals(rxhip_engine* e, int32_t node_type, double* mean, double* cov, int32_t layout);

/* device views of the same results, layout [T][chain][d] and [T][chain][d][d]; valid until the
 * next run / destroy */
rxhip_status rxhip_get_marginals_device(rxhip_engine* e, int32_t var_id, const double** mean_dev,
                                        const double** cov_dev);

/* posteriors of SELECTED chains, chain-major: mean [n][T][d], cov [n][T][d][d] (either may be NULL) — what `infer` returns
 * for each of those chains alone (`result.posteriors[:x]`, src/inference/batch.jl:325-340); a strided device gather + one
 * copy, so that a host can inspect a few chains of a 20 GB batch result without moving the batch. */
rxhip_status rxhip_get_marginals_chains(rxhip_engine* e, int32_t var_id, const int64_t* chains, int64_t n, double* mean,
                                        double* cov);

/* replaces: score(model, BetheFreeEnergy{Float64}, checks) |> ScoreActor
 * (src/model/plugins/reactivemp_free_energy.jl:84-126, src/score/actor.jl:38-63).
 * per_iteration[i] = Bethe free energy of the whole batch (sum over chains) at iteration i of
 * the last rxhip_run; length = iterations of that run. */
rxhip_status rxhip_get_free_energy(rxhip_engine* e, double* per_iteration);
/* free energy of each chain (= what `infer` returns for that chain alone), last iteration; n_chains doubles */
rxhip_status rxhip_get_free_energy_per_chain(rxhip_engine* e, double* per_chain);
/* device address of the batch free energy of the last iteration (1 double) — the buffer the
 * multi-GPU host all-reduces over RCCL */
rxhip_status rxhip_get_free_energy_device(rxhip_engine* e, double** fe_dev);
/* enqueue (on the engine's stream, asynchronously) a device-to-device copy of the batch free
 * energy of the last enqueued iteration into dst_dev[0] — lets the multi-GPU host hand the
 * scalar to RCCL without a host round trip */
rxhip_status rxhip_copy_free_energy_to_device(rxhip_engine* e, double* dst_dev);

/* replaces: the after_message_rule_call / after_product_of_two_messages event counts
 * (src/callbacks/events.jl, counted in test/callbacks/trace_tests.jl:93-104): the number of
 * reference rule calls / pairwise products the work done by the last rxhip_run is equivalent to. */
rxhip_status rxhip_counters(rxhip_engine* e, uint64_t* rule_calls, uint64_t* products, uint64_t* marginals);

/* ------------------------------------------------------------------------------------------
 * Univariate Gaussian mixture, mean-field VMP (BASELINE config 5; reference model
 * test/models/mixtures/gmm_univariate_tests.jl:7-26, K-component form as gmm_multivariate_tests.jl:22-31):
 *     s ~ Dirichlet(alpha0);  m[k] ~ Normal(mean = mu0[k], variance = v0[k]);  p[k] ~ Gamma(shape = a0[k], rate = b0[k]);
 *     z[i] ~ Categorical(s);  y[i] ~ NormalMixture(switch = z[i], m = m, p = p);   q = q(z)q(s)Πq(m[k])Πq(p[k])
 * K = 1 is the iid Gaussian with unknown mean and precision (test/models/models_tests.jl:114-128).
 * init_*: the `@initialization` marginals q(m[k]) = N(mean, var), q(p[k]) = Gamma(shape, rate), q(s) = Dirichlet.
 * Replaces create_model + postprocess_plugin for this family; then rxhip_set_data(RXHIP_VAR_Y, y, N, 0),
 * rxhip_run(iterations, want_free_energy), rxhip_get_free_energy (one value per VMP iteration).
 * PINNING: the order of the mean-field updates inside an iteration (q(z) from the previous marginals; q(s), q(m); q(p) with
 * the new q(m)) is an assumption — the reference's reactive order is not documented and no Julia run is available.  What
 * is pinned to the reference is the FIXED POINT (the multivariate golden free energy); intermediate iterates
 * (rxhip_gmm_get_history rows, free energies of early iterations) match the oracle's restatement, not yet RxInfer itself
 * (gate: tests/golden/dump_rxinfer_reference.jl).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int64_t N;      /* observations held by THIS engine (one shard when several GPUs split the data) */
    int32_t K;      /* components, 1..16 */
    const double* mu0; const double* v0; const double* a0; const double* b0; const double* alpha0;        /* [K] priors */
    const double* init_m_mean; const double* init_m_var; const double* init_p_shape; const double* init_p_rate;
    const double* init_s_alpha;                                                                          /* [K] q init */
    int32_t materialize_responsibilities; /* 0: never, 1: q(z) of the last iteration is written to HBM ([N][K]) */
    int32_t device;
    void* stream;
} rxhip_gmm_desc;
rxhip_status rxhip_gmm_create(const rxhip_gmm_desc* desc, rxhip_engine** out);
/* posteriors after every iteration of the last run (KeepEach): hist[iterations][5][K] =
 * (mean m, var m, shape p, rate p, alpha s)   — replaces the marginal actors of src/inference/batch.jl:325-340 */
rxhip_status rxhip_gmm_get_history(rxhip_engine* e, double* hist);
/* q(z[i]) of the last iteration, [N][K]; requires materialize_responsibilities = 1 */
rxhip_status rxhip_gmm_get_responsibilities(rxhip_engine* e, double* resp);
/* split-phase iteration for several GPUs: accumulate() streams this shard and leaves the 3K'+1
 * responsibility-weighted statistics (K' = padded K, see n) in a device buffer; the host all-reduces that
 * buffer over RCCL; update() forms the new marginals / free energy from the (global) statistics.
 * rxhip_run == begin_run + iterations × (accumulate, update). */
rxhip_status rxhip_gmm_begin_run(rxhip_engine* e, int32_t iterations);
rxhip_status rxhip_gmm_accumulate(rxhip_engine* e);
rxhip_status rxhip_gmm_statistics_device(rxhip_engine* e, double** stats_dev, int32_t* n);
rxhip_status rxhip_gmm_update(rxhip_engine* e, int32_t want_free_energy);

/* ------------------------------------------------------------------------------------------
 * Multivariate Gaussian mixture, mean-field VMP (reference model test/models/mixtures/gmm_multivariate_tests.jl:6-32):
 *     m[k] ~ MvNormal(mean = mu0[k], cov = S0[k]);  w[k] ~ Wishart(nu0[k], V0[k]);  s ~ Dirichlet(alpha0);
 *     z[i] ~ Categorical(s);  y[i] ~ NormalMixture(switch = z[i], m = m, p = w)           (w: precision matrices)
 * d = 1…4; K ≤ 16 (d ≤ 2) or K ≤ 8 (d = 3, 4).  init_*: the `@initialization` marginals q(m[k]) = N(mean, cov),
 * q(w[k]) = Wishart(nu, V), q(s) = Dirichlet.  Same handle protocol as the univariate engine: rxhip_set_data(RXHIP_VAR_Y,
 * y [N][d], N*d, 0), rxhip_run / the split-phase rxhip_gmm_begin_run, _accumulate, _statistics_device, _update (statistics:
 * K'(1 + d + d(d+1)/2) + 1 doubles), rxhip_get_free_energy, rxhip_gmm_get_responsibilities.
 * rxhip_gmm_get_history: hist[iterations][K][2 + d + 2d²] = per component  mean[d] | cov[d][d] | nu | V[d][d] | alpha.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int64_t N;
    int32_t K;
    int32_t d;
    const double* mu0;   /* [K][d]    */
    const double* S0;    /* [K][d][d] prior covariance of m[k] */
    const double* nu0;   /* [K]       Wishart degrees of freedom (> d − 1) */
    const double* V0;    /* [K][d][d] Wishart scale */
    const double* alpha0;       /* [K] */
    const double* init_m_mean;  /* [K][d]    */
    const double* init_m_cov;   /* [K][d][d] */
    const double* init_w_nu;    /* [K]       */
    const double* init_w_V;     /* [K][d][d] */
    const double* init_s_alpha; /* [K]       */
    int32_t materialize_responsibilities;
    int32_t device;
    void* stream;
} rxhip_mvgmm_desc;
rxhip_status rxhip_mvgmm_create(const rxhip_mvgmm_desc* desc, rxhip_engine** out);

/* ------------------------------------------------------------------------------------------
 * Hierarchical Gaussian filter, online (BASELINE config 4; reference model + driver
 * test/models/statespace/hgf_tests.jl:9-70): per observation the one-step graph
 *     zt_min ~ Normal(zm, zv); xt_min ~ Normal(xm, xv); zt ~ Normal(mean = zt_min, var = z_variance);
 *     xt ~ GCV(xt_min, zt, kappa, omega); y ~ Normal(mean = xt, var = y_variance);  q = q(xt, xt_min) q(zt)
 * is iterated `iterations` times (rxhip_run's argument = `iterations = vmp_iters` of infer) and the posteriors
 * of zt, xt feed the next observation's priors (@autoupdates).  n_series independent series run side by side.
 * Data: rxhip_set_data(RXHIP_VAR_Y, y, T*n_series, layout) with k = 1.  Replaces streaming_inference's
 * per-event loop (src/inference/streaming.jl:349-407) for this model.
 * rxhip_get_free_energy: per iteration, Σ_series of the mean-over-observations free energy
 * (free_energy_history, src/score/actor.jl:98-104); rxhip_get_free_energy_per_chain: per series, last iteration.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int64_t T;
    int64_t n_series;
    double kappa, omega, z_variance, y_variance;
    double z0_mean, z0_var, x0_mean, x0_var; /* @initialization q(zt), q(xt) */
    int32_t n_gh;                            /* Gauss–Hermite points (GaussHermiteCubature(n)), 1..32 */
    int32_t device;
    void* stream;
} rxhip_hgf_desc;
rxhip_status rxhip_hgf_create(const rxhip_hgf_desc* desc, rxhip_engine** out);
/* history[:zt], history[:xt] (KeepLast per observation): means and variances, each T*n_series doubles in `layout` */
rxhip_status rxhip_hgf_get_history(rxhip_engine* e, double* z_mean, double* z_var, double* x_mean, double* x_var,
                                   int32_t layout);

/* ------------------------------------------------------------------------------------------
 * Noise-free drift chain (reference model test/models/statespace/ulgssm_tests.jl:8-15):
 *     x_prior ~ Normal(μ = m0, v = v0);  x[t] ~ x[t-1] + c;  y[t] ~ Normal(μ = x[t], v = obs_var)     t = 1…T
 * (prior_through_transition = 1: the spelling above, x[1] = x_prior + c; 0: the prior sits on x[1] itself).  Every
 * transition is a deterministic typeof(+) node, so the sweep through the `+`(:out) / `+`(:in1) rules and the message
 * products is a reduction over time (csrc/drift_kernels.hpp).  Same handle protocol as the state-space engine:
 * rxhip_set_data(RXHIP_VAR_Y, y, T*n_chains, layout), rxhip_run, rxhip_get_marginals (d = 1: mean, variance of x[1…T]),
 * rxhip_get_free_energy[_per_chain].  rxhip_create builds it from the graph (RXHIP_NODE_ADD).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int64_t T;
    int64_t n_chains;
    double m0, v0, c, obs_var;
    int32_t prior_through_transition;
    int32_t device;
    void* stream;
} rxhip_drift_chain_desc;
rxhip_status rxhip_drift_chain_create(const rxhip_drift_chain_desc* desc, rxhip_engine** out);

/* ------------------------------------------------------------------------------------------
 * Several GPUs (one process per GPU; chains / series / points shard).  The path's only exchange is the sum
 * over shards of the Bethe free energy (reference: the single `sumreduce` of src/model/plugins/reactivemp_free_energy.jl:99-123
 * over ALL nodes and variables of the model) and, for the mixture, of the responsibility-weighted statistics that
 * form the messages toward m[k], p[k], s (products over all points, reactivemp_inference.jl:365-374).  Both run over
 * RCCL (xGMI) on the engine's stream, in place on the engine's device buffers, as all-gather + a sum in ascending rank
 * order: every rank obtains bit-identical totals, and the same totals on every run.  librccl is opened on first use.
 * The communicator is a plain ncclComm_t: either the host's own (e.g. from AMDGPU.jl / torch) or one made with the three
 * helpers below — rank 0 creates the 128-byte id, the host moves it to the other ranks by any means (file, socket, MPI.jl),
 * every rank calls rxhip_comm_init_rank (collective).  Errors of the helpers: rxhip_comm_last_error() (thread-local).
 * ------------------------------------------------------------------------------------------ */
rxhip_status rxhip_comm_unique_id(char* id128);
rxhip_status rxhip_comm_init_rank(void** comm, int32_t nranks, const char* id128, int32_t rank, int32_t device /* −1: current */);
rxhip_status rxhip_comm_destroy(void* comm);
const char* rxhip_comm_last_error(void);
/* after rxhip_run[_async](…, want_free_energy = 1): per_iteration free energies of the last run become the sums over all
 * ranks (asynchronous on the engine's stream; rxhip_get_free_energy afterwards returns the global values on every rank).  Every engine kind,
 * the node-array executor included: its replicas shard over the ranks like chains do, and this sum is the path's only exchange. */
rxhip_status rxhip_allreduce_free_energy(rxhip_engine* e, void* rccl_comm);
/* between rxhip_gmm_accumulate and rxhip_gmm_update: the statistics buffer becomes the sum over all ranks */
rxhip_status rxhip_gmm_allreduce_statistics(rxhip_engine* e, void* rccl_comm);

/* ------------------------------------------------------------------------------------------
 * measurement hooks (no reference counterpart; RxInferBenchmarkCallbacks is the closest,
 * src/callbacks/benchmark.jl:99-155)
 * ------------------------------------------------------------------------------------------ */
enum {
    RXHIP_K_SEG_AGGREGATE = 0, /* phase 1: per-segment Kalman "elements" (reads y)            */
    RXHIP_K_BOUNDARY_SCAN = 1, /* phase 2: prefix/suffix scan over segment boundaries          */
    RXHIP_K_FORWARD = 2,       /* phase 3: forward messages + evidence terms (reads y, writes fwd) */
    RXHIP_K_BACKWARD = 3,      /* phase 4: backward messages + marginals (reads fwd, writes posteriors) */
    RXHIP_K_FE_REDUCE = 4,     /* Bethe free-energy reduction                                  */
    RXHIP_K_GMM_PASS = 5,      /* mixture: responsibilities + weighted statistics (streams y)  */
    RXHIP_K_GMM_REDUCE = 6,    /* mixture: block partials -> totals                            */
    RXHIP_K_GMM_UPDATE = 7,    /* mixture: new marginals of m[k], p[k], s + free energy        */
    RXHIP_K_HGF_FILTER = 8,    /* hierarchical Gaussian filter: all observations × VMP iterations */
    RXHIP_K_DRIFT_CHAIN = 9,   /* noise-free drift chain: time reduction + marginals + free energy */
    RXHIP_K_COUNT = 10
};
/* enable (1) / disable (0) per-kernel HIP-event timing on the engine's stream */
rxhip_status rxhip_set_profiling(rxhip_engine* e, int32_t enabled);
/* average duration in milliseconds of each kernel (index RXHIP_K_*) over all launches since
 * profiling was enabled / last reset, and the number of launches averaged */
rxhip_status rxhip_get_kernel_times(rxhip_engine* e, double* ms_avg, uint64_t* launches);
rxhip_status rxhip_reset_kernel_times(rxhip_engine* e);
/* device time in milliseconds of the kernels that ran ONCE at creation because their results depend on the model only
 * (gain, covariance and smoother-gain tables of a batch that shares one model); 0 for engines without
 * such tables.  Waits for those kernels.  The reference has no counterpart: it recomputes these messages for every chain
 * and every call (src/inference/batch.jl:391-430). */
rxhip_status rxhip_get_model_tables_ms(rxhip_engine* e, double* ms);
/* Where the host time of the engine's creation went, in milliseconds: ms4[0] host arithmetic on model tables (models of state
 * dimension < 32; 0 for tables that came from the cache), ms4[1] device kernels that build tables (enqueue — or completion where
 * their status is needed before the engine exists), ms4[2] uploads (tables, constants; host → device copies + their sync),
 * ms4[3] device memory (hipMalloc, or the engine pool).  A 23 GB engine is one hipMalloc whose cost depends on the state of
 * the device's page tables, not on this library: bench.py reports the four next to engine_create_ms.
 * replaces: the split `create_model` / inference timing of src/callbacks/benchmark.jl:172-207 */
rxhip_status rxhip_get_create_stages(rxhip_engine* e, double* ms4);
/* Posterior covariances of a batch of chains that share ONE model do not depend on the data: they are one [T][d][d] table per model.
 * mode 0 (default): every sweep writes them into the posterior array of every chain — what the reference's marginal actors do
 *   (`src/inference/batch.jl:325-340`: every marginal of every variable, every iteration), and what every BASELINE timing here includes.
 * mode 1: shared-model batches on the MFMA path (4 < d ≤ 64, one model, ≥ 4 workgroup chains) keep the table and write the per-chain array
 *   when somebody asks for it (rxhip_get_marginals / _device / _chains, rxhip_lgssm_infer with cov != NULL, predictions, node-local
 *   joints) — at most once until the next run that rewrites it; means and free energy are unaffected.  Every other engine ignores the
 *   mode.  (d = 8 × 1024 chains × T = 1000: 0.56 -> 0.40 ms per sweep; the per-chain copies are 30 % of it.) */
rxhip_status rxhip_set_covariance_mode(rxhip_engine* e, int32_t mode);
/* The fixed-point exits of the sweeps ("Fixed-point exits" at the top of this file) per engine.  enabled = 0: every recursion of every later sweep of this
 * engine runs in full to the end of every segment — no frozen stretches on the MFMA path, no mean-only records / early exits of the per-chain d, dy ≤ 4
 * kernels (what RXHIP_NO_FROZEN / RXHIP_ELEM_FULL do under the test hooks, as an API a host can rely on); 1: the default.  The model tables an engine was
 * created with (built once per model, shared between engines) keep their own convergence tests.  State-space engines; others: RXHIP_ERR_UNSUPPORTED. */
rxhip_status rxhip_set_fixed_point_exits(rxhip_engine* e, int32_t enabled);
/* the hipStream_t the engine launches on */
rxhip_status rxhip_get_stream(rxhip_engine* e, void** stream);
/* the number of time segments the schedule uses and their length */
rxhip_status rxhip_get_schedule(rxhip_engine* e, int32_t* segments, int64_t* segment_len);

const char* rxhip_last_error(const rxhip_engine* e); /* never NULL; valid until the next call on e */
const char* rxhip_status_string(rxhip_status s);
const char* rxhip_version(void);
int32_t rxhip_device_count(void); /* number of visible HIP devices (0 if none) */
rxhip_status rxhip_destroy(rxhip_engine* e);
/* frees the engines, device blocks and pinned blocks parked by destroyed engines (see the note on process-wide state at the top) */
rxhip_status rxhip_release_cached_memory(void);
/* enabled = 0: no process-wide caching at all from now on — idle streams, device blocks, pinned blocks, per-model tables and parked engines are released now
 * and nothing is parked or shared again (a destroyed engine frees everything it owns, engines of the same model build their own tables); 1 (the default)
 * restores the pools.  Callable at any time from any thread; handles that are alive keep what they hold until they are destroyed. */
rxhip_status rxhip_set_caching(int32_t enabled);
/* The state-space engines of d > 4 carry filtered PRECISIONS (information form, parallel in time) and are validated inside an envelope of model conditioning:
 * kappa = max(cond(V0p^-1 + B'Q^-1 B), cond((1 - rho^2) P^-1 + B'Q^-1 B)), rho the spectral radius of A, V0p the prior of the first observed state, cond of the matrix scaled to unit diagonal (units of the state do not count) — a vague prior or a slowly
 * forgetting transition in directions the observations do not see.  Beyond 2e3 (d <= 16) / 1e3 (d > 16) rxhip_lgssm_create returns RXHIP_ERR_UNSUPPORTED with NO
 * handle (text: rxhip_lowering_error()), and rxhip_create runs the same graph on the node-array executor instead, which holds 1e-10 on such models
 * (csrc/model_envelope.hpp; measured: scripts/calib_dense_envelope.py, profiles/r06/dense_envelope.txt — up to 19 posterior standard deviations wrong at
 * kappa = 3e5 without the check).  enabled = 0 switches the check off for the process (a host that knows its models, measurements); 1 (the default) restores it.
 * The reference has no counterpart: its rules run in covariance form one message at a time.  Engines of d <= 4 carry no check (a long chain has no second home);
 * their measured envelope: filtering within the contract up to kappa = 1e7, smoothing while lambda_max(Q^-1 (B V0p B' + Q)) <= 1e5 and kappa <= 1e6
 * (profiles/r06/dense_envelope.txt). */
rxhip_status rxhip_set_conditioning_guard(int32_t enabled);

#ifdef __cplusplus
}
#endif
#endif /* RXHIP_H */
