# RxHip.jl — Julia host shim over librxhip (include/rxhip.h).
#
# Host code stays in Julia; the device is reached through a thin `ccall` layer (no AMDGPU.jl kernels,
# no KernelAbstractions).  This file is what a maintainer adds next to
# src/model/plugins/reactivemp_inference.jl; it was written against the C header and CANNOT be executed
# in the build image (no Julia toolchain there, SURVEY.md §0 F3).  The Python package
# rxinfer.jl_amd/rxhip binds the same symbols and is what the parity tests drive.
module RxHip

using LinearAlgebra

const librxhip = get(ENV, "RXHIP_LIB", joinpath(@__DIR__, "..", "csrc", "librxhip.so"))

# ---- status ------------------------------------------------------------------------------------
const RXHIP_OK = Int32(0)
const RXHIP_ERR_NOT_POSDEF = Int32(3)
const RXHIP_LAYOUT_TIME_CHAIN = Int32(0)
const RXHIP_LAYOUT_CHAIN_TIME = Int32(1)
const RXHIP_VAR_Y = Int32(0)
const RXHIP_VAR_X = Int32(1)
const RXHIP_VAR_U = Int32(2)

struct RxHipError <: Exception
    status::Int32
    msg::String
end
Base.showerror(io::IO, e::RxHipError) = print(io, "rxhip status ", e.status, ": ", e.msg)

# mirrors rxhip_lgssm_desc (include/rxhip.h); field order and types must match the C struct
struct LgssmDesc
    d::Int32
    dy::Int32
    T::Int64
    n_chains::Int64
    n_models::Int32
    prior_through_transition::Int32
    A::Ptr{Float64}
    B::Ptr{Float64}
    P::Ptr{Float64}
    Q::Ptr{Float64}
    m0::Ptr{Float64}
    V0::Ptr{Float64}
    chain_model::Ptr{Int32}
    segments::Int32
    device::Int32
    stream::Ptr{Cvoid}
    horizon::Int64
    allow_missing::Int32
    step_model::Ptr{Int32}
    state_offset::Ptr{Float64}
    obs_offset::Ptr{Float64}
end

mutable struct Engine
    handle::Ptr{Cvoid}
    d::Int
    dy::Int
    T::Int
    n_chains::Int
    iterations::Int
end

function check(e::Engine, st::Int32)
    st == RXHIP_OK && return nothing
    msg = unsafe_string(ccall((:rxhip_last_error, librxhip), Cstring, (Ptr{Cvoid},), e.handle))
    # non-zero status becomes a Julia exception so that inference_process_error
    # (src/inference/inference.jl:328-386) and catch_exception (src/inference/batch.jl:440-446) behave as today
    st == RXHIP_ERR_NOT_POSDEF && throw(PosDefException(0))
    throw(RxHipError(st, msg))
end

rowmajor(M::AbstractMatrix) = vec(collect(transpose(Matrix{Float64}(M))))   # Julia is column-major
rowmajor(v::AbstractVector) = Vector{Float64}(v)

"""
    Engine(A, B, P, Q, m0, V0; T, n_chains = 1, prior_through_transition = false, segments = 0, device = -1)

Replaces `create_model` + `postprocess_plugin(::ReactiveMPInferencePlugin, model)` for the linear Gaussian
state-space family (src/inference/batch.jl:252, src/model/plugins/reactivemp_inference.jl:272-326).
`P` is the state-noise covariance, `Q` the observation-noise covariance.
"""
function Engine(A, B, P, Q, m0, V0; T::Integer, n_chains::Integer = 1, prior_through_transition::Bool = false,
                segments::Integer = 0, device::Integer = -1, chain_model::Union{Nothing, AbstractVector{<:Integer}} = nothing,
                stream = nothing, horizon::Integer = 0, allow_missing::Bool = false,
                step_model::Union{Nothing, AbstractVector{<:Integer}} = nothing,
                state_offset::Union{Nothing, AbstractMatrix} = nothing, obs_offset::Union{Nothing, AbstractMatrix} = nothing)
    # one model: plain matrices; several: vectors of matrices (A[m], B[m], …) with chain_model[c] ∈ 0:n_models-1 (a model per
    # chain) or step_model[t] ∈ 0:n_models-1 (per-step constants `A[t] * x[t-1]`, shared by the chains; T + horizon entries)
    multi = A isa AbstractVector{<:AbstractMatrix}
    n_models = multi ? length(A) : 1
    cat(xs) = multi ? reduce(vcat, rowmajor.(xs)) : rowmajor(xs)
    d, dy = multi ? (size(A[1], 1), size(B[1], 1)) : (size(A, 1), size(B, 1))
    a, b, p, q, m, v = cat(A), cat(B), cat(P), cat(Q), cat(m0), cat(V0)
    cm = chain_model === nothing ? Int32[] : Vector{Int32}(chain_model)
    (chain_model === nothing || length(cm) == n_chains) || throw(ArgumentError("chain_model needs one entry per chain"))
    sm = step_model === nothing ? Int32[] : Vector{Int32}(step_model)
    (step_model === nothing || length(sm) == T + horizon) || throw(ArgumentError("step_model needs one entry per time index"))
    # known inputs, d × (T + horizon) / dy × (T + horizon) (one column per time index: row-major [t][·] on the C side)
    cx = state_offset === nothing ? Float64[] : vec(Matrix{Float64}(state_offset))
    cy = obs_offset === nothing ? Float64[] : vec(Matrix{Float64}(obs_offset))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    st = GC.@preserve a b p q m v cm sm cx cy begin
        desc = LgssmDesc(d, dy, T, n_chains, n_models, prior_through_transition ? 1 : 0, pointer(a), pointer(b), pointer(p),
                         pointer(q), pointer(m), pointer(v), isempty(cm) ? Ptr{Int32}(C_NULL) : pointer(cm), segments, device,
                         stream_handle(stream), horizon, allow_missing ? 1 : 0, isempty(sm) ? Ptr{Int32}(C_NULL) : pointer(sm),
                         isempty(cx) ? Ptr{Float64}(C_NULL) : pointer(cx), isempty(cy) ? Ptr{Float64}(C_NULL) : pointer(cy))
        ccall((:rxhip_lgssm_create, librxhip), Int32, (Ref{LgssmDesc}, Ref{Ptr{Cvoid}}), desc, h)
    end
    e = Engine(h[], d, dy, T + horizon, n_chains, 0)   # T counts the rows of the result arrays (observed + horizon)
    if st != RXHIP_OK
        h[] != C_NULL && (try check(e, st) finally ccall((:rxhip_destroy, librxhip), Int32, (Ptr{Cvoid},), h[]) end)
        throw(RxHipError(st, unsafe_string(ccall((:rxhip_status_string, librxhip), Cstring, (Int32,), st))))
    end
    finalizer(destroy!, e)   # ownership: the engine owns its device memory until destroy
    return e
end

# mirrors rxhip_noise_prior
struct NoisePrior
    nu0::Float64
    S0::Ptr{Float64}
    init_nu::Float64
    init_V::Ptr{Float64}
end

"""
    NoiseEngine(A, B, P, m0, V0; T, nu0, S0, init_nu = nu0, init_V = S0, n_chains = 1, ...)

The state-space chain with an UNKNOWN observation-noise precision — `W ~ Wishart(nu0, S0); y[t] ~ MvNormal(μ = B * x[t], Λ = W)` under
`@constraints q(x, W) = q(x)q(W)` (chain: test/models/statespace/mlgssm_test.jl:9-14, node pair: test/models/iid/mv_iid_precision_tests.jl:11-15).
`run!(e, iterations; free_energy = true)` alternates one BP sweep of every chain with every chain's Wishart update on the device
(rxhip_lgssm_noise_create, csrc/noise_kernels.hpp); `noise_posterior(e)` returns q(W) of every chain after the last iteration.
"""
function NoiseEngine(A, B, P, m0, V0; T::Integer, nu0::Real, S0::AbstractMatrix, init_nu::Real = nu0, init_V::AbstractMatrix = S0,
                     n_chains::Integer = 1, prior_through_transition::Bool = false, segments::Integer = 0, device::Integer = -1, stream = nothing)
    d, dy = size(A, 1), size(B, 1)
    a, b, p, m, v, s0, iv = rowmajor(A), rowmajor(B), rowmajor(P), rowmajor(m0), rowmajor(V0), rowmajor(S0), rowmajor(init_V)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    st = GC.@preserve a b p m v s0 iv begin
        desc = LgssmDesc(d, dy, T, n_chains, 1, prior_through_transition ? 1 : 0, pointer(a), pointer(b), pointer(p), Ptr{Float64}(C_NULL),
                         pointer(m), pointer(v), Ptr{Int32}(C_NULL), segments, device, stream_handle(stream), 0, 0, Ptr{Int32}(C_NULL),
                         Ptr{Float64}(C_NULL), Ptr{Float64}(C_NULL))
        prior = NoisePrior(nu0, pointer(s0), init_nu, pointer(iv))
        ccall((:rxhip_lgssm_noise_create, librxhip), Int32, (Ref{LgssmDesc}, Ref{NoisePrior}, Ref{Ptr{Cvoid}}), desc, prior, h)
    end
    e = Engine(h[], d, dy, T, n_chains, 0)
    if st != RXHIP_OK
        h[] != C_NULL && (try check(e, st) finally ccall((:rxhip_destroy, librxhip), Int32, (Ptr{Cvoid},), h[]) end)
        throw(RxHipError(st, unsafe_string(ccall((:rxhip_status_string, librxhip), Cstring, (Int32,), st))))
    end
    finalizer(destroy!, e)
    return e
end

"""Later `run!` calls continue from the current q(W) instead of the `@initialization` marginal (`rxhip_lgssm_noise_continue`): the plugin takes
one VMP iteration per `fire!`, as the loop of src/inference/batch.jl:391-430 does."""
noise_continue!(e::Engine, on::Bool = true) =
    check(e, ccall((:rxhip_lgssm_noise_continue, librxhip), Int32, (Ptr{Cvoid}, Int32), e.handle, on ? 1 : 0))

"""The same for an engine of the node-array executor (`rxhip_tree_continue`): without it every `run!(e; iterations = 1)` of the plugin's `fire!` would
restart from the `@initialization` q(W) and `infer(iterations = N)` would return the iteration-1 posterior N times."""
tree_continue!(e::Engine, on::Bool = true) =
    check(e, ccall((:rxhip_tree_continue, librxhip), Int32, (Ptr{Cvoid}, Int32), e.handle, on ? 1 : 0))

"""q(W) of every chain after the last iteration: (ν [chains], V [dy, dy, chains])."""
function noise_posterior(e::Engine)
    nu = Vector{Float64}(undef, e.n_chains)
    V = Array{Float64}(undef, e.dy, e.dy, e.n_chains)      # symmetric blocks: row- and column-major coincide
    GC.@preserve nu V check(e, ccall((:rxhip_lgssm_noise_get, librxhip), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), e.handle, nu, V))
    return nu, V
end

function destroy!(e::Engine)
    e.handle == C_NULL && return
    ccall((:rxhip_destroy, librxhip), Int32, (Ptr{Cvoid},), e.handle)
    e.handle = C_NULL
    return
end

"""`new_observation!(datavar, value)` (src/inference/batch.jl:405-407): y is a Vector (chains) of Vector (time) of Vector{Float64}."""
function set_data!(e::Engine, y::AbstractVector)
    chains = e.n_chains == 1 && eltype(y) <: AbstractVector{<:Real} ? (y,) : y
    flat = Vector{Float64}(undef, e.n_chains * length(first(chains)) * e.dy)   # the observed rows only (without the horizon)
    k = 1
    for yc in chains, yt in yc, v in yt
        flat[k] = v; k += 1
    end
    GC.@preserve flat check(e, ccall((:rxhip_set_data, librxhip), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}, Csize_t, Int32),
                                     e.handle, RXHIP_VAR_Y, flat, length(flat), RXHIP_LAYOUT_CHAIN_TIME))
end

"""The iteration loop of src/inference/batch.jl:391-430 (synchronous)."""
function run!(e::Engine; iterations::Integer = 1, free_energy::Bool = false)
    check(e, ccall((:rxhip_run, librxhip), Int32, (Ptr{Cvoid}, Int32, Int32), e.handle, iterations, free_energy ? 1 : 0))
    e.iterations = iterations
end

"""posteriors[:x] as (means, covariances), chain-major; replaces obtain_marginal + mean_cov (reactivemp_inference.jl:626-629)."""
function marginals(e::Engine)
    mean = Array{Float64}(undef, e.d, e.T, e.n_chains)            # column-major view of [chain][T][d]
    cov = Array{Float64}(undef, e.d, e.d, e.T, e.n_chains)
    GC.@preserve mean cov check(e, ccall((:rxhip_get_marginals, librxhip), Int32,
                                         (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}, Int32),
                                         e.handle, RXHIP_VAR_X, mean, cov, RXHIP_LAYOUT_CHAIN_TIME))
    return mean, cov   # covariances are symmetric, so the row-/column-major distinction is immaterial
end

"""`result.predictions[:y]` (`predictvars = (y = KeepLast(),)`, reactivemp_inference.jl:619-624): mean dy × T × chains,
cov dy × dy × T × chains — leave-one-out predictive for observed steps, forecasts for the `horizon` missing ones."""
function predictions(e::Engine)
    mean = Array{Float64}(undef, e.dy, e.T, e.n_chains)
    cov = Array{Float64}(undef, e.dy, e.dy, e.T, e.n_chains)
    GC.@preserve mean cov check(e, ccall((:rxhip_get_predictions, librxhip), Int32,
                                         (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}, Int32),
                                         e.handle, RXHIP_VAR_Y, mean, cov, RXHIP_LAYOUT_CHAIN_TIME))
    return mean, cov
end

"""Data inputs u[t] of `A * x[t-1] + B_u * u[t]` (graph engines, `rxhip_set_data(RXHIP_VAR_U)`): `flat` is [chain][t][du] row-major."""
function set_inputs!(e::Engine, flat::Vector{Float64})
    GC.@preserve flat check(e, ccall((:rxhip_set_data, librxhip), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}, Csize_t, Int32),
                                     e.handle, RXHIP_VAR_U, flat, length(flat), RXHIP_LAYOUT_CHAIN_TIME))
end

"""`get_node_local_marginals` of the transition nodes `x[t] ~ MvNormal(μ = A * x[t-1], Σ = P)`, t = 2 … T: the joint q(out, μ) in
(out, μ) order — mean 2d × (T-1) × chains, cov 2d × 2d × (T-1) × chains (`rxhip_get_node_marginals`; any d, dy ≤ 64)."""
function node_marginals(e::Engine)
    d2, n = 2 * e.d, e.T - 1
    mean = Array{Float64}(undef, d2, n, e.n_chains)
    cov = Array{Float64}(undef, d2, d2, n, e.n_chains)
    GC.@preserve mean cov check(e, ccall((:rxhip_get_node_marginals, librxhip), Int32,
                                         (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}, Int32),
                                         e.handle, Int32(1), mean, cov, RXHIP_LAYOUT_CHAIN_TIME))
    return mean, permutedims(cov, (2, 1, 3, 4))   # row-major blocks -> Julia's column-major
end

"""score(model, BetheFreeEnergy, checks) |> ScoreActor (reactivemp_free_energy.jl:84-126, score/actor.jl:38-63)."""
function free_energy(e::Engine)
    fe = Vector{Float64}(undef, e.iterations)
    GC.@preserve fe check(e, ccall((:rxhip_get_free_energy, librxhip), Int32, (Ptr{Cvoid}, Ptr{Float64}), e.handle, fe))
    return fe
end

"""The body of a static `infer(...)` in one round trip (`rxhip_lgssm_infer`): observations `y` (vector of dy-vectors, one chain) in;
posterior mean d × T, covariance d × d × T and the free energy out.  `filtering = true`: the streaming twin."""
function infer!(e::Engine, y::AbstractVector; iterations::Integer = 1, free_energy::Bool = false, filtering::Bool = false)
    flat = reduce(vcat, Vector{Float64}.(y))
    mean = Array{Float64}(undef, e.d, e.T, e.n_chains)
    cov = Array{Float64}(undef, e.d, e.d, e.T, e.n_chains)
    fe = Vector{Float64}(undef, e.n_chains)
    e.n_chains == 1 || throw(ArgumentError("infer!: one chain (batches: set_data! / run! / marginals)"))
    GC.@preserve flat mean cov fe check(e, ccall((:rxhip_lgssm_infer, librxhip), Int32,
        (Ptr{Cvoid}, Ptr{Float64}, Csize_t, Int32, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        e.handle, flat, length(flat), iterations, free_energy ? 1 : 0, filtering ? 1 : 0, mean, cov, fe))
    e.iterations = iterations
    return mean, cov, (free_energy ? fe[1] : nothing)
end

"""One observation at a time, as `RxInferenceEngine` consumes a datastream (src/inference/streaming.jl:349-407): `y` holds the new
observation of every chain (dy × chains; `NaN` = missing); returns q(x) after it (mean d × chains, cov d × d × chains) and
−log p(y_k | y_<k) per chain.  The belief stays on the device between calls (`filter_reset!` starts over from the prior)."""
function filter_step!(e::Engine, y::AbstractMatrix)
    flat = vec(Matrix{Float64}(y))                       # column-major dy × chains = row-major [chain][dy]
    mean = Array{Float64}(undef, e.d, e.n_chains)
    cov = Array{Float64}(undef, e.d, e.d, e.n_chains)
    fe = Vector{Float64}(undef, e.n_chains)
    GC.@preserve flat mean cov fe check(e, ccall((:rxhip_filter_step, librxhip), Int32,
                                                  (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                                                  e.handle, flat, mean, cov, fe))
    return mean, cov, fe                                  # covariances are symmetric: no transposition needed
end
"""New known inputs (d × T / dy × T matrices or `nothing` = zeros) for an engine created with offsets (`rxhip_lgssm_set_offsets`)."""
function set_offsets!(e::Engine; state_offset = nothing, obs_offset = nothing)
    cx = state_offset === nothing ? Float64[] : vec(Matrix{Float64}(state_offset))
    cy = obs_offset === nothing ? Float64[] : vec(Matrix{Float64}(obs_offset))
    GC.@preserve cx cy check(e, ccall((:rxhip_lgssm_set_offsets, librxhip), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), e.handle,
                                      isempty(cx) ? Ptr{Float64}(C_NULL) : pointer(cx), isempty(cy) ? Ptr{Float64}(C_NULL) : pointer(cy)))
end
filter_reset!(e::Engine) = check(e, ccall((:rxhip_filter_reset, librxhip), Int32, (Ptr{Cvoid},), e.handle))

"""Streaming twin: `infer(model = linear_gaussian_ssm_filtering(...), data = (y_t = observations,), autoupdates = ...,
historyvars = (x_t = KeepLast(),), keephistory = n)` of the benchmark notebook (cell 7; driver
src/inference/streaming.jl:349-407).  Afterwards `marginals(e)` is `result.history[:x_t]` and `free_energy(e)[1]` the
mean-over-observations free energy (`free_energy_history`)."""
function run_filter!(e::Engine; free_energy::Bool = false)
    check(e, ccall((:rxhip_run_filter, librxhip), Int32, (Ptr{Cvoid}, Int32), e.handle, free_energy ? 1 : 0))
    e.iterations = 1
end

# ---- mean-field families (same handle type; the generic entry points run!/free_energy/counters apply) -----

# mirrors rxhip_gmm_desc
struct GmmDesc
    N::Int64
    K::Int32
    mu0::Ptr{Float64}; v0::Ptr{Float64}; a0::Ptr{Float64}; b0::Ptr{Float64}; alpha0::Ptr{Float64}
    init_m_mean::Ptr{Float64}; init_m_var::Ptr{Float64}; init_p_shape::Ptr{Float64}; init_p_rate::Ptr{Float64}
    init_s_alpha::Ptr{Float64}
    materialize_responsibilities::Int32
    device::Int32
    stream::Ptr{Cvoid}
end

"""
    MixtureEngine(y; mu0, v0, a0, b0, alpha0, q_m_mean, q_m_var, q_p_shape, q_p_rate, q_s_alpha)

`univariate_gaussian_mixture_model` (test/models/mixtures/gmm_univariate_tests.jl:7-26, K components) with the
`@initialization` marginals; K = 1 is `iid_gaussians_params` (test/models/models_tests.jl:114-128)."""
function MixtureEngine(y::Vector{Float64}; mu0, v0, a0, b0, alpha0, q_m_mean, q_m_var, q_p_shape, q_p_rate, q_s_alpha,
                       materialize_z::Bool = false, device::Integer = -1)
    K = length(mu0)
    vs = map(x -> Vector{Float64}(x), (mu0, v0, a0, b0, alpha0, q_m_mean, q_m_var, q_p_shape, q_p_rate, q_s_alpha))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    st = GC.@preserve vs begin
        desc = GmmDesc(length(y), K, map(pointer, vs)..., materialize_z ? 1 : 0, device, C_NULL)
        ccall((:rxhip_gmm_create, librxhip), Int32, (Ref{GmmDesc}, Ref{Ptr{Cvoid}}), desc, h)
    end
    e = Engine(h[], 1, 1, length(y), 1, 0)
    st == RXHIP_OK || throw(RxHipError(st, unsafe_string(ccall((:rxhip_status_string, librxhip), Cstring, (Int32,), st))))
    finalizer(destroy!, e)
    GC.@preserve y check(e, ccall((:rxhip_set_data, librxhip), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}, Csize_t, Int32),
                                  e.handle, RXHIP_VAR_Y, y, length(y), RXHIP_LAYOUT_TIME_CHAIN))
    return e
end

"""posteriors of m, p, s after every iteration (`returnvars = KeepEach()`): hist[k, j, it], j = (mean m, var m, shape p, rate p, α s)."""
function mixture_history(e::Engine, K::Integer)
    hist = Array{Float64}(undef, K, 5, e.iterations)      # column-major view of [iterations][5][K]
    GC.@preserve hist check(e, ccall((:rxhip_gmm_get_history, librxhip), Int32, (Ptr{Cvoid}, Ptr{Float64}), e.handle, hist))
    return hist
end

# mirrors rxhip_mvgmm_desc
struct MvGmmDesc
    N::Int64
    K::Int32
    d::Int32
    mu0::Ptr{Float64}; S0::Ptr{Float64}; nu0::Ptr{Float64}; V0::Ptr{Float64}; alpha0::Ptr{Float64}
    init_m_mean::Ptr{Float64}; init_m_cov::Ptr{Float64}; init_w_nu::Ptr{Float64}; init_w_V::Ptr{Float64}; init_s_alpha::Ptr{Float64}
    materialize_responsibilities::Int32
    device::Int32
    stream::Ptr{Cvoid}
end

"""
    MvMixtureEngine(y; mu0, S0, nu0, V0, alpha0, q_m_mean, q_m_cov, q_w_nu, q_w_V, q_s_alpha)

`multivariate_gaussian_mixture_model` (test/models/mixtures/gmm_multivariate_tests.jl:6-32); y is d × N (one observation per
column = the C layout [N][d]); matrices per component are symmetric, so Julia's column-major blocks are passed as they are."""
function MvMixtureEngine(y::Matrix{Float64}; mu0, S0, nu0, V0, alpha0, q_m_mean, q_m_cov, q_w_nu, q_w_V, q_s_alpha, device::Integer = -1)
    d, N = size(y)
    K = length(nu0)
    flat(x) = x isa AbstractVector{<:AbstractArray} ? reduce(vcat, vec.(x)) : Vector{Float64}(vec(x))
    vs = map(x -> Vector{Float64}(flat(x)), (mu0, S0, nu0, V0, alpha0, q_m_mean, q_m_cov, q_w_nu, q_w_V, q_s_alpha))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    st = GC.@preserve vs begin
        desc = MvGmmDesc(N, K, d, map(pointer, vs)..., 0, device, C_NULL)
        ccall((:rxhip_mvgmm_create, librxhip), Int32, (Ref{MvGmmDesc}, Ref{Ptr{Cvoid}}), desc, h)
    end
    e = Engine(h[], d, d, N, 1, 0)
    st == RXHIP_OK || throw(RxHipError(st, unsafe_string(ccall((:rxhip_status_string, librxhip), Cstring, (Int32,), st))))
    finalizer(destroy!, e)
    GC.@preserve y check(e, ccall((:rxhip_set_data, librxhip), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}, Csize_t, Int32),
                                  e.handle, RXHIP_VAR_Y, y, length(y), RXHIP_LAYOUT_TIME_CHAIN))
    return e
end

# mirrors rxhip_hgf_desc
struct HgfDesc
    T::Int64
    n_series::Int64
    kappa::Float64; omega::Float64; z_variance::Float64; y_variance::Float64
    z0_mean::Float64; z0_var::Float64; x0_mean::Float64; x0_var::Float64
    n_gh::Int32
    device::Int32
    stream::Ptr{Cvoid}
end

"""`hgf_online_inference` of test/models/statespace/hgf_tests.jl:43-70 for `size(y, 2)` independent series (y: T × series)."""
function HgfEngine(y::Matrix{Float64}; kappa, omega, z_variance, y_variance, q_zt = (0.0, 5.0), q_xt = (0.0, 5.0),
                   n_gh::Integer = 31, device::Integer = -1)
    T, S = size(y)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    desc = HgfDesc(T, S, kappa, omega, z_variance, y_variance, q_zt[1], q_zt[2], q_xt[1], q_xt[2], n_gh, device, C_NULL)
    st = ccall((:rxhip_hgf_create, librxhip), Int32, (Ref{HgfDesc}, Ref{Ptr{Cvoid}}), desc, h)
    e = Engine(h[], 1, 1, T, S, 0)
    st == RXHIP_OK || throw(RxHipError(st, unsafe_string(ccall((:rxhip_status_string, librxhip), Cstring, (Int32,), st))))
    finalizer(destroy!, e)
    # Julia's T × series matrix is series-major in memory = [chain][time]
    GC.@preserve y check(e, ccall((:rxhip_set_data, librxhip), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}, Csize_t, Int32),
                                  e.handle, RXHIP_VAR_Y, y, length(y), RXHIP_LAYOUT_CHAIN_TIME))
    return e
end

"""result.history[:zt], result.history[:xt] as (mean, var) matrices T × series."""
function hgf_history(e::Engine)
    zm, zv, xm, xv = (Matrix{Float64}(undef, e.T, e.n_chains) for _ in 1:4)
    GC.@preserve zm zv xm xv check(e, ccall((:rxhip_hgf_get_history, librxhip), Int32,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32), e.handle, zm, zv, xm, xv, RXHIP_LAYOUT_CHAIN_TIME))
    return (zt = (zm, zv), xt = (xm, xv))
end

function counters(e::Engine)
    r, p, m = Ref{UInt64}(0), Ref{UInt64}(0), Ref{UInt64}(0)
    check(e, ccall((:rxhip_counters, librxhip), Int32, (Ptr{Cvoid}, Ref{UInt64}, Ref{UInt64}, Ref{UInt64}), e.handle, r, p, m))
    return (rule_calls = r[], products = p[], marginals = m[])
end

# ---- the rest of the C ABI as thin wrappers: asynchronous runs, device-pointer hand-over (AMDGPU.jl arrays stay on the device), profiling, housekeeping ----

"per-chain (per-replica) free energy of the last iteration (`rxhip_get_free_energy_per_chain`)"
function free_energy_per_chain(e::Engine)
    fe = Vector{Float64}(undef, e.n_chains)
    GC.@preserve fe check(e, ccall((:rxhip_get_free_energy_per_chain, librxhip), Int32, (Ptr{Cvoid}, Ptr{Float64}), e.handle, fe))
    return fe
end

"enqueue a run on the engine's stream and return (`rxhip_run_async`); `sync!` or any getter waits for it"
function run_async!(e::Engine; iterations::Integer = 1, free_energy::Bool = true)
    check(e, ccall((:rxhip_run_async, librxhip), Int32, (Ptr{Cvoid}, Int32, Int32), e.handle, Int32(iterations), Int32(free_energy ? 1 : 0)))
    e.iterations = iterations
    return e
end
"the streaming twin, enqueued (`rxhip_run_filter_async`)"
run_filter_async!(e::Engine; free_energy::Bool = true) =
    (check(e, ccall((:rxhip_run_filter_async, librxhip), Int32, (Ptr{Cvoid}, Int32), e.handle, Int32(free_energy ? 1 : 0))); e.iterations = 1; e)

"observations that already live on the device (`rxhip_set_data_device`): `ptr` a device pointer to `n` doubles in `layout`; the caller keeps the buffer alive"
set_data_device!(e::Engine, ptr::Ptr{Float64}, n::Integer; var_id::Integer = 0, layout::Integer = RXHIP_LAYOUT_TIME_CHAIN) =
    check(e, ccall((:rxhip_set_data_device, librxhip), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}, Csize_t, Int32), e.handle, Int32(var_id), ptr, Csize_t(n), Int32(layout)))

"device views of the posteriors, layout [T][chain][d] / [T][chain][d][d], valid until the next run (`rxhip_get_marginals_device`)"
function marginals_device(e::Engine; var_id::Integer = 1)
    m, c = Ref{Ptr{Float64}}(C_NULL), Ref{Ptr{Float64}}(C_NULL)
    check(e, ccall((:rxhip_get_marginals_device, librxhip), Int32, (Ptr{Cvoid}, Int32, Ref{Ptr{Float64}}, Ref{Ptr{Float64}}), e.handle, Int32(var_id), m, c))
    return m[], c[]
end

"device pointer to the per-iteration free energies of the last run (`rxhip_get_free_energy_device`)"
function free_energy_device(e::Engine)
    q = Ref{Ptr{Float64}}(C_NULL)
    check(e, ccall((:rxhip_get_free_energy_device, librxhip), Int32, (Ptr{Cvoid}, Ref{Ptr{Float64}}), e.handle, q))
    return q[]
end
"the last iteration's free energy into a caller's device buffer, on the engine's stream (`rxhip_copy_free_energy_to_device`)"
copy_free_energy_to_device!(e::Engine, dst::Ptr{Float64}) =
    check(e, ccall((:rxhip_copy_free_energy_to_device, librxhip), Int32, (Ptr{Cvoid}, Ptr{Float64}), e.handle, dst))

"the hipStream_t the engine launches on (`rxhip_get_stream`)"
function stream(e::Engine)
    q = Ref{Ptr{Cvoid}}(C_NULL)
    check(e, ccall((:rxhip_get_stream, librxhip), Int32, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}), e.handle, q))
    return q[]
end

"(segments, segment length) of the time-parallel schedule (`rxhip_get_schedule`)"
function schedule(e::Engine)
    s, l = Ref{Int32}(0), Ref{Int64}(0)
    check(e, ccall((:rxhip_get_schedule, librxhip), Int32, (Ptr{Cvoid}, Ref{Int32}, Ref{Int64}), e.handle, s, l))
    return (segments = Int(s[]), segment_len = Int(l[]))
end

const RXHIP_K_COUNT = 10   # include/rxhip.h rxhip_kernel_id
"per-kernel device times (`RxInferBenchmarkCallbacks`, src/callbacks/benchmark.jl:99-155): `set_profiling!`, then runs, then `kernel_times`"
set_profiling!(e::Engine, on::Bool) = check(e, ccall((:rxhip_set_profiling, librxhip), Int32, (Ptr{Cvoid}, Int32), e.handle, Int32(on ? 1 : 0)))
reset_kernel_times!(e::Engine) = check(e, ccall((:rxhip_reset_kernel_times, librxhip), Int32, (Ptr{Cvoid},), e.handle))
function kernel_times(e::Engine)
    ms, n = Vector{Float64}(undef, RXHIP_K_COUNT), Vector{UInt64}(undef, RXHIP_K_COUNT)
    GC.@preserve ms n check(e, ccall((:rxhip_get_kernel_times, librxhip), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{UInt64}), e.handle, ms, n))
    return (ms_avg = ms, launches = n)
end

"per-chain known inputs c[t], d[t] of a batch (`rxhip_lgssm_set_chain_offsets`): d × T × chains and dy × T × chains (either may be `nothing`)"
function set_chain_offsets!(e::Engine, state_offset, obs_offset)
    so = state_offset === nothing ? Float64[] : vec(Array{Float64}(state_offset))
    oo = obs_offset === nothing ? Float64[] : vec(Array{Float64}(obs_offset))
    GC.@preserve so oo check(e, ccall((:rxhip_lgssm_set_chain_offsets, librxhip), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int32), e.handle,
                                       isempty(so) ? Ptr{Float64}(C_NULL) : pointer(so), isempty(oo) ? Ptr{Float64}(C_NULL) : pointer(oo), RXHIP_LAYOUT_CHAIN_TIME))
end

"responsibilities q(z[i] = k) of the mixture engine after a run, K × N (`rxhip_gmm_get_responsibilities`)"
function gmm_responsibilities(e::Engine, K::Integer, N::Integer)
    r = Array{Float64}(undef, K, N)
    GC.@preserve r check(e, ccall((:rxhip_gmm_get_responsibilities, librxhip), Int32, (Ptr{Cvoid}, Ptr{Float64}), e.handle, r))
    return r
end
"device pointer and length of the mixture's statistics buffer between `accumulate` and `update` (`rxhip_gmm_statistics_device`)"
function gmm_statistics_device(e::Engine)
    q, n = Ref{Ptr{Float64}}(C_NULL), Ref{Int32}(0)
    check(e, ccall((:rxhip_gmm_statistics_device, librxhip), Int32, (Ptr{Cvoid}, Ref{Ptr{Float64}}, Ref{Int32}), e.handle, q, n))
    return q[], Int(n[])
end

version() = unsafe_string(ccall((:rxhip_version, librxhip), Cstring, ()))
device_count() = Int(ccall((:rxhip_device_count, librxhip), Int32, ()))
"1 if a device schedule exists for state dimension d and observation dimension dy (`rxhip_lgssm_supported`)"
lgssm_supported(d::Integer, dy::Integer) = ccall((:rxhip_lgssm_supported, librxhip), Int32, (Int32, Int32), Int32(d), Int32(dy)) != 0
"hand the library's parked engines, arenas and pinned blocks back to the runtime (`rxhip_release_cached_memory`; INTEGRATION.md: lifetime of the pools)"
release_cached_memory() = ccall((:rxhip_release_cached_memory, librxhip), Int32, ()) == 0
"switch the library's process-wide pools off (`false`: nothing parked, nothing shared between handles) or back on (`rxhip_set_caching`)"
set_caching!(on::Bool) = ccall((:rxhip_set_caching, librxhip), Int32, (Int32,), on ? 1 : 0) == 0
"the conditioning envelope of the information-form chain engines (d > 4): `false` switches the check at creation off (`rxhip_set_conditioning_guard`; include/rxhip.h)"
set_conditioning_guard!(on::Bool) = ccall((:rxhip_set_conditioning_guard, librxhip), Int32, (Int32,), on ? 1 : 0) == 0

"device time (ms) of the kernels that ran once at creation because their results depend on the model only"
function model_tables_ms(e::Engine)
    ms = Ref{Float64}(0.0)
    check(e, ccall((:rxhip_get_model_tables_ms, librxhip), Int32, (Ptr{Cvoid}, Ref{Float64}), e.handle, ms))
    return ms[]
end

"host milliseconds of the engine's creation by stage: (tables_host, tables_device, upload, alloc) — the split of `create_model` time"
function create_stages(e::Engine)
    ms = zeros(Float64, 4)
    GC.@preserve ms check(e, ccall((:rxhip_get_create_stages, librxhip), Int32, (Ptr{Cvoid}, Ptr{Float64}), e.handle, pointer(ms)))
    return (tables_host_ms = ms[1], tables_device_ms = ms[2], upload_ms = ms[3], alloc_ms = ms[4])
end

"false: every recursion of the engine's later sweeps in full — no frozen stretches, no early exits (include/rxhip.h rxhip_set_fixed_point_exits)"
set_fixed_point_exits!(e::Engine, enabled::Bool) =
    check(e, ccall((:rxhip_set_fixed_point_exits, librxhip), Int32, (Ptr{Cvoid}, Int32), e.handle, Int32(enabled ? 1 : 0)))

"0: every sweep writes the covariance of every chain (default); 1: shared-model batches on the MFMA path write the per-chain array on request"
set_covariance_mode!(e::Engine, mode::Integer) =
    check(e, ccall((:rxhip_set_covariance_mode, librxhip), Int32, (Ptr{Cvoid}, Int32), e.handle, Int32(mode)))

# ---- generic graph entry: rxhip_graph_desc / rxhip_create (used by HIPInferencePlugin.jl) ----------------------------
const RXHIP_ERR_UNSUPPORTED = Int32(2)

# mirrors rxhip_graph_desc (include/rxhip.h) field for field
struct GraphDesc
    n_variables::Int64
    var_kind::Ptr{Int32}
    var_rows::Ptr{Int32}
    var_cols::Ptr{Int32}
    var_const::Ptr{Int64}
    n_factors::Int64
    factor_type::Ptr{Int32}
    factor_iface::Ptr{Int64}
    const_pool::Ptr{Float64}
    n_const::Int64
    n_replicas::Int64
    factor_iface_ptr::Ptr{Int64}
    var_init_family::Ptr{Int32}
    var_init::Ptr{Int64}
    gh_points::Int32
    n_observations::Int64
    allow_missing::Int32
    factor_cluster::Ptr{Int32}
end

# mirrors rxhip_lgssm_lowered
mutable struct LgssmLowered
    d::Int32
    dy::Int32
    T::Int64
    prior_through_transition::Int32
    A::Ptr{Float64}; B::Ptr{Float64}; P::Ptr{Float64}; Q::Ptr{Float64}; m0::Ptr{Float64}; V0::Ptr{Float64}
    state_var::Ptr{Int64}
    data_var::Ptr{Int64}
    deterministic::Int32
    c::Ptr{Float64}
    n_models::Int32
    step_model::Ptr{Int32}
    has_offsets::Int32
    state_offset::Ptr{Float64}
    obs_offset::Ptr{Float64}
    du::Int32
    input_matrix::Ptr{Float64}
    input_var::Ptr{Int64}
    LgssmLowered() = new(0, 0, 0, 0, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, 0, C_NULL, 0, C_NULL, 0, C_NULL, C_NULL,
                         0, C_NULL, C_NULL)
end

# rxhip_lgssm_lowered BY VALUE — the first member of rxhip_lgssm_noise_lowered.  A mutable Julia struct is stored by reference inside another
# struct; this immutable twin of LgssmLowered (same fields, same order) is stored inline, as C does.
struct LgssmLoweredFields
    d::Int32
    dy::Int32
    T::Int64
    prior_through_transition::Int32
    A::Ptr{Float64}; B::Ptr{Float64}; P::Ptr{Float64}; Q::Ptr{Float64}; m0::Ptr{Float64}; V0::Ptr{Float64}
    state_var::Ptr{Int64}
    data_var::Ptr{Int64}
    deterministic::Int32
    c::Ptr{Float64}
    n_models::Int32
    step_model::Ptr{Int32}
    has_offsets::Int32
    state_offset::Ptr{Float64}
    obs_offset::Ptr{Float64}
    du::Int32
    input_matrix::Ptr{Float64}
    input_var::Ptr{Int64}
end
LgssmLoweredFields(; state_var = Ptr{Int64}(C_NULL), data_var = Ptr{Int64}(C_NULL)) =
    LgssmLoweredFields(0, 0, 0, 0, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, state_var, data_var, 0, C_NULL, 0, C_NULL, 0, C_NULL, C_NULL, 0, C_NULL, C_NULL)

# mirrors rxhip_lgssm_noise_lowered
mutable struct LgssmNoiseLowered
    chain::LgssmLoweredFields
    precision_var::Int64
    nu0::Float64
    init_nu::Float64
    S0::Ptr{Float64}
    init_V::Ptr{Float64}
    LgssmNoiseLowered(chain = LgssmLoweredFields()) = new(chain, -1, 0.0, 0.0, C_NULL, C_NULL)
end

"""The engine's stream: `nothing` (engine-owned) or an AMDGPU.jl stream, whose raw `hipStream_t` is handed over so that the
host's own kernels / copies and the engine's launches are ordered on one queue (AMDGPU.jl only for handles)."""
stream_handle(::Nothing) = Ptr{Cvoid}(C_NULL)
stream_handle(s) = Base.unsafe_convert(Ptr{Cvoid}, s)     # AMDGPU.HIPStream -> hipStream_t

with_desc(f, t, n_replicas::Integer, n_observations::Integer; allow_missing::Bool = false) = GC.@preserve t begin
    f(GraphDesc(length(t.var_kind), pointer(t.var_kind), pointer(t.var_rows), pointer(t.var_cols), pointer(t.var_const),
                length(t.factor_type), pointer(t.factor_type), pointer(t.factor_iface), pointer(t.const_pool), length(t.const_pool),
                n_replicas, pointer(t.factor_iface_ptr), pointer(t.var_init_family), pointer(t.var_init), t.gh_points, n_observations,
                allow_missing ? 1 : 0, pointer(t.factor_cluster)))
end

lowering_error() = unsafe_string(ccall((:rxhip_lowering_error, librxhip), Cstring, ()))
"the node-array executor asked for directly (`rxhip_tree_create`; `create_from_tables` reaches it through `rxhip_create` for graphs no family matches)"
function tree_create(g::Ref{GraphDesc}; device::Integer = -1, stream = C_NULL)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    st = ccall((:rxhip_tree_create, librxhip), Int32, (Ptr{GraphDesc}, Int32, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), g, Int32(device), stream, h)
    st == 0 || error("rxhip_tree_create: status $st: $(lowering_error())")
    return h[]
end

"the graph compiler alone (host only): would the node-array executor take this graph, and with what schedule? (include/rxhip.h rxhip_tree_plan)"
function tree_plan(g::Ref{GraphDesc})
    info = Ref(TreeInfo(0, 0, 0, 0, 0, 0, 0, 0, 0, 0.0, 0, 0, 0, 0, 0, 0))
    rc, pr, mg = Ref{UInt64}(0), Ref{UInt64}(0), Ref{UInt64}(0)
    st = ccall((:rxhip_tree_plan, librxhip), Int32, (Ptr{GraphDesc}, Ref{TreeInfo}, Ref{UInt64}, Ref{UInt64}, Ref{UInt64}), g, info, rc, pr, mg)
    st == 0 || error("rxhip_tree_plan: status $st: $(lowering_error())")
    return (info = info[], rule_calls = rc[], products = pr[], marginals = mg[])
end

"largest relative asymmetry of a constant parameter the last lowering call accepted and symmetrised (include/rxhip.h rxhip_lowering_asymmetry)"
lowering_asymmetry() = ccall((:rxhip_lowering_asymmetry, librxhip), Cdouble, ())

"""
    create_from_tables(tables; n_replicas = 1, n_observations = 0, segments = 0, device = -1, stream = nothing, allow_missing = false)

`rxhip_create`: lowers the graph tables and builds the engine of the family the node types select (`allow_missing`: a
state-space engine that takes `missing` = NaN observations anywhere in the data, rxhip_lgssm_desc.allow_missing).  Throws
`RxHipError(RXHIP_ERR_UNSUPPORTED, why)` for graphs without a device schedule (the plugin then uses ReactiveMP)."""
function create_from_tables(t; n_replicas::Integer = 1, n_observations::Integer = 0, segments::Integer = 0, device::Integer = -1, stream = nothing,
                            allow_missing::Bool = false)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    st = with_desc(t, n_replicas, n_observations; allow_missing = allow_missing) do desc
        ccall((:rxhip_create, librxhip), Int32, (Ref{GraphDesc}, Int32, Int32, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), desc, segments, device,
              stream_handle(stream), h)
    end
    if st != RXHIP_OK
        msg = lowering_error()
        if h[] != C_NULL
            isempty(msg) && (msg = unsafe_string(ccall((:rxhip_last_error, librxhip), Cstring, (Ptr{Cvoid},), h[])))
            ccall((:rxhip_destroy, librxhip), Int32, (Ptr{Cvoid},), h[])
        end
        st == RXHIP_ERR_NOT_POSDEF && throw(PosDefException(0))
        throw(RxHipError(st, msg))
    end
    probe = Engine(h[], 0, 0, 0, n_replicas, 0)
    if tree_info(probe) !== nothing   # the pattern matcher had no family for this graph: the node-array executor took it (rxhip_tree_* entry points)
        finalizer(destroy!, probe)
        return probe
    end
    lay = lowered_layout(t)
    e = Engine(h[], lay.d, lay.width, length(lay.data_ids), n_replicas, 0)
    finalizer(destroy!, e)
    return e
end

"""Layout of a graph the node-array executor runs: every data variable (id order) with its offset in the staging vector, every random variable
that is not a precision variable (the `out` of a Wishart / Gamma node) with its dimension, the precision variables."""
function tree_layout(t)
    prec = Set{Int64}()
    for f in eachindex(t.factor_type)
        t.factor_type[f] in (Int32(12), Int32(5), Int32(15)) && push!(prec, t.factor_iface[t.factor_iface_ptr[f] + 1])
    end
    # the discrete side of a mixture layer: switches (out of Categorical / Bernoulli) and their probability vectors (out of Dirichlet / Beta) — read with
    # rxhip_tree_get_discrete, not as Gaussian marginals
    switch, probs = Set{Int64}(), Set{Int64}()
    for f in eachindex(t.factor_type)
        t.factor_type[f] in (Int32(8), Int32(9)) && push!(switch, t.factor_iface[t.factor_iface_ptr[f] + 1])
        t.factor_type[f] in (Int32(6), Int32(7)) && push!(probs, t.factor_iface[t.factor_iface_ptr[f] + 1])
    end
    data = Int64[i - 1 for i in eachindex(t.var_kind) if t.var_kind[i] == Int32(1)]
    rnd = Int64[i - 1 for i in eachindex(t.var_kind) if t.var_kind[i] == Int32(0) && !((i - 1) in prec) && !((i - 1) in switch) && !((i - 1) in probs)]
    # univariate variables (their marginals are NormalMeanVariance, not one-dimensional MvNormals): what a Normal(…) node touches, carried through `*` and `+`
    scal = Set{Int64}()
    ifs(f) = t.factor_iface[(t.factor_iface_ptr[f] + 1):t.factor_iface_ptr[f + 1]]
    for f in eachindex(t.factor_type)
        t.factor_type[f] in (Int32(3), Int32(4)) && union!(scal, ifs(f)[1:2])
    end
    grew = true
    while grew
        grew = false
        for f in eachindex(t.factor_type)
            t.factor_type[f] in (Int32(2), Int32(13)) || continue
            io = t.factor_type[f] == Int32(2) ? ifs(f)[[1, 3]] : ifs(f)
            if any(in(scal), io) && !all(in(scal), io) && all(i -> t.var_rows[i + 1] == 1, io)
                union!(scal, io)
                grew = true
            end
        end
    end
    offs, o = Int[], 0
    for id in data
        push!(offs, o)
        o += Int(t.var_rows[id + 1])
    end
    return (family = :tree, data_ids = data, data_offsets = offs, data_total = o, state_ids = rnd, state_dims = Int[Int(t.var_rows[id + 1]) for id in rnd],
            precision_ids = sort!(collect(prec)), scalar_ids = scal, switch_ids = sort!(collect(switch)), probability_ids = sort!(collect(probs)),
            bernoulli = Set{Int64}(t.factor_iface[t.factor_iface_ptr[f] + 1] for f in eachindex(t.factor_type) if t.factor_type[f] in (Int32(9), Int32(7))), gamma = Dict(id => any(f -> t.factor_type[f] != Int32(12) && t.factor_iface[t.factor_iface_ptr[f] + 1] == id,
                                                                        eachindex(t.factor_type)) for id in prec))
end

"""Family of the graph and the variable ids (0-based, time order) of its observations / states, from the host-only lowering
entry points (`rxhip_graph_lower_*`, two-call protocol)."""
function lowered_layout(t)
    has(code) = any(==(Int32(code)), t.factor_type)
    if has(11)
        return (family = :hgf, d = 1, width = 1, data_ids = Int64[findfirst(==(Int32(1)), t.var_kind) - 1], state_ids = Int64[])
    elseif !has(10) && (has(12) || has(5) || has(15)) && (has(2) || any(eachindex(t.factor_type)) do f   # as rxhip_create decides: a precision prior over a CHAIN
            io = t.factor_iface[(t.factor_iface_ptr[f] + 1):t.factor_iface_ptr[f + 1]]
            t.factor_type[f] in (Int32(1), Int32(3), Int32(4), Int32(14)) && length(io) == 3 && t.var_kind[io[1] + 1] == Int32(0) && t.var_kind[io[2] + 1] == Int32(0)
        end)
        # a precision prior (Wishart; Gamma for scalar observations) on top of a chain with `*` nodes: the state-space chain with an unknown
        # observation-noise precision (rxhip_graph_lower_lgssm_noise)
        low = LgssmNoiseLowered()
        st = with_desc(t, 1, 0) do desc
            ccall((:rxhip_graph_lower_lgssm_noise, librxhip), Int32, (Ref{GraphDesc}, Ref{LgssmNoiseLowered}), desc, low)
        end
        st == RXHIP_OK || throw(RxHipError(st, lowering_error()))
        T = low.chain.T
        sv, dv = Vector{Int64}(undef, T), Vector{Int64}(undef, T)
        GC.@preserve sv dv begin
            low.chain = LgssmLoweredFields(state_var = pointer(sv), data_var = pointer(dv))
            st = with_desc(t, 1, 0) do desc
                ccall((:rxhip_graph_lower_lgssm_noise, librxhip), Int32, (Ref{GraphDesc}, Ref{LgssmNoiseLowered}), desc, low)
            end
        end
        st == RXHIP_OK || throw(RxHipError(st, lowering_error()))
        return (family = :lgssm_noise, d = Int(low.chain.d), width = Int(low.chain.dy), data_ids = dv, state_ids = sv, precision_id = low.precision_var,
                gamma = !has(12))
    elseif has(10) || has(4) || (has(14) && has(12))
        ids = Int64[]
        for f in eachindex(t.factor_type)   # the observation nodes' `out` interface, in node order = data order
            out = t.factor_iface[t.factor_iface_ptr[f] + 1]
            # NormalMixture / Normal(mean, precision) observation nodes; MvNormal(μ, Λ) counts when its `out` is data (iid family)
            (t.factor_type[f] in (Int32(10), Int32(4)) || (t.factor_type[f] == Int32(14) && t.var_kind[out + 1] == Int32(1))) && push!(ids, out)
        end
        d = Int(t.var_rows[ids[1] + 1])
        return (family = has(12) ? :mvmixture : :mixture, d = d, width = d, data_ids = ids, state_ids = Int64[])
    end
    low = LgssmLowered()
    st = with_desc(t, 1, 0) do desc
        ccall((:rxhip_graph_lower_lgssm, librxhip), Int32, (Ref{GraphDesc}, Ref{LgssmLowered}), desc, low)
    end
    st == RXHIP_OK || throw(RxHipError(st, lowering_error()))
    sv, dv, uv = Vector{Int64}(undef, low.T), Vector{Int64}(undef, low.T), Vector{Int64}(undef, low.T)
    GC.@preserve sv dv uv begin
        low.state_var, low.data_var, low.input_var = pointer(sv), pointer(dv), pointer(uv)
        st = with_desc(t, 1, 0) do desc
            ccall((:rxhip_graph_lower_lgssm, librxhip), Int32, (Ref{GraphDesc}, Ref{LgssmLowered}), desc, low)
        end
    end
    st == RXHIP_OK || throw(RxHipError(st, lowering_error()))
    # data inputs `A * x[t-1] + B_u * u[t]`: variable id of u[t] per time index (−1: that transition has none), dimension du
    return (family = low.deterministic != 0 ? :drift : :lgssm, d = Int(low.d), width = Int(low.dy), data_ids = dv, state_ids = sv,
            du = Int(low.du), input_ids = low.du > 0 ? uv : Int64[])
end

"""Split-phase VMP iteration of the mixture engines (`rxhip_gmm_begin_run` once, then accumulate + update per iteration):
what the plugin's `fire!` runs for a mixture graph, because `rxhip_run` restarts from the `@initialization` marginals."""
function vmp_iteration!(e::Engine; free_energy::Bool, max_iterations::Integer = 1000)
    if e.iterations == 0
        check(e, ccall((:rxhip_gmm_begin_run, librxhip), Int32, (Ptr{Cvoid}, Int32), e.handle, max_iterations))
    end
    e.iterations < max_iterations || throw(RxHipError(Int32(7), "more than max_iterations = $(max_iterations) VMP iterations"))
    check(e, ccall((:rxhip_gmm_accumulate, librxhip), Int32, (Ptr{Cvoid},), e.handle))
    check(e, ccall((:rxhip_gmm_update, librxhip), Int32, (Ptr{Cvoid}, Int32), e.handle, free_energy ? 1 : 0))
    check(e, ccall((:rxhip_sync, librxhip), Int32, (Ptr{Cvoid},), e.handle))
    e.iterations += 1
end

"""Univariate mixture: the marginals of the iteration just run as `(mean m, var m, shape p, rate p, α s)[k]`; multivariate:
per component `mean[d] | cov[d][d] | ν | V[d][d] | α` — the last block of `rxhip_gmm_get_history`."""
function last_mixture_state(e::Engine, K::Integer; d::Integer = 0)
    stride = d == 0 ? 5 * K : K * (2 + d + 2 * d * d)
    hist = Vector{Float64}(undef, stride * e.iterations)
    GC.@preserve hist check(e, ccall((:rxhip_gmm_get_history, librxhip), Int32, (Ptr{Cvoid}, Ptr{Float64}), e.handle, hist))
    return hist[(end - stride + 1):end]
end

"""posteriors of selected chains only (`rxhip_get_marginals_chains`): chains are 0-based ids."""
function marginals_of_chains(e::Engine, chains::Vector{Int64})
    mean = Array{Float64}(undef, e.d, e.T, length(chains))
    cov = Array{Float64}(undef, e.d, e.d, e.T, length(chains))
    GC.@preserve chains mean cov check(e, ccall((:rxhip_get_marginals_chains, librxhip), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Int64}, Int64, Ptr{Float64}, Ptr{Float64}), e.handle, RXHIP_VAR_X, chains, length(chains), mean, cov))
    return mean, cov
end

# ---- the level-scheduled node-array executor (include/rxhip.h rxhip_tree_*): any acyclic Gaussian graph -----------------------------
# mirrors rxhip_tree_info
struct TreeInfo
    n_ops::Int64
    n_levels::Int64
    n_messages::Int64
    doubles_per_replica::Int64
    bytes_per_sweep::Int64
    dmax::Int32
    mode::Int32
    replicas_per_workgroup::Int32
    n_precision_vars::Int32
    last_iteration_ms::Float64
    io_bytes_per_sweep::Int64
    n_strands::Int64
    n_strand_levels::Int64
    longest_strand::Int32
    kernels::Int32
    strand_bytes_per_sweep::Int64
    fe_bytes_per_sweep::Int64
end
# mirrors rxhip_rule_call
struct RuleCall
    node_type::Int32
    iface::Int32
    d_out::Int32
    d_in::Int32
    n::Int64
    constant::Ptr{Float64}
    in_form::Int32
    in_a::Ptr{Float64}
    in_B::Ptr{Float64}
    in2_a::Ptr{Float64}
    in2_B::Ptr{Float64}
    out_form::Int32
    out_a::Ptr{Float64}
    out_B::Ptr{Float64}
end

"""`rxhip_tree_get_info`, or `nothing` when the handle belongs to one of the pattern-matched families (what `rxhip_create` built
tells the plugin which set of entry points drives the engine)."""
function tree_info(e::Engine)
    info = Ref(TreeInfo(0, 0, 0, 0, 0, 0, 0, 0, 0, 0.0, 0, 0, 0, 0, 0, 0))
    st = ccall((:rxhip_tree_get_info, librxhip), Int32, (Ptr{Cvoid}, Ref{TreeInfo}), e.handle, info)
    return st == RXHIP_OK ? info[] : nothing
end

"""Observations of the listed data variables (0-based ids), `flat` = their values side by side, one replica."""
function tree_set_data!(e::Engine, ids::Vector{Int64}, flat::Vector{Float64})
    GC.@preserve ids flat check(e, ccall((:rxhip_tree_set_data, librxhip), Int32, (Ptr{Cvoid}, Ptr{Int64}, Int64, Ptr{Float64}),
                                        e.handle, ids, length(ids), flat))
end

"""Posteriors of the listed random variables (0-based ids, dimensions `dims`): vectors of means and of covariance matrices (replica 1)."""
function tree_marginals(e::Engine, ids::Vector{Int64}, dims::Vector{Int})
    R = e.n_chains
    mean, cov = Vector{Float64}(undef, R * sum(dims)), Vector{Float64}(undef, R * sum(d -> d * d, dims))
    GC.@preserve ids mean cov check(e, ccall((:rxhip_tree_get_marginals, librxhip), Int32,
        (Ptr{Cvoid}, Ptr{Int64}, Int64, Ptr{Float64}, Ptr{Float64}), e.handle, ids, length(ids), mean, cov))
    ms, Vs, mo, co = Vector{Vector{Float64}}(), Vector{Matrix{Float64}}(), 0, 0
    for d in dims   # [var][replica][d] / [var][replica][d][d], row-major: replica 1 of every variable
        push!(ms, mean[(mo + 1):(mo + d)])
        push!(Vs, collect(transpose(reshape(cov[(co + 1):(co + d * d)], d, d))))
        mo += R * d
        co += R * d * d
    end
    return ms, Vs
end

"""q(W) = Wishart(ν, V) of a precision variable (replica 1); a Gamma(a, b) variable comes back as Wishart₁(2a, 1/(2b))."""
function tree_precision(e::Engine, id::Integer, d::Integer)
    R = e.n_chains
    nu, V = Vector{Float64}(undef, R), Vector{Float64}(undef, R * d * d)
    GC.@preserve nu V check(e, ccall((:rxhip_tree_get_precision, librxhip), Int32, (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}), e.handle, id, nu, V))
    return nu[1], collect(transpose(reshape(V[1:(d * d)], d, d)))
end

"""q(z = k) of a mixture node's switch, or the concentrations of q(s) of its probability vector (replica 1)."""
function tree_discrete(e::Engine, id::Integer)
    K = Ref{Int32}(0)
    check(e, ccall((:rxhip_tree_get_discrete, librxhip), Int32, (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Int32}), e.handle, id, C_NULL, K))
    out = Vector{Float64}(undef, e.n_chains * K[])
    GC.@preserve out check(e, ccall((:rxhip_tree_get_discrete, librxhip), Int32, (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Int32}), e.handle, id, out, C_NULL))
    return out[1:K[]]
end

"""One message rule on the device (`rxhip_rule_eval`): the A/B hook next to `ReactiveMP.@call_rule`.  `a`: d × n, `B`: d × d × n (column-major
Julia arrays of symmetric matrices: the C side reads them row-major, the same bytes)."""
function rule_eval(node_type::Integer, iface::Integer, constant, a::Matrix{Float64}, B::Array{Float64, 3}; a2 = nothing, B2 = nothing,
                   in_form::Integer = 0, out_form::Integer = 0, d_out::Integer = size(a, 1), d_in::Integer = size(a, 1), device::Integer = -1)
    n = size(a, 2)
    dres = node_type == 2 ? (iface == 0 ? d_out : d_in) : d_out
    oa, oB = Matrix{Float64}(undef, dres, n), Array{Float64, 3}(undef, dres, dres, n)
    c = constant === nothing ? Float64[] : collect(Float64, transpose(constant))[:]   # row-major for the C side
    GC.@preserve c a B a2 B2 oa oB begin
        call = RuleCall(node_type, iface, d_out, d_in, n, isempty(c) ? C_NULL : pointer(c), in_form, pointer(a), pointer(B),
                        a2 === nothing ? C_NULL : pointer(a2), B2 === nothing ? C_NULL : pointer(B2), out_form, pointer(oa), pointer(oB))
        st = ccall((:rxhip_rule_eval, librxhip), Int32, (Ref{RuleCall}, Int32), call, device)
        st == RXHIP_OK || throw(RxHipError(st, lowering_error()))
    end
    return oa, oB
end

# mirrors rxhip_drift_chain_desc
struct DriftChainDesc
    T::Int64
    n_chains::Int64
    m0::Float64; v0::Float64; c::Float64; obs_var::Float64
    prior_through_transition::Int32
    device::Int32
    stream::Ptr{Cvoid}
end
"""`univariate_lgssm_model` of test/models/statespace/ulgssm_tests.jl:8-15 (noise-free `x[i] ~ x_prev + c` transitions)."""
function DriftChainEngine(; T::Integer, m0, v0, c, obs_var, n_chains::Integer = 1, prior_through_transition::Bool = true,
                          device::Integer = -1, stream = nothing)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    desc = DriftChainDesc(T, n_chains, m0, v0, c, obs_var, prior_through_transition ? 1 : 0, device, stream_handle(stream))
    st = ccall((:rxhip_drift_chain_create, librxhip), Int32, (Ref{DriftChainDesc}, Ref{Ptr{Cvoid}}), desc, h)
    e = Engine(h[], 1, 1, T, n_chains, 0)
    st == RXHIP_OK || throw(RxHipError(st, unsafe_string(ccall((:rxhip_status_string, librxhip), Cstring, (Int32,), st))))
    finalizer(destroy!, e)
    return e
end

# ---- several GPUs: the free-energy / statistics exchange over RCCL (no torch, no MPI dependency in this file) ----------
"""128-byte RCCL id, created by rank 0 and moved to the other ranks by the host (file, socket, MPI.jl, Distributed)."""
function comm_unique_id()
    id = Vector{UInt8}(undef, 128)
    st = ccall((:rxhip_comm_unique_id, librxhip), Int32, (Ptr{UInt8},), id)
    st == RXHIP_OK || throw(RxHipError(st, unsafe_string(ccall((:rxhip_comm_last_error, librxhip), Cstring, ()))))
    return id
end
function comm_init_rank(nranks::Integer, id::Vector{UInt8}, rank::Integer; device::Integer = -1)
    c = Ref{Ptr{Cvoid}}(C_NULL)
    st = ccall((:rxhip_comm_init_rank, librxhip), Int32, (Ref{Ptr{Cvoid}}, Int32, Ptr{UInt8}, Int32, Int32), c, nranks, id, rank, device)
    st == RXHIP_OK || throw(RxHipError(st, unsafe_string(ccall((:rxhip_comm_last_error, librxhip), Cstring, ()))))
    return c[]
end
comm_destroy(comm::Ptr{Cvoid}) = ccall((:rxhip_comm_destroy, librxhip), Int32, (Ptr{Cvoid},), comm)
"""after `run!(…, free_energy = true)`: the free energies become the sums over all ranks (bit-identical on every rank)."""
allreduce_free_energy!(e::Engine, comm::Ptr{Cvoid}) =
    check(e, ccall((:rxhip_allreduce_free_energy, librxhip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), e.handle, comm))
"""between `rxhip_gmm_accumulate` and `rxhip_gmm_update` of a sharded mixture"""
allreduce_statistics!(e::Engine, comm::Ptr{Cvoid}) =
    check(e, ccall((:rxhip_gmm_allreduce_statistics, librxhip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), e.handle, comm))

end # module
