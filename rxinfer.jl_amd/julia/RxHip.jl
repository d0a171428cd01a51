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

rowmajor(M::AbstractMatrix) = collect(transpose(Matrix{Float64}(M)))   # Julia is column-major
rowmajor(v::AbstractVector) = Vector{Float64}(v)

"""
    Engine(A, B, P, Q, m0, V0; T, n_chains = 1, prior_through_transition = false, segments = 0, device = -1)

Replaces `create_model` + `postprocess_plugin(::ReactiveMPInferencePlugin, model)` for the linear Gaussian
state-space family (src/inference/batch.jl:252, src/model/plugins/reactivemp_inference.jl:272-326).
`P` is the state-noise covariance, `Q` the observation-noise covariance.
"""
function Engine(A, B, P, Q, m0, V0; T::Integer, n_chains::Integer = 1, prior_through_transition::Bool = false,
                segments::Integer = 0, device::Integer = -1)
    d, dy = size(A, 1), size(B, 1)
    a, b, p, q, m, v = rowmajor(A), rowmajor(B), rowmajor(P), rowmajor(Q), rowmajor(m0), rowmajor(V0)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    st = GC.@preserve a b p q m v begin
        desc = LgssmDesc(d, dy, T, n_chains, 1, prior_through_transition ? 1 : 0, pointer(a), pointer(b), pointer(p),
                         pointer(q), pointer(m), pointer(v), Ptr{Int32}(C_NULL), segments, device, C_NULL)
        ccall((:rxhip_lgssm_create, librxhip), Int32, (Ref{LgssmDesc}, Ref{Ptr{Cvoid}}), desc, h)
    end
    e = Engine(h[], d, dy, T, n_chains, 0)
    if st != RXHIP_OK
        h[] != C_NULL && (try check(e, st) finally ccall((:rxhip_destroy, librxhip), Int32, (Ptr{Cvoid},), h[]) end)
        throw(RxHipError(st, unsafe_string(ccall((:rxhip_status_string, librxhip), Cstring, (Int32,), st))))
    end
    finalizer(destroy!, e)   # ownership: the engine owns its device memory until destroy
    return e
end

function destroy!(e::Engine)
    e.handle == C_NULL && return
    ccall((:rxhip_destroy, librxhip), Int32, (Ptr{Cvoid},), e.handle)
    e.handle = C_NULL
    return
end

"""`new_observation!(datavar, value)` (src/inference/batch.jl:405-407): y is a Vector (chains) of Vector (time) of Vector{Float64}."""
function set_data!(e::Engine, y::AbstractVector)
    flat = Vector{Float64}(undef, e.n_chains * e.T * e.dy)
    k = 1
    chains = e.n_chains == 1 && eltype(y) <: AbstractVector{<:Real} ? (y,) : y
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

"""score(model, BetheFreeEnergy, checks) |> ScoreActor (reactivemp_free_energy.jl:84-126, score/actor.jl:38-63)."""
function free_energy(e::Engine)
    fe = Vector{Float64}(undef, e.iterations)
    GC.@preserve fe check(e, ccall((:rxhip_get_free_energy, librxhip), Int32, (Ptr{Cvoid}, Ptr{Float64}), e.handle, fe))
    return fe
end

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

# ---- plugin sketch -----------------------------------------------------------------------------------
# A sibling of ReactiveMPInferencePlugin (src/model/plugins/reactivemp_inference.jl:207-326), selected with
# `options = (backend = :hip,)` (the closed option key set at :129-143 must learn `:backend`, `:device`,
# `:segments`).  `postprocess_plugin` walks the finished GraphPPL graph exactly as :272-326 does, recognises
# the chain  MvNormalMeanCovariance ← typeof(*) ← x[t-1]  /  MvNormalMeanCovariance(y[t]) ← typeof(*) ← x[t]
# (node types from GraphPPL.fform, interface names from GraphPPL.getname(edge), constants from
# GraphPPL.value), and fills an `Engine`; `new_observation!` maps to `set_data!`, `obtain_marginal` to a
# Rocket `of(...)` observable over `marginals(e)`, `score(…, BetheFreeEnergy, …)` to `of(free_energy(e)...)`.
# Anything it does not recognise falls through to the stock ReactiveMP plugin.

end # module
