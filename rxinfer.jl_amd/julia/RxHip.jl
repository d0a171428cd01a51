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
