# HIPInferencePlugin.jl — the GraphPPL plugin that puts librxhip behind `infer(...)`.
#
# A sibling of `ReactiveMPInferencePlugin` (src/model/plugins/reactivemp_inference.jl:207-326): same plugin type, same
# place in the plugin collection (src/inference/batch.jl:177-182, src/inference/streaming.jl:613-618), same objects handed
# back to the drivers (`GraphVariableRef`s whose variables answer `new_observation!`, `get_stream_of_marginals`, … and a
# `score(model, BetheFreeEnergy, checks)` observable) — but instead of one ReactiveMP object per node it walks the finished
# graph ONCE, fills the struct-of-arrays tables of `rxhip_graph_desc` (include/rxhip.h) and calls `rxhip_create`.
# Graphs without a device schedule (RXHIP_ERR_UNSUPPORTED) fall back to the stock plugin, node for node.
#
# To be `include`d inside `module RxInfer` after reactivemp_inference.jl and reactivemp_free_energy.jl (it uses their
# names: GraphVariableRef, ReactiveMPExtraVariableKey, InitMarExtraKey, BetheFreeEnergy, …) and after RxHip.jl.
# It was written against the reference sources and the C header and has NOT been executed: the build image has no Julia
# (SURVEY.md §0 F3).  What the walk produces is pinned another way: `dump_graph` below writes the same tables as JSON,
# rxinfer.jl_amd/rxhip/graph.py reads that format (`from_dump`), and tests/golden/graph_dumps/ holds the dumps of the four
# reference test models, which the GPU tests lower, run and compare with the reference's golden free energies.
#
# Integration (three hunks, see INTEGRATION.md §3):
#   batch.jl:181      RxInfer.ReactiveMPInferencePlugin(_options)  ->  RxInfer.inference_plugin(_options)
#   batch.jl:188      modelplugins + ReactiveMPFreeEnergyPlugin(fe_objective)  ->  … + free_energy_plugin(_options, fe_objective)
#   reactivemp_inference.jl:129-137   available_options gains :backend, :device, :segments, :chains

import GraphPPL
import ReactiveMP
import Rocket
using LinearAlgebra

# ---- options ------------------------------------------------------------------------------------------------------
"""Options of the HIP backend: `options = (backend = :hip, device = 0, segments = 0)`; everything else as today."""
struct HIPInferenceOptions{O}
    device::Int32
    segments::Int32
    fallback::O          # the ReactiveMPInferenceOptions the stock plugin would have received
end

struct HIPInferencePlugin{O}
    options::HIPInferenceOptions{O}
end
getoptions(plugin::HIPInferencePlugin) = plugin.options

"""Replaces the constructor call at src/inference/batch.jl:181 / streaming.jl:617."""
function inference_plugin(options::NamedTuple)
    if get(options, :backend, :reactivemp) === :hip
        rest = Base.structdiff(options, NamedTuple{(:backend, :device, :segments)})
        return HIPInferencePlugin(HIPInferenceOptions(Int32(get(options, :device, -1)), Int32(get(options, :segments, 0)),
                                                      convert(ReactiveMPInferenceOptions, rest)))
    end
    return ReactiveMPInferencePlugin(convert(ReactiveMPInferenceOptions, options))
end

GraphPPL.plugin_type(::HIPInferencePlugin) = GraphPPL.FactorAndVariableNodesPlugin()

# nothing to record per node while the `@model` body runs (reactivemp_inference.jl:228-270 stores per-node dependencies
# and stream postprocessors, neither of which exists on a static device schedule)
GraphPPL.preprocess_plugin(::HIPInferencePlugin, model::GraphPPL.Model, context::GraphPPL.Context, label::GraphPPL.NodeLabel,
                           nodedata::GraphPPL.NodeData, options::GraphPPL.NodeCreationOptions) = (label, nodedata)

# ---- node vocabulary: GraphPPL.fform -> RXHIP_NODE_* and the interface order of include/rxhip.h --------------------
const RXHIP_VARKIND_RANDOM, RXHIP_VARKIND_DATA, RXHIP_VARKIND_CONST = Int32(0), Int32(1), Int32(2)
const RXHIP_INIT_NONE, RXHIP_INIT_NORMAL, RXHIP_INIT_GAMMA, RXHIP_INIT_DIRICHLET, RXHIP_INIT_MVNORMAL, RXHIP_INIT_WISHART =
    Int32(0), Int32(1), Int32(2), Int32(3), Int32(4), Int32(5)

hip_node(::Type) = nothing
hip_node(::Type{<:ReactiveMP.MvNormalMeanCovariance}) = (Int32(1), "MvNormalMeanCovariance", (:out, :μ, :Σ))
hip_node(::typeof(*)) = (Int32(2), "*", (:out, :A, :in))
hip_node(::Type{<:ReactiveMP.NormalMeanVariance}) = (Int32(3), "NormalMeanVariance", (:out, :μ, :v))
hip_node(::Type{<:ReactiveMP.NormalMeanPrecision}) = (Int32(4), "NormalMeanPrecision", (:out, :μ, :τ))
hip_node(::Type{<:ReactiveMP.GammaShapeRate}) = (Int32(5), "GammaShapeRate", (:out, :α, :β))
hip_node(::Type{<:ReactiveMP.Dirichlet}) = (Int32(6), "Dirichlet", (:out, :a))
hip_node(::Type{<:ReactiveMP.Beta}) = (Int32(7), "Beta", (:out, :a, :b))
hip_node(::Type{<:ReactiveMP.Categorical}) = (Int32(8), "Categorical", (:out, :p))
hip_node(::Type{<:ReactiveMP.Bernoulli}) = (Int32(9), "Bernoulli", (:out, :p))
hip_node(::Type{<:ReactiveMP.NormalMixture}) = (Int32(10), "NormalMixture", (:out, :switch, :m, :p))   # m, p: indexed edges
hip_node(::Type{<:ReactiveMP.GCV}) = (Int32(11), "GCV", (:y, :x, :z, :κ, :ω))
hip_node(::Type{<:ReactiveMP.Wishart}) = (Int32(12), "Wishart", (:out, :ν, :S))
hip_node(::typeof(+)) = (Int32(13), "+", (:out, :in1, :in2))
hip_node(::Type{<:ReactiveMP.MvNormalMeanPrecision}) = (Int32(14), "MvNormalMeanPrecision", (:out, :μ, :Λ))
hip_node(::Type{<:ReactiveMP.GammaShapeScale}) = (Int32(15), "GammaShapeScale", (:out, :α, :θ))
hip_node(f::Function) = nothing

# row-major flattening of a constant's value (Julia arrays are column-major)
flat_value(x::Real) = (Float64[x], 1, 1)
flat_value(x::AbstractVector{<:Real}) = (Vector{Float64}(x), length(x), 1)
flat_value(x::AbstractMatrix{<:Real}) = (vec(collect(transpose(Matrix{Float64}(x)))), size(x, 1), size(x, 2))
flat_value(x::ReactiveMP.PointMass) = flat_value(ReactiveMP.mean(x))
flat_value(x) = nothing   # a constant the device path cannot represent -> unsupported graph

# `@initialization q(x) = …` marginals (InitMarExtraKey, src/model/plugins/initialization_plugin.jl:201-202)
init_params(q) = nothing
init_params(q::ReactiveMP.UnivariateNormalDistributionsFamily) = (RXHIP_INIT_NORMAL, Float64[ReactiveMP.mean(q), ReactiveMP.var(q)])
init_params(q::ReactiveMP.GammaDistributionsFamily) = (RXHIP_INIT_GAMMA, Float64[ReactiveMP.shape(q), ReactiveMP.rate(q)])
init_params(q::ReactiveMP.Dirichlet) = (RXHIP_INIT_DIRICHLET, Vector{Float64}(ReactiveMP.probvec(q)))   # concentration vector
init_params(q::ReactiveMP.Beta) = (RXHIP_INIT_DIRICHLET, Float64[ReactiveMP.params(q)...])
init_params(q::ReactiveMP.MultivariateNormalDistributionsFamily) =
    (RXHIP_INIT_MVNORMAL, vcat(Vector{Float64}(ReactiveMP.mean(q)), vec(collect(transpose(Matrix{Float64}(ReactiveMP.cov(q)))))))
init_params(q::ReactiveMP.Wishart) = let (ν, S) = ReactiveMP.params(q)
    (RXHIP_INIT_WISHART, vcat(Float64(ν), vec(collect(transpose(Matrix{Float64}(S))))))
end

# ---- the tables (owning Julia vectors; GraphDescC points into them) -----------------------------------------------
mutable struct GraphTables
    var_kind::Vector{Int32}
    var_rows::Vector{Int32}
    var_cols::Vector{Int32}
    var_const::Vector{Int64}
    var_init_family::Vector{Int32}
    var_init::Vector{Int64}
    var_name::Vector{String}              # for dump_graph / error messages only
    var_index::Vector{Any}
    factor_type::Vector{Int32}
    factor_name::Vector{String}
    factor_iface_ptr::Vector{Int64}       # CSR offsets, 0-based
    factor_iface::Vector{Int64}           # variable ids, 0-based
    factor_iface_names::Vector{String}
    factor_cluster::Vector{Int32}         # per entry of factor_iface: the cluster of q the interface belongs to WITHIN its node (0-based)
    const_pool::Vector{Float64}
    gh_points::Int32
    id_of::Dict{GraphPPL.NodeLabel, Int64}
    GraphTables() = new(Int32[], Int32[], Int32[], Int64[], Int32[], Int64[], String[], Any[], Int32[], String[], Int64[0], Int64[],
                        String[], Int32[], Float64[], Int32(0), Dict{GraphPPL.NodeLabel, Int64}())
end

struct UnsupportedGraph <: Exception
    why::String
end

"""
    build_tables(model) -> GraphTables

The walk of `postprocess_plugin(::ReactiveMPInferencePlugin, model)` (reactivemp_inference.jl:272-326) with table rows
instead of ReactiveMP objects: `variable_nodes` first (kind, shape, constant value, `@initialization` marginal), then
`factor_nodes` (`GraphPPL.fform`, `GraphPPL.neighbors` + `getname(edge)` in the node's interface order).
Throws `UnsupportedGraph` for anything the ABI has no code for.
"""
function build_tables(model::GraphPPL.Model)
    t = GraphTables()
    GraphPPL.variable_nodes(model) do label, nodedata
        props = GraphPPL.getproperties(nodedata)::GraphPPL.VariableNodeProperties
        id = Int64(length(t.var_kind))
        t.id_of[label] = id
        push!(t.var_name, string(GraphPPL.getname(props)))
        push!(t.var_index, GraphPPL.index(props))
        rows, cols, coff = Int32(0), Int32(1), Int64(-1)   # random / data: the shape is fixed below from the node they hang on
        if GraphPPL.is_constant(props)
            fv = flat_value(GraphPPL.value(props))
            fv === nothing && throw(UnsupportedGraph("constant $(GraphPPL.getname(props)) of type $(typeof(GraphPPL.value(props)))"))
            vals, r, c = fv
            rows, cols, coff = Int32(r), Int32(c), Int64(length(t.const_pool))
            append!(t.const_pool, vals)
            push!(t.var_kind, RXHIP_VARKIND_CONST)
        elseif GraphPPL.is_data(props)
            push!(t.var_kind, RXHIP_VARKIND_DATA)
        else
            # the same sanity checks the stock plugin performs (reactivemp_inference.jl:279-286)
            GraphPPL.degree(model, label) !== 0 || error(lazy"Unused random variable has been found $(label).")
            GraphPPL.degree(model, label) !== 1 ||
                error(lazy"Half-edge has been found: $(label). To terminate half-edges 'Uninformative' node can be used.")
            push!(t.var_kind, RXHIP_VARKIND_RANDOM)
        end
        push!(t.var_rows, rows); push!(t.var_cols, cols); push!(t.var_const, coff)
        fam, ioff = RXHIP_INIT_NONE, Int64(-1)
        if GraphPPL.hasextra(nodedata, InitMarExtraKey)
            ip = init_params(GraphPPL.getextra(nodedata, InitMarExtraKey))
            ip === nothing && throw(UnsupportedGraph("@initialization marginal of $(GraphPPL.getname(props))"))
            fam, ioff = ip[1], Int64(length(t.const_pool))
            append!(t.const_pool, ip[2])
        end
        GraphPPL.hasextra(nodedata, InitMsgExtraKey) && throw(UnsupportedGraph("@initialization of a message (μ(x) = …)"))
        push!(t.var_init_family, fam); push!(t.var_init, ioff)
    end
    GraphPPL.factor_nodes(model) do label, nodedata
        props = GraphPPL.getproperties(nodedata)::GraphPPL.FactorNodeProperties
        spec = hip_node(GraphPPL.fform(props))
        spec === nothing && throw(UnsupportedGraph("node $(GraphPPL.fform(props))"))
        code, name, order = spec
        # (interface name, index within an indexed interface such as m[k] of NormalMixture) -> variable id
        found = Dict{Tuple{Symbol, Int}, Int64}()
        # the factorisation of q around this node, exactly what the stock plugin hands to `factornode(fform, interfaces, factorization)`
        # (reactivemp_inference.jl:499-506): a tuple of clusters of NEIGHBOUR indices, ((1, 2), (3,)) = q(out, μ) q(Σ)
        factorization = GraphPPL.getextra(nodedata, GraphPPL.VariationalConstraintsFactorizationIndicesKey)
        cluster_of_neighbour = Dict{Int, Int32}()
        for (c, cluster) in enumerate(factorization), n in cluster
            cluster_of_neighbour[Int(n)] = Int32(c - 1)
        end
        cluster_of = Dict{Tuple{Symbol, Int}, Int32}()
        for (n, (vlabel, edge, _)) in enumerate(GraphPPL.neighbors(props))
            k = something(edge.index, 1)
            found[(GraphPPL.getname(edge), k)] = t.id_of[vlabel]
            haskey(cluster_of_neighbour, n) || throw(UnsupportedGraph("node $(name): interface $(GraphPPL.getname(edge)) is in no factorisation cluster"))
            cluster_of[(GraphPPL.getname(edge), k)] = cluster_of_neighbour[n]
        end
        for iname in order
            k = 1
            haskey(found, (iname, 1)) || throw(UnsupportedGraph("node $(name) without interface $(iname)"))
            while haskey(found, (iname, k))     # m[1..K], p[1..K] of NormalMixture are K consecutive entries
                push!(t.factor_iface, found[(iname, k)])
                push!(t.factor_cluster, cluster_of[(iname, k)])
                push!(t.factor_iface_names, k == 1 && !haskey(found, (iname, 2)) ? string(iname) : string(iname, "[", k, "]"))
                k += 1
            end
        end
        push!(t.factor_iface_ptr, Int64(length(t.factor_iface)))
        push!(t.factor_type, code); push!(t.factor_name, name)
        meta = GraphPPL.getextra(nodedata, GraphPPL.MetaExtraKey, nothing)
        if code == Int32(11) && meta !== nothing        # GCVMetadata(GaussHermiteCubature(n)), hgf_tests.jl:37-40
            t.gh_points = Int32(length(ReactiveMP.getweights(ReactiveMP.get_approximation(meta))))
        elseif meta !== nothing
            throw(UnsupportedGraph("@meta on node $(name)"))   # user meta redirects rule dispatch (inference_tests.jl:2049-2066)
        end
        # (the clusters go into rxhip_graph_desc.factor_cluster; every lowering pass and the executor's compiler hold them against the one
        #  factorisation per node type their schedule implements and answer a mismatch with RXHIP_ERR_UNSUPPORTED -> UnsupportedGraph -> stock plugin)
    end
    fix_shapes!(t)
    return t
end

# random / data variables take their dimension from a constant they share a node with: the mean of a Gaussian node has
# the length of the node's output, the output of `A * x` has A's row count, `x` its column count.
function fix_shapes!(t::GraphTables)
    changed = true
    while changed
        changed = false
        for f in 1:length(t.factor_type)
            ids = t.factor_iface[(t.factor_iface_ptr[f] + 1):t.factor_iface_ptr[f + 1]] .+ 1
            code = t.factor_type[f]
            want = Dict{Int, Int32}()
            if code in (Int32(1), Int32(3))                    # Gaussian nodes: out, μ have the (co)variance's row count
                r = maximum(t.var_rows[ids])
                r > 0 && (want[ids[1]] = r; want[ids[2]] = r)
            elseif code == Int32(2)                             # out = A * in
                t.var_rows[ids[2]] > 0 && (want[ids[1]] = t.var_rows[ids[2]]; want[ids[3]] = t.var_cols[ids[2]])
            elseif code == Int32(13)
                r = maximum(t.var_rows[ids]); r > 0 && (for i in ids; want[i] = r; end)
            elseif code == Int32(10)                            # NormalMixture: out as m[1]
                r = t.var_rows[ids[3]]; r > 0 && (want[ids[1]] = r)
            else                                                # scalar families
                for i in ids; want[i] = max(t.var_rows[i], Int32(1)); end
            end
            for (i, r) in want
                if t.var_rows[i] == 0
                    t.var_rows[i] = r; changed = true
                end
            end
        end
    end
    for i in eachindex(t.var_rows)
        t.var_rows[i] == 0 && (t.var_rows[i] = Int32(1))
    end
end

"""JSON dump of the tables: the exchange format `rxhip.graph.from_dump` reads (tests/golden/graph_dumps/*.json.gz)."""
function dump_graph(io::IO, t::GraphTables; n_replicas::Integer = 1, n_observations::Integer = 0)
    esc(s) = replace(s, "\\" => "\\\\", "\"" => "\\\"")
    print(io, "{\"format\":\"rxhip-graph-1\",\"n_replicas\":", n_replicas, ",\"n_observations\":", n_observations, ",\"gh_points\":", t.gh_points,
          ",\"variables\":[")
    kinds = ("random", "data", "constant")
    for i in eachindex(t.var_kind)
        i > 1 && print(io, ",")
        print(io, "{\"name\":\"", esc(t.var_name[i]), "\",\"kind\":\"", kinds[t.var_kind[i] + 1], "\",\"rows\":", t.var_rows[i], ",\"cols\":", t.var_cols[i])
        if t.var_const[i] >= 0
            n = t.var_rows[i] * t.var_cols[i]
            print(io, ",\"value\":[", join(repr.(t.const_pool[(t.var_const[i] + 1):(t.var_const[i] + n)]), ","), "]")
        end
        if t.var_init_family[i] != RXHIP_INIT_NONE
            fam = ("", "normal", "gamma", "dirichlet", "mvnormal", "wishart")[t.var_init_family[i] + 1]
            stop = minimum(vcat([o for o in vcat(t.var_const, t.var_init) if o > t.var_init[i]], length(t.const_pool)))
            print(io, ",\"init\":{\"family\":\"", fam, "\",\"params\":[", join(repr.(t.const_pool[(t.var_init[i] + 1):stop]), ","), "]}")
        end
        print(io, "}")
    end
    print(io, "],\"factors\":[")
    for f in eachindex(t.factor_type)
        f > 1 && print(io, ",")
        lo, hi = t.factor_iface_ptr[f] + 1, t.factor_iface_ptr[f + 1]
        print(io, "{\"type\":\"", esc(t.factor_name[f]), "\",\"interfaces\":[",
              join(("[\"" * esc(t.factor_iface_names[k]) * "\"," * string(t.factor_iface[k]) * "]" for k in lo:hi), ","), "],\"clusters\":[",
              join((string(t.factor_cluster[k]) for k in lo:hi), ","), "]}")
    end
    print(io, "]}")
end

# ---- engine-backed variables: what `getvardict` / the drivers see ---------------------------------------------------
mutable struct HIPGraphEngine
    engine::RxHip.Engine
    tables::GraphTables
    family::Symbol                       # :lgssm, :lgssm_noise, :drift, :mixture, :mvmixture, :hgf, :tree (the node-array executor: any acyclic Gaussian graph)
    data_ids::Vector{Int64}              # data variables in the order rxhip_set_data expects (time order)
    data_slot::Dict{Int64, Int}          # variable id -> position in `staging`
    staging::Vector{Float64}
    received::Int
    want_free_energy::Bool
    marginals::Dict{Int64, Any}          # variable id -> RecentSubject of Marginal
    free_energy::Any                     # RecentSubject{Float64} or nothing
    state_ids::Vector{Int64}             # x[t] in time order (state-space families)
    width::Int                           # doubles per observation
    component_ids::Any                   # mixtures: (m = ids, p = ids, s = id, beta = Bool)
    input_slot::Dict{Int64, Int}         # data inputs u[t] (`A * x[t-1] + B_u * u[t]`): variable id -> time index (1-based)
    input_staging::Vector{Float64}       # [t][du], zero where a transition has no input
    du::Int
    predictions::Dict{Int64, Any}        # data variable id -> RecentSubject of its prediction, created on first request
    masked::Bool                         # the engine takes `missing` observations (rebuilt on the first one, see `fire!`)
    options::Any                         # HIPInferenceOptions (segments, device) for that rebuild
    tree::Any                            # family :tree: RxHip.tree_layout (data offsets, random / precision variable ids), else nothing
end

"""`randomvar` stand-in: a marginal stream the engine pushes into after every sweep."""
struct HIPRandomVariable <: ReactiveMP.AbstractVariable
    id::Int64
    stream::Rocket.RecentSubjectInstance
end
"""`datavar` stand-in: `new_observation!` stages the value; the sweep fires when every observation of the iteration arrived
(the reference's cascade runs inside the same call, src/inference/batch.jl:405-407)."""
struct HIPDataVariable <: ReactiveMP.AbstractVariable
    id::Int64
    graph::Base.RefValue{Union{Nothing, HIPGraphEngine}}
end
struct HIPConstVariable{V} <: ReactiveMP.AbstractVariable
    value::V
end

ReactiveMP.israndom(::HIPRandomVariable) = true
ReactiveMP.isdata(::HIPRandomVariable) = false
ReactiveMP.isconst(::HIPRandomVariable) = false
ReactiveMP.israndom(::HIPDataVariable) = false
ReactiveMP.isdata(::HIPDataVariable) = true
ReactiveMP.isconst(::HIPDataVariable) = false
ReactiveMP.israndom(::HIPConstVariable) = false
ReactiveMP.isdata(::HIPConstVariable) = false
ReactiveMP.isconst(::HIPConstVariable) = true

ReactiveMP.get_stream_of_marginals(v::HIPRandomVariable) = v.stream
# `obtain_prediction` (reactivemp_inference.jl:619-624): the message toward a data variable, MvN_y(:out) — formed on the device
# by rxhip_get_predictions after the sweep, for the state-space families (any d, dy ≤ 64)
function ReactiveMP.get_stream_of_predictions(v::HIPDataVariable)
    g = v.graph[]
    g === nothing && error("HIP data variable is not attached to an engine")
    g.family === :lgssm || error("the HIP backend predicts the data variables of state-space graphs only; use options = (backend = :reactivemp,)")
    return get!(() -> Rocket.RecentSubject(ReactiveMP.Marginal), g.predictions, v.id)
end
ReactiveMP.get_stream_of_predictions(v::HIPRandomVariable) = v.stream   # a random variable's prediction is its marginal

function ReactiveMP.new_observation!(v::HIPDataVariable, value)
    g = v.graph[]
    g === nothing && error("HIP data variable is not attached to an engine")
    if haskey(g.input_slot, v.id)   # a control input u[t]: staged like an observation, never missing
        ismissing(value) && error("a data input of the transition cannot be missing")
        vals = value isa Real ? (Float64(value),) : value
        length(vals) == g.du || error("input of length $(length(vals)), expected $(g.du)")
        t = g.input_slot[v.id]
        @inbounds for (k, x) in enumerate(vals)
            g.input_staging[(t - 1) * g.du + k] = x
        end
        g.received += 1
        g.received == length(g.data_ids) + length(g.input_slot) && fire!(g)
        return nothing
    end
    slot = g.data_slot[v.id]
    if g.family === :tree   # ragged observations: every data variable at its own offset of the staging vector
        ismissing(value) && error("the node-array executor takes no missing observations; use options = (backend = :reactivemp,)")
        vals = value isa Real ? (Float64(value),) : value
        o = g.tree.data_offsets[slot]
        @inbounds for (k, x) in enumerate(vals)
            g.staging[o + k] = x
        end
        g.received += 1
        g.received == length(g.data_ids) && fire!(g)
        return nothing
    end
    if ismissing(value)   # `missing` = NaN on the device: no message from this observation branch (static.md:98-123)
        g.family === :lgssm || error("the HIP backend takes missing observations in state-space graphs only")
        vals = ntuple(_ -> NaN, g.width)
    else
        vals = value isa Real ? (Float64(value),) : value
        any(ismissing, vals) && (vals = ntuple(_ -> NaN, g.width))   # a partly missing vector is missing
    end
    length(vals) == g.width || error("observation of length $(length(vals)), expected $(g.width)")
    @inbounds for (k, x) in enumerate(vals)
        g.staging[(slot - 1) * g.width + k] = x
    end
    g.received += 1
    g.received == length(g.data_ids) + length(g.input_slot) && fire!(g)
    return nothing
end

"""One iteration of the loop at batch.jl:391-430: push the staged data, run ONE sweep / VMP iteration, publish."""
function fire!(g::HIPGraphEngine)
    g.received = 0
    if !g.masked && (g.family === :lgssm || g.family === :tree) && any(isnan, g.staging)
        # the first `missing` observation: the time-parallel tables assume every step observed (state-space family) and the executor keeps its data leaves in
        # the form their reader wants (tree family) — rebuild the engine with rxhip_graph_desc.allow_missing: the masked schedule / data leaves in precision
        # form, where a NaN is the zero message.  RXHIP_ERR_UNSUPPORTED surfaces as the error it is.
        RxHip.destroy!(g.engine)
        g.engine = RxHip.create_from_tables(g.tables; segments = g.options.segments, device = g.options.device, allow_missing = true)
        g.family === :tree && RxHip.tree_continue!(g.engine)
        g.masked = true
    end
    e = g.engine
    if g.family === :tree   # one sweep (+ one update of every q(W)) per call: the loop of batch.jl:391-430 re-pushes the data every iteration
        RxHip.tree_set_data!(e, g.data_ids, g.staging)
        RxHip.run!(e; iterations = 1, free_energy = g.want_free_energy)
        publish_marginals!(g)
        g.want_free_energy && g.free_energy !== nothing && Rocket.next!(g.free_energy, RxHip.free_energy(e)[end])
        return nothing
    end
    isempty(g.input_slot) || RxHip.set_inputs!(e, g.input_staging)
    GC.@preserve g begin
        RxHip.check(e, ccall((:rxhip_set_data, RxHip.librxhip), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}, Csize_t, Int32),
                             e.handle, RxHip.RXHIP_VAR_Y, g.staging, length(g.staging), RxHip.RXHIP_LAYOUT_CHAIN_TIME))
    end
    if g.family === :mixture || g.family === :mvmixture
        RxHip.vmp_iteration!(e; free_energy = g.want_free_energy)   # continues from the marginals of the previous iteration
    elseif g.family === :lgssm_noise
        RxHip.run!(e; iterations = 1, free_energy = g.want_free_energy)   # one sweep + one Wishart update, from the q(W) of the previous call
    else
        RxHip.run!(e; iterations = 1, free_energy = g.want_free_energy)   # trees: one sweep is the fixed point
    end
    publish_marginals!(g)
    if !isempty(g.predictions)
        mean, cov = RxHip.predictions(e)                         # dy × T, dy × dy × T
        for (t, id) in enumerate(g.data_ids)
            haskey(g.predictions, id) || continue
            Rocket.next!(g.predictions[id], as_marginal(ReactiveMP.MvNormalMeanCovariance(mean[:, t, 1], Symmetric(cov[:, :, t, 1]))))
        end
    end
    if g.want_free_energy && g.free_energy !== nothing
        Rocket.next!(g.free_energy, RxHip.free_energy(e)[end])
    end
    return nothing
end

as_marginal(d) = ReactiveMP.Marginal(d, false, false, nothing)

function publish_marginals!(g::HIPGraphEngine)
    e = g.engine
    if g.family === :tree
        ms, Vs = RxHip.tree_marginals(e, g.tree.state_ids, g.tree.state_dims)
        for (k, id) in enumerate(g.tree.state_ids)
            haskey(g.marginals, id) || continue
            q = (id in g.tree.scalar_ids) ? ReactiveMP.NormalMeanVariance(ms[k][1], Vs[k][1, 1]) :
                ReactiveMP.MvNormalMeanCovariance(ms[k], Symmetric(Vs[k]))
            Rocket.next!(g.marginals[id], as_marginal(q))
        end
        for id in g.tree.precision_ids
            haskey(g.marginals, id) || continue
            ν, V = RxHip.tree_precision(e, id, Int(g.tables.var_rows[id + 1]))
            Rocket.next!(g.marginals[id], as_marginal(g.tree.gamma[id] ? ReactiveMP.GammaShapeRate(ν / 2, 1 / (2 * V[1, 1])) : ReactiveMP.Wishart(ν, Symmetric(V))))
        end
        # a mixture layer's discrete side: q(z) of every switch, q(s) of every probability vector (the Bernoulli / Beta spelling: the FIRST component is `true`)
        for id in g.tree.switch_ids
            haskey(g.marginals, id) || continue
            p = RxHip.tree_discrete(e, id)
            Rocket.next!(g.marginals[id], as_marginal((id in g.tree.bernoulli) ? ReactiveMP.Bernoulli(p[1]) : ReactiveMP.Categorical(p)))
        end
        for id in g.tree.probability_ids
            haskey(g.marginals, id) || continue
            a = RxHip.tree_discrete(e, id)
            Rocket.next!(g.marginals[id], as_marginal((id in g.tree.bernoulli) ? ReactiveMP.Beta(a[1], a[2]) : ReactiveMP.Dirichlet(a)))
        end
        return nothing
    end
    if g.family === :lgssm || g.family === :lgssm_noise
        mean, cov = RxHip.marginals(e)                          # d × T, d × d × T
        for (t, id) in enumerate(g.state_ids)
            haskey(g.marginals, id) || continue
            Rocket.next!(g.marginals[id], as_marginal(ReactiveMP.MvNormalMeanCovariance(mean[:, t, 1], Symmetric(cov[:, :, t, 1]))))
        end
        if g.family === :lgssm_noise && haskey(g.marginals, g.component_ids.p[1])
            ν, V = RxHip.noise_posterior(e)                     # q(W) = Wishart(ν, V); scalar observations with a Gamma prior: Gamma(ν/2, 1/(2V))
            q = g.component_ids.beta ? ReactiveMP.GammaShapeRate(ν[1] / 2, 1 / (2 * V[1, 1, 1])) : ReactiveMP.Wishart(ν[1], Symmetric(V[:, :, 1]))
            Rocket.next!(g.marginals[g.component_ids.p[1]], as_marginal(q))
        end
    elseif g.family === :drift
        mean, var = RxHip.marginals(e)
        for (t, id) in enumerate(g.state_ids)
            haskey(g.marginals, id) && Rocket.next!(g.marginals[id], as_marginal(ReactiveMP.NormalMeanVariance(mean[1, t, 1], var[1, 1, t, 1])))
        end
    elseif g.family === :mixture
        K = length(g.component_ids.m)
        st = RxHip.last_mixture_state(e, K)                     # (mean m, var m, shape p, rate p, α s)[k]
        for k in 1:K
            haskey(g.marginals, g.component_ids.m[k]) &&
                Rocket.next!(g.marginals[g.component_ids.m[k]], as_marginal(ReactiveMP.NormalMeanVariance(st[k], st[K + k])))
            haskey(g.marginals, g.component_ids.p[k]) &&
                Rocket.next!(g.marginals[g.component_ids.p[k]], as_marginal(ReactiveMP.GammaShapeRate(st[2K + k], st[3K + k])))
        end
        if g.component_ids.s >= 0 && haskey(g.marginals, g.component_ids.s)
            α = st[(4K + 1):(5K)]
            q = g.component_ids.beta ? ReactiveMP.Beta(α[1], α[2]) : ReactiveMP.Dirichlet(α)
            Rocket.next!(g.marginals[g.component_ids.s], as_marginal(q))
        end
    elseif g.family === :mvmixture
        K, d = length(g.component_ids.m), g.width
        sz = 2 + d + 2 * d * d
        st = RxHip.last_mixture_state(e, K; d = d)
        for k in 1:K
            b = st[((k - 1) * sz + 1):(k * sz)]
            μ, Σ = b[1:d], collect(transpose(reshape(b[(d + 1):(d + d * d)], d, d)))
            ν, V = b[d + d * d + 1], collect(transpose(reshape(b[(d + d * d + 2):(d + 2 * d * d + 1)], d, d)))
            haskey(g.marginals, g.component_ids.m[k]) &&
                Rocket.next!(g.marginals[g.component_ids.m[k]], as_marginal(ReactiveMP.MvNormalMeanCovariance(μ, Symmetric(Σ))))
            haskey(g.marginals, g.component_ids.p[k]) && Rocket.next!(g.marginals[g.component_ids.p[k]], as_marginal(ReactiveMP.Wishart(ν, Symmetric(V))))
        end
        if g.component_ids.s >= 0 && haskey(g.marginals, g.component_ids.s)
            Rocket.next!(g.marginals[g.component_ids.s], as_marginal(ReactiveMP.Dirichlet([st[k * sz] for k in 1:K])))
        end
        # q(z[i]) stays on the device (N × K responsibilities; `rxhip_gmm_get_responsibilities` with materialize = 1)
    end
end

"""variable ids of m[k], p[k] (or w[k]) and s of a mixture graph, from the first observation node's interfaces
(out, switch, m[1..K], p[1..K]) and the switch's Categorical / Bernoulli node; iid Normal(mean, precision): (out, μ, τ)."""
function component_ids(t::GraphTables)
    isobs(c) = t.factor_type[c] == Int32(10) || t.factor_type[c] == Int32(4) ||
               (t.factor_type[c] == Int32(14) && t.var_kind[t.factor_iface[t.factor_iface_ptr[c] + 1] + 1] == Int32(1))   # MvNormal(μ = m, Λ = P) on data
    f = findfirst(isobs, eachindex(t.factor_type))
    f === nothing && return (m = Int64[], p = Int64[], s = Int64(-1), beta = false)
    ids = t.factor_iface[(t.factor_iface_ptr[f] + 1):t.factor_iface_ptr[f + 1]]
    t.factor_type[f] != Int32(10) && return (m = [ids[2]], p = [ids[3]], s = Int64(-1), beta = false)
    K = (length(ids) - 2) ÷ 2
    s, beta = Int64(-1), false
    for c in eachindex(t.factor_type)
        if t.factor_type[c] in (Int32(8), Int32(9)) && t.factor_iface[t.factor_iface_ptr[c] + 1] == ids[2]
            s, beta = t.factor_iface[t.factor_iface_ptr[c] + 2], t.factor_type[c] == Int32(9)
        end
    end
    return (m = ids[3:(2 + K)], p = ids[(3 + K):(2 + 2K)], s = s, beta = beta)
end

# ---- postprocess: graph -> tables -> rxhip_create, or the stock plugin ------------------------------------------------
const HIPEngineKey = GraphPPL.NodeDataExtraKey{:rxhip_engine, Any}()

function GraphPPL.postprocess_plugin(plugin::HIPInferencePlugin, model::GraphPPL.Model)
    tables = try
        build_tables(model)
    catch err
        err isa UnsupportedGraph || rethrow()
        getoptions(plugin).fallback.warn && @warn "HIP backend: $(err.why) has no device schedule; using ReactiveMP"
        return GraphPPL.postprocess_plugin(ReactiveMPInferencePlugin(getoptions(plugin).fallback), model)
    end
    engine = try
        RxHip.create_from_tables(tables; segments = getoptions(plugin).segments, device = getoptions(plugin).device)
    catch err
        (err isa RxHip.RxHipError && err.status == RxHip.RXHIP_ERR_UNSUPPORTED) || rethrow()
        getoptions(plugin).fallback.warn && @warn "HIP backend: $(err.msg); using ReactiveMP"
        return GraphPPL.postprocess_plugin(ReactiveMPInferencePlugin(getoptions(plugin).fallback), model)
    end
    tree = RxHip.tree_info(engine) === nothing ? nothing : RxHip.tree_layout(tables)   # the executor took the graph: its own layout
    lowered = tree === nothing ? RxHip.lowered_layout(tables) : (family = :tree, data_ids = tree.data_ids, state_ids = tree.state_ids)       # family, data / state variable ids in time order, observation width
    if lowered.family === :hgf
        # one-step graphs driven by `@autoupdates` belong to the streaming driver's event loop (streaming.jl:349-407); the
        # device runs ALL observations in one call, which is reached through RxHip.HgfEngine / RxHip.run_filter!, not here
        RxHip.destroy!(engine)
        return GraphPPL.postprocess_plugin(ReactiveMPInferencePlugin(getoptions(plugin).fallback), model)
    end
    gref = Ref{Union{Nothing, HIPGraphEngine}}(nothing)
    marginals = Dict{Int64, Any}()
    # one variable object per graph variable, stored under the key the drivers read (`getvariable`, :589-592)
    GraphPPL.variable_nodes(model) do label, nodedata
        props = GraphPPL.getproperties(nodedata)::GraphPPL.VariableNodeProperties
        id = tables.id_of[label]
        if GraphPPL.is_random(props)
            subject = Rocket.RecentSubject(ReactiveMP.Marginal)
            marginals[id] = subject
            GraphPPL.setextra!(nodedata, ReactiveMPExtraVariableKey, HIPRandomVariable(id, subject))
        elseif GraphPPL.is_data(props)
            GraphPPL.setextra!(nodedata, ReactiveMPExtraVariableKey, HIPDataVariable(id, gref))
        else
            GraphPPL.setextra!(nodedata, ReactiveMPExtraVariableKey, HIPConstVariable(GraphPPL.value(props)))
        end
    end
    width = Int(tables.var_rows[lowered.data_ids[1] + 1])
    # the chain with an unknown noise precision: `p` carries the variable id of W, `beta` whether its prior was spelled as a Gamma
    comps = lowered.family === :tree ? (m = Int64[], p = Int64[], s = Int64(-1), beta = false) : lowered.family === :lgssm_noise ? (m = Int64[], p = [lowered.precision_id], s = Int64(-1), beta = lowered.gamma) : component_ids(tables)
    lowered.family === :lgssm_noise && RxHip.noise_continue!(engine)
    lowered.family === :tree && RxHip.tree_continue!(engine)   # one VMP iteration per `fire!`: go on from the current q(W), not from @initialization
    g = HIPGraphEngine(engine, tables, lowered.family, lowered.data_ids, Dict(id => k for (k, id) in enumerate(lowered.data_ids)),
                       zeros(Float64, tree === nothing ? width * length(lowered.data_ids) : tree.data_total), 0, false, marginals, nothing, lowered.state_ids, width,
                       comps,
                       Dict{Int64, Int}(id => t for (t, id) in enumerate(get(lowered, :input_ids, Int64[])) if id >= 0),
                       zeros(Float64, get(lowered, :du, 0) * length(lowered.data_ids)), get(lowered, :du, 0),
                       Dict{Int64, Any}(), false, getoptions(plugin), tree)
    gref[] = g
    GraphPPL.setextra!(GraphPPL.getcontext(model), HIPEngineKey, g)   # one handle per model; found again by `score`
    return nothing
end

# ---- free energy ------------------------------------------------------------------------------------------------
"""Replaces `ReactiveMPFreeEnergyPlugin` (reactivemp_free_energy.jl:22-82) when the HIP plugin owns the graph."""
struct HIPFreeEnergyPlugin{O}
    objective::O
end
free_energy_plugin(options::NamedTuple, objective) =
    get(options, :backend, :reactivemp) === :hip ? HIPFreeEnergyPlugin(objective) : ReactiveMPFreeEnergyPlugin(objective)

GraphPPL.plugin_type(::HIPFreeEnergyPlugin) = GraphPPL.FactorAndVariableNodesPlugin()
GraphPPL.preprocess_plugin(::HIPFreeEnergyPlugin, model::GraphPPL.Model, context::GraphPPL.Context, label::GraphPPL.NodeLabel,
                           nodedata::GraphPPL.NodeData, options::GraphPPL.NodeCreationOptions) = (label, nodedata)

function GraphPPL.postprocess_plugin(plugin::HIPFreeEnergyPlugin, model::GraphPPL.Model)
    ctx = GraphPPL.getcontext(model)
    if GraphPPL.hasextra(ctx, HIPEngineKey)       # the HIP plugin took the graph: the device evaluates the Bethe sum
        g = GraphPPL.getextra(ctx, HIPEngineKey)::HIPGraphEngine
        g.want_free_energy = true
        g.free_energy = Rocket.RecentSubject(Float64)
        return nothing
    end
    # fallen back to ReactiveMP: wire the stock per-node score streams
    return GraphPPL.postprocess_plugin(ReactiveMPFreeEnergyPlugin(plugin.objective), model)
end

"""`score(model, BetheFreeEnergy{T}, checks)` (reactivemp_free_energy.jl:84-126): the observable the ScoreActor subscribes to."""
function hip_score(model::ProbabilisticModel, objective::BetheFreeEnergy{T}, diagnostic_checks) where {T}
    ctx = GraphPPL.getcontext(getmodel(model))
    GraphPPL.hasextra(ctx, HIPEngineKey) || return nothing
    g = GraphPPL.getextra(ctx, HIPEngineKey)::HIPGraphEngine
    # NaN / Inf checks happen on the device (RXHIP_ERR_NONFINITE_FE -> exception in `fire!`), cf. src/score/diagnostics.jl:19-51
    return g.free_energy |> Rocket.map(T, identity)
end
# score(model, objective, checks) = something(hip_score(model, objective, checks), <stock method body>)   — one-line guard at :84
