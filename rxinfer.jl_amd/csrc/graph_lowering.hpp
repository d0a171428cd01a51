// graph_lowering.hpp — host-side lowering of a generic factor-graph descriptor (include/rxhip.h
// rxhip_graph_desc) to the structured schedule descriptors.  Replaces, for the graph shapes that have a
// device schedule, the per-node object construction of
// GraphPPL.postprocess_plugin(::ReactiveMPInferencePlugin, model) (src/model/plugins/reactivemp_inference.jl:272-326):
// instead of one ReactiveMP object per variable / factor, one pass over the SoA tables.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <map>
#include <unordered_map>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rxhip.h"

namespace rxhip_lower {

inline std::string& last_error() {
    static thread_local std::string e;
    return e;
}
// the largest relative asymmetry max|W − W′| / max|W| among the constant parameters the current lowering call accepted as "symmetric within round-off"
// and replaced by their symmetric part (spd_inverse_checked below; rxhip_lowering_asymmetry() hands it to the caller); the ABI entry points reset it
inline double& last_asymmetry() {
    static thread_local double a = 0.0;
    return a;
}
inline rxhip_status unsupported(const std::string& why) {
    last_error() = why;
    return RXHIP_ERR_UNSUPPORTED;
}
inline rxhip_status badarg(const std::string& why) {
    last_error() = why;
    return RXHIP_ERR_BADARG;
}

struct Lgssm {
    int d = 0, dy = 0;
    long long T = 0;
    int ptt = 0;
    int deterministic = 0;  // 1: every transition is `x[t] ~ x[t-1] + c` (typeof(+) with a constant, no state noise)
    std::vector<double> A, B, P, Q, m0, V0, c;  // A, B, P, Q: [n_models] matrices
    std::vector<long long> state_var, data_var;
    int n_models = 1;
    std::vector<int> step_model;  // [T] when n_models > 1: the constants of time index t (per-step A[t], P[t], B[t], Q[t])
    std::vector<double> cx, cy;   // known inputs [T][d] / [T][dy] (`A * x[t-1] + c`, `B * x[t] + d`); empty: none
    // inputs that are data: `A * x[t-1] + B_u * u[t]` with u[t] a data variable
    int du = 0;                       // dimension of u (0: none)
    std::vector<double> Bu;           // [d][du] (identity when u is added without a `*` node)
    std::vector<long long> input_var; // [T] variable id of u[t] (-1: this transition has no input)
};

// interface k of factor f / number of interfaces (3-wide table or CSR)
inline long long iface(const rxhip_graph_desc* g, long long f, int k) {
    return g->factor_iface_ptr ? g->factor_iface[g->factor_iface_ptr[f] + k] : g->factor_iface[f * 3 + k];
}
inline int n_iface(const rxhip_graph_desc* g, long long f) {
    return g->factor_iface_ptr ? (int)(g->factor_iface_ptr[f + 1] - g->factor_iface_ptr[f]) : 3;
}
inline const char* node_name(int type) {
    switch (type) {
    case RXHIP_NODE_MVNORMAL_MEAN_COV: return "MvNormalMeanCovariance";
    case RXHIP_NODE_MULTIPLY: return "*";
    case RXHIP_NODE_NORMAL_MEAN_VARIANCE: return "NormalMeanVariance";
    case RXHIP_NODE_NORMAL_MEAN_PRECISION: return "NormalMeanPrecision";
    case RXHIP_NODE_GAMMA_SHAPE_RATE: return "GammaShapeRate";
    case RXHIP_NODE_DIRICHLET: return "Dirichlet";
    case RXHIP_NODE_BETA: return "Beta";
    case RXHIP_NODE_CATEGORICAL: return "Categorical";
    case RXHIP_NODE_BERNOULLI: return "Bernoulli";
    case RXHIP_NODE_NORMAL_MIXTURE: return "NormalMixture";
    case RXHIP_NODE_GCV: return "GCV";
    case RXHIP_NODE_WISHART: return "Wishart";
    case RXHIP_NODE_ADD: return "+";
    case RXHIP_NODE_MVNORMAL_MEAN_PRECISION: return "MvNormalMeanPrecision";
    case RXHIP_NODE_GAMMA_SHAPE_SCALE: return "GammaShapeScale";
    default: return "?";
    }
}
// The factorisation of q around every node (rxhip_graph_desc.factor_cluster = the reference's VariationalConstraintsFactorizationIndicesKey,
// src/model/plugins/reactivemp_inference.jl:499-506) against the ONE factorisation per node type the schedules implement (include/rxhip.h).
// `gaussian_meanfield` (the node-array executor only): receives, per factor, 1 where a Gaussian node is asked for q(out) q(μ) — without it such a node
// is refused like every other mismatch, with the node named.  Call after the interface tables have been validated.
inline rxhip_status check_factorisation(const rxhip_graph_desc* g, std::vector<char>* gaussian_meanfield = nullptr) {
    if (gaussian_meanfield) gaussian_meanfield->assign((size_t)g->n_factors, 0);
    if (!g->factor_cluster) return RXHIP_OK;
    for (long long f = 0; f < g->n_factors; ++f) {
        const int n = n_iface(g, f), t = g->factor_type[f];
        const long long base = g->factor_iface_ptr ? g->factor_iface_ptr[f] : 3 * f;
        auto rnd = [&](int k) { return g->var_kind[iface(g, f, k)] == RXHIP_VARKIND_RANDOM; };
        auto cl = [&](int k) { return g->factor_cluster[base + k]; };
        auto refuse = [&](const char* asked, const char* have) {
            return unsupported("factor " + std::to_string(f) + " (" + node_name(t) + "): the model's constraints ask for " + asked + ", the device schedule of this node implements " +
                               have + " (rxhip_graph_desc.factor_cluster)");
        };
        // which pairs of random interfaces the schedule keeps in one factor of q
        auto joint = [&](int a, int b) -> bool {
            switch (t) {
            case RXHIP_NODE_MVNORMAL_MEAN_COV: case RXHIP_NODE_NORMAL_MEAN_VARIANCE: case RXHIP_NODE_MVNORMAL_MEAN_PRECISION: case RXHIP_NODE_NORMAL_MEAN_PRECISION:
                return a < 2 && b < 2;
            case RXHIP_NODE_MULTIPLY: case RXHIP_NODE_ADD: return true;
            case RXHIP_NODE_GCV: return a < 2 && b < 2;
            default: return false;
            }
        };
        const bool gauss = t == RXHIP_NODE_MVNORMAL_MEAN_COV || t == RXHIP_NODE_NORMAL_MEAN_VARIANCE || t == RXHIP_NODE_MVNORMAL_MEAN_PRECISION || t == RXHIP_NODE_NORMAL_MEAN_PRECISION;
        for (int a = 0; a < n; ++a) {
            if (!rnd(a)) continue;
            for (int b = a + 1; b < n; ++b) {
                if (!rnd(b) || iface(g, f, a) == iface(g, f, b)) continue;
                const bool same = cl(a) == cl(b), want = joint(a, b);
                if (same == want) continue;
                if (gauss && a == 0 && b == 1 && !same) {   // q(out) q(μ)
                    if (!gaussian_meanfield) return refuse("q(out) q(μ) (mean-field between the Gaussian interfaces)", "the structured q(out, μ); only the node-array executor runs the mean-field form");
                    (*gaussian_meanfield)[(size_t)f] = 1;
                    continue;
                }
                if (gauss) return refuse("a joint factor of q over a Gaussian interface and the precision", "q(out, μ) q(precision)");
                if (t == RXHIP_NODE_MULTIPLY || t == RXHIP_NODE_ADD) return refuse("a factorised q around a deterministic node", "the joint over its random interfaces");
                if (t == RXHIP_NODE_GCV) return refuse(same ? "a joint factor with the volatility input z" : "q(y) q(x)", "q(y, x) q(z)");
                return refuse("a structured factor of q", "the mean-field factorisation (every random interface its own factor)");
            }
        }
    }
    return RXHIP_OK;
}
inline rxhip_status check_tables_only(const rxhip_graph_desc* g);
// tables, then the factorisation every pattern-matched family assumes (a Gaussian node under q(out) q(μ) is the executor's: refused here)
inline rxhip_status check_tables(const rxhip_graph_desc* g) {
    if (rxhip_status st = check_tables_only(g)) return st;
    return check_factorisation(g);
}
inline rxhip_status check_tables_only(const rxhip_graph_desc* g) {
    if (!g || g->n_variables <= 0 || g->n_factors <= 0 || !g->var_kind || !g->var_rows || !g->var_cols || !g->var_const ||
        !g->factor_type || !g->factor_iface || (g->n_const > 0 && !g->const_pool))
        return badarg("graph descriptor has null tables");
    if (g->factor_iface_ptr) {
        if (g->factor_iface_ptr[0] < 0) return badarg("factor_iface_ptr must start at a non-negative offset");
        for (long long f = 0; f < g->n_factors; ++f)
            if (g->factor_iface_ptr[f + 1] < g->factor_iface_ptr[f]) return badarg("factor_iface_ptr is not non-decreasing");
    }
    for (long long f = 0; f < g->n_factors; ++f) {
        const int n = n_iface(g, f);
        if (n <= 0) return badarg("factor without interfaces");
        for (int k = 0; k < n; ++k)
            if (iface(g, f, k) < 0 || iface(g, f, k) >= g->n_variables) return badarg("factor interface refers to an unknown variable");
    }
    for (long long v = 0; v < g->n_variables; ++v) {
        if (g->var_rows[v] <= 0 || g->var_cols[v] <= 0) return badarg("variable with a non-positive shape");
        if (g->var_kind[v] == RXHIP_VARKIND_CONST && g->var_const[v] >= 0 &&
            g->var_const[v] + (long long)g->var_rows[v] * g->var_cols[v] > g->n_const)
            return badarg("constant value lies outside the constant pool");
    }
    return RXHIP_OK;
}
inline bool has_node(const rxhip_graph_desc* g, int type) {
    for (long long f = 0; f < g->n_factors; ++f)
        if (g->factor_type[f] == type) return true;
    return false;
}
// scalar constant
inline bool const_scalar(const rxhip_graph_desc* g, long long v, double* out) {
    if (g->var_kind[v] != RXHIP_VARKIND_CONST || g->var_const[v] < 0 || g->var_rows[v] != 1 || g->var_cols[v] != 1) return false;
    if (g->var_const[v] + 1 > g->n_const) return false;
    *out = g->const_pool[g->var_const[v]];
    return true;
}
// `@initialization` marginal of a random variable: n parameters of the given family
inline bool init_params(const rxhip_graph_desc* g, long long v, int family, int n, const double** out) {
    if (!g->var_init_family || !g->var_init) return false;
    if (g->var_init_family[v] != family || g->var_init[v] < 0 || g->var_init[v] + n > g->n_const) return false;
    *out = g->const_pool + g->var_init[v];
    return true;
}

// value of a constant variable, with shape check
inline bool const_value(const rxhip_graph_desc* g, long long v, int rows, int cols, const double** out) {
    if (g->var_kind[v] != RXHIP_VARKIND_CONST || g->var_const[v] < 0) return false;
    if (g->var_rows[v] != rows || g->var_cols[v] != cols) return false;
    if (g->var_const[v] + (long long)rows * cols > g->n_const) return false;
    *out = g->const_pool + g->var_const[v];
    return true;
}
// two constant variables with equal values (both must carry a value inside the pool)
inline bool same_const(const rxhip_graph_desc* g, long long a, long long b) {
    if (a == b) return true;
    if (g->var_kind[a] != RXHIP_VARKIND_CONST || g->var_kind[b] != RXHIP_VARKIND_CONST) return false;
    if (g->var_rows[a] != g->var_rows[b] || g->var_cols[a] != g->var_cols[b]) return false;
    const long long n = (long long)g->var_rows[a] * g->var_cols[a];
    if (g->var_const[a] < 0 || g->var_const[b] < 0 || g->var_const[a] + n > g->n_const || g->var_const[b] + n > g->n_const) return false;
    return std::memcmp(g->const_pool + g->var_const[a], g->const_pool + g->var_const[b], (size_t)n * sizeof(double)) == 0;
}

// Recognise a linear Gaussian state-space chain.  A "Gaussian node" is MvNormalMeanCovariance(out, μ, Σ) or — for scalar
// chains — NormalMeanVariance(out, μ, v), Σ / v constant:
//     prior:        Gaussian(out = x_first, μ = const)
//     per state x:  Gaussian(out = y (data), μ = B·x)                 observation; B·x is either the output of a
//                   Gaussian(out = x_next (random), μ = A·x)          `*`(out, A const, in = x) node or x itself (B, A = I)
// (`y ~ MvNormal(μ = B * x, Σ = Q)`, `x ~ Normal(μ = x_prev, v = …)` and their mixtures); constants that differ from step to
// step (`A[t] * x[t-1]`, `Σ = P[t]`) are grouped into models by value (Lgssm::step_model).  Also the noise-free drift chain of test/models/statespace/ulgssm_tests.jl:8-15,
//     x[t] ~ x[t-1] + c        `+`(out = x_next, in1 = x, in2 = c const)  (or in1 const)
// whose every transition is such a node.  A `+` with a constant IN FRONT of a Gaussian mean (`A * x[t-1] + c`, `B * x[t] + d`,
// `x[t-1] + c` with state noise) is a known input of that time index (Lgssm::cx / cy).  Node order in the tables is irrelevant.
// `NormalMeanPrecision(μ, τ)` / `MvNormalMeanPrecision(μ, Λ)` nodes with a CONSTANT precision are the same factors as their
// covariance-parametrised forms (test/inference/prediction_tests.jl:197-213 spells a whole random-walk chain this way): a
// private copy of the tables gets a new constant variable W⁻¹ per such node and the node type of the covariance form, and the
// chain lowering below never sees a precision.  Nodes with a random precision (the iid Gaussian×Gamma family) are left alone.
// Inverse of a constant precision (the covariance the kernels run on): symmetric within ROUND-OFF or refused — the lowering
// must not repair an input error by symmetrising afterwards — then through the Cholesky factor (positive definite or refused).
// "Round-off" is scaled to what a host produces: a precision computed as inv(Σ) carries an asymmetry of ≈ eps·cond(Σ)·max|W|, so the
// bound is 1e-8·max|W| (cond up to ≈ 10⁷; the reference places no symmetry requirement on MvNormalMeanPrecision arguments at all);
// below it the symmetric part ½(W + W′) is what gets inverted — and the measured asymmetry is kept for rxhip_lowering_asymmetry(), so that a caller who
// wants the tight bound can have it.  0: ok, 1: not symmetric, 2: not positive definite.
inline int spd_inverse_checked(int d, const double* W, std::vector<double>& inv) {
    double amax = 0.0, asym = 0.0;
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
            amax = std::max(amax, std::fabs(W[(size_t)i * d + j]));
            asym = std::max(asym, std::fabs(W[(size_t)i * d + j] - W[(size_t)j * d + i]));
        }
    if (!(asym <= 1e-8 * amax)) return 1;   // also catches NaN
    if (asym > 0.0) last_asymmetry() = std::max(last_asymmetry(), asym / amax);   // accepted and symmetrised: the caller can ask how far
    std::vector<double> L((size_t)d * d, 0.0), Li((size_t)d * d, 0.0);
    for (int j = 0; j < d; ++j) {
        double s = W[(size_t)j * d + j];
        for (int k = 0; k < j; ++k) s -= L[(size_t)j * d + k] * L[(size_t)j * d + k];
        if (!(s > 0.0)) return 2;
        const double ljj = std::sqrt(s);
        L[(size_t)j * d + j] = ljj;
        for (int i = j + 1; i < d; ++i) {
            double t = 0.5 * (W[(size_t)i * d + j] + W[(size_t)j * d + i]);
            for (int k = 0; k < j; ++k) t -= L[(size_t)i * d + k] * L[(size_t)j * d + k];
            L[(size_t)i * d + j] = t / ljj;
        }
    }
    for (int i = 0; i < d; ++i) {   // Li = L⁻¹, row by row
        Li[(size_t)i * d + i] = 1.0;
        for (int k = 0; k < i; ++k) {
            const double l = L[(size_t)i * d + k];
            for (int j = 0; j <= k; ++j) Li[(size_t)i * d + j] -= l * Li[(size_t)k * d + j];
        }
        const double r = 1.0 / L[(size_t)i * d + i];
        for (int j = 0; j <= i; ++j) Li[(size_t)i * d + j] *= r;
    }
    inv.assign((size_t)d * d, 0.0);   // W⁻¹ = Li'Li, symmetric by construction
    for (int i = 0; i < d; ++i)
        for (int j = 0; j <= i; ++j) {
            double t = 0.0;
            for (int k = i; k < d; ++k) t += Li[(size_t)k * d + i] * Li[(size_t)k * d + j];
            inv[(size_t)i * d + j] = inv[(size_t)j * d + i] = t;
        }
    return 0;
}
struct NormalisedGraph {
    rxhip_graph_desc g;
    std::vector<int32_t> var_kind, var_rows, var_cols, factor_type, var_init_family;
    std::vector<int64_t> var_const, factor_iface, var_init;
    std::vector<double> pool;
};
inline rxhip_status normalise_precision_nodes(const rxhip_graph_desc* g0, NormalisedGraph& N, bool& changed) {
    changed = false;
    for (long long f = 0; f < g0->n_factors && !changed; ++f) {
        const int t = g0->factor_type[f];
        if ((t == RXHIP_NODE_NORMAL_MEAN_PRECISION || t == RXHIP_NODE_MVNORMAL_MEAN_PRECISION) && n_iface(g0, f) == 3 &&
            g0->var_kind[iface(g0, f, 2)] == RXHIP_VARKIND_CONST)
            changed = true;
    }
    if (!changed) return RXHIP_OK;
    if (g0->factor_iface_ptr) { changed = false; return RXHIP_OK; }  // CSR tables: not a 3-interface Gaussian graph
    const long long NV = g0->n_variables, NF = g0->n_factors;
    N.var_kind.assign(g0->var_kind, g0->var_kind + NV);
    N.var_rows.assign(g0->var_rows, g0->var_rows + NV);
    N.var_cols.assign(g0->var_cols, g0->var_cols + NV);
    N.var_const.assign(g0->var_const, g0->var_const + NV);
    N.factor_type.assign(g0->factor_type, g0->factor_type + NF);
    N.factor_iface.assign(g0->factor_iface, g0->factor_iface + 3 * NF);
    N.pool.assign(g0->const_pool, g0->const_pool + g0->n_const);
    if (g0->var_init_family) N.var_init_family.assign(g0->var_init_family, g0->var_init_family + NV);
    if (g0->var_init) N.var_init.assign(g0->var_init, g0->var_init + NV);
    for (long long f = 0; f < NF; ++f) {
        const int t = g0->factor_type[f];
        if (t != RXHIP_NODE_NORMAL_MEAN_PRECISION && t != RXHIP_NODE_MVNORMAL_MEAN_PRECISION) continue;
        const long long w = iface(g0, f, 2);
        if (g0->var_kind[w] != RXHIP_VARKIND_CONST) continue;
        const int d = g0->var_rows[w];
        const double* W;
        if (g0->var_cols[w] != d || !const_value(g0, w, d, d, &W)) return badarg("precision of a Gaussian node is not a square constant");
        std::vector<double> inv;
        switch (spd_inverse_checked(d, W, inv)) {
            case 1: return badarg("precision of a Gaussian node is not symmetric");
            case 2: return badarg("precision of a Gaussian node is not positive definite");
            default: break;
        }
        const long long nv = (long long)N.var_kind.size();
        N.var_kind.push_back(RXHIP_VARKIND_CONST);
        N.var_rows.push_back(d);
        N.var_cols.push_back(d);
        N.var_const.push_back((long long)N.pool.size());
        if (!N.var_init_family.empty()) N.var_init_family.push_back(RXHIP_INIT_NONE);
        if (!N.var_init.empty()) N.var_init.push_back(-1);
        N.pool.insert(N.pool.end(), inv.begin(), inv.end());
        N.factor_iface[(size_t)3 * f + 2] = nv;
        N.factor_type[f] = t == RXHIP_NODE_NORMAL_MEAN_PRECISION ? RXHIP_NODE_NORMAL_MEAN_VARIANCE : RXHIP_NODE_MVNORMAL_MEAN_COV;
    }
    N.g = *g0;
    N.g.n_variables = (long long)N.var_kind.size();
    N.g.var_kind = N.var_kind.data(); N.g.var_rows = N.var_rows.data(); N.g.var_cols = N.var_cols.data(); N.g.var_const = N.var_const.data();
    N.g.factor_type = N.factor_type.data(); N.g.factor_iface = N.factor_iface.data();
    N.g.const_pool = N.pool.data(); N.g.n_const = (long long)N.pool.size();
    N.g.var_init_family = N.var_init_family.empty() ? nullptr : N.var_init_family.data();
    N.g.var_init = N.var_init.empty() ? nullptr : N.var_init.data();
    return RXHIP_OK;
}

inline rxhip_status lower_lgssm(const rxhip_graph_desc* g, Lgssm& L) {
    if (rxhip_status st = check_tables(g)) return st;
    NormalisedGraph norm;
    {
        bool changed = false;
        if (rxhip_status st = normalise_precision_nodes(g, norm, changed)) return st;
        if (changed) g = &norm.g;
    }
    const long long NV = g->n_variables, NF = g->n_factors;
    for (long long f = 0; f < NF; ++f)
        if (n_iface(g, f) != 3) return unsupported("node with " + std::to_string(n_iface(g, f)) + " interfaces in a state-space chain");
    // `*` node producing each (anonymous) variable; Gaussian node by its μ variable; `+` nodes by their random input
    std::vector<long long> mul_of_out(NV, -1), add_of_out(NV, -1), writer(NV, -1), mul_data(NV, -1);
    std::vector<long long> add_input(NF, -1);  // `+` node -> its data-side input variable (u[t] itself or the output of B_u * u[t])
    std::vector<std::vector<long long>> mul_of_in(NV), gauss_by_mu(NV), add_of_in(NV);
    long long prior = -1;
    bool scalar_nodes = false;
    std::vector<long long> pending_adds;
    auto writes = [&](long long v, long long f) -> bool {  // one factor "produces" a variable: rejects merges and cycles early
        if (writer[v] >= 0) return false;
        writer[v] = f;
        return true;
    };
    for (long long f = 0; f < NF; ++f) {
        const long long io[3] = {iface(g, f, 0), iface(g, f, 1), iface(g, f, 2)};
        const int t = g->factor_type[f];
        if (t == RXHIP_NODE_MULTIPLY) {
            if (g->var_kind[io[1]] != RXHIP_VARKIND_CONST) return unsupported("`*` node with a non-constant matrix (no device schedule)");
            if (g->var_kind[io[2]] == RXHIP_VARKIND_DATA && g->var_kind[io[0]] == RXHIP_VARKIND_RANDOM) {
                if (!writes(io[0], f)) return unsupported("variable produced by two nodes");
                mul_data[io[0]] = f;  // `B_u * u[t]`: a deterministic function of data, consumed by a `+` below
                continue;
            }
            if (g->var_kind[io[2]] != RXHIP_VARKIND_RANDOM || g->var_kind[io[0]] != RXHIP_VARKIND_RANDOM)
                return unsupported("`*` node must connect two random variables");
            if (mul_of_out[io[0]] >= 0 || !writes(io[0], f)) return unsupported("variable produced by two nodes");
            mul_of_out[io[0]] = f;
            mul_of_in[io[2]].push_back(f);
        } else if (t == RXHIP_NODE_MVNORMAL_MEAN_COV || t == RXHIP_NODE_NORMAL_MEAN_VARIANCE) {
            if (t == RXHIP_NODE_NORMAL_MEAN_VARIANCE) scalar_nodes = true;
            if (g->var_kind[io[2]] != RXHIP_VARKIND_CONST) return unsupported("Gaussian node with a non-constant covariance");
            if (g->var_kind[io[0]] == RXHIP_VARKIND_CONST) return unsupported("Gaussian node with a constant output");
            if (g->var_kind[io[0]] == RXHIP_VARKIND_RANDOM && !writes(io[0], f)) return unsupported("random variable that is the output of two nodes: not a chain");
            if (g->var_kind[io[1]] == RXHIP_VARKIND_CONST) {
                if (prior >= 0) return unsupported("more than one prior node: not a single chain");
                prior = f;
            } else if (g->var_kind[io[1]] == RXHIP_VARKIND_RANDOM) {
                gauss_by_mu[io[1]].push_back(f);  // an anonymous `A * x` feeds exactly one; a state may feed its observation and its transition
            } else
                return unsupported("Gaussian node whose mean is a data variable");
        } else if (t == RXHIP_NODE_ADD) {
            bool c1 = g->var_kind[io[1]] == RXHIP_VARKIND_CONST, c2 = g->var_kind[io[2]] == RXHIP_VARKIND_CONST;
            if (c1 && c2) return unsupported("`+` node with two constant inputs");
            if (!c1 && !c2) {
                // no constant: one side must be data — u[t] itself or (later resolved) the output of `B_u * u[t]`
                const bool d1 = g->var_kind[io[1]] == RXHIP_VARKIND_DATA, d2 = g->var_kind[io[2]] == RXHIP_VARKIND_DATA;
                if (d1 == d2) { pending_adds.push_back(f); continue; }  // decided once every `*` node has been seen
                add_input[f] = d1 ? io[1] : io[2];
                c1 = d1; c2 = d2;
            }
            const long long xin = c1 ? io[2] : io[1];
            if (g->var_kind[xin] != RXHIP_VARKIND_RANDOM || g->var_kind[io[0]] != RXHIP_VARKIND_RANDOM) return unsupported("`+` node must connect two random variables");
            if (!writes(io[0], f)) return unsupported("random variable that is the output of two nodes: not a chain");
            add_of_in[xin].push_back(f);
            add_of_out[io[0]] = f;
        } else
            return unsupported("node type " + std::to_string(t) + " has no device schedule");
    }
    for (long long f : pending_adds) {  // `+` of two random variables: one of them is `B_u * u[t]`
        const long long io[3] = {iface(g, f, 0), iface(g, f, 1), iface(g, f, 2)};
        const bool m1 = mul_data[io[1]] >= 0, m2 = mul_data[io[2]] >= 0;
        if (m1 == m2) return unsupported("`+` node needs a constant or a data-driven input");
        add_input[f] = m1 ? io[1] : io[2];
        const long long xin = m1 ? io[2] : io[1];
        if (g->var_kind[io[0]] != RXHIP_VARKIND_RANDOM) return unsupported("`+` node must connect two random variables");
        if (!writes(io[0], f)) return unsupported("random variable that is the output of two nodes: not a chain");
        add_of_in[xin].push_back(f);
        add_of_out[io[0]] = f;
    }
    if (prior < 0) return unsupported("no prior node (Gaussian node with constant mean)");
    // an anonymous `A * x` feeds exactly one consumer: a Gaussian mean, or a `+` with a constant (known input) in front of one
    for (long long v = 0; v < NV; ++v) {
        if (mul_of_out[v] >= 0 && gauss_by_mu[v].size() + add_of_in[v].size() != 1)
            return unsupported(gauss_by_mu[v].empty() && add_of_in[v].empty() ? "`*` node whose output feeds no Gaussian mean" : "mean variable shared by two nodes");
        if (add_of_out[v] >= 0 && gauss_by_mu[v].size() > 1) return unsupported("mean variable shared by two Gaussian nodes");
    }
    long long x = iface(g, prior, 0);
    if (g->var_kind[x] != RXHIP_VARKIND_RANDOM) return unsupported("prior node on a non-random variable");
    const int d = g->var_rows[x];
    if (scalar_nodes && d != 1) return unsupported("NormalMeanVariance node in a vector-valued chain");
    const double *m0, *V0;
    if (!const_value(g, iface(g, prior, 1), d, 1, &m0) || !const_value(g, iface(g, prior, 2), d, d, &V0))
        return badarg("prior constants have the wrong shape");
    // a branch out of state x: the Gaussian node it feeds and the constant matrix in between (-1: identity)
    struct Branch { long long gauss, matrix, offset; };  // offset: the constant of a `+` node in front of the Gaussian mean (-1: none)
    long long vA = -2, vB = -2, vC = -1;  // -2: not seen yet, -1: identity
    constexpr long long NONE = -3;                           // no transition into this time index (t = 1 of a chain whose prior sits on x[1])
    std::vector<long long> sA, sP, sB, sQ, sCx, sCy;         // per time index: the constant variables of its transition / observation
    int n_noisy = 0, n_det = 0;
    long long used_factors = 1;
    bool first = true;
    std::vector<char> visited(NV, 0);
    L = Lgssm();
    for (long long guard = 0; guard <= NV; ++guard) {
        if (visited[x]) return unsupported("state visited twice: the graph has a cycle, not a chain");
        visited[x] = 1;
        std::vector<Branch> br;
        if (mul_of_out[x] >= 0) return unsupported("the output of a `*` node used as a state");
        // the `+` node's constant — or, for a data-driven input, the node itself encoded as −(10 + f)
        auto add_const = [&](long long f) -> long long {
            if (add_input[f] >= 0) return -(10 + f);
            return g->var_kind[iface(g, f, 1)] == RXHIP_VARKIND_CONST ? iface(g, f, 1) : iface(g, f, 2);
        };
        long long add = -1;  // a `+` whose output is the next STATE: the noise-free drift transition
        for (long long f : mul_of_in[x]) {
            const long long v = iface(g, f, 0);
            if (!add_of_in[v].empty()) {  // A * x + c
                const long long a = add_of_in[v][0], w = iface(g, a, 0);
                if (gauss_by_mu[w].empty()) return unsupported("`A * x + c` that feeds no Gaussian mean");
                br.push_back({gauss_by_mu[w][0], iface(g, f, 1), add_const(a)});
                used_factors += 1 + ((add_input[a] >= 0 && g->var_kind[add_input[a]] != RXHIP_VARKIND_DATA) ? 1 : 0);  // `+` (and `B_u * u`)
            } else
                br.push_back({gauss_by_mu[v][0], iface(g, f, 1), -1});
        }
        for (long long gn : gauss_by_mu[x]) br.push_back({gn, -1, -1});  // Gaussian nodes reading the state directly (A, B = I)
        bool onward = false;  // a transition out of x other than through a `+`
        for (const Branch& b : br) onward = onward || g->var_kind[iface(g, b.gauss, 0)] == RXHIP_VARKIND_RANDOM;
        for (long long a : add_of_in[x]) {
            // x + c is either an anonymous mean (a known input of a transition / an observation with A or B = I) or the NEXT
            // STATE of a noise-free drift chain, which is observed through a Gaussian node of its own
            const long long w = iface(g, a, 0);
            const bool continues = !mul_of_in[w].empty() || !add_of_in[w].empty();
            const bool one_gauss = gauss_by_mu[w].size() == 1;
            const bool to_random = one_gauss && g->var_kind[iface(g, gauss_by_mu[w][0], 0)] == RXHIP_VARKIND_RANDOM;
            const bool to_data = one_gauss && g->var_kind[iface(g, gauss_by_mu[w][0], 0)] == RXHIP_VARKIND_DATA;
            if (!continues && (to_random || (to_data && (onward || n_noisy > 0)))) {
                br.push_back({gauss_by_mu[w][0], -1, add_const(a)});
                used_factors += 1 + ((add_input[a] >= 0 && g->var_kind[add_input[a]] != RXHIP_VARKIND_DATA) ? 1 : 0);
                onward = onward || to_random;
            } else {
                if (add >= 0) return unsupported("state with two `+` transitions: not a chain");
                add = a;
            }
        }
        long long obs = -1, obs_m = -1, obs_c = -1, tr = -1, tr_m = -1, tr_c = -1;
        for (const Branch& b : br) {
            const long long target = iface(g, b.gauss, 0);
            if (g->var_kind[target] == RXHIP_VARKIND_DATA) {
                if (obs >= 0) return unsupported("state with two observation branches");
                obs = b.gauss; obs_m = b.matrix; obs_c = b.offset;
            } else {
                if (tr >= 0) return unsupported("state with two transitions: not a chain");
                tr = b.gauss; tr_m = b.matrix; tr_c = b.offset;
            }
            used_factors += b.matrix >= 0 ? 2 : 1;
        }
        if (add >= 0 && tr >= 0) return unsupported("state with two transitions: not a chain");
        if (obs >= 0) {
            const long long q = iface(g, obs, 2);
            if (vB == -2) vB = obs_m;
            sB.push_back(obs_m);
            sQ.push_back(q);
            sCy.push_back(obs_c);
            L.state_var.push_back(x);
            L.data_var.push_back(iface(g, obs, 0));
        } else if (first && (tr >= 0 || add >= 0)) {
            L.ptt = 1;  // x0 ~ prior without an observation: test/models/statespace/mlgssm_test.jl:9-17
        } else
            return unsupported("state variable without an observation");
        first = false;
        if (tr >= 0) {
            const long long pv = iface(g, tr, 2);
            if (vA == -2) vA = tr_m;
            sA.resize(L.state_var.size() + 1, NONE);  // the transition INTO the next observed state
            sP.resize(L.state_var.size() + 1, NONE);
            sA.back() = tr_m;
            sP.back() = pv;
            sCx.resize(L.state_var.size() + 1, -1);
            sCx.back() = tr_c;
            ++n_noisy;
            x = iface(g, tr, 0);
        } else if (add >= 0) {
            if (add_input[add] >= 0) return unsupported("noise-free transition with a data input");
            const long long cv = add_const(add);
            if (vC < 0) vC = cv;
            else if (!same_const(g, vC, cv)) return unsupported("time-varying drift");
            ++n_det;
            used_factors += 1;
            x = iface(g, add, 0);
        } else
            break;
        if (g->var_rows[x] != d) return unsupported("state dimension changes along the chain");
    }
    if (used_factors != NF) return unsupported("graph has factors outside the state-space chain");
    if (n_det > 0 && n_noisy > 0) return unsupported("chain mixing noisy and noise-free (`+`) transitions");
    L.T = (long long)L.state_var.size();
    if (L.T <= 0 || vB == -2) return unsupported("chain without observations");
    L.d = d;
    L.dy = g->var_rows[L.data_var[0]];
    const int dy = L.dy;
    L.m0.assign(m0, m0 + d);
    L.V0.assign(V0, V0 + (size_t)d * d);
    // the constants of every time index, grouped into models: first by variable id, then by value
    const long long T = L.T;
    sA.resize((size_t)T, NONE);
    sP.resize((size_t)T, NONE);
    for (long long t = 0; t < T; ++t)
        if (g->var_rows[L.data_var[t]] != dy) return unsupported("observation dimension changes along the chain");
    if (n_noisy > 0 && sA[0] == NONE) {  // no transition into the first state: its A, P are never used — take the next step's
        sA[0] = T > 1 ? sA[1] : -1;
        sP[0] = T > 1 ? sP[1] : NONE;
    }
    auto append = [&](std::vector<double>& dst, long long v, int r, int c, int fill) -> bool {  // fill: 1 identity, 0 zero
        if (v >= 0) {
            const double* q;
            if (!const_value(g, v, r, c, &q)) return false;
            dst.insert(dst.end(), q, q + (size_t)r * c);
        } else {
            const size_t o = dst.size();
            dst.resize(o + (size_t)r * c, 0.0);
            if (fill) for (int i = 0; i < (r < c ? r : c); ++i) dst[o + (size_t)i * c + i] = 1.0;
        }
        return true;
    };
    std::map<std::array<long long, 4>, int> by_id;
    std::unordered_map<std::string, int> by_value;
    std::vector<int> step((size_t)T);
    for (long long t = 0; t < T; ++t) {
        const std::array<long long, 4> ids = {sA[t], sP[t], sB[t], sQ[t]};
        if (t > 0) {  // the common case — GraphPPL makes a new constant variable per use, with equal values: compare with the last step
            const std::array<long long, 4> pv = {sA[t - 1], sP[t - 1], sB[t - 1], sQ[t - 1]};
            bool same = true;
            for (int k = 0; k < 4 && same; ++k) same = ids[k] == pv[k] || (ids[k] >= 0 && pv[k] >= 0 && same_const(g, ids[k], pv[k]));
            if (same) { step[t] = step[t - 1]; continue; }
        }
        auto it = by_id.find(ids);
        if (it != by_id.end()) { step[t] = it->second; continue; }
        if (sB[t] < 0 && dy != d) return unsupported("observation without a `*` node must have the state's dimension");
        std::vector<double> a, pm, b, q;
        // a chain of `+` nodes has A = I, P = 0; a single time step has no transition in the graph (A = P = I, never used)
        if (!append(a, sA[t] == NONE ? -1 : sA[t], d, d, 1)) return badarg("transition matrix has the wrong shape");
        if (!append(pm, sP[t] == NONE ? -1 : sP[t], d, d, n_det > 0 ? 0 : 1)) return badarg("state noise has the wrong shape");
        if (!append(b, sB[t], dy, d, 1)) return badarg("observation matrix has the wrong shape");
        if (sQ[t] < 0 || !append(q, sQ[t], dy, dy, 1)) return badarg("observation noise has the wrong shape");
        std::string key;
        for (const std::vector<double>* m : {&a, &pm, &b, &q}) key.append((const char*)m->data(), m->size() * sizeof(double));
        auto iv = by_value.find(key);
        int mdl;
        if (iv != by_value.end()) mdl = iv->second;
        else {
            mdl = (int)by_value.size();
            by_value.emplace(std::move(key), mdl);
            L.A.insert(L.A.end(), a.begin(), a.end());
            L.P.insert(L.P.end(), pm.begin(), pm.end());
            L.B.insert(L.B.end(), b.begin(), b.end());
            L.Q.insert(L.Q.end(), q.begin(), q.end());
        }
        by_id.emplace(ids, mdl);
        step[t] = mdl;
    }
    // known inputs of every time index
    sCx.resize((size_t)T, -1);
    bool any_cx = false, any_cy = false;
    for (long long t = 0; t < T; ++t) {
        any_cx = any_cx || sCx[t] >= 0;
        any_cy = any_cy || sCy[t] >= 0;
    }
    // data-driven inputs of the transitions: u[t] and B_u (one matrix for the whole chain)
    long long vBu = -2;  // -2: none seen, -1: identity (u added directly)
    for (long long t = 0; t < T; ++t) {
        if (sCy[t] <= -10) return unsupported("data-driven offset of an observation");
        if (sCx[t] > -10) continue;
        const long long f = -(sCx[t] + 10), v = add_input[f];
        long long uvar = v, bu = -1;
        if (g->var_kind[v] != RXHIP_VARKIND_DATA) {
            const long long mf = mul_data[v];
            uvar = iface(g, mf, 2);
            bu = iface(g, mf, 1);
        }
        if (vBu == -2) { vBu = bu; L.du = g->var_rows[uvar]; L.input_var.assign((size_t)T, -1); }
        else if ((vBu < 0) != (bu < 0) || (bu >= 0 && !same_const(g, vBu, bu))) return unsupported("input matrix B_u changes along the chain");
        if (g->var_rows[uvar] != L.du) return unsupported("input dimension changes along the chain");
        L.input_var[(size_t)t] = uvar;
        sCx[t] = -1;  // no constant part at this step
    }
    if (L.du > 0) {
        if (n_det > 0) return unsupported("noise-free `+` chain with data inputs");
        if (vBu >= 0) {
            const double* q;
            if (!const_value(g, vBu, d, L.du, &q)) return badarg("input matrix B_u has the wrong shape");
            L.Bu.assign(q, q + (size_t)d * L.du);
        } else {
            if (L.du != d) return unsupported("an input added without a `*` node must have the state's dimension");
            L.Bu.assign((size_t)d * d, 0.0);
            for (int i = 0; i < d; ++i) L.Bu[(size_t)i * d + i] = 1.0;
        }
        any_cx = false;
        for (long long t = 0; t < T; ++t) any_cx = any_cx || sCx[t] >= 0;
        L.cx.assign((size_t)T * d, 0.0);  // offsets are reserved whenever inputs exist
        L.cy.assign((size_t)T * dy, 0.0);
    }
    if (any_cx || any_cy) {
        if (n_det > 0) return unsupported("noise-free `+` chain with further offsets");
        if (L.cx.empty()) {
        L.cx.assign((size_t)T * d, 0.0);
        L.cy.assign((size_t)T * dy, 0.0);
        }
        for (long long t = 0; t < T; ++t) {
            const double* q;
            if (sCx[t] >= 0) {
                if (!const_value(g, sCx[t], d, 1, &q)) return badarg("state offset has the wrong shape");
                std::memcpy(&L.cx[(size_t)t * d], q, sizeof(double) * d);
            }
            if (sCy[t] >= 0) {
                if (!const_value(g, sCy[t], dy, 1, &q)) return badarg("observation offset has the wrong shape");
                std::memcpy(&L.cy[(size_t)t * dy], q, sizeof(double) * dy);
            }
        }
    }
    L.n_models = (int)by_value.size();
    if (L.n_models > 1) {
        if (n_det > 0) return unsupported("noise-free `+` chain with time-varying observation noise");
        L.step_model.swap(step);
    }
    L.c.assign(d, 0.0);
    if (n_det > 0) {
        const double* pc;
        if (!const_value(g, vC, d, 1, &pc)) return badarg("drift constant has the wrong shape");
        L.c.assign(pc, pc + d);
        L.deterministic = 1;
        if (d != 1 || vB >= 0) return unsupported("noise-free `+` chains have a device schedule for scalar states observed directly");
    }
    last_error().clear();
    return RXHIP_OK;
}

// ------------------------------------------------------------------------------------------------------------
// The state-space chain with an UNKNOWN observation-noise precision (rxhip_lgssm_noise_create): one Wishart node on a random W —
// or, for scalar observations, one Gamma node on a random τ (Gamma(a, b) = Wishart₁(ν = 2a, S = 1/(2b))) — and every observation node
//     y[t] ~ MvNormal(μ = B * x[t], Λ = W)          (`Normal(mean = …, precision = τ)` for dy = 1)
// precision-parametrised on that ONE variable.  The lowering takes the prior node out, gives the observation nodes a placeholder
// covariance (the engine never reads desc->Q) and hands the rest to lower_lgssm: what comes back must be a plain chain of one model.
struct LgssmNoise {
    Lgssm chain;
    long long w_var = -1;
    double nu0 = 0.0, init_nu = 0.0;
    std::vector<double> S0, init_V;
};
inline rxhip_status lower_lgssm_noise(const rxhip_graph_desc* g, LgssmNoise& L) {
    if (rxhip_status st = check_tables(g)) return st;
    if (g->factor_iface_ptr) return unsupported("CSR interface tables: not a three-interface Gaussian graph");
    const long long NV = g->n_variables, NF = g->n_factors;
    long long fw = -1;
    bool gamma_form = false;
    for (long long f = 0; f < NF; ++f) {
        const int t = g->factor_type[f];
        if (t == RXHIP_NODE_WISHART || t == RXHIP_NODE_GAMMA_SHAPE_RATE || t == RXHIP_NODE_GAMMA_SHAPE_SCALE) {
            if (fw >= 0) return unsupported("more than one precision prior in a state-space graph");
            fw = f;
            gamma_form = t != RXHIP_NODE_WISHART;
        }
    }
    if (fw < 0) return unsupported("no Wishart / Gamma prior node");
    const long long w = iface(g, fw, 0);
    if (g->var_kind[w] != RXHIP_VARKIND_RANDOM) return unsupported("precision prior on a non-random variable");
    const int dy = g->var_rows[w];
    if (gamma_form && dy != 1) return badarg("Gamma prior on a precision that is not a scalar");   // (random variables carry their dimension in var_rows only)
    // the prior's constants
    L = LgssmNoise();
    L.w_var = w;
    if (gamma_form) {
        double a, b;
        if (!const_scalar(g, iface(g, fw, 1), &a) || !const_scalar(g, iface(g, fw, 2), &b)) return unsupported("Gamma prior with non-constant parameters");
        if (g->factor_type[fw] == RXHIP_NODE_GAMMA_SHAPE_SCALE) b = 1.0 / b;
        if (!(a > 0.0) || !(b > 0.0)) return badarg("Gamma prior parameters must be positive");
        L.nu0 = 2.0 * a;
        L.S0.assign(1, 1.0 / (2.0 * b));
    } else {
        const double* S;
        if (!const_scalar(g, iface(g, fw, 1), &L.nu0) || !const_value(g, iface(g, fw, 2), dy, dy, &S)) return unsupported("Wishart prior with non-constant parameters");
        L.S0.assign(S, S + (size_t)dy * dy);
    }
    // the `@initialization` marginal q(W) (the reference refuses to start mean-field VMP without one)
    const double* q;
    if (init_params(g, w, RXHIP_INIT_WISHART, 1 + dy * dy, &q)) {
        L.init_nu = q[0];
        L.init_V.assign(q + 1, q + 1 + (size_t)dy * dy);
    } else if (dy == 1 && init_params(g, w, RXHIP_INIT_GAMMA, 2, &q)) {
        if (!(q[0] > 0.0) || !(q[1] > 0.0)) return badarg("@initialization marginal of the precision must have positive parameters");
        L.init_nu = 2.0 * q[0];
        L.init_V.assign(1, 1.0 / (2.0 * q[1]));
    } else
        return badarg("mean-field VMP needs an @initialization marginal for the observation precision");
    // private tables: the prior node removed, the observation nodes in covariance form on a placeholder constant
    NormalisedGraph N;
    N.var_kind.assign(g->var_kind, g->var_kind + NV);
    N.var_rows.assign(g->var_rows, g->var_rows + NV);
    N.var_cols.assign(g->var_cols, g->var_cols + NV);
    N.var_const.assign(g->var_const, g->var_const + NV);
    N.pool.assign(g->const_pool, g->const_pool + g->n_const);
    if (g->var_init_family) N.var_init_family.assign(g->var_init_family, g->var_init_family + NV);
    if (g->var_init) N.var_init.assign(g->var_init, g->var_init + NV);
    const long long placeholder = NV;
    N.var_kind.push_back(RXHIP_VARKIND_CONST);
    N.var_rows.push_back(dy);
    N.var_cols.push_back(dy);
    N.var_const.push_back((long long)N.pool.size());
    if (!N.var_init_family.empty()) N.var_init_family.push_back(RXHIP_INIT_NONE);
    if (!N.var_init.empty()) N.var_init.push_back(-1);
    for (int i = 0; i < dy; ++i)
        for (int j = 0; j < dy; ++j) N.pool.push_back(i == j ? 1.0 : 0.0);
    long long n_obs = 0;
    for (long long f = 0; f < NF; ++f) {
        if (f == fw) continue;
        int t = g->factor_type[f];
        long long io[3] = {iface(g, f, 0), iface(g, f, 1), iface(g, f, 2)};
        for (int k = 0; k < 2; ++k)
            if (io[k] == w) return unsupported("the precision variable is used as something else than a precision");
        if (io[2] == w) {
            if ((t != RXHIP_NODE_MVNORMAL_MEAN_PRECISION && t != RXHIP_NODE_NORMAL_MEAN_PRECISION) || g->var_kind[io[0]] != RXHIP_VARKIND_DATA)
                return unsupported("the random precision must be the Λ of observation nodes only");
            t = t == RXHIP_NODE_NORMAL_MEAN_PRECISION ? RXHIP_NODE_NORMAL_MEAN_VARIANCE : RXHIP_NODE_MVNORMAL_MEAN_COV;
            io[2] = placeholder;
            ++n_obs;
        }
        N.factor_type.push_back(t);
        N.factor_iface.insert(N.factor_iface.end(), io, io + 3);
    }
    if (n_obs == 0) return unsupported("precision prior that no observation node uses");
    N.g = *g;
    N.g.n_variables = (long long)N.var_kind.size();
    N.g.n_factors = NF - 1;
    N.g.var_kind = N.var_kind.data(); N.g.var_rows = N.var_rows.data(); N.g.var_cols = N.var_cols.data(); N.g.var_const = N.var_const.data();
    N.g.factor_type = N.factor_type.data(); N.g.factor_iface = N.factor_iface.data();
    N.g.const_pool = N.pool.data(); N.g.n_const = (long long)N.pool.size();
    N.g.var_init_family = N.var_init_family.empty() ? nullptr : N.var_init_family.data();
    N.g.var_init = N.var_init.empty() ? nullptr : N.var_init.data();
    if (rxhip_status st = lower_lgssm(&N.g, L.chain)) return st;
    const Lgssm& C = L.chain;
    if (C.T != n_obs) return unsupported("observation nodes with a constant covariance next to the ones with the random precision");
    if (C.dy != dy) return badarg("the precision's dimension is not the observations'");
    if (C.n_models != 1 || C.deterministic || !C.cx.empty() || C.du > 0) return unsupported("unknown observation precision: plain chains of one model only");
    if (C.d > 4 || C.dy > 4) return unsupported("unknown observation precision: d, dy <= 4");
    if (g->allow_missing) return unsupported("unknown observation precision with missing observations");
    last_error().clear();
    return RXHIP_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Mean-field mixture (NormalMixture, K ≤ 16) and its K = 1 form (iid Gaussian with unknown mean and precision)
struct Gmm {
    long long N = 0;
    int K = 0;
    std::vector<double> mu0, v0, a0, b0, alpha0, qm_mean, qm_var, qp_shape, qp_rate, qs_alpha;
    std::vector<long long> data_var;
};
inline rxhip_status lower_gmm(const rxhip_graph_desc* g, Gmm& M) {
    if (rxhip_status st = check_tables(g)) return st;
    const long long NV = g->n_variables, NF = g->n_factors;
    M = Gmm();
    std::vector<long long> prior_of(NV, -1), cat_of(NV, -1);
    std::vector<long long> mix;
    long long s_var = -1, s_prior = -1;
    for (long long f = 0; f < NF; ++f) {
        const int t = g->factor_type[f], n = n_iface(g, f);
        const long long out = iface(g, f, 0);
        switch (t) {
        case RXHIP_NODE_NORMAL_MEAN_VARIANCE:
        case RXHIP_NODE_GAMMA_SHAPE_RATE:
        case RXHIP_NODE_GAMMA_SHAPE_SCALE:
            if (n != 3) return badarg("prior node must have 3 interfaces");
            if (g->var_kind[out] != RXHIP_VARKIND_RANDOM) return unsupported("Normal / Gamma node on a non-random variable in a mixture graph");
            if (prior_of[out] >= 0) return unsupported("variable with two prior nodes");
            prior_of[out] = f;
            break;
        case RXHIP_NODE_DIRICHLET:
        case RXHIP_NODE_BETA:
            if (n != (t == RXHIP_NODE_DIRICHLET ? 2 : 3)) return badarg("Dirichlet / Beta node with the wrong number of interfaces");
            if (s_prior >= 0) return unsupported("more than one switch prior");
            s_prior = f;
            s_var = out;
            break;
        case RXHIP_NODE_CATEGORICAL:
        case RXHIP_NODE_BERNOULLI:
            if (n != 2) return badarg("Categorical / Bernoulli node must have 2 interfaces");
            if (cat_of[out] >= 0) return unsupported("switch variable with two Categorical nodes");
            cat_of[out] = f;
            break;
        case RXHIP_NODE_NORMAL_MIXTURE:
        case RXHIP_NODE_NORMAL_MEAN_PRECISION:
            mix.push_back(f);
            break;
        default:
            return unsupported("node type " + std::to_string(t) + " has no place in a mixture graph");
        }
    }
    if (mix.empty()) return unsupported("no NormalMixture / Normal(mean, precision) observation nodes");
    const bool iid = g->factor_type[mix[0]] == RXHIP_NODE_NORMAL_MEAN_PRECISION;
    const int K = iid ? 1 : (n_iface(g, mix[0]) - 2) / 2;
    if (K < 1 || K > 16 || (!iid && n_iface(g, mix[0]) != 2 + 2 * K)) return unsupported("mixture with an unsupported number of components");
    std::vector<long long> mv(K), pv(K);
    for (int k = 0; k < K; ++k) {
        mv[k] = iface(g, mix[0], iid ? 1 : 2 + k);
        pv[k] = iface(g, mix[0], iid ? 2 : 2 + K + k);
    }
    long long used = 0;
    for (long long f : mix) {
        if (g->factor_type[f] != g->factor_type[mix[0]] || n_iface(g, f) != n_iface(g, mix[0])) return unsupported("observation nodes of different shapes");
        const long long y = iface(g, f, 0);
        if (g->var_kind[y] != RXHIP_VARKIND_DATA || g->var_rows[y] != 1) return unsupported("mixture observation must be a scalar data variable");
        for (int k = 0; k < K; ++k)
            if (iface(g, f, iid ? 1 : 2 + k) != mv[k] || iface(g, f, iid ? 2 : 2 + K + k) != pv[k])
                return unsupported("observation nodes do not share the component variables");
        if (!iid) {
            const long long z = iface(g, f, 1);
            if (g->var_kind[z] != RXHIP_VARKIND_RANDOM || cat_of[z] < 0) return unsupported("switch variable without a Categorical / Bernoulli node");
            if (iface(g, cat_of[z], 1) != s_var) return unsupported("switch variables do not share one probability vector");
            cat_of[z] = -2;  // consumed
            used += 1;
        }
        M.data_var.push_back(y);
        used += 1;
    }
    for (long long v = 0; v < NV; ++v)
        if (cat_of[v] >= 0) return unsupported("Categorical node outside the mixture");
    M.N = (long long)mix.size();
    M.K = K;
    auto grab = [&](std::vector<double>& a, std::vector<double>& b, const std::vector<long long>& vars, int node, int fam,
                    std::vector<double>& qa, std::vector<double>& qb, const char* what) -> rxhip_status {
        for (int k = 0; k < K; ++k) {
            const long long f = prior_of[vars[k]];
            // `Gamma(shape = …, scale = …)` (and the positional `Gamma(α, θ)` of Distributions) is the rate form with β = 1/θ
            const bool scale_form = f >= 0 && node == RXHIP_NODE_GAMMA_SHAPE_RATE && g->factor_type[f] == RXHIP_NODE_GAMMA_SHAPE_SCALE;
            if (f < 0 || (g->factor_type[f] != node && !scale_form)) return unsupported(std::string("component ") + what + " without its prior node");
            double x, y;
            if (!const_scalar(g, iface(g, f, 1), &x) || !const_scalar(g, iface(g, f, 2), &y)) return unsupported(std::string("prior of ") + what + " with non-constant parameters");
            if (scale_form) {
                if (!(y > 0.0)) return badarg("Gamma prior with a non-positive scale");
                y = 1.0 / y;
            }
            a.push_back(x);
            b.push_back(y);
            const double* q;
            if (!init_params(g, vars[k], fam, 2, &q)) return badarg(std::string("mean-field VMP needs an @initialization marginal for every ") + what);
            qa.push_back(q[0]);
            qb.push_back(q[1]);
            used += 1;
        }
        return RXHIP_OK;
    };
    if (rxhip_status st = grab(M.mu0, M.v0, mv, RXHIP_NODE_NORMAL_MEAN_VARIANCE, RXHIP_INIT_NORMAL, M.qm_mean, M.qm_var, "m[k]")) return st;
    if (rxhip_status st = grab(M.a0, M.b0, pv, RXHIP_NODE_GAMMA_SHAPE_RATE, RXHIP_INIT_GAMMA, M.qp_shape, M.qp_rate, "p[k]")) return st;
    if (iid) {
        if (s_prior >= 0) return unsupported("switch prior in a graph without a mixture node");
        M.alpha0.assign(1, 1.0);
        M.qs_alpha.assign(1, 1.0);
    } else {
        if (s_prior < 0) return unsupported("mixture without a Dirichlet / Beta prior on the switch probabilities");
        if (g->factor_type[s_prior] == RXHIP_NODE_BETA) {
            double a, b;
            if (K != 2 || !const_scalar(g, iface(g, s_prior, 1), &a) || !const_scalar(g, iface(g, s_prior, 2), &b)) return unsupported("Beta switch prior needs K = 2 and constant parameters");
            M.alpha0 = {a, b};
        } else {
            const double* al;
            if (!const_value(g, iface(g, s_prior, 1), K, 1, &al)) return unsupported("Dirichlet prior with non-constant or mis-sized parameters");
            M.alpha0.assign(al, al + K);
        }
        const double* q;
        if (init_params(g, s_var, RXHIP_INIT_DIRICHLET, K, &q)) M.qs_alpha.assign(q, q + K);
        else M.qs_alpha.assign(K, 1.0);
        used += 1;
    }
    if (used != NF) return unsupported("graph has factors outside the mixture");
    last_error().clear();
    return RXHIP_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Hierarchical Gaussian filter, one-step graph
struct Hgf {
    double kappa = 0, omega = 0, z_variance = 0, y_variance = 0, z0m = 0, z0v = 0, x0m = 0, x0v = 0;
    int n_gh = 31;
    long long zt = -1, xt = -1, y = -1;
};
inline rxhip_status lower_hgf(const rxhip_graph_desc* g, Hgf& H) {
    if (rxhip_status st = check_tables(g)) return st;
    H = Hgf();
    long long gcv = -1;
    for (long long f = 0; f < g->n_factors; ++f) {
        if (g->factor_type[f] == RXHIP_NODE_GCV) {
            if (gcv >= 0) return unsupported("more than one GCV node: not the one-step filter graph");
            gcv = f;
        } else if (g->factor_type[f] != RXHIP_NODE_NORMAL_MEAN_VARIANCE)
            return unsupported("node type " + std::to_string(g->factor_type[f]) + " has no place in the HGF step graph");
    }
    if (gcv < 0 || n_iface(g, gcv) != 5 || g->n_factors != 5) return unsupported("HGF step graph = 4 Normal nodes + 1 GCV node");
    const long long xt = iface(g, gcv, 0), xt_min = iface(g, gcv, 1), zt = iface(g, gcv, 2);
    if (!const_scalar(g, iface(g, gcv, 3), &H.kappa) || !const_scalar(g, iface(g, gcv, 4), &H.omega)) return unsupported("GCV with non-constant κ / ω");
    int seen = 0;
    long long zt_min = -1, zprior_out = -1;
    for (long long f = 0; f < g->n_factors; ++f) {
        if (f == gcv) continue;
        if (n_iface(g, f) != 3) return badarg("Normal node must have 3 interfaces");
        const long long out = iface(g, f, 0), mu = iface(g, f, 1), v = iface(g, f, 2);
        if (out == zt) {  // zt ~ Normal(mean = zt_min, var = const)
            if (g->var_kind[mu] != RXHIP_VARKIND_RANDOM || !const_scalar(g, v, &H.z_variance)) return unsupported("upper layer must be a random walk with constant variance");
            zt_min = mu;
            seen |= 1;
        } else if (g->var_kind[out] == RXHIP_VARKIND_DATA) {  // y ~ Normal(mean = xt, var = const)
            if (mu != xt || !const_scalar(g, v, &H.y_variance)) return unsupported("observation must read xt with constant variance");
            H.y = out;
            seen |= 2;
        } else if (out == xt_min) {  // xt_min ~ Normal(data, data)
            if (g->var_kind[mu] != RXHIP_VARKIND_DATA || g->var_kind[v] != RXHIP_VARKIND_DATA) return unsupported("xt_min prior must be fed by @autoupdates data variables");
            seen |= 4;
        } else {  // zt_min ~ Normal(data, data)
            if (g->var_kind[mu] != RXHIP_VARKIND_DATA || g->var_kind[v] != RXHIP_VARKIND_DATA) return unsupported("zt_min prior must be fed by @autoupdates data variables");
            if (zprior_out >= 0) return unsupported("unexpected Normal node");
            zprior_out = out;
            seen |= 8;
        }
    }
    if (seen != 15 || zprior_out != zt_min) return unsupported("HGF step graph is incomplete");
    const double *qz, *qx;
    if (!init_params(g, zt, RXHIP_INIT_NORMAL, 2, &qz) || !init_params(g, xt, RXHIP_INIT_NORMAL, 2, &qx))
        return badarg("the HGF filter needs @initialization marginals q(zt), q(xt)");
    H.z0m = qz[0]; H.z0v = qz[1]; H.x0m = qx[0]; H.x0v = qx[1];
    H.n_gh = g->gh_points > 0 ? g->gh_points : 31;
    H.zt = zt; H.xt = xt;
    last_error().clear();
    return RXHIP_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Multivariate mixture: MvNormalMeanCovariance priors on m[k] (constant mean / covariance), Wishart priors on w[k],
// Dirichlet + Categorical switch, NormalMixture observation nodes over d-vector data
struct MvGmm {
    long long N = 0;
    int K = 0, d = 0;
    std::vector<double> mu0, S0, nu0, V0, alpha0, qm_mean, qm_cov, qw_nu, qw_V, qs_alpha;
    std::vector<long long> data_var;
};
inline rxhip_status lower_mvgmm(const rxhip_graph_desc* g, MvGmm& M) {
    if (rxhip_status st = check_tables(g)) return st;
    const long long NV = g->n_variables, NF = g->n_factors;
    M = MvGmm();
    std::vector<long long> prior_of(NV, -1), cat_of(NV, -1), mix, iid;
    long long s_var = -1, s_prior = -1;
    for (long long f = 0; f < NF; ++f) {
        const int t = g->factor_type[f], n = n_iface(g, f);
        const long long out = iface(g, f, 0);
        if (t == RXHIP_NODE_MVNORMAL_MEAN_PRECISION && n == 3 && g->var_kind[out] == RXHIP_VARKIND_DATA) {
            iid.push_back(f);  // y[i] ~ MvNormal(μ = m, Λ = P): the K = 1 form, no switch
        } else if (t == RXHIP_NODE_MVNORMAL_MEAN_COV || t == RXHIP_NODE_MVNORMAL_MEAN_PRECISION || t == RXHIP_NODE_WISHART) {
            if (n != 3) return badarg("prior node must have 3 interfaces");
            if (g->var_kind[out] != RXHIP_VARKIND_RANDOM || prior_of[out] >= 0) return unsupported("MvNormal / Wishart node is not a component prior");
            prior_of[out] = f;
        } else if (t == RXHIP_NODE_DIRICHLET) {
            if (n != 2 || s_prior >= 0) return unsupported("more than one switch prior");
            s_prior = f;
            s_var = out;
        } else if (t == RXHIP_NODE_CATEGORICAL) {
            if (n != 2 || cat_of[out] >= 0) return unsupported("switch variable with two Categorical nodes");
            cat_of[out] = f;
        } else if (t == RXHIP_NODE_NORMAL_MIXTURE)
            mix.push_back(f);
        else
            return unsupported("node type " + std::to_string(t) + " has no place in a multivariate mixture graph");
    }
    if (!iid.empty()) {
        // iid multivariate Gaussian with unknown mean and precision (test/models/iid/mv_iid_precision_tests.jl:11-15)
        if (!mix.empty() || s_prior >= 0) return unsupported("iid observation nodes next to a mixture");
        const long long mvar = iface(g, iid[0], 1), pvar = iface(g, iid[0], 2);
        const int d = g->var_rows[iface(g, iid[0], 0)];
        if (d < 1 || d > 4) return unsupported("iid multivariate Gaussian with an unsupported dimension");
        if (g->var_kind[mvar] != RXHIP_VARKIND_RANDOM || g->var_kind[pvar] != RXHIP_VARKIND_RANDOM) return unsupported("iid Gaussian with a known mean or precision");
        for (long long f : iid) {
            if (iface(g, f, 1) != mvar || iface(g, f, 2) != pvar || g->var_rows[iface(g, f, 0)] != d) return unsupported("observation nodes do not share mean and precision");
            M.data_var.push_back(iface(g, f, 0));
        }
        const long long fm = prior_of[mvar], fw = prior_of[pvar];
        if (fm < 0 || g->factor_type[fm] == RXHIP_NODE_WISHART || fw < 0 || g->factor_type[fw] != RXHIP_NODE_WISHART)
            return unsupported("iid Gaussian without its MvNormal / Wishart priors");
        const double *mu, *S, *V, *q;
        double nu;
        if (!const_value(g, iface(g, fm, 1), d, 1, &mu) || !const_value(g, iface(g, fm, 2), d, d, &S)) return unsupported("MvNormal prior with non-constant parameters");
        if (!const_scalar(g, iface(g, fw, 1), &nu) || !const_value(g, iface(g, fw, 2), d, d, &V)) return unsupported("Wishart prior with non-constant parameters");
        M.mu0.assign(mu, mu + d);
        M.S0.assign(S, S + d * d);
        if (g->factor_type[fm] == RXHIP_NODE_MVNORMAL_MEAN_PRECISION) {  // Λ given: the descriptor carries the covariance
            std::vector<double> inv;
            switch (spd_inverse_checked(d, S, inv)) {
                case 1: return badarg("prior precision of the mean is not symmetric");
                case 2: return badarg("prior precision of the mean is not positive definite");
                default: break;
            }
            M.S0 = inv;
        }
        M.nu0.assign(1, nu);
        M.V0.assign(V, V + d * d);
        if (!init_params(g, mvar, RXHIP_INIT_MVNORMAL, d + d * d, &q)) return badarg("mean-field VMP needs an @initialization marginal for the mean");
        M.qm_mean.assign(q, q + d);
        M.qm_cov.assign(q + d, q + d + d * d);
        if (!init_params(g, pvar, RXHIP_INIT_WISHART, 1 + d * d, &q)) return badarg("mean-field VMP needs an @initialization marginal for the precision");
        M.qw_nu.assign(1, q[0]);
        M.qw_V.assign(q + 1, q + 1 + d * d);
        M.alpha0.assign(1, 1.0);
        M.qs_alpha.assign(1, 1.0);
        if ((long long)iid.size() + 2 != NF) return unsupported("graph has factors outside the iid model");
        M.N = (long long)iid.size();
        M.K = 1;
        M.d = d;
        last_error().clear();
        return RXHIP_OK;
    }
    if (mix.empty() || s_prior < 0) return unsupported("no NormalMixture nodes / no switch prior");
    const int K = (n_iface(g, mix[0]) - 2) / 2;
    const int d = g->var_rows[iface(g, mix[0], 0)];
    if (K < 1 || K > 16 || n_iface(g, mix[0]) != 2 + 2 * K || d < 1 || d > 4) return unsupported("mixture with an unsupported number of components / dimension");
    std::vector<long long> mv(K), wv(K);
    for (int k = 0; k < K; ++k) { mv[k] = iface(g, mix[0], 2 + k); wv[k] = iface(g, mix[0], 2 + K + k); }
    long long used = 1;
    for (long long f : mix) {
        if (n_iface(g, f) != 2 + 2 * K) return unsupported("observation nodes of different shapes");
        const long long y = iface(g, f, 0), z = iface(g, f, 1);
        if (g->var_kind[y] != RXHIP_VARKIND_DATA || g->var_rows[y] != d) return unsupported("mixture observation must be a d-vector data variable");
        for (int k = 0; k < K; ++k)
            if (iface(g, f, 2 + k) != mv[k] || iface(g, f, 2 + K + k) != wv[k]) return unsupported("observation nodes do not share the component variables");
        if (g->var_kind[z] != RXHIP_VARKIND_RANDOM || cat_of[z] < 0 || iface(g, cat_of[z], 1) != s_var) return unsupported("switch variable without its Categorical node");
        cat_of[z] = -2;
        M.data_var.push_back(y);
        used += 2;
    }
    for (int k = 0; k < K; ++k) {
        const long long fm = prior_of[mv[k]], fw = prior_of[wv[k]];
        if (fm < 0 || g->factor_type[fm] != RXHIP_NODE_MVNORMAL_MEAN_COV || fw < 0 || g->factor_type[fw] != RXHIP_NODE_WISHART)
            return unsupported("component without its MvNormal / Wishart prior node");
        const double *mu, *S, *V, *q;
        double nu;
        if (!const_value(g, iface(g, fm, 1), d, 1, &mu) || !const_value(g, iface(g, fm, 2), d, d, &S)) return unsupported("MvNormal prior with non-constant parameters");
        if (!const_scalar(g, iface(g, fw, 1), &nu) || !const_value(g, iface(g, fw, 2), d, d, &V)) return unsupported("Wishart prior with non-constant parameters");
        M.mu0.insert(M.mu0.end(), mu, mu + d);
        M.S0.insert(M.S0.end(), S, S + d * d);
        M.nu0.push_back(nu);
        M.V0.insert(M.V0.end(), V, V + d * d);
        if (!init_params(g, mv[k], RXHIP_INIT_MVNORMAL, d + d * d, &q)) return badarg("mean-field VMP needs an @initialization marginal for every m[k]");
        M.qm_mean.insert(M.qm_mean.end(), q, q + d);
        M.qm_cov.insert(M.qm_cov.end(), q + d, q + d + d * d);
        if (!init_params(g, wv[k], RXHIP_INIT_WISHART, 1 + d * d, &q)) return badarg("mean-field VMP needs an @initialization marginal for every w[k]");
        M.qw_nu.push_back(q[0]);
        M.qw_V.insert(M.qw_V.end(), q + 1, q + 1 + d * d);
        used += 2;
    }
    const double* al;
    if (!const_value(g, iface(g, s_prior, 1), K, 1, &al)) return unsupported("Dirichlet prior with non-constant or mis-sized parameters");
    M.alpha0.assign(al, al + K);
    const double* q;
    if (init_params(g, s_var, RXHIP_INIT_DIRICHLET, K, &q)) M.qs_alpha.assign(q, q + K);
    else M.qs_alpha.assign(K, 1.0);
    if (used != NF) return unsupported("graph has factors outside the mixture");
    M.N = (long long)mix.size();
    M.K = K;
    M.d = d;
    last_error().clear();
    return RXHIP_OK;
}

}  // namespace rxhip_lower
