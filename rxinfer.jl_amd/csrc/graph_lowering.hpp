// graph_lowering.hpp — host-side lowering of a generic factor-graph descriptor (include/rxhip.h
// rxhip_graph_desc) to the structured schedule descriptors.  Replaces, for the graph shapes that have a
// device schedule, the per-node object construction of
// GraphPPL.postprocess_plugin(::ReactiveMPInferencePlugin, model) (src/model/plugins/reactivemp_inference.jl:272-326):
// instead of one ReactiveMP object per variable / factor, one pass over the SoA tables.
#pragma once
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
    std::vector<double> A, B, P, Q, m0, V0;
    std::vector<long long> state_var, data_var;
};

// value of a constant variable, with shape check
inline bool const_value(const rxhip_graph_desc* g, long long v, int rows, int cols, const double** out) {
    if (g->var_kind[v] != RXHIP_VARKIND_CONST || g->var_const[v] < 0) return false;
    if (g->var_rows[v] != rows || g->var_cols[v] != cols) return false;
    if (g->var_const[v] + (long long)rows * cols > g->n_const) return false;
    *out = g->const_pool + g->var_const[v];
    return true;
}
inline bool same_const(const rxhip_graph_desc* g, long long a, long long b) {
    if (a == b) return true;
    if (g->var_rows[a] != g->var_rows[b] || g->var_cols[a] != g->var_cols[b]) return false;
    const size_t n = (size_t)g->var_rows[a] * g->var_cols[a];
    return std::memcmp(g->const_pool + g->var_const[a], g->const_pool + g->var_const[b], n * sizeof(double)) == 0;
}

// Recognise   prior:  MvN(out = x_first, μ = const, Σ = const)
//             per state x:  [`*`(out = b, A = B, in = x);  MvN(out = y (data), μ = b, Σ = Q)]      (observation)
//                           [`*`(out = a, A = A, in = x);  MvN(out = x_next (random), μ = a, Σ = P)] (transition)
// with time-invariant constants.  Node order in the tables is irrelevant.
inline rxhip_status lower_lgssm(const rxhip_graph_desc* g, Lgssm& L) {
    if (!g || g->n_variables <= 0 || g->n_factors <= 0 || !g->var_kind || !g->var_rows || !g->var_cols || !g->var_const ||
        !g->factor_type || !g->factor_iface || (g->n_const > 0 && !g->const_pool))
        return badarg("graph descriptor has null tables");
    const long long NV = g->n_variables, NF = g->n_factors;
    for (long long f = 0; f < NF; ++f)
        for (int k = 0; k < 3; ++k)
            if (g->factor_iface[f * 3 + k] < 0 || g->factor_iface[f * 3 + k] >= NV) return badarg("factor interface refers to an unknown variable");
    // `*` node producing each (anonymous) variable, and MvN nodes by their μ variable
    std::vector<long long> mul_of_out(NV, -1), mvn_of_mu(NV, -1);
    std::vector<std::vector<long long>> mul_of_in(NV);
    long long prior = -1;
    for (long long f = 0; f < NF; ++f) {
        const int64_t* io = g->factor_iface + f * 3;
        if (g->factor_type[f] == RXHIP_NODE_MULTIPLY) {
            if (g->var_kind[io[1]] != RXHIP_VARKIND_CONST) return unsupported("`*` node with a non-constant matrix (no device schedule)");
            if (g->var_kind[io[2]] != RXHIP_VARKIND_RANDOM || g->var_kind[io[0]] != RXHIP_VARKIND_RANDOM)
                return unsupported("`*` node must connect two random variables");
            if (mul_of_out[io[0]] >= 0) return unsupported("variable produced by two `*` nodes");
            mul_of_out[io[0]] = f;
            mul_of_in[io[2]].push_back(f);
        } else if (g->factor_type[f] == RXHIP_NODE_MVNORMAL_MEAN_COV) {
            if (g->var_kind[io[2]] != RXHIP_VARKIND_CONST) return unsupported("MvNormalMeanCovariance with a non-constant covariance");
            if (g->var_kind[io[1]] == RXHIP_VARKIND_CONST) {
                if (prior >= 0) return unsupported("more than one prior node: not a single chain");
                prior = f;
            } else {
                if (mvn_of_mu[io[1]] >= 0) return unsupported("mean variable shared by two MvNormal nodes");
                mvn_of_mu[io[1]] = f;
            }
        } else
            return unsupported("node type " + std::to_string(g->factor_type[f]) + " has no device schedule");
    }
    if (prior < 0) return unsupported("no prior node (MvNormalMeanCovariance with constant mean)");
    long long x = g->factor_iface[prior * 3 + 0];
    if (g->var_kind[x] != RXHIP_VARKIND_RANDOM) return unsupported("prior node on a non-random variable");
    const int d = g->var_rows[x];
    const double *m0, *V0;
    if (!const_value(g, g->factor_iface[prior * 3 + 1], d, 1, &m0) || !const_value(g, g->factor_iface[prior * 3 + 2], d, d, &V0))
        return badarg("prior constants have the wrong shape");
    long long vA = -1, vP = -1, vB = -1, vQ = -1;
    long long used_factors = 1;
    bool first = true;
    L = Lgssm();
    while (true) {
        long long obs_mul = -1, tr_mul = -1;
        for (long long f : mul_of_in[x]) {
            const long long outv = g->factor_iface[f * 3 + 0];
            const long long mv = mvn_of_mu[outv];
            if (mv < 0) return unsupported("`*` node whose output feeds no MvNormal mean");
            const long long target = g->factor_iface[mv * 3 + 0];
            if (g->var_kind[target] == RXHIP_VARKIND_DATA) {
                if (obs_mul >= 0) return unsupported("state with two observation branches");
                obs_mul = f;
            } else if (g->var_kind[target] == RXHIP_VARKIND_RANDOM) {
                if (tr_mul >= 0) return unsupported("state with two transitions: not a chain");
                tr_mul = f;
            } else
                return unsupported("MvNormal with a constant output");
        }
        if (obs_mul >= 0) {
            const long long mv = mvn_of_mu[g->factor_iface[obs_mul * 3 + 0]];
            const long long b = g->factor_iface[obs_mul * 3 + 1], q = g->factor_iface[mv * 3 + 2];
            if (vB < 0) { vB = b; vQ = q; }
            else if (!same_const(g, vB, b) || !same_const(g, vQ, q)) return unsupported("time-varying observation model");
            L.state_var.push_back(x);
            L.data_var.push_back(g->factor_iface[mv * 3 + 0]);
            used_factors += 2;
        } else if (first && tr_mul >= 0) {
            L.ptt = 1;  // x0 ~ prior without an observation: test/models/statespace/mlgssm_test.jl:9-17
        } else
            return unsupported("state variable without an observation");
        first = false;
        if (tr_mul < 0) break;
        const long long mv = mvn_of_mu[g->factor_iface[tr_mul * 3 + 0]];
        const long long a = g->factor_iface[tr_mul * 3 + 1], pv = g->factor_iface[mv * 3 + 2];
        if (vA < 0) { vA = a; vP = pv; }
        else if (!same_const(g, vA, a) || !same_const(g, vP, pv)) return unsupported("time-varying transition model");
        used_factors += 2;
        x = g->factor_iface[mv * 3 + 0];
        if (g->var_rows[x] != d) return unsupported("state dimension changes along the chain");
    }
    if (used_factors != NF) return unsupported("graph has factors outside the state-space chain");
    L.T = (long long)L.state_var.size();
    if (L.T <= 0 || vB < 0) return unsupported("chain without observations");
    L.d = d;
    L.dy = g->var_rows[vB];
    const double *pA = nullptr, *pP = nullptr, *pB, *pQ;
    if (!const_value(g, vB, L.dy, d, &pB) || !const_value(g, vQ, L.dy, L.dy, &pQ)) return badarg("observation constants have the wrong shape");
    if (vA >= 0 && (!const_value(g, vA, d, d, &pA) || !const_value(g, vP, d, d, &pP))) return badarg("transition constants have the wrong shape");
    L.m0.assign(m0, m0 + d);
    L.V0.assign(V0, V0 + (size_t)d * d);
    L.B.assign(pB, pB + (size_t)L.dy * d);
    L.Q.assign(pQ, pQ + (size_t)L.dy * L.dy);
    if (pA) {
        L.A.assign(pA, pA + (size_t)d * d);
        L.P.assign(pP, pP + (size_t)d * d);
    } else {  // single time step: no transition in the graph
        L.A.assign((size_t)d * d, 0.0);
        L.P.assign((size_t)d * d, 0.0);
        for (int i = 0; i < d; ++i) L.A[i * d + i] = L.P[i * d + i] = 1.0;
    }
    last_error().clear();
    return RXHIP_OK;
}

}  // namespace rxhip_lower
