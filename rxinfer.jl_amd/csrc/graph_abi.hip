// graph_abi.hip — the graph entry points of the C ABI: the host-only lowering passes (graph_lowering.hpp: which pattern-matched family a materialised
// GraphPPL model belongs to, and its structured descriptor), rxhip_create (pattern matcher first, the level-scheduled node-array executor for every
// graph no family matches), the executor's rxhip_tree_* wrappers around tree_engine.hpp, rxhip_rule_eval.  No kernels live here.
#include "engine.hpp"

#include <cstring>

#include "graph_lowering.hpp"

using namespace rxhip;

extern "C" {

static void fill_lowered(const rxhip_lower::Lgssm& L, rxhip_lgssm_lowered* out, bool with_Q);
rxhip_status rxhip_graph_lower_lgssm(const rxhip_graph_desc* g, rxhip_lgssm_lowered* out) {
    if (!out) return RXHIP_ERR_BADARG;
    rxhip_lower::last_asymmetry() = 0.0;
    rxhip_lower::Lgssm L;
    rxhip_status st = rxhip_lower::lower_lgssm(g, L);
    if (st) return st;
    fill_lowered(L, out, true);
    return RXHIP_OK;
}
rxhip_status rxhip_graph_lower_lgssm_noise(const rxhip_graph_desc* g, rxhip_lgssm_noise_lowered* out) {
    if (!out) return RXHIP_ERR_BADARG;
    rxhip_lower::last_asymmetry() = 0.0;
    rxhip_lower::LgssmNoise N;
    rxhip_status st = rxhip_lower::lower_lgssm_noise(g, N);
    if (st) return st;
    fill_lowered(N.chain, &out->chain, false);
    out->precision_var = N.w_var;
    out->nu0 = N.nu0;
    out->init_nu = N.init_nu;
    if (out->S0) std::memcpy(out->S0, N.S0.data(), sizeof(double) * N.S0.size());
    if (out->init_V) std::memcpy(out->init_V, N.init_V.data(), sizeof(double) * N.init_V.size());
    return RXHIP_OK;
}
static void fill_lowered(const rxhip_lower::Lgssm& L, rxhip_lgssm_lowered* out, bool with_Q) {
    out->d = L.d; out->dy = L.dy; out->T = L.T; out->prior_through_transition = L.ptt;
    out->deterministic = L.deterministic;
    out->n_models = L.n_models;
    out->has_offsets = L.cx.empty() ? 0 : 1;
    out->du = L.du;
    if (out->input_matrix && L.du > 0) std::memcpy(out->input_matrix, L.Bu.data(), sizeof(double) * L.Bu.size());
    if (out->input_var) for (long long t = 0; t < L.T; ++t) out->input_var[t] = L.du > 0 ? L.input_var[t] : -1;
    if (out->state_offset) for (size_t q = 0; q < (size_t)L.T * L.d; ++q) out->state_offset[q] = L.cx.empty() ? 0.0 : L.cx[q];
    if (out->obs_offset) for (size_t q = 0; q < (size_t)L.T * L.dy; ++q) out->obs_offset[q] = L.cy.empty() ? 0.0 : L.cy[q];
    if (out->step_model) for (long long t = 0; t < L.T; ++t) out->step_model[t] = L.n_models > 1 ? L.step_model[t] : 0;
    if (out->c) std::memcpy(out->c, L.c.data(), L.c.size() * sizeof(double));
    auto cp = [](double* dst, const std::vector<double>& v) { if (dst) std::memcpy(dst, v.data(), v.size() * sizeof(double)); };
    cp(out->A, L.A); cp(out->B, L.B); cp(out->P, L.P); if (with_Q) cp(out->Q, L.Q); cp(out->m0, L.m0); cp(out->V0, L.V0);
    if (out->state_var) for (long long t = 0; t < L.T; ++t) out->state_var[t] = L.state_var[t];
    if (out->data_var) for (long long t = 0; t < L.T; ++t) out->data_var[t] = L.data_var[t];
}
const char* rxhip_lowering_error(void) { return rxhip_lower::last_error().c_str(); }
double rxhip_lowering_asymmetry(void) { return rxhip_lower::last_asymmetry(); }

rxhip_status rxhip_graph_lower_gmm(const rxhip_graph_desc* g, rxhip_gmm_lowered* out) {
    if (!out) return RXHIP_ERR_BADARG;
    rxhip_lower::Gmm M;
    rxhip_status st = rxhip_lower::lower_gmm(g, M);
    if (st) return st;
    out->N = M.N; out->K = M.K;
    auto cp = [](double* dst, const std::vector<double>& v) { if (dst) std::memcpy(dst, v.data(), v.size() * sizeof(double)); };
    cp(out->mu0, M.mu0); cp(out->v0, M.v0); cp(out->a0, M.a0); cp(out->b0, M.b0); cp(out->alpha0, M.alpha0);
    cp(out->init_m_mean, M.qm_mean); cp(out->init_m_var, M.qm_var); cp(out->init_p_shape, M.qp_shape);
    cp(out->init_p_rate, M.qp_rate); cp(out->init_s_alpha, M.qs_alpha);
    if (out->data_var) for (long long i = 0; i < M.N; ++i) out->data_var[i] = M.data_var[i];
    return RXHIP_OK;
}
rxhip_status rxhip_graph_lower_mvgmm(const rxhip_graph_desc* g, rxhip_mvgmm_lowered* out) {
    if (!out) return RXHIP_ERR_BADARG;
    rxhip_lower::last_asymmetry() = 0.0;
    rxhip_lower::MvGmm M;
    rxhip_status st = rxhip_lower::lower_mvgmm(g, M);
    if (st) return st;
    out->N = M.N; out->K = M.K; out->d = M.d;
    auto cp = [](double* dst, const std::vector<double>& v) { if (dst) std::memcpy(dst, v.data(), v.size() * sizeof(double)); };
    cp(out->mu0, M.mu0); cp(out->S0, M.S0); cp(out->nu0, M.nu0); cp(out->V0, M.V0); cp(out->alpha0, M.alpha0);
    cp(out->init_m_mean, M.qm_mean); cp(out->init_m_cov, M.qm_cov); cp(out->init_w_nu, M.qw_nu); cp(out->init_w_V, M.qw_V);
    cp(out->init_s_alpha, M.qs_alpha);
    if (out->data_var) for (long long i = 0; i < M.N; ++i) out->data_var[i] = M.data_var[i];
    return RXHIP_OK;
}
rxhip_status rxhip_graph_lower_hgf(const rxhip_graph_desc* g, rxhip_hgf_lowered* out) {
    if (!out) return RXHIP_ERR_BADARG;
    rxhip_lower::Hgf H;
    rxhip_status st = rxhip_lower::lower_hgf(g, H);
    if (st) return st;
    out->kappa = H.kappa; out->omega = H.omega; out->z_variance = H.z_variance; out->y_variance = H.y_variance;
    out->z0_mean = H.z0m; out->z0_var = H.z0v; out->x0_mean = H.x0m; out->x0_var = H.x0v;
    out->n_gh = H.n_gh; out->zt_var = H.zt; out->xt_var = H.xt; out->y_var = H.y;
    return RXHIP_OK;
}

// the node-array executor behind an rxhip_engine handle (everything lives behind e->tree)
rxhip_status rxhip_tree_create(const rxhip_graph_desc* g, int32_t device, void* stream, rxhip_engine** out) {
    if (!out) return RXHIP_ERR_BADARG;
    *out = nullptr;
    rxhip::tree::Engine* t = nullptr;
    std::string err;
    rxhip_lower::last_asymmetry() = 0.0;
    const rxhip_status st = rxhip::tree::create(g, device, stream, &t, err);
    if (st) { rxhip_lower::last_error() = err; return st; }
    rxhip_engine* e = new rxhip_engine();
    e->kind = 5;
    e->tree = t;
    e->device = rxhip::tree::device_of(t);
    e->stream = (hipStream_t)rxhip::tree::stream_of(t);   // (owned by the executor; the cross-GPU free-energy sum is enqueued on it)
    e->n_chains = g->n_replicas > 0 ? g->n_replicas : 1;
    *out = e;
    return RXHIP_OK;
}
rxhip_status rxhip_tree_plan(const rxhip_graph_desc* g, rxhip_tree_info* out, uint64_t* rule_calls, uint64_t* products, uint64_t* marginals) {
    std::string err;
    rxhip_lower::last_asymmetry() = 0.0;
    const rxhip_status st = rxhip::tree::plan(g, out, rule_calls, products, marginals, err);
    if (st) rxhip_lower::last_error() = err;
    return st;
}
rxhip_status rxhip_tree_set_data(rxhip_engine* e, const int64_t* vars, int64_t n_vars, const double* host) {
    if (!e || !e->tree) return e ? fail(e, RXHIP_ERR_BADARG, "rxhip_tree_set_data: not an engine of the node-array executor") : RXHIP_ERR_BADARG;
    e->err.clear();
    return rxhip::tree::set_data(e->tree, vars, n_vars, host, e->err);
}
rxhip_status rxhip_tree_get_marginals(rxhip_engine* e, const int64_t* vars, int64_t n_vars, double* mean, double* cov) {
    if (!e || !e->tree) return e ? fail(e, RXHIP_ERR_BADARG, "rxhip_tree_get_marginals: not an engine of the node-array executor") : RXHIP_ERR_BADARG;
    e->err.clear();
    return rxhip::tree::get_marginals(e->tree, vars, n_vars, mean, cov, e->err);
}
rxhip_status rxhip_tree_get_precision(rxhip_engine* e, int64_t var, double* nu, double* V) {
    if (!e || !e->tree) return e ? fail(e, RXHIP_ERR_BADARG, "rxhip_tree_get_precision: not an engine of the node-array executor") : RXHIP_ERR_BADARG;
    e->err.clear();
    return rxhip::tree::get_precision(e->tree, var, nu, V, e->err);
}
rxhip_status rxhip_tree_get_discrete(rxhip_engine* e, int64_t var, double* out, int32_t* n_components) {
    if (!e || !e->tree) return e ? fail(e, RXHIP_ERR_BADARG, "rxhip_tree_get_discrete: not an engine of the node-array executor") : RXHIP_ERR_BADARG;
    e->err.clear();
    return rxhip::tree::get_discrete(e->tree, var, out, n_components, e->err);
}
rxhip_status rxhip_tree_get_info(rxhip_engine* e, rxhip_tree_info* out) {
    if (!e || !e->tree || !out) return RXHIP_ERR_BADARG;
    rxhip::tree::info(e->tree, out);
    return RXHIP_OK;
}
rxhip_status rxhip_tree_continue(rxhip_engine* e, int32_t on) {
    if (!e || !e->tree) return e ? fail(e, RXHIP_ERR_BADARG, "rxhip_tree_continue: not an engine of the node-array executor") : RXHIP_ERR_BADARG;
    rxhip::tree::set_continue(e->tree, on != 0);
    return RXHIP_OK;
}
rxhip_status rxhip_rule_eval(const rxhip_rule_call* call, int32_t device) {
    std::string err;
    const rxhip_status st = rxhip::tree::rule_eval(call, device, err);
    if (st) rxhip_lower::last_error() = err;   // (no handle to hang the text on: rxhip_lowering_error() returns it)
    return st;
}

static rxhip_status create_pattern_matched(const rxhip_graph_desc* g, int32_t segments, int32_t device, void* stream, rxhip_engine** out);
rxhip_status rxhip_create(const rxhip_graph_desc* g, int32_t segments, int32_t device, void* stream, rxhip_engine** out) {
    if (!out) return RXHIP_ERR_BADARG;
    // the pattern matcher first: its engines are the fast paths of the families they know; every graph it has no schedule for goes to the
    // level-scheduled node-array executor, and only what THAT rejects (a cycle, a non-Gaussian node it has no rule for) is RXHIP_ERR_UNSUPPORTED
    rxhip_status st = create_pattern_matched(g, segments, device, stream, out);
    if (st != RXHIP_ERR_UNSUPPORTED || (out && *out)) return st;
    const std::string why = rxhip_lower::last_error();
    st = rxhip_tree_create(g, device, stream, out);
    if (st == RXHIP_ERR_UNSUPPORTED) rxhip_lower::last_error() = why + " | node-array executor: " + rxhip_lower::last_error();
    return st;
}
static rxhip_status create_pattern_matched(const rxhip_graph_desc* g, int32_t segments, int32_t device, void* stream, rxhip_engine** out) {
    if (!out) return RXHIP_ERR_BADARG;
    *out = nullptr;
    rxhip_lower::last_asymmetry() = 0.0;
    if (rxhip_status st0 = rxhip_lower::check_tables(g)) return st0;
    // family by the node types present (the lowering passes reject everything that is not exactly their graph)
    if (rxhip_lower::has_node(g, RXHIP_NODE_GCV)) {
        rxhip_lower::Hgf H;
        rxhip_status st = rxhip_lower::lower_hgf(g, H);
        if (st) return st;
        if (g->n_observations <= 0) { rxhip_lower::last_error() = "streaming graph: n_observations must be positive"; return RXHIP_ERR_BADARG; }
        rxhip_hgf_desc d;
        std::memset(&d, 0, sizeof d);
        d.T = g->n_observations; d.n_series = g->n_replicas > 0 ? g->n_replicas : 1;
        d.kappa = H.kappa; d.omega = H.omega; d.z_variance = H.z_variance; d.y_variance = H.y_variance;
        d.z0_mean = H.z0m; d.z0_var = H.z0v; d.x0_mean = H.x0m; d.x0_var = H.x0v;
        d.n_gh = H.n_gh; d.device = device; d.stream = stream;
        return rxhip_hgf_create(&d, out);
    }
    // a precision prior on top of a state-space chain (no mixture node, a `*` node or a Gaussian transition between random variables):
    // the chain with unknown observation noise
    auto noise_chain = [&]() -> bool {
        if (rxhip_lower::has_node(g, RXHIP_NODE_NORMAL_MIXTURE)) return false;
        if (!rxhip_lower::has_node(g, RXHIP_NODE_WISHART) && !rxhip_lower::has_node(g, RXHIP_NODE_GAMMA_SHAPE_RATE) && !rxhip_lower::has_node(g, RXHIP_NODE_GAMMA_SHAPE_SCALE)) return false;
        if (g->factor_iface_ptr) return false;
        for (long long f = 0; f < g->n_factors; ++f) {
            const int t = g->factor_type[f];
            if (t == RXHIP_NODE_MULTIPLY) return true;
            if ((t == RXHIP_NODE_MVNORMAL_MEAN_COV || t == RXHIP_NODE_NORMAL_MEAN_VARIANCE || t == RXHIP_NODE_MVNORMAL_MEAN_PRECISION || t == RXHIP_NODE_NORMAL_MEAN_PRECISION) &&
                g->var_kind[rxhip_lower::iface(g, f, 0)] == RXHIP_VARKIND_RANDOM && g->var_kind[rxhip_lower::iface(g, f, 1)] == RXHIP_VARKIND_RANDOM)
                return true;
        }
        return false;
    };
    if (noise_chain()) {
        rxhip_lower::LgssmNoise N;
        rxhip_status st = rxhip_lower::lower_lgssm_noise(g, N);
        if (st) return st;
        const rxhip_lower::Lgssm& L = N.chain;
        rxhip_lgssm_desc d;
        std::memset(&d, 0, sizeof d);
        d.d = L.d; d.dy = L.dy; d.T = L.T; d.n_chains = g->n_replicas > 0 ? g->n_replicas : 1; d.n_models = 1;
        d.prior_through_transition = L.ptt;
        d.A = L.A.data(); d.B = L.B.data(); d.P = L.P.data(); d.Q = nullptr; d.m0 = L.m0.data(); d.V0 = L.V0.data();
        d.segments = segments; d.device = device; d.stream = stream;
        rxhip_noise_prior pr;
        pr.nu0 = N.nu0; pr.S0 = N.S0.data(); pr.init_nu = N.init_nu; pr.init_V = N.init_V.data();
        return rxhip_lgssm_noise_create(&d, &pr, out);
    }
    if (rxhip_lower::has_node(g, RXHIP_NODE_WISHART)) {
        rxhip_lower::MvGmm M;
        rxhip_status st = rxhip_lower::lower_mvgmm(g, M);
        if (st) return st;
        rxhip_mvgmm_desc d;
        std::memset(&d, 0, sizeof d);
        d.N = M.N; d.K = M.K; d.d = M.d;
        d.mu0 = M.mu0.data(); d.S0 = M.S0.data(); d.nu0 = M.nu0.data(); d.V0 = M.V0.data(); d.alpha0 = M.alpha0.data();
        d.init_m_mean = M.qm_mean.data(); d.init_m_cov = M.qm_cov.data(); d.init_w_nu = M.qw_nu.data(); d.init_w_V = M.qw_V.data();
        d.init_s_alpha = M.qs_alpha.data();
        d.device = device; d.stream = stream;
        return rxhip_mvgmm_create(&d, out);
    }
    // `NormalMeanPrecision` with a RANDOM precision is the iid Gaussian×Gamma model; with constant precisions it is a Gaussian
    // chain written in precision form (test/inference/prediction_tests.jl:197-213) and belongs to the state-space lowering
    bool random_precision = false;
    for (long long f = 0; f < g->n_factors && !random_precision; ++f)
        random_precision = g->factor_type[f] == RXHIP_NODE_NORMAL_MEAN_PRECISION && rxhip_lower::n_iface(g, f) == 3 &&
                           g->var_kind[rxhip_lower::iface(g, f, 2)] == RXHIP_VARKIND_RANDOM;
    if (rxhip_lower::has_node(g, RXHIP_NODE_NORMAL_MIXTURE) || random_precision) {
        rxhip_lower::Gmm M;
        rxhip_status st = rxhip_lower::lower_gmm(g, M);
        if (st) return st;
        rxhip_gmm_desc d;
        std::memset(&d, 0, sizeof d);
        d.N = M.N; d.K = M.K;
        d.mu0 = M.mu0.data(); d.v0 = M.v0.data(); d.a0 = M.a0.data(); d.b0 = M.b0.data(); d.alpha0 = M.alpha0.data();
        d.init_m_mean = M.qm_mean.data(); d.init_m_var = M.qm_var.data(); d.init_p_shape = M.qp_shape.data();
        d.init_p_rate = M.qp_rate.data(); d.init_s_alpha = M.qs_alpha.data();
        d.device = device; d.stream = stream;
        return rxhip_gmm_create(&d, out);
    }
    rxhip_lower::Lgssm L;
    rxhip_status st = rxhip_lower::lower_lgssm(g, L);
    if (st) return st;
    if (L.deterministic) {
        rxhip_drift_chain_desc dd;
        std::memset(&dd, 0, sizeof dd);
        dd.T = L.T; dd.n_chains = g->n_replicas > 0 ? g->n_replicas : 1;
        dd.m0 = L.m0[0]; dd.v0 = L.V0[0]; dd.c = L.c[0]; dd.obs_var = L.Q[0];
        dd.prior_through_transition = L.ptt; dd.device = device; dd.stream = stream;
        return rxhip_drift_chain_create(&dd, out);
    }
    rxhip_lgssm_desc d;
    std::memset(&d, 0, sizeof d);
    d.d = L.d; d.dy = L.dy; d.T = L.T; d.n_chains = g->n_replicas > 0 ? g->n_replicas : 1; d.n_models = L.n_models;
    d.prior_through_transition = L.ptt;
    std::vector<double> m0s, V0s;
    if (L.n_models > 1) {  // per-step constants: every model carries the (one) prior of the chain
        for (int m = 0; m < L.n_models; ++m) {
            m0s.insert(m0s.end(), L.m0.begin(), L.m0.end());
            V0s.insert(V0s.end(), L.V0.begin(), L.V0.end());
        }
        d.step_model = L.step_model.data();
    }
    d.A = L.A.data(); d.B = L.B.data(); d.P = L.P.data(); d.Q = L.Q.data();
    d.m0 = L.n_models > 1 ? m0s.data() : L.m0.data(); d.V0 = L.n_models > 1 ? V0s.data() : L.V0.data();
    d.allow_missing = g->allow_missing;
    if (!L.cx.empty()) { d.state_offset = L.cx.data(); d.obs_offset = L.cy.data(); }
    d.segments = segments; d.device = device; d.stream = stream;
    st = rxhip_lgssm_create(&d, out);
    if (!st && L.du > 0) {  // data inputs: u arrives through rxhip_set_data(RXHIP_VAR_U); its constant part (if any) stays in c
        (*out)->du = L.du;
        (*out)->h_Bu = L.Bu;
        (*out)->h_umask.assign((size_t)L.T, 0);
        for (long long t = 0; t < L.T; ++t) (*out)->h_umask[(size_t)t] = L.input_var[(size_t)t] >= 0 ? 1 : 0;
    }
    return st;
}


}  // extern "C"
