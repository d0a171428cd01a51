// tree_compiler.hpp — the node-array executor's graph COMPILER (host only): rxhip_graph_desc → Program (op tables sorted by dependency level, slot offsets, the
// constant pool, the strand schedule, the reference-equivalent counts).  Included by tree_engine.hip alone (the engine: device state, schedules, launches);
// `rxhip_tree_plan` runs exactly this and nothing of the device.  What the passes do is described where they stand: parse / expand_mixtures (virtual nodes of
// NormalMixture and GCV) / check_family / build_edges / build_deps / analyse (forms, aliases, levels, hubs) / allocate / emit_all (rules, products, marginals, then
// Bethe terms, residual moments, q(W) / q(z) / q(s) / γ updates, the sum tree) / finish (dead messages, byte counts) / count / build_strands.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/rxhip.h"
#include "graph_lowering.hpp"  // check_factorisation, last_asymmetry
#include "tree_kernels.hpp"    // opcodes, op words, flags
#include "tree_wave.hpp"       // DMAX_WAVE

namespace rxhip {
namespace tree {

namespace {

struct Fail {
    rxhip_status st;
    std::string msg;
};
[[noreturn]] void fail(rxhip_status st, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw Fail{st, buf};
}

bool host_chol_inv(int n, const double* A, double* out, double* logdet) {
    std::vector<double> L((size_t)n * n, 0.0), Li((size_t)n * n, 0.0);
    double ld = 0.0;
    for (int j = 0; j < n; ++j) {
        double s = A[j * n + j];
        for (int k = 0; k < j; ++k) s -= L[j * n + k] * L[j * n + k];
        if (!(s > 0.0) || !std::isfinite(s)) return false;
        L[j * n + j] = std::sqrt(s);
        ld += std::log(s);
        for (int i = j + 1; i < n; ++i) {
            double t = 0.5 * (A[i * n + j] + A[j * n + i]);
            for (int k = 0; k < j; ++k) t -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = t / L[j * n + j];
        }
    }
    for (int j = 0; j < n; ++j) {
        Li[j * n + j] = 1.0 / L[j * n + j];
        for (int i = j + 1; i < n; ++i) {
            double t = 0.0;
            for (int k = j; k < i; ++k) t += L[i * n + k] * Li[k * n + j];
            Li[i * n + j] = -t / L[i * n + i];
        }
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            double t = 0.0;
            for (int k = i; k < n; ++k) t += Li[k * n + i] * Li[k * n + j];
            out[i * n + j] = out[j * n + i] = t;
        }
    if (logdet) *logdet = ld;
    return true;
}
// log|det A| of a square matrix (LU with partial pivoting); false: singular to working precision
bool host_logabsdet(int n, const double* A, double* out) {
    std::vector<double> M(A, A + (size_t)n * n);
    double ld = 0.0;
    for (int k = 0; k < n; ++k) {
        int piv = k;
        for (int i = k + 1; i < n; ++i)
            if (std::fabs(M[(size_t)i * n + k]) > std::fabs(M[(size_t)piv * n + k])) piv = i;
        const double p = M[(size_t)piv * n + k];
        if (!(std::fabs(p) > 0.0) || !std::isfinite(p)) return false;
        if (piv != k)
            for (int j = 0; j < n; ++j) std::swap(M[(size_t)k * n + j], M[(size_t)piv * n + j]);
        ld += std::log(std::fabs(p));
        for (int i = k + 1; i < n; ++i) {
            const double f = M[(size_t)i * n + k] / p;
            for (int j = k; j < n; ++j) M[(size_t)i * n + j] -= f * M[(size_t)k * n + j];
        }
    }
    *out = ld;
    return true;
}
double host_digamma(double x) {
    double r = 0.0;
    while (x < 6.0) { r -= 1.0 / x; x += 1.0; }
    const double f = 1.0 / (x * x);
    return r + std::log(x) - 0.5 / x - f * (1.0 / 12.0 - f * (1.0 / 120.0 - f * (1.0 / 252.0 - f * (1.0 / 240.0 - f * (1.0 / 132.0)))));
}

enum VarClass { VC_CONST = 0, VC_DATA = 1, VC_DERIVED = 2, VC_PREC = 3, VC_GAUSS = 4, VC_CAT = 5, VC_DIR = 6 };   // (CAT: the switch of a mixture node; DIR: its probability vector)
enum NodeClass { NC_NOISE = 0, NC_MUL = 1, NC_ADD = 2, NC_PRIOR = 3, NC_SKIP = 4, NC_GCVZ = 5 };   // (GCVZ: the z side of a GCV node)   // (SKIP: NormalMixture / Categorical / Dirichlet — they act through virtual nodes and state ops)

struct Program {
    int dmax = 1;
    std::vector<int> ops, aux, lvl_ptr;
    std::vector<double> cpool;
    long long msg_doubles = 0, marg_doubles = 0, val_doubles = 0, data_doubles = 0, prec_doubles = 0, term_slots = 0, stat_doubles = 0;
    int fe_root = -1;
    int fe_level = 0;   // first level of the second phase (Bethe terms, residual moments, q(W) updates, sums)
    int n_ops = 0, n_levels = 0, n_messages = 0;
    std::vector<int> dim, vclass, marg_off, val_off, prec_off;
    std::vector<int> discrete_k;   // per variable: components of a switch / probability vector (else 0)
    std::vector<int64_t> data_vars;
    std::vector<double> prec_init;   // [prec_doubles] initial state (replica-independent)
    uint64_t rule_calls = 0, products = 0, marginals = 0;
    long long bytes_per_sweep = 0;
    int max_width = 0;
    // the strand schedule of the sweep phase (build_strands): the ops again, strand by strand, with register inputs and suppressed stores
    std::vector<int> sops, strands, slvl_ptr;   // [n_sweep_ops][OP_WORDS]; [n_strands][2] = (first op, ops); strands of level l: slvl_ptr[l] … slvl_ptr[l + 1]
    long long bytes_per_sweep_strands = 0;      // message bytes this schedule moves through HBM (register hand-overs left out)
    long long io_bytes = 0;                     // what has to move whatever the schedule: the data in, the posteriors of the named variables out
    long long fe_bytes = 0;                     // the second phase's reads and writes per replica
    int longest_strand = 0;
    bool has_gcv = false;               // GCV nodes: lane-per-item kernels only
    bool has_valnoise = false;          // scalar Gaussian nodes with a data-valued variance / precision: lane-per-item kernels only
    bool has_mix = false;               // NormalMixture nodes: q(z), q(s) live in the precision-state array; lane-per-item kernels only (dimensions ≤ 8)
    bool has_mf = false;                // some Gaussian node runs under q(out) q(μ): the marginals of its interfaces are STATE (start: the @initialization marginals)
    std::vector<double> marg_init;      // [marg_doubles] when has_mf
    bool fe_heavy = false;   // the second phase holds OP_FE_ADD2 or OP_PREC_UPDATE ops (else the light kernel instance runs it)
    int n_push = 0, lazy_level = -1;   // marginals stored as images of other marginals (OP_MARG_PUSH): a level of their own behind everything a run executes, launched on demand
    std::vector<char> is_push;         // per variable
};

struct Compiler {
    const rxhip_graph_desc* g;
    Program& P;
    int64_t nv, nf;
    std::vector<int64_t> iptr;          // CSR of factor interfaces
    const int64_t* ifv;
    std::vector<int> nclass;
    std::vector<char> mf;   // per factor: a Gaussian node the model's constraints run under q(out) q(μ) — mean field between its two Gaussian interfaces
    // NormalMixture (out, switch, m[1..K], p[1..K]) under MeanField(): the node's log-factor is Σ_k z_k log N(out | m_k, p_k⁻¹), so toward (out, m_k, p_k) it acts as K
    // Gaussian precision nodes each WEIGHTED by π_k = q(z = k).  The compiler appends those K virtual three-interface nodes behind the real factors (everything
    // downstream — leaf rules, Bethe terms, residual moments, entropy bookkeeping — sees ordinary Gaussian nodes carrying a weight) and adds two state ops: q(z) of
    // every switch before the sweep (OP_CAT_UPDATE, from the marginals of the previous iteration) and q(s) with the switches' terms behind it (OP_DIR_UPDATE).
    // Schedule and arithmetic: the mixture engine's (vmp_engines.hip: q(z) from the previous marginals, then the Gaussian sweep and q(s), then q(p) with the
    // new q(m)), node by node instead of summed over the data set; tests/test_tree_mixture_gpu.py holds the two together.
    std::vector<int64_t> xifv;          // interfaces of the real factors, then of the virtual ones
    std::vector<int> xtype;             // node types alike
    std::vector<int> wz, wk;            // per factor: the switch variable and component that weight it (−1)
    // GCV(y, x, z, κ, ω) under q(y, x) q(z), κ and ω constants (test/models/statespace/hgf_tests.jl:28-35): toward (y, x) a scalar Gaussian precision node whose precision
    // γ(z) = exp(−(κ z + ω)) lives in a STATE slot (E γ, E log γ of the previous iteration's q(z)) — virtual node `fa` — and toward z a message op of its own — virtual
    // node `fz` (tree_kernels.hpp OP_GCV_Z): the joint of (y, x), ψ, the cubature-matched q(z) and the Gaussian moments of the ELQ message for z's other neighbours
    struct Gcv { int node, y, x, z, kv, ov, fa, fz, state, stat; };
    std::vector<Gcv> gcvs;
    std::vector<int> gcv_of;            // per factor: index into gcvs for its two virtual nodes (−1)
    std::vector<char> gcv_z;            // per variable: the volatility input of a GCV node
    struct Mix { int node, out, z, K; std::vector<int> m, p; };
    std::vector<Mix> mixes;
    std::vector<int> cat_s, dir_a, catK;   // per variable: a switch's probability-vector variable; a Dirichlet variable's concentration constant (≤ −2: Beta, alpha_pool); components
    std::vector<double> alpha_pool;
    std::vector<int> alpha_off_;           // per Dirichlet / Beta variable: constant-pool offset of its prior's concentrations (−1)
    const double* alpha0(int sv) const { return dir_a[sv] <= -2 ? alpha_pool.data() + (-2 - dir_a[sv]) : cptr(dir_a[sv]); }
    int alpha_off(int sv) {
        if (alpha_off_.empty()) alpha_off_.assign(nv, -1);
        if (alpha_off_[sv] < 0) {
            alpha_off_[sv] = (int)P.cpool.size();
            P.cpool.insert(P.cpool.end(), alpha0(sv), alpha0(sv) + catK[sv]);
        }
        return alpha_off_[sv];
    }
    int ftype(int f) const { return xtype[f]; }
    // edges: (factor, interface) with a Gaussian variable
    struct Edge { int f, k, v; };
    std::vector<Edge> edges;
    std::vector<std::vector<int>> var_edges;      // per variable: its edges in factor order
    std::vector<std::vector<int>> fac_edges;      // per factor: edge id per interface (−1)
    // message m: f2v(e) = e, v2f(e) = E + e
    int E = 0;
    std::vector<char> null_, needed, form, done;
    std::vector<int> alias, off, level;
    std::vector<std::vector<int>> deps, users;
    std::vector<int> cval_off, cmat_off, noise_off_cov, noise_off_prec, prior_off;   // constant-pool offsets per variable (−1)
    struct OpRec { int level; int w[OP_WORDS]; };
    std::vector<OpRec> recs;

    Compiler(const rxhip_graph_desc* g_, Program& p) : g(g_), P(p) {}

    int64_t iface(int f, int k) const { return ifv[iptr[f] + k]; }
    int n_iface(int f) const { return (int)(iptr[f + 1] - iptr[f]); }
    bool clamped(int v) const { return P.vclass[v] == VC_CONST || P.vclass[v] == VC_DATA || P.vclass[v] == VC_DERIVED; }
    int msz(int d) const { return d + d * (d + 1) / 2; }

    const double* cptr(int v) const { return g->const_pool + g->var_const[v]; }
    int const_value(int v) {   // vector value of a constant variable in the pool
        if (cval_off[v] < 0) {
            cval_off[v] = (int)P.cpool.size();
            const int n = g->var_rows[v] * g->var_cols[v];
            P.cpool.insert(P.cpool.end(), cptr(v), cptr(v) + n);
        }
        return cval_off[v];
    }
    int const_matrix(int v, int rows, int cols) {
        if ((int64_t)g->var_rows[v] * g->var_cols[v] != (int64_t)rows * cols) fail(RXHIP_ERR_BADARG, "constant %d: %d x %d expected", v, rows, cols);
        return const_value(v);
    }
    int noise_block(int v, int d, bool is_precision) {   // Σ | W | log|W|; memoised per (constant, parametrisation): a constant that one node reads as a covariance and another as a precision gets two blocks
        std::vector<int>& noise_off = is_precision ? noise_off_prec : noise_off_cov;
        if (noise_off[v] >= 0) return noise_off[v];
        if ((int64_t)g->var_rows[v] * g->var_cols[v] != (int64_t)d * d) fail(RXHIP_ERR_BADARG, "noise parameter %d: %d x %d expected", v, d, d);
        std::vector<double> M(cptr(v), cptr(v) + (size_t)d * d), Mi((size_t)d * d);
        double amax = 0.0, asym = 0.0;
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) {
                amax = std::max(amax, std::fabs(M[i * d + j]));
                asym = std::max(asym, std::fabs(M[i * d + j] - M[j * d + i]));
            }
        // (the bound of the state-space lowering, graph_lowering.hpp spd_inverse_checked: a precision computed as inv(Σ) on the host carries eps·cond·max|W|
        //  of asymmetry; below it the symmetric part is what gets factorised)
        if (!(asym <= 1e-8 * amax)) fail(RXHIP_ERR_NOT_POSDEF, "noise parameter %d is not symmetric", v);
        if (asym > 0.0) rxhip_lower::last_asymmetry() = std::max(rxhip_lower::last_asymmetry(), asym / amax);
        double ld = 0.0;
        if (!host_chol_inv(d, M.data(), Mi.data(), &ld)) fail(RXHIP_ERR_NOT_POSDEF, "noise parameter %d is not positive definite", v);
        noise_off[v] = (int)P.cpool.size();
        const std::vector<double>&Sg = is_precision ? Mi : M, &W = is_precision ? M : Mi;
        P.cpool.insert(P.cpool.end(), Sg.begin(), Sg.end());
        P.cpool.insert(P.cpool.end(), W.begin(), W.end());
        P.cpool.push_back(is_precision ? ld : -ld);
        return noise_off[v];
    }

    void expand_mixtures() {
        wz.assign((size_t)nf, -1); wk.assign((size_t)nf, -1);
        cat_s.assign((size_t)nv, -1); dir_a.assign((size_t)nv, -1); catK.assign((size_t)nv, 0);
        const int64_t nf0 = nf;
        gcv_of.assign((size_t)nf, -1);
        gcv_z.assign((size_t)nv, 0);
        bool any = false;
        for (int64_t f = 0; f < nf0; ++f) any = any || xtype[f] == RXHIP_NODE_NORMAL_MIXTURE || xtype[f] == RXHIP_NODE_GCV;
        if (!any) return;
        xifv.assign(g->factor_iface, g->factor_iface + iptr[nf0]);
        for (int64_t f = 0; f < nf0; ++f) {
            if (xtype[f] != RXHIP_NODE_GCV) continue;
            if (n_iface((int)f) != 5) fail(RXHIP_ERR_BADARG, "factor %lld (GCV): interfaces (y, x, z, κ, ω) expected", (long long)f);
            Gcv gc{(int)f, (int)iface((int)f, 0), (int)iface((int)f, 1), (int)iface((int)f, 2), (int)iface((int)f, 3), (int)iface((int)f, 4), 0, 0, -1, -1};
            for (int v : {gc.y, gc.x, gc.z, gc.kv, gc.ov})
                if (g->var_rows[v] != 1 || g->var_cols[v] != 1) fail(RXHIP_ERR_UNSUPPORTED, "factor %lld (GCV): scalar interfaces expected", (long long)f);
            if (gcv_z[gc.z]) fail(RXHIP_ERR_UNSUPPORTED, "factor %lld (GCV): its volatility input drives another GCV node as well", (long long)f);
            gcv_z[gc.z] = 1;
            gc.fa = (int)xtype.size();
            xtype.push_back(RXHIP_NODE_NORMAL_MEAN_PRECISION);
            xifv.push_back(gc.y); xifv.push_back(gc.x); xifv.push_back(gc.z);
            iptr.push_back((int64_t)xifv.size());
            gc.fz = (int)xtype.size();
            xtype.push_back(RXHIP_NODE_GCV);
            xifv.push_back(gc.z); xifv.push_back(gc.y); xifv.push_back(gc.x);
            iptr.push_back((int64_t)xifv.size());
            for (int k = 0; k < 2; ++k) { wz.push_back(-1); wk.push_back(-1); gcv_of.push_back((int)gcvs.size()); }
            gcvs.push_back(gc);
            P.has_gcv = true;
        }
        for (int64_t f = 0; f < nf0; ++f) {
            if (xtype[f] != RXHIP_NODE_NORMAL_MIXTURE) continue;
            const int n = n_iface((int)f), K = (n - 2) / 2;
            if (K < 1 || n != 2 + 2 * K) fail(RXHIP_ERR_BADARG, "factor %lld (NormalMixture): interfaces (out, switch, m[1..K], p[1..K]) expected", (long long)f);
            Mix mx{(int)f, (int)iface((int)f, 0), (int)iface((int)f, 1), K, {}, {}};
            const int d = g->var_rows[mx.out];
            for (int k = 0; k < K; ++k) {
                mx.m.push_back((int)iface((int)f, 2 + k));
                mx.p.push_back((int)iface((int)f, 2 + K + k));
                xtype.push_back(d == 1 ? RXHIP_NODE_NORMAL_MEAN_PRECISION : RXHIP_NODE_MVNORMAL_MEAN_PRECISION);
                xifv.push_back(mx.out); xifv.push_back(mx.m[k]); xifv.push_back(mx.p[k]);
                iptr.push_back((int64_t)xifv.size());
                wz.push_back(mx.z); wk.push_back(k); gcv_of.push_back(-1);
            }
            mixes.push_back(std::move(mx));
        }
        nf = (int64_t)xtype.size();
        ifv = xifv.data();
        P.has_mix = !mixes.empty();
    }
    void classify_mixtures() {
        for (int64_t f = 0; f < nf; ++f) {
            const int t = ftype((int)f);
            if (t == RXHIP_NODE_CATEGORICAL || (t == RXHIP_NODE_BERNOULLI && P.has_mix)) {   // (Bernoulli(s): the two-component spelling, z = true the FIRST component)
                if (n_iface((int)f) != 2) fail(RXHIP_ERR_BADARG, "factor %lld (Categorical): interfaces (out, p) expected", (long long)f);
                const int z = (int)iface((int)f, 0), sv = (int)iface((int)f, 1);
                if (g->var_kind[z] != RXHIP_VARKIND_RANDOM || cat_s[z] >= 0) fail(RXHIP_ERR_UNSUPPORTED, "factor %lld (Categorical): the output must be a random variable with this one prior", (long long)f);
                cat_s[z] = sv;
                P.vclass[z] = VC_CAT;
            } else if (t == RXHIP_NODE_DIRICHLET) {
                if (n_iface((int)f) != 2) fail(RXHIP_ERR_BADARG, "factor %lld (Dirichlet): interfaces (out, a) expected", (long long)f);
                const int sv = (int)iface((int)f, 0), a = (int)iface((int)f, 1);
                if (g->var_kind[sv] != RXHIP_VARKIND_RANDOM || dir_a[sv] != -1 || P.vclass[a] != VC_CONST)
                    fail(RXHIP_ERR_UNSUPPORTED, "factor %lld (Dirichlet): a random output with this one prior and a constant concentration expected", (long long)f);
                dir_a[sv] = a;
                P.vclass[sv] = VC_DIR;
                catK[sv] = g->var_rows[a] * g->var_cols[a];
                for (int k = 0; k < catK[sv]; ++k)
                    if (!(cptr(a)[k] > 0.0)) fail(RXHIP_ERR_BADARG, "factor %lld (Dirichlet): concentrations must be positive", (long long)f);
            } else if (t == RXHIP_NODE_BETA && P.has_mix) {   // Beta(a, b) on the probability of the first component = Dirichlet([a, b])
                if (n_iface((int)f) != 3) fail(RXHIP_ERR_BADARG, "factor %lld (Beta): interfaces (out, a, b) expected", (long long)f);
                const int sv = (int)iface((int)f, 0), a = (int)iface((int)f, 1), b = (int)iface((int)f, 2);
                if (g->var_kind[sv] != RXHIP_VARKIND_RANDOM || dir_a[sv] != -1 || P.vclass[a] != VC_CONST || P.vclass[b] != VC_CONST || !(cptr(a)[0] > 0.0) || !(cptr(b)[0] > 0.0))
                    fail(RXHIP_ERR_UNSUPPORTED, "factor %lld (Beta): a random output with this one prior and positive constant parameters expected", (long long)f);
                dir_a[sv] = (int)alpha_pool.size();
                alpha_pool.push_back(cptr(a)[0]);
                alpha_pool.push_back(cptr(b)[0]);
                dir_a[sv] = -2 - dir_a[sv];   // (≤ −2: an entry of alpha_pool, not a constant variable)
                P.vclass[sv] = VC_DIR;
                catK[sv] = 2;
            }
        }
        if (!mixes.empty() && g->allow_missing) fail(RXHIP_ERR_UNSUPPORTED, "`missing` observations in a graph with NormalMixture nodes have no schedule here (allow_missing)");
        for (const Mix& mx : mixes) {
            if (P.vclass[mx.z] != VC_CAT) fail(RXHIP_ERR_UNSUPPORTED, "factor %d (NormalMixture): the switch must be a random variable with a Categorical prior", mx.node);
            if (catK[mx.z] != 0) fail(RXHIP_ERR_UNSUPPORTED, "factor %d (NormalMixture): its switch drives another mixture node as well", mx.node);
            catK[mx.z] = mx.K;
            const int sv = cat_s[mx.z];
            if (P.vclass[sv] == VC_DIR) {
                if (catK[sv] != mx.K) fail(RXHIP_ERR_BADARG, "factor %d (NormalMixture): %d components, a probability vector of %d", mx.node, mx.K, catK[sv]);
            } else if (P.vclass[sv] == VC_CONST) {
                if (g->var_rows[sv] * g->var_cols[sv] != mx.K) fail(RXHIP_ERR_BADARG, "factor %d (NormalMixture): %d components, a probability vector of another length", mx.node, mx.K);
                for (int k = 0; k < mx.K; ++k)
                    if (!(cptr(sv)[k] > 0.0)) fail(RXHIP_ERR_BADARG, "factor %d (NormalMixture): constant switch probabilities must be positive", mx.node);
            } else
                fail(RXHIP_ERR_UNSUPPORTED, "factor %d (NormalMixture): the switch's probability vector must be a constant or a Dirichlet variable", mx.node);
        }
        P.discrete_k = catK;
        for (int64_t v = 0; v < nv; ++v)
            if (P.vclass[v] == VC_CAT && catK[v] == 0) fail(RXHIP_ERR_UNSUPPORTED, "variable %lld: a Categorical variable that is not the switch of a NormalMixture node has no schedule here", (long long)v);
    }

    void parse() {
        nv = g->n_variables;
        nf = g->n_factors;
        if (nv <= 0 || nf <= 0 || !g->var_kind || !g->var_rows || !g->var_cols || !g->var_const || !g->factor_type || !g->factor_iface)
            fail(RXHIP_ERR_BADARG, "graph descriptor: missing tables");
        if (nv > (1ll << 28) || nf > (1ll << 28)) fail(RXHIP_ERR_UNSUPPORTED, "graph too large for the executor's 32-bit tables");
        iptr.resize(nf + 1);
        for (int64_t f = 0; f <= nf; ++f) iptr[f] = g->factor_iface_ptr ? g->factor_iface_ptr[f] : 3 * f;
        ifv = g->factor_iface;
        xtype.assign(g->factor_type, g->factor_type + nf);
        for (int64_t f = 0; f < nf; ++f)
            for (int k = 0; k < n_iface((int)f); ++k)
                if (iface((int)f, k) < 0 || iface((int)f, k) >= nv) fail(RXHIP_ERR_BADARG, "factor %lld: interface %d names no variable", (long long)f, k);
        expand_mixtures();
        P.dim.assign(nv, 0);
        P.vclass.assign(nv, VC_GAUSS);
        nclass.assign(nf, -1);
        for (int64_t v = 0; v < nv; ++v) {
            P.dim[v] = g->var_rows[v];
            const int k = g->var_kind[v];
            if (k == RXHIP_VARKIND_CONST) {
                P.vclass[v] = VC_CONST;
                if (g->var_const[v] < 0 || g->var_const[v] + (int64_t)g->var_rows[v] * g->var_cols[v] > g->n_const) fail(RXHIP_ERR_BADARG, "constant %lld: value outside the pool", (long long)v);
            } else if (k == RXHIP_VARKIND_DATA) P.vclass[v] = VC_DATA;
            else if (k != RXHIP_VARKIND_RANDOM) fail(RXHIP_ERR_BADARG, "variable %lld: unknown kind %d", (long long)v, k);
            if (g->var_rows[v] < 1) fail(RXHIP_ERR_BADARG, "variable %lld: no rows", (long long)v);
        }
        for (int64_t f = 0; f < nf; ++f) {
            const int t = ftype((int)f);
            if (t == RXHIP_NODE_NORMAL_MIXTURE || t == RXHIP_NODE_CATEGORICAL || t == RXHIP_NODE_DIRICHLET || ((t == RXHIP_NODE_BERNOULLI || t == RXHIP_NODE_BETA) && P.has_mix)) { nclass[f] = NC_SKIP; continue; }
            if (t == RXHIP_NODE_GCV) { nclass[f] = gcv_of[f] >= 0 ? NC_GCVZ : NC_SKIP; continue; }   // (the node itself; its z side)
            switch (t) {
            case RXHIP_NODE_MVNORMAL_MEAN_COV: case RXHIP_NODE_NORMAL_MEAN_VARIANCE: case RXHIP_NODE_MVNORMAL_MEAN_PRECISION: case RXHIP_NODE_NORMAL_MEAN_PRECISION:
                nclass[f] = NC_NOISE; break;
            case RXHIP_NODE_MULTIPLY: nclass[f] = NC_MUL; break;
            case RXHIP_NODE_ADD: nclass[f] = NC_ADD; break;
            case RXHIP_NODE_WISHART: case RXHIP_NODE_GAMMA_SHAPE_RATE: case RXHIP_NODE_GAMMA_SHAPE_SCALE: nclass[f] = NC_PRIOR; break;
            default: fail(RXHIP_ERR_UNSUPPORTED, "node type %d is outside the Gaussian tree family of the node-array executor", t);
            }
            if (n_iface((int)f) != 3) fail(RXHIP_ERR_BADARG, "factor %lld: three interfaces expected", (long long)f);
        }
        // the factorisation the model's constraints ask of every node against the one this schedule implements (q(out, μ) q(W) on Gaussian nodes, joint
        // deterministic nodes): a mismatch is refused with the node named — never answered with the other variational family's posterior
        if (rxhip_lower::check_factorisation(g, &mf)) fail(RXHIP_ERR_UNSUPPORTED, "%s", rxhip_lower::last_error().c_str());
        mf.resize((size_t)nf, 1);   // (the components of a mixture node: mean field between `out` and the mean wherever both are random)
        for (const Gcv& gc : gcvs) mf[gc.fa] = 0;   // (q(y, x): structured)
        classify_mixtures();
        // precision variables
        for (int64_t f = 0; f < nf; ++f)
            if (nclass[f] == NC_PRIOR) {
                const int v = (int)iface((int)f, 0);
                if (g->var_kind[v] != RXHIP_VARKIND_RANDOM) fail(RXHIP_ERR_UNSUPPORTED, "a Wishart / Gamma node with a clamped output has no schedule");
                if (P.vclass[v] == VC_PREC) fail(RXHIP_ERR_UNSUPPORTED, "precision variable %d has two priors", v);
                P.vclass[v] = VC_PREC;
                for (int k = 1; k < 3; ++k)
                    if (P.vclass[iface((int)f, k)] != VC_CONST) fail(RXHIP_ERR_UNSUPPORTED, "Wishart / Gamma priors need constant parameters");
            }
        // derived clamped values
        bool changed = true;
        while (changed) {
            changed = false;
            for (int64_t f = 0; f < nf; ++f) {
                if (nclass[f] != NC_MUL && nclass[f] != NC_ADD) continue;
                const int o = (int)iface((int)f, 0);
                if (P.vclass[o] != VC_GAUSS) continue;
                const int a = (int)iface((int)f, 1), b = (int)iface((int)f, 2);
                if (clamped(a) && clamped(b)) { P.vclass[o] = VC_DERIVED; changed = true; }
            }
        }
    }

    void check_family() {
        int dmx = 1;
        for (int64_t f = 0; f < nf; ++f) {
            const int t = ftype((int)f), a = (int)iface((int)f, 0), b = (int)iface((int)f, 1), c = (int)iface((int)f, 2);
            if (nclass[f] == NC_NOISE) {
                const bool prec_node = t == RXHIP_NODE_MVNORMAL_MEAN_PRECISION || t == RXHIP_NODE_NORMAL_MEAN_PRECISION;
                if (P.vclass[a] == VC_PREC || P.vclass[b] == VC_PREC) fail(RXHIP_ERR_UNSUPPORTED, "a precision variable on a Gaussian node's out / mean interface");
                if (P.dim[a] != P.dim[b]) fail(RXHIP_ERR_BADARG, "factor %lld: out and mean differ in dimension", (long long)f);
                if (gcv_of[f] >= 0) {
                    // γ(z) of a GCV node: a state slot, not a variable
                } else if (P.vclass[c] == VC_PREC) {
                    if (!prec_node) fail(RXHIP_ERR_UNSUPPORTED, "a random covariance has no rule here (precision-parametrised nodes only)");
                    if (P.dim[c] != P.dim[a]) fail(RXHIP_ERR_BADARG, "factor %lld: precision variable of another dimension", (long long)f);
                } else if (P.vclass[c] == VC_DATA && P.dim[a] == 1 && g->var_rows[c] * g->var_cols[c] == 1) {
                    P.has_valnoise = true;   // a scalar node whose variance / precision arrives with the data (`Normal(mean = m_prev, var = v_prev)` of @autoupdates)
                } else if (P.vclass[c] != VC_CONST) fail(RXHIP_ERR_UNSUPPORTED, "factor %lld: the third interface of a Gaussian node must be a constant, a Wishart / Gamma variable or (scalar nodes) a data variable", (long long)f);
                dmx = std::max(dmx, P.dim[a]);
            } else if (nclass[f] == NC_MUL) {
                if (P.vclass[b] != VC_CONST) fail(RXHIP_ERR_UNSUPPORTED, "factor %lld: `*` needs a constant matrix", (long long)f);
                if (P.vclass[a] == VC_PREC || P.vclass[c] == VC_PREC) fail(RXHIP_ERR_UNSUPPORTED, "`*` on a precision variable");
                if ((int64_t)g->var_rows[b] * g->var_cols[b] != (int64_t)P.dim[a] * P.dim[c]) fail(RXHIP_ERR_BADARG, "factor %lld: matrix is not %d x %d", (long long)f, P.dim[a], P.dim[c]);
                if (P.vclass[a] == VC_GAUSS && P.vclass[c] != VC_GAUSS) fail(RXHIP_ERR_BADARG, "factor %lld: `*` of a clamped input with a random output", (long long)f);
                if (clamped(a) && P.vclass[a] != VC_DERIVED) fail(RXHIP_ERR_UNSUPPORTED, "factor %lld: `*` with an observed output", (long long)f);
                dmx = std::max(dmx, std::max(P.dim[a], P.dim[c]));
            } else if (nclass[f] == NC_ADD) {
                if (P.dim[a] != P.dim[b] || P.dim[a] != P.dim[c]) fail(RXHIP_ERR_BADARG, "factor %lld: `+` of different dimensions", (long long)f);
                for (int v : {a, b, c})
                    if (P.vclass[v] == VC_PREC) fail(RXHIP_ERR_UNSUPPORTED, "`+` on a precision variable");
                if (clamped(a) && P.vclass[a] != VC_DERIVED) fail(RXHIP_ERR_UNSUPPORTED, "factor %lld: `+` with an observed output", (long long)f);
                dmx = std::max(dmx, P.dim[a]);
            }
        }
        if (dmx > wave::DMAX_WAVE) fail(RXHIP_ERR_UNSUPPORTED, "the node-array executor runs dimensions <= %d (this graph: %d)", wave::DMAX_WAVE, dmx);
        // q(out) q(μ) means something only where both interfaces are random Gaussian variables (a clamped one is a cluster of its own anyway)
        for (int64_t f = 0; f < nf; ++f) {
            mf[f] = mf[f] && nclass[f] == NC_NOISE && P.vclass[iface((int)f, 0)] == VC_GAUSS && P.vclass[iface((int)f, 1)] == VC_GAUSS;
            P.has_mf = P.has_mf || mf[f];
        }
        if (P.has_mix && dmx > 8) fail(RXHIP_ERR_UNSUPPORTED, "NormalMixture nodes run on the lane-per-item kernels: dimensions <= 8 (this graph: %d)", dmx);
        if (P.has_gcv && dmx > 8) fail(RXHIP_ERR_UNSUPPORTED, "GCV nodes run on the lane-per-item kernels: dimensions <= 8 (this graph: %d)", dmx);
        for (const Gcv& gc : gcvs) {
            if (P.vclass[gc.kv] != VC_CONST || P.vclass[gc.ov] != VC_CONST) fail(RXHIP_ERR_UNSUPPORTED, "factor %d (GCV): κ and ω must be constants", gc.node);
            for (int v : {gc.y, gc.x, gc.z})
                if (P.vclass[v] != VC_GAUSS) fail(RXHIP_ERR_UNSUPPORTED, "factor %d (GCV): y, x and z must be random Gaussian variables", gc.node);
        }
        if (P.has_valnoise && dmx > 8) fail(RXHIP_ERR_UNSUPPORTED, "data-valued variances run on the lane-per-item kernels: dimensions <= 8 (this graph: %d)", dmx);
        P.has_mf = P.has_mf || P.has_mix;   // (the switch's rule reads the marginals of the means of the previous iteration: marginals are state)
        // <= 8: the register instances (lane per op and replica); above: the graph's own maximum, staged in LDS by a wavefront per op and replica
        P.dmax = dmx <= 1 ? 1 : dmx <= 2 ? 2 : dmx <= 4 ? 4 : dmx <= 8 ? 8 : dmx;
    }

    void build_edges() {
        for (const Mix& mx : mixes) {
            for (int k = 0; k < mx.K; ++k) {
                if (P.vclass[mx.m[k]] == VC_PREC || P.vclass[mx.m[k]] == VC_CAT || P.vclass[mx.m[k]] == VC_DIR) fail(RXHIP_ERR_UNSUPPORTED, "factor %d (NormalMixture): m[%d] is not a Gaussian variable", mx.node, k + 1);
                if (P.dim[mx.m[k]] != P.dim[mx.out] || P.dim[mx.p[k]] != P.dim[mx.out]) fail(RXHIP_ERR_BADARG, "factor %d (NormalMixture): component %d differs in dimension from `out`", mx.node, k + 1);
            }
            if (P.vclass[mx.out] == VC_PREC || P.vclass[mx.out] == VC_CAT || P.vclass[mx.out] == VC_DIR) fail(RXHIP_ERR_UNSUPPORTED, "factor %d (NormalMixture): `out` is not a Gaussian variable or a value", mx.node);
        }
        var_edges.assign(nv, {});
        fac_edges.assign(nf, std::vector<int>(3, -1));
        std::vector<int> uf(nv + 2 * nf);   // (a node under q(out) q(μ) is two leaf factors as far as cycles go: its interfaces do not exchange messages)
        std::iota(uf.begin(), uf.end(), 0);
        auto find = [&](int x) { while (uf[x] != x) { uf[x] = uf[uf[x]]; x = uf[x]; } return x; };
        for (int64_t f = 0; f < nf; ++f) {
            if (nclass[f] == NC_PRIOR || nclass[f] == NC_SKIP) continue;
            for (int k = 0; k < 3; ++k) {
                if (k == 1 && nclass[f] == NC_MUL) continue;
                if (k == 2 && nclass[f] == NC_NOISE) continue;
                if (k != 0 && nclass[f] == NC_GCVZ) continue;
                const int v = (int)iface((int)f, k);
                if (P.vclass[v] != VC_GAUSS) continue;
                const int ra = find(v), rb = find((int)(nv + f + ((mf[f] && k == 1) ? nf : 0)));
                if (ra == rb) fail(RXHIP_ERR_UNSUPPORTED, "the Gaussian variables do not form a tree (a cycle through variable %d): loopy graphs have no exact schedule", v);
                uf[ra] = rb;
                fac_edges[f][k] = (int)edges.size();
                var_edges[v].push_back((int)edges.size());
                edges.push_back({(int)f, k, v});
            }
        }
        E = (int)edges.size();
        bool any_prec = false;
        for (int64_t v = 0; v < nv; ++v) any_prec = any_prec || P.vclass[v] == VC_PREC;
        // (no Gaussian variable but a precision variable: `y[i] ~ MvNormal(μ = m, Λ = P)` with a KNOWN mean, test/models/iid/mv_iid_precision_known_mean_tests.jl —
        //  no message at all, the schedule is the nodes' residual moments, the q(W) updates and the Bethe sum)
        if (E == 0 && !any_prec) fail(RXHIP_ERR_UNSUPPORTED, "no random variable of the Gaussian / Wishart / Gamma family in the graph");
    }

    // ---- dependencies of every message ----
    void build_deps() {
        deps.assign(2 * E, {});
        for (int e = 0; e < E; ++e) {
            const Edge& ed = edges[e];
            const int f = ed.f;
            auto other = [&](int k) { return fac_edges[f][k]; };
            if (nclass[f] == NC_GCVZ) {   // the joint of (y, x) from their messages into the node, and the Gaussian message into z the product is matched against
                const Gcv& gc = gcvs[gcv_of[f]];
                for (int k = 0; k < 2; ++k) {
                    if (fac_edges[gc.fa][k] < 0) fail(RXHIP_ERR_UNSUPPORTED, "factor %d (GCV): y and x must be random variables", gc.node);
                    deps[e].push_back(E + fac_edges[gc.fa][k]);
                }
            } else if (nclass[f] == NC_NOISE) {
                if (!mf[f] && other(1 - ed.k) >= 0) deps[e].push_back(E + other(1 - ed.k));   // (mean field: the rule reads the other interface's MARGINAL, not its message)
            } else if (nclass[f] == NC_MUL) {
                const int o = other(ed.k == 0 ? 2 : 0);
                if (o >= 0) deps[e].push_back(E + o);
                else if (ed.k == 2) fail(RXHIP_ERR_BADARG, "internal: `*` toward its input without a random output");
            } else {
                for (int k = 0; k < 3; ++k)
                    if (k != ed.k && other(k) >= 0) deps[e].push_back(E + other(k));
            }
            for (int e2 : var_edges[ed.v])
                if (e2 != e) deps[E + e].push_back(e2);
        }
        users.assign(2 * E, {});
        for (int m = 0; m < 2 * E; ++m)
            for (int dpm : deps[m]) users[dpm].push_back(m);
    }

    // what the rule that consumes v2f(e) reads it as: 0 moment, 1 precision, 2 either
    int wanted_form(int e) const {
        const Edge& ed = edges[e];
        if (nclass[ed.f] == NC_MUL) return ed.k == 2 ? 0 : 1;   // the message from `in` feeds (:out) in moment form; from `out` feeds (:in) in precision form
        if (nclass[ed.f] == NC_ADD) {
            int ng = 0;
            for (int k = 0; k < 3; ++k) ng += fac_edges[ed.f][k] >= 0;
            return ng == 3 ? 0 : 2;
        }
        return 2;
    }
    bool factor_uses_v2f(int f) const {
        if (nclass[f] == NC_GCVZ) return true;
        if (mf[f]) return false;
        int ng = 0;
        for (int k = 0; k < 3; ++k) ng += fac_edges[f][k] >= 0;
        return ng >= 2;
    }

    // ---- hubs: a variable of degree n > HUB_MIN whose neighbours each need their own product of the n − 1 other messages ----
    // Reduced one by one that is O(n²) partial products (a star of 3 000 leaves: 440 000 ops).  Instead the inbound messages sit at the leaves of ONE
    // FAN_IN-ary tree of partial products (positions = the variable's edge order; a node's product is built once, when first asked for), and the product that
    // leaves out position j is the product of the SIBLINGS along j's path to the root: ≤ (FAN_IN − 1)·height inputs.  O(n) ops for all n products; the
    // marginal is the product of the top nodes.  (Products in precision form are sums: the fold order changes the rounding, not the result.)
    static constexpr int HUB_MIN = 16;
    struct Hub {
        int height = 0;                                   // node heights 0 (leaves = edges) … height − 1 (children of the root)
        std::vector<std::vector<int>> lvl;                // [height][node]: −2 not yet known, −1 empty, else the level of the node's product
        std::vector<std::vector<std::pair<int, int>>> slot;   // [height][node]: (offset, form) once emitted; offset −2 not yet, −1 empty
        std::vector<int> pos_of_edge;                     // edge id → position (sparse: by edge id − first edge … kept as a map below)
    };
    std::vector<int> hub_of;                              // per variable: index into hubs, −1
    std::vector<Hub> hubs;
    std::vector<int> edge_pos;                            // per edge: its position among its variable's edges
    bool is_hub(int v) const { return hub_of[v] >= 0; }
    void build_hubs() {
        hub_of.assign(nv, -1);
        edge_pos.assign(E, 0);
        for (int64_t v = 0; v < nv; ++v) {
            const int n = (int)var_edges[v].size();
            for (int i = 0; i < n; ++i) edge_pos[var_edges[v][i]] = i;
            if (n <= HUB_MIN) continue;
            Hub h;
            int width = n;
            while (width > 1) {
                h.lvl.push_back(std::vector<int>((size_t)width, -2));
                h.slot.push_back(std::vector<std::pair<int, int>>((size_t)width, {-2, 0}));
                width = (width + FAN_IN - 1) / FAN_IN;
                ++h.height;
            }
            hub_of[v] = (int)hubs.size();
            hubs.push_back(std::move(h));
        }
    }
    // level of the product of node (k, idx) of variable v's tree (−1: no live message below it); every live leaf below it has been analysed
    int hub_node_level(int v, int k, int idx) {
        Hub& h = hubs[hub_of[v]];
        int& memo = h.lvl[k][idx];
        if (memo != -2) return memo;
        if (k == 0) {
            const int e = var_edges[v][idx];
            return memo = (null_[e] ? -1 : level[e]);
        }
        int cnt = 0, mx = -1;
        const int nchild = (int)h.lvl[k - 1].size();
        for (int c = idx * FAN_IN; c < std::min(nchild, (idx + 1) * FAN_IN); ++c) {
            const int l = hub_node_level(v, k - 1, c);
            if (l >= 0) { ++cnt; mx = std::max(mx, l); }
        }
        return memo = (cnt == 0 ? -1 : cnt == 1 ? mx : mx + 1);
    }
    // the siblings along position j's path: (height, node) pairs
    template <class F>
    void hub_siblings(int v, int j, F f) {
        const Hub& h = hubs[hub_of[v]];
        int idx = j;
        for (int k = 0; k < h.height; ++k) {
            const int parent = idx / FAN_IN, nk = (int)h.lvl[k].size();
            for (int c = parent * FAN_IN; c < std::min(nk, (parent + 1) * FAN_IN); ++c)
                if (c != idx) f(k, c);
            idx = parent;
        }
    }
    // (offset, form) of node (k, idx)'s product, emitting it (and what it needs) on first use; offset −1: empty
    std::pair<int, int> hub_node_slot(int v, int k, int idx, int L0) {
        Hub& h = hubs[hub_of[v]];
        std::pair<int, int>& memo = h.slot[k][idx];
        if (memo.first != -2) return memo;
        if (k == 0) {
            const int e = var_edges[v][idx];
            return memo = (null_[e] ? std::pair<int, int>{-1, 0} : std::pair<int, int>{src_off(e), (int)form[e]});
        }
        std::vector<std::pair<int, int>> ins;
        const int nchild = (int)h.lvl[k - 1].size(), d = P.dim[v];
        for (int c = idx * FAN_IN; c < std::min(nchild, (idx + 1) * FAN_IN); ++c) {
            const auto sl = hub_node_slot(v, k - 1, c, L0);
            if (sl.first >= 0) ins.push_back(sl);
        }
        if (ins.empty()) return memo = {-1, 0};
        if (ins.size() == 1) return memo = ins[0];
        OpRec& r = emit(L0 + hub_node_level(v, k, idx), OP_PRODUCT, d);
        r.w[W_OUT] = (int)P.msg_doubles;
        P.msg_doubles += msz(d);
        r.w[W_LIST] = (int)P.aux.size();
        r.w[W_N] = (int)ins.size();
        for (auto& in : ins) { P.aux.push_back(in.first); P.aux.push_back(in.second); }
        P.bytes_per_sweep += 8ll * msz(d) * (long long)(ins.size() + 1);
        return memo = {r.w[W_OUT], 1};
    }

    void analyse() {
        build_hubs();
        const int M = 2 * E;
        null_.assign(M, 0); needed.assign(M, 0); form.assign(M, 1); done.assign(M, 0);
        alias.assign(M, -1); off.assign(M, -1); level.assign(M, 0);
        std::vector<int> indeg(M);
        std::vector<int> q;
        for (int m = 0; m < M; ++m) { indeg[m] = (int)deps[m].size(); if (!indeg[m]) q.push_back(m); }
        size_t head = 0;
        int processed = 0;
        while (head < q.size()) {
            const int m = q[head++];
            ++processed;
            if (m >= E) {   // variable -> factor
                const int e = m - E;
                std::vector<int> live;
                for (int dpm : deps[m]) if (!null_[dpm]) live.push_back(dpm);
                needed[m] = factor_uses_v2f(edges[e].f);
                if (live.empty()) null_[m] = 1;
                else if (live.size() == 1) { alias[m] = alias[live[0]] >= 0 ? alias[live[0]] : live[0]; form[m] = form[live[0]]; level[m] = level[live[0]]; }
                else if (is_hub(edges[e].v)) {   // the product of the siblings along this edge's path through the variable's tree (above)
                    form[m] = 1;
                    int cnt = 0, lv = 0;
                    hub_siblings(edges[e].v, edge_pos[e], [&](int k, int c) {
                        const int l = hub_node_level(edges[e].v, k, c);
                        if (l >= 0) { ++cnt; lv = std::max(lv, l); }
                    });
                    level[m] = lv + rounds(std::max(cnt, 1));
                } else { form[m] = 1; int lv = 0; for (int x : live) lv = std::max(lv, level[x]); level[m] = lv + rounds((int)live.size()); }
            } else {        // factor -> variable
                const Edge& ed = edges[m];
                needed[m] = 1;
                int lv = -1;
                bool any_null = false;
                for (int dpm : deps[m]) { any_null = any_null || null_[dpm]; lv = std::max(lv, level[dpm]); }
                level[m] = lv + 1;
                if (any_null) null_[m] = 1;
                else if (nclass[ed.f] == NC_NOISE) {
                    if (deps[m].empty()) {   // leaf: the form its single consumer wants, precision form otherwise
                        int wf = 2;
                        const int ov = (int)iface(ed.f, 1 - ed.k);
                        const bool may_miss = g->allow_missing && !mf[ed.f] && (P.vclass[ov] == VC_DATA || P.vclass[ov] == VC_DERIVED);
                        if (var_edges[ed.v].size() == 2 && !may_miss) {   // (a `missing` observation is the zero of the precision form: such leaves stay in it)
                            const int e2 = var_edges[ed.v][0] == m ? var_edges[ed.v][1] : var_edges[ed.v][0];
                            wf = wanted_form(e2);
                        }
                        form[m] = (wf == 0 && wz[ed.f] < 0) ? 0 : 1;   // (a weighted leaf — a mixture component — stays in precision form: weight 0 is the zero message)
                    } else {
                        // a moment-form sum into a variable with three or more edges: every reader is a product or the marginal, each of which would invert
                        // it for itself — the op stores the precision form instead (one inverse where there were two or three)
                        form[m] = form[deps[m][0]];
                        if (!form[m] && var_edges[ed.v].size() >= 3 && !is_hub(ed.v)) form[m] = 1;
                    }
                } else if (nclass[ed.f] == NC_GCVZ) form[m] = 0;   // (the ELQ message's Gaussian moments)
                else if (nclass[ed.f] == NC_MUL) form[m] = ed.k == 0 ? 0 : 1;
                else if (deps[m].size() == 2) form[m] = ed.k == 0 ? 0 : form[deps[m][0]];   // `+`: (:out) adds moments; (:in) keeps the form of the message from `out` (deps[m][0])
                else form[m] = form[deps[m][0]];
            }
            done[m] = 1;
            for (int u : users[m]) if (--indeg[u] == 0) q.push_back(u);
        }
        if (processed != M) fail(RXHIP_ERR_UNSUPPORTED, "internal: the message dependencies of this graph are not acyclic");
        for (int64_t v = 0; v < nv; ++v)
            if (P.vclass[v] == VC_GAUSS) {
                bool any = false;
                for (int e : var_edges[v]) any = any || !null_[e];
                if (!any) fail(RXHIP_ERR_BADARG, "random variable %lld receives no message (no prior and no data reach it)", (long long)v);
            }
        for (const Gcv& gc : gcvs) {
            if (null_[fac_edges[gc.fz][0]]) fail(RXHIP_ERR_UNSUPPORTED, "factor %d (GCV): y, x and z each need a proper message from the rest of the graph", gc.node);
            for (int e2 : var_edges[gc.z]) {   // z's other neighbours see the ELQ message through its Gaussian moments: structured Gaussian nodes only
                const int f2 = edges[e2].f;
                if (f2 != gc.fz && !(nclass[f2] == NC_NOISE && !mf[f2]))
                    fail(RXHIP_ERR_UNSUPPORTED, "factor %d (GCV): its volatility input (variable %d) may only touch structured Gaussian nodes besides", gc.node, gc.z);
            }
        }
    }

    // ---- emission ----
    int value_source(int v, int& flag_bit) {   // offset + whether it is a per-replica slot
        if (P.vclass[v] == VC_CONST) { flag_bit = 0; return const_value(v); }
        flag_bit = 1;
        return P.val_off[v];
    }
    OpRec& emit(int lvl, int op, int d0) {
        recs.push_back({});
        OpRec& r = recs.back();
        r.level = lvl;
        std::memset(r.w, 0, sizeof r.w);
        r.w[W_OP] = op; r.w[W_D0] = d0;
        r.w[W_IN0] = r.w[W_IN1] = r.w[W_IN2] = r.w[W_OUT] = r.w[W_PREC] = r.w[W_TERM] = -1;
        return r;
    }
    int src_off(int m) const { return off[alias[m] >= 0 ? alias[m] : m]; }
    void noise_params(OpRec& r, int f, int d) {
        const int c = (int)iface(f, 2), t = ftype(f);
        if (gcv_of[f] >= 0) r.w[W_PREC] = gcvs[gcv_of[f]].state;
        else if (P.vclass[c] == VC_PREC) r.w[W_PREC] = P.prec_off[c];
        else if (P.vclass[c] == VC_DATA) {
            r.w[W_C0] = P.val_off[c];
            r.w[W_FLAGS] |= F_NOISE_VAL | ((t == RXHIP_NODE_MVNORMAL_MEAN_PRECISION || t == RXHIP_NODE_NORMAL_MEAN_PRECISION) ? F_NOISE_VAL_PREC : 0);
        } else r.w[W_C0] = noise_block(c, d, t == RXHIP_NODE_MVNORMAL_MEAN_PRECISION || t == RXHIP_NODE_NORMAL_MEAN_PRECISION);
    }

    // A product or marginal over MANY inbound messages (a hub variable: the mean of 10^5 iid observations) as a tree of partial products, eight inputs
    // per op, one level per round — O(n) work in O(log n) levels instead of one lane summing n messages.  Returns the reduced list and the level the
    // final op may run at.
    static constexpr int FAN_IN = 8;
    static int rounds(int n) { int r = 1; while (n > FAN_IN) { n = (n + FAN_IN - 1) / FAN_IN; ++r; } return r; }   // ops in sequence for n inputs
    std::vector<std::pair<int, int>> reduce_inputs(std::vector<std::pair<int, int>> ins, int d, int& lvl) {
        while ((int)ins.size() > FAN_IN) {
            std::vector<std::pair<int, int>> nxt;
            for (size_t i = 0; i < ins.size(); i += FAN_IN) {
                const size_t n = std::min<size_t>(FAN_IN, ins.size() - i);
                if (n == 1) { nxt.push_back(ins[i]); continue; }
                OpRec& r = emit(lvl, OP_PRODUCT, d);
                r.w[W_OUT] = (int)P.msg_doubles;
                P.msg_doubles += msz(d);
                r.w[W_LIST] = (int)P.aux.size();
                r.w[W_N] = (int)n;
                for (size_t q = 0; q < n; ++q) { P.aux.push_back(ins[i + q].first); P.aux.push_back(ins[i + q].second); }
                P.bytes_per_sweep += 8ll * msz(d) * (long long)(n + 1);
                nxt.push_back({r.w[W_OUT], 1});
            }
            ins.swap(nxt);
            ++lvl;
        }
        return ins;
    }

    void allocate() {
        P.marg_off.assign(nv, -1); P.val_off.assign(nv, -1); P.prec_off.assign(nv, -1);
        cval_off.assign(nv, -1); cmat_off.assign(nv, -1); noise_off_cov.assign(nv, -1); noise_off_prec.assign(nv, -1); prior_off.assign(nv, -1);
        long long vo = 0;
        for (int64_t v = 0; v < nv; ++v)
            if (P.vclass[v] == VC_DATA) { P.val_off[v] = (int)vo; vo += P.dim[v]; P.data_vars.push_back(v); }
        P.data_doubles = vo;
        for (int64_t v = 0; v < nv; ++v)
            if (P.vclass[v] == VC_DERIVED) { P.val_off[v] = (int)vo; vo += P.dim[v]; }
        P.val_doubles = vo;
        long long mo = 0, po = 0;
        for (int64_t v = 0; v < nv; ++v) {
            if (P.vclass[v] == VC_GAUSS) { P.marg_off[v] = (int)mo; mo += msz(P.dim[v]) + 1; }
            if (P.vclass[v] == VC_PREC) { const int d = P.dim[v]; P.prec_off[v] = (int)po; po += 2 + d * (d + 1) / 2 + 2 * d * d; }
            if (P.vclass[v] == VC_CAT) { P.prec_off[v] = (int)po; po += catK[v]; }          // q(z): π[K]
            if (P.vclass[v] == VC_DIR) { P.prec_off[v] = (int)po; po += 2 * catK[v]; }      // q(s): α[K] | E log s[K]
        }
        for (size_t i = 0; i < gcvs.size(); ++i) { gcvs[i].state = (int)po; po += 5; gcvs[i].stat = (int)i; }   // (ψ: the first residual-moment slots)   // γ(z): [· | · | E γ | 1 / E γ | E log γ] — the layout of a scalar precision variable's state
        P.marg_doubles = mo; P.prec_doubles = po;
        long long so = 0;
        for (int m = 0; m < 2 * E; ++m)
            if (!null_[m] && needed[m] && alias[m] < 0) { off[m] = (int)so; so += msz(P.dim[edges[m % E].v]); ++P.n_messages; }
        for (int m = 0; m < 2 * E; ++m)   // an alias of a message nobody else needed (cannot happen: every f2v is needed)
            if (alias[m] >= 0 && off[alias[m]] < 0) fail(RXHIP_ERR_BADARG, "internal: alias of an unallocated message");
        if (so > (1ll << 30) || mo > (1ll << 30) || vo > (1ll << 30)) fail(RXHIP_ERR_UNSUPPORTED, "graph too large for the executor's 32-bit slot offsets");
        P.msg_doubles = so;
    }

    void init_precision() {
        P.prec_init.assign((size_t)P.prec_doubles, 0.0);
        for (const Gcv& gc : gcvs) {   // γ(z) under the `@initialization` marginal of z (hgf_tests.jl:47-50)
            if (!(g->var_init_family && g->var_init && g->var_init_family[gc.z] == RXHIP_INIT_NORMAL && g->var_init[gc.z] >= 0))
                fail(RXHIP_ERR_BADARG, "factor %d (GCV): its volatility input (variable %d) needs a Normal @initialization marginal", gc.node, gc.z);
            const double* q = g->const_pool + g->var_init[gc.z];
            const double kappa = cptr(gc.kv)[0], omega = cptr(gc.ov)[0];
            if (!(q[1] > 0.0)) fail(RXHIP_ERR_BADARG, "factor %d (GCV): initial marginal of variable %d is not a proper Gaussian", gc.node, gc.z);
            double* st = P.prec_init.data() + gc.state;
            st[2] = std::exp(-omega - kappa * q[0] + 0.5 * kappa * kappa * q[1]);
            st[3] = 1.0 / st[2];
            st[4] = -(kappa * q[0] + omega);
        }
        for (int64_t v = 0; v < nv; ++v) {
            double* st = P.prec_init.data() + (P.prec_off[v] >= 0 ? P.prec_off[v] : 0);
            const int K = catK[v];
            if (P.vclass[v] == VC_CAT)
                for (int k = 0; k < K; ++k) st[k] = 1.0 / K;
            if (P.vclass[v] == VC_DIR) {   // `@initialization` q(s), default the prior
                const double* a = alpha0((int)v);
                if (g->var_init_family && g->var_init && g->var_init_family[v] == RXHIP_INIT_DIRICHLET && g->var_init[v] >= 0) a = g->const_pool + g->var_init[v];
                double sum = 0.0;
                for (int k = 0; k < K; ++k) { if (!(a[k] > 0.0)) fail(RXHIP_ERR_BADARG, "initial marginal of variable %lld: positive concentrations expected", (long long)v); sum += a[k]; }
                for (int k = 0; k < K; ++k) { st[k] = a[k]; st[K + k] = host_digamma(a[k]) - host_digamma(sum); }
            }
        }
        for (int64_t f = 0; f < nf; ++f) {
            if (nclass[f] != NC_PRIOR) continue;
            const int v = (int)iface((int)f, 0), d = P.dim[v], t = ftype((int)f);
            const double* a = cptr((int)iface((int)f, 1));
            const double* b = cptr((int)iface((int)f, 2));
            double nu0;
            std::vector<double> S0((size_t)d * d), S0i((size_t)d * d);
            if (t == RXHIP_NODE_WISHART) {
                if ((int64_t)g->var_rows[iface((int)f, 2)] * g->var_cols[iface((int)f, 2)] != (int64_t)d * d) fail(RXHIP_ERR_BADARG, "Wishart scale of variable %d is not %d x %d", v, d, d);
                nu0 = a[0];
                std::copy(b, b + (size_t)d * d, S0.begin());
            } else {
                if (d != 1) fail(RXHIP_ERR_BADARG, "a Gamma prior on a vector variable");
                const double rate = t == RXHIP_NODE_GAMMA_SHAPE_RATE ? b[0] : 1.0 / b[0];
                nu0 = 2.0 * a[0];
                S0[0] = 1.0 / (2.0 * rate);
            }
            double ldS0 = 0.0;
            if (!(nu0 > d - 1.0) || !host_chol_inv(d, S0.data(), S0i.data(), &ldS0)) fail(RXHIP_ERR_NOT_POSDEF, "prior of precision variable %d is not a proper Wishart / Gamma", v);
            prior_off[v] = (int)P.cpool.size();
            P.cpool.push_back(nu0);
            P.cpool.insert(P.cpool.end(), S0i.begin(), S0i.end());
            P.cpool.push_back(ldS0);
            // initial marginal: `@initialization`, default the prior
            double nu = nu0;
            std::vector<double> V = S0;
            if (g->var_init_family && g->var_init && g->var_init_family[v] != RXHIP_INIT_NONE && g->var_init[v] >= 0) {
                const double* q = g->const_pool + g->var_init[v];
                if (g->var_init_family[v] == RXHIP_INIT_WISHART) { nu = q[0]; std::copy(q + 1, q + 1 + (size_t)d * d, V.begin()); }
                else if (g->var_init_family[v] == RXHIP_INIT_GAMMA && d == 1) { nu = 2.0 * q[0]; V[0] = 1.0 / (2.0 * q[1]); }
                else fail(RXHIP_ERR_BADARG, "initial marginal of precision variable %d: Wishart (or Gamma for scalars) expected", v);
            }
            std::vector<double> What((size_t)d * d), Whi((size_t)d * d);
            for (size_t i = 0; i < What.size(); ++i) What[i] = nu * V[i];
            double ldW = 0.0;
            if (!(nu > d - 1.0) || !host_chol_inv(d, What.data(), Whi.data(), &ldW)) fail(RXHIP_ERR_NOT_POSDEF, "initial marginal of precision variable %d is not proper", v);
            double* st = P.prec_init.data() + P.prec_off[v];
            const int tri = d * (d + 1) / 2;
            st[0] = nu;
            for (int i = 0, k = 0; i < d; ++i)
                for (int j = 0; j <= i; ++j) st[1 + k++] = V[i * d + j];
            std::copy(What.begin(), What.end(), st + 1 + tri);
            std::copy(Whi.begin(), Whi.end(), st + 1 + tri + d * d);
            double el = d * std::log(2.0) + (ldW - d * std::log(nu));
            for (int i = 0; i < d; ++i) el += host_digamma(0.5 * nu - 0.5 * i);
            st[1 + tri + 2 * d * d] = el;
        }
    }

    // the @initialization marginals of the Gaussian variables a mean-field node reads (InitMarExtraKey, src/model/plugins/initialization_plugin.jl:201-202): the
    // reference refuses to run such a model without them too
    void init_marginals() {
        if (!P.has_mf) return;
        P.marg_init.assign((size_t)P.marg_doubles, 0.0);
        std::vector<char> need(nv, 0);
        for (int64_t f = 0; f < nf; ++f)
            if (mf[f]) need[iface((int)f, 0)] = need[iface((int)f, 1)] = 1;
        for (const Mix& mx : mixes) {
            for (int mk : mx.m) need[mk] = need[mk] || P.vclass[mk] == VC_GAUSS;
            need[mx.out] = need[mx.out] || P.vclass[mx.out] == VC_GAUSS;
        }
        for (int64_t v = 0; v < nv; ++v) {
            if (!need[v]) continue;
            const int d = P.dim[v];
            auto family = [&](int64_t x) { return (g->var_init_family && g->var_init && g->var_init[x] >= 0) ? g->var_init_family[x] : (int)RXHIP_INIT_NONE; };
            auto has_init = [&](int64_t x) { return family(x) == RXHIP_INIT_MVNORMAL || (family(x) == RXHIP_INIT_NORMAL && P.dim[x] == 1); };
            std::vector<double> m((size_t)d), V((size_t)d * d), Vi((size_t)d * d);
            if (has_init(v)) {
                const double* q = g->const_pool + g->var_init[v];
                std::copy(q, q + d, m.begin());
                std::copy(q + d, q + d + (size_t)d * d, V.begin());
            } else {
                // the anonymous output of `A * x` cannot be named in an @initialization block: it starts as the image (A m, A V Aᵀ) of its input's initial marginal
                int64_t fm = -1;
                for (int64_t f = 0; f < nf && fm < 0; ++f)
                    if (nclass[f] == NC_MUL && iface((int)f, 0) == v && has_init(iface((int)f, 2))) fm = f;
                if (fm < 0)
                    fail(RXHIP_ERR_BADARG, "variable %lld sits on a Gaussian node under q(out) q(μ) and has no @initialization marginal (Normal / MvNormal): the first iteration has nothing to read",
                         (long long)v);
                const int64_t u = iface((int)fm, 2);
                const int du = P.dim[u];
                const double *A = cptr((int)iface((int)fm, 1)), *q = g->const_pool + g->var_init[u];
                for (int i = 0; i < d; ++i) {
                    double sm = 0.0;
                    for (int k = 0; k < du; ++k) sm += A[i * du + k] * q[k];
                    m[i] = sm;
                    for (int j = 0; j < d; ++j) {
                        double sv = 0.0;
                        for (int k = 0; k < du; ++k)
                            for (int l = 0; l < du; ++l) sv += A[i * du + k] * q[du + k * du + l] * A[j * du + l];
                        V[(size_t)i * d + j] = sv;
                    }
                }
            }
            double ld = 0.0;
            if (!host_chol_inv(d, V.data(), Vi.data(), &ld)) fail(RXHIP_ERR_NOT_POSDEF, "initial marginal of variable %lld is not a proper Gaussian", (long long)v);
            double* st = P.marg_init.data() + P.marg_off[v];
            std::copy(m.begin(), m.end(), st);
            for (int i = 0, k = 0; i < d; ++i)
                for (int j = 0; j <= i; ++j) st[d + k++] = 0.5 * (V[i * d + j] + V[j * d + i]);
            st[d + d * (d + 1) / 2] = ld;
        }
    }

    void emit_all() {
        // derived values first (their own dependency order)
        {
            std::vector<int> lv(nv, 0);
            bool changed = true;
            std::vector<char> emitted(nv, 0);
            while (changed) {
                changed = false;
                for (int64_t f = 0; f < nf; ++f) {
                    if (nclass[f] != NC_MUL && nclass[f] != NC_ADD) continue;
                    const int o = (int)iface((int)f, 0);
                    if (P.vclass[o] != VC_DERIVED || emitted[o]) continue;
                    const int a = (int)iface((int)f, 1), b = (int)iface((int)f, 2);
                    auto ready = [&](int v) { return P.vclass[v] != VC_DERIVED || emitted[v]; };
                    if (nclass[f] == NC_MUL) {
                        if (!ready(b)) continue;
                        OpRec& r = emit(lv[b] , OP_DERIVE_MUL, P.dim[o]);
                        r.w[W_D1] = P.dim[b];
                        r.w[W_C0] = const_matrix(a, P.dim[o], P.dim[b]);
                        int bit; r.w[W_VAL] = value_source(b, bit); if (bit) r.w[W_FLAGS] |= F_VAL_SLOT;
                        r.w[W_OUT] = P.val_off[o];
                        lv[o] = lv[b] + 1;
                    } else {
                        if (!ready(a) || !ready(b)) continue;
                        const int l = std::max(lv[a], lv[b]);
                        OpRec& r = emit(l, OP_DERIVE_ADD, P.dim[o]);
                        int bit; r.w[W_VAL] = value_source(a, bit); if (bit) r.w[W_FLAGS] |= F_VAL_SLOT;
                        r.w[W_VAL2] = value_source(b, bit); if (bit) r.w[W_FLAGS] |= F_VAL2_SLOT;
                        r.w[W_OUT] = P.val_off[o];
                        lv[o] = l + 1;
                    }
                    emitted[o] = 1;
                    changed = true;
                }
            }
            derived_levels = 0;
            for (int64_t v = 0; v < nv; ++v) derived_levels = std::max(derived_levels, lv[v]);
        }
        // q(z) of every mixture node, from the marginals q(m), q(p), q(s) (and q(out)) of the PREVIOUS iteration: a level of its own in front of the messages
        for (const Mix& mx : mixes) {
            const int d = P.dim[mx.out];
            OpRec& r = emit(derived_levels, OP_CAT_UPDATE, d);
            r.w[W_N] = mx.K;
            r.w[W_OUT] = P.prec_off[mx.z];
            if (P.vclass[mx.out] == VC_GAUSS) { r.w[W_VAL] = P.marg_off[mx.out]; r.w[W_FLAGS] |= F_VAL_MARG; }
            else { int bit; r.w[W_VAL] = value_source(mx.out, bit); if (bit) r.w[W_FLAGS] |= F_VAL_SLOT; }
            const int sv = cat_s[mx.z];
            if (P.vclass[sv] == VC_DIR) r.w[W_IN0] = P.prec_off[sv];
            else r.w[W_C0] = log_probabilities(sv, mx.K);
            r.w[W_LIST] = (int)P.aux.size();
            for (int k = 0; k < mx.K; ++k) {   // per component: where the mean's marginal (or constant value) and the precision's state (or constant block) are
                const int mk = mx.m[k], pk = mx.p[k];
                P.aux.push_back(P.vclass[mk] == VC_GAUSS ? P.marg_off[mk] : -1 - const_value(mk));
                P.aux.push_back(P.vclass[pk] == VC_PREC ? P.prec_off[pk] : -1 - noise_block(pk, d, true));
            }
        }
        if (!mixes.empty()) ++derived_levels;
        const int L0 = derived_levels;   // messages start behind the derived values
        int maxl = 0;
        for (int m = 0; m < 2 * E; ++m) {
            if (null_[m] || !needed[m] || alias[m] >= 0) continue;
            const Edge& ed = edges[m % E];
            const int d = P.dim[ed.v], lv = L0 + level[m];
            maxl = std::max(maxl, lv);
            if (m >= E) {   // product
                std::vector<std::pair<int, int>> ins;
                if (is_hub(ed.v)) {
                    hub_siblings(ed.v, edge_pos[m - E], [&](int k, int c) {
                        const auto sl = hub_node_slot(ed.v, k, c, L0);
                        if (sl.first >= 0) ins.push_back(sl);
                    });
                } else
                    for (int dpm : deps[m])
                        if (!null_[dpm]) ins.push_back({src_off(dpm), (int)form[dpm]});
                int lvp = lv - rounds((int)ins.size()) + 1;   // a hub: partial products first; `lv` (analyse) is the level of the final op
                ins = reduce_inputs(ins, d, lvp);
                OpRec& r = emit(lvp, OP_PRODUCT, d);
                r.w[W_OUT] = off[m];
                r.w[W_LIST] = (int)P.aux.size();
                r.w[W_N] = (int)ins.size();
                for (auto& in : ins) { P.aux.push_back(in.first); P.aux.push_back(in.second); P.bytes_per_sweep += 8ll * msz(d); }
                P.bytes_per_sweep += 8ll * msz(d);
                continue;
            }
            const int f = ed.f;
            if (nclass[f] == NC_GCVZ) {
                const Gcv& gc = gcvs[gcv_of[f]];
                OpRec& r = emit(lv, OP_GCV_Z, 1);
                const int my = E + fac_edges[gc.fa][0], mx = E + fac_edges[gc.fa][1];
                r.w[W_IN0] = src_off(my); if (form[my]) r.w[W_FLAGS] |= F_IN0_WP;
                r.w[W_IN1] = src_off(mx); if (form[mx]) r.w[W_FLAGS] |= F_IN1_WP;
                r.w[W_PREC] = gc.state;
                r.w[W_C0] = gcv_constants(gc);
                r.w[W_C1] = gc.stat;
                r.w[W_OUT] = off[m];
                P.bytes_per_sweep += 8ll * 4 * msz(1);
                continue;
            }
            if (nclass[f] == NC_NOISE) {
                const int oe = fac_edges[f][1 - ed.k];
                if (oe < 0 || mf[f]) {
                    OpRec& r = emit(lv, OP_LEAF, d);
                    if (mf[f]) { r.w[W_VAL] = P.marg_off[iface(f, 1 - ed.k)]; r.w[W_FLAGS] |= F_VAL_MARG; }   // N(E[other interface], Σ): the mean of last iteration's marginal
                    else {
                        int bit; r.w[W_VAL] = value_source((int)iface(f, 1 - ed.k), bit); if (bit) r.w[W_FLAGS] |= F_VAL_SLOT;
                        if (g->allow_missing && P.vclass[iface(f, 1 - ed.k)] != VC_CONST) {
                            if (P.vclass[iface(f, 2)] == VC_PREC)
                                fail(RXHIP_ERR_UNSUPPORTED, "factor %d: `missing` observations under a random precision (the count of the q(W) update would depend on the data)", f);
                            r.w[W_FLAGS] |= F_MAY_MISS;
                        }
                    }
                    if (form[m]) r.w[W_FLAGS] |= F_OUT_WP;
                    noise_params(r, f, d);
                    weigh(r, f);
                    r.w[W_OUT] = off[m];
                } else {
                    OpRec& r = emit(lv, OP_NOISE, d);
                    r.w[W_IN0] = src_off(E + oe);
                    if (form[E + oe]) r.w[W_FLAGS] |= F_IN0_WP;
                    if (form[m]) r.w[W_FLAGS] |= F_OUT_WP;
                    noise_params(r, f, d);
                    r.w[W_OUT] = off[m];
                    P.bytes_per_sweep += 8ll * msz(d);
                }
            } else if (nclass[f] == NC_MUL) {
                const int dout = P.dim[iface(f, 0)], din = P.dim[iface(f, 2)];
                OpRec& r = emit(lv, ed.k == 0 ? OP_MUL_OUT : OP_MUL_IN, dout);
                r.w[W_D1] = din;
                r.w[W_C0] = const_matrix((int)iface(f, 1), dout, din);
                const int se = fac_edges[f][ed.k == 0 ? 2 : 0];
                r.w[W_IN0] = src_off(E + se);
                if (form[E + se]) r.w[W_FLAGS] |= F_IN0_WP;
                r.w[W_OUT] = off[m];
                P.bytes_per_sweep += 8ll * msz(ed.k == 0 ? din : dout);
            } else {   // `+`
                std::vector<int> oth;
                for (int k = 0; k < 3; ++k) if (k != ed.k) oth.push_back(k);
                const int e0 = fac_edges[f][oth[0]], e1 = fac_edges[f][oth[1]];
                if (e0 >= 0 && e1 >= 0) {
                    // out = in1 + in2: (:out) adds the inputs; (:in_k) subtracts the other input from the message toward out
                    const int first = ed.k == 0 ? e0 : fac_edges[f][0], second = ed.k == 0 ? e1 : (oth[0] == 0 ? e1 : e0);
                    OpRec& r = emit(lv, ed.k == 0 ? OP_ADD_OUT : OP_ADD_IN, d);
                    r.w[W_IN0] = src_off(E + first); if (form[E + first]) r.w[W_FLAGS] |= F_IN0_WP;
                    r.w[W_IN1] = src_off(E + second); if (form[E + second]) r.w[W_FLAGS] |= F_IN1_WP;
                    r.w[W_OUT] = off[m];
                    P.bytes_per_sweep += 16ll * msz(d);
                } else {
                    const int ge = e0 >= 0 ? e0 : e1, ck = e0 >= 0 ? oth[1] : oth[0];
                    OpRec& r = emit(lv, OP_SHIFT, d);
                    r.w[W_IN0] = src_off(E + ge);
                    if (form[E + ge]) r.w[W_FLAGS] |= F_IN0_WP | F_OUT_WP;
                    int bit; r.w[W_VAL] = value_source((int)iface(f, ck), bit); if (bit) r.w[W_FLAGS] |= F_VAL_SLOT;
                    if (ed.k != 0) r.w[W_FLAGS] |= F_NEG;   // toward an input: out − constant
                    r.w[W_OUT] = off[m];
                    P.bytes_per_sweep += 8ll * msz(d);
                }
            }
            P.bytes_per_sweep += 8ll * msz(d);
        }
        // marginals: one level behind the last message
        const int LM = maxl + 1;
        int lm_last = LM;
        // the output of `A * x` with x random: its marginal is the image of x's (OP_MARG_PUSH, on demand) — no product of the
        // messages on its two edges in the sweep; one level only (the input's own marginal comes from messages)
        push_from.assign(nv, -1);
        push_fac.assign(nv, -1);
        for (int64_t f = 0; f < nf; ++f) {
                if (nclass[f] != NC_MUL) continue;
                const int o = (int)iface((int)f, 0), in = (int)iface((int)f, 2);
                if (P.vclass[o] == VC_GAUSS && P.vclass[in] == VC_GAUSS) { push_from[o] = in; push_fac[o] = (int)f; }
            }
        for (int64_t f = 0; f < nf; ++f)   // (a mean-field rule reads the STORED marginal of its other interface every iteration: such a variable keeps the message route)
            if (mf[f]) push_from[iface((int)f, 0)] = push_from[iface((int)f, 1)] = -1;
        for (const Mix& mx : mixes) {   // (so does the switch's rule)
            for (int mk : mx.m) push_from[mk] = -1;
            push_from[mx.out] = -1;
        }
        for (int64_t v = 0; v < nv; ++v)
            if (push_from[v] >= 0 && push_from[push_from[v]] >= 0) push_from[v] = -2;   // the input is an image itself: this one takes the message route
        for (int64_t v = 0; v < nv; ++v)
            if (push_from[v] == -2) push_from[v] = -1;
        for (int64_t v = 0; v < nv; ++v) {
            if (P.vclass[v] != VC_GAUSS || push_from[v] >= 0) continue;
            if (gcv_z[v]) {   // the volatility input of a GCV node: the ELQ message times the product of all other messages, by cubature
                const Gcv* gc = nullptr;
                for (const Gcv& c2 : gcvs) if (c2.z == (int)v) gc = &c2;
                const int mz = E + fac_edges[gc->fz][0];
                if (null_[mz]) fail(RXHIP_ERR_UNSUPPORTED, "factor %d (GCV): its volatility input receives no Gaussian message to match the product against", gc->node);
                OpRec& r = emit(LM, OP_GCV_ZMARG, 1);
                r.w[W_IN0] = src_off(mz); if (form[mz]) r.w[W_FLAGS] |= F_IN0_WP;
                r.w[W_IN1] = off[fac_edges[gc->fz][0]];   // (not loaded: names the op whose ψ this one reads, so that every schedule orders the two)
                r.w[W_C0] = gcv_constants(*gc);
                r.w[W_C1] = gc->stat;
                r.w[W_OUT] = P.marg_off[v];
                lm_last = std::max(lm_last, LM);
                continue;
            }
            const int d = P.dim[v];
            std::vector<std::pair<int, int>> ins;
            int hub_lv = 0;
            if (is_hub((int)v)) {   // the top nodes of the variable's tree (their products exist already wherever a neighbour needed them)
                const Hub& h = hubs[hub_of[v]];
                const int top = h.height - 1;
                for (int c = 0; c < (int)h.lvl[top].size(); ++c) {
                    const auto sl = hub_node_slot((int)v, top, c, L0);
                    if (sl.first >= 0) { ins.push_back(sl); hub_lv = std::max(hub_lv, L0 + hub_node_level((int)v, top, c) + 1); }
                }
            } else
                for (int e : var_edges[v])
                    if (!null_[e]) ins.push_back({src_off(e), (int)form[e]});
            int lvm = std::max(LM, hub_lv);   // (a hub nobody takes a product from builds its tree for the marginal alone: behind the last node)
            ins = reduce_inputs(ins, d, lvm);
            lm_last = std::max(lm_last, lvm);
            OpRec& r = emit(lvm, OP_MARGINAL, d);
            r.w[W_OUT] = P.marg_off[v];
            if (g->allow_missing) r.w[W_FLAGS] |= F_MAY_MISS;   // (the op looks for the case "one moment-form message, the others the zero of the precision form")
            r.w[W_LIST] = (int)P.aux.size();
            r.w[W_N] = (int)ins.size();
            for (auto& in : ins) { P.aux.push_back(in.first); P.aux.push_back(in.second); P.bytes_per_sweep += 8ll * msz(d); }
            P.bytes_per_sweep += 8ll * (msz(d) + 1);
        }
        // second phase: the Bethe terms and residual moments.  A term that needs the marginal of an image variable (push_from) reads the marginal it is the image
        // of and forms (A m, A V Aᵀ, log|A V Aᵀ|) on the fly (F_PUSH_*), and books that variable's own entropy term while it has the log-determinant (F_FOLD_ENT)
        const int LF = lm_last + 1;
        std::vector<int> terms;
        std::vector<std::pair<size_t, int>> fold;   // (op record, variable): ops that have log|V| of an image variable at hand
        // word_ld (−1: none): where the op finds 2·log|det A| of a SQUARE map — log|A V Aᵀ| = log|V| + 2 log|det A|, so the image's log-determinant costs an
        // addition instead of a Cholesky sweep (at d = 64: an inverse's worth of work per Bethe term)
        auto marg_of = [&](OpRec& r, int v, int word, int bit, int word_a, int word_du, int word_ld) {
            if (push_from[v] >= 0) {
                const int u = push_from[v];
                r.w[word] = P.marg_off[u];
                r.w[W_FLAGS] |= bit;
                r.w[word_a] = const_matrix((int)iface(push_fac[v], 1), P.dim[v], P.dim[u]);
                r.w[word_du] = P.dim[u];
                if (word_ld >= 0) r.w[word_ld] = push_logdet(v);
            } else
                r.w[word] = P.marg_off[v];
        };
        std::vector<int> ent_coef(nv, 0);
        std::vector<std::vector<int>> prec_stats(nv), prec_weight(nv);
        std::vector<int> prec_nodes(nv, 0);
        long long stat_o = (long long)gcvs.size();   // (ψ of every GCV node first: allocate())
        auto new_term = [&]() { terms.push_back((int)P.term_slots); return (int)P.term_slots++; };
        auto msg_in = [&](OpRec& r, int word, int bit, int m) {
            if (null_[m]) { r.w[word] = -1; return; }
            r.w[word] = src_off(m);
            if (form[m]) r.w[W_FLAGS] |= bit;
        };
        for (int64_t v = 0; v < nv; ++v)
            if (P.vclass[v] == VC_GAUSS) ent_coef[v] = (int)var_edges[v].size() - 1;
        for (int64_t f = 0; f < nf; ++f) {
            const int a = (int)iface((int)f, 0), b = (int)iface((int)f, 1), c = (int)iface((int)f, 2);
            if (nclass[f] == NC_NOISE) {
                const int d = P.dim[a];
                const bool ga = P.vclass[a] == VC_GAUSS, gb = P.vclass[b] == VC_GAUSS, rw = P.vclass[c] == VC_PREC || gcv_of[f] >= 0;
                OpRec& r = emit(LF, ga && gb ? OP_FE_NOISE2 : (ga || gb) ? OP_FE_NOISE1 : OP_FE_NOISE0, d);
                noise_params(r, (int)f, d);
                if (mf[f]) {   // q(out) q(μ): the average energy from the two marginals; the clusters' entropies go with the variables' terms
                    r.w[W_OP] = OP_FE_NOISE_MF;
                    r.w[W_VAL] = P.marg_off[a];
                    r.w[W_VAL2] = P.marg_off[b];
                    ent_coef[a] -= 1;
                    ent_coef[b] -= 1;
                } else if (ga && gb) {
                    // the joint from ONE inbound message and the two marginals (tree_kernels.hpp / tree_wave_kernels.hpp OP_FE_NOISE2M); side a = the interface
                    // whose message to the node is stored in precision form (no conversion), the out side when both are
                    const int m0 = E + fac_edges[f][0], m1 = E + fac_edges[f][1];
                    bool use1 = !null_[m1] && form[m1] && (null_[m0] || !form[m0]);
                    if (gcv_z[a] || gcv_z[b]) {   // the marginal of a GCV node's volatility input is not the product of its Gaussian messages: it takes side a, mean from the joint
                        use1 = gcv_z[b] && !gcv_z[a];
                        if (null_[use1 ? m1 : m0]) fail(RXHIP_ERR_UNSUPPORTED, "factor %lld: no message from the volatility input", (long long)f);
                        r.w[W_FLAGS] |= F_JOINT_MEAN;
                        if (gcv_z[a] && gcv_z[b]) {   // a transition between two volatility states: side b from its message as well
                            if (null_[m1]) fail(RXHIP_ERR_UNSUPPORTED, "factor %lld: no message from the volatility input", (long long)f);
                            r.w[W_FLAGS] |= F_JOINT_B;
                        }
                    }
                    r.w[W_OP] = OP_FE_NOISE2M;
                    msg_in(r, W_IN0, F_IN0_WP, use1 ? m1 : m0);
                    marg_of(r, use1 ? b : a, W_VAL, F_PUSH_A, W_IN1, W_LIST, -1);
                    marg_of(r, use1 ? a : b, W_VAL2, F_PUSH_B, W_IN2, W_N, W_D1);
                    if (r.w[W_FLAGS] & F_JOINT_B) msg_in(r, W_IN1, F_IN1_WP, m1);   // (side a is interface 0 here; neither side is an image)
                    r.w[W_OUT] = 0;
                    if (push_from[use1 ? a : b] >= 0) fold.push_back({recs.size() - 1, use1 ? a : b});
                } else if (ga || gb) {
                    if (g->allow_missing && P.vclass[ga ? b : a] != VC_CONST) r.w[W_FLAGS] |= F_MAY_MISS;
                    marg_of(r, ga ? a : b, W_IN0, F_PUSH_A, W_IN1, W_D1, W_IN2);
                    r.w[W_OUT] = 0;
                    if (push_from[ga ? a : b] >= 0) fold.push_back({recs.size() - 1, ga ? a : b});
                    int bit; r.w[W_VAL] = value_source(ga ? b : a, bit); if (bit) r.w[W_FLAGS] |= F_VAL_SLOT;
                } else {
                    if (g->allow_missing && (P.vclass[a] != VC_CONST || P.vclass[b] != VC_CONST)) r.w[W_FLAGS] |= F_MAY_MISS;
                    int bit; r.w[W_VAL] = value_source(a, bit); if (bit) r.w[W_FLAGS] |= F_VAL_SLOT;
                    r.w[W_VAL2] = value_source(b, bit); if (bit) r.w[W_FLAGS] |= F_VAL2_SLOT;
                }
                r.w[W_TERM] = new_term();
                weigh(r, (int)f);
                if (rw) {
                    r.w[W_FLAGS] |= F_STAT;
                    r.w[W_C1] = (int)stat_o;
                    if (gcv_of[f] >= 0) { r.w[W_C1] = gcvs[gcv_of[f]].stat; stat_o -= d * d; }   // (its own slot, shared with the sweep's message op)
                    else {
                        prec_stats[c].push_back((int)stat_o);
                        prec_weight[c].push_back(wz[f] >= 0 ? P.prec_off[wz[f]] + wk[f] : -1);
                    }
                    stat_o += d * d;
                }
            } else if (nclass[f] == NC_MUL) {
                if (P.vclass[c] == VC_GAUSS && P.vclass[a] == VC_GAUSS) ent_coef[c] -= 1;   // −H[q(in)]
            } else if (nclass[f] == NC_ADD && P.vclass[a] == VC_GAUSS) {
                const bool g1 = P.vclass[b] == VC_GAUSS, g2 = P.vclass[c] == VC_GAUSS;
                if (g1 && g2) {
                    OpRec& r = emit(LF, OP_FE_ADD2, P.dim[a]);
                    msg_in(r, W_IN0, F_IN0_WP, E + fac_edges[f][1]);
                    msg_in(r, W_IN1, F_IN1_WP, E + fac_edges[f][2]);
                    msg_in(r, W_IN2, F_IN2_WP, E + fac_edges[f][0]);
                    r.w[W_TERM] = new_term();
                } else ent_coef[g1 ? b : c] -= 1;
            }
        }
        for (const Gcv& gc : gcvs) ent_coef[gc.z] -= 1;   // the node's second cluster: −H[q(z)]
        for (auto& fo : fold)   // the entropy term of an image variable goes to the first op that computes its log-determinant anyway
            if (ent_coef[fo.second] != 0) {
                recs[fo.first].w[W_FLAGS] |= F_FOLD_ENT;
                recs[fo.first].w[W_OUT] = ent_coef[fo.second];
                ent_coef[fo.second] = 0;
            }
        for (int64_t v = 0; v < nv; ++v)
            if (P.vclass[v] == VC_GAUSS && ent_coef[v] != 0) {
                OpRec& r = emit(LF, OP_FE_ENT, P.dim[v]);
                if (push_from[v] >= 0) {
                    const int u = push_from[v];
                    r.w[W_IN0] = P.marg_off[u];
                    r.w[W_FLAGS] |= F_PUSH_A;
                    r.w[W_C0] = const_matrix((int)iface(push_fac[v], 1), P.dim[v], P.dim[u]);
                    r.w[W_D1] = P.dim[u];
                    r.w[W_IN1] = push_logdet(v);
                } else
                    r.w[W_IN0] = P.marg_off[v];
                r.w[W_N] = ent_coef[v];
                r.w[W_TERM] = new_term();
            }
        P.stat_doubles = stat_o;
        // q(W) updates
        const int LP = LF + 1;
        for (int64_t v = 0; v < nv; ++v) {
            if (P.vclass[v] != VC_PREC) continue;
            OpRec& r = emit(LP, OP_PREC_UPDATE, P.dim[v]);
            r.w[W_C0] = prior_off[v];
            r.w[W_PREC] = P.prec_off[v];
            r.w[W_LIST] = (int)P.aux.size();
            r.w[W_N] = (int)prec_stats[v].size();
            bool weighted = false;
            for (int wo : prec_weight[v]) weighted = weighted || wo >= 0;
            if (weighted) r.w[W_FLAGS] |= F_WEIGHT;   // the list holds (moments, weight | −1) pairs: ν = ν0 + Σ π, V⁻¹ = S0⁻¹ + Σ π E[rrᵀ] (the moments arrive weighted)
            for (size_t q = 0; q < prec_stats[v].size(); ++q) {
                P.aux.push_back(prec_stats[v][q]);
                if (weighted) P.aux.push_back(prec_weight[v][q]);
            }
            r.w[W_TERM] = new_term();
        }
        for (const Gcv& gc : gcvs) {   // γ(z) under the new q(z), the GCV average energy
            OpRec& r = emit(LP, OP_GCV_PREC, 1);
            r.w[W_PREC] = gc.state;
            r.w[W_C0] = gcv_constants(gc);
            r.w[W_IN0] = P.marg_off[gc.z];
            r.w[W_C1] = gc.stat;
            r.w[W_TERM] = new_term();
        }
        // q(s) of every probability vector with its switches' terms: −Σ_i Σ_k π_ik E log s_k (new q(s)), −Σ_i H[q(z_i)], the Dirichlet prior node U − H[q(s)];
        // switches with a CONSTANT probability vector: one op per vector with the first two terms
        {
            std::vector<int> owners;
            for (const Mix& mx : mixes)
                if (std::find(owners.begin(), owners.end(), cat_s[mx.z]) == owners.end()) owners.push_back(cat_s[mx.z]);
            for (int sv : owners) {
                int K = 0, cnt = 0;
                const int lst = (int)P.aux.size();
                for (const Mix& mx : mixes)
                    if (cat_s[mx.z] == sv) { P.aux.push_back(P.prec_off[mx.z]); K = mx.K; ++cnt; }
                OpRec& r = emit(LP, OP_DIR_UPDATE, 1);
                r.w[W_N] = K;
                r.w[W_LIST] = lst;
                r.w[W_VAL2] = cnt;
                if (P.vclass[sv] == VC_DIR) {
                    r.w[W_PREC] = P.prec_off[sv];
                    r.w[W_C0] = alpha_off(sv);
                } else
                    r.w[W_C0] = log_probabilities(sv, K);
                r.w[W_TERM] = new_term();
            }
        }
        // fixed-order tree sum of the terms (chunks of 32)
        int lv = LP + 1;
        std::vector<int> cur = terms;
        if (cur.empty()) { cur.push_back((int)P.term_slots++); }
        while (cur.size() > 1) {
            std::vector<int> nxt;
            for (size_t i = 0; i < cur.size(); i += 32) {
                const size_t n = std::min<size_t>(32, cur.size() - i);
                OpRec& r = emit(lv, OP_SUM_TERMS, 1);
                r.w[W_LIST] = (int)P.aux.size();
                r.w[W_N] = (int)n;
                for (size_t q = 0; q < n; ++q) P.aux.push_back(cur[i + q]);
                r.w[W_TERM] = (int)P.term_slots;
                nxt.push_back((int)P.term_slots++);
            }
            cur.swap(nxt);
            ++lv;
        }
        P.fe_root = cur[0];
        P.fe_level = LF;
        // the stored marginals of the image variables: nobody on the device reads them — formed when a caller asks (rxhip_tree_get_marginals), one level behind
        // everything a run executes
        P.n_push = 0;
        for (int64_t v = 0; v < nv; ++v) {
            if (P.vclass[v] != VC_GAUSS || push_from[v] < 0) continue;
            const int f = push_fac[v], u = push_from[v];
            OpRec& r = emit(lv, OP_MARG_PUSH, P.dim[v]);
            r.w[W_D1] = P.dim[u];
            r.w[W_C0] = const_matrix((int)iface(f, 1), P.dim[v], P.dim[u]);
            r.w[W_IN0] = P.marg_off[u];
            r.w[W_OUT] = P.marg_off[v];
            r.w[W_IN1] = push_logdet((int)v);
            ++P.n_push;
        }
        P.lazy_level = P.n_push ? lv : -1;
    }
    int derived_levels = 0;
    std::vector<int> logp_off;   // per constant probability vector: constant-pool offset of its logarithms (−1)
    int log_probabilities(int v, int K) {
        if (logp_off.empty()) logp_off.assign(nv, -1);
        if (logp_off[v] < 0) {
            logp_off[v] = (int)P.cpool.size();
            for (int k = 0; k < K; ++k) P.cpool.push_back(std::log(cptr(v)[k]));
        }
        return logp_off[v];
    }
    std::vector<int> gcv_coff;
    int gcv_constants(const Gcv& gc) {
        const int gi = (int)(&gc - gcvs.data());
        if (gcv_coff.empty()) gcv_coff.assign(gcvs.size(), -1);
        if (gcv_coff[gi] >= 0) return gcv_coff[gi];
        int n = g->gh_points > 0 ? (int)g->gh_points : 31;
        if (n > 64) fail(RXHIP_ERR_UNSUPPORTED, "Gauss–Hermite cubature with more than 64 points");
        // Gauss–Hermite nodes and weights: Newton iteration on the orthonormal recurrence (as the HGF engine's table, csrc/rxhip.hip)
        std::vector<double> x((size_t)n), w((size_t)n);
        const double PIM4 = 0.7511255444649425;
        const int m = (n + 1) / 2;
        double z = 0.0, pp = 0.0;
        for (int i = 0; i < m; ++i) {
            if (i == 0) z = std::sqrt((double)(2 * n + 1)) - 1.85575 * std::pow((double)(2 * n + 1), -0.16667);
            else if (i == 1) z -= 1.14 * std::pow((double)n, 0.426) / z;
            else if (i == 2) z = 1.86 * z - 0.86 * x[0];
            else if (i == 3) z = 1.91 * z - 0.91 * x[1];
            else z = 2.0 * z - x[(size_t)i - 2];
            for (int its = 0; its < 100; ++its) {
                double p1 = PIM4, p2 = 0.0;
                for (int j = 0; j < n; ++j) {
                    const double p3 = p2;
                    p2 = p1;
                    p1 = z * std::sqrt(2.0 / (j + 1)) * p2 - std::sqrt((double)j / (j + 1)) * p3;
                }
                pp = std::sqrt(2.0 * n) * p2;
                const double z1 = z;
                z = z1 - p1 / pp;
                if (std::fabs(z - z1) <= 1e-15 * (1.0 + std::fabs(z))) break;
            }
            x[(size_t)i] = z;
            x[(size_t)(n - 1 - i)] = -z;
            w[(size_t)i] = 2.0 / (pp * pp);
            w[(size_t)(n - 1 - i)] = w[(size_t)i];
        }
        gcv_coff[gi] = (int)P.cpool.size();
        P.cpool.push_back(cptr(gc.kv)[0]);
        P.cpool.push_back(cptr(gc.ov)[0]);
        P.cpool.push_back((double)n);
        P.cpool.insert(P.cpool.end(), x.begin(), x.end());
        for (double wi : w) P.cpool.push_back(wi / 1.7724538509055160273);
        return gcv_coff[gi];
    }
    void weigh(OpRec& r, int f) {   // a component of a mixture node: the op scales its message / energy / residual moments by π_k = q(z = k)
        if (wz[f] < 0) return;
        r.w[W_FLAGS] |= F_WEIGHT;
        r.w[W_LIST] = P.prec_off[wz[f]] + wk[f];
    }
    std::vector<int> push_from, push_fac;   // per variable: the variable whose marginal it is the image of (−1), through which `*` node
    std::vector<int> push_ld;               // per variable: constant-pool offset of 2·log|det A| of a square, nonsingular map (−1: none; −2: not asked yet)
    int push_logdet(int v) {
        if (push_ld.empty()) push_ld.assign(nv, -2);
        if (push_ld[v] != -2) return push_ld[v];
        const int u = push_from[v], a = (int)iface(push_fac[v], 1);
        double ld;
        if (P.dim[v] == P.dim[u] && host_logabsdet(P.dim[v], cptr(a), &ld)) {
            push_ld[v] = (int)P.cpool.size();
            P.cpool.push_back(2.0 * ld);
        } else
            push_ld[v] = -1;
        return push_ld[v];
    }

    // the messages an op reads: (kind 0: descriptor word idx | kind 1: list entry idx, offset, dimension)
    struct In { int kind, idx, off, d; };
    static bool produces_msg(int op) { return op == OP_LEAF || op == OP_NOISE || op == OP_MUL_OUT || op == OP_MUL_IN || op == OP_ADD_OUT || op == OP_ADD_IN || op == OP_SHIFT || op == OP_PRODUCT || op == OP_GCV_Z; }
    std::vector<In> op_inputs(const int* w) const {
        std::vector<In> v;
        switch (w[W_OP]) {
        case OP_NOISE: case OP_SHIFT: case OP_MUL_IN: v.push_back({0, W_IN0, w[W_IN0], w[W_D0]}); break;
        case OP_MUL_OUT: v.push_back({0, W_IN0, w[W_IN0], w[W_D1]}); break;
        case OP_GCV_Z: for (int k : {W_IN0, W_IN1}) v.push_back({0, k, w[k], 1}); break;
        case OP_GCV_ZMARG: v.push_back({0, W_IN0, w[W_IN0], 1}); v.push_back({0, W_IN1, w[W_IN1], 1}); break;
        case OP_ADD_OUT: case OP_ADD_IN: v.push_back({0, W_IN0, w[W_IN0], w[W_D0]}); v.push_back({0, W_IN1, w[W_IN1], w[W_D0]}); break;
        case OP_PRODUCT: case OP_MARGINAL:
            for (int q = 0; q < w[W_N]; ++q) v.push_back({1, q, P.aux[(size_t)w[W_LIST] + 2 * q], w[W_D0]});
            break;
        case OP_FE_NOISE2M:
            if (w[W_IN0] >= 0) v.push_back({0, W_IN0, w[W_IN0], w[W_D0]});
            if ((w[W_FLAGS] & F_JOINT_B) && w[W_IN1] >= 0) v.push_back({0, W_IN1, w[W_IN1], w[W_D0]});
            break;
        case OP_FE_NOISE2: case OP_FE_ADD2:
            for (int k : {W_IN0, W_IN1, W_IN2})
                if ((k != W_IN2 || w[W_OP] == OP_FE_ADD2) && w[k] >= 0) v.push_back({0, k, w[k], w[W_D0]});
            break;
        default: break;
        }
        return v;
    }
    // messages nobody reads (since the marginals of `A * x` outputs stopped being products of messages: the message toward such an output when the node behind
    // it is observed, the product that fed it): their ops go, and then whatever only they were reading
    void eliminate_dead_messages() {
        std::unordered_map<int, int> prod;
        std::vector<int> readers(recs.size(), 0);
        std::vector<char> dead(recs.size(), 0);
        for (size_t i = 0; i < recs.size(); ++i)
            if (produces_msg(recs[i].w[W_OP])) prod[recs[i].w[W_OUT]] = (int)i;
        for (size_t i = 0; i < recs.size(); ++i)
            for (const In& in : op_inputs(recs[i].w)) {
                auto it = prod.find(in.off);
                if (it != prod.end()) ++readers[it->second];
            }
        std::vector<int> work;
        for (size_t i = 0; i < recs.size(); ++i)
            if (produces_msg(recs[i].w[W_OP]) && readers[i] == 0) work.push_back((int)i);
        while (!work.empty()) {
            const int i = work.back();
            work.pop_back();
            if (dead[i]) continue;
            dead[i] = 1;
            for (const In& in : op_inputs(recs[i].w)) {
                auto it = prod.find(in.off);
                if (it != prod.end() && --readers[it->second] == 0) work.push_back(it->second);
            }
        }
        std::vector<OpRec> keep;
        keep.reserve(recs.size());
        for (size_t i = 0; i < recs.size(); ++i)
            if (!dead[i]) keep.push_back(recs[i]);
        recs.swap(keep);
    }

    void finish() {
        eliminate_dead_messages();
        // the sweep's message traffic when every message goes through HBM: 8·(d + d(d+1)/2) per message read or written, + the log-determinant slot of a marginal
        P.bytes_per_sweep = 0;
        P.n_messages = 0;
        for (const OpRec& r : recs) {
            if (r.level >= P.fe_level) continue;
            for (const In& in : op_inputs(r.w)) P.bytes_per_sweep += 8ll * msz(in.d);
            if (produces_msg(r.w[W_OP])) { P.bytes_per_sweep += 8ll * msz(r.w[W_OP] == OP_MUL_IN ? r.w[W_D1] : r.w[W_D0]); ++P.n_messages; }
            else if (r.w[W_OP] == OP_MARGINAL) P.bytes_per_sweep += 8ll * msz(r.w[W_D0]) + 8;
        }
        // the second phase: a message per FE_NOISE2 / FE_ADD2 input, a marginal (mean, packed covariance, log-determinant) or a mean per marginal read, data
        // values, the statistics of the q(W) updates, one double per term written or summed
        P.fe_bytes = 0;
        for (const OpRec& r : recs) {
            if (r.level < P.fe_level || r.w[W_OP] == OP_MARG_PUSH) continue;
            const int* w = r.w;
            const int d = w[W_D0];
            for (const In& in : op_inputs(w)) P.fe_bytes += 8ll * msz(in.d);
            switch (w[W_OP]) {
            case OP_FE_NOISE2M:
                P.fe_bytes += 8ll * ((w[W_FLAGS] & F_PUSH_A) ? w[W_LIST] : d) + 8ll * (msz((w[W_FLAGS] & F_PUSH_B) ? w[W_N] : d) + 1) + 8;
                break;
            case OP_FE_NOISE1:
                P.fe_bytes += 8ll * (msz((w[W_FLAGS] & F_PUSH_A) ? w[W_D1] : d) + 1) + ((w[W_FLAGS] & F_VAL_SLOT) ? 8ll * d : 0) + 8;
                break;
            case OP_FE_NOISE_MF: P.fe_bytes += 16ll * (msz(d) + 1) + 8; break;
            case OP_FE_NOISE0: P.fe_bytes += ((w[W_FLAGS] & F_VAL_SLOT) ? 8ll * d : 0) + ((w[W_FLAGS] & F_VAL2_SLOT) ? 8ll * d : 0) + 8; break;
            case OP_FE_ENT: P.fe_bytes += ((w[W_FLAGS] & F_PUSH_A) ? 8ll * (msz(w[W_D1]) + 1) : 8) + 8; break;
            case OP_FE_NOISE2: case OP_FE_ADD2: P.fe_bytes += 8; break;
            case OP_SUM_TERMS: P.fe_bytes += 8ll * w[W_N] + 8; break;
            case OP_PREC_UPDATE: P.fe_bytes += 8ll * d * d * w[W_N] + 8ll * (2 + d * (d + 1) / 2 + 2 * d * d) + 8; break;
            default: break;
            }
            if (w[W_FLAGS] & F_STAT) P.fe_bytes += 8ll * d * d;
        }
        P.is_push.assign(nv, 0);
        for (int64_t v = 0; v < nv; ++v) P.is_push[v] = push_from[v] >= 0;
        std::stable_sort(recs.begin(), recs.end(), [](const OpRec& a, const OpRec& b) { return a.level != b.level ? a.level < b.level : a.w[W_OP] < b.w[W_OP]; });
        P.n_ops = (int)recs.size();
        for (const OpRec& r : recs) P.fe_heavy = P.fe_heavy || r.w[W_OP] == OP_FE_ADD2 || r.w[W_OP] == OP_PREC_UPDATE || r.w[W_OP] == OP_DIR_UPDATE || r.w[W_OP] == OP_GCV_PREC;
        P.ops.resize((size_t)P.n_ops * OP_WORDS);
        int nl = recs.empty() ? 0 : recs.back().level + 1;
        P.lvl_ptr.assign(nl + 1, 0);
        for (int i = 0; i < P.n_ops; ++i) {
            std::memcpy(&P.ops[(size_t)i * OP_WORDS], recs[i].w, sizeof recs[i].w);
            P.lvl_ptr[recs[i].level + 1]++;
        }
        for (int l = 0; l < nl; ++l) { P.max_width = std::max(P.max_width, P.lvl_ptr[l + 1]); P.lvl_ptr[l + 1] += P.lvl_ptr[l]; }
        P.n_levels = nl;
        if (P.aux.empty()) P.aux.push_back(0);
        if (P.cpool.empty()) P.cpool.push_back(0.0);
    }

    // rule calls as the reference's trace counts them: what the marginals of the NAMED variables pull in (anonymous outputs of deterministic nodes —
    // `B * x[t]` — are not requested by a user; the CPU restatements the tests compare with count the same way)
    void count() {
        std::vector<char> det_out(nv, 0);
        for (int64_t f = 0; f < nf; ++f)
            if (nclass[f] == NC_MUL || nclass[f] == NC_ADD) det_out[iface((int)f, 0)] = 1;
        std::vector<char> dem(2 * E, 0);
        std::vector<int> st;
        std::vector<char> mf_read(nv, 0);   // a mean-field rule reads the marginal of its other interface: that marginal is computed whoever asks, anonymous or not
        for (int64_t f = 0; f < nf; ++f)
            if (mf[f]) mf_read[iface((int)f, 0)] = mf_read[iface((int)f, 1)] = 1;
        for (int64_t v = 0; v < nv; ++v)
            if (P.vclass[v] == VC_GAUSS && (!det_out[v] || mf_read[v])) {
                if (!det_out[v]) ++P.marginals;
                for (int e : var_edges[v]) if (!null_[e] && !dem[e]) { dem[e] = 1; st.push_back(e); }
            }
        for (const Gcv& gc : gcvs) {   // q(z) of a GCV node's volatility input is matched against the product of all its other messages
            const int m = E + fac_edges[gc.fz][0];
            if (!null_[m] && !dem[m]) { dem[m] = 1; st.push_back(m); }
        }
        while (!st.empty()) {
            const int m = st.back(); st.pop_back();
            for (int dpm : deps[m]) if (!null_[dpm] && !dem[dpm]) { dem[dpm] = 1; st.push_back(dpm); }
        }
        for (int m = 0; m < E; ++m) P.rule_calls += dem[m];
        for (int m = E; m < 2 * E; ++m)
            if (dem[m] && alias[m] < 0) {
                int n = 0;
                for (int dpm : deps[m]) n += !null_[dpm];
                P.products += n > 1 ? n - 1 : 0;
            }
    }

    // ---- the strand schedule (tree_kernels.hpp k_tree_strands) ----
    // The sweep's ops (sorted by level: a topological order) are cut into strands: an op joins the strand whose LAST op produced one of its inputs when every
    // other input comes from the same strand or from a strand of a lower strand level — so a lane can walk a whole strand without waiting for anybody, and
    // strands of one level are independent of each other.  Otherwise it opens a strand one level above its inputs'.  Processing in level order makes the op
    // on the critical path (the lowest level among the readers of a message) the one that continues the strand.  A message is written to HBM unless its only
    // reader — sweep or Bethe phase — is the next op of its strand.
    void build_strands() {
        const int n_sweep = P.lvl_ptr[std::min(P.fe_level, P.n_levels)];
        auto W = [&](int i) { return &P.ops[(size_t)i * OP_WORDS]; };
        std::unordered_map<int, int> prod;
        for (int i = 0; i < n_sweep; ++i)
            if (produces_msg(W(i)[W_OP])) prod[W(i)[W_OUT]] = i;
        auto inputs = [&](int i) { return op_inputs(W(i)); };
        std::vector<int> readers(P.n_ops, 0), oplevel(P.n_ops, 0);
        for (int l = 0; l < P.n_levels; ++l)
            for (int i = P.lvl_ptr[l]; i < P.lvl_ptr[l + 1]; ++i) oplevel[i] = l;
        for (int i = 0; i < P.n_ops; ++i)
            for (const In& in : inputs(i)) {
                auto it = prod.find(in.off);
                if (it != prod.end()) ++readers[it->second];
            }
        std::vector<int> strand_of(n_sweep, -1), reg_kind(n_sweep, -1), reg_idx(n_sweep, -1), next_in_strand(n_sweep, -1);
        std::vector<std::vector<int>> members;
        std::vector<int> slevel, tail;
        const int L0 = derived_levels;
        for (int i = 0; i < n_sweep; ++i) {
            const int op = W(i)[W_OP];
            const std::vector<In> ins = inputs(i);
            int best = -1, best_k = -1;
            for (size_t k = 0; k < ins.size(); ++k) {
                auto it = prod.find(ins[k].off);
                if (it == prod.end()) fail(RXHIP_ERR_BADARG, "internal: op %d reads a message nobody produces", i);
                const int j = it->second;
                if (tail[strand_of[j]] != j) continue;
                if (best < 0 || oplevel[j] > oplevel[best]) { best = j; best_k = (int)k; }
            }
            if (best >= 0) {
                const int sb = strand_of[best];
                for (size_t k = 0; k < ins.size() && best >= 0; ++k) {
                    const int s2 = strand_of[prod[ins[k].off]];
                    if (s2 != sb && slevel[s2] >= slevel[sb]) best = -1;
                }
            }
            if (best >= 0) {
                const int sb = strand_of[best];
                strand_of[i] = sb;
                members[sb].push_back(i);
                tail[sb] = i;
                next_in_strand[best] = i;
                reg_kind[i] = ins[best_k].kind;
                reg_idx[i] = ins[best_k].idx;
            } else {
                int lv = (op == OP_DERIVE_MUL || op == OP_DERIVE_ADD || op == OP_CAT_UPDATE) ? oplevel[i] : L0;
                for (const In& in : ins) lv = std::max(lv, slevel[strand_of[prod[in.off]]] + 1);
                strand_of[i] = (int)members.size();
                members.push_back({i});
                slevel.push_back(lv);
                tail.push_back(i);
            }
        }
        const int ns = (int)members.size();
        std::vector<int> order(ns);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return slevel[a] != slevel[b] ? slevel[a] < slevel[b] : members[a].size() > members[b].size(); });
        const int nsl = ns ? slevel[order.back()] + 1 : 0;
        P.slvl_ptr.assign(nsl + 1, 0);
        P.sops.reserve((size_t)n_sweep * OP_WORDS);
        auto msz8 = [&](int d) { return 8ll * msz(d); };
        for (int s : order) {
            ++P.slvl_ptr[slevel[s] + 1];
            P.strands.push_back((int)(P.sops.size() / OP_WORDS));
            P.strands.push_back((int)members[s].size());
            P.longest_strand = std::max(P.longest_strand, (int)members[s].size());
            for (int i : members[s]) {
                int w[OP_WORDS];
                std::memcpy(w, W(i), sizeof w);
                const std::vector<In> ins = inputs(i);
                if (reg_kind[i] == 0) w[reg_idx[i]] = OFF_REG;
                else if (reg_kind[i] == 1) {   // a private copy of the list with the register entry
                    const int l0 = w[W_LIST], nn = w[W_N];
                    w[W_LIST] = (int)P.aux.size();
                    for (int q = 0; q < nn; ++q) {
                        P.aux.push_back(q == reg_idx[i] ? OFF_REG : P.aux[(size_t)l0 + 2 * q]);
                        P.aux.push_back(P.aux[(size_t)l0 + 2 * q + 1]);
                    }
                }
                for (size_t k = 0; k < ins.size(); ++k)
                    if (!(ins[k].kind == reg_kind[i] && ins[k].idx == reg_idx[i])) P.bytes_per_sweep_strands += msz8(ins[k].d);
                const int op = w[W_OP];
                if (produces_msg(op)) {
                    const int dout = op == OP_MUL_IN ? w[W_D1] : w[W_D0];
                    if (readers[i] == 1 && next_in_strand[i] >= 0) w[W_FLAGS] |= F_NO_STORE;
                    else P.bytes_per_sweep_strands += msz8(dout);
                } else if (op == OP_MARGINAL) P.bytes_per_sweep_strands += msz8(w[W_D0]) + 8;
                P.sops.insert(P.sops.end(), w, w + OP_WORDS);
            }
        }
        for (int l = 0; l < nsl; ++l) P.slvl_ptr[l + 1] += P.slvl_ptr[l];
        if (P.sops.empty()) P.sops.assign(OP_WORDS, 0);
        if (P.strands.empty()) P.strands.assign(2, 0);
        // the floor of any schedule: the data in, the posteriors of the named variables (mean, packed covariance, log-determinant slot) out
        std::vector<char> det_out(nv, 0);
        for (int64_t f = 0; f < nf; ++f)
            if (nclass[f] == NC_MUL || nclass[f] == NC_ADD) det_out[iface((int)f, 0)] = 1;
        P.io_bytes = 8ll * P.data_doubles;
        for (int64_t v = 0; v < nv; ++v)
            if (P.vclass[v] == VC_GAUSS && !det_out[v]) P.io_bytes += msz8(P.dim[v]) + 8;
    }

    void compile() {
        parse();
        check_family();
        build_edges();
        build_deps();
        analyse();
        allocate();
        init_precision();
        init_marginals();
        emit_all();
        finish();
        count();
        build_strands();
    }
};

}  // namespace

}  // namespace tree
}  // namespace rxhip
