/*
 * rxoracle.c — CPU restatement of the ReactiveMP Gaussian sum-product rules, message product,
 * marginals and Bethe free energy that RxInfer fires on a linear Gaussian state-space model.
 *
 * TEST INFRASTRUCTURE ONLY (see rxoracle.h).  fp64, plain C, written for clarity: every
 * function below names the reference rule / call site it restates.  The message schedule is
 * the reference's: per time step 6 rule calls, 4 pairwise products,
 * 1 marginal; parametrisations change exactly where ExponentialFamily changes them
 * (mean/covariance <-> weighted-mean/precision through `cholinv`).
 */
#define _GNU_SOURCE   /* madvise(MADV_HUGEPAGE) */
#include "rxoracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define LOG2PI 1.8378770664093454835606594728112

const char* rxo_version(void) { return "rxoracle 0.1 (restates ReactiveMP ~6.0 / ExponentialFamily 2.1 rules)"; }

/* ------------------------------------------------------------------------------------------
 * FastCholesky.jl restatement: cholinv / chollogdet (FastCholesky.jl 1.3.0, `cholinv(x) =
 * inv(fastcholesky(x))`).  Non-SPD input is an error, as in the reference CI
 * (.github/workflows/CI.yml:72, JULIA_FASTCHOLESKY_THROW_ERROR_NON_SYMMETRIC=1).
 * ------------------------------------------------------------------------------------------ */
static int chol_lower(int n, const double* A, double* L) {
    memset(L, 0, sizeof(double) * (size_t)n * n);
    for (int j = 0; j < n; ++j) {
        double s = A[j * n + j];
        for (int k = 0; k < j; ++k) s -= L[j * n + k] * L[j * n + k];
        if (!(s > 0.0)) return RXO_ERR_NOT_POSDEF;
        double ljj = sqrt(s);
        L[j * n + j] = ljj;
        for (int i = j + 1; i < n; ++i) {
            double t = A[i * n + j];
            for (int k = 0; k < j; ++k) t -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = t / ljj;
        }
    }
    return RXO_OK;
}

/* out = inv(A) for SPD A, logdet (nullable) = log det A.  work: 2*n*n doubles. */
static int cholinv(int n, const double* A, double* out, double* logdet, double* work) {
    double* L = work;
    double* Li = work + (size_t)n * n;
    int rc = chol_lower(n, A, L);
    if (rc) return rc;
    if (logdet) {
        double ld = 0.0;
        for (int i = 0; i < n; ++i) ld += log(L[i * n + i]);
        *logdet = 2.0 * ld;
    }
    /* Li = inv(L) (lower) */
    memset(Li, 0, sizeof(double) * (size_t)n * n);
    for (int j = 0; j < n; ++j) {
        Li[j * n + j] = 1.0 / L[j * n + j];
        for (int i = j + 1; i < n; ++i) {
            double s = 0.0;
            for (int k = j; k < i; ++k) s -= L[i * n + k] * Li[k * n + j];
            Li[i * n + j] = s / L[i * n + i];
        }
    }
    /* out = Li' * Li */
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = 0.0;
            for (int k = i; k < n; ++k) s += Li[k * n + i] * Li[k * n + j];
            out[i * n + j] = s;
            out[j * n + i] = s;
        }
    return RXO_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* small dense helpers (row-major)                                                            */
static void matvec(int n, int m, const double* A, const double* x, double* y) { /* y = A x, A n×m */
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int k = 0; k < m; ++k) s += A[i * m + k] * x[k];
        y[i] = s;
    }
}
static void matTvec(int n, int m, const double* A, const double* x, double* y) { /* y = A' x, A n×m */
    for (int k = 0; k < m; ++k) y[k] = 0.0;
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < m; ++k) y[k] += A[i * m + k] * x[i];
}
/* C = A S A'  with A n×m, S m×m ; tmp n*m */
static void congruence(int n, int m, const double* A, const double* S, double* C, double* tmp) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) {
            double s = 0.0;
            for (int k = 0; k < m; ++k) s += A[i * m + k] * S[k * m + j];
            tmp[i * m + j] = s;
        }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int k = 0; k < m; ++k) s += tmp[i * m + k] * A[j * m + k];
            C[i * n + j] = s;
        }
}
/* C = A' S A with A n×m, S n×n -> C m×m ; tmp n*m */
static void congruenceT(int n, int m, const double* A, const double* S, double* C, double* tmp) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) {
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += S[i * n + k] * A[k * m + j];
            tmp[i * m + j] = s;
        }
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) {
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += A[k * m + i] * tmp[k * m + j];
            C[i * m + j] = s;
        }
}

/* ------------------------------------------------------------------------------------------
 * ExponentialFamily.jl restatement: the two Gaussian parametrisations on the path and their
 * conversions:
 *   MvNormalMeanCovariance(μ,Σ)            <->   MvNormalWeightedMeanPrecision(ξ,Λ), ξ = Λμ
 *   mean_cov(ξ,Λ): Σ = cholinv(Λ), μ = Σξ       weightedmean_precision(μ,Σ): Λ = cholinv(Σ), ξ = Λμ
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int n;
    double* mean; /* or ξ */
    double* mat;  /* Σ or Λ */
} gauss;

typedef struct {
    rxo_counters c;
    double* work; /* scratch */
    int count;
} ctx;

static int to_other_param(int n, const double* v, const double* M, double* v2, double* M2, double* logdetM,
                          double* work) {
    int rc = cholinv(n, M, M2, logdetM, work);
    if (rc) return rc;
    matvec(n, n, M2, v, v2);
    return RXO_OK;
}

/* BayesBase.prod(GenericProd, Gaussian, Gaussian) -> MvNormalWeightedMeanPrecision(ξ1+ξ2, Λ1+Λ2)
 * (prod constraint GenericProd from src/constraints/form/form_ensure_supported.jl:13) */
static void prod_wmp(int n, const double* x1, const double* L1, const double* x2, const double* L2, double* x,
                     double* L, ctx* c) {
    for (int i = 0; i < n; ++i) x[i] = x1[i] + x2[i];
    for (int i = 0; i < n * n; ++i) L[i] = L1[i] + L2[i];
    if (c->count) c->c.products++;
}

/* @rule MvNormalMeanCovariance(:out, Marginalisation) (m_μ, q_Σ::PointMass) = N(mean(m_μ), cov(m_μ)+Σ)
 * @rule MvNormalMeanCovariance(:μ,   Marginalisation) (m_out, q_Σ::PointMass): same arithmetic      */
static void rule_mvn_additive(int n, const double* mean_in, const double* cov_in, const double* Sigma,
                              double* mean_out, double* cov_out, ctx* c) {
    for (int i = 0; i < n; ++i) mean_out[i] = mean_in[i];
    for (int i = 0; i < n * n; ++i) cov_out[i] = (cov_in ? cov_in[i] : 0.0) + Sigma[i];
    if (c->count) c->c.rule_calls++;
}
/* @rule typeof(*)(:out, Marginalisation) (m_A::PointMass, m_in) = N(Aμ, AΣA')   A: n×m */
static void rule_mul_out(int n, int m, const double* A, const double* mean_in, const double* cov_in,
                         double* mean_out, double* cov_out, ctx* c) {
    matvec(n, m, A, mean_in, mean_out);
    congruence(n, m, A, cov_in, cov_out, c->work);
    if (c->count) c->c.rule_calls++;
}
/* @rule typeof(*)(:in, Marginalisation) (m_out, m_A::PointMass) = WMP(A'ξ, A'ΛA)   A: n×m */
static void rule_mul_in(int n, int m, const double* A, const double* xi_out, const double* L_out,
                        double* xi_in, double* L_in, ctx* c) {
    matTvec(n, m, A, xi_out, xi_in);
    congruenceT(n, m, A, L_out, L_in, c->work);
    if (c->count) c->c.rule_calls++;
}

/* entropy(MvNormal) = ½(n·log(2πe) + logdet Σ) */
static double gauss_entropy_from_logdetcov(int n, double logdetcov) {
    return 0.5 * (n * (LOG2PI + 1.0) + logdetcov);
}

/* BayesBase.CountingReal: finite part + number of infinities */
typedef struct {
    double v;
    long ninf;
} creal;

/* ------------------------------------------------------------------------------------------ */
/* optional tap on the node-local joints q(out, μ) of the transition nodes (set by rxo_lgssm_bp_joints) */
static __thread double* g_joint_mean = NULL; /* [T-1][2d]     */
static __thread double* g_joint_cov = NULL;  /* [T-1][2d][2d] */

/* The message stores of one chain (16 MB + 96 MB at d = 4, T = 10⁵).  rxo_lgssm_bp_batch keeps ONE set per thread for all the chains the
   thread runs: with a fresh malloc per chain the all-core timing of bench.py's cpu_baseline was page faults and mmap-lock contention
   (6.7× one core on 256 cores). */
typedef struct { double *work, *big, *fwdp, *fwdx; } bp_ws;
static void bp_ws_free(bp_ws* w) {
    free(w->work); free(w->big); free(w->fwdp); free(w->fwdx);
    memset(w, 0, sizeof *w);
}
/* message stores: 2 MB-aligned with transparent huge pages asked for (a page fault per 4 KB of a fresh 100 MB block, taken by every
   thread of the process at once, serialises on the address-space lock) */
static double* store_alloc(size_t bytes) {
    void* p = NULL;
    if (bytes < ((size_t)4 << 20)) return (double*)malloc(bytes);
    if (posix_memalign(&p, (size_t)2 << 20, (bytes + (((size_t)2 << 20) - 1)) & ~(((size_t)2 << 20) - 1))) return NULL;
#ifdef MADV_HUGEPAGE
    (void)madvise(p, bytes, MADV_HUGEPAGE);
#endif
    return (double*)p;
}
static int bp_ws_reserve(bp_ws* w, int d, int dy, int T, int ptt, int want_fe) {
    const size_t n = (size_t)T + (ptt ? 1 : 0), dm = (size_t)(d > dy ? d : dy), vs = (size_t)d, ms = (size_t)d * d;
    memset(w, 0, sizeof *w);
    w->work = (double*)malloc(sizeof(double) * (16 * dm * dm + 64));
    w->big = (double*)malloc(sizeof(double) * (40 * dm * dm + 64));
    w->fwdp = store_alloc(sizeof(double) * n * (vs + ms));
    if (want_fe) w->fwdx = store_alloc(sizeof(double) * n * (vs + ms) * 6 + sizeof(double) * n);
    if (!w->work || !w->big || !w->fwdp || (want_fe && !w->fwdx)) { bp_ws_free(w); return RXO_ERR_BADARG; }
    return RXO_OK;
}
static int lgssm_bp_ws(int d, int dy, int T, const double* A, const double* B, const double* P, const double* Q,
                       const double* m0, const double* V0, int ptt, const double* y, double* post_mean,
                       double* post_cov, double* free_energy, rxo_counters* counters, bp_ws* ws);
int rxo_lgssm_bp(int d, int dy, int T, const double* A, const double* B, const double* P, const double* Q,
                 const double* m0, const double* V0, int ptt, const double* y, double* post_mean,
                 double* post_cov, double* free_energy, rxo_counters* counters) {
    if (d <= 0 || dy <= 0 || T <= 0) return RXO_ERR_BADARG;
    bp_ws ws;
    int rc = bp_ws_reserve(&ws, d, dy, T, ptt, free_energy != NULL);
    if (rc) return rc;
    rc = lgssm_bp_ws(d, dy, T, A, B, P, Q, m0, V0, ptt, y, post_mean, post_cov, free_energy, counters, &ws);
    bp_ws_free(&ws);
    return rc;
}
static int lgssm_bp_ws(int d, int dy, int T, const double* A, const double* B, const double* P, const double* Q,
                       const double* m0, const double* V0, int ptt, const double* y, double* post_mean,
                       double* post_cov, double* free_energy, rxo_counters* counters, bp_ws* ws) {
    const int n = T + (ptt ? 1 : 0); /* number of state variables; state k observes y[k-ptt] */
    const int want_fe = free_energy != NULL;
    const int dm = d > dy ? d : dy;
    const size_t vs = (size_t)d, ms = (size_t)d * d;
    int rc = RXO_OK;

    ctx c;
    memset(&c, 0, sizeof c);
    c.count = 1;
    c.work = ws->work;
    double* big = ws->big;

    /* stored messages */
    double* fwdp_x = ws->fwdp; /* (fwd ⊗ obs) as WMP */
    double* fwdp_L = fwdp_x + n * vs;
    double *fwdx_m = NULL, *fwdx_V = NULL, *amsg_m = NULL, *amsg_V = NULL, *tox_x = NULL, *tox_L = NULL,
           *mum_m = NULL, *mum_V = NULL, *bwd_x = NULL, *bwd_L = NULL, *q_m = NULL, *q_V = NULL, *q_ld = NULL;
    if (want_fe) {
        fwdx_m = ws->fwdx;
        fwdx_V = fwdx_m + n * vs;
        amsg_m = fwdx_V + n * ms;
        amsg_V = amsg_m + n * vs;
        tox_x = amsg_V + n * ms;
        tox_L = tox_x + n * vs;
        mum_m = tox_L + n * ms;
        mum_V = mum_m + n * vs;
        bwd_x = mum_V + n * ms;
        bwd_L = bwd_x + n * vs;
        q_m = bwd_L + n * ms;
        q_V = q_m + n * vs;
        q_ld = q_V + n * ms;
    }

    /* scratch vectors / matrices (sized for max(d,dy)) */
    double* t_m = big;                  /* mean-like */
    double* t_V = t_m + dm;             /* cov-like  */
    double* t_x = t_V + dm * dm;        /* ξ-like    */
    double* t_L = t_x + dm;             /* Λ-like    */
    double* f_m = t_L + dm * dm;        /* current fwd_x mean */
    double* f_V = f_m + dm;             /* current fwd_x cov  */
    double* o_x = f_V + dm * dm;        /* obs ξ */
    double* o_L = o_x + dm;             /* obs Λ */
    double* a_m = o_L + dm * dm;
    double* a_V = a_m + dm;
    double* ny_m = a_V + dm * dm;       /* N(y,Q) mean (dy) */
    double* ny_V = ny_m + dm;           /* N(y,Q) cov (dy×dy) */
    double* ny_x = ny_V + dm * dm;
    double* ny_L = ny_x + dm;
    double* b_x = ny_L + dm * dm;       /* backward ξ */
    double* b_L = b_x + dm;
    double* u_x = b_L + dm * dm;
    double* u_L = u_x + dm;
    double* chw = u_L + dm * dm;        /* cholinv work, 2*(2dm)^2 = 8 dm^2 */

    /* ---------------- forward pass ---------------- */
    for (int k = 0; k < n; ++k) {
        if (k == 0) {
            /* prior node: @rule MvNormalMeanCovariance(:out)(q_μ::PointMass, q_Σ::PointMass) */
            rule_mvn_additive(d, m0, NULL, V0, f_m, f_V, &c);
        } else {
            /* x[k-1] -> `*`_A : message already formed as fwdp[k-1] (WMP). `*`(:out) needs mean/cov */
            rc = to_other_param(d, fwdp_x + (k - 1) * vs, fwdp_L + (k - 1) * ms, t_m, t_V, NULL, chw);
            if (rc) goto done;
            rule_mul_out(d, d, A, t_m, t_V, a_m, a_V, &c);
            rule_mvn_additive(d, a_m, a_V, P, f_m, f_V, &c);
            if (want_fe) {
                memcpy(amsg_m + k * vs, a_m, sizeof(double) * vs);
                memcpy(amsg_V + k * ms, a_V, sizeof(double) * ms);
            }
        }
        if (want_fe) {
            memcpy(fwdx_m + k * vs, f_m, sizeof(double) * vs);
            memcpy(fwdx_V + k * ms, f_V, sizeof(double) * ms);
        }
        /* fwd_x as WMP (prod converts a mean/cov operand with weightedmean_precision) */
        rc = to_other_param(d, f_m, f_V, t_x, t_L, NULL, chw);
        if (rc) goto done;
        if (k >= ptt) {
            const double* yk = y + (size_t)(k - ptt) * dy;
            /* MvN_y(:μ) with observed out: N(y, Q); then `*`_B(:in) */
            rule_mvn_additive(dy, yk, NULL, Q, ny_m, ny_V, &c);
            rc = to_other_param(dy, ny_m, ny_V, ny_x, ny_L, NULL, chw);
            if (rc) goto done;
            rule_mul_in(dy, d, B, ny_x, ny_L, o_x, o_L, &c);
            /* outbound toward next `*`_A: product (fwd_x ⊗ obs_x), left to right */
            prod_wmp(d, t_x, t_L, o_x, o_L, fwdp_x + k * vs, fwdp_L + k * ms, &c);
        } else {
            memcpy(fwdp_x + k * vs, t_x, sizeof(double) * vs);
            memcpy(fwdp_L + k * ms, t_L, sizeof(double) * ms);
        }
    }
    /* the last variable has no outgoing `*`_A, so its (fwd ⊗ obs) product toward it is never
       requested by a subscriber; it is formed for the marginal instead (counted there) */
    if (c.count && n - 1 >= ptt) c.c.products--;

    /* ---------------- backward pass + marginals ---------------- */
    int have_bwd = 0;
    for (int k = n - 1; k >= 0; --k) {
        const int has_obs = k >= ptt;
        /* marginal q(x_k) = ((fwd ⊗ obs) ⊗ bwd), left-to-right fold (reactivemp_inference.jl:365-374) */
        if (has_obs && c.count) c.c.products++; /* (fwd ⊗ obs) inside the marginal fold */
        if (have_bwd)
            prod_wmp(d, fwdp_x + k * vs, fwdp_L + k * ms, b_x, b_L, u_x, u_L, &c);
        else {
            memcpy(u_x, fwdp_x + k * vs, sizeof(double) * vs);
            memcpy(u_L, fwdp_L + k * ms, sizeof(double) * ms);
        }
        double ldL;
        rc = to_other_param(d, u_x, u_L, t_m, t_V, &ldL, chw);
        if (rc) goto done;
        if (c.count) c.c.marginals++;
        if (k >= ptt) {
            memcpy(post_mean + (size_t)(k - ptt) * vs, t_m, sizeof(double) * vs);
            memcpy(post_cov + (size_t)(k - ptt) * ms, t_V, sizeof(double) * ms);
        }
        if (want_fe) {
            memcpy(q_m + k * vs, t_m, sizeof(double) * vs);
            memcpy(q_V + k * ms, t_V, sizeof(double) * ms);
            q_ld[k] = -ldL; /* logdet cov */
            if (have_bwd) {
                memcpy(bwd_x + k * vs, b_x, sizeof(double) * vs);
                memcpy(bwd_L + k * ms, b_L, sizeof(double) * ms);
            }
        }
        /* message x_k -> MvN_x(k) (out interface): product of the other inbound messages (obs, bwd) */
        if (has_obs) {
            const double* yk = y + (size_t)(k - ptt) * dy;
            /* obs message is memoised in the reference; recomputed here without counting */
            c.count = 0;
            rule_mvn_additive(dy, yk, NULL, Q, ny_m, ny_V, &c);
            rc = to_other_param(dy, ny_m, ny_V, ny_x, ny_L, NULL, chw);
            if (rc) goto done;
            rule_mul_in(dy, d, B, ny_x, ny_L, o_x, o_L, &c);
            c.count = 1;
        }
        /* (only MvN_x(k), k >= 1, subscribes to this message: the prior node's other interfaces
           are constants, and messages toward constants are never computed) */
        if (k >= 1) {
            if (has_obs && have_bwd)
                prod_wmp(d, o_x, o_L, b_x, b_L, u_x, u_L, &c);
            else if (has_obs) {
                memcpy(u_x, o_x, sizeof(double) * vs);
                memcpy(u_L, o_L, sizeof(double) * ms);
            } else {
                memcpy(u_x, b_x, sizeof(double) * vs);
                memcpy(u_L, b_L, sizeof(double) * ms);
            }
            if (want_fe) {
                memcpy(tox_x + k * vs, u_x, sizeof(double) * vs);
                memcpy(tox_L + k * ms, u_L, sizeof(double) * ms);
            }
        }
        if (k >= 1) {
            /* MvN_x(k)(:μ)(m_out, q_Σ) = N(mean(m_out), cov(m_out) + P) */
            rc = to_other_param(d, u_x, u_L, t_m, t_V, NULL, chw);
            if (rc) goto done;
            rule_mvn_additive(d, t_m, t_V, P, a_m, a_V, &c);
            if (want_fe) {
                memcpy(mum_m + k * vs, a_m, sizeof(double) * vs);
                memcpy(mum_V + k * ms, a_V, sizeof(double) * ms);
            }
            /* `*`_A(k)(:in)(m_out, m_A) = WMP(A'ξ, A'ΛA) */
            rc = to_other_param(d, a_m, a_V, t_x, t_L, NULL, chw);
            if (rc) goto done;
            rule_mul_in(d, d, A, t_x, t_L, b_x, b_L, &c);
            have_bwd = 1;
        }
    }
    if (counters) *counters = c.c;

    /* ---------------- Bethe free energy (reactivemp_free_energy.jl:51-126) ---------------- */
    if (want_fe) {
        c.count = 0;
        creal nodes = {0.0, 0}, vars = {0.0, 0};
        long point_entropies = 0;
        const int d2 = 2 * d;
        double* Lj = (double*)malloc(sizeof(double) * (size_t)(3 * d2 * d2 + 4 * d2) + sizeof(double) * 8 * d2 * d2);
        double* Vj = Lj + d2 * d2;
        double* xj = Vj + d2 * d2;
        double* mj = xj + d2;
        double* jw = mj + d2; /* cholinv work 2*(2d)^2 = 8 d^2 */
        double* Wp = jw + 8 * d * d + d2 * d2; /* cholinv(P) */
        double ldP, ldV0, ldQ;
        double* Wq = (double*)malloc(sizeof(double) * (size_t)(dy * dy + d * d));
        double* W0 = Wq + dy * dy;
        rc = cholinv(d, P, Wp, &ldP, chw);
        if (!rc) rc = cholinv(dy, Q, Wq, &ldQ, chw);
        if (!rc) rc = cholinv(d, V0, W0, &ldV0, chw);
        for (int k = 0; k < n && !rc; ++k) {
            const int has_obs = k >= ptt;
            const double* qm = q_m + k * vs;
            const double* qV = q_V + k * ms;
            const double Hx = gauss_entropy_from_logdetcov(d, q_ld[k]);
            if (k == 0) {
                /* prior node MvNormalMeanCovariance(out = x, μ = const, Σ = const):
                   U = ½[d log2π + logdet V0 + tr(V0⁻¹ (V + (m-m0)(m-m0)'))], clusters: (out), (μ), (Σ) */
                double tr = 0.0;
                for (int i = 0; i < d; ++i)
                    for (int j = 0; j < d; ++j)
                        tr += W0[i * d + j] * (qV[j * d + i] + (qm[j] - m0[j]) * (qm[i] - m0[i]));
                nodes.v += 0.5 * (d * LOG2PI + ldV0 + tr) - Hx;
                nodes.ninf += 2; /* -H[PointMass μ] - H[PointMass Σ] */
                point_entropies += 2;
            } else {
                /* `*`_A(k): deterministic node, contribution −H[q(in)] (+∞ counted for const A) */
                nodes.v += -gauss_entropy_from_logdetcov(d, q_ld[k - 1]);
                nodes.ninf += 1;
                point_entropies += 1;
                /* MvN_x(k): joint q(out, μ) from @marginalrule MvNormalMeanCovariance(:out_μ)
                   (SURVEY Appendix A.3): Λj = [[Λo+W, −W],[−W, Λμ+W]], ξj = [ξo; ξμ] */
                rc = to_other_param(d, amsg_m + k * vs, amsg_V + k * ms, t_x, t_L, NULL, chw); /* m_μ as WMP */
                if (rc) break;
                const double* xo = tox_x + k * vs;
                const double* Lo = tox_L + k * ms;
                for (int i = 0; i < d; ++i) {
                    xj[i] = xo[i];
                    xj[d + i] = t_x[i];
                    for (int j = 0; j < d; ++j) {
                        Lj[i * d2 + j] = Lo[i * d + j] + Wp[i * d + j];
                        Lj[i * d2 + d + j] = -Wp[i * d + j];
                        Lj[(d + i) * d2 + j] = -Wp[i * d + j];
                        Lj[(d + i) * d2 + d + j] = t_L[i * d + j] + Wp[i * d + j];
                    }
                }
                double ldLj;
                rc = cholinv(d2, Lj, Vj, &ldLj, jw);
                if (rc) break;
                matvec(d2, d2, Vj, xj, mj);
                if (g_joint_mean && k - 1 - ptt >= 0) { /* node between the observed states k-1 and k */
                    memcpy(g_joint_mean + (size_t)(k - 1 - ptt) * d2, mj, sizeof(double) * d2);
                    memcpy(g_joint_cov + (size_t)(k - 1 - ptt) * d2 * d2, Vj, sizeof(double) * d2 * d2);
                }
                double tr = 0.0;
                for (int i = 0; i < d; ++i)
                    for (int j = 0; j < d; ++j) {
                        double e = Vj[j * d2 + i] - Vj[j * d2 + d + i] - Vj[(d + j) * d2 + i] +
                                   Vj[(d + j) * d2 + d + i] + (mj[j] - mj[d + j]) * (mj[i] - mj[d + i]);
                        tr += Wp[i * d + j] * e;
                    }
                double U = 0.5 * (d * LOG2PI + ldP + tr);
                nodes.v += U - gauss_entropy_from_logdetcov(d2, -ldLj);
                nodes.ninf += 1; /* Σ = const P */
                point_entropies += 1;
                /* variable a_k (anonymous, degree 2): + H[q(a_k)], q(a_k) = prod(`*`_A(:out), MvN_x(:μ)) */
                rc = to_other_param(d, mum_m + k * vs, mum_V + k * ms, u_x, u_L, NULL, chw);
                if (rc) break;
                for (int i = 0; i < d * d; ++i) u_L[i] += t_L[i];
                double lda;
                rc = cholinv(d, u_L, t_V, &lda, chw);
                if (rc) break;
                vars.v += gauss_entropy_from_logdetcov(d, -lda);
            }
            int deg = 1 + (has_obs ? 1 : 0) + (k < n - 1 ? 1 : 0);
            vars.v += (deg - 1) * Hx;
            if (has_obs) {
                const double* yk = y + (size_t)(k - ptt) * dy;
                /* `*`_B(k): −H[q(x_k)] */
                nodes.v += -Hx;
                nodes.ninf += 1;
                point_entropies += 1;
                /* message x_k -> `*`_B = prod(fwd_x, bwd_x); `*`_B(:out) = N(Bm, BVB') */
                rc = to_other_param(d, fwdx_m + k * vs, fwdx_V + k * ms, t_x, t_L, NULL, chw);
                if (rc) break;
                if (k < n - 1) {
                    for (int i = 0; i < d; ++i) t_x[i] += bwd_x[k * vs + i];
                    for (int i = 0; i < d * d; ++i) t_L[i] += bwd_L[k * ms + i];
                }
                rc = to_other_param(d, t_x, t_L, t_m, t_V, NULL, chw);
                if (rc) break;
                rule_mul_out(dy, d, B, t_m, t_V, a_m, a_V, &c); /* dy-dim */
                /* q(b_k) = prod(N(Bm,BVB'), N(y,Q)) */
                rc = to_other_param(dy, a_m, a_V, u_x, u_L, NULL, chw);
                if (rc) break;
                matvec(dy, dy, Wq, yk, ny_x);
                for (int i = 0; i < dy; ++i) u_x[i] += ny_x[i];
                for (int i = 0; i < dy * dy; ++i) u_L[i] += Wq[i];
                double ldb;
                rc = to_other_param(dy, u_x, u_L, t_m, t_V, &ldb, chw);
                if (rc) break;
                double Hb = gauss_entropy_from_logdetcov(dy, -ldb);
                /* MvN_y(k): out = PointMass(y): U = ½[dy log2π + logdet Q + tr(Q⁻¹(Vb + (y−mb)(y−mb)'))] */
                double tr = 0.0;
                for (int i = 0; i < dy; ++i)
                    for (int j = 0; j < dy; ++j)
                        tr += Wq[i * dy + j] * (t_V[j * dy + i] + (yk[j] - t_m[j]) * (yk[i] - t_m[i]));
                nodes.v += 0.5 * (dy * LOG2PI + ldQ + tr) - Hb;
                nodes.ninf += 2; /* −H[PointMass y] − H[PointMass Q] */
                point_entropies += 2;
                vars.v += Hb; /* variable b_k, degree 2 */
            }
        }
        if (!rc) {
            /* float(nodes + vars − point_entropies): infinities must cancel exactly */
            long ninf = nodes.ninf + vars.ninf - point_entropies;
            double fe = nodes.v + vars.v;
            if (ninf != 0) fe = ninf > 0 ? INFINITY : -INFINITY;
            *free_energy = fe;
            if (!isfinite(fe)) rc = RXO_ERR_NONFINITE_FE;
        }
        free(Lj);
        free(Wq);
    }

done:   /* the message stores belong to the caller's workspace */
    return rc;
}

/* Node-local joint marginals q(out = x[t], μ = A x[t-1]) of the transition nodes between observed states, t = 2..T, exactly as
   the @marginalrule MvNormalMeanCovariance(:out_μ) of the Bethe sum above forms them: mean [T-1][2d],
   cov [T-1][2d][2d], (out, μ) order.  Test infrastructure: checks rxhip_get_node_marginals. */
int rxo_lgssm_bp_joints(int d, int dy, int T, const double* A, const double* B, const double* P, const double* Q,
                        const double* m0, const double* V0, int ptt, const double* y, double* joint_mean, double* joint_cov) {
    if (!joint_mean || !joint_cov) return RXO_ERR_BADARG;
    double* pm = (double*)malloc(sizeof(double) * (size_t)T * (d + (size_t)d * d));
    double fe = 0.0;
    g_joint_mean = joint_mean;
    g_joint_cov = joint_cov;
    int rc = rxo_lgssm_bp(d, dy, T, A, B, P, Q, m0, V0, ptt, y, pm, pm + (size_t)T * d, &fe, NULL);
    g_joint_mean = g_joint_cov = NULL;
    free(pm);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
int rxo_lgssm_bp_batch(int d, int dy, int T, int n_chains, const double* A, const double* B, const double* P,
                       const double* Q, const double* m0, const double* V0, int ptt, const double* y,
                       double* post_mean, double* post_cov, double* fe, int nthreads, rxo_counters* counters) {
    if (n_chains <= 0) return RXO_ERR_BADARG;
    int rc_all = RXO_OK;
    uint64_t rc_rules = 0, rc_prods = 0, rc_margs = 0;
    if (d <= 0 || dy <= 0 || T <= 0) return RXO_ERR_BADARG;
    if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads) reduction(+ : rc_rules, rc_prods, rc_margs)
#endif
    {
    /* one workspace and one set of chain-major staging buffers per thread, reused for every chain the thread runs */
    bp_ws ws;
    int rc_ws = bp_ws_reserve(&ws, d, dy, T, ptt, fe != NULL);
    double* yc = (double*)malloc(sizeof(double) * (size_t)T * dy);
    double* pm = (double*)malloc(sizeof(double) * (size_t)T * d);
    double* pc = (double*)malloc(sizeof(double) * (size_t)T * d * d);
    if (!yc || !pm || !pc) rc_ws = RXO_ERR_BADARG;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (int ch = 0; ch < n_chains; ++ch) {
        rxo_counters cc;
        memset(&cc, 0, sizeof cc);
        double f = 0.0;
        int rc = rc_ws;
        if (!rc) {
            for (int t = 0; t < T; ++t)
                memcpy(yc + (size_t)t * dy, y + ((size_t)t * n_chains + ch) * dy, sizeof(double) * dy);
            rc = lgssm_bp_ws(d, dy, T, A, B, P, Q, m0, V0, ptt, yc, pm, pc, fe ? &f : NULL, &cc, &ws);
        }
        if (rc) {
#ifdef _OPENMP
#pragma omp critical
#endif
            rc_all = rc;
            continue;
        }
        for (int t = 0; t < T; ++t) {
            memcpy(post_mean + ((size_t)t * n_chains + ch) * d, pm + (size_t)t * d, sizeof(double) * d);
            memcpy(post_cov + ((size_t)t * n_chains + ch) * d * d, pc + (size_t)t * d * d, sizeof(double) * d * d);
        }
        if (fe) fe[ch] = f;
        rc_rules += cc.rule_calls;
        rc_prods += cc.products;
        rc_margs += cc.marginals;
    }
    free(yc);
    free(pm);
    free(pc);
    if (!rc_ws) bp_ws_free(&ws);
    }
    if (counters) {
        counters->rule_calls = rc_rules;
        counters->products = rc_prods;
        counters->marginals = rc_margs;
    }
    (void)nthreads;
    return rc_all;
}

/* ------------------------------------------------------------------------------------------
 * Streaming / filtering driver (see rxoracle.h): one-step graph per observation, reference rule order.
 * The per-observation Bethe free energy of the one-step TREE equals −log p(y_t | y_<t)
 * (docs/src/manuals/variational/bethe-free-energy.md:70; the identity is checked for the full sweep in
 * tests/test_oracle.py) and is evaluated in that innovation form.
 * ------------------------------------------------------------------------------------------ */
int rxo_lgssm_filter(int d, int dy, int T, const double* A, const double* B, const double* P, const double* Q,
                     const double* m0, const double* V0, int prior_through_transition, const double* y,
                     double* hist_mean, double* hist_cov, double* fe, rxo_counters* counters) {
    if (d <= 0 || dy <= 0 || T <= 0) return RXO_ERR_BADARG;
    const int n = d > dy ? d : dy;
    const size_t nn = (size_t)n * n;
    double* buf = (double*)malloc(sizeof(double) * (14 * nn + 8 * (size_t)n));
    if (!buf) return RXO_ERR_BADARG;
    double *work = buf, *chw = work + nn, *Vp = chw + 2 * nn, *Lp = Vp + nn, *Lq = Lp + nn, *Lb = Lq + nn, *Lf = Lb + nn,
           *Vf = Lf + nn, *Sm = Vf + nn, *Si = Sm + nn, *tmpm = Si + nn, *V = tmpm + nn, *vecs = V + 2 * nn;
    double *m = vecs, *mp = m + n, *xp = mp + n, *xq = xp + n, *xb = xq + n, *xf = xb + n, *r = xf + n, *sr = r + n;
    ctx c;
    memset(&c, 0, sizeof c);
    c.work = work;
    c.count = 1;
    int rc = RXO_OK;
    double fsum = 0.0;
    memcpy(m, m0, sizeof(double) * d);
    memcpy(V, V0, sizeof(double) * d * d);
    /* MvN_y(:μ) with the clamped observation gives N(y, Q); its precision is constant */
    if ((rc = cholinv(dy, Q, Lq, NULL, chw))) goto done;
    for (int t = 0; t < T; ++t) {
        const double* yt = y + (size_t)t * dy;
        c.c.rule_calls++; /* prior node MvN(:out) with the data mean / covariance of @autoupdates */
        if (t > 0 || prior_through_transition) {
            rule_mul_out(d, d, A, m, V, mp, tmpm, &c);            /* `*`_A(:out)  */
            rule_mvn_additive(d, mp, tmpm, P, mp, Vp, &c);        /* MvN_x(:out)  */
        } else {
            memcpy(mp, m, sizeof(double) * d);
            memcpy(Vp, V, sizeof(double) * d * d);
        }
        c.c.rule_calls++;                                         /* MvN_y(:μ)    */
        matvec(dy, dy, Lq, yt, xq);
        rule_mul_in(dy, d, B, xq, Lq, xb, Lb, &c);                /* `*`_B(:in)   */
        if ((rc = to_other_param(d, mp, Vp, xp, Lp, NULL, chw))) goto done; /* weightedmean_precision(forward msg) */
        prod_wmp(d, xp, Lp, xb, Lb, xf, Lf, &c);                  /* q(x_t)       */
        if ((rc = to_other_param(d, xf, Lf, m, Vf, NULL, chw))) goto done;  /* mean_cov(q(x_t)) */
        memcpy(V, Vf, sizeof(double) * d * d);
        c.c.marginals++;
        memcpy(hist_mean + (size_t)t * d, m, sizeof(double) * d);
        memcpy(hist_cov + (size_t)t * d * d, V, sizeof(double) * d * d);
        if (fe) {
            double ld;
            congruence(dy, d, B, Vp, Sm, work);
            for (int i = 0; i < dy * dy; ++i) Sm[i] += Q[i];
            if ((rc = cholinv(dy, Sm, Si, &ld, chw))) goto done;
            matvec(dy, d, B, mp, r);
            for (int i = 0; i < dy; ++i) r[i] = yt[i] - r[i];
            matvec(dy, dy, Si, r, sr);
            double q = 0.0;
            for (int i = 0; i < dy; ++i) q += r[i] * sr[i];
            fsum += 0.5 * (dy * LOG2PI + ld + q);
        }
    }
    if (fe) {
        *fe = fsum / (double)T;
        if (!isfinite(*fe)) rc = RXO_ERR_NONFINITE_FE;
    }
done:
    if (counters) *counters = c.c;
    free(buf);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * Noise-free drift chain (test/models/statespace/ulgssm_tests.jl:8-15), reference schedule, scalars:
 *     x_prior ~ Normal(μ = m0, v = v0);  x[t] ~ x[t-1] + c;  y[t] ~ Normal(μ = x[t], v = obs_var)
 * Rules restated (ReactiveMP, univariate Normal family; messages in mean–variance form, products in weighted-mean–
 * precision form):
 *   NormalMeanVariance(:out)(m_μ::PointMass, q_v::PointMass) = N(m0, v0)             (prior node)
 *   NormalMeanVariance(:μ)(m_out::PointMass(y), q_v::PointMass) = N(y, v)            (observation node)
 *   typeof(+)(:out)(m_in1, m_in2::PointMass(c)) = N(mean + c, var);  typeof(+)(:in1)(m_out, m_in2) = N(mean − c, var)
 * Bethe free energy: prior node U − H[q(x_prior)]; every `+` node −H[q(in1)] (deterministic node; its constant input
 * is a counted −∞, reactivemp_force_marginal_computation_plugin.jl:52-98); observation nodes U over q(x[t]) − H[q(x[t])];
 * variables (degree − 1)·H; point entropies cancelled as reactivemp_free_energy.jl:108-123.
 * ------------------------------------------------------------------------------------------ */
int rxo_drift_chain_bp(long long T, const double* y, double m0, double v0, double c, double obs_var,
                       int prior_through_transition, double* post_mean, double* post_var, double* free_energy,
                       rxo_counters* counters) {
    if (T <= 0 || !y || !post_mean || !post_var) return RXO_ERR_BADARG;
    if (!(v0 > 0.0) || !(obs_var > 0.0)) return RXO_ERR_NOT_POSDEF;
    const int ptt = prior_through_transition ? 1 : 0;
    const long long n = T + ptt; /* states x_prior (if ptt), x[1..T] */
    rxo_counters cn = {0, 0, 0};
    /* inbound messages at every state, (ξ, w) form; have_* flags */
    double* fx = (double*)malloc(sizeof(double) * (size_t)n * 8);
    if (!fx) return RXO_ERR_BADARG;
    double *fw = fx + n, *bx = fw + n, *bw = bx + n, *ox = bw + n, *ow = ox + n, *qm = ow + n, *qv = qm + n;
    /* forward */
    double m = m0, v = v0;
    cn.rule_calls++; /* prior node (:out) */
    for (long long k = 0; k < n; ++k) {
        if (k > 0) { /* `+`(:out)(message from x[k-1] toward the node = prod of its other inbound messages) */
            double xi = fx[k - 1], w = fw[k - 1];
            if (k - 1 >= ptt) { xi += ox[k - 1]; w += ow[k - 1]; cn.products++; }
            m = xi / w + c;
            v = 1.0 / w;
            cn.rule_calls++;
        }
        fx[k] = m / v;
        fw[k] = 1.0 / v;
        if (k >= ptt) { /* observation node (:μ): N(y, obs_var) */
            ox[k] = y[k - ptt] / obs_var;
            ow[k] = 1.0 / obs_var;
            cn.rule_calls++;
        } else
            ox[k] = ow[k] = 0.0;
    }
    /* backward: toward x[k-1] through `+`(:in1) */
    bx[n - 1] = bw[n - 1] = 0.0;
    for (long long k = n - 1; k >= 1; --k) {
        double xi = ox[k], w = ow[k];
        if (k < n - 1) { xi += bx[k]; w += bw[k]; cn.products++; }
        const double mm = xi / w - c, vv = 1.0 / w;
        bx[k - 1] = mm / vv;
        bw[k - 1] = 1.0 / vv;
        cn.rule_calls++;
    }
    /* marginals */
    for (long long k = 0; k < n; ++k) {
        double xi = fx[k], w = fw[k];
        int parts = 1;
        if (k >= ptt) { xi += ox[k]; w += ow[k]; ++parts; }
        if (k < n - 1) { xi += bx[k]; w += bw[k]; ++parts; }
        cn.products += (uint64_t)(parts - 1);
        cn.marginals++;
        qm[k] = xi / w;
        qv[k] = 1.0 / w;
        if (k >= ptt) { post_mean[k - ptt] = qm[k]; post_var[k - ptt] = qv[k]; }
    }
    if (counters) *counters = cn;
    int rc = RXO_OK;
    if (free_energy) {
        creal nodes = {0.0, 0}, vars = {0.0, 0};
        long point_entropies = 0;
        for (long long k = 0; k < n; ++k) {
            const double H = 0.5 * (LOG2PI + 1.0 + log(qv[k]));
            if (k == 0) { /* prior node: clusters (out), (μ const), (v const) */
                nodes.v += 0.5 * (LOG2PI + log(v0) + ((qm[0] - m0) * (qm[0] - m0) + qv[0]) / v0) - H;
                nodes.ninf += 2;
                point_entropies += 2;
            } else { /* `+` node k: −H[q(in1)] and the counted point mass of c */
                nodes.v += -0.5 * (LOG2PI + 1.0 + log(qv[k - 1]));
                nodes.ninf += 1;
                point_entropies += 1;
            }
            if (k >= ptt) { /* observation node: out = PointMass(y), v const */
                const double e = y[k - ptt] - qm[k];
                nodes.v += 0.5 * (LOG2PI + log(obs_var) + (e * e + qv[k]) / obs_var) - H;
                nodes.ninf += 2;
                point_entropies += 2;
            }
            const int deg = 1 + (k >= ptt ? 1 : 0) + (k < n - 1 ? 1 : 0);
            vars.v += (deg - 1) * H;
        }
        const long ninf = nodes.ninf + vars.ninf - point_entropies;
        double fe = nodes.v + vars.v;
        if (ninf != 0) fe = ninf > 0 ? INFINITY : -INFINITY;
        *free_energy = fe;
        if (!isfinite(fe)) rc = RXO_ERR_NONFINITE_FE;
    }
    free(fx);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * Predictions (`predictvars = (y = KeepLast(),)`, src/model/plugins/reactivemp_inference.jl:619-624: the stream of the
 * message toward a data variable).  For y[t] that is MvN_y(:out)(m_μ, q_Σ) = N(B m, B V B' + Q) with (m, V) the
 * message x[t] -> `*`_B, i.e. the product of the OTHER messages into x[t] — the forward message and the backward message,
 * NOT its own observation.  The last H time indices have no observation (`missing`): their posteriors are the forward
 * predictions and nothing flows back from them.  Reference rule order: forward `*`_A(:out), MvN_x(:out); observation
 * MvN_y(:μ), `*`_B(:in); backward MvN_x(:μ), `*`_A(:in); products left to right.
 * y: [T][dy] observed part.  pred_mean [T+H][dy], pred_cov [T+H][dy][dy]; post_mean / post_cov (nullable): posteriors of
 * x for the H unobserved steps ([H][d], [H][d][d]).  prior_through_transition as rxo_lgssm_bp.
 * ------------------------------------------------------------------------------------------ */
int rxo_lgssm_predict(int d, int dy, int T, int H, const double* A, const double* B, const double* P, const double* Q,
                      const double* m0, const double* V0, int ptt, const double* y, double* pred_mean, double* pred_cov,
                      double* post_mean, double* post_cov) {
    if (d <= 0 || dy <= 0 || T <= 0 || H < 0 || !pred_mean || !pred_cov) return RXO_ERR_BADARG;
    const int dm = d > dy ? d : dy;
    const size_t vs = (size_t)d, ms = (size_t)d * d;
    int rc = RXO_OK;
    ctx c;
    memset(&c, 0, sizeof c);
    c.work = (double*)malloc(sizeof(double) * (size_t)(16 * dm * dm + 64));
    double* st = (double*)malloc(sizeof(double) * ((size_t)T * 2 * (vs + ms) + (size_t)(24 * dm * dm + 64)));
    if (!c.work || !st) { free(c.work); free(st); return RXO_ERR_BADARG; }
    double* fw_x = st;                 /* forward message into x[t] as WMP: ξ [T][d] */
    double* fw_L = fw_x + T * vs;      /* Λ [T][d][d] */
    double* bw_x = fw_L + T * ms;      /* backward message into x[t] (WMP); zero at t = T−1 */
    double* bw_L = bw_x + T * vs;
    double* f_m = bw_L + T * ms;       /* scratch */
    double* f_V = f_m + dm;
    double* t_m = f_V + dm * dm;
    double* t_V = t_m + dm;
    double* t_x = t_V + dm * dm;
    double* t_L = t_x + dm;
    double* o_x = t_L + dm * dm;
    double* o_L = o_x + dm;
    double* ny_x = o_L + dm * dm;
    double* ny_L = ny_x + dm;
    double* a_m = ny_L + dm * dm;
    double* a_V = a_m + dm;
    double* chw = a_V + dm * dm;       /* 8 dm² */
    double* Qi = chw + 8 * dm * dm;
    rc = cholinv(dy, Q, Qi, NULL, chw);
    /* ---- forward: the message MvN_x(:out) -> x[t], then x[t] -> next `*`_A = prod(fwd, obs) ---- */
    for (int t = 0; t < T && !rc; ++t) {
        if (t == 0) {
            if (ptt) {  /* x0 ~ prior; x[1] ~ MvN(A x0, P) */
                rule_mul_out(d, d, A, m0, V0, a_m, a_V, &c);
                rule_mvn_additive(d, a_m, a_V, P, f_m, f_V, &c);
            } else
                rule_mvn_additive(d, m0, NULL, V0, f_m, f_V, &c);
        } else {
            /* x[t-1] -> `*`_A: prod(fwd(t-1), obs(t-1)) in WMP, then mean_cov */
            rc = to_other_param(d, t_x, t_L, t_m, t_V, NULL, chw);
            if (rc) break;
            rule_mul_out(d, d, A, t_m, t_V, a_m, a_V, &c);
            rule_mvn_additive(d, a_m, a_V, P, f_m, f_V, &c);
        }
        rc = to_other_param(d, f_m, f_V, fw_x + t * vs, fw_L + t * ms, NULL, chw);
        if (rc) break;
        /* observation branch: MvN_y(:μ)(y, Q) = N(y, Q); `*`_B(:in) = WMP(B'Q⁻¹y, B'Q⁻¹B) */
        matvec(dy, dy, Qi, y + (size_t)t * dy, ny_x);
        rule_mul_in(dy, d, B, ny_x, Qi, o_x, o_L, &c);
        prod_wmp(d, fw_x + t * vs, fw_L + t * ms, o_x, o_L, t_x, t_L, &c);
    }
    /* (t_x, t_L) = filtered belief of the last observed step: the forecast start */
    double* fc_x = (double*)malloc(sizeof(double) * (vs + ms));
    double* fc_L = fc_x + vs;
    if (!fc_x) rc = RXO_ERR_BADARG;
    if (!rc) { memcpy(fc_x, t_x, sizeof(double) * vs); memcpy(fc_L, t_L, sizeof(double) * ms); }
    /* ---- backward messages into x[t]: `*`_A(:in)(MvN_x(:μ)(prod(obs(t+1), bwd(t+1)))) ---- */
    if (!rc) {
        memset(bw_x + (size_t)(T - 1) * vs, 0, sizeof(double) * vs);
        memset(bw_L + (size_t)(T - 1) * ms, 0, sizeof(double) * ms);
    }
    for (int t = T - 2; t >= 0 && !rc; --t) {
        matvec(dy, dy, Qi, y + (size_t)(t + 1) * dy, ny_x);
        rule_mul_in(dy, d, B, ny_x, Qi, o_x, o_L, &c);
        if (t + 1 < T - 1) prod_wmp(d, o_x, o_L, bw_x + (size_t)(t + 1) * vs, bw_L + (size_t)(t + 1) * ms, t_x, t_L, &c);
        else { memcpy(t_x, o_x, sizeof(double) * vs); memcpy(t_L, o_L, sizeof(double) * ms); }
        rc = to_other_param(d, t_x, t_L, t_m, t_V, NULL, chw);  /* toward MvN_x(:out side) as mean / covariance */
        if (rc) break;
        rule_mvn_additive(d, t_m, t_V, P, a_m, a_V, &c);          /* MvN_x(:μ) */
        rc = to_other_param(d, a_m, a_V, t_x, t_L, NULL, chw);
        if (rc) break;
        rule_mul_in(d, d, A, t_x, t_L, bw_x + (size_t)t * vs, bw_L + (size_t)t * ms, &c);
    }
    /* ---- predictions of the observed y[t]: x[t] -> `*`_B = prod(fwd, bwd) ---- */
    for (int t = 0; t < T && !rc; ++t) {
        if (t < T - 1) prod_wmp(d, fw_x + t * vs, fw_L + t * ms, bw_x + t * vs, bw_L + t * ms, t_x, t_L, &c);
        else { memcpy(t_x, fw_x + t * vs, sizeof(double) * vs); memcpy(t_L, fw_L + t * ms, sizeof(double) * ms); }
        rc = to_other_param(d, t_x, t_L, t_m, t_V, NULL, chw);
        if (rc) break;
        rule_mul_out(dy, d, B, t_m, t_V, a_m, a_V, &c);
        rule_mvn_additive(dy, a_m, a_V, Q, pred_mean + (size_t)t * dy, pred_cov + (size_t)t * dy * dy, &c);
    }
    /* ---- the unobserved tail: forward messages only ---- */
    if (!rc && H > 0) {
        rc = to_other_param(d, fc_x, fc_L, t_m, t_V, NULL, chw);
        for (int h = 0; h < H && !rc; ++h) {
            rule_mul_out(d, d, A, t_m, t_V, a_m, a_V, &c);
            rule_mvn_additive(d, a_m, a_V, P, f_m, f_V, &c);   /* q(x[T+h]) = the forward message (nothing else arrives) */
            if (post_mean) memcpy(post_mean + (size_t)h * vs, f_m, sizeof(double) * vs);
            if (post_cov) memcpy(post_cov + (size_t)h * ms, f_V, sizeof(double) * ms);
            rule_mul_out(dy, d, B, f_m, f_V, a_m, a_V, &c);
            rule_mvn_additive(dy, a_m, a_V, Q, pred_mean + (size_t)(T + h) * dy, pred_cov + (size_t)(T + h) * dy * dy, &c);
            memcpy(t_m, f_m, sizeof(double) * vs);
            memcpy(t_V, f_V, sizeof(double) * ms);
        }
    }
    free(fc_x);
    free(st);
    free(c.work);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * Independent textbook implementation, used ONLY to validate the restatement above
 * (identity: BP on a tree == Kalman filter + RTS smoother; Bethe FE == −log p(y)).
 * ------------------------------------------------------------------------------------------ */
/* offsets (known inputs): x[t] ~ N(A x[t-1] + cx[t], P), y[t] ~ N(B x[t] + cy[t], Q); NULL = none.  cx[0] enters only
   through the prior's transition (ptt). */
static __thread const double* g_cx = NULL; /* [T][d]  */
static __thread const double* g_cy = NULL; /* [T][dy] */
static int kalman_rts_impl(int d, int dy, int T, const double* A0, const double* B0, const double* P0,
                           const double* Q0, const double* m00, const double* V00, const int* sm, int ptt, const double* y,
                           double* post_mean, double* post_cov, double* neg_loglik) {
    const double *cx = g_cx, *cy = g_cy;
    /* time-varying constants: time index t uses model sm[t] (transition INTO x[t], observation of y[t]) */
#define MDL(t) (sm ? (size_t)sm[t] : (size_t)0)
    const double *A = A0 + MDL(0) * d * d, *B = B0 + MDL(0) * dy * d, *P = P0 + MDL(0) * d * d, *Q = Q0 + MDL(0) * dy * dy;
    const double *m0 = m00 + MDL(0) * d, *V0 = V00 + MDL(0) * d * d;
    const int dm = d > dy ? d : dy;
    const size_t vs = d, ms = (size_t)d * d;
    double* mf = (double*)malloc(sizeof(double) * T * (vs + ms));
    double* Vf = mf + T * vs;
    double* w = (double*)malloc(sizeof(double) * (size_t)(20 * dm * dm + 8 * dm));
    double *mp = w, *Vp = mp + dm, *S = Vp + dm * dm, *Si = S + dm * dm, *K = Si + dm * dm, *tmp = K + dm * dm,
           *e = tmp + dm * dm, *BV = e + dm, *chw = BV + dm * dm, *G = chw + 2 * dm * dm, *D = G + dm * dm,
           *t2 = D + dm * dm;
    int rc = RXO_OK;
    double nll = 0.0;
    double *pm0 = (double*)malloc(sizeof(double) * (vs + ms)), *pV0 = pm0 + vs;
    if (ptt) {
        matvec(d, d, A, m0, pm0);
        if (cx) for (size_t i = 0; i < vs; ++i) pm0[i] += cx[i];
        congruence(d, d, A, V0, pV0, tmp);
        for (size_t i = 0; i < ms; ++i) pV0[i] += P[i];
    } else {
        memcpy(pm0, m0, sizeof(double) * vs);
        memcpy(pV0, V0, sizeof(double) * ms);
    }
    for (int t = 0; t < T && !rc; ++t) {
        A = A0 + MDL(t) * d * d; B = B0 + MDL(t) * dy * d; P = P0 + MDL(t) * d * d; Q = Q0 + MDL(t) * dy * dy;
        if (t == 0) {
            memcpy(mp, pm0, sizeof(double) * vs);
            memcpy(Vp, pV0, sizeof(double) * ms);
        } else {
            matvec(d, d, A, mf + (t - 1) * vs, mp);
            if (cx) for (size_t i = 0; i < vs; ++i) mp[i] += cx[(size_t)t * vs + i];
            congruence(d, d, A, Vf + (t - 1) * ms, Vp, tmp);
            for (size_t i = 0; i < ms; ++i) Vp[i] += P[i];
        }
        {   /* a `missing` observation (any NaN entry; docs/src/manuals/inference/static.md:98-123): the observation
               branch sends no message, the filtered belief is the prediction and the step has no evidence term */
            int miss = 0;
            for (int i = 0; i < dy; ++i) miss |= (y[(size_t)t * dy + i] != y[(size_t)t * dy + i]);
            if (miss) {
                memcpy(mf + t * vs, mp, sizeof(double) * vs);
                memcpy(Vf + t * ms, Vp, sizeof(double) * ms);
                continue;
            }
        }
        congruence(dy, d, B, Vp, S, tmp);
        for (int i = 0; i < dy * dy; ++i) S[i] += Q[i];
        double ldS;
        rc = cholinv(dy, S, Si, &ldS, chw);
        if (rc) break;
        matvec(dy, d, B, mp, e);
        for (int i = 0; i < dy; ++i) e[i] = y[(size_t)t * dy + i] - e[i] - (cy ? cy[(size_t)t * dy + i] : 0.0);
        double q = 0.0;
        for (int i = 0; i < dy; ++i)
            for (int j = 0; j < dy; ++j) q += e[i] * Si[i * dy + j] * e[j];
        nll += 0.5 * (dy * LOG2PI + ldS + q);
        /* BV = B Vp (dy×d); K = BV' Si (d×dy) */
        for (int i = 0; i < dy; ++i)
            for (int j = 0; j < d; ++j) {
                double s = 0.0;
                for (int k = 0; k < d; ++k) s += B[i * d + k] * Vp[k * d + j];
                BV[i * d + j] = s;
            }
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < dy; ++j) {
                double s = 0.0;
                for (int k = 0; k < dy; ++k) s += BV[k * d + i] * Si[k * dy + j];
                K[i * dy + j] = s;
            }
        for (int i = 0; i < d; ++i) {
            double s = mp[i];
            for (int j = 0; j < dy; ++j) s += K[i * dy + j] * e[j];
            mf[t * vs + i] = s;
        }
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) {
                double s = Vp[i * d + j];
                for (int k = 0; k < dy; ++k) s -= K[i * dy + k] * BV[k * d + j];
                Vf[t * ms + i * d + j] = s;
            }
        for (int i = 0; i < d; ++i) /* symmetrise */
            for (int j = 0; j < i; ++j) {
                double s = 0.5 * (Vf[t * ms + i * d + j] + Vf[t * ms + j * d + i]);
                Vf[t * ms + i * d + j] = Vf[t * ms + j * d + i] = s;
            }
    }
    if (!rc) {
        memcpy(post_mean + (size_t)(T - 1) * vs, mf + (T - 1) * vs, sizeof(double) * vs);
        memcpy(post_cov + (size_t)(T - 1) * ms, Vf + (T - 1) * ms, sizeof(double) * ms);
        for (int t = T - 2; t >= 0 && !rc; --t) {
            A = A0 + MDL(t + 1) * d * d; P = P0 + MDL(t + 1) * d * d; /* the transition into x[t+1] */
            matvec(d, d, A, mf + t * vs, mp);
            if (cx) for (size_t i = 0; i < vs; ++i) mp[i] += cx[(size_t)(t + 1) * vs + i];
            congruence(d, d, A, Vf + t * ms, Vp, tmp);
            for (size_t i = 0; i < ms; ++i) Vp[i] += P[i];
            rc = cholinv(d, Vp, Si, NULL, chw);
            if (rc) break;
            /* G = Vf A' Vp^-1 */
            for (int i = 0; i < d; ++i)
                for (int j = 0; j < d; ++j) {
                    double s = 0.0;
                    for (int k = 0; k < d; ++k) s += Vf[t * ms + i * d + k] * A[j * d + k];
                    tmp[i * d + j] = s;
                }
            for (int i = 0; i < d; ++i)
                for (int j = 0; j < d; ++j) {
                    double s = 0.0;
                    for (int k = 0; k < d; ++k) s += tmp[i * d + k] * Si[k * d + j];
                    G[i * d + j] = s;
                }
            for (int i = 0; i < d; ++i) e[i] = post_mean[(size_t)(t + 1) * vs + i] - mp[i];
            for (int i = 0; i < d; ++i) {
                double s = mf[t * vs + i];
                for (int j = 0; j < d; ++j) s += G[i * d + j] * e[j];
                post_mean[(size_t)t * vs + i] = s;
            }
            for (size_t i = 0; i < ms; ++i) D[i] = post_cov[(size_t)(t + 1) * ms + i] - Vp[i];
            congruence(d, d, G, D, t2, tmp);
            for (size_t i = 0; i < ms; ++i) post_cov[(size_t)t * ms + i] = Vf[t * ms + i] + t2[i];
        }
    }
    if (neg_loglik) *neg_loglik = nll;
    free(mf);
    free(w);
    free(pm0);
    return rc;
}
#undef MDL
int rxo_lgssm_kalman_rts(int d, int dy, int T, const double* A, const double* B, const double* P,
                         const double* Q, const double* m0, const double* V0, int ptt, const double* y,
                         double* post_mean, double* post_cov, double* neg_loglik) {
    return kalman_rts_impl(d, dy, T, A, B, P, Q, m0, V0, NULL, ptt, y, post_mean, post_cov, neg_loglik);
}
/* Known inputs / offsets: x[t] ~ N(A x[t-1] + cx[t], P), y[t] ~ N(B x[t] + cy[t], Q) (either array may be NULL), optionally with
   per-step constants (step_model NULL: one model).  Test infrastructure: checks rxhip_lgssm_desc.state_offset / obs_offset. */
int rxo_lgssm_kalman_rts_affine(int d, int dy, int T, int n_models, const double* A, const double* B, const double* P,
                                const double* Q, const double* m0, const double* V0, const int* step_model, int ptt,
                                const double* cx, const double* cy, const double* y, double* post_mean, double* post_cov,
                                double* neg_loglik) {
    if (n_models <= 0) return RXO_ERR_BADARG;
    if (step_model)
        for (int t = 0; t < T; ++t)
            if (step_model[t] < 0 || step_model[t] >= n_models) return RXO_ERR_BADARG;
    g_cx = cx;
    g_cy = cy;
    const int rc = kalman_rts_impl(d, dy, T, A, B, P, Q, m0, V0, step_model, ptt, y, post_mean, post_cov, neg_loglik);
    g_cx = g_cy = NULL;
    return rc;
}
/* Time-varying constants (`A[t] * x[t-1]`, `Σ = P[t]` … in the @model loop): A, B, P, Q, m0, V0 hold n_models models,
   step_model[t] names the model of time index t.  Test infrastructure: checks rxhip_lgssm_desc.step_model. */
int rxo_lgssm_kalman_rts_tv(int d, int dy, int T, int n_models, const double* A, const double* B, const double* P,
                            const double* Q, const double* m0, const double* V0, const int* step_model, int ptt,
                            const double* y, double* post_mean, double* post_cov, double* neg_loglik) {
    if (!step_model || n_models <= 0) return RXO_ERR_BADARG;
    for (int t = 0; t < T; ++t)
        if (step_model[t] < 0 || step_model[t] >= n_models) return RXO_ERR_BADARG;
    return kalman_rts_impl(d, dy, T, A, B, P, Q, m0, V0, step_model, ptt, y, post_mean, post_cov, neg_loglik);
}


/* ==========================================================================================
 * Mean-field VMP for the univariate Gaussian mixture (NormalMixture node) — rules restated from
 * SURVEY.md Appendix A.4 / A.5; see rxoracle.h for the model and the assumed schedule.
 * ========================================================================================== */
static double digamma_(double x) {
    double r = 0.0;
    while (x < 6.0) {
        r -= 1.0 / x;
        x += 1.0;
    }
    double f = 1.0 / (x * x);
    /* asymptotic series */
    return r + log(x) - 0.5 / x -
           f * (1.0 / 12 - f * (1.0 / 120 - f * (1.0 / 252 - f * (1.0 / 240 - f * (1.0 / 132 - f * (691.0 / 32760 - f / 12))))));
}

/* split-phase form (what several GPUs do: every rank accumulates its shard, the 3K+1 statistics are summed over
 * ranks, every rank applies the same update).  state = (mean m, var m, shape p, rate p, alpha s), 5K doubles. */
int rxo_gmm_accumulate(long long N, int K, const double* y, const double* state, double* stats, double* resp,
                       rxo_counters* counters) {
    if (N < 0 || K <= 0 || K > 64) return RXO_ERR_BADARG;
    const double *mm = state, *mv = state + K, *pa = state + 2 * K, *pb = state + 3 * K, *al = state + 4 * K;
    double Ep[64], Elp[64], Els[64], lg[64], pi[64];
    double *S0 = stats, *S1 = stats + K, *S2 = stats + 2 * K;
    uint64_t rules = 0, prods = 0, margs = 0;
    double asum = 0.0, Hz = 0.0;
    for (int k = 0; k < K; ++k) asum += al[k];
    for (int k = 0; k < K; ++k) {
        Ep[k] = pa[k] / pb[k];                     /* E[p] of GammaShapeRate */
        Elp[k] = digamma_(pa[k]) - log(pb[k]);     /* E[log p] */
        Els[k] = digamma_(al[k]) - digamma_(asum); /* E[log s_k] of Dirichlet */
        S0[k] = S1[k] = S2[k] = 0.0;
    }
    /* q(z_i) from the marginals of the previous iteration; responsibility-weighted statistics */
    for (long long i = 0; i < N; ++i) {
        /* @rule NormalMixture(:switch): ∝ exp(−U_k), U_k = NormalMeanPrecision average energy;
           @rule Categorical(:out)(q_p::Dirichlet): ∝ exp(E log s_k); q(z_i) = normalised product */
        double mx = -INFINITY;
        for (int k = 0; k < K; ++k) {
            double d = y[i] - mm[k];
            double U = 0.5 * (LOG2PI - Elp[k] + Ep[k] * (mv[k] + d * d));
            lg[k] = Els[k] - U;
            if (lg[k] > mx) mx = lg[k];
        }
        double Z = 0.0;
        for (int k = 0; k < K; ++k) {
            pi[k] = exp(lg[k] - mx);
            Z += pi[k];
        }
        rules += 2; prods += 1; margs += 1;
        for (int k = 0; k < K; ++k) {
            pi[k] /= Z;
            if (pi[k] > 0.0) Hz -= pi[k] * log(pi[k]);
            if (resp) resp[i * K + k] = pi[k];
            /* the messages toward m[k] (N(y_i, precision π_ik E[p_k])), s (Dirichlet(1 + π)) and p[k] are functions of
               these sums: Σπ, Σπy, Σπy² */
            S0[k] += pi[k];
            S1[k] += pi[k] * y[i];
            S2[k] += pi[k] * y[i] * y[i];
            rules += 3; prods += 3;
        }
    }
    stats[3 * K] = Hz;
    if (counters) { counters->rule_calls += rules; counters->products += prods; counters->marginals += margs; }
    return RXO_OK;
}

int rxo_gmm_update(int K, const double* mu0, const double* v0, const double* a0, const double* b0, const double* alpha0,
                   const double* stats, double* state, double* fe, rxo_counters* counters) {
    if (K <= 0 || K > 64) return RXO_ERR_BADARG;
    double *mm = state, *mv = state + K, *pa = state + 2 * K, *pb = state + 3 * K, *al = state + 4 * K;
    const double *S0 = stats, *S1 = stats + K, *S2 = stats + 2 * K;
    const double Hz = stats[3 * K];
    int rc = RXO_OK;
    for (int k = 0; k < K; ++k) {
        /* product of the prior message and the N messages NormalMixture(m[k]) in (ξ, Λ) form, E[p] of the previous q(p) */
        const double Ep = pa[k] / pb[k];
        const double xi = mu0[k] / v0[k] + Ep * S1[k], lam = 1.0 / v0[k] + Ep * S0[k];
        mv[k] = 1.0 / lam;
        mm[k] = xi * mv[k];
        /* Dirichlet×Dirichlet = α1 + α2 − 1 over the N messages Dirichlet(1 + π) */
        al[k] = alpha0[k] + S0[k];
    }
    /* @rule NormalMixture(p[k]): GammaShapeRate(1 + π/2, π ½[(y−m̄)² + v]) with the NEW q(m[k]);
       Gamma×Gamma = (a1+a2−1, b1+b2).  Σ_i π_ik[(y_i−m̄)² + v] = S2 − 2m̄S1 + m̄²S0 + vS0 */
    for (int k = 0; k < K; ++k) {
        pa[k] = a0[k] + 0.5 * S0[k];
        pb[k] = b0[k] + 0.5 * (S2[k] - 2.0 * mm[k] * S1[k] + mm[k] * mm[k] * S0[k] + mv[k] * S0[k]);
        if (counters) counters->marginals += 3;
        if (!(mv[k] > 0.0) || !(pb[k] > 0.0)) rc = RXO_ERR_NOT_POSDEF;
    }
    if (fe) {
        /* Bethe free energy at the marginals of this iteration:
           Σ_i U_mix,i + Σ_i U_cat,i − Σ_i H[z_i] + Σ_k (U_m − H[m_k]) + Σ_k (U_p − H[p_k]) + (U_s − H[s]) */
        double F = -Hz, as2 = 0.0, a0s = 0.0;
        for (int k = 0; k < K; ++k) { as2 += al[k]; a0s += alpha0[k]; }
        double lB = -lgamma(as2), lB0 = -lgamma(a0s), Hs_t = 0.0, Us_t = 0.0;
        for (int k = 0; k < K; ++k) {
            double Epk = pa[k] / pb[k], Elpk = digamma_(pa[k]) - log(pb[k]), Elsk = digamma_(al[k]) - digamma_(as2);
            /* NormalMixture average energy Σ_i π_ik U_k(i) through the responsibility-weighted statistics */
            F += 0.5 * ((LOG2PI - Elpk) * S0[k] + Epk * (mv[k] * S0[k] + S2[k] - 2.0 * mm[k] * S1[k] + mm[k] * mm[k] * S0[k]));
            F += -S0[k] * Elsk; /* Categorical average energy */
            /* prior nodes minus entropies */
            double dm = mm[k] - mu0[k];
            F += 0.5 * (LOG2PI + log(v0[k]) + (dm * dm + mv[k]) / v0[k]) - 0.5 * (LOG2PI + 1.0 + log(mv[k]));
            F += (-a0[k] * log(b0[k]) + lgamma(a0[k]) - (a0[k] - 1.0) * Elpk + b0[k] * Epk) -
                 (pa[k] - log(pb[k]) + lgamma(pa[k]) + (1.0 - pa[k]) * digamma_(pa[k]));
            lB += lgamma(al[k]);
            lB0 += lgamma(alpha0[k]);
            Us_t += (alpha0[k] - 1.0) * Elsk;
            Hs_t += (al[k] - 1.0) * digamma_(al[k]);
        }
        double Us = lB0 - Us_t;
        double Hs = lB + (as2 - K) * digamma_(as2) - Hs_t;
        if (K > 1) F += Us - Hs; /* K = 1: no switch variable in the iid model (Dirichlet(1) terms vanish identically) */
        *fe = F;
        if (!isfinite(F)) rc = RXO_ERR_NONFINITE_FE;
    }
    return rc;
}

int rxo_gmm_vmp(long long N, int K, const double* y, const double* mu0, const double* v0, const double* a0,
                const double* b0, const double* alpha0, const double* init_m_mean, const double* init_m_var,
                const double* init_p_shape, const double* init_p_rate, const double* init_s_alpha, int iterations,
                double* hist, double* fe, double* resp, rxo_counters* counters) {
    if (N <= 0 || K <= 0 || K > 64 || iterations <= 0) return RXO_ERR_BADARG;
    double state[5 * 64], stats[3 * 64 + 1];
    rxo_counters c = {0, 0, 0};
    for (int k = 0; k < K; ++k) {
        state[k] = init_m_mean[k];
        state[K + k] = init_m_var[k];
        state[2 * K + k] = init_p_shape[k];
        state[3 * K + k] = init_p_rate[k];
        state[4 * K + k] = init_s_alpha[k];
    }
    int rc = RXO_OK;
    for (int it = 0; it < iterations; ++it) {
        int r = rxo_gmm_accumulate(N, K, y, state, stats, (resp && it == iterations - 1) ? resp : NULL, &c);
        if (!r) r = rxo_gmm_update(K, mu0, v0, a0, b0, alpha0, stats, state, fe ? fe + it : NULL, &c);
        if (r) rc = r;
        if (r == RXO_ERR_BADARG) break;
        if (hist) memcpy(hist + (size_t)it * 5 * K, state, sizeof(double) * 5 * K);
    }
    if (counters) *counters = c;
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * Multivariate Gaussian mixture, mean-field VMP (test/models/mixtures/gmm_multivariate_tests.jl:6-32):
 *     m[k] ~ MvNormal(mean = mu0[k], cov = S0[k]);  w[k] ~ Wishart(nu0[k], V0[k]);  s ~ Dirichlet(alpha0);
 *     z[i] ~ Categorical(s);  y[i] ~ NormalMixture(switch = z[i], m = m, p = w)      (p: precision matrices)
 * Rules as in the univariate case with NormalMeanPrecision -> MvNormalMeanPrecision and Gamma -> Wishart
 * (ExponentialFamily: E[W] = νV, E log|W| = ψ_d(ν/2) + d log 2 + log|V|):
 *   NormalMixture(:switch)  ∝ exp(−U_k),  U_k = ½[d log 2π − E log|W_k| + tr(E[W_k] E[(y−m_k)(y−m_k)'])]
 *   NormalMixture(m[k])     MvNormalMeanPrecision(y_i, π_ik E[W_k]); product with the prior in (ξ, Λ)
 *   NormalMixture(w[k])     natural parameters of the Wishart product: ν += π_ik, V⁻¹ += π_ik E[(y_i−m_k)(y_i−m_k)']
 * Same schedule as rxo_gmm_vmp: q(z) from the previous marginals, then q(s), q(m) with the previous E[W], then q(w)
 * with the new q(m); the sum over points of the w-message is evaluated through Σπ, Σπy, Σπyy' (exact algebra).
 * For d = 1 the model IS the univariate one (Wishart(ν, V) = Gamma(shape ν/2, rate 1/(2V))): tests/test_oracle.py
 * checks that the two restatements agree to rounding, free energy included.
 * state / init / hist layout per component: mean[d] | cov[d][d] | nu | V[d][d] | alpha   (SZ = 2 + d + 2d² doubles)
 * ------------------------------------------------------------------------------------------ */
static double mvdigamma_(double a, int d) {
    double s = 0.0;
    for (int i = 0; i < d; ++i) s += digamma_(a - 0.5 * i);
    return s;
}
static double mvlgamma_(double a, int d) {
    double s = 0.25 * d * (d - 1) * 1.1447298858494001741434273513531; /* log π */
    for (int i = 0; i < d; ++i) s += lgamma(a - 0.5 * i);
    return s;
}
int rxo_mvgmm_vmp(long long N, int K, int d, const double* y, const double* mu0, const double* S0, const double* nu0,
                  const double* V0, const double* alpha0, const double* init, int iterations, double* hist, double* fe,
                  double* resp) {
    if (N <= 0 || K <= 0 || K > 64 || d <= 0 || d > 8 || iterations <= 0) return RXO_ERR_BADARG;
    const int dd = d * d, SZ = 2 + d + 2 * dd;
    const double LOG2 = 0.69314718055994530942;
    double* st = (double*)malloc(sizeof(double) * (size_t)K * SZ);
    double* w = (double*)calloc((size_t)(K * (6 * dd + 2 * d + 8) + 8 * dd + 4 * d + 64), sizeof(double));
    if (!st || !w) { free(st); free(w); return RXO_ERR_BADARG; }
    memcpy(st, init, sizeof(double) * (size_t)K * SZ);
    double *EW = w, *S0i = EW + K * dd, *V0i = S0i + K * dd, *Lam = V0i + K * dd, *S2 = Lam + K * dd, *Sc = S2 + K * dd,
           *xi = Sc + K * dd, *S1 = xi + K * d, *Elw = S1 + K * d, *Els = Elw + K, *lg = Els + K, *pi = lg + K, *S0k = pi + K,
           *tmp = S0k + K, *chw = tmp + 2 * dd, *dv = chw + 4 * dd;
    int rc = RXO_OK;
    double ldS0[64], ldV0[64];
    for (int k = 0; k < K; ++k) {
        if ((rc = cholinv(d, S0 + k * dd, S0i + k * dd, &ldS0[k], chw))) goto done;
        if ((rc = cholinv(d, V0 + k * dd, V0i + k * dd, &ldV0[k], chw))) goto done;
    }
    for (int it = 0; it < iterations; ++it) {
        double asum = 0.0, Hz = 0.0;
        for (int k = 0; k < K; ++k) asum += st[k * SZ + SZ - 1];
        for (int k = 0; k < K; ++k) {
            const double* sk = st + k * SZ;
            const double nu = sk[d + dd];
            const double* V = sk + d + dd + 1;
            double ldV;
            if ((rc = cholinv(d, V, tmp, &ldV, chw))) goto done; /* log|V| */
            for (int i = 0; i < dd; ++i) EW[k * dd + i] = nu * V[i];
            Elw[k] = mvdigamma_(0.5 * nu, d) + d * LOG2 + ldV;
            Els[k] = digamma_(sk[SZ - 1]) - digamma_(asum);
            for (int i = 0; i < dd; ++i) { Lam[k * dd + i] = S0i[k * dd + i]; S2[k * dd + i] = 0.0; }
            matvec(d, d, S0i + k * dd, mu0 + k * d, xi + k * d);
            for (int a = 0; a < d; ++a) S1[k * d + a] = 0.0;
            S0k[k] = 0.0;
        }
        for (long long i = 0; i < N; ++i) {
            const double* yi = y + i * d;
            double mx = -INFINITY;
            for (int k = 0; k < K; ++k) {
                const double* sk = st + k * SZ;
                double q = 0.0;
                for (int a = 0; a < d; ++a) dv[a] = yi[a] - sk[a];
                for (int a = 0; a < d; ++a)
                    for (int b = 0; b < d; ++b) q += EW[k * dd + a * d + b] * (dv[a] * dv[b] + sk[d + a * d + b]);
                lg[k] = Els[k] - 0.5 * (d * LOG2PI - Elw[k] + q);
                if (lg[k] > mx) mx = lg[k];
            }
            double Z = 0.0;
            for (int k = 0; k < K; ++k) { pi[k] = exp(lg[k] - mx); Z += pi[k]; }
            for (int k = 0; k < K; ++k) {
                pi[k] /= Z;
                if (pi[k] > 0.0) Hz -= pi[k] * log(pi[k]);
                if (resp && it == iterations - 1) resp[i * K + k] = pi[k];
                for (int a = 0; a < d; ++a) {
                    double s1 = 0.0;
                    for (int b = 0; b < d; ++b) {
                        Lam[k * dd + a * d + b] += pi[k] * EW[k * dd + a * d + b]; /* message toward m[k]: precision */
                        s1 += EW[k * dd + a * d + b] * yi[b];
                        S2[k * dd + a * d + b] += pi[k] * yi[a] * yi[b];
                    }
                    xi[k * d + a] += pi[k] * s1;                                   /* … and weighted mean */
                    S1[k * d + a] += pi[k] * yi[a];
                }
                S0k[k] += pi[k];
            }
        }
        double F = -Hz, as2 = 0.0, a0s = 0.0;
        for (int k = 0; k < K; ++k) {
            double* sk = st + k * SZ;
            double ldC;
            if ((rc = cholinv(d, Lam + k * dd, sk + d, NULL, chw))) goto done; /* cov of q(m[k]) */
            matvec(d, d, sk + d, xi + k * d, sk);
            if ((rc = cholinv(d, sk + d, tmp, &ldC, chw))) goto done;          /* log|cov| for the entropy */
            sk[SZ - 1] = alpha0[k] + S0k[k];
            /* Sc = Σ_i π_ik E[(y_i − m)(y_i − m)'] with the new q(m[k]) */
            for (int a = 0; a < d; ++a)
                for (int b = 0; b < d; ++b)
                    Sc[k * dd + a * d + b] = S2[k * dd + a * d + b] - sk[a] * S1[k * d + b] - S1[k * d + a] * sk[b] +
                                             S0k[k] * (sk[a] * sk[b] + sk[d + a * d + b]);
            for (int i = 0; i < dd; ++i) tmp[i] = V0i[k * dd + i] + Sc[k * dd + i];
            double ldVn;
            if ((rc = cholinv(d, tmp, sk + d + dd + 1, &ldVn, chw))) goto done; /* new V = (V0⁻¹ + Sc)⁻¹, ldVn = log|V⁻¹| */
            sk[d + dd] = nu0[k] + S0k[k];
            as2 += sk[SZ - 1];
            a0s += alpha0[k];
        }
        if (hist) memcpy(hist + (size_t)it * K * SZ, st, sizeof(double) * (size_t)K * SZ);
        if (fe) {
            double lB = -lgamma(as2), lB0 = -lgamma(a0s), Hs_t = 0.0, Us_t = 0.0;
            for (int k = 0; k < K; ++k) {
                const double* sk = st + k * SZ;
                const double nu = sk[d + dd], *V = sk + d + dd + 1, al = sk[SZ - 1];
                double ldV, ldC;
                if ((rc = cholinv(d, V, tmp, &ldV, chw))) goto done;
                if ((rc = cholinv(d, sk + d, tmp, &ldC, chw))) goto done;
                const double Elwk = mvdigamma_(0.5 * nu, d) + d * LOG2 + ldV, Elsk = digamma_(al) - digamma_(as2);
                double trWS = 0.0, trV0W = 0.0, trS0 = 0.0;
                for (int a = 0; a < d; ++a)
                    for (int b = 0; b < d; ++b) {
                        trWS += nu * V[a * d + b] * Sc[k * dd + b * d + a];
                        trV0W += V0i[k * dd + a * d + b] * nu * V[b * d + a];
                        trS0 += S0i[k * dd + a * d + b] * (sk[d + b * d + a] + (sk[b] - mu0[k * d + b]) * (sk[a] - mu0[k * d + a]));
                    }
                /* NormalMixture and Categorical average energies */
                F += 0.5 * (S0k[k] * (d * LOG2PI - Elwk) + trWS) - S0k[k] * Elsk;
                /* MvNormal prior node of m[k] minus H[q(m[k])] */
                F += 0.5 * (d * LOG2PI + ldS0[k] + trS0) - 0.5 * (d * (LOG2PI + 1.0) + ldC);
                /* Wishart prior node of w[k] minus H[q(w[k])] */
                const double n0 = nu0[k];
                F += -(0.5 * (n0 - d - 1.0) * Elwk - 0.5 * trV0W - 0.5 * n0 * d * LOG2 - 0.5 * n0 * ldV0[k] - mvlgamma_(0.5 * n0, d));
                F -= 0.5 * (d + 1.0) * ldV + 0.5 * d * (d + 1.0) * LOG2 + mvlgamma_(0.5 * nu, d) -
                     0.5 * (nu - d - 1.0) * mvdigamma_(0.5 * nu, d) + 0.5 * nu * d;
                lB += lgamma(al);
                lB0 += lgamma(alpha0[k]);
                Us_t += (alpha0[k] - 1.0) * Elsk;
                Hs_t += (al - 1.0) * digamma_(al);
            }
            if (K > 1) F += (lB0 - Us_t) - (lB + (as2 - K) * digamma_(as2) - Hs_t);
            fe[it] = F;
            if (!isfinite(F)) { rc = RXO_ERR_NONFINITE_FE; goto done; }
        }
    }
done:
    free(st);
    free(w);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * A state-space chain whose observation-noise precision is unknown — the first COMPOSED graph: the chain of
 * test/models/statespace/mlgssm_test.jl:9-14 with the observation nodes in the precision parametrisation and a Wishart prior on it
 * (the node pair of test/models/iid/mv_iid_precision_tests.jl:11-15):
 *     x[1] ~ MvNormal(μ = m0, Σ = V0);  x[t] ~ MvNormal(μ = A x[t-1], Σ = P);  W ~ Wishart(ν0, S0);  y[t] ~ MvNormal(μ = B x[t], Λ = W)
 * with q(x[1..T], W) = q(x[1..T]) q(W) (`@constraints`; the factorisation handling of reactivemp_inference.jl:499-501).  Mean-field VMP:
 *   q(x)   the MvNormalMeanPrecision(:μ) messages toward B x[t] carry E[W] = νV (ExponentialFamily mean of a Wishart), everything else on
 *          the chain is sum-product: the smoother of the Gaussian chain with Q = E[W]⁻¹ (rxo_lgssm_kalman_rts, pinned to rxo_lgssm_bp);
 *   q(W)   MvNormalMeanPrecision(:Λ) sends Wishart(dy + 2, E[(y−Bx)(y−Bx)′]⁻¹) per observation; the product with the prior in natural
 *          parameters: ν = ν0 + T, V⁻¹ = S0⁻¹ + Σ_t [(y_t − B m_t)(y_t − B m_t)′ + B V_t B′];
 *   order  per iteration q(x) with the previous q(W), then q(W) with the new q(x) (the order rxo_mvgmm_vmp assumes for q(m), q(w));
 *   F      Bethe free energy of the iteration's marginals.  With Ŵ_old = E_old[W] the chain's own terms are the Gaussian model's evidence
 *          −log p̃(y | Q = Ŵ_old⁻¹) minus the observation nodes' energies in that model; adding the observation nodes' average energies under
 *          the new q(W) and the Wishart prior node's energy minus H[q(W)] gives
 *              F = −log p̃(y) + T/2 (log|Ŵ_old| − E_new log|W|) + ½ tr((Ŵ_new − Ŵ_old) Σ_t E[r_t r_t′]) + KL(q_new(W) ‖ p(W)).
 * In the limit A = I, P → 0, B = I the chain is the iid model with unknown mean: tests/test_oracle.py checks this function against
 * rxo_mvgmm_vmp (K = 1) there, free energy included.  w_hist: [iterations][1 + dy·dy] = ν | V; fe: [iterations] (nullable).
 * Test infrastructure: the checker of rxhip_lgssm_noise_create.
 * ------------------------------------------------------------------------------------------ */
int rxo_lgssm_noise_vmp(int d, int dy, int T, const double* A, const double* B, const double* P, const double* m0, const double* V0,
                        int ptt, const double* y, double nu0, const double* S0, double init_nu, const double* init_V, int iterations,
                        double* post_mean, double* post_cov, double* w_hist, double* fe) {
    if (d <= 0 || dy <= 0 || T <= 0 || iterations <= 0 || !(nu0 > dy - 1.0) || !(init_nu > dy - 1.0)) return RXO_ERR_BADARG;
    const double LOG2 = 0.69314718055994530942;
    const size_t qq = (size_t)dy * dy;
    double* w = (double*)calloc(8 * qq + 2 * (size_t)dy * d + 4 * (size_t)(dy > d ? dy : d) * (dy > d ? dy : d) + 2 * (size_t)dy, sizeof(double));
    if (!w) return RXO_ERR_BADARG;
    double *V = w, *What = V + qq, *Q = What + qq, *S = Q + qq, *S0i = S + qq, *Vn = S0i + qq, *Wn = Vn + qq, *tmp = Wn + qq,
           *BV = tmp + qq, *r = BV + (size_t)dy * d, *chw = r + 2 * dy;
    int rc = RXO_OK;
    double ldS0, nu = init_nu;
    memcpy(V, init_V, sizeof(double) * qq);
    if ((rc = cholinv(dy, S0, S0i, &ldS0, chw))) goto done;   /* ldS0 = log|S0| */
    for (int it = 0; it < iterations; ++it) {
        double ldW, nll = 0.0;
        for (size_t i = 0; i < qq; ++i) What[i] = nu * V[i];
        if ((rc = cholinv(dy, What, Q, &ldW, chw))) goto done;   /* Q = E[W]⁻¹, ldW = log|Ŵ_old| */
        if ((rc = rxo_lgssm_kalman_rts(d, dy, T, A, B, P, Q, m0, V0, ptt, y, post_mean, post_cov, &nll))) goto done;
        memset(S, 0, sizeof(double) * qq);
        for (int t = 0; t < T; ++t) {
            const double *m = post_mean + (size_t)t * d, *C = post_cov + (size_t)t * d * d;
            for (int a = 0; a < dy; ++a) {
                double sm = y[(size_t)t * dy + a];
                for (int k = 0; k < d; ++k) sm -= B[a * d + k] * m[k];
                r[a] = sm;
                for (int k = 0; k < d; ++k) {
                    double sv = 0.0;
                    for (int l = 0; l < d; ++l) sv += B[a * d + l] * C[l * d + k];
                    BV[a * d + k] = sv;
                }
            }
            for (int a = 0; a < dy; ++a)
                for (int b = 0; b < dy; ++b) {
                    double sv = r[a] * r[b];
                    for (int k = 0; k < d; ++k) sv += BV[a * d + k] * B[b * d + k];
                    S[a * dy + b] += sv;
                }
        }
        for (size_t i = 0; i < qq; ++i) tmp[i] = S0i[i] + 0.5 * (S[i] + S[(i % dy) * dy + i / dy]);
        double ldVi;
        if ((rc = cholinv(dy, tmp, Vn, &ldVi, chw))) goto done;   /* V_new, ldVi = log|V_new⁻¹| */
        const double nun = nu0 + (double)T, ldV = -ldVi;
        for (size_t i = 0; i < qq; ++i) Wn[i] = nun * Vn[i];
        if (w_hist) {
            w_hist[(size_t)it * (1 + qq)] = nun;
            memcpy(w_hist + (size_t)it * (1 + qq) + 1, Vn, sizeof(double) * qq);
        }
        if (fe) {
            const double Elw = mvdigamma_(0.5 * nun, dy) + dy * LOG2 + ldV;
            double trdS = 0.0, trS0W = 0.0;
            for (int a = 0; a < dy; ++a)
                for (int b = 0; b < dy; ++b) {
                    trdS += (Wn[a * dy + b] - What[a * dy + b]) * S[b * dy + a];
                    trS0W += S0i[a * dy + b] * Wn[b * dy + a];
                }
            double F = nll + 0.5 * (double)T * (ldW - Elw) + 0.5 * trdS;
            /* Wishart prior node of W minus H[q(W)] (the lines of rxo_mvgmm_vmp) */
            F += -(0.5 * (nu0 - dy - 1.0) * Elw - 0.5 * trS0W - 0.5 * nu0 * dy * LOG2 - 0.5 * nu0 * ldS0 - mvlgamma_(0.5 * nu0, dy));
            F -= 0.5 * (dy + 1.0) * ldV + 0.5 * dy * (dy + 1.0) * LOG2 + mvlgamma_(0.5 * nun, dy) - 0.5 * (nun - dy - 1.0) * mvdigamma_(0.5 * nun, dy) +
                 0.5 * nun * dy;
            fe[it] = F;
            if (!isfinite(F)) { rc = RXO_ERR_NONFINITE_FE; goto done; }
        }
        nu = nun;
        memcpy(V, Vn, sizeof(double) * qq);
    }
done:
    free(w);
    return rc;
}

/* ==========================================================================================
 * Hierarchical Gaussian filter (GCV node) — see rxoracle.h for the model, provenance and assumptions.
 * ========================================================================================== */
int rxo_gauss_hermite(int n, double* x, double* w) {
    /* Newton iteration on the orthonormal Hermite recurrence (Numerical Recipes gauher, fp64) */
    if (n < 1 || n > 64) return RXO_ERR_BADARG;
    const double PIM4 = 0.7511255444649425; /* π^{-1/4} */
    int m = (n + 1) / 2;
    double z = 0.0, pp = 0.0;
    for (int i = 0; i < m; ++i) {
        if (i == 0) z = sqrt((double)(2 * n + 1)) - 1.85575 * pow((double)(2 * n + 1), -0.16667);
        else if (i == 1) z -= 1.14 * pow((double)n, 0.426) / z;
        else if (i == 2) z = 1.86 * z - 0.86 * x[0];
        else if (i == 3) z = 1.91 * z - 0.91 * x[1];
        else z = 2.0 * z - x[i - 2];
        for (int its = 0; its < 100; ++its) {
            double p1 = PIM4, p2 = 0.0;
            for (int j = 0; j < n; ++j) {
                double p3 = p2;
                p2 = p1;
                p1 = z * sqrt(2.0 / (j + 1)) * p2 - sqrt((double)j / (j + 1)) * p3;
            }
            pp = sqrt(2.0 * n) * p2;
            double z1 = z;
            z = z1 - p1 / pp;
            if (fabs(z - z1) <= 1e-15 * (1.0 + fabs(z))) break;
        }
        x[i] = z;
        x[n - 1 - i] = -z;
        w[i] = 2.0 / (pp * pp);
        w[n - 1 - i] = w[i];
    }
    return RXO_OK;
}

int rxo_hgf_filter(long long T, const double* y, double kappa, double omega, double z_variance, double y_variance,
                   double z0m, double z0v, double x0m, double x0v, int vmp_iters, int n_gh, double* zm_o, double* zv_o,
                   double* xm_o, double* xv_o, double* fe, rxo_counters* counters) {
    if (T <= 0 || vmp_iters <= 0 || n_gh < 1 || n_gh > 64) return RXO_ERR_BADARG;
    double gx[64], gw[64];
    int rc = rxo_gauss_hermite(n_gh, gx, gw);
    if (rc) return rc;
    const double SQRTPI = 1.7724538509055160273;
    double qzm = z0m, qzv = z0v, qxm = x0m, qxv = x0v; /* current marginals q(zt), q(xt) */
    uint64_t rules = 0, prods = 0, margs = 0;
    if (fe) for (int n = 0; n < vmp_iters; ++n) fe[n] = 0.0;
    for (long long t = 0; t < T; ++t) {
        /* @autoupdates: priors of this step from the previous posteriors */
        const double zm = qzm, zv = qzv, xm = qxm, xv = qxv;
        const double fzv = zv + z_variance; /* NormalMeanVariance(:out)(m_μ, q_v): forward message to zt */
        rules += 3;                         /* two prior nodes + transition node; the obs message: */
        rules += 1;
        for (int n = 0; n < vmp_iters; ++n) {
            /* expected precision of the GCV node under q(zt):  A·B,  A = exp(−ω), B = exp(−κ z̄ + κ² v_z / 2) */
            const double A = exp(-omega);
            const double B = exp(-kappa * qzm + 0.5 * kappa * kappa * qzv);
            const double g = A * B;
            /* @marginalrule GCV(:y_x)(m_y = N(y_t, y_variance), m_x = N(xm, xv), q_z): joint precision / weighted mean */
            const double l11 = 1.0 / y_variance + g, l22 = 1.0 / xv + g, l12 = -g;
            const double det = l11 * l22 - l12 * l12;
            if (!(det > 0.0)) { rc = RXO_ERR_NOT_POSDEF; goto out; }
            const double v11 = l22 / det, v22 = l11 / det, v12 = -l12 / det;
            const double x1 = y[t] / y_variance, x2 = xm / xv;
            const double m1 = v11 * x1 + v12 * x2, m2 = v12 * x1 + v22 * x2;
            const double psi = (m1 - m2) * (m1 - m2) + v11 + v22 - 2.0 * v12;
            /* @rule GCV(:z)(q_y_x, q_κ, q_ω): ExponentialLinearQuadratic(a = κ, b = ψA, c = −κ, d = 0);
               product with the forward message N(zm, fzv) moment-matched by approximate_meancov (Gauss–Hermite) */
            const double a = kappa, b = psi * A, c = -kappa;
            double norm = 0.0, mean = 0.0, cs[64], pts[64];
            const double sc = sqrt(2.0 * fzv);
            for (int i = 0; i < n_gh; ++i) {
                const double pt = zm + sc * gx[i];
                const double gv = exp(-0.5 * (a * pt + b * exp(c * pt)));
                const double cv = gw[i] / SQRTPI * gv;
                pts[i] = pt;
                cs[i] = cv;
                mean += pt * cv;
                norm += cv;
            }
            mean /= norm;
            double var = 0.0;
            for (int i = 0; i < n_gh; ++i) var += cs[i] * (pts[i] - mean) * (pts[i] - mean);
            var /= norm;
            if (!(var > 0.0) || !isfinite(mean)) { rc = RXO_ERR_NONFINITE_FE; goto out; }
            qzm = mean; qzv = var; qxm = m1; qxv = v11;
            rules += 2; prods += 2; margs += 3;
            if (fe) {
                const double Bn = exp(-kappa * qzm + 0.5 * kappa * kappa * qzv);
                /* The transition node's joint q(zt, zt_min) (@marginalrule NormalMeanVariance(:out_μ)) and the message
                   toward zt_min see the GCV z-message through its Gaussian moments: mean_var(ExponentialLinearQuadratic)
                   = approximate_meancov of pdf(z)·exp(z²/2) against N(0, 1).  (With this treatment the reference's golden
                   value test/models/statespace/hgf_tests.jl:113 is reproduced to 1e-5; see tests/golden.) */
                double en = 0.0, em = 0.0, ecs[64], epts[64];
                for (int i = 0; i < n_gh; ++i) {
                    const double pt = 1.4142135623730951 * gx[i];
                    const double cv = gw[i] / SQRTPI * exp(-0.5 * (a * pt + b * exp(c * pt)) + 0.5 * pt * pt);
                    epts[i] = pt; ecs[i] = cv; em += pt * cv; en += cv;
                }
                em /= en;
                double ev = 0.0;
                for (int i = 0; i < n_gh; ++i) ev += ecs[i] * (epts[i] - em) * (epts[i] - em);
                ev /= en;
                if (!(ev > 0.0) || !isfinite(em)) { rc = RXO_ERR_NONFINITE_FE; goto out; }
                const double wb = 1.0 / z_variance, w00 = 1.0 / ev + wb, w11 = 1.0 / zv + wb;
                const double dW = w00 * w11 - wb * wb;
                const double s00 = w11 / dW, s11 = w00 / dW, s01 = wb / dW;
                const double j0 = s00 * (em / ev) + s01 * (zm / zv), j1 = s01 * (em / ev) + s11 * (zm / zv);
                const double mu_m = j1, var_m = s11;
                const double e2 = (j0 - j1) * (j0 - j1) + s00 + s11 - 2.0 * s01;
                double F = 0.0;
                F += 0.5 * (LOG2PI + log(zv) + ((mu_m - zm) * (mu_m - zm) + var_m) / zv);           /* prior zt_min */
                F += 0.5 * (LOG2PI + log(xv) + ((m2 - xm) * (m2 - xm) + v22) / xv);                  /* prior xt_min */
                F += 0.5 * (LOG2PI + log(z_variance) + e2 / z_variance);                             /* transition   */
                F -= 0.5 * (2.0 * (LOG2PI + 1.0) - log(dW));                                         /* −H[zt,zt_min] */
                F += 0.5 * (LOG2PI + (qzm * kappa + omega) + psi * A * Bn);                          /* GCV average energy */
                F -= 0.5 * (2.0 * (LOG2PI + 1.0) + log(v11 * v22 - v12 * v12));                      /* −H[xt,xt_min] */
                F += 0.5 * (LOG2PI + log(y_variance) + ((y[t] - m1) * (y[t] - m1) + v11) / y_variance); /* observation */
                fe[n] += F;
            }
        }
        zm_o[t] = qzm; zv_o[t] = qzv; xm_o[t] = qxm; xv_o[t] = qxv;
    }
    if (fe) for (int n = 0; n < vmp_iters; ++n) fe[n] /= (double)T;
out:
    if (counters) { counters->rule_calls = rules; counters->products = prods; counters->marginals = margs; }
    return rc;
}
