// vmp_engines.hip — the mean-field engines of the pattern-matched families behind the C ABI: the uni- and multivariate Gaussian-mixture engines
// (csrc/gmm_kernels.hpp, mvgmm_kernels.hpp; SURVEY §8 a9/a10: rxhip_gmm_* / rxhip_mvgmm_create) and the hierarchical Gaussian filter
// (csrc/hgf_kernels.hpp; a11: rxhip_hgf_create, its run).  The runtime they share with the state-space engines — streams, profiling events,
// the engine handle — is rxhip.hip's (engine.hpp).  No kernels of the state-space path live here.
#include "../../include/rxhip.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <vector>

#include "gmm_kernels.hpp"
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Warray-bounds"   // (k_hgf_filter<FE = false> indexes its FE-sized scratch inside `if (FE)` branches that are dead in that instance)
#include "hgf_kernels.hpp"
#pragma clang diagnostic pop
#include "mvgmm_kernels.hpp"
#include "engine.hpp"

using namespace rxhip;

namespace host {
static bool chol_inv(int n, const double* A, double* out, double* logdet) { return rxhip::host_chol_inv(n, A, out, logdet); }
}  // namespace host

// ------------------------------------------------------------------------------------------
// Gaussian-mixture VMP engine
static int gmm_kt(int K) { return K <= 1 ? 1 : K <= 2 ? 2 : K <= 4 ? 4 : K <= 8 ? 8 : 16; }
static MvgParams mvg_params(rxhip_engine* e) {
    MvgParams p;
    p.N = e->g.N; p.K = e->g.K; p.y = e->d_y; p.resp = e->g.d_resp; p.state = e->g.d_par; p.drv = e->g.d_drv;
    p.prior = e->g.d_prior; p.partial = e->g.d_partial; p.totals = e->g.d_totals; p.hist = e->g.d_hist; p.fe = e->g.d_fe;
    p.iteration = e->g.it; p.nblocks = e->g.nblocks; p.write_resp = 0; p.status = e->d_status;
    return p;
}
template <int D, int KT>
struct MvgLaunch {
    static void init(const MvgParams& p, hipStream_t s) { hipLaunchKernelGGL((k_mvg_init<D, KT>), dim3(1), dim3(64), 0, s, p); }
    static void pass(const MvgParams& p, bool resp, hipStream_t s) {
        if (resp) hipLaunchKernelGGL((k_mvg_pass<D, KT, true>), dim3(p.nblocks), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((k_mvg_pass<D, KT, false>), dim3(p.nblocks), dim3(256), 0, s, p);
    }
    static void reduce(const MvgParams& p, hipStream_t s) {
        hipLaunchKernelGGL(k_mvg_reduce, dim3(KT * MvgDim<D>::STAT + 1), dim3(256), 0, s, p, KT * MvgDim<D>::STAT + 1);
    }
    static void update(const MvgParams& p, bool fe, hipStream_t s) {
        if (fe) hipLaunchKernelGGL((k_mvg_update<D, KT, true>), dim3(1), dim3(64), 0, s, p);
        else hipLaunchKernelGGL((k_mvg_update<D, KT, false>), dim3(1), dim3(64), 0, s, p);
    }
};
// component tile: the statistics of a lane live in registers, KT·(1 + d + d(d+1)/2) ≤ 128 doubles
static int mvg_kt(int d, int K) {
    const int cap = d <= 2 ? 16 : 8;
    const int kt = K <= 4 ? 4 : K <= 8 ? 8 : 16;
    return kt <= cap ? kt : 0;
}
#define MVG_DISPATCH(d, kt, CALL)                                                    \
    switch ((d) * 100 + (kt)) {                                                      \
        case 104: MvgLaunch<1, 4>::CALL; break;  case 108: MvgLaunch<1, 8>::CALL; break;  case 116: MvgLaunch<1, 16>::CALL; break; \
        case 204: MvgLaunch<2, 4>::CALL; break;  case 208: MvgLaunch<2, 8>::CALL; break;  case 216: MvgLaunch<2, 16>::CALL; break; \
        case 304: MvgLaunch<3, 4>::CALL; break;  case 308: MvgLaunch<3, 8>::CALL; break;                                          \
        case 404: MvgLaunch<4, 4>::CALL; break;  default: MvgLaunch<4, 8>::CALL; break;                                           \
    }


static GmmParams gmm_params(rxhip_engine* e) {
    GmmParams p;
    p.N = e->g.N; p.K = e->g.K; p.y = e->d_y; p.resp = e->g.d_resp; p.par = e->g.d_par; p.drv = e->g.d_drv;
    p.prior = e->g.d_prior; p.partial = e->g.d_partial; p.totals = e->g.d_totals; p.hist = e->g.d_hist; p.fe = e->g.d_fe;
    p.iteration = e->g.it; p.nblocks = e->g.nblocks; p.write_resp = 0; p.status = e->d_status;
    return p;
}
template <int KT>
struct GmmLaunch {
    static void init(const GmmParams& p, hipStream_t s) { hipLaunchKernelGGL((k_gmm_init<KT>), dim3(1), dim3(256), 0, s, p); }
    static void pass(const GmmParams& p, bool resp, hipStream_t s) {
        if (resp) hipLaunchKernelGGL((k_gmm_pass<KT, true>), dim3(p.nblocks), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((k_gmm_pass<KT, false>), dim3(p.nblocks), dim3(256), 0, s, p);
    }
    static void reduce(const GmmParams& p, hipStream_t s) { hipLaunchKernelGGL((k_gmm_reduce<KT>), dim3(3 * KT + 1), dim3(256), 0, s, p); }
    static void update(const GmmParams& p, bool fe, hipStream_t s) {
        if (fe) hipLaunchKernelGGL((k_gmm_update<KT, true>), dim3(1), dim3(64), 0, s, p);
        else hipLaunchKernelGGL((k_gmm_update<KT, false>), dim3(1), dim3(64), 0, s, p);
    }
};
#define GMM_DISPATCH(kt, CALL)                   \
    switch (kt) {                                \
        case 1: GmmLaunch<1>::CALL; break;       \
        case 2: GmmLaunch<2>::CALL; break;       \
        case 4: GmmLaunch<4>::CALL; break;       \
        case 8: GmmLaunch<8>::CALL; break;       \
        default: GmmLaunch<16>::CALL; break;     \
    }


extern "C" {

rxhip_status rxhip_gmm_create(const rxhip_gmm_desc* ds, rxhip_engine** out) {
    if (!out) return RXHIP_ERR_BADARG;
    *out = nullptr;
    if (!ds || ds->N <= 0 || ds->K <= 0 || !ds->mu0 || !ds->v0 || !ds->a0 || !ds->b0 || !ds->alpha0 || !ds->init_m_mean ||
        !ds->init_m_var || !ds->init_p_shape || !ds->init_p_rate || !ds->init_s_alpha)
        return RXHIP_ERR_BADARG;
    if (ds->K > 16) return RXHIP_ERR_UNSUPPORTED;
    for (int k = 0; k < ds->K; ++k)
        if (!(ds->v0[k] > 0) || !(ds->a0[k] > 0) || !(ds->b0[k] > 0) || !(ds->alpha0[k] > 0) || !(ds->init_m_var[k] > 0) ||
            !(ds->init_p_shape[k] > 0) || !(ds->init_p_rate[k] > 0) || !(ds->init_s_alpha[k] > 0))
            return RXHIP_ERR_NOT_POSDEF;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return RXHIP_ERR_NO_DEVICE;
    rxhip_engine* e = new rxhip_engine();
    *out = e;
    e->kind = 1;
    e->g.N = ds->N;
    e->g.K = ds->K;
    e->g.KT = gmm_kt(ds->K);
    e->g.materialize = ds->materialize_responsibilities ? 1 : 0;
    e->n_chains = 1;
    e->T = ds->N;
    e->dy = 1;
    if (ds->device >= 0) {
        if (ds->device >= ndev) return fail(e, RXHIP_ERR_BADARG, "device %d out of range (%d visible)", ds->device, ndev);
        e->device = ds->device;
    } else
        HIPCHK(e, hipGetDevice(&e->device));
    SET_DEVICE(e);
    if (ds->stream) e->stream = (hipStream_t)ds->stream;
    else {
        HIPCHK(e, stream_acquire(e->device, &e->stream));
        e->own_stream = true;
    }
    const int KT = e->g.KT, K = e->g.K;
    e->g.nq = 3 * KT + 1;
    e->g.hist_stride = 5 * K;
    e->g.state_size = 5 * KT;
    long long nb = (e->g.N + 255) / 256;
    if (nb > 1024) nb = 1024;  // 4 workgroups per CU, grid-stride over the observations
    e->g.nblocks = (int)nb;
    std::vector<double> prior(5 * KT, 1.0), init(5 * KT, 1.0);
    const double* pr[5] = {ds->mu0, ds->v0, ds->a0, ds->b0, ds->alpha0};
    const double* in[5] = {ds->init_m_mean, ds->init_m_var, ds->init_p_shape, ds->init_p_rate, ds->init_s_alpha};
    for (int f = 0; f < 5; ++f)
        for (int k = 0; k < K; ++k) {
            prior[f * KT + k] = pr[f][k];
            init[f * KT + k] = in[f][k];
        }
    HIPCHK(e, hipMalloc(&e->g.d_prior, sizeof(double) * 5 * KT));
    HIPCHK(e, hipMalloc(&e->g.d_init, sizeof(double) * 5 * KT));
    HIPCHK(e, hipMalloc(&e->g.d_par, sizeof(double) * 5 * KT));
    HIPCHK(e, hipMalloc(&e->g.d_drv, sizeof(double) * 3 * KT));
    HIPCHK(e, hipMalloc(&e->g.d_partial, sizeof(double) * (size_t)nb * (3 * KT + 1)));
    HIPCHK(e, hipMalloc(&e->g.d_totals, sizeof(double) * (3 * KT + 1)));
    HIPCHK(e, hipMemcpy(e->g.d_prior, prior.data(), sizeof(double) * 5 * KT, hipMemcpyHostToDevice));
    HIPCHK(e, hipMemcpy(e->g.d_init, init.data(), sizeof(double) * 5 * KT, hipMemcpyHostToDevice));
    if (e->g.materialize) HIPCHK(e, hipMalloc(&e->g.d_resp, sizeof(double) * (size_t)e->g.N * K));
    HIPCHK(e, hipMalloc(&e->d_status, sizeof(int)));
    HIPCHK(e, hipMemset(e->d_status, 0, sizeof(int)));
    return RXHIP_OK;
}

rxhip_status rxhip_mvgmm_create(const rxhip_mvgmm_desc* ds, rxhip_engine** out) {
    if (!out) return RXHIP_ERR_BADARG;
    *out = nullptr;
    if (!ds || ds->N <= 0 || ds->K <= 0 || ds->d <= 0 || !ds->mu0 || !ds->S0 || !ds->nu0 || !ds->V0 || !ds->alpha0 ||
        !ds->init_m_mean || !ds->init_m_cov || !ds->init_w_nu || !ds->init_w_V || !ds->init_s_alpha)
        return RXHIP_ERR_BADARG;
    if (ds->d > 4 || mvg_kt(ds->d, ds->K) == 0) return RXHIP_ERR_UNSUPPORTED;
    const int d = ds->d, dd = d * d, K = ds->K, KT = mvg_kt(d, K);
    const int SZ = 2 + d + 2 * dd, PRI = d + 2 * dd + 4, STAT = 1 + d + d * (d + 1) / 2, DRV = 1 + d * (d + 1) / 2 + d;
    for (int k = 0; k < K; ++k)
        if (!(ds->nu0[k] > d - 1) || !(ds->init_w_nu[k] > d - 1) || !(ds->alpha0[k] > 0) || !(ds->init_s_alpha[k] > 0)) return RXHIP_ERR_NOT_POSDEF;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return RXHIP_ERR_NO_DEVICE;
    rxhip_engine* e = new rxhip_engine();
    *out = e;
    e->kind = 1;
    e->g.mvd = d;
    e->g.N = ds->N; e->g.K = K; e->g.KT = KT;
    e->g.materialize = ds->materialize_responsibilities ? 1 : 0;
    e->g.nq = KT * STAT + 1; e->g.hist_stride = K * SZ; e->g.state_size = K * SZ;
    e->n_chains = 1; e->T = ds->N; e->dy = d; e->d = d;
    if (ds->device >= 0) {
        if (ds->device >= ndev) return fail(e, RXHIP_ERR_BADARG, "device %d out of range (%d visible)", ds->device, ndev);
        e->device = ds->device;
    } else
        HIPCHK(e, hipGetDevice(&e->device));
    SET_DEVICE(e);
    if (ds->stream) e->stream = (hipStream_t)ds->stream;
    else {
        HIPCHK(e, stream_acquire(e->device, &e->stream));
        e->own_stream = true;
    }
    long long nb = (e->g.N + 255) / 256;
    if (nb > 1024) nb = 1024;
    e->g.nblocks = (int)nb;
    // prior block per component: mu0 | S0⁻¹ | nu0 | V0⁻¹ | alpha0 | log|S0| | log|V0|   (inverses / log-determinants once, here)
    std::vector<double> prior((size_t)K * PRI), init((size_t)K * SZ), tmp(dd);
    for (int k = 0; k < K; ++k) {
        double* pr = prior.data() + (size_t)k * PRI;
        double ldS = 0, ldV = 0;
        for (int a = 0; a < d; ++a) pr[a] = ds->mu0[k * d + a];
        if (!host::chol_inv(d, ds->S0 + (size_t)k * dd, pr + d, &ldS)) return fail(e, RXHIP_ERR_NOT_POSDEF, "prior covariance of m[%d] is not positive definite", k);
        pr[d + dd] = ds->nu0[k];
        if (!host::chol_inv(d, ds->V0 + (size_t)k * dd, pr + d + dd + 1, &ldV)) return fail(e, RXHIP_ERR_NOT_POSDEF, "Wishart scale of w[%d] is not positive definite", k);
        pr[d + 2 * dd + 1] = ds->alpha0[k];
        pr[d + 2 * dd + 2] = ldS;
        pr[d + 2 * dd + 3] = ldV;
        double* in = init.data() + (size_t)k * SZ;
        for (int a = 0; a < d; ++a) in[a] = ds->init_m_mean[k * d + a];
        for (int q = 0; q < dd; ++q) in[d + q] = ds->init_m_cov[(size_t)k * dd + q];
        in[d + dd] = ds->init_w_nu[k];
        for (int q = 0; q < dd; ++q) in[d + dd + 1 + q] = ds->init_w_V[(size_t)k * dd + q];
        in[SZ - 1] = ds->init_s_alpha[k];
        if (!host::chol_inv(d, in + d, tmp.data(), nullptr) || !host::chol_inv(d, in + d + dd + 1, tmp.data(), nullptr))
            return fail(e, RXHIP_ERR_NOT_POSDEF, "initial marginal of component %d is not positive definite", k);
    }
    HIPCHK(e, hipMalloc(&e->g.d_prior, sizeof(double) * prior.size()));
    HIPCHK(e, hipMalloc(&e->g.d_init, sizeof(double) * init.size()));
    HIPCHK(e, hipMalloc(&e->g.d_par, sizeof(double) * init.size()));
    HIPCHK(e, hipMalloc(&e->g.d_drv, sizeof(double) * (size_t)KT * DRV));
    HIPCHK(e, hipMalloc(&e->g.d_partial, sizeof(double) * (size_t)nb * e->g.nq));
    HIPCHK(e, hipMalloc(&e->g.d_totals, sizeof(double) * e->g.nq));
    HIPCHK(e, hipMemcpy(e->g.d_prior, prior.data(), sizeof(double) * prior.size(), hipMemcpyHostToDevice));
    HIPCHK(e, hipMemcpy(e->g.d_init, init.data(), sizeof(double) * init.size(), hipMemcpyHostToDevice));
    if (e->g.materialize) HIPCHK(e, hipMalloc(&e->g.d_resp, sizeof(double) * (size_t)e->g.N * K));
    HIPCHK(e, hipMalloc(&e->d_status, sizeof(int)));
    HIPCHK(e, hipMemset(e->d_status, 0, sizeof(int)));
    return RXHIP_OK;
}

rxhip_status rxhip_gmm_begin_run(rxhip_engine* e, int32_t iterations) {
    TREE_GUARD(e);
    if (!e || e->kind != 1) return RXHIP_ERR_BADARG;
    if (iterations <= 0) return fail(e, RXHIP_ERR_BADARG, "run: iterations must be positive");
    if (!e->have_data) return fail(e, RXHIP_ERR_STATE, "run: no observations (call rxhip_set_data first)");
    SET_DEVICE(e);
    if (iterations > e->g.hist_cap) {
        HIPCHK(e, hipStreamSynchronize(e->stream));
        if (e->g.d_hist) HIPCHK(e, hipFree(e->g.d_hist));
        if (e->g.d_fe) HIPCHK(e, hipFree(e->g.d_fe));
        e->g.d_hist = e->g.d_fe = nullptr;
        HIPCHK(e, hipMalloc(&e->g.d_hist, sizeof(double) * (size_t)iterations * e->g.hist_stride));
        HIPCHK(e, hipMalloc(&e->g.d_fe, sizeof(double) * iterations));
        e->g.hist_cap = iterations;
    }
    HIPCHK(e, hipMemsetAsync(e->g.d_fe, 0, sizeof(double) * iterations, e->stream));
    HIPCHK(e, hipMemcpyAsync(e->g.d_par, e->g.d_init, sizeof(double) * e->g.state_size, hipMemcpyDeviceToDevice, e->stream));
    e->g.it = 0;
    e->g.iterations = iterations;
    if (e->g.mvd) {
        MvgParams p = mvg_params(e);
        MVG_DISPATCH(e->g.mvd, e->g.KT, init(p, e->stream));
    } else {
        GmmParams p = gmm_params(e);
        GMM_DISPATCH(e->g.KT, init(p, e->stream));
    }
    HIPCHK(e, hipGetLastError());
    e->rule_calls = e->products = e->marginals = 0;
    return RXHIP_OK;
}
rxhip_status rxhip_gmm_accumulate(rxhip_engine* e) {
    TREE_GUARD(e);
    if (!e || e->kind != 1) return RXHIP_ERR_BADARG;
    if (e->g.it >= e->g.iterations) return fail(e, RXHIP_ERR_STATE, "accumulate: no iteration left (call rxhip_gmm_begin_run)");
    SET_DEVICE(e);
    const bool resp = e->g.materialize && e->g.it == e->g.iterations - 1;
    rxhip_status st;
    if ((st = prof_begin(e, RXHIP_K_GMM_PASS))) return st;
    if (e->g.mvd) {
        MvgParams p = mvg_params(e);
        MVG_DISPATCH(e->g.mvd, e->g.KT, pass(p, resp, e->stream));
    } else {
        GmmParams p = gmm_params(e);
        GMM_DISPATCH(e->g.KT, pass(p, resp, e->stream));
    }
    if ((st = prof_end(e))) return st;
    if ((st = prof_begin(e, RXHIP_K_GMM_REDUCE))) return st;
    if (e->g.mvd) {
        MvgParams p = mvg_params(e);
        MVG_DISPATCH(e->g.mvd, e->g.KT, reduce(p, e->stream));
    } else {
        GmmParams p = gmm_params(e);
        GMM_DISPATCH(e->g.KT, reduce(p, e->stream));
    }
    if ((st = prof_end(e))) return st;
    HIPCHK(e, hipGetLastError());
    return RXHIP_OK;
}
rxhip_status rxhip_gmm_statistics_device(rxhip_engine* e, double** stats_dev, int32_t* n) {
    TREE_GUARD(e);
    if (!e || e->kind != 1) return RXHIP_ERR_BADARG;
    if (stats_dev) *stats_dev = e->g.d_totals;
    if (n) *n = e->g.nq;
    return RXHIP_OK;
}
rxhip_status rxhip_gmm_update(rxhip_engine* e, int32_t want_fe) {
    TREE_GUARD(e);
    if (!e || e->kind != 1) return RXHIP_ERR_BADARG;
    if (e->g.it >= e->g.iterations) return fail(e, RXHIP_ERR_STATE, "update: no iteration left");
    SET_DEVICE(e);
    rxhip_status st;
    if ((st = prof_begin(e, RXHIP_K_GMM_UPDATE))) return st;
    if (e->g.mvd) {
        MvgParams p = mvg_params(e);
        MVG_DISPATCH(e->g.mvd, e->g.KT, update(p, want_fe != 0, e->stream));
    } else {
        GmmParams p = gmm_params(e);
        GMM_DISPATCH(e->g.KT, update(p, want_fe != 0, e->stream));
    }
    if ((st = prof_end(e))) return st;
    HIPCHK(e, hipGetLastError());
    e->g.it++;
    e->last_iterations = e->g.it;
    e->last_want_fe = want_fe != 0;
    e->ran = true;
    // reference-equivalent event counts per iteration (the oracle counts its own invocations the same way)
    const uint64_t N = (uint64_t)e->g.N, K = (uint64_t)e->g.K;
    e->rule_calls += N * (2 + 3 * K);
    e->products += N * (1 + 3 * K);
    e->marginals += N + 3 * K;
    return RXHIP_OK;
}
rxhip_status rxhip_gmm_get_history(rxhip_engine* e, double* hist) {
    TREE_GUARD(e);
    if (!e || e->kind != 1 || !hist) return RXHIP_ERR_BADARG;
    if (!e->ran) return fail(e, RXHIP_ERR_STATE, "get_history: no run yet");
    SET_DEVICE(e);
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipMemcpy(hist, e->g.d_hist, sizeof(double) * (size_t)e->g.it * e->g.hist_stride, hipMemcpyDeviceToHost));
    return RXHIP_OK;
}
rxhip_status rxhip_gmm_get_responsibilities(rxhip_engine* e, double* resp) {
    TREE_GUARD(e);
    if (!e || e->kind != 1 || !resp) return RXHIP_ERR_BADARG;
    if (!e->g.materialize) return fail(e, RXHIP_ERR_STATE, "responsibilities were not materialised (desc.materialize_responsibilities)");
    if (!e->ran || e->g.it < e->g.iterations) return fail(e, RXHIP_ERR_STATE, "get_responsibilities: run not finished");
    SET_DEVICE(e);
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipMemcpy(resp, e->g.d_resp, sizeof(double) * (size_t)e->g.N * e->g.K, hipMemcpyDeviceToHost));
    return RXHIP_OK;
}


// Gauss–Hermite nodes / weights (Newton iteration on the orthonormal recurrence)
static void gauss_hermite_host(int n, double* x, double* w) {
    const double PIM4 = 0.7511255444649425;
    const int m = (n + 1) / 2;
    double z = 0.0, pp = 0.0;
    for (int i = 0; i < m; ++i) {
        if (i == 0) z = std::sqrt((double)(2 * n + 1)) - 1.85575 * std::pow((double)(2 * n + 1), -0.16667);
        else if (i == 1) z -= 1.14 * std::pow((double)n, 0.426) / z;
        else if (i == 2) z = 1.86 * z - 0.86 * x[0];
        else if (i == 3) z = 1.91 * z - 0.91 * x[1];
        else z = 2.0 * z - x[i - 2];
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
        x[i] = z;
        x[n - 1 - i] = -z;
        w[i] = 2.0 / (pp * pp);
        w[n - 1 - i] = w[i];
    }
}

rxhip_status rxhip_hgf_create(const rxhip_hgf_desc* ds, rxhip_engine** out) {
    if (!out) return RXHIP_ERR_BADARG;
    *out = nullptr;
    if (!ds || ds->T <= 0 || ds->n_series <= 0 || ds->n_gh < 1) return RXHIP_ERR_BADARG;
    if (ds->n_gh > 32) return RXHIP_ERR_UNSUPPORTED;
    if (!(ds->z_variance > 0) || !(ds->y_variance > 0) || !(ds->z0_var > 0) || !(ds->x0_var > 0)) return RXHIP_ERR_NOT_POSDEF;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return RXHIP_ERR_NO_DEVICE;
    rxhip_engine* e = new rxhip_engine();
    *out = e;
    e->kind = 2;
    e->h.ds = *ds;
    e->T = ds->T;
    e->n_chains = ds->n_series;
    e->dy = 1;
    e->d = 1;
    if (ds->device >= 0) {
        if (ds->device >= ndev) return fail(e, RXHIP_ERR_BADARG, "device %d out of range (%d visible)", ds->device, ndev);
        e->device = ds->device;
    } else
        HIPCHK(e, hipGetDevice(&e->device));
    SET_DEVICE(e);
    if (ds->stream) e->stream = (hipStream_t)ds->stream;
    else {
        HIPCHK(e, stream_acquire(e->device, &e->stream));
        e->own_stream = true;
    }
    double gh[64] = {0}, gx[32], gw[32];
    gauss_hermite_host(ds->n_gh, gx, gw);
    for (int i = 0; i < ds->n_gh; ++i) {
        gh[i] = gx[i];
        gh[32 + i] = gw[i] / 1.7724538509055160273;
    }
    HIPCHK(e, hipMalloc(&e->h.d_gh, sizeof(gh)));
    HIPCHK(e, hipMemcpy(e->h.d_gh, gh, sizeof(gh), hipMemcpyHostToDevice));
    HIPCHK(e, hipMalloc(&e->h.d_out, sizeof(double) * 4 * (size_t)ds->T * ds->n_series));
    HIPCHK(e, hipMalloc(&e->d_fe_chain, sizeof(double) * (size_t)ds->n_series));
    HIPCHK(e, hipMalloc(&e->d_status, sizeof(int)));
    HIPCHK(e, hipMemset(e->d_status, 0, sizeof(int)));
    return RXHIP_OK;
}


}  // extern "C"

rxhip_status rxhip::hgf_run_async(rxhip_engine* e, int32_t iterations, int32_t want_fe) {
    if (iterations <= 0) return fail(e, RXHIP_ERR_BADARG, "run: iterations must be positive");
    if (!e->have_data) return fail(e, RXHIP_ERR_STATE, "run: no observations (call rxhip_set_data first)");
    SET_DEVICE(e);
    const size_t C = (size_t)e->n_chains, T = (size_t)e->T;
    if (iterations > e->h.fe_cap) {
        HIPCHK(e, hipStreamSynchronize(e->stream));
        if (e->h.d_fe_series) HIPCHK(e, hipFree(e->h.d_fe_series));
        if (e->h.d_fe_total) HIPCHK(e, hipFree(e->h.d_fe_total));
        e->h.d_fe_series = e->h.d_fe_total = nullptr;
        HIPCHK(e, hipMalloc(&e->h.d_fe_series, sizeof(double) * (size_t)iterations * C));
        HIPCHK(e, hipMalloc(&e->h.d_fe_total, sizeof(double) * iterations));
        e->h.fe_cap = iterations;
    }
    HIPCHK(e, hipMemsetAsync(e->h.d_fe_series, 0, sizeof(double) * (size_t)iterations * C, e->stream));
    HgfParams p;
    p.T = e->T; p.n_series = e->n_chains; p.y = e->d_y;
    p.zm = e->h.d_out; p.zv = e->h.d_out + T * C; p.xm = e->h.d_out + 2 * T * C; p.xv = e->h.d_out + 3 * T * C;
    p.fe_series = e->h.d_fe_series; p.gh = e->h.d_gh;
    const rxhip_hgf_desc& d = e->h.ds;
    p.kappa = d.kappa; p.omega = d.omega; p.z_variance = d.z_variance; p.y_variance = d.y_variance;
    p.z0m = d.z0_mean; p.z0v = d.z0_var; p.x0m = d.x0_mean; p.x0v = d.x0_var;
    p.iters = iterations; p.n_gh = d.n_gh; p.status = e->d_status;
    rxhip_status st;
    if ((st = prof_begin(e, RXHIP_K_HGF_FILTER))) return st;
    const unsigned nb = (unsigned)((C + HGF_SERIES_PER_WAVE - 1) / HGF_SERIES_PER_WAVE);
    if (want_fe) hipLaunchKernelGGL((k_hgf_filter<true>), dim3(nb), dim3(64), 0, e->stream, p);
    else hipLaunchKernelGGL((k_hgf_filter<false>), dim3(nb), dim3(64), 0, e->stream, p);
    if ((st = prof_end(e))) return st;
    if (want_fe) {
        hipLaunchKernelGGL(k_hgf_fe, dim3(iterations), dim3(256), 0, e->stream, p, e->h.d_fe_total);
        HIPCHK(e, hipMemcpyAsync(e->d_fe_chain, e->h.d_fe_series + (size_t)(iterations - 1) * C, sizeof(double) * C,
                                 hipMemcpyDeviceToDevice, e->stream));
    }
    HIPCHK(e, hipGetLastError());
    e->last_iterations = iterations;
    e->last_want_fe = want_fe != 0;
    e->ran = true;
    e->rule_calls = (uint64_t)C * T * (4 + 2 * (uint64_t)iterations);
    e->products = (uint64_t)C * T * 2 * (uint64_t)iterations;
    e->marginals = (uint64_t)C * T * 3 * (uint64_t)iterations;
    return RXHIP_OK;
}

