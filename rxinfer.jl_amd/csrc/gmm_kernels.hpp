// gmm_kernels.hpp — mean-field VMP for the univariate Gaussian mixture (BASELINE config 5) on gfx950.
//
// Reference rules replaced (bodies live in the un-vendored ReactiveMP.jl):
//   a10 NormalMixture(:switch | :m[k] | :p[k]) and its average energy   test/models/mixtures/gmm_univariate_tests.jl:16-19
//   a9  NormalMeanPrecision average energy, Normal / Gamma prior nodes, Gamma×Gamma and Gaussian products
//       (aliases src/model/graphppl.jl:340-370, 399-423), Categorical / Dirichlet switch prior
//       (test/models/mixtures/gmm_multivariate_tests.jl:26)
//   a7  Bethe free energy of the mean-field factorisation (reactivemp_free_energy.jl:51-126)
// K = 1 is the iid Gaussian with unknown mean and precision (test/models/models_tests.jl:114-128).
//
// The reference evaluates O(N·K) rule calls per iteration, one heap object each.  Here one VMP iteration is
//   k_gmm_pass   : stream the observations once (8 B/point), per point K logits -> softmax -> q(z_i);
//                  accumulate the responsibility-weighted statistics S0 = Σπ, S1 = Σπy, S2 = Σπy² and the
//                  entropy Σ H[q(z_i)] in registers; fixed-shape wave/block reduction -> per-block partials
//                  (optionally q(z) is materialised, 8K B/point)
//   k_gmm_reduce : block partials -> totals (fixed order: deterministic run to run).  With several GPUs the
//                  host all-reduces these 3K+1 doubles over RCCL — the path's one exchange step
//   k_gmm_update : products of the messages toward m[k], p[k], s in closed form from the statistics,
//                  Bethe free energy, and the per-component constants of the next pass
// Schedule (assumed; the reference's reactive order is undocumented): per iteration
// q(z_i) from the marginals of the previous iteration, then q(s), q(m[k]) from the new q(z), then q(p[k]) from
// the new q(z) and the new q(m[k]) (the order under which the reference test's own initialisation converges).
#pragma once
#include <hip/hip_runtime.h>

#include "lgssm_kernels.hpp"

namespace rxhip {

constexpr double kLog2Pi = 1.8378770664093454835606594728112;

struct GmmParams {
    long long N;
    int K;            // real number of components
    const double* y;  // [N]
    double* resp;     // [N][K] or nullptr
    // parameters, all [KT]-strided arrays of KT doubles (KT = padded K)
    double* par;      // [5][KT]: m mean, m var, p shape, p rate, s alpha   (current marginals)
    double* drv;      // [3][KT]: c_k, h_k, m_k  — logit_k(y) = c_k − h_k (y − m_k)²
    const double* prior;  // [5][KT]: mu0, v0, a0, b0, alpha0
    double* partial;  // [blocks][3KT+1]
    double* totals;   // [3KT+1]
    double* hist;     // [iterations][5][K]
    double* fe;       // [iterations]
    int iteration;
    int nblocks;
    int write_resp;
    int* status;
};

__device__ __forceinline__ double digamma_dev(double x) {
    double r = 0.0;
    while (x < 6.0) {
        r -= 1.0 / x;
        x += 1.0;
    }
    const double f = 1.0 / (x * x);
    return r + log(x) - 0.5 / x -
           f * (1.0 / 12 - f * (1.0 / 120 - f * (1.0 / 252 - f * (1.0 / 240 - f * (1.0 / 132 - f * (691.0 / 32760 - f / 12))))));
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;  // valid in lane 0; fixed tree shape
}

// exp without the library routine's range checks: k = rint(x log2 e), r = x − k ln 2 in two pieces, degree-12 Taylor
// polynomial on |r| ≤ ½ ln 2, scaling by 2^k with v_ldexp_f64 (≈1 ulp).  The argument is clamped instead: below −745 the
// result is a denormal no sum can see, above 709.7 it saturates near DBL_MAX.  The clamps swallow a NaN argument — callers
// detect non-finite inputs themselves (the mixture passes poison their entropy sum, the HGF kernel checks its moments).
__device__ __forceinline__ double exp_core(double x) {
    const double k = __builtin_rint(x * 1.4426950408889634074);
    double r = __builtin_fma(k, -6.93147180369123816490e-01, x);
    r = __builtin_fma(k, -1.90821492927058770002e-10, r);
    double p = 2.08767569878680989792e-09;   // 1/12!
    p = __builtin_fma(p, r, 2.50521083854417187751e-08);   // 1/11!
    p = __builtin_fma(p, r, 2.75573192239858906526e-07);   // 1/10!
    p = __builtin_fma(p, r, 2.75573192239858906526e-06);   // 1/9!
    p = __builtin_fma(p, r, 2.48015873015873015873e-05);   // 1/8!
    p = __builtin_fma(p, r, 1.98412698412698412698e-04);   // 1/7!
    p = __builtin_fma(p, r, 1.38888888888888888889e-03);   // 1/6!
    p = __builtin_fma(p, r, 8.33333333333333333333e-03);   // 1/5!
    p = __builtin_fma(p, r, 4.16666666666666666667e-02);   // 1/4!
    p = __builtin_fma(p, r, 1.66666666666666666667e-01);   // 1/3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_amdgcn_ldexp(p, (int)k);
}
__device__ __forceinline__ double exp_nonpos(double x) { return exp_core(fmax(x, -745.0)); }                 // x ≤ 0
__device__ __forceinline__ double exp_bounded(double x) { return exp_core(fmin(fmax(x, -745.0), 709.7)); }  // any x

// per-component constants of the responsibility rule from the current marginals:
//   logit_k(y) = E log s_k − ½[log2π − E log p_k + E p_k (v_k + (y − m̄_k)²)]  = c_k − h_k (y − m̄_k)²  (+ const)
template <int KT>
__device__ __forceinline__ void gmm_derive(const GmmParams& p, int k) {
    const double* par = p.par;
    double asum = 0.0;
    for (int j = 0; j < p.K; ++j) asum += par[4 * KT + j];
    if (k < p.K) {
        const double Ep = par[2 * KT + k] / par[3 * KT + k];
        const double Elp = digamma_dev(par[2 * KT + k]) - log(par[3 * KT + k]);
        const double Els = digamma_dev(par[4 * KT + k]) - digamma_dev(asum);
        p.drv[k] = Els + 0.5 * Elp - 0.5 * Ep * par[KT + k];
        p.drv[KT + k] = 0.5 * Ep;
        p.drv[2 * KT + k] = par[k];
    } else {  // padding components never receive responsibility
        p.drv[k] = -1e300;
        p.drv[KT + k] = 0.0;
        p.drv[2 * KT + k] = 0.0;
    }
}

template <int KT>
__global__ void __launch_bounds__(256) k_gmm_init(GmmParams p) {
    const int k = threadIdx.x;
    if (k < KT) gmm_derive<KT>(p, k);
}

template <int KT, bool RESP>
__global__ void __launch_bounds__(256) k_gmm_pass(GmmParams p) {
    __shared__ double sh[4][3 * KT + 1];
    // The 3·KT per-component constants do not fit the scalar register file at KT = 16 (72 SGPRs were spilled to VGPR lanes and
    // restored with 54 v_readlane per point — VALU issue slots of a VALU-bound kernel).  They are read from LDS inside the
    // loop instead (wave-uniform broadcast reads on the otherwise idle LDS pipe); the per-iteration opaque zero keeps the
    // reads in the loop (hoisted, they would occupy 96 vector registers).  Small KT keeps them in scalar registers.
    constexpr bool CST_LDS = KT >= 16;
    __shared__ __attribute__((aligned(16))) double sc[CST_LDS ? 4 * KT : 4];  // [k][c, h, m, pad]
    double c[CST_LDS ? 1 : KT], h[CST_LDS ? 1 : KT], m[CST_LDS ? 1 : KT];
    if constexpr (CST_LDS) {
        for (int q = threadIdx.x; q < 3 * KT; q += 256) sc[4 * (q % KT) + q / KT] = p.drv[q];
        __syncthreads();
    } else {
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            c[k] = p.drv[k];
            h[k] = p.drv[KT + k];
            m[k] = p.drv[2 * KT + k];
        }
    }
    double S0[KT], S1[KT], S2[KT], Hz = 0.0;
#pragma unroll
    for (int k = 0; k < KT; ++k) S0[k] = S1[k] = S2[k] = 0.0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < p.N; i += stride) {
        const double y = p.y[i];
        double lg[KT], mx = -1e308;
        int z = 0;
        if constexpr (CST_LDS) asm volatile("v_mov_b32 %0, 0" : "=v"(z));
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            double ck, hk, mk;
            if constexpr (CST_LDS) {
                const double2 ch = *reinterpret_cast<const double2*>(sc + 4 * k + z);
                ck = ch.x; hk = ch.y; mk = sc[4 * k + 2 + z];
            } else {
                ck = c[k]; hk = h[k]; mk = m[k];
            }
            const double d = y - mk;
            lg[k] = ck - hk * d * d;
            mx = fmax(mx, lg[k]);
        }
        double Z = 0.0, e[KT], sl = 0.0;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            e[k] = exp_nonpos(lg[k] - mx);
            Z += e[k];
        }
        const double zi = 1.0 / Z, y2 = y * y;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            const double pi = e[k] * zi;
            sl += pi * (lg[k] - mx);
            S0[k] += pi;
            S1[k] += pi * y;
            S2[k] += pi * y2;
            e[k] = pi;
        }
        Hz += (log(Z) - sl) + (y - y);  // H[q(z_i)] = −Σ π log π;  y − y: NaN for a non-finite observation (exp_nonpos hides it)
        if (RESP) {
            double* r = p.resp + i * p.K;
            for (int k = 0; k < p.K && k < KT; ++k) r[k] = e[k];
        }
    }
    // fixed-shape reduction: wave shuffles, then the 4 waves of the block through LDS
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        const double a = wave_sum(S0[k]), b = wave_sum(S1[k]), cc = wave_sum(S2[k]);
        if (lane == 0) {
            sh[w][k] = a;
            sh[w][KT + k] = b;
            sh[w][2 * KT + k] = cc;
        }
    }
    {
        const double a = wave_sum(Hz);
        if (lane == 0) sh[w][3 * KT] = a;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < 3 * KT + 1; q += 256)
        p.partial[(size_t)blockIdx.x * (3 * KT + 1) + q] = ((sh[0][q] + sh[1][q]) + sh[2][q]) + sh[3][q];
}

template <int KT>
__global__ void __launch_bounds__(256) k_gmm_reduce(GmmParams p) {
    __shared__ double sh[256];
    constexpr int NQ = 3 * KT + 1;
    const int q = blockIdx.x;  // one workgroup per statistic
    double s = 0.0;
    for (int b = threadIdx.x; b < p.nblocks; b += 256) s += p.partial[(size_t)b * NQ + q];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int wd = 128; wd > 0; wd >>= 1) {
        if ((int)threadIdx.x < wd) sh[threadIdx.x] += sh[threadIdx.x + wd];
        __syncthreads();
    }
    if (threadIdx.x == 0) p.totals[q] = sh[0];
}

template <int KT, bool FE>
__global__ void __launch_bounds__(64) k_gmm_update(GmmParams p) {
    __shared__ double fsh[64];
    const int k = threadIdx.x;
    const int K = p.K;
    double fk = 0.0;
    double nm = 0, nv = 1, na = 1, nb = 1, nal = 1, S0 = 0, S1 = 0, S2 = 0;
    const double* pr = p.prior;
    bool bad = false;
    if (k < K) {
        S0 = p.totals[k];
        S1 = p.totals[KT + k];
        S2 = p.totals[2 * KT + k];
        const double Ep_old = p.par[2 * KT + k] / p.par[3 * KT + k];
        // q(m[k]) = N(μ0, v0) × Π_i N(y_i, (π_ik E p_k)⁻¹)            (ξ, Λ) sums
        const double L = 1.0 / pr[KT + k] + Ep_old * S0;
        const double xi = pr[k] / pr[KT + k] + Ep_old * S1;
        nv = 1.0 / L;
        nm = xi * nv;
        // q(p[k]) = Gamma(a0, b0) × Π_i Gamma(1 + π/2, π ½[(y_i − m̄)² + v])
        na = pr[2 * KT + k] + 0.5 * S0;
        nb = pr[3 * KT + k] + 0.5 * (S2 - 2.0 * nm * S1 + nm * nm * S0 + nv * S0);  // with the NEW q(m[k])
        // q(s) = Dirichlet(α0) × Π_i Dirichlet(1 + π_i)
        nal = pr[4 * KT + k] + S0;
        bad = !(nv > 0.0) || !(nb > 0.0);
    }
    __syncthreads();
    if (k < K) {
        p.par[k] = nm;
        p.par[KT + k] = nv;
        p.par[2 * KT + k] = na;
        p.par[3 * KT + k] = nb;
        p.par[4 * KT + k] = nal;
        double* h = p.hist + (size_t)p.iteration * 5 * K;
        h[k] = nm;
        h[K + k] = nv;
        h[2 * K + k] = na;
        h[3 * K + k] = nb;
        h[4 * K + k] = nal;
    }
    __syncthreads();
    if (FE) {
        double asum = 0.0, a0sum = 0.0;
        for (int j = 0; j < K; ++j) {
            asum += p.par[4 * KT + j];
            a0sum += pr[4 * KT + j];
        }
        if (k < K) {
            const double Ep = na / nb, Elp = digamma_dev(na) - log(nb), dga = digamma_dev(nal), Els = dga - digamma_dev(asum);
            const double mu0 = pr[k], v0 = pr[KT + k], a0 = pr[2 * KT + k], b0 = pr[3 * KT + k], al0 = pr[4 * KT + k];
            fk += 0.5 * ((kLog2Pi - Elp) * S0 + Ep * (nv * S0 + S2 - 2.0 * nm * S1 + nm * nm * S0));  // Σ_i π_ik U_k(i)
            fk += -S0 * Els;                                                                           // Categorical
            const double dm = nm - mu0;
            fk += 0.5 * (kLog2Pi + log(v0) + (dm * dm + nv) / v0) - 0.5 * (kLog2Pi + 1.0 + log(nv));    // U_m − H[m]
            fk += (-a0 * log(b0) + lgamma(a0) - (a0 - 1.0) * Elp + b0 * Ep) -
                  (na - log(nb) + lgamma(na) + (1.0 - na) * digamma_dev(na));                          // U_p − H[p]
            if (K > 1) fk += (lgamma(al0) - (al0 - 1.0) * Els) - (lgamma(nal) - (nal - 1.0) * dga);    // Dirichlet parts
        }
        fsh[k] = (k < K) ? fk : 0.0;
        __syncthreads();
        if (k == 0) {
            double F = -p.totals[3 * KT];  // −Σ_i H[q(z_i)]
            for (int j = 0; j < K; ++j) F += fsh[j];
            if (K > 1) F += -lgamma(a0sum) - (-lgamma(asum) + (asum - K) * digamma_dev(asum));
            p.fe[p.iteration] = F;
            if (!is_finite(F)) atomicOr(p.status, ST_NONFINITE);
        }
    }
    if (bad) atomicOr(p.status, ST_NOT_POSDEF);
    __syncthreads();
    if (k < KT) gmm_derive<KT>(p, k);
}

}  // namespace rxhip
