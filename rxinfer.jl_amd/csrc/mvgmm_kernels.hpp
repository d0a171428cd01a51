// mvgmm_kernels.hpp — mean-field VMP for the MULTIVARIATE Gaussian mixture (d = 1…4) on gfx950.
//
// Reference model and rules replaced (bodies in the un-vendored ReactiveMP.jl / ExponentialFamily.jl):
//   a10 NormalMixture(:switch | :m[k] | :p[k]) with MvNormalMeanPrecision components and Wishart precisions
//       test/models/mixtures/gmm_multivariate_tests.jl:6-32 (m[k] ~ MvNormal(mean, cov), w[k] ~ Wishart(ν, V),
//       s ~ Dirichlet, z[i] ~ Categorical(s), y[i] ~ NormalMixture(switch = z[i], m = m, p = w))
//   a7  Bethe free energy of the mean-field factorisation (MvNormal / Wishart / Dirichlet / Categorical node energies
//       and entropies in closed form)
// Same structure and schedule as gmm_kernels.hpp (univariate): one pass streams the observations (8d B/point), forms
// q(z_i) and accumulates per component Σπ, Σπy, Σπyy' (1 + d + d(d+1)/2 numbers) and Σ_i H[q(z_i)] in registers;
// fixed-shape reductions; one thread per component then forms q(m[k]) (with the previous E[W]), q(w[k]) (with the new
// q(m[k])), q(s), the free energy and the constants of the next pass.  Several GPUs: the statistics are the all-reduce
// payload, exactly as in the univariate engine.  d = 1 is the univariate model (Wishart(ν, V) = Gamma(ν/2, 1/(2V))).
#pragma once
#include <hip/hip_runtime.h>

#include "gmm_kernels.hpp"

namespace rxhip {

template <int D>
struct MvgDim {
    static constexpr int NS = D * (D + 1) / 2;
    static constexpr int STAT = 1 + D + NS;        // Σπ | Σπy | Σπyy' (packed lower)
    static constexpr int SZ = 2 + D + 2 * D * D;   // mean | cov | nu | V | alpha
    static constexpr int DRV = 1 + NS + D;         // c_k | H_k = ½E[W] (packed, off-diagonals doubled) | m̄_k
    static constexpr int PRI = D + 2 * D * D + 4;  // mu0 | S0⁻¹ | nu0 | V0⁻¹ | alpha0 | log|S0| | log|V0|
};

struct MvgParams {
    long long N;
    int K;
    const double* y;      // [N][D]
    double* resp;         // [N][K] or nullptr
    double* state;        // [K][SZ]   current marginals
    double* drv;          // [KT][DRV] constants of the responsibility rule
    const double* prior;  // [K][PRI]
    double* partial;      // [blocks][KT·STAT + 1]
    double* totals;       // [KT·STAT + 1]
    double* hist;         // [iterations][K][SZ]
    double* fe;           // [iterations]
    int iteration, nblocks, write_resp;
    int* status;
};

__device__ __forceinline__ double mvdigamma_dev(double a, int d) {
    double s = 0.0;
    for (int i = 0; i < d; ++i) s += digamma_dev(a - 0.5 * i);
    return s;
}
__device__ __forceinline__ double mvlgamma_dev(double a, int d) {
    double s = 0.25 * d * (d - 1) * 1.1447298858494001741434273513531;
    for (int i = 0; i < d; ++i) s += lgamma(a - 0.5 * i);
    return s;
}
template <int D>
__device__ __forceinline__ void load_sym(const double* M, Sym<D>& S) {  // row-major full -> packed
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) S(i, j) = 0.5 * (M[i * D + j] + M[j * D + i]);
}
template <int D>
__device__ __forceinline__ void store_sym(const Sym<D>& S, double* M) {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) M[i * D + j] = S(i, j);
}

// constants of the responsibility rule from the current marginals (one thread per component):
//   logit_k(y) = E log s_k − ½[d log 2π − E log|W_k| + tr(E[W_k]((y − m̄)(y − m̄)' + Σ_m))]
//              = c_k − (y − m̄_k)' H_k (y − m̄_k)      (+ a constant common to all k)
template <int D, int KT>
__device__ __forceinline__ bool mvg_derive(const MvgParams& p, int k) {
    using MD = MvgDim<D>;
    double* dr = p.drv + k * MD::DRV;
    if (k >= p.K) {  // padding components never receive responsibility
        dr[0] = -1e300;
        for (int i = 1; i < MD::DRV; ++i) dr[i] = 0.0;
        return true;
    }
    double asum = 0.0;
    for (int j = 0; j < p.K; ++j) asum += p.state[j * MD::SZ + MD::SZ - 1];
    const double* st = p.state + k * MD::SZ;
    const double nu = st[D + D * D];
    Sym<D> V, Vi, Cm;
    load_sym<D>(st + D + D * D + 1, V);
    load_sym<D>(st + D, Cm);
    double detV;
    const bool ok = spd_inv<D>(V, Vi, detV);
    const double Elw = mvdigamma_dev(0.5 * nu, D) + D * 0.69314718055994530942 + log(detV);
    const double Els = digamma_dev(st[MD::SZ - 1]) - digamma_dev(asum);
    double tr = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) tr += nu * V(i, j) * Cm(j, i);
    dr[0] = Els + 0.5 * Elw - 0.5 * tr;
    int q = 1;
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) dr[q++] = (i == j ? 0.5 : 1.0) * nu * V(i, j);  // quadratic form over the lower triangle
#pragma unroll
    for (int i = 0; i < D; ++i) dr[q++] = st[i];
    return ok;
}

template <int D, int KT>
__global__ void __launch_bounds__(64) k_mvg_init(MvgParams p) {
    const int k = threadIdx.x;
    if (k < KT && !mvg_derive<D, KT>(p, k)) atomicOr(p.status, ST_NOT_POSDEF);
}

template <int D, int KT, bool RESP>
__global__ void __launch_bounds__(256) k_mvg_pass(MvgParams p) {
    using MD = MvgDim<D>;
    constexpr int NQ = KT * MD::STAT + 1;
    __shared__ double sdrv[KT * MD::DRV];
    __shared__ double sh[4][NQ];
    for (int q = threadIdx.x; q < KT * MD::DRV; q += 256) sdrv[q] = p.drv[q];
    __syncthreads();
    double S[KT][MD::STAT], Hz = 0.0;
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
        for (int q = 0; q < MD::STAT; ++q) S[k][q] = 0.0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < p.N; i += stride) {
        double y[D];
#pragma unroll
        for (int a = 0; a < D; ++a) y[a] = p.y[i * D + a];
        double lg[KT], mx = -1e308;
        // the derived parameters stay in LDS: the address is laundered per point, else the compiler hoists all KT · DRV loads out of the loop
        // into registers — next to the KT · STAT accumulators that was 512 registers and 184 bytes of scratch at d = 4, K = 8
        const double* sd = sdrv;
        asm volatile("" : "+v"(sd));
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            const double* dr = sd + k * MD::DRV;  // wave-uniform: broadcast LDS reads
            double dv[D], qf = 0.0;
#pragma unroll
            for (int a = 0; a < D; ++a) dv[a] = y[a] - dr[1 + MD::NS + a];
            int q = 1;
#pragma unroll
            for (int a = 0; a < D; ++a)
#pragma unroll
                for (int b = 0; b <= a; ++b) qf += dr[q++] * dv[a] * dv[b];
            lg[k] = dr[0] - qf;
            mx = fmax(mx, lg[k]);
        }
        double Z = 0.0, e[KT], sl = 0.0;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            e[k] = exp_nonpos(lg[k] - mx);
            Z += e[k];
        }
        const double zi = 1.0 / Z;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            const double pi = e[k] * zi;
            sl += pi * (lg[k] - mx);
            S[k][0] += pi;
            int q = 1 + D;
#pragma unroll
            for (int a = 0; a < D; ++a) {
                const double pa = pi * y[a];
                S[k][1 + a] += pa;
#pragma unroll
                for (int b = 0; b <= a; ++b) S[k][q++] += pa * y[b];
            }
            e[k] = pi;
        }
        double ysum = 0.0;
#pragma unroll
        for (int a = 0; a < D; ++a) ysum += y[a];
        Hz += (log(Z) - sl) + (ysum - ysum);  // H[q(z_i)] = −Σ π log π;  NaN for a non-finite observation (exp_nonpos hides it)
        if (RESP) {
            double* r = p.resp + i * p.K;
            for (int k = 0; k < p.K && k < KT; ++k) r[k] = e[k];
        }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
        for (int q = 0; q < MD::STAT; ++q) {
            const double a = wave_sum(S[k][q]);
            if (lane == 0) sh[w][k * MD::STAT + q] = a;
        }
    {
        const double a = wave_sum(Hz);
        if (lane == 0) sh[w][NQ - 1] = a;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < NQ; q += 256) p.partial[(size_t)blockIdx.x * NQ + q] = ((sh[0][q] + sh[1][q]) + sh[2][q]) + sh[3][q];
}

static __global__ void __launch_bounds__(256) k_mvg_reduce(MvgParams p, int nq) {
    __shared__ double sh[256];
    const int q = blockIdx.x;  // one workgroup per statistic
    double s = 0.0;
    for (int b = threadIdx.x; b < p.nblocks; b += 256) s += p.partial[(size_t)b * nq + q];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int wd = 128; wd > 0; wd >>= 1) {
        if ((int)threadIdx.x < wd) sh[threadIdx.x] += sh[threadIdx.x + wd];
        __syncthreads();
    }
    if (threadIdx.x == 0) p.totals[q] = sh[0];
}

template <int D, int KT, bool FE>
__global__ void __launch_bounds__(64) k_mvg_update(MvgParams p) {
    using MD = MvgDim<D>;
    __shared__ double fsh[64];
    const int k = threadIdx.x, K = p.K;
    bool ok = true;
    double fk = 0.0;
    if (k < K) {
        const double* tot = p.totals + k * MD::STAT;
        const double* pr = p.prior + k * MD::PRI;
        double* st = p.state + k * MD::SZ;
        const double S0 = tot[0];
        double S1[D];
        Sym<D> S2;
#pragma unroll
        for (int a = 0; a < D; ++a) S1[a] = tot[1 + a];
#pragma unroll
        for (int q = 0; q < MD::NS; ++q) S2.v[q] = tot[1 + D + q];
        // previous q(w[k]): E[W] = νV
        const double nu_old = st[D + D * D];
        Sym<D> EW, S0i, V0i;
        load_sym<D>(st + D + D * D + 1, EW);
#pragma unroll
        for (int q = 0; q < MD::NS; ++q) EW.v[q] *= nu_old;
        load_sym<D>(pr + D, S0i);
        load_sym<D>(pr + D + D * D + 1, V0i);
        const double nu0 = pr[D + D * D], al0 = pr[D + 2 * D * D + 1], ldS0 = pr[D + 2 * D * D + 2], ldV0 = pr[D + 2 * D * D + 3];
        // q(m[k]) = N(μ0, S0) × Π_i N(y_i, (π_ik E[W])⁻¹):  Λ = S0⁻¹ + Σπ E[W],  ξ = S0⁻¹μ0 + E[W] Σπy
        Sym<D> Lam, Cm;
#pragma unroll
        for (int q = 0; q < MD::NS; ++q) Lam.v[q] = S0i.v[q] + S0 * EW.v[q];
        double xi[D], mu0[D], mb[D];
#pragma unroll
        for (int a = 0; a < D; ++a) mu0[a] = pr[a];
#pragma unroll
        for (int a = 0; a < D; ++a) {
            double s = 0.0;
#pragma unroll
            for (int b = 0; b < D; ++b) s += S0i(a, b) * mu0[b] + EW(a, b) * S1[b];
            xi[a] = s;
        }
        double detL;
        ok = spd_inv<D>(Lam, Cm, detL) && ok;
        symv<D>(Cm, xi, mb);
        // q(w[k]) = Wishart(ν0, V0) × Π_i (…):  ν = ν0 + Σπ,  V⁻¹ = V0⁻¹ + Σ_i π E[(y_i − m)(y_i − m)'] with the NEW q(m[k])
        Sym<D> Sc, Vin, Vn;
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) Sc(a, b) = S2(a, b) - mb[a] * S1[b] - S1[a] * mb[b] + S0 * (mb[a] * mb[b] + Cm(a, b));
#pragma unroll
        for (int q = 0; q < MD::NS; ++q) Vin.v[q] = V0i.v[q] + Sc.v[q];
        double detVin;
        ok = spd_inv<D>(Vin, Vn, detVin) && ok;
        const double nu = nu0 + S0, al = al0 + S0;
        // store the new marginals (+ history)
#pragma unroll
        for (int a = 0; a < D; ++a) st[a] = mb[a];
        store_sym<D>(Cm, st + D);
        st[D + D * D] = nu;
        store_sym<D>(Vn, st + D + D * D + 1);
        st[MD::SZ - 1] = al;
        double* h = p.hist + ((size_t)p.iteration * K + k) * MD::SZ;
        for (int q = 0; q < MD::SZ; ++q) h[q] = st[q];
        if (FE) {
            const double ldV = -log(detVin), ldC = -log(detL);
            const double Elw = mvdigamma_dev(0.5 * nu, D) + D * 0.69314718055994530942 + ldV;
            double trWS = 0.0, trV0W = 0.0, trS0 = 0.0;
#pragma unroll
            for (int a = 0; a < D; ++a)
#pragma unroll
                for (int b = 0; b < D; ++b) {
                    trWS += nu * Vn(a, b) * Sc(b, a);
                    trV0W += V0i(a, b) * nu * Vn(b, a);
                    trS0 += S0i(a, b) * (Cm(b, a) + (mb[b] - mu0[b]) * (mb[a] - mu0[a]));
                }
            fk += 0.5 * (S0 * (D * kLog2Pi - Elw) + trWS);                                                   // Σ_i π_ik U_k(i)
            fk += 0.5 * (D * kLog2Pi + ldS0 + trS0) - 0.5 * (D * (kLog2Pi + 1.0) + ldC);                      // U_m − H[m]
            fk += -(0.5 * (nu0 - D - 1.0) * Elw - 0.5 * trV0W - 0.5 * nu0 * D * 0.69314718055994530942 - 0.5 * nu0 * ldV0 -
                    mvlgamma_dev(0.5 * nu0, D));                                                              // U_w
            fk -= 0.5 * (D + 1.0) * ldV + 0.5 * D * (D + 1.0) * 0.69314718055994530942 + mvlgamma_dev(0.5 * nu, D) -
                  0.5 * (nu - D - 1.0) * mvdigamma_dev(0.5 * nu, D) + 0.5 * nu * D;                           // − H[w]
        }
    }
    __syncthreads();
    if (FE) {
        double asum = 0.0, a0sum = 0.0;
        for (int j = 0; j < K; ++j) {
            asum += p.state[j * MD::SZ + MD::SZ - 1];
            a0sum += p.prior[j * MD::PRI + D + 2 * D * D + 1];
        }
        if (k < K) {
            const double al = p.state[k * MD::SZ + MD::SZ - 1], al0 = p.prior[k * MD::PRI + D + 2 * D * D + 1];
            const double dga = digamma_dev(al), Els = dga - digamma_dev(asum), S0 = p.totals[k * MD::STAT];
            fk += -S0 * Els;                                                                              // Categorical
            if (K > 1) fk += (lgamma(al0) - (al0 - 1.0) * Els) - (lgamma(al) - (al - 1.0) * dga);         // Dirichlet parts
        }
        fsh[k] = (k < K) ? fk : 0.0;
        __syncthreads();
        if (k == 0) {
            double F = -p.totals[KT * MD::STAT];  // −Σ_i H[q(z_i)]
            for (int j = 0; j < K; ++j) F += fsh[j];
            if (K > 1) F += -lgamma(a0sum) - (-lgamma(asum) + (asum - K) * digamma_dev(asum));
            p.fe[p.iteration] = F;
            if (!is_finite(F)) atomicOr(p.status, ST_NONFINITE);
        }
    }
    if (!ok) atomicOr(p.status, ST_NOT_POSDEF);
    __syncthreads();
    if (k < KT && !mvg_derive<D, KT>(p, k)) atomicOr(p.status, ST_NOT_POSDEF);
}

}  // namespace rxhip
