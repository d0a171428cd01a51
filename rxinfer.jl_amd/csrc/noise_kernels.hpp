// noise_kernels.hpp — the Wishart side of the first COMPOSED graph (round 4): a state-space chain whose observation-noise precision is
// unknown,
//     x[1] ~ MvNormal(μ = m0, Σ = V0);  x[t] ~ MvNormal(μ = A x[t-1], Σ = P);  W ~ Wishart(ν0, S0);  y[t] ~ MvNormal(μ = B x[t], Λ = W)
// with q(x[1..T], W) = q(x[1..T]) q(W): the chain of test/models/statespace/mlgssm_test.jl:9-14 (benchmark model, notebook cell 4) with the
// observation nodes in the precision parametrisation and the node pair of test/models/iid/mv_iid_precision_tests.jl:11-15 on W
// (factorisation handling: src/model/plugins/reactivemp_inference.jl:499-501).  Every rule it needs was on the device already — the chain
// rules in lgssm_kernels.hpp, the Wishart rules in mvgmm_kernels.hpp — what was missing is the schedule that alternates them:
//   q(x)  MvNormalMeanPrecision(:μ) with q(W) sends the Gaussian with precision E[W] = νV toward B x[t]; the rest of the chain is
//         sum-product: ONE sweep of the state-space kernels with this chain's constants Q⁻¹ ← E[W]  (each chain its own model block:
//         B′E[W]B, B′E[W], E[W], dy log 2π − log|E[W]| — written by the kernels below, no host round trip between iterations);
//   q(W)  MvNormalMeanPrecision(:Λ) sends Wishart(dy + 2, E[(y − Bx)(y − Bx)′]⁻¹) per observation; product with the prior in natural
//         parameters: ν = ν0 + T, V⁻¹ = S0⁻¹ + Σ_t [(y_t − B m_t)(y_t − B m_t)′ + B V_t B′]   (accumulated per segment by the backward sweep itself,
//         k_backward_noise in lgssm_kernels.hpp; engines whose sweep is one segment: k_noise_moments, an HBM-bound pass over the posteriors,
//         lane = chain, time slices; either way the partials are reduced in a fixed order; k_noise_update: one lane per chain finishes the 4×4 algebra);
//   F     Bethe free energy of the iteration's marginals = the sweep's −log p̃(y | Q = E_old[W]⁻¹) + T/2 (log|E_old W| − E_new log|W|)
//         + ½ tr((E_new W − E_old W) Σ_t E[r_t r_t′]) + KL(q_new(W) ‖ p(W)): one extra slot per chain in the sweep's free-energy partials.
// Order per iteration: q(x) with the previous q(W), then q(W) with the new q(x) (the order the mixture engines assume for q(m), q(w); the CPU
// checker of tests/test_noise_vmp_gpu.py restates it and is pinned to the iid model in the constant-state limit).  d, dy ≤ 4 (the lanes path).
#pragma once
#include "lgssm_kernels.hpp"
#include "mvgmm_kernels.hpp"

namespace rxhip {

struct NoiseParams {
    long long T, n_chains;
    int S;                  // segments of the sweep: the extra free-energy slot is S + 1
    const double* y;        // [T][chain][dy]
    const double* mean;     // [T][chain][d]     posteriors of the sweep that just ran
    const double* cov;      // [T][chain][d][d]
    const double* B;        // [dy][d]
    double* cst;            // [chain][CstLayout::SIZE]: the observation-side constants are rewritten per chain
    const double* prior;    // ν0 | S0⁻¹ [dy][dy] | log|S0| | ν_init | V_init [dy][dy]
    double* state;          // [chain][1 + dy·dy]: ν | V of q(W)
    double* hist;           // null, or [iterations][chain][1 + dy·dy]
    double* fe_part;        // [slot][chain]
    int iteration;
    int* status;
    double* part;           // [slices][NS][chain]: partial residual second moments of k_noise_moments
    int slices;             // time slices (fixed per engine: the summation order is the same on every run)
    int moments_in_sweep;   // the backward sweep left the partial moments per SEGMENT in `part` (slices = segments): k_noise_moments is not launched
};
constexpr int NOISE_MAX_SLICES = 128;
template <int DY>
struct NoisePrior {
    static constexpr int NU0 = 0, S0I = 1, LDS0 = 1 + DY * DY, NUI = LDS0 + 1, VI = NUI + 1, SIZE = VI + DY * DY;
};

// the observation-side constants of one chain from E[W] (CstLayout: LOBS = B′WB, G = B′W, QI = W, C0 = dy log 2π − log|W|)
template <int D, int DY>
__device__ __forceinline__ void noise_write_constants(double* c, const double* B, const Sym<DY>& W, double logdetW) {
    using CL = CstLayout<D, DY>;
    double G[D][DY];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int a = 0; a < DY; ++a) {
            double s = 0.0;
#pragma unroll
            for (int b = 0; b < DY; ++b) s += B[b * D + i] * W(b, a);
            G[i][a] = s;
            c[CL::G + i * DY + a] = s;
        }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double s = 0.0;
#pragma unroll
            for (int a = 0; a < DY; ++a) s += G[i][a] * B[a * D + j];
            c[CL::LOBS + sidx(i, j)] = s;
        }
#pragma unroll
    for (int k = 0; k < Dim<DY>::NS; ++k) c[CL::QI + k] = W.v[k];
    c[CL::C0] = DY * 1.8378770664093454835606594728112 - logdetW;
}

// run start: q(W) ← the @initialization marginal, every chain's constants from its E[W]
template <int D, int DY>
__global__ void __launch_bounds__(64) k_noise_reset(NoiseParams p) {
    using NP = NoisePrior<DY>;
    const long long chain = (long long)blockIdx.x * 64 + threadIdx.x;
    if (chain >= p.n_chains) return;
    const double nu = p.prior[NP::NUI];
    Sym<DY> V, W, Wi;
    double* st = p.state + chain * (1 + DY * DY);
    st[0] = nu;
#pragma unroll
    for (int a = 0; a < DY; ++a)
#pragma unroll
        for (int b = 0; b < DY; ++b) {
            const double v = p.prior[NP::VI + a * DY + b];
            st[1 + a * DY + b] = v;
            if (b <= a) { V(a, b) = v; W(a, b) = nu * v; }
        }
    double det;
    if (!spd_inv<DY>(W, Wi, det)) atomicOr(p.status, ST_NOT_POSDEF);
    noise_write_constants<D, DY>(p.cst + chain * CstLayout<D, DY>::SIZE, p.B, W, log(det));
}

// after the sweep of an iteration, 1: Σ_t [(y_t − B m_t)(y_t − B m_t)′ + B V_t B′] per chain.  Lane = chain (the posterior arrays are
// [t][chain][·]: a wave reads 64 chains' worth of one time index as contiguous runs), workgroup row = time slice: slice s takes
// t ≡ s (mod slices), its partial sums go to part[s][·][chain].  2 GB of posteriors at d = 4 × 1024 chains × T = 10⁴: an HBM-bound pass.
template <int D, int DY>
__global__ void __launch_bounds__(64) k_noise_moments(NoiseParams p) {
    constexpr int NS = Dim<DY>::NS;
    const long long chain = (long long)blockIdx.x * 64 + threadIdx.x;
    const int slice = blockIdx.y;
    if (chain >= p.n_chains) return;
    double Bm[DY][D];
#pragma unroll
    for (int a = 0; a < DY; ++a)
#pragma unroll
        for (int k = 0; k < D; ++k) Bm[a][k] = p.B[a * D + k];
    double acc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) acc[k] = 0.0;
    for (long long t = slice; t < p.T; t += p.slices) {
        const double* m = p.mean + (t * p.n_chains + chain) * D;
        const double* C = p.cov + (t * p.n_chains + chain) * D * D;
        const double* yt = p.y + (t * p.n_chains + chain) * DY;
        double r[DY], BV[DY][D];
#pragma unroll
        for (int a = 0; a < DY; ++a) {
            double s = yt[a];
#pragma unroll
            for (int k = 0; k < D; ++k) s -= Bm[a][k] * m[k];
            r[a] = s;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                double v = 0.0;
#pragma unroll
                for (int l = 0; l < D; ++l) v += Bm[a][l] * C[l * D + k];
                BV[a][k] = v;
            }
        }
#pragma unroll
        for (int a = 0; a < DY; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) {
                double v = r[a] * r[b];
#pragma unroll
                for (int k = 0; k < D; ++k) v += BV[a][k] * Bm[b][k];
                acc[sidx(a, b)] += v;
            }
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) p.part[((long long)slice * NS + k) * p.n_chains + chain] = acc[k];
}

// 2: the Wishart update of every chain (one lane per chain: the slices in ascending order — the same sum on every run), its free-energy
// slot, the constants of the next sweep
template <int D, int DY>
__global__ void __launch_bounds__(64) k_noise_update(NoiseParams p) {
    using NP = NoisePrior<DY>;
    constexpr int NS = Dim<DY>::NS;
    const long long chain = (long long)blockIdx.x * 64 + threadIdx.x;
    if (chain >= p.n_chains) return;
    double red0[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) red0[k] = 0.0;
#pragma unroll 8   // (same ascending order; the loads of eight slices travel together instead of one round trip per slice)
    for (int sl = 0; sl < p.slices; ++sl)
#pragma unroll
        for (int k = 0; k < NS; ++k) red0[k] += p.part[((long long)sl * NS + k) * p.n_chains + chain];
    Sym<DY> S, Vo, Wo, Vi, Vn, Wn, tmp;
#pragma unroll
    for (int k = 0; k < NS; ++k) S.v[k] = red0[k];
    double* st = p.state + chain * (1 + DY * DY);
    const double nuo = st[0], nu0 = p.prior[NP::NU0], nun = nu0 + (double)p.T;
    bool ok = true;
#pragma unroll
    for (int a = 0; a < DY; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) {
            Vo(a, b) = st[1 + a * DY + b];
            Wo(a, b) = nuo * Vo(a, b);
            Vi(a, b) = p.prior[NP::S0I + a * DY + b] + S(a, b);   // V_new⁻¹ = S0⁻¹ + Σ E[r r′]
        }
    double detWo, detVi;
    ok = spd_inv<DY>(Wo, tmp, detWo) && ok;     // (only log|E_old W| is needed)
    ok = spd_inv<DY>(Vi, Vn, detVi) && ok;
    const double ldWo = log(detWo), ldV = -log(detVi);
    double trdS = 0.0, trS0W = 0.0;
#pragma unroll
    for (int a = 0; a < DY; ++a)
#pragma unroll
        for (int b = 0; b < DY; ++b) {
            const double wn = nun * Vn(a, b);
            if (b <= a) Wn(a, b) = wn;
            trdS += (wn - Wo(a, b)) * S(a, b);
            trS0W += p.prior[NP::S0I + a * DY + b] * wn;
        }
    const double LOG2 = 0.69314718055994530942;
    const double Elw = mvdigamma_dev(0.5 * nun, DY) + DY * LOG2 + ldV;
    double F = 0.5 * (double)p.T * (ldWo - Elw) + 0.5 * trdS;
    // Wishart prior node of W minus H[q(W)]
    F += -(0.5 * (nu0 - DY - 1.0) * Elw - 0.5 * trS0W - 0.5 * nu0 * DY * LOG2 - 0.5 * nu0 * p.prior[NP::LDS0] - mvlgamma_dev(0.5 * nu0, DY));
    F -= 0.5 * (DY + 1.0) * ldV + 0.5 * DY * (DY + 1.0) * LOG2 + mvlgamma_dev(0.5 * nun, DY) - 0.5 * (nun - DY - 1.0) * mvdigamma_dev(0.5 * nun, DY) +
         0.5 * nun * DY;
    p.fe_part[(long long)(p.S + 1) * p.n_chains + chain] = -F;   // (the reduction negates the sum of the slots)
    st[0] = nun;
#pragma unroll
    for (int a = 0; a < DY; ++a)
#pragma unroll
        for (int b = 0; b < DY; ++b) st[1 + a * DY + b] = Vn(a, b);
    if (p.hist) {
        double* h = p.hist + ((long long)p.iteration * p.n_chains + chain) * (1 + DY * DY);
        h[0] = nun;
#pragma unroll
        for (int a = 0; a < DY; ++a)
#pragma unroll
            for (int b = 0; b < DY; ++b) h[1 + a * DY + b] = Vn(a, b);
    }
    double detWn;
    ok = spd_inv<DY>(Wn, tmp, detWn) && ok;
    noise_write_constants<D, DY>(p.cst + chain * CstLayout<D, DY>::SIZE, p.B, Wn, log(detWn));
    if (!ok) atomicOr(p.status, ST_NOT_POSDEF);
}

}  // namespace rxhip
