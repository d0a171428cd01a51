// dense8_kernels.hpp — the information-form sweep of a chain with d ≤ 8 INSIDE one wavefront (round 4).
//
// What this is for: batches whose chains own their Riccati recursion — `missing` observations anywhere in the data
// (docs/src/manuals/inference/static.md:98-123, test/inference/prediction_tests.jl:197-213): the covariances differ per chain and
// time index, so nothing can be hoisted into per-model tables and every (chain, step) pays a d×d inverse.  When the chains fill the chip on
// their own the masked schedule runs ONE segment per chain (dense_mseg_kernels.hpp), i.e. T dependent steps per chain: the sweep time is
// the LATENCY of a step.  On the MFMA path a step of a d ≤ 8 chain is one 16×16 tile, half of it padding, with an in-wave 16×16 panel
// inverse: 3.3 µs forward + 1.3 µs backward (profiles/r03/dense_missing_parallel_kernels.txt) — d = 8 × 1024 chains × T = 1000 with 10 %
// missing took 4.8 ms, 9× the fully observed sweep.  Here a chain is one wavefront and an 8×8 matrix is one element per lane
// (lane = 8 i + j): no matrix cores (a 8×8×8 product is 8 FMAs per lane), no workgroup barrier, no padding —
//   inverse   symmetric sweep operator on 2×2 pivot blocks, 4 rounds; a round needs rows 2q, 2q + 1 at column j (the 16 lanes that hold them
//             publish them in LDS: one write, one two-element read) and at column i (= element (i, 2q), (i, 2q + 1) by symmetry: the same
//             8-lane group — DPP row_newbcast, no memory), and the pivot block (uniform LDS reads)
//   products  operands pass through a wave-private LDS block (one write, 8 broadcast reads); constant operands (K = P⁻¹A) sit in registers
// A wavefront that has its SIMD to itself (1024 chains = 1024 wavefronts = one per SIMD) runs such a step as one chain of dependent
// instructions: ≈ 400 of them, 7 cycles each.  With TWO chains per wavefront (NCH = 2: every lane holds its element of both) the two
// chains' instruction streams interleave and fill each other's stalls.
// The arithmetic is kd_forward_info / kd_backward_info's (dense_kernels.hpp: M = Λ_f + A′P⁻¹A, C = M⁻¹, G′ = K C, Λ_f′ = PLW − K C K′ with
// the constant without B′Q⁻¹B where y_t is missing; m_s = C ξ_f + G m_s⁺, V_s = C + G V_s⁺ G′), the same boundary data, constants (DenseCst of
// the 16-padded model — the leading 8×8 block is the model, further padding dimensions are decoupled identities) and free-energy slots, so
// the residual kernel (kd_fe_resid_mfma) and the reduction follow unchanged.  Records are this file's own (8×8 natural order inside the
// record stride of the MFMA path): node-local joints of such an engine come from the sequential re-run, filtering runs keep the MFMA kernels.
#pragma once
#include "dense_kernels.hpp"

namespace rxhip {

constexpr int K8_MAT = 64, K8_VEC = 16;
constexpr int K8_LDS_PER_CHAIN = 3 * K8_MAT + K8_VEC + 16;   // three matrices, a vector, the pivot rows of the inverse
// record of a time index on this path: ξ_f | B′Q⁻¹y (km_gy) | C ξ_f, 16 doubles each as on the MFMA path, then C and G′ as 8×8 blocks in natural
// order — 176 doubles instead of the 560 of a 16×16 tile record (the host passes this stride to km_gy: MsegParams::rec)
constexpr int K8_REC = 48 + 2 * 64, K8_HDR = 48;

__device__ __forceinline__ double k8_gather(double x, int src_lane) {   // x of lane src_lane (any lane of the wave)
    const int a = src_lane << 2;
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(a, __double2hiint(x)), __builtin_amdgcn_ds_bpermute(a, __double2loint(x)));
}
// element (i, P) of the matrix for the lane that holds (i, j): the same 8-lane group — lanes 0–7 of a DPP row hold an even matrix row, lanes
// 8–15 the odd one
template <int P>
__device__ __forceinline__ double k8_row_element(double x, bool iodd) {
    const int hi = __double2hiint(x), lo = __double2loint(x);
    const int eh = __builtin_amdgcn_mov_dpp(hi, 0x150 + P, 0xf, 0xf, true), el = __builtin_amdgcn_mov_dpp(lo, 0x150 + P, 0xf, 0xf, true);
    const int oh = __builtin_amdgcn_mov_dpp(hi, 0x158 + P, 0xf, 0xf, true), ol = __builtin_amdgcn_mov_dpp(lo, 0x158 + P, 0xf, 0xf, true);
    return __hiloint2double(iodd ? oh : eh, iodd ? ol : el);
}
// One round of the symmetric sweep operator on the 2×2 pivot block of rows 2Q, 2Q + 1, for NCH matrices at once (one element of each per lane)
template <int Q, int NCH>
__device__ __forceinline__ void k8_sweep_round(double (&m)[NCH], int lane, int j, bool iodd, bool jodd, bool idiag, double* const (&sb)[NCH], LogProd (&lp)[NCH],
                                               bool (&ok)[NCH]) {
    constexpr int P0 = 2 * Q, P1 = 2 * Q + 1;
    const bool mine = (lane >> 4) == Q;   // this lane holds an element of the pivot rows
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        if (mine) sb[c][lane & 15] = m[c];
    wave_lds_fence();
    double rj0[NCH], rj1[NCH], a[NCH], b[NCH], cc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        rj0[c] = sb[c][j];        // M[P0][j]
        rj1[c] = sb[c][8 + j];    // M[P1][j]
        a[c] = sb[c][P0];
        b[c] = sb[c][P1];
        cc[c] = sb[c][8 + P1];
    }
    wave_lds_fence();             // the next round's publish stays behind these reads
    const bool ip = ((lane >> 4) == Q), jp = (j >> 1) == Q;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const double ri0 = k8_row_element<P0>(m[c], iodd), ri1 = k8_row_element<P1>(m[c], iodd);   // M[i][P0] = M[P0][i], M[i][P1]
        const double det = __builtin_fma(a[c], cc[c], -b[c] * b[c]);
        ok[c] = ok[c] & (a[c] > 0.0) & (det > 0.0);
        lp[c].mul(det);
        double rd = __builtin_amdgcn_rcp(det);
        double e = __builtin_fma(-det, rd, 1.0);
        rd = __builtin_fma(rd, e, rd);
        e = __builtin_fma(-det, rd, 1.0);
        rd = __builtin_fma(rd, e, rd);
        const double ar = a[c] * rd, br = b[c] * rd, cr = cc[c] * rd;
        // (pivot block)⁻¹ · rows, at column j and at column i
        const double u0 = __builtin_fma(cr, rj0[c], -br * rj1[c]), u1 = __builtin_fma(ar, rj1[c], -br * rj0[c]);
        const double w0 = __builtin_fma(cr, ri0, -br * ri1), w1 = __builtin_fma(ar, ri1, -br * ri0);
        double r = __builtin_fma(-ri1, u1, __builtin_fma(-ri0, u0, m[c]));
        const double rowv = iodd ? u1 : u0, colv = jodd ? w1 : w0, blk = idiag ? (iodd ? -ar : -cr) : br;   // −(pivot block)⁻¹ on the block itself
        r = jp ? colv : r;
        r = ip ? rowv : r;
        r = (ip & jp) ? blk : r;
        m[c] = r;
    }
}
// In-wave inverse of symmetric positive definite 8×8 matrices, one element per lane (i = lane >> 3, j = lane & 7): four rounds of the sweep
// operator on 2×2 pivot blocks — the rounds are a chain of dependent exchanges, so their number, not their arithmetic, is the latency of the
// inverse.  After the four rounds the arrays hold −M⁻¹.  The pivot blocks are the Schur complements' diagonal blocks: the product of their
// determinants is det M, and a block that is not positive definite says M is not.  sb: 16 doubles of wave-private LDS per matrix.
template <int NCH>
__device__ __forceinline__ void k8_inverse(double (&m)[NCH], int lane, double* const (&sb)[NCH], LogProd (&lp)[NCH], bool (&ok)[NCH]) {
    const int i = lane >> 3, j = lane & 7;
    const bool iodd = i & 1, jodd = j & 1, idiag = i == j;
    k8_sweep_round<0, NCH>(m, lane, j, iodd, jodd, idiag, sb, lp, ok);
    k8_sweep_round<1, NCH>(m, lane, j, iodd, jodd, idiag, sb, lp, ok);
    k8_sweep_round<2, NCH>(m, lane, j, iodd, jodd, idiag, sb, lp, ok);
    k8_sweep_round<3, NCH>(m, lane, j, iodd, jodd, idiag, sb, lp, ok);
#pragma unroll
    for (int c = 0; c < NCH; ++c) m[c] = -m[c];
}
__device__ __forceinline__ double k8_symmetrise(double x, int i, int j) { return 0.5 * (x + k8_gather(x, 8 * j + i)); }

// forward sweep of ONE segment per chain (S = 1) of the masked schedule: DenseParams::mseg = 2 (the boundary vector is ξ_f(b_0)), pack = 1.
// NCH chains per wavefront: workgroup b runs chains NCH·b … NCH·b + NCH − 1 of the slice (the host launches NCH = 2 on even slices).
template <bool FE, int NCH>
__global__ void __launch_bounds__(64) k8_forward(DenseParams p) {
    constexpr int D16 = 16;
    using C1 = DenseCfg<1>;
    __shared__ __attribute__((aligned(16))) double lds[NCH * K8_LDS_PER_CHAIN];
    const int lane = threadIdx.x, i = lane >> 3, j = lane & 7;
    double *bufA[NCH], *bufB[NCH], *vbuf[NCH], *sb[NCH];
    long long chain[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        double* base = lds + c * K8_LDS_PER_CHAIN;
        bufA[c] = base;               // C_t
        bufB[c] = base + K8_MAT;      // G_t′
        vbuf[c] = base + 3 * K8_MAT;  // ξ_f exchange
        sb[c] = base + 3 * K8_MAT + K8_VEC;
        chain[c] = (long long)blockIdx.x * NCH + c + p.chain0;
    }
    const DenseModel M = dense_model(p, chain[0]);   // one model per engine on this path
    const DenseCst cl = DenseCst::make(D16, p.dy);
    const double* cst = M.cst;
    // constants of this lane: its elements of W, PLW, PLWM; row i and row j of K = P⁻¹A (operands of both products of a step)
    const double w_ij = cst[cl.oW + i * D16 + j], plw_ij = cst[cl.oPLW + i * D16 + j], plwm_ij = cst[cl.oPLWM + i * D16 + j];
    double ki[8], kj[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        ki[k] = cst[cl.oK + i * D16 + k];
        kj[k] = cst[cl.oK + j * D16 + k];
    }
    const long long T = p.T, len = T - 1;
    auto clamp_t = [&](long long t) { return t < T ? t : T - 1; };
    double lam[NCH], xi[NCH][8], xi_own[NCH], gyn[NCH], gyn2[NCH], obn[NCH], obn2[NCH];
    bool ok[NCH];
    LogProd lp[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        lam[c] = p.mbnd[((size_t)chain[c] * p.S * 2 + 0) * (D16 * D16) + i * D16 + j] + w_ij;   // M_1 = Λ_f(b_0) + A′P⁻¹A
        xi_own[c] = p.fstart_m[(chain[c] * p.S) * D16 + i];   // ξ_f: the whole vector in every lane, and component i on its own
#pragma unroll
        for (int k = 0; k < 8; ++k) xi[c][k] = p.fstart_m[(chain[c] * p.S) * D16 + k];
        ok[c] = true;
        gyn[c] = p.filt[(chain[c] * T + clamp_t(1)) * K8_REC + D16 + i];
        gyn2[c] = p.filt[(chain[c] * T + clamp_t(2)) * K8_REC + D16 + i];
        obn[c] = p.obs[chain[c] * T + clamp_t(1)];
        obn2[c] = p.obs[chain[c] * T + clamp_t(2)];
    }
    for (long long s = 0; s < len; ++s) {
        const long long t = s + 1;
        double gyc[NCH];
        bool miss[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            double* rec = p.filt + (chain[c] * T + (t - 1)) * K8_REC;
            gyc[c] = gyn[c];
            miss[c] = obn[c] == 0.0;
            // two steps ahead: B′Q⁻¹y and the mask (first touch of those lines: HBM latency)
            gyn[c] = gyn2[c];
            obn[c] = obn2[c];
            gyn2[c] = p.filt[(chain[c] * T + clamp_t(t + 2)) * K8_REC + D16 + i];
            obn2[c] = p.obs[chain[c] * T + clamp_t(t + 2)];
            if (j == 0) rec[i] = xi_own[c];   // ξ_f(t − 1)
        }
        k8_inverse<NCH>(lam, lane, sb, lp, ok);   // C_{t−1} = (Λ_f + A′P⁻¹A)⁻¹
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            p.filt[(chain[c] * T + (t - 1)) * K8_REC + K8_HDR + lane] = lam[c];
            bufA[c][lane] = lam[c];
        }
        wave_lds_fence();
        double g[NCH], cx[NCH];   // G′ = K C;  C ξ_f — two accumulators each: the sums are chains of dependent FMAs
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            double g0 = 0.0, g1 = 0.0, c0 = 0.0, c1 = 0.0;
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                g0 = __builtin_fma(ki[k], bufA[c][8 * k + j], g0);
                g1 = __builtin_fma(ki[k + 1], bufA[c][8 * (k + 1) + j], g1);
                c0 = __builtin_fma(bufA[c][8 * i + k], xi[c][k], c0);
                c1 = __builtin_fma(bufA[c][8 * i + k + 1], xi[c][k + 1], c1);
            }
            g[c] = g0 + g1;
            cx[c] = c0 + c1;
            double* rec = p.filt + (chain[c] * T + (t - 1)) * K8_REC;
            rec[K8_HDR + 64 + lane] = g[c];
            if (j == 0) rec[2 * D16 + i] = cx[c];   // C_{t−1} ξ_f(t − 1): the backward sweep's constant term
            bufB[c][lane] = g[c];
        }
        wave_lds_fence();
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            double a0 = 0.0, a1 = 0.0, x0 = 0.0, x1 = 0.0;   // (G′K′)[i][j] = Σ_k G′[i][k] K[j][k];  ξ_p = G′ ξ_f
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                const double g0 = bufB[c][8 * i + k], g1 = bufB[c][8 * i + k + 1];
                a0 = __builtin_fma(g0, kj[k], a0);
                a1 = __builtin_fma(g1, kj[k + 1], a1);
                x0 = __builtin_fma(g0, xi[c][k], x0);
                x1 = __builtin_fma(g1, xi[c][k + 1], x1);
            }
            // M_{t+1} = Λ_f(t) + A′P⁻¹A = PLW − K C K′ (without B′Q⁻¹B where y_t is missing), exactly symmetric for the sweep operator
            lam[c] = k8_symmetrise((miss[c] ? plwm_ij : plw_ij) - (a0 + a1), i, j);
            xi_own[c] = gyc[c] + (x0 + x1);      // ξ_f(t)
            if (j == 0) vbuf[c][i] = xi_own[c];
        }
        wave_lds_fence();
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int k = 0; k < 8; ++k) xi[c][k] = vbuf[c][k];
        wave_lds_fence();   // the next step's stores to bufA / vbuf stay behind these reads
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        p.vend[(chain[c] * p.S) * C1::TRI + lane] = lam[c] - w_ij;   // Λ_f at the last time index (8×8, natural order: k8_backward reads it)
        if (j == 0) p.filt[(chain[c] * T + len) * K8_REC + i] = xi_own[c];
        if (FE && lane == 0) dense_fe_write(p, 1, chain[c], lp[c].value(), 0.0, 0.0);
        if (!ok[c] && lane == 0) atomicOr(p.status, ST_NOT_POSDEF);
    }
}

template <bool FE, int NCH>
__global__ void __launch_bounds__(64) k8_backward(DenseParams p) {
    constexpr int D16 = 16;
    using C1 = DenseCfg<1>;
    __shared__ __attribute__((aligned(16))) double lds[NCH * K8_LDS_PER_CHAIN];
    const int lane = threadIdx.x, i = lane >> 3, j = lane & 7;
    double *bufG[NCH], *bufV[NCH], *bufH[NCH], *vbuf[NCH], *sb[NCH];
    long long chain[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        double* base = lds + c * K8_LDS_PER_CHAIN;
        bufG[c] = base;                // G_t′
        bufV[c] = base + K8_MAT;       // V_s(t + 1)
        bufH[c] = base + 2 * K8_MAT;   // H = G V_s
        vbuf[c] = base + 3 * K8_MAT;   // m_s exchange
        sb[c] = base + 3 * K8_MAT + K8_VEC;
        chain[c] = (long long)blockIdx.x * NCH + c + p.chain0;
    }
    const DenseModel M = dense_model(p, chain[0]);
    const DenseCst cl = DenseCst::make(D16, p.dy);
    const double* cst = M.cst;
    const long long T = p.T, te = T - 1;
    const int dout = p.d_out;
    bool ok[NCH];
    LogProd lpe[NCH];
    auto store_posterior = [&](int c, long long t, double mi, double vij) {
        if (j == 0 && i < dout) p.mean[(t * p.n_chains + chain[c]) * dout + i] = mi;
        if (i < dout && j < dout) p.cov[(t * p.n_chains + chain[c]) * (size_t)dout * dout + i * dout + j] = vij;
    };
    // smoothed belief at the last time index: V_s = Λ_f(T−1)⁻¹ (no backward message behind it), m_s = V_s (ξ_f + ξβ)
    double v[NCH], ms[NCH][8];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        v[c] = p.vend[(chain[c] * p.S) * C1::TRI + lane];
        ok[c] = true;
    }
    k8_inverse<NCH>(v, lane, sb, lpe, ok);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        bufV[c][lane] = v[c];
        if (j == 0) vbuf[c][i] = p.filt[(chain[c] * T + te) * K8_REC + i] + p.beta_xi[(chain[c] * (p.S + 1) + 1) * D16 + i];
    }
    wave_lds_fence();
    double mi0[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        double mi = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) mi = __builtin_fma(bufV[c][8 * i + k], vbuf[c][k], mi);
        mi0[c] = mi;
    }
    wave_lds_fence();
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        if (j == 0) vbuf[c][i] = mi0[c];
    wave_lds_fence();
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
        for (int k = 0; k < 8; ++k) ms[c][k] = vbuf[c][k];
        store_posterior(c, te, mi0[c], v[c]);
    }
    wave_lds_fence();
    // the records of the two steps below travel under the current one (they come from HBM: one step of ≈ 0.7 µs does not cover the round trip)
    double gN[NCH][2], cN[NCH][2], cxN[NCH][2];
    auto prefetch = [&](int c, long long t, int slot) {
        const double* rec = p.filt + (chain[c] * T + (t >= 0 ? t : 0)) * K8_REC;
        cN[c][slot] = rec[K8_HDR + lane];
        gN[c][slot] = rec[K8_HDR + 64 + lane];
        cxN[c][slot] = rec[2 * D16 + i];
    };
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        prefetch(c, te - 1, 0);
        prefetch(c, te - 2, 1);
    }
    for (long long t = te - 1; t >= 0; --t) {
        double g[NCH], cc[NCH], cx[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            g[c] = gN[c][0]; cc[c] = cN[c][0]; cx[c] = cxN[c][0];
            gN[c][0] = gN[c][1]; cN[c][0] = cN[c][1]; cxN[c][0] = cxN[c][1];
            prefetch(c, t - 2, 1);
            bufG[c][lane] = g[c];          // bufV holds V_s(t + 1) (written at the end of the previous step)
        }
        wave_lds_fence();
        double mi[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {   // H = G V_s: G[i][k] = G′[k][i];  m_s(t) = C ξ_f + G m_s(t + 1)
            double h0 = 0.0, h1 = 0.0, m0 = cx[c], m1 = 0.0;
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                const double g0 = bufG[c][8 * k + i], g1 = bufG[c][8 * (k + 1) + i];
                h0 = __builtin_fma(g0, bufV[c][8 * k + j], h0);
                h1 = __builtin_fma(g1, bufV[c][8 * (k + 1) + j], h1);
                m0 = __builtin_fma(g0, ms[c][k], m0);
                m1 = __builtin_fma(g1, ms[c][k + 1], m1);
            }
            mi[c] = m0 + m1;
            bufH[c][lane] = h0 + h1;
            if (j == 0) vbuf[c][i] = mi[c];
        }
        wave_lds_fence();
#pragma unroll
        for (int c = 0; c < NCH; ++c) {   // V_s(t) = C + H G′
            double v0 = cc[c], v1 = 0.0;
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                v0 = __builtin_fma(bufH[c][8 * i + k], bufG[c][8 * k + j], v0);
                v1 = __builtin_fma(bufH[c][8 * i + k + 1], bufG[c][8 * (k + 1) + j], v1);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) ms[c][k] = vbuf[c][k];
            v[c] = k8_symmetrise(v0 + v1, i, j);
        }
        wave_lds_fence();          // every read of bufV / bufG / vbuf of this step is done
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            bufV[c][lane] = v[c];
            store_posterior(c, t, mi[c], v[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (FE && lane == 0) {   // the slot kd_backward_info writes for segment 0 of a one-segment chain (the residual forms are kd_fe_resid's)
            double f = lpe[c].value();                      // log|Λ_f(T)|
            f += 2.0 * cst[cl.oFEC];
            f -= ((double)p.T - p.nobs[chain[c]]) * cst[cl.oC0];   // dy log 2π + log|Q| of the observed time indices only
            dense_fe_write(p, 0, chain[c], f, 0.0, 0.0);
        }
        if (!ok[c] && lane == 0) atomicOr(p.status, ST_NOT_POSDEF);
    }
}

}  // namespace rxhip
