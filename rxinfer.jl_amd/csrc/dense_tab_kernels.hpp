// dense_tab_kernels.hpp — the per-model tables of the MFMA path, built ON THE DEVICE (round 3).
//
// What this replaces: build_dense_tables() of rxhip.hip — ≈55 ms of host Riccati recursions, Cholesky inverses and d×d products
// per never-seen d = 64 model, plus the upload of 130 MB of tables; the reference's counterpart is create_model, which is inside
// every published timing (benchmarks/…Benchmark.ipynb:186-196).  Same arithmetic, same table layouts (DenseCst, the
// aggregation maps Ψ_i | Θ_i, DenseParams::scanm / qtab / canon), but every d×d product is an MFMA contraction and every SPD
// inverse the panel inverse of dense_kernels.hpp, run by a handful of workgroups:
//   kt_consts   1 workgroup     the constant block (DenseCst): P⁻¹, Q⁻¹, K = P⁻¹A, W, PLW, the first filtered belief, the whitening
//                               maps L_P⁻¹, L_Q⁻¹ of the free-energy residuals (a right-looking Cholesky in LDS), transposed copies
//   kt_gains    1 workgroup     the known-start filter over one segment (L steps, sequential): K_i, U_i, Φ_i per offset and the
//                               segment elements (Π, C, J, C⁻¹, C⁻¹Π, J + Π'C⁻¹Π) of the full and of the last segment
//   kt_agg      2 workgroups    the aggregation maps Ψ_i = Φ_L⋯Φ_{i+1}K_i, Θ_i = U_i − Z_iK_i (backward recursion, one per segment length)
//   kt_scan     2 workgroups    the Riccati recursions of the boundary scan (prefix: covariance at every segment start; suffix:
//                               precision of the backward message at every segment end) UNTIL THEY CONVERGE — a time-invariant model
//                               reaches its fixed point after a few segments, and every later segment is the canonical one
//                               (DenseParams::canon): no copies, no further arithmetic
//   kt_qcanon   1 thread        which groups of the two-level scan repeat the previous group (integer logic on canon)
//   kt_qtab     ng × 2          the composed maps of the groups that do not
// All matrices are d×d with d = 16·NT (the model padded by the caller: identity blocks, B and Q padded to d rows with Q = I on the
// padding diagonal), row-major in a global workspace that stays in L2; operands are read straight from it in the MFMA operand
// layouts (these kernels are latency chains of a few hundred small products, ≈2–3 ms per model at d = 64 — not a roofline path).
// Restrictions: d ≥ 32 (smaller models are built on the host in well under a millisecond) and dy ≤ d.
#pragma once
#ifndef RXHIP_TAB_SAME_TOL
#define RXHIP_TAB_SAME_TOL 5e-15   // (launch_tables.hpp)
#endif
#include "dense_kernels.hpp"

namespace rxhip {

struct TabParams {
    int d, dy, ptt;
    long long T, L, Llast;
    int S, sg, ng;
    const double* in;   // padded inputs: A | P | V0 | B | Q | m0 (5 matrices d×d, then d doubles)
    double* ws;         // workspace (see TabWs)
    double* cst;        // DenseCst block
    double* tab;        // aggregation tables [2][L·dyp][2d]
    double* scanm;      // [S][6][d][d]
    double* qtab;       // [2][S][d][d]
    int* canon;         // [4][S]
    int* status;        // ST_NOT_POSDEF
    // kt_consts over several models at once (one workgroup per model, blockIdx.x): strides of `in`, `ws`, `cst` in doubles (0: one model)
    long long in_stride, ws_stride, cst_stride;
};
// workspace layout (doubles): NSLOT named d×d matrices, then K_i | U_i | Φ_i for every offset of a segment
struct TabWs {
    enum { QI = 0, PINV, G, LOBS, HF, V1, V1I, VF1, KC, WC, T1, T2, T3, T4, T5, T6, V, PI, J, VP, SI, KK, UU, PHI, HFPI,
           SPINV = T3, SWC = T4, SLOBS = T5,   // masked schedule (TabParams::S = 0): exactly symmetric copies, written when kt_consts is done with its scratch
           AG0 = 32,   // two element sets of 6 matrices each: Π, C, J, C⁻¹, C⁻¹Π, J + Π'C⁻¹Π
           PER0 = 48,  // per-program scratch of kt_agg (2 programs) and kt_scan (2): 8 matrices each
           NSLOT = 48 + 8 * 4 };
    static __host__ __device__ size_t doubles(int d, long long L) { return ((size_t)NSLOT + 3 * (size_t)L) * d * d; }
};

// The building blocks are NOT inlined: a phase kernel is a chain of a few hundred of them, and inlined the compiler schedules across
// all of it (512 registers and kilobytes of scratch per lane in the first version — scratch that size also makes the runtime
// re-provision the queue's scratch space).  One call per product costs nothing next to the product.  (A non-inlined block is compiled
// without the caller's launch bounds: 280–312 registers at d ≥ 48, one workgroup per CU — kernels that run hundreds of workgroups at once
// are fused instead: dense_mseg_kernels.hpp.)
// SYNC = false: no barrier behind the stores — for a product whose result the NEXT building block does not read (and whose destination
// nobody is still reading): its stores drain under the next block's loads
template <int NT, bool TA, bool TB, bool SYNC = true>
__device__ __attribute__((noinline)) void tab_mm(double* dst0, const double* a0, const double* b0, double alpha, const double* c0, double beta, int w, int lane) {
    constexpr int D = 16 * NT;
    // every operand lives in the global workspace: say so (behind a non-inlined call the pointers are of unknown origin, and flat loads
    // count on the LDS counter as well)
    double* dst = as_global(dst0);
    const double *a = as_global(a0), *b = as_global(b0), *c = c0 ? as_global(c0) : nullptr;
    // Every operand fragment is loaded BEFORE the first product (the generic mm_acc interleaves them and keeps ≈8 loads in flight:
    // ten L2 round trips per product), and the contraction index is permuted so that an operand read along k is one 16-byte load for
    // two k-steps (steps 2m, 2m + 1 of lane quarter kq ↔ k = 8m + 2kq, 8m + 2kq + 1 — both operands use the same map, so the sum is the same).
    typedef double v2d __attribute__((ext_vector_type(2)));
    constexpr int KS = D / 4;
    const int il = lane & 15, kq = lane >> 4, i = 16 * w + il;
    double av[KS], bv[NT][KS];
#pragma unroll
    for (int m = 0; m < KS / 2; ++m) {
        const int k = 8 * m + 2 * kq;
        if (TA) { av[2 * m] = a[k * D + i]; av[2 * m + 1] = a[(k + 1) * D + i]; }
        else { const v2d v = *reinterpret_cast<const v2d*>(a + i * D + k); av[2 * m] = v.x; av[2 * m + 1] = v.y; }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int j = 16 * t + il;
            if (TB) { const v2d v = *reinterpret_cast<const v2d*>(b + j * D + k); bv[t][2 * m] = v.x; bv[t][2 * m + 1] = v.y; }
            else { bv[t][2 * m] = b[k * D + j]; bv[t][2 * m + 1] = b[(k + 1) * D + j]; }
        }
    }
    double cv[NT][4];
    if (c) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) cv[t][r] = c[acc_row<NT>(w, lane, r) * D + acc_col<NT>(lane, t)];
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(av[s]));   // all of the above is in flight before anything is consumed
    d4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], bv[t][s], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ii = acc_row<NT>(w, lane, r), j = acc_col<NT>(lane, t);
            double v = alpha * acc[t][r];
            if (c) v += beta * cv[t][r];
            dst[ii * D + j] = v;
        }
    if (SYNC) __syncthreads();
}
// leading dimension of the LDS staging matrix of the fused kernels of dense_mseg_kernels.hpp
constexpr int tab_stage_ld(int NT) { return 16 * NT + 2; }
template <int NT>
__device__ __attribute__((noinline)) bool tab_inv(double* dst, const double* a, double* lds, double* logdet_out, int w, int lane) {
    constexpr int D = 16 * NT;
    Acc<NT> acc;
    acc_load<NT>(acc, a, D, w, lane);
    LogProd lp;
    const bool ok = blk_inverse<NT>(acc, lds, w, lane, lp);
    acc_store<NT>(acc, dst, D, w, lane);
    if (w == 0 && lane == 0) lds[blk_scratch_doubles(NT)] = lp.value();
    __syncthreads();
    if (logdet_out) *logdet_out = lds[blk_scratch_doubles(NT)];
    __syncthreads();
    return ok;
}
// dst = (alpha·½(a + a′) + gamma·c)⁻¹: the symmetrised sum is formed in the accumulator registers on the way in (one building block less
// in front of every inverse of the masked schedule)
template <int NT>
__device__ __attribute__((noinline)) bool tab_inv_symadd(double* dst0, double alpha, const double* a0, double gamma, const double* c0, const double* e0, double* lds, int w, int lane) {
    constexpr int D = 16 * NT;
    double* dst = as_global(dst0);
    const double *a = as_global(a0), *c = as_global(c0), *e3 = e0 ? as_global(e0) : nullptr;   // optional third term (+ sym(e))
    Acc<NT> acc;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = acc_row<NT>(w, lane, r), j = acc_col<NT>(lane, t);
            double v = alpha * 0.5 * (a[i * D + j] + a[j * D + i]) + gamma * 0.5 * (c[i * D + j] + c[j * D + i]);
            if (e3) v += 0.5 * (e3[i * D + j] + e3[j * D + i]);
            acc.v[t][r] = v;
        }
    LogProd lp;
    const bool ok = blk_inverse<NT>(acc, lds, w, lane, lp);
    acc_store<NT>(acc, dst, D, w, lane);
    __syncthreads();
    return ok;
}
template <int NT>
__device__ __attribute__((noinline)) void tab_lin(double* dst0, double alpha, const double* a0, double beta, const double* b0, bool tb, int tid) {
    constexpr int D = 16 * NT, MM = D * D, NTH = 64 * NT;
    double* dst = as_global(dst0);
    const double *a = a0 ? as_global(a0) : nullptr, *b = b0 ? as_global(b0) : nullptr;
    double v[MM / NTH];
#pragma unroll
    for (int u = 0; u < MM / NTH; ++u) {
        const int k = tid + u * NTH, i = k / D, j = k - i * D;
        double x = a ? alpha * a[k] : 0.0;
        if (b) x += beta * (tb ? b[j * D + i] : b[k]);
        v[u] = x;
    }
    __syncthreads();   // a transposed read of dst itself (symmetrisation in place) is complete before anything is written
#pragma unroll
    for (int u = 0; u < MM / NTH; ++u) dst[tid + u * NTH] = v[u];
    __syncthreads();
}

// dst = alpha·½(a + a') + gamma·c   (dst may alias a or c)
template <int NT>
__device__ __attribute__((noinline)) void tab_symadd(double* dst0, double alpha, const double* a0, double gamma, const double* c0, int tid) {
    constexpr int D = 16 * NT, MM = D * D, NTH = 64 * NT;
    double* dst = as_global(dst0);
    const double *a = as_global(a0), *c = as_global(c0);
    double v[MM / NTH];
#pragma unroll
    for (int u = 0; u < MM / NTH; ++u) {
        const int k = tid + u * NTH, i = k / D, j = k - i * D;
        v[u] = alpha * 0.5 * (a[k] + a[j * D + i]) + gamma * c[k];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < MM / NTH; ++u) dst[tid + u * NTH] = v[u];
    __syncthreads();
}

template <int NT>
struct TabOps {
    static constexpr int D = 16 * NT, MM = D * D, NTH = 64 * NT;
    int tid, w, lane;
    double* lds;   // ≥ blk_scratch_doubles(NT) + NTH doubles
    __device__ __forceinline__ void sync() const { __syncthreads(); }
    // dst = alpha·op(a)·op(b) + beta·c      (c may be null or dst; dst must differ from a and b)
    template <bool TA, bool TB, bool SYNC = true>
    __device__ __forceinline__ void mm(double* dst, const double* a, const double* b, double alpha = 1.0, const double* c = nullptr, double beta = 0.0) const {
        tab_mm<NT, TA, TB, SYNC>(dst, a, b, alpha, c, beta, w, lane);
    }
    // dst = (alpha·sym(a) + gamma·sym(c))⁻¹
    __device__ __forceinline__ bool inv_symadd(double* dst, double alpha, const double* a, double gamma, const double* c, const double* e = nullptr) const {
        return tab_inv_symadd<NT>(dst, alpha, a, gamma, c, e, lds, w, lane);
    }
    // dst = alpha·a + beta·op(b)   (elementwise; a, b may be null; dst may alias a, and b)
    __device__ __forceinline__ void lin(double* dst, double alpha, const double* a, double beta = 0.0, const double* b = nullptr, bool tb = false) const {
        tab_lin<NT>(dst, alpha, a, beta, b, tb, tid);
    }
    __device__ __forceinline__ void sym(double* dst, const double* a) const { lin(dst, 0.5, a, 0.5, a, true); }
    __device__ __forceinline__ void symadd(double* dst, double alpha, const double* a, double gamma, const double* c) const {
        tab_symadd<NT>(dst, alpha, a, gamma, c, tid);
    }
    __device__ __forceinline__ void eye(double* dst, double diag) const {
        for (int k = tid; k < MM; k += NTH) dst[k] = (k / D == k % D) ? diag : 0.0;
        sync();
    }
    // dst = a⁻¹ (SPD); *logdet (may be null) receives log det a (valid in every thread); returns false if a is not positive definite
    __device__ __forceinline__ bool inv(double* dst, const double* a, double* logdet) const {
        return tab_inv<NT>(dst, a, lds, logdet, w, lane);
    }
    // two iterates of a recursion over symmetric positive (semi)definite matrices agree to rounding: |a − b|_ij ≤ tol · sqrt(a_ii a_jj) for EVERY entry —
    // each on the scale of its own row and column (relative to the largest entry the test was blind to a slowly converging block whose units put it
    // decades below another one)   (uniform result)
    __device__ __forceinline__ bool same(const double* a, const double* b, double tol) const {
        constexpr int D = 16 * NT;
        double bad = 0.0;
        for (int k = tid; k < MM; k += NTH) {
            const int i = k / D, j = k - i * D;
            const double sc = sqrt(fabs(a[i * D + i] * a[j * D + j]));
            bad = fmax(bad, (fabs(a[k] - b[k]) <= tol * sc) ? 0.0 : 1.0);
        }
        double* red = lds;
        red[tid] = bad;
        sync();
        for (int n = NTH; n > 1;) {   // any thread count (192 threads at d = 48)
            const int h = (n + 1) / 2;
            if (tid < n - h) red[tid] = fmax(red[tid], red[tid + h]);
            sync();
            n = h;
        }
        const bool r = red[0] == 0.0;
        sync();
        return r;
    }
    // out[i·si + j·sj] = alpha·a[i][j] + beta·b[i][j]   for i < rows, j < cols
    __device__ __forceinline__ void put(double* out, long long si, long long sj, int rows, int cols, const double* a, double alpha = 1.0,
                                        const double* b = nullptr, double beta = 0.0) const {
        for (int k = tid; k < rows * cols; k += NTH) {
            const int i = k / cols, j = k - i * cols;
            double v = alpha * a[i * D + j];
            if (b) v += beta * b[i * D + j];
            out[i * si + j * sj] = v;
        }
    }
    // L⁻¹ of the Cholesky factor a = L L' (lower), in LDS: right-looking factorisation, then forward substitution by rows
    __device__ __forceinline__ bool chol_linv(double* dst, const double* a, double* Ls /* D×(D+1) LDS */) const {
        constexpr int LD = D + 1;
        for (int k = tid; k < MM; k += NTH) Ls[(k / D) * LD + (k % D)] = a[k];
        sync();
        bool ok = true;
        for (int j = 0; j < D; ++j) {
            const double pv = Ls[j * LD + j];
            ok = ok && pv > 0.0;
            const double l = sqrt(pv > 0.0 ? pv : 1.0), r = 1.0 / l;
            sync();
            if (tid >= j && tid < D) Ls[tid * LD + j] = tid == j ? l : Ls[tid * LD + j] * r;   // column j of L
            sync();
            for (int k = tid; k < (D - j - 1) * (D - j - 1); k += NTH) {   // trailing update (lower and upper: harmless)
                const int i = j + 1 + k / (D - j - 1), c = j + 1 + k % (D - j - 1);
                Ls[i * LD + c] -= Ls[i * LD + j] * Ls[c * LD + j];
            }
            sync();
        }
        // row i of L⁻¹: (e_i − Σ_{k<i} L[i][k]·Li[k][:]) / L[i][i]; thread j owns column j
        for (int i = 0; i < D; ++i) {
            if (tid < D) {
                double s = tid == i ? 1.0 : 0.0;
                for (int k = tid; k < i; ++k) s -= Ls[i * LD + k] * dst[k * D + tid];   // Li[k][j] = 0 for k < j
                dst[i * D + tid] = tid <= i ? s / Ls[i * LD + i] : 0.0;
            }
            sync();
        }
        return ok;
    }
};

// ------------------------------------------------------------------------------------------------------------------------
template <int NT>
__global__ void __launch_bounds__(64 * NT) kt_consts(TabParams p) {
    constexpr int D = 16 * NT, MM = D * D;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    TabOps<NT> o{(int)threadIdx.x, (int)threadIdx.x >> 6, (int)threadIdx.x & 63, smem};
    double* Ls = smem + blk_scratch_doubles(NT) + 2 * 64 * NT;   // Cholesky work matrix D×(D+1)
    const int tid = o.tid, dy = p.dy;
    const double* in = p.in + (size_t)blockIdx.x * (size_t)p.in_stride;   // this workgroup's model
    double* wsb = p.ws + (size_t)blockIdx.x * (size_t)p.ws_stride;
    const double *A = in, *P = in + MM, *V0 = in + 2 * MM, *B = in + 3 * MM, *Q = in + 4 * MM, *m0 = in + 5 * MM;
    auto W = [&](int slot) { return wsb + (size_t)slot * MM; };
    const DenseCst c = DenseCst::make(D, dy);
    double* cst = p.cst + (size_t)blockIdx.x * (size_t)p.cst_stride;
    bool ok = true;
    double ldQ, ldP, ldV1, ldLf;
    ok = o.inv(W(TabWs::QI), Q, &ldQ) && ok;            // Q⁻¹ (identity on the padding)
    ok = o.inv(W(TabWs::PINV), P, &ldP) && ok;          // P⁻¹
    o.template mm<true, false>(W(TabWs::G), B, W(TabWs::QI));            // G = B'Q⁻¹  [d × dy]
    o.template mm<false, false>(W(TabWs::LOBS), W(TabWs::G), B);         // B'Q⁻¹B
    o.template mm<false, false>(W(TabWs::HF), B, A);                     // B A  [dy × d]
    double* m1 = smem + blk_scratch_doubles(NT) + 64 * NT;   // [D] in LDS
    if (p.ptt) {
        if (tid < D) {
            double s = 0.0;
            for (int k = 0; k < D; ++k) s += A[tid * D + k] * m0[k];
            m1[tid] = s;
        }
        o.template mm<false, false>(W(TabWs::T1), A, V0);
        o.template mm<false, true>(W(TabWs::V1), W(TabWs::T1), A, 1.0, P, 1.0);   // A V0 A' + P
    } else {
        if (tid < D) m1[tid] = m0[tid];
        o.lin(W(TabWs::V1), 1.0, V0);
    }
    o.sync();
    ok = o.inv(W(TabWs::V1I), W(TabWs::V1), &ldV1) && ok;
    o.lin(W(TabWs::T1), 1.0, W(TabWs::V1I), 1.0, W(TabWs::LOBS));   // Λ_f(1)
    ok = o.inv(W(TabWs::VF1), W(TabWs::T1), &ldLf) && ok;
    o.template mm<false, false>(W(TabWs::KC), W(TabWs::PINV), A);           // K = P⁻¹A
    o.template mm<true, false>(W(TabWs::WC), A, W(TabWs::KC));              // W = A'P⁻¹A
    o.template mm<false, false>(W(TabWs::T2), W(TabWs::VF1), W(TabWs::G));  // K1 = V_f(1) G  [d × dy]
    // whitening maps of the free-energy residuals
    ok = o.chol_linv(W(TabWs::T3), P, Ls) && ok;                            // L_P⁻¹
    o.template mm<false, false>(W(TabWs::T4), W(TabWs::T3), A);             // L_P⁻¹A
    ok = o.chol_linv(W(TabWs::T5), Q, Ls) && ok;                            // L_Q⁻¹ (identity on the padding)
    o.template mm<false, false>(W(TabWs::T6), W(TabWs::T5), B);             // L_Q⁻¹B
    // ---- the constant block ----
    o.put(cst + c.oA, D, 1, D, D, A);
    {   // symmetrised: P, P⁻¹, W, PLW
        for (int k = tid; k < MM; k += 64 * NT) {
            const int i = k / D, j = k % D;
            const double ps = 0.5 * (P[k] + P[j * D + i]);
            const double pi = 0.5 * (W(TabWs::PINV)[k] + W(TabWs::PINV)[j * D + i]);
            const double wc = 0.5 * (W(TabWs::WC)[k] + W(TabWs::WC)[j * D + i]);
            const double lo = 0.5 * (W(TabWs::LOBS)[k] + W(TabWs::LOBS)[j * D + i]);
            cst[c.oP + k] = ps;
            cst[c.oPI + k] = pi;
            cst[c.oW + k] = wc;
            cst[c.oPLW + k] = pi + lo + wc;
            cst[c.oPLWM + k] = pi + wc;
        }
    }
    if (tid == 0) { cst[c.oLDP] = ldP; cst[c.oLDP + 1] = ldV1; }
    o.put(cst + c.oLOBS, D, 1, D, D, W(TabWs::LOBS));
    o.put(cst + c.oVF1, D, 1, D, D, W(TabWs::VF1));
    o.put(cst + c.oG, dy, 1, D, dy, W(TabWs::G));
    o.put(cst + c.oQI, dy, 1, dy, dy, W(TabWs::QI));
    o.put(cst + c.oHF, D, 1, dy, D, W(TabWs::HF));
    o.put(cst + c.oK1, dy, 1, D, dy, W(TabWs::T2));
    o.put(cst + c.oAT, 1, D, D, D, A);                       // A'
    o.put(cst + c.oGT, 1, D, D, dy, W(TabWs::G));            // G'   [dy][d]
    o.put(cst + c.oHFT, 1, dy, dy, D, W(TabWs::HF));         // (BA)' [d][dy]
    o.put(cst + c.oK1T, 1, D, D, dy, W(TabWs::T2));          // K1'  [dy][d]
    o.put(cst + c.oK, D, 1, D, D, W(TabWs::KC));
    o.put(cst + c.oKT, 1, D, D, D, W(TabWs::KC));
    o.put(cst + c.oV1I, D, 1, D, D, W(TabWs::V1I));
    o.put(cst + c.oBT, 1, dy, dy, D, B);                     // B'   [d][dy]
    o.put(cst + c.oLPX, 2 * D, 1, D, D, W(TabWs::T3));
    o.put(cst + c.oLPX + D, 2 * D, 1, D, D, W(TabWs::T4), -1.0);
    {
        const int dy4 = (dy + 3) & ~3, ky = dy4 + D, dyr = (dy + 15) / 16 * 16;
        for (int k = tid; k < dyr * ky; k += 64 * NT) cst[c.oLQX + k] = 0.0;
        o.sync();
        o.put(cst + c.oLQX, ky, 1, dy, dy, W(TabWs::T5));
        o.put(cst + c.oLQX + dy4, ky, 1, dy, D, W(TabWs::T6), -1.0);
    }
    o.sync();
    // vectors and scalars: X1 = V1⁻¹m1, S1 = m1'X1, C1 = V_f(1) X1
    double* x1 = smem + blk_scratch_doubles(NT);
    if (tid < D) {
        double s = 0.0;
        for (int k = 0; k < D; ++k) s += W(TabWs::V1I)[tid * D + k] * m1[k];
        x1[tid] = s;
        cst[c.oX1 + tid] = s;
        cst[c.oM1 + tid] = m1[tid];
    }
    o.sync();
    if (tid < D) {
        double s = 0.0;
        for (int k = 0; k < D; ++k) s += W(TabWs::VF1)[tid * D + k] * x1[k];
        cst[c.oC1 + tid] = s;
    }
    if (tid == 0) {
        double s1 = 0.0;
        for (int k = 0; k < D; ++k) s1 += x1[k] * m1[k];
        cst[c.oS1] = s1;
        cst[c.oC0] = dy * 1.8378770664093454835606594728112 + ldQ;
        cst[c.oLD1] = ldLf + ldV1;
        cst[c.oFEC] = 0.5 * (ldV1 + (double)(p.T - 1) * ldP + (double)p.T * (dy * 1.8378770664093454835606594728112 + ldQ));
    }
    if (p.S == 0) {   // constants of the masked schedule only (dense_mseg_kernels.hpp): its element pass reads P⁻¹, A′P⁻¹A, B′Q⁻¹B symmetrised, once
        o.sync();
        o.sym(W(TabWs::SPINV), W(TabWs::PINV));
        o.sym(W(TabWs::SWC), W(TabWs::WC));
        o.sym(W(TabWs::SLOBS), W(TabWs::LOBS));
    }
    if (!ok && tid == 0) atomicOr(p.status, ST_NOT_POSDEF);
}

// the known-start filter over one segment: gains and closed-loop maps of every offset, segment elements at L and at Llast
template <int NT>
__global__ void __launch_bounds__(64 * NT) kt_gains(TabParams p) {
    constexpr int D = 16 * NT, MM = D * D;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    TabOps<NT> o{(int)threadIdx.x, (int)threadIdx.x >> 6, (int)threadIdx.x & 63, smem};
    const double *A = p.in, *P = p.in + MM, *B = p.in + 3 * MM, *Q = p.in + 4 * MM;
    auto W = [&](int slot) { return p.ws + (size_t)slot * MM; };
    double* per = p.ws + (size_t)TabWs::NSLOT * MM;   // K_i | U_i | Φ_i
    const double* HF = W(TabWs::HF);
    double *V = W(TabWs::V), *Pi = W(TabWs::PI), *J = W(TabWs::J), *Vp = W(TabWs::VP), *Si = W(TabWs::SI), *HFPi = W(TabWs::HFPI);
    double *T1 = W(TabWs::T1), *T2 = W(TabWs::T2), *T3 = W(TabWs::T3);
    bool ok = true;
    o.eye(V, 0.0);
    o.eye(J, 0.0);
    o.eye(Pi, 1.0);
    for (long long i = 1; i <= p.L; ++i) {
        double* K = per + (size_t)(3 * (i - 1) + 0) * MM;
        double* U = per + (size_t)(3 * (i - 1) + 1) * MM;
        double* Phi = per + (size_t)(3 * (i - 1) + 2) * MM;
        o.template mm<false, false>(T1, A, V);
        o.template mm<false, true>(Vp, T1, A, 1.0, P, 1.0);              // V_p = A V A' + P
        o.template mm<false, false>(T1, B, Vp);                          // B V_p
        o.template mm<false, true>(T2, T1, B, 1.0, Q, 1.0);              // S = B V_p B' + Q
        ok = o.inv(Si, T2, nullptr) && ok;
        o.template mm<true, false>(K, T1, Si);                           // K = V_p B' S⁻¹
        o.template mm<false, false>(HFPi, HF, Pi);                       // (BA) Π
        o.template mm<true, false>(U, HFPi, Si);                         // U = ((BA)Π)' S⁻¹
        o.template mm<false, false>(J, U, HFPi, 1.0, J, 1.0);            // J += U (BA)Π
        o.template mm<false, false>(T2, K, T1, -1.0, Vp, 1.0);           // V_p − K B V_p
        o.sym(V, T2);
        o.template mm<false, false>(Phi, K, HF, -1.0, A, 1.0);           // Φ = A − K (BA)
        o.template mm<false, false>(T3, Phi, Pi);
        o.lin(Pi, 1.0, T3);
        for (int which = 0; which < 2; ++which) {
            if (i != (which == 0 ? p.L : p.Llast)) continue;
            double* g = W(TabWs::AG0 + 6 * which);   // Π, C, J, C⁻¹, X = C⁻¹Π, JJ = J + Π'X
            o.lin(g, 1.0, Pi);
            o.lin(g + MM, 1.0, V);
            o.sym(g + 2 * MM, J);
            ok = o.inv(g + 3 * MM, V, nullptr) && ok;
            o.template mm<false, false>(g + 4 * MM, g + 3 * MM, Pi);
            o.template mm<true, false>(g + 5 * MM, Pi, g + 4 * MM, 1.0, g + 2 * MM, 1.0);
        }
    }
    if (!ok && o.tid == 0) atomicOr(p.status, ST_NOT_POSDEF);
}

// aggregation maps of one segment length (blockIdx.x = 0: L, 1: Llast): backward recursion over the offsets
template <int NT>
__global__ void __launch_bounds__(64 * NT) kt_agg(TabParams p) {
    constexpr int D = 16 * NT, MM = D * D;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    TabOps<NT> o{(int)threadIdx.x, (int)threadIdx.x >> 6, (int)threadIdx.x & 63, smem};
    const int which = blockIdx.x, dy = p.dy, dyp = (dy + 3) & ~3;
    const long long Lx = which == 0 ? p.L : p.Llast;
    auto W = [&](int slot) { return p.ws + (size_t)slot * MM; };
    const double* per = p.ws + (size_t)TabWs::NSLOT * MM;
    const double* HF = W(TabWs::HF);
    double* mine = W(TabWs::PER0 + 8 * which);
    double *Pc = mine, *Z = mine + MM, *Psi = mine + 2 * MM, *ZK = mine + 3 * MM, *T = mine + 4 * MM, *UH = mine + 5 * MM;
    o.eye(Pc, 1.0);
    o.eye(Z, 0.0);
    for (long long i = Lx; i >= 1; --i) {
        const double* K = per + (size_t)(3 * (i - 1) + 0) * MM;
        const double* U = per + (size_t)(3 * (i - 1) + 1) * MM;
        const double* Phi = per + (size_t)(3 * (i - 1) + 2) * MM;
        o.template mm<false, false>(Psi, Pc, K);
        o.template mm<false, false>(ZK, Z, K);
        double* te = p.tab + ((size_t)which * p.L + (size_t)(i - 1)) * dyp * 2 * D;
        o.put(te, 1, 2 * D, D, dy, Psi);                       // te[j·2D + a] = Ψ[a][j]
        o.put(te + D, 1, 2 * D, D, dy, U, 1.0, ZK, -1.0);      // te[j·2D + D + a] = U[a][j] − (Z K)[a][j]
        o.template mm<false, false>(T, Pc, Phi);
        o.lin(Pc, 1.0, T);
        o.template mm<false, false>(UH, U, HF);
        o.template mm<false, false>(T, Z, Phi, 1.0, UH, 1.0);
        o.lin(Z, 1.0, T);
    }
}

// boundary-scan matrices: blockIdx.x = 0 prefix (covariance at every segment start, maps 0–2), 1 suffix (maps 3–5)
template <int NT>
__global__ void __launch_bounds__(64 * NT) kt_scan(TabParams p) {
    constexpr int D = 16 * NT, MM = D * D;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    TabOps<NT> o{(int)threadIdx.x, (int)threadIdx.x >> 6, (int)threadIdx.x & 63, smem};
    const int dir = blockIdx.x, S = p.S, tid = o.tid;
    auto W = [&](int slot) { return p.ws + (size_t)slot * MM; };
    double* mine = W(TabWs::PER0 + 8 * (2 + dir));
    double *Vi = mine, *Wm = mine + MM, *tt = mine + 2 * MM, *M1 = mine + 3 * MM, *M2 = mine + 4 * MM, *cur = mine + 5 * MM, *nxt = mine + 6 * MM;
    bool ok = true;
    if (S <= 0) return;
    if (dir == 0) {
        const double* g = W(TabWs::AG0);   // Π, C, J of the full segments
        o.lin(cur, 1.0, W(TabWs::VF1));
        bool conv = false;
        for (int s = 0; s < S; ++s) {
            double* sm = p.scanm + (size_t)s * 6 * MM;
            if (conv) {   // the maps of the previous segment: canonical index instead of a copy
                if (tid == 0) p.canon[s] = p.canon[s - 1];
                o.sync();
                continue;
            }
            if (tid == 0) p.canon[s] = s;
            o.lin(sm + 2 * MM, 1.0, cur);
            if (s == S - 1) break;
            ok = o.inv(Vi, cur, nullptr) && ok;
            o.lin(tt, 1.0, Vi, 1.0, g + 2 * MM);
            ok = o.inv(Wm, tt, nullptr) && ok;
            o.template mm<false, false>(M2, g, Wm);                 // M2 = Π W
            o.template mm<false, false>(M1, M2, Vi);                // M1 = M2 V⁻¹
            o.template mm<false, true>(tt, M2, g);                  // M2 Π'
            o.lin(sm, 1.0, nullptr, 1.0, M1, true);                 // maps stored transposed
            o.lin(sm + MM, 1.0, nullptr, 1.0, M2, true);
            o.lin(nxt, 0.5, tt, 0.5, tt, true);
            o.lin(nxt, 1.0, nxt, 1.0, g + MM);                      // sym(M2 Π') + C
            conv = o.same(nxt, cur, RXHIP_TAB_SAME_TOL);
            o.lin(cur, 1.0, nxt);
        }
    } else {
        o.eye(cur, 0.0);   // Λβ(b_S) = 0
        bool conv = false;
        for (int s = S - 1; s >= 1; --s) {
            const double* g = W(TabWs::AG0 + (s == S - 1 ? 6 : 0));   // C⁻¹ = g+3MM, X = g+4MM, JJ = g+5MM
            double* sm = p.scanm + (size_t)s * 6 * MM;
            if (conv) {
                if (tid == 0) p.canon[S + s] = p.canon[S + s + 1];
                o.sync();
                continue;
            }
            if (tid == 0) p.canon[S + s] = s;
            o.lin(sm + 5 * MM, 1.0, cur);
            o.lin(tt, 1.0, g + 3 * MM, 1.0, cur);
            ok = o.inv(Wm, tt, nullptr) && ok;
            o.template mm<true, false>(M1, g + 4 * MM, Wm);         // N1 = X'W
            o.template mm<false, false>(M2, M1, cur);               // N2 = N1 Λ
            o.template mm<false, false>(tt, M1, g + 4 * MM);        // N1 X
            o.lin(sm + 3 * MM, 1.0, nullptr, 1.0, M1, true);
            o.lin(sm + 4 * MM, 1.0, nullptr, 1.0, M2, true);
            o.lin(nxt, -0.5, tt, -0.5, tt, true);
            o.lin(nxt, 1.0, nxt, 1.0, g + 5 * MM);                  // JJ − sym(N1 X)
            conv = s < S - 1 && o.same(nxt, cur, RXHIP_TAB_SAME_TOL);            // the last segment has its own length: compare full-length steps only
            o.lin(cur, 1.0, nxt);
        }
        // segment 0: Λβ(b_1).  Its suffix maps 3, 4 are never read; slot 5 is (kd_prepare_bnd), so segment 0 is always its own canon
        o.lin(p.scanm + 5 * MM, 1.0, cur);
        if (tid == 0) p.canon[S + 0] = 0;
    }
    if (!ok && tid == 0) atomicOr(p.status, ST_NOT_POSDEF);
}

// canonical indices of the composed maps: a group whose step maps are, one by one, the canonical maps of the previous group's
// steps has the previous group's products.  (Sequential integer logic, one thread.)
static __global__ void kt_qcanon(TabParams p) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int S = p.S, n = S - 1, sg = p.sg;
    for (int dir = 0; dir < 2; ++dir) {
        int* qc = p.canon + (size_t)(2 + dir) * S;
        const int* mc = p.canon + (size_t)dir * S;
        for (int q = 0; q < S; ++q) qc[q] = q;
        for (int g0 = sg; g0 < n; g0 += sg) {
            const int g1 = g0 + sg < n ? g0 + sg : n;
            bool same = true;   // (a shorter last group composes fewer maps: its products are prefixes of the previous group's)
            for (int st = g0; st < g1 && same; ++st) {
                const int seg = dir ? S - 1 - st : st, pseg = dir ? S - 1 - (st - sg) : st - sg;
                same = mc[seg] == mc[pseg];
            }
            if (same)
                for (int st = g0; st < g1; ++st) qc[st + 1] = qc[st + 1 - sg];
        }
    }
}

// composed maps of one group of the two-level scan: Q_{st+1} = Map_st ⋯ Map_{group start}  (stored transposed, like the maps)
template <int NT>
__global__ void __launch_bounds__(64 * NT) kt_qtab(TabParams p) {
    constexpr int D = 16 * NT, MM = D * D;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    TabOps<NT> o{(int)threadIdx.x, (int)threadIdx.x >> 6, (int)threadIdx.x & 63, smem};
    const int grp = blockIdx.x, dir = blockIdx.y, S = p.S, n = S - 1, sg = p.sg;
    const int g0 = grp * sg, g1 = g0 + sg < n ? g0 + sg : n;
    if (g0 >= n) return;
    const int* qc = p.canon + (size_t)(2 + dir) * S;
    if (qc[g0 + 1] != g0 + 1) return;   // a repeat of an earlier group (whole groups repeat or none of their steps does)
    const int* mc = p.canon + (size_t)dir * S;
    double* qt = p.qtab + (size_t)dir * S * MM;
    for (int st = g0; st < g1; ++st) {
        const int seg = dir ? S - 1 - st : st;
        const double* mt = p.scanm + ((size_t)mc[seg] * 6 + (dir ? 3 : 0)) * MM;   // Map' (stored transposed)
        double* q = qt + (size_t)(st + 1) * MM;
        if (st == g0) o.lin(q, 1.0, mt);
        else o.template mm<false, false>(q, q - MM, mt);   // (Map Q)' = Q' Map'
    }
}

}  // namespace rxhip
