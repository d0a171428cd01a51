// dense_kernels.hpp — LGSSM message passing for LARGE state dimension (d = 16·NT ≤ 64,
// BASELINE config 3: d = dy = 64) on the fp64 matrix cores of gfx950.
//
// Same reference rules as lgssm_kernels.hpp (a3 MvNormalMeanCovariance, a4 typeof(*), a5 product,
// a6 marginal, a7 Bethe free energy) and the same exact parallel-in-time schedule, but the unit of
// execution is different: a d×d covariance no longer fits one lane, so ONE WORKGROUP (NT waves) owns one
// (chain, time-segment) and every d×d object is distributed over its threads in the accumulator layout
// of v_mfma_f64_16x16x4_f64:
//     wave w owns tile-row w (rows 16w..16w+15), all NT tile-columns;
//     lane l, tile t, register r  <->  element (row 16w + (l>>4) + 4r, col 16t + (l&15))
// (the f64 C/D map differs from the f32 one: cdna_hip_programming.md §3).
//   * dense contractions (A V A', V_f A' Λ_p, G D G', …) are MFMA: operands are read from LDS (or from
//     L2 for the constant matrices) in the A/B operand layout, 16 k-steps of 4, NT MFMAs per k-step;
//   * SPD inverses (cholinv in the reference) are an in-place Gauss–Jordan sweep directly on the
//     accumulator registers: per pivot the owners publish the pivot row and column through LDS, one
//     barrier, and every thread does a rank-1 update of its 4×NT elements; the pivots give logdet;
//   * vectors live in LDS; matvecs are row-per-thread dot products.
// Segment boundaries: the data-independent matrix parts of the boundary scan (covariance at every
// segment start, precision of the backward message at every segment end, and the d×d maps that carry
// the means) are per-model tables built at create time; the kernels carry only the data-dependent
// vectors across boundaries.  Inside a segment every step recomputes the full matrix algebra.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#include "lgssm_kernels.hpp"

namespace rxhip {

typedef double d4 __attribute__((ext_vector_type(4)));

// per-model constant block for the dense path (doubles; all matrices dense row-major)
struct DenseCst {
    int d, dy;
    long long oA, oP, oLOBS, oG, oQI, oHF, oC0, oX1, oS1, oLD1, oC1, oK1, oVF1, oAT, oGT, oHFT, oK1T, oPI, oK, oKT, oW, oV1I, oM1,
        oBT, oFEC, oPLW, oPLWM, oLDP, oLPX, oLQX, size;
    __host__ __device__ static DenseCst make(int d, int dy) {
        DenseCst c;
        c.d = d;
        c.dy = dy;
        long long o = 0;
        c.oA = o; o += (long long)d * d;
        c.oP = o; o += (long long)d * d;
        c.oLOBS = o; o += (long long)d * d;
        c.oG = o; o += (long long)d * dy;
        c.oQI = o; o += (long long)dy * dy;
        c.oHF = o; o += (long long)dy * d;
        c.oC0 = o; o += 1;   // dy log 2π + logdet Q
        c.oX1 = o; o += d;   // V1⁻¹ m1
        c.oS1 = o; o += 1;   // m1' V1⁻¹ m1
        c.oLD1 = o; o += 1;  // log(det Λf(1) · det V1)
        c.oC1 = o; o += d;   // Vf(1) V1⁻¹ m1
        c.oK1 = o; o += (long long)d * dy;  // Vf(1) G
        c.oVF1 = o; o += (long long)d * d;  // Vf(1)
        // transposed copies: out[i] = Σ_k M'[k][i] x[k] is coalesced over threads i
        c.oAT = o; o += (long long)d * d;    // A'   [d][d]
        c.oGT = o; o += (long long)dy * d;   // G'   [dy][d]
        c.oHFT = o; o += (long long)d * dy;  // (BA)' [d][dy]
        c.oK1T = o; o += (long long)dy * d;  // K1'  [dy][d]
        // information-form smoother (kd_forward_info / kd_backward_info)
        c.oPI = o; o += (long long)d * d;    // P⁻¹
        c.oK = o; o += (long long)d * d;     // K = P⁻¹A
        c.oKT = o; o += (long long)d * d;    // K'
        c.oW = o; o += (long long)d * d;     // W = A'P⁻¹A
        c.oV1I = o; o += (long long)d * d;   // V1⁻¹ (prior of the first state, through the transition if so modelled)
        c.oM1 = o; o += d;                   // m1
        c.oBT = o; o += (long long)d * dy;   // B'   [d][dy]
        c.oFEC = o; o += 1;                  // ½[log|V1| + (T−1) log|P| + T(dy log 2π + log|Q|)]
        c.oPLW = o; o += (long long)d * d;   // P⁻¹ + B'Q⁻¹B + A'P⁻¹A (symmetric): M_{t+1} = PLW − K G_t
        c.oPLWM = o; o += (long long)d * d;  // the same without B'Q⁻¹B: the step after a `missing` observation (masked sweeps)
        c.oLDP = o; o += 2;                  // log|P|, log|V1|: the free-energy constant of a chain with per-step constants is a sum over its steps
        // whitening maps of the free-energy residuals (kd_fe_resid_mfma): with P = L_P L_P', Q = L_Q L_Q'
        //   r_x'P⁻¹r_x = |L_P⁻¹ x̂_{t+1} − (L_P⁻¹A) x̂_t|²,   r_y'Q⁻¹r_y = |L_Q⁻¹ y_t − (L_Q⁻¹B) x̂_t|²
        o = (o + 1) & ~1LL;                                                              // 16-byte loads of both maps
        c.oLPX = o; o += (long long)d * 2 * d;                                           // [d][2d]  L_P⁻¹ | −L_P⁻¹A
        c.oLQX = o; o += (long long)(((dy + 15) / 16) * 16) * (((dy + 3) & ~3) + d);     // [dy↑16][dy↑4 + d]  L_Q⁻¹ | −L_Q⁻¹B
        c.size = (o + 7) / 8 * 8;
        return c;
    }
};

struct DenseParams {
    long long T, n_chains;
    int S;
    long long L;
    int d, dy;
    const double* y;      // [T][chain][dy]
    double* filt;         // [chain][T][REC]   m_f(t) | A m_f(t) | C_t = V_f − G_t A V_f (lower tiles) | G_t = V_f A' V_p(t+1)⁻¹ (smoother gain)
    int d_out;            // state dimension of the MODEL (≤ d): the kernels run on d = 16·NT with decoupled padding dimensions
                          // (A = 0, P = V0 = I, B = 0 there), only the leading d_out block of every posterior is written
    int filter;           // 1: filtering run (forward pass only; the filtered belief is written as the marginal)
    double* vend;         // [chain][S][TRI]   V_f at the last step of every segment (lower tiles)
    double* mean;         // [T][chain][d]
    double* cov;          // [T][chain][d][d]
    const double* cst;    // DenseCst block (model 0; the dense path takes one model)
    const double* tab;    // [2][L·dyp][2d]  aggregation tables Ψ_i | Θ_i (set 0: full segments, set 1: the last segment), dyp = dy padded to 4
    double* aggpart;      // [chain][agg_kc][S][2d]  partial sums of kd_agg_gemm
    int agg_oc, agg_kc;   // offsets per K-chunk, K-chunks
    long long Llast;      // length of the last segment
    const double* scanm;  // [S][6][d][d]   0:M1' 1:M2' 2:V(b_s)  3:N1' 4:N2' 5:Λβ(b_{s+1})  (maps stored transposed)
    double* elem;         // [chain][S][2][d]   b, η
    double* fstart_m;     // [chain][S][d]      filtered mean at b_s
    double* beta_xi;      // [chain][S+1][d]    ξβ at b_s
    // two-level boundary scan (see kd_scan_local / kd_scan_fix): scan steps st = 0 … S−2 in groups of `sg`
    const double* qtab;   // [2][S][d][d]  index q = st + 1: product of the step maps from the start of q's group through st (transposed)
    double* loc;          // [chain][2][S][d]  index q: state after step q − 1 of a scan that starts every group from zero
    const int* canon;     // [4][S] canonical indices of the boundary maps: a time-invariant model's Riccati recursions converge, and from
                          // then on the maps of a segment are bit-identical copies of the previous one's (build_dense_tables).  [0][s]: the
                          // first segment whose prefix maps (scanm 0–2) equal segment s's; [1][s]: suffix maps (3–5); [2][q] / [3][q]: the first q'
                          // with qtab[dir][q'] == qtab[dir][q].  The scan kernels address maps through it, so a converged stretch of the
                          // recursion keeps ONE 32 KB map in registers instead of streaming a new copy from HBM every round.  NULL: identity.
    double* bnd;          // [S][2][d][d]  data-independent boundary inverses (kd_prepare_bnd, once per engine): 0: Λ_f(b_s) = V(b_s)⁻¹;  1: V_s(b_{s+1}) = (Λ_f + Λβ)⁻¹ (s < S − 1)
    int sg, ng;           // group size, number of groups
    double* fe_part;      // [slot][user chain]
    int* status;
    // Packed layout (d ≤ 8, NT = 1): TWO chains of the batch share one 16×16 tile as a block-diagonal model — chain 2c in the
    // leading d_sub = 8 dimensions, chain 2c + 1 in the trailing ones (each padded to 8 with decoupled dimensions).  Block-
    // diagonal matrices stay block-diagonal under every operation of the sweep, the tile's zeros cost nothing extra, and the
    // observations / posteriors of the pair are adjacent in memory already.  pack = 1: one chain per tile row set (default).
    int pack, d_sub, dy_sub;
    // several models in one engine (chain c runs model chain_model[c]): per-model table pointers; NULL = the single model above
    const struct DenseModel* models;
    const int* chain_model;
    double* vlast;        // null, or [d][d]: V_s(T−1) of this run (the model pass of dense_split_kernels.hpp keeps it)
    // `missing` observations, parallel in time (dense_mseg_kernels.hpp): per-chain boundaries and the observation mask
    int mseg;
    const double* obs;    // [chain][T]  1: y[t] observed
    const double* nobs;   // [chain]     number of observed time indices
    const double* mbnd;   // [chain][S][2][d][d]  Λ_f(b_s) | V_s(b_{s+1})
    // per-step constants on the masked schedule (desc.step_model): time index t runs the constant block cst + step_model[t]·cst_stride;
    // fe_const[chain] = log|V1| + Σ_t log|P_t| + Σ_{t observed} (dy log 2π + log|Q_t|) (km_feconst); model_sel: the pass of the residual kernel
    const int* step_model;
    long long cst_stride;
    const double* fe_const;
    int model_sel;
    long long oW_off;     // DenseCst::oW (km_filter_out reads the constant block through MsegParams)
    long long chain0;     // first workgroup chain of this launch: a grid dimension holds 65 535 blocks, larger batches are launched in slices
    int wave8;            // masked schedule, one segment per chain, d ≤ 8: the sweep inside one wavefront per chain (dense8_kernels.hpp)
    int full_records;     // 1: the records of a frozen stretch carry their (repeated) matrices as well — the model pass of the model / data split, whose table
                          // kernel (kd_split_tables) reads C_t and G_t′ of EVERY time index
    int no_frozen;        // test hook RXHIP_NO_FROZEN=1: kd_forward_info / kd_backward_info take every step in full (no FROZEN / BFROZEN stretches)
};
struct DenseModel {
    const double *cst, *tab, *scanm, *qtab;
    double* bnd;
    const int* canon;
};
template <class T>
__device__ __forceinline__ T* as_global(T* ptr) {   // tells the compiler the address space (global_load instead of flat_load)
    const unsigned long long v = (unsigned long long)ptr;   // uniform values only: into scalar registers whatever loaded them
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    __attribute__((address_space(1))) T* g = (__attribute__((address_space(1))) T*)(((unsigned long long)hi << 32) | lo);
    asm volatile("" : "+s"(g));   // (without it the two casts cancel before anything can learn from them)
    return (T*)g;
}
template <class T>
__device__ __forceinline__ T* as_global_v(T* ptr) {   // the same for a pointer in vector registers (e.g. read back from LDS)
    __attribute__((address_space(1))) T* g = (__attribute__((address_space(1))) T*)ptr;
    asm volatile("" : "+v"(g));
    return (T*)g;
}
__device__ __forceinline__ DenseModel dense_model(const DenseParams& p, long long chain) {
    // field by field and through an empty asm: returning either struct makes the compiler select between two ADDRESSES — one of them inside the kernel-argument
    // segment — and read the fields through a vector load of it: a round trip to wherever the kernel arguments live in front of every
    // dense kernel (the free-energy residual kernel spent 7 of its 17 µs there, the last workgroups 15)
    DenseModel m{as_global(p.cst), as_global(p.tab), as_global(p.scanm), as_global(p.qtab), as_global(p.bnd), as_global(p.canon)};   // values now, not loads to be re-addressed
    if (p.models) {
        const DenseModel* q = p.models + p.chain_model[chain];
        m.cst = q->cst; m.tab = q->tab; m.scanm = q->scanm; m.qtab = q->qtab; m.bnd = q->bnd; m.canon = q->canon;
    }
    // ... and pointers to GLOBAL memory: a pointer of unknown origin is read with flat_load, which counts on the LDS counter too
    m.cst = as_global(m.cst); m.tab = as_global(m.tab); m.scanm = as_global(m.scanm); m.qtab = as_global(m.qtab);
    m.bnd = as_global(m.bnd); m.canon = as_global(m.canon);
    return m;
}
// Symmetric d×d results (M_{t+1} in the forward kernel, V_s(t) in the backward one, C_t in the records between them) are computed /
// stored once per unordered pair of tile indices: tile (w, t) belongs to wave w when (t − w) mod NT ≤ NT/2, the even-NT tie
// (distance NT/2) going to the lower half of the waves — 3, 3, 2, 2 tiles per wave at NT = 4.
__host__ __device__ inline bool dense_owned_tile(int NT, int w, int t) {
    int dist = t - w;
    if (dist < 0) dist += NT;
    const int NS = NT / 2 + 1, nsw = (NT % 2 == 0 && NT > 1 && w >= NT / 2) ? NS - 1 : NS;
    return dist < NS - 1 || (dist == NS - 1 && dist < nsw);
}
// matrix `slot` (0–5) of the boundary-scan tables of segment `seg`, through the canonical index of its direction (tables built
// on the device hold the maps of a converged recursion once; host-built tables hold copies, and the index is consistent with them)
__device__ __forceinline__ const double* scan_mat(const DenseModel& M, int S, long long seg, int slot, size_t MM) {
    const long long cs = M.canon ? M.canon[(slot >= 3 ? S : 0) + seg] : seg;
    return M.scanm + ((size_t)cs * 6 + slot) * MM;
}

// ---- posterior / free-energy output addressing (one chain per workgroup, or the packed pair) -------------------------
// p.n_chains counts what a workgroup owns (a chain, or a pair); the result arrays are indexed by USER chains.
__device__ __forceinline__ void dense_store_mean(const DenseParams& p, long long t, long long chain, int tid, double v) {
    if (p.pack == 2) {
        const int sub = tid >> 3, i = tid & 7;
        if (tid < 16 && i < p.d_out) p.mean[((t * p.n_chains + chain) * 2 + sub) * p.d_out + i] = v;
    } else if (tid < p.d_out)
        p.mean[(t * p.n_chains + chain) * p.d_out + tid] = v;
}
__device__ __forceinline__ double dense_load_mean(const DenseParams& p, long long t, long long chain, int j) {
    if (p.pack == 2) {
        const int sub = j >> 3, i = j & 7;
        return (j < 16 && i < p.d_out) ? p.mean[((t * p.n_chains + chain) * 2 + sub) * p.d_out + i] : 0.0;
    }
    return j < p.d_out ? p.mean[(t * p.n_chains + chain) * p.d_out + j] : 0.0;
}
// the same element without control flow: offset into p.mean (0 when the element does not exist) and whether it exists — for
// loops that want many such loads in flight (a branch per load makes the compiler wait for each before the next)
__device__ __forceinline__ long long dense_mean_offset(const DenseParams& p, long long t, long long chain, int j, bool& exists) {
    const bool pk = p.pack == 2;
    const int sub = j >> 3, i = pk ? (j & 7) : j;
    exists = pk ? (j < 16 && i < p.d_out) : (j < p.d_out);
    const long long row = pk ? (t * p.n_chains + chain) * 2 + sub : t * p.n_chains + chain;
    return exists ? row * p.d_out + i : 0;
}
// one free-energy partial of workgroup-chain `chain`: `indep` does not depend on the data (log-determinants, constants: both
// chains of a pair share the model, so each owns half of the pair's value), dep0 / dep1 are the data-dependent parts of the
// first / second chain of a pair (dep1 = 0 and dep0 = the whole when unpacked).  Stored negated, as every slot.
__device__ __forceinline__ void dense_fe_write(const DenseParams& p, long long slot, long long chain, double indep, double dep0, double dep1) {
    if (p.pack == 2) {
        p.fe_part[slot * (2 * p.n_chains) + 2 * chain] = -0.5 * (0.5 * indep + dep0);
        p.fe_part[slot * (2 * p.n_chains) + 2 * chain + 1] = -0.5 * (0.5 * indep + dep1);
    } else
        p.fe_part[slot * p.n_chains + chain] = -0.5 * (indep + dep0 + dep1);
}

template <int NT>
struct DenseCfg {
    static constexpr int D = 16 * NT;
    static constexpr int LD = D + 2;           // LDS leading dimension (doubles): A-operand reads conflict-free
    static constexpr int THREADS = 64 * NT;
    static constexpr int NTRI = NT * (NT + 1) / 2;
    static constexpr int TRI = NTRI * 256;           // a symmetric matrix as lower tiles in register order
    static constexpr int HDR = 3 * D;                // ξ_f(t) | B'Q⁻¹y_t (aggregation kernel) | C_t ξ_f(t) (forward kernel)
    static constexpr int REC = HDR + 2 * D * D;      // record of a time index: header | C_t | G_t'  (both in accumulator order)
    static constexpr int MAT = D * LD;          // doubles per LDS matrix
};

// ---- accumulator-layout helpers ------------------------------------------------------------------
template <int NT>
struct Acc {
    double v[NT][4];  // v[tile-col][reg]   (plain scalars: every index below is a compile-time constant)
};
template <int NT>
__device__ __forceinline__ int acc_row(int w, int lane, int r) { return 16 * w + (lane >> 4) + 4 * r; }
template <int NT>
__device__ __forceinline__ int acc_col(int lane, int t) { return 16 * t + (lane & 15); }

template <int NT>
__device__ __forceinline__ void acc_zero(Acc<NT>& a) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) a.v[t][r] = 0.0;
}
// row-major matrix (leading dimension ld) <-> accumulator layout
template <int NT>
__device__ __forceinline__ void acc_load(Acc<NT>& a, const double* M, int ld, int w, int lane) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) a.v[t][r] = M[acc_row<NT>(w, lane, r) * ld + acc_col<NT>(lane, t)];
}
template <int NT>
__device__ __forceinline__ void acc_store(const Acc<NT>& a, double* M, int ld, int w, int lane) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) M[acc_row<NT>(w, lane, r) * ld + acc_col<NT>(lane, t)] = a.v[t][r];
}
// a <- ½(a + M') with M = a copy of a in LDS: restores exact symmetry (the sweep inverse reads one triangle's worth of every
// pivot row; in a long recursion rounding-level asymmetry would otherwise be fed back and grow)
template <int NT>
__device__ __forceinline__ void acc_symmetrise(Acc<NT>& a, const double* M, int ld, int w, int lane) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) a.v[t][r] = 0.5 * (a.v[t][r] + M[acc_col<NT>(lane, t) * ld + acc_row<NT>(w, lane, r)]);
}
// transposed store: M[col][row] = a(row, col)
template <int NT>
__device__ __forceinline__ void acc_store_T(const Acc<NT>& a, double* M, int ld, int w, int lane) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) M[acc_col<NT>(lane, t) * ld + acc_row<NT>(w, lane, r)] = a.v[t][r];
}
// posterior store: the leading n×n block, row-major with leading dimension n (n = D unless the model was padded)
template <int NT>
__device__ __forceinline__ void acc_store_out(const Acc<NT>& a, double* M, int n, int w, int lane) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = acc_row<NT>(w, lane, r), j = acc_col<NT>(lane, t);
            if (i < n && j < n) M[i * n + j] = a.v[t][r];
        }
}
// posterior covariance of (t, chain): the leading d_out × d_out block, or — packed pairs — the two diagonal 8×8 blocks of the
// tile, each to its own chain
template <int NT>
__device__ __forceinline__ void dense_store_cov(const DenseParams& p, const Acc<NT>& a, long long t, long long chain, int w, int lane) {
    const size_t dd = (size_t)p.d_out * p.d_out;
    if (NT == 1 && p.pack == 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = acc_row<NT>(w, lane, r), j = acc_col<NT>(lane, 0);
            const int si = i >> 3, ii = i & 7, jj = j & 7;
            if (si == (j >> 3) && ii < p.d_out && jj < p.d_out) p.cov[((t * p.n_chains + chain) * 2 + si) * dd + (size_t)ii * p.d_out + jj] = a.v[0][r];
        }
    } else
        acc_store_out<NT>(a, p.cov + (t * p.n_chains + chain) * dd, p.d_out, w, lane);
}
template <int NT>
__device__ __forceinline__ void acc_add_mat(Acc<NT>& a, const double* M, int ld, int w, int lane, double sgn) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) a.v[t][r] += sgn * M[acc_row<NT>(w, lane, r) * ld + acc_col<NT>(lane, t)];
}
// lower-triangle tiles of a symmetric matrix in register order: [tile idx][r][lane] — every access of a
// wave is 512 contiguous bytes.  Tiles above the diagonal are reconstructed by symmetry through LDS.
template <int NT>
__device__ __forceinline__ void tri_load(Acc<NT>& a, const double* rec, int w, int lane) {  // tiles t ≤ w of this wave's tile row
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) a.v[t][r] = (t <= w) ? rec[(w * (w + 1) / 2 + t) * 256 + r * 64 + lane] : 0.0;
}
template <int NT>
__device__ __forceinline__ void tri_regs_to_lds(const Acc<NT>& a, double* M, int ld, int w, int lane) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (t <= w) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = acc_row<NT>(w, lane, r), j = acc_col<NT>(lane, t);
                M[i * ld + j] = a.v[t][r];
                if (t != w) M[j * ld + i] = a.v[t][r];
            }
        }
}
template <int NT>
__device__ __forceinline__ void acc_store_tri(const Acc<NT>& a, double* rec, int w, int lane) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (t <= w) {
            double* p = rec + (w * (w + 1) / 2 + t) * 256;
#pragma unroll
            for (int r = 0; r < 4; ++r) p[r * 64 + lane] = a.v[t][r];
        }
}
// load the lower tiles from a record into an LDS matrix (full symmetric, row-major ld)
template <int NT>
__device__ __forceinline__ void tri_to_lds(const double* rec, double* M, int ld, int w, int lane) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (t <= w) {
            const double* p = rec + (w * (w + 1) / 2 + t) * 256;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double v = p[r * 64 + lane];
                const int i = acc_row<NT>(w, lane, r), j = acc_col<NT>(lane, t);
                M[i * ld + j] = v;
                if (t != w) M[j * ld + i] = v;  // diagonal tiles are stored in full (deterministic: no double writer)
            }
        }
}

// a full matrix in accumulator register order: [wave][tile][r][lane] — every access of a wave is 512 contiguous bytes
template <int NT>
__device__ __forceinline__ void acc_store_full(const Acc<NT>& a, double* g, int w, int lane) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) g[((w * NT + t) * 4 + r) * 64 + lane] = a.v[t][r];
}
template <int NT>
__device__ __forceinline__ void acc_load_full(Acc<NT>& a, const double* g, int w, int lane) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) a.v[t][r] = g[((w * NT + t) * 4 + r) * 64 + lane];
}

// ---- MFMA contraction: acc += X·Y, X and Y addressed as X[i][k], Y[k][j] through element functors ----
// A operand of v_mfma_f64_16x16x4_f64: lane l holds X[i = l&15][k = l>>4]; B operand: Y[k = l>>4][j = l&15].
template <int NT, bool TX, bool TY>
__device__ __forceinline__ void mm_acc(Acc<NT>& c, const double* X, int ldx, const double* Y, int ldy, int w, int lane) {
    constexpr int D = 16 * NT;
    const int i = 16 * w + (lane & 15), kq = lane >> 4, jl = lane & 15;
    d4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (d4){c.v[t][0], c.v[t][1], c.v[t][2], c.v[t][3]};
#pragma unroll
    for (int kk = 0; kk < D / 4; ++kk) {
        const int k = 4 * kk + kq;
        const double a = TX ? X[k * ldx + i] : X[i * ldx + k];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int j = 16 * t + jl;
            const double b = TY ? Y[j * ldy + k] : Y[k * ldy + j];
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        c.v[t][0] = acc[t][0];
        c.v[t][1] = acc[t][1];
        c.v[t][2] = acc[t][2];
        c.v[t][3] = acc[t][3];
    }
}

// ---- SPD inverse, in place on the accumulator registers (restates FastCholesky.cholinv for large blocks) ----
// Block sweep operator, FOUR pivots per barrier.  The rows K = {16·pb + q + 4i, i = 0..3} of tile-row pb live in
// the same 16 lanes of wave pb (lane>>4 == q, registers i = 0..3), so one LDS publish delivers four pivot rows
// R (4 × D).  With D4 = R[:, K] (the 4×4 pivot block) the step
//     A_JJ <- A_JJ − R_J' D4⁻¹ R_J,     A_KJ <- D4⁻¹ R_J,     A_KK <- −D4⁻¹
// is a rank-4 update of every 16×16 tile: one v_mfma_f64_16x16x4_f64 per tile with suitably modified operands (see
// below) instead of ≈190 vector FMAs per thread; D4⁻¹ (SPD, via the LDL' inverse of lgssm_kernels.hpp) is recomputed
// redundantly by every thread, which is cheaper than a second barrier.  After the D/4 blocks the array holds −A⁻¹ (any pivot
// order is valid for SPD matrices).  det A = Π det D4.  16 barriers per 64×64 inverse instead of 64.
// rowbuf: 2 buffers × D columns × 4 rows (layout [col][4]: a thread fetches the four pivot-row values of a
// column with two ds_read_b128).
// 4×4 SPD inverse by cofactors (2×2 minors, Laplace expansion): every product is independent, so the dependent chain
// is ≈8 fp64 operations + one reciprocal instead of the four sequential pivots of an LDL' — this inverse sits on the
// critical path of every sweep round.  Positive definiteness = positivity of the four leading minors.
__device__ __forceinline__ bool spd_inv4_cof(const Sym<4>& A, Sym<4>& B, double& det) {
    const double a00 = A(0, 0), a10 = A(1, 0), a11 = A(1, 1), a20 = A(2, 0), a21 = A(2, 1), a22 = A(2, 2), a30 = A(3, 0),
                 a31 = A(3, 1), a32 = A(3, 2), a33 = A(3, 3);
    const double s0 = a00 * a11 - a10 * a10, s1 = a00 * a21 - a10 * a20, s2 = a00 * a31 - a10 * a30;
    const double s3 = a10 * a21 - a11 * a20, s4 = a10 * a31 - a11 * a30, s5 = a20 * a31 - a21 * a30;
    const double c5 = a22 * a33 - a32 * a32, c4 = a21 * a33 - a31 * a32, c3 = a21 * a32 - a31 * a22;
    const double c2 = a20 * a33 - a30 * a32, c1 = a20 * a32 - a30 * a22, c0 = a20 * a31 - a30 * a21;
    det = (s0 * c5 - s1 * c4 + s2 * c3) + (s3 * c2 - s4 * c1 + s5 * c0);
    const double m3 = a20 * s3 - a21 * s1 + a22 * s0;  // leading 3×3 minor
    const bool ok = (a00 > 0.0) && (s0 > 0.0) && (m3 > 0.0) && (det > 0.0);
    const double id = rcp_pos(det);
    B(0, 0) = (a11 * c5 - a21 * c4 + a31 * c3) * id;
    B(1, 0) = (-a10 * c5 + a21 * c2 - a31 * c1) * id;
    B(1, 1) = (a00 * c5 - a20 * c2 + a30 * c1) * id;
    B(2, 0) = (a10 * c4 - a11 * c2 + a31 * c0) * id;
    B(2, 1) = (-a00 * c4 + a10 * c2 - a30 * c0) * id;
    B(2, 2) = (a30 * s4 - a31 * s2 + a33 * s0) * id;
    B(3, 0) = (-a10 * c3 + a11 * c1 - a21 * c0) * id;
    B(3, 1) = (a00 * c3 - a10 * c1 + a20 * c0) * id;
    B(3, 2) = (-a30 * s3 + a31 * s1 - a32 * s0) * id;
    B(3, 3) = m3 * id;
    return ok;
}
// the same, but B = adj(A) = det·A⁻¹ (unscaled cofactors) and id = 1/det: a caller that needs one column scales four numbers
__device__ __forceinline__ bool spd_adj4_cof(const Sym<4>& A, Sym<4>& B, double& det, double& id) {
    const double a00 = A(0, 0), a10 = A(1, 0), a11 = A(1, 1), a20 = A(2, 0), a21 = A(2, 1), a22 = A(2, 2), a30 = A(3, 0),
                 a31 = A(3, 1), a32 = A(3, 2), a33 = A(3, 3);
    const double s0 = a00 * a11 - a10 * a10, s1 = a00 * a21 - a10 * a20, s2 = a00 * a31 - a10 * a30;
    const double s3 = a10 * a21 - a11 * a20, s4 = a10 * a31 - a11 * a30, s5 = a20 * a31 - a21 * a30;
    const double c5 = a22 * a33 - a32 * a32, c4 = a21 * a33 - a31 * a32, c3 = a21 * a32 - a31 * a22;
    const double c2 = a20 * a33 - a30 * a32, c1 = a20 * a32 - a30 * a22, c0 = a20 * a31 - a30 * a21;
    det = (s0 * c5 - s1 * c4 + s2 * c3) + (s3 * c2 - s4 * c1 + s5 * c0);
    const double m3 = a20 * s3 - a21 * s1 + a22 * s0;  // leading 3×3 minor
    const bool ok = (a00 > 0.0) && (s0 > 0.0) && (m3 > 0.0) && (det > 0.0);
    id = rcp_pos(det);
    B(0, 0) = (a11 * c5 - a21 * c4 + a31 * c3);
    B(1, 0) = (-a10 * c5 + a21 * c2 - a31 * c1);
    B(1, 1) = (a00 * c5 - a20 * c2 + a30 * c1);
    B(2, 0) = (a10 * c4 - a11 * c2 + a31 * c0);
    B(2, 1) = (-a00 * c4 + a10 * c2 - a30 * c0);
    B(2, 2) = (a30 * s4 - a31 * s2 + a33 * s0);
    B(3, 0) = (-a10 * c3 + a11 * c1 - a21 * c0);
    B(3, 1) = (a00 * c3 - a10 * c1 + a20 * c0);
    B(3, 2) = (-a30 * s3 + a31 * s1 - a32 * s0);
    B(3, 3) = m3;
    return ok;
}

// ---- SPD inverse, blocked: 16×16 panels (round 3) ---------------------------------------------------------------------
// The rank-4 sweep above pays, per 64×64 inverse, 16 workgroup barriers and 16 redundant 4×4 adjugates in all 256 threads
// (≈3 150 VALU instructions per wave against 64 MFMAs: the dependent chain publish → barrier → adjugate → MFMA bounds the
// forward step, not either pipe).  blk_inverse sweeps one 16×16 PANEL per workgroup barrier:
//   block step k (pivot block row k = wave k's registers):
//     wave k      publishes its tile row R = A[k, :] as it sits in its registers ([tile][reg][lane]: conflict-free, contiguous),
//                 inverts the diagonal tile D = A[k, k] INSIDE THE WAVE (diag_inverse16: four rank-4 rounds through 512 bytes of
//                 wave-private LDS, no workgroup barrier; one column of the 4×4 pivot inverse per lane by Cramer's rule on
//                 rotated columns — ≈45 flops per lane and round instead of ≈130), publishes −D⁻¹                 [barrier]
//     wave k      A[k, t] ← D⁻¹ R_t straight from its own registers (no LDS read), A[k, k] ← 2I − D⁻¹
//     wave w ≠ k  Z̃ = −D⁻¹ R_w (4 MFMAs); trailing tiles A[w, t] += Z̃' R_t (4 MFMAs each); column tile A[w, k] ← R_w' D⁻¹
//                 (the same two operands swapped — no cancellation against the old tile)
//   The accumulator layout of v_mfma_f64_16x16x4_f64 (lane (j, q), register r ↔ row q + 4r, column j) IS the B-operand layout
//   of a k-step that runs over rows {q + 4r : q} in register order, and — for the transposed factor — the A-operand layout:
//   Z̃ leaves the matrix pipe in exactly the registers the trailing products read it from, and R / D⁻¹ are read back from LDS
//   at [reg][lane], the address they were written to.  No transposition, no bank conflict, 4 barriers per 64×64 inverse.
// Accuracy: the rank-4 rounds inside a diagonal tile use the modified-operand form of the round-1 sweep (pivot rows and columns come out
// of the same MFMA as the trailing update through ±1 entries), whose error is ε·|D|² relative — harmless for |D| = O(1), not
// for precisions of 10⁶.  The matrix is therefore equilibrated first, EXACTLY: a_ij ← a_ij·2^(h_i + h_j), h_i = −⌊exponent(a_ii)/2⌋
// (powers of two: no rounding), so every diagonal entry — and with it every entry of every Schur complement — is below 2; the
// result is scaled back the same way.  scripts/sim_blk_inverse.py models every lane, register and LDS slot of this routine
// on the CPU (12 decades between diagonal entries: 1.5e-14 element-wise).
// LDS: blk_scratch_doubles(NT) doubles (21 KB at NT = 4), double-buffered over block steps.
__host__ __device__ constexpr int blk_half_doubles(int NT) { return (4 * NT + 4) * 64 + 8; }
__host__ __device__ constexpr int blk_scratch_doubles(int NT) { return 2 * blk_half_doubles(NT) + 16 * NT; }

// wave-local LDS exchange point: LDS instructions of one wavefront complete in issue order; this only keeps the compiler from
// moving the reads above the writes
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

// One rank-4 round of the in-wave inverse of a 16×16 tile t[r] ↔ D[q + 4r][j] (lane = 16q + j).  Pivot rows K_u = Q + 4u sit in the
// 16 lanes with q == Q.  sb: 64 doubles, [column j][u].
template <int Q>
__device__ __forceinline__ void diag_round(double (&t)[4], double* sb, int j, int q, const double (&eu)[4], double euq, bool& bad, double& detprod) {
    typedef double v4d __attribute__((ext_vector_type(4)));
    if (q == Q) {
        reinterpret_cast<double2*>(sb + 4 * j)[0] = make_double2(t[0], t[1]);
        reinterpret_cast<double2*>(sb + 4 * j)[1] = make_double2(t[2], t[3]);
    }
    wave_lds_fence();
    double ri[4];
    {
        const double2 x0 = reinterpret_cast<const double2*>(sb + 4 * j)[0], x1 = reinterpret_cast<const double2*>(sb + 4 * j)[1];
        ri[0] = x0.x; ri[1] = x0.y; ri[2] = x1.x; ri[3] = x1.y;
    }
    double yb = sb[4 * j + q];
    // pivot block D4[u][c] = R[u][Q + 4c], columns rotated so that this lane group's own column c = q comes first (the address
    // is uniform over the 16 lanes of a group: broadcast reads)
    double col[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const double* src = sb + 4 * (Q + 4 * ((q + m) & 3));
        const double2 x0 = reinterpret_cast<const double2*>(src)[0], x1 = reinterpret_cast<const double2*>(src)[1];
        col[m][0] = x0.x; col[m][1] = x0.y; col[m][2] = x1.x; col[m][3] = x1.y;
    }
    wave_lds_fence();  // the next round's publish must not overtake these reads
    // R̃ = R − E_K: the ±1 entries that make pivot rows, pivot columns and the pivot block fall out of the one MFMA below
    const double fQ = (j & 3) == Q ? 1.0 : 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u) ri[u] = __builtin_fma(-fQ, eu[u], ri[u]);
    yb = __builtin_fma(-fQ, euq, yb);
    const double(&a)[4] = col[0];
    const double(&b)[4] = col[1];
    const double(&c)[4] = col[2];
    const double(&d)[4] = col[3];
    const double m01 = c[0] * d[1] - c[1] * d[0], m02 = c[0] * d[2] - c[2] * d[0], m03 = c[0] * d[3] - c[3] * d[0];
    const double m12 = c[1] * d[2] - c[2] * d[1], m13 = c[1] * d[3] - c[3] * d[1], m23 = c[2] * d[3] - c[3] * d[2];
    const double cof0 = (b[1] * m23 - b[2] * m13) + b[3] * m12;
    const double cof1 = (b[2] * m03 - b[0] * m23) - b[3] * m02;
    const double cof2 = (b[0] * m13 - b[1] * m03) + b[3] * m01;
    const double cof3 = (b[1] * m02 - b[0] * m12) - b[2] * m01;
    const double det = (a[0] * cof0 + a[1] * cof1) + (a[2] * cof2 + a[3] * cof3);   // ± det D4 (sign of the rotation; q = 0: +)
    const double num = (ri[0] * cof0 + ri[1] * cof1) + (ri[2] * cof2 + ri[3] * cof3);
    // Cramer: component q of the solution of D4 x = −r̃_j; the rotation's sign is common to numerator and denominator
    double rc = __builtin_amdgcn_rcp(det);
    double e = __builtin_fma(-det, rc, 1.0);
    rc = __builtin_fma(rc, e, rc);
    e = __builtin_fma(-det, rc, 1.0);
    rc = __builtin_fma(rc, e, rc);
    const double xa = -num * rc;
    // positive definiteness of the pivot block: nested trailing principal minors, in the lanes that see the natural column order
    const bool good = (d[3] > 0.0) & (m23 > 0.0) & (cof0 > 0.0) & (det > 0.0);   // bitwise: no short-circuit branches
    bad = bad | ((q == 0) & !good);
    detprod *= det;  // meaningful in the lanes with q == 0
    v4d acc = {t[0], t[1], t[2], t[3]};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, yb, acc, 0, 0, 0);
    t[0] = acc[0]; t[1] = acc[1]; t[2] = acc[2]; t[3] = acc[3];
}
// t ← D⁻¹ (in place, accumulator layout); detprod = det D in the lanes with q == 0.  Wave-private scratch sb (64 doubles).
__device__ __forceinline__ void diag_inverse16(double (&t)[4], double* sb, int lane, bool& bad, double& detprod) {
    const int j = lane & 15, q = lane >> 4, ju = j >> 2;
    const double eu[4] = {ju == 0 ? 1.0 : 0.0, ju == 1 ? 1.0 : 0.0, ju == 2 ? 1.0 : 0.0, ju == 3 ? 1.0 : 0.0};
    const double euq = ju == q ? 1.0 : 0.0;
    detprod = 1.0;
    diag_round<0>(t, sb, j, q, eu, euq, bad, detprod);
    diag_round<1>(t, sb, j, q, eu, euq, bad, detprod);
    diag_round<2>(t, sb, j, q, eu, euq, bad, detprod);
    diag_round<3>(t, sb, j, q, eu, euq, bad, detprod);
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = (j == q + 4 * r ? 2.0 : 0.0) - t[r];
}

struct NoPrefetch { __device__ __forceinline__ void operator()() const {} };
// Seeding the in-wave tile inverse with the one of the previous TIME step (kd_forward_info, round 4).  Along a chain the pivot tile D_k(t) of
// block step k differs from D_k(t − 1) by the Riccati recursion's contraction — by rounding once the recursion has converged, which at the
// segment starts of a long chain is everywhere but in its first segments.  With X₀ = D_k(t − 1)⁻¹ in hand one Newton – Schulz step
//     W = D X₀,   X₁ = X₀ + X₀′(I − W)
// is two chains of four MFMAs straight on the accumulator registers (an MFMA whose A operand is a tile in accumulator layout multiplies by the
// tile's TRANSPOSE, hence X₀′: written this way the antisymmetric part of X₀ cancels instead of doubling — X₀ + X₀′ − X₀′DX₀ is symmetric to
// rounding whatever X₀ was) in place of four rank-4 rounds with an LDS round trip, a 4×4 Cramer solve and a reciprocal each (≈ 2 300 → 650 cycles,
// of a wave the other three wait for).  Taken when every |W − I| entry is below 10⁻⁸ (the result is then exact to 10⁻¹⁴; else the exact rounds
// run and reseed).  The determinant follows from log det D(t) = log det D(t − 1) + tr(W − I) − O(‖W − I‖²): the trace is kept as a per-LANE
// partial sum, added into a per-lane running total once per step, and reduced over lanes ONCE at the end of the segment.
struct NoSeed { static constexpr bool ON = false; };
struct DiagSeed {
    static constexpr bool ON = true;
#ifdef RXHIP_TEST_SEEDCOUNT
    static constexpr int LDS_DOUBLES = 280;
#else
    static constexpr int LDS_DOUBLES = 264;   // per wave: the tile [4][64], the determinant at the last exact inversion
#endif
    double* x0 = nullptr;
    bool valid = false;     // wave-uniform: x0 holds the previous step's inverse
    bool use = false;       // attempt at all (not on masked or per-step-constant sweeps: every step's tile is another matrix)
    double r = 0.0, a = 0.0;   // lane partials: Σ tr(W − I) since the last exact inversion; Σ over steps of r
};
// `prefetch` is called once, before the last block step: loads the caller needs right after the inverse travel under it
// The equilibration exponents of a matrix, from its diagonal tile `dt` (tile (w, w) in accumulator layout): written by the 16
// lanes that hold a diagonal element.  blk_inverse does this itself (PUB) — or the caller does, where it has the tile in
// registers anyway and a workgroup barrier to share (kd_forward_info: the barrier in front of the symmetrisation).
template <int NT>
__device__ __forceinline__ void blk_publish_exponents(const double (&dt)[4], double* scratch, int w, int lane) {
    int* sc = reinterpret_cast<int*>(scratch + 2 * blk_half_doubles(NT));
    const int j = lane & 15, q = lane >> 4;
    double dv = 1.0;
    bool own = false;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (j == q + 4 * r) { dv = dt[r]; own = true; }
    if (own) {
        const bool pos = dv > 0.0 && is_finite(dv);
        const int h = pos ? -(__builtin_amdgcn_frexp_exp(dv) >> 1) : 0x40000000;
        sc[16 * w + j] = h;
        sc[16 * NT + 16 * w + 4 * (j & 3) + (j >> 2)] = h;
    }
}
// Sum over the 16 lanes of a DPP row, in every lane of the row: four DPP stages (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror,
// row_mirror — the reduction of hgf_kernels.hpp) instead of four dependent ds_bpermute round trips with a wait behind each.
template <int CTRL>
__device__ __forceinline__ int dpp_row_mov(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ int row16_sum(int v) {
    v += dpp_row_mov<0xB1>(v);
    v += dpp_row_mov<0x4E>(v);
    v += dpp_row_mov<0x141>(v);
    v += dpp_row_mov<0x140>(v);
    return v;
}
template <int CTRL>
__device__ __forceinline__ double dpp_row_mov(double v) {
    return __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true), __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ double rd_lane(double v, int l) {   // the value of lane l, in every lane
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double row16_sum(double v) {
    v += dpp_row_mov<0xB1>(v);
    v += dpp_row_mov<0x4E>(v);
    v += dpp_row_mov<0x141>(v);
    v += dpp_row_mov<0x140>(v);
    return v;
}
// FINAL = false: no barrier at the end — the caller must pass a workgroup barrier before anything else writes to `scratch`
// PUB = false: the exponents are in the scratch already (blk_publish_exponents + a workgroup barrier by the caller)
template <int NT, class PF = NoPrefetch, bool FINAL = true, bool PUB = true, class SEED = NoSeed>
__device__ __forceinline__ bool blk_inverse(Acc<NT>& a, double* scratch, int w_, int lane_, LogProd& lp, PF prefetch = PF(), SEED* seed = nullptr) {
    typedef double v4d __attribute__((ext_vector_type(4)));
    constexpr int HALF = blk_half_doubles(NT);
    // the wave index as a scalar (uniform branches below); the lane index laundered, so that the lane constants and LDS addresses
    // derived from it are recomputed per call instead of being hoisted out of the caller's time loop into (spilled) registers
    const int w = __builtin_amdgcn_readfirstlane(w_);
    int lane = lane_;
    asm volatile("" : "+v"(lane));
    const int j = lane & 15, q = lane >> 4;
    int* sc = reinterpret_cast<int*>(scratch + 2 * HALF);  // [2][16 NT]: exponents in natural order | permuted (row q + 4r at 4q + r)
    bool bad = false;
    // ---- exact equilibration by powers of two ----
    if (PUB) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (t == w) blk_publish_exponents<NT>(a.v[t], scratch, w, lane);
        lds_barrier();
    }
    int hr[4], hc[NT];
    {
        const int4 x = *reinterpret_cast<const int4*>(sc + 16 * NT + 16 * w + 4 * q);
        hr[0] = x.x; hr[1] = x.y; hr[2] = x.z; hr[3] = x.w;
        int hsum = 0;
        bool inval = false;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            hc[t] = sc[16 * t + j];
            inval = inval || hc[t] == 0x40000000;
            hc[t] = hc[t] == 0x40000000 ? 0 : hc[t];
            hsum += hc[t];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) hr[r] = hr[r] == 0x40000000 ? 0 : hr[r];
        bad = __any(inval);  // every wave sees all 16 NT exponents: uniform over the workgroup
        if (w == 0) {        // Σ h over the whole diagonal: det A = det A_s · 2^(−2 Σ h)
            hsum = row16_sum(hsum);
            if (lane == 0) lp.expo -= 2 * (long long)hsum;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) a.v[t][r] = __builtin_ldexp(a.v[t][r], hr[r] + hc[t]);
    }
#ifdef RXHIP_TEST_SEEDCOUNT
    long long tb_ = __builtin_readcyclecounter();
#define RXHIP_BPH(i) do { if constexpr (SEED::ON) { const long long tn_ = __builtin_readcyclecounter(); if (lane == 0) seed->x0[272 + (i)] += (double)(tn_ - tb_); tb_ = tn_; } } while (0)
#else
#define RXHIP_BPH(i) do { } while (0)
#endif
    // ---- block steps ----
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        RXHIP_BPH(k == 0 ? 0 : 3);
        double* hb = scratch + (k & 1) * HALF;  // R [NT][4][64] | −D⁻¹ [4][64] (first the wave-private scratch of the tile inverse) | det, flag
        double* nd = hb + NT * 256;
        double di[4];
        if (k == NT - 1) prefetch();
        if (w == k) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) hb[(t * 4 + r) * 64 + lane] = a.v[t][r];
#pragma unroll
            for (int r = 0; r < 4; ++r) di[r] = a.v[k][r];
            double detp;
            bool badk = false;
            // (A first attempt at seeding — X ← X(2I − DX) with a re-symmetrisation chain and a wave-wide reduction for the determinant per tile —
            // cost as much as the rounds it replaced: 0.433 against 0.425 ms.  The form below has neither.  DESIGN §6e.)
            if constexpr (SEED::ON) {
                bool done = false;
#ifdef RXHIP_TEST_SEEDCOUNT
                const long long tc0 = __builtin_readcyclecounter();
#endif
                if (__builtin_expect(seed->valid && seed->use, 1)) {   // (the hot path: the exact rounds below are laid out of line)
                    double x0[4], e[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) x0[r] = seed->x0[r * 64 + lane];
                    v4d wv = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int r = 0; r < 4; ++r) wv = __builtin_amdgcn_mfma_f64_16x16x4f64(di[r], x0[r], wv, 0, 0, 0);   // W = D′X₀
                    bool big = false;
                    double trp = 0.0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool dg = j == q + 4 * r;
                        e[r] = wv[r] - (dg ? 1.0 : 0.0);
                        big = big | !(__builtin_fabs(e[r]) <= 1e-8);   // (NaN: big)
                        trp += dg ? e[r] : 0.0;
                    }
                    if (__builtin_expect(!__any(big), 1)) {
                        v4d v = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int r = 0; r < 4; ++r) v = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[r], -e[r], v, 0, 0, 0);   // X₀′(I − W)
#pragma unroll
                        for (int r = 0; r < 4; ++r) di[r] = x0[r] + v[r];
                        seed->r += trp;
                        detp = seed->x0[256];
                        done = true;
#ifdef RXHIP_TEST_SEEDCOUNT
                        seed->x0[257] += 1.0;
#endif
                    }
                }
                if (__builtin_expect(!done, 0)) {
                    diag_inverse16(di, nd, lane, badk, detp);
                    seed->r = 0.0;
                    seed->valid = true;
                    if (lane == 0) seed->x0[256] = detp;
                }
                seed->a += seed->r;
#pragma unroll
                for (int r = 0; r < 4; ++r) seed->x0[r * 64 + lane] = di[r];
#ifdef RXHIP_TEST_SEEDCOUNT
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) seed->x0[done ? 258 : 259] += (double)(__builtin_readcyclecounter() - tc0);
#endif
            } else {
                diag_inverse16(di, nd, lane, badk, detp);
            }
            badk = __any(badk);
#pragma unroll
            for (int r = 0; r < 4; ++r) nd[r * 64 + lane] = -di[r];
            if (lane == 0) {
                nd[256] = detp;
                nd[257] = badk ? 1.0 : 0.0;
            }
        }
        RXHIP_BPH(1);
        lds_barrier();
        RXHIP_BPH(2);
        if (nd[257] != 0.0) bad = true;
        if (w == 0 && lane == 0) lp.mul(nd[256]);
        if (w == k) {
            // own tile row from registers: A[k, t] ← D⁻¹ R_t;  pivot block ← −D⁻¹
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t == k) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) a.v[t][r] = -di[r];
                } else {
                    v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(di[r], a.v[t][r], acc, 0, 0, 0);
                    a.v[t][0] = acc[0]; a.v[t][1] = acc[1]; a.v[t][2] = acc[2]; a.v[t][3] = acc[3];
                }
            }
        } else {
            double ndr[4], rw[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) ndr[r] = nd[r * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) rw[r] = hb[(w * 4 + r) * 64 + lane];
            v4d z = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 4; ++r) z = __builtin_amdgcn_mfma_f64_16x16x4f64(ndr[r], rw[r], z, 0, 0, 0);  // Z̃ = −D⁻¹ R_w
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t == k) {  // column tile: (D⁻¹ R_w)' = R_w' D⁻¹
                    v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(rw[r], -ndr[r], acc, 0, 0, 0);
                    a.v[t][0] = acc[0]; a.v[t][1] = acc[1]; a.v[t][2] = acc[2]; a.v[t][3] = acc[3];
                } else {
                    double rt[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) rt[r] = hb[(t * 4 + r) * 64 + lane];
                    v4d acc = {a.v[t][0], a.v[t][1], a.v[t][2], a.v[t][3]};
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(z[r], rt[r], acc, 0, 0, 0);
                    a.v[t][0] = acc[0]; a.v[t][1] = acc[1]; a.v[t][2] = acc[2]; a.v[t][3] = acc[3];
                }
            }
        }
    }
    RXHIP_BPH(3);
    // the array holds −A_s⁻¹: negate, scale back
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) a.v[t][r] = __builtin_ldexp(-a.v[t][r], hr[r] + hc[t]);
    RXHIP_BPH(4);
    if (FINAL) lds_barrier();  // the scratch (exponent table, last half) may be reused by the caller
    return !bad;
}
// the SPD inverse the sweep kernels call: RXHIP_INV_BLOCKED selects the panel form
#define RXHIP_INV_BLOCKED 1   // (the rank-4 sweep inverse of rounds 1–2 lost every digit on badly scaled input: it lives on in scripts/ only)
#ifndef RXHIP_KF_RELOAD
#define RXHIP_KF_RELOAD 1
#endif
#ifndef RXHIP_FWD_FROZEN
#define RXHIP_FWD_FROZEN 1
#endif
#ifndef RXHIP_FZ_CONFIRM
#define RXHIP_FZ_CONFIRM 1   // consecutive steps on which BOTH tests of the recursion's matrix (below) must pass before a segment leaves the matrix work: one
                             // such step is two consecutive repeats, M_{t-1} -> M_t by test (a) and M_t -> M_{t+1} by test (b).  (2: measured on the BASELINE d = 64
                             // chain, 1000 segments of 10 steps that start on the fixed point — 3 full steps per segment instead of 2, k_forward 0.23 -> 0.46 ms)
#endif
#ifndef RXHIP_FZ_LANE_TOL
#define RXHIP_FZ_LANE_TOL 1.4e-14   // kd_forward_info, test (a): a lane's plain sum over its 16 equilibrated entries (each O(1) at the diagonal) repeats to 16 · 4 ulp of ONE,
                                    // its 1.37^k-weighted sum to 26 times that (measured on the BASELINE d = 64 model: at 2 ulp of the sums no segment ever froze)
#endif
#ifndef RXHIP_FZ_TOL
#define RXHIP_FZ_TOL 1.5e-14 // kd_backward_info, the verification step: |V_s(t) − V_s(t+1)|_ij ≤ TOL · sqrt(V_ii V_jj) (64 ulp on the entry's own scale)
#endif

template <int NT, class PF = NoPrefetch, bool FINAL = true, bool PUB = true, class SEED = NoSeed>
__device__ __forceinline__ bool spd_inverse(Acc<NT>& a, double* scratch, int w, int lane, LogProd& lp, PF prefetch = PF(), SEED* seed = nullptr) {
    return blk_inverse<NT, PF, FINAL, PUB, SEED>(a, scratch, w, lane, lp, prefetch, seed);
}

// ---- vectors in LDS ------------------------------------------------------------------------------
// out[i] = s0·add[i] + Σ_k M[i][k] x[k]   (M: n×m row-major ld; thread i < n does row i)
__device__ __forceinline__ void matvec_lds(double* out, const double* M, int ld, int n, int m, const double* x,
                                           const double* add, double sadd, int tid) {
    if (tid < n) {
        double s = add ? sadd * add[tid] : 0.0;
#pragma unroll 8
        for (int k = 0; k < m; ++k) s += M[tid * ld + k] * x[k];
        out[tid] = s;
    }
}
// out[k] = Σ_i M[i][k] x[i]  (transposed)
__device__ __forceinline__ void matTvec_lds(double* out, const double* M, int ld, int n, int m, const double* x,
                                            const double* add, double sadd, int tid) {
    if (tid < m) {
        double s = add ? sadd * add[tid] : 0.0;
        for (int i = 0; i < n; ++i) s += M[i * ld + tid] * x[i];
        out[tid] = s;
    }
}
// out[i] = sadd·add[i] + Σ_k MT[k][i] x[k]  with MT = M' stored row-major [m][n] in global memory:
// consecutive threads read consecutive addresses (coalesced), the m loads of a thread are independent.
__device__ __forceinline__ void matvec_gT(double* out, const double* MT, int n, int m, const double* x, const double* add,
                                          double sadd, int tid) {
    if (tid < n) {
        double s0 = add ? sadd * add[tid] : 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int k = 0;
        for (; k + 15 < m; k += 16) {  // 16 independent (coalesced) loads in flight per thread
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = MT[(size_t)(k + u) * n + tid];
#pragma unroll
            for (int u = 0; u < 16; u += 4) {
                s0 += v[u] * x[k + u];
                s1 += v[u + 1] * x[k + u + 1];
                s2 += v[u + 2] * x[k + u + 2];
                s3 += v[u + 3] * x[k + u + 3];
            }
        }
        for (; k < m; ++k) s0 += MT[(size_t)k * n + tid] * x[k];
        out[tid] = (s0 + s1) + (s2 + s3);
    }
}
// out[r] = Σ_k MT[k][r] x[k] for r = i, i + width, … < n   (MT = M' row-major [m][n] in global memory; one thread group of
// `width` threads, local index i) — lets different thread groups of a workgroup run different matvecs at the same time
__device__ __forceinline__ void matvec_gT_group(double* out, const double* MT, int n, int m, const double* x, int i, int width) {
    for (int r = i; r < n; r += width) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int k = 0;
        for (; k + 15 < m; k += 16) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = MT[(size_t)(k + u) * n + r];
#pragma unroll
            for (int u = 0; u < 16; u += 4) {
                s0 += v[u] * x[k + u];
                s1 += v[u + 1] * x[k + u + 1];
                s2 += v[u + 2] * x[k + u + 2];
                s3 += v[u + 3] * x[k + u + 3];
            }
        }
        for (; k < m; ++k) s0 += MT[(size_t)k * n + r] * x[k];
        out[r] = (s0 + s1) + (s2 + s3);
    }
}

// three block-wide dot products in one reduction (result valid in every thread); red: 3·nthreads doubles.
// Works for any thread count (192 threads at d = 48).
__device__ __forceinline__ void block_dot3(const double* a0, const double* b0, int n0, const double* a1, const double* b1,
                                           int n1, const double* a2, const double* b2, int n2, double* red, int tid,
                                           int nthreads, double (&out)[3]) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int i = tid; i < n0; i += nthreads) s0 += a0[i] * b0[i];
    for (int i = tid; i < n1; i += nthreads) s1 += a1[i] * b1[i];
    for (int i = tid; i < n2; i += nthreads) s2 += a2[i] * b2[i];
    red[tid] = s0;
    red[nthreads + tid] = s1;
    red[2 * nthreads + tid] = s2;
    lds_barrier();
    for (int n = nthreads; n > 1;) {
        const int h = (n + 1) / 2;
        if (tid < n - h) {
            red[tid] += red[tid + h];
            red[nthreads + tid] += red[nthreads + tid + h];
            red[2 * nthreads + tid] += red[2 * nthreads + tid + h];
        }
        lds_barrier();
        n = h;
    }
    out[0] = red[0];
    out[1] = red[nthreads];
    out[2] = red[2 * nthreads];
    lds_barrier();
}

// ------------------------------------------------------------------------------------------------
// LDS carve (dynamic): matrices + vectors + the scratch of the SPD inverse
template <int NT>
struct DenseLds {
    using C = DenseCfg<NT>;
    static constexpr int NVEC = 12;
    // scratch of spd_inverse: the panel buffers of blk_inverse, or the pivot-row buffers of gj_inverse
    static constexpr int SCR = RXHIP_INV_BLOCKED ? blk_scratch_doubles(NT) : 8 * C::D;
    // kernels that keep two LDS matrices (two workgroups per CU at d = 64) lend the inverse a matrix that is dead while it runs
    static constexpr bool ALIAS = RXHIP_INV_BLOCKED && C::MAT >= blk_scratch_doubles(NT);
    static constexpr int SCR2 = ALIAS ? 8 * C::D : (SCR > 8 * C::D ? SCR : 8 * C::D);   // what those kernels carve besides
    // kd_forward_info at d ≥ 48 keeps a seed of the tile inverse per wave there (DiagSeed) — 162 304 bytes for two workgroups at d = 64, of 163 840
    static constexpr bool SEEDED = ALIAS;
    static constexpr int FWD_FROZEN = 16;   // kd_forward_info: two functionals of M per wave, the log-determinant state in front of the inverse
    static constexpr int FWD_TAIL = (SEEDED && NT * DiagSeed::LDS_DOUBLES > SCR2 ? NT * DiagSeed::LDS_DOUBLES : SCR2) + FWD_FROZEN;
    static constexpr size_t bytes(int dmax) {   // scan kernels (vectors + the staging of dense_affine_rounds), kd_prepare_bnd (scratch only)
        return sizeof(double) * ((size_t)NVEC * dmax + (SCR > 8 * C::D ? SCR : 8 * C::D) + 3 * C::THREADS + 32 + 2 * 32 * C::D + 48);
    }
    // kd_forward (filtering runs): 3 matrices, 5 vectors, the scratch, 2 × 4 partial-sum rows
    static constexpr size_t fwd_bytes(int dmax) {
        return sizeof(double) * ((size_t)3 * C::MAT + (size_t)13 * dmax + (SCR > 8 * C::D ? SCR : 8 * C::D));
    }
    // kd_forward_info: 2 matrices, ξ_f | u | 4 partial-sum rows, the scratch
    static constexpr size_t fwd_info_bytes(int dmax) {
        return sizeof(double) * ((size_t)2 * C::MAT + (size_t)10 * dmax + FWD_TAIL);
    }
    // kd_backward_info: 2 matrices, m_s | C ξ_f, the scratch / matvec partials
    static constexpr size_t bwd_info_bytes(int dmax) {
        return sizeof(double) * ((size_t)2 * C::MAT + (size_t)2 * dmax + SCR2);
    }
    // kd_agg_finish: 6 vectors, the map (B'Q⁻¹)' and a tile of 16 observations
    static constexpr size_t agg_bytes(int dy) {
        return sizeof(double) * ((size_t)6 * ((((C::D > dy ? C::D : dy) + 1) & ~1)) + (size_t)C::D * dy + (size_t)16 * dy + 16);
    }
};

// n doubles into LDS, NB loads per thread in flight: dst[k] = load(k).  Written as a plain loop (load, store, next) the compiler
// waits for every load before its store — one memory round trip per pass, and staging a 64×64 map with 256 threads is sixteen of
// them (≈1 µs each out of L2).  The index is clamped instead of tested, and an empty asm between the loads and the stores keeps
// the compiler from moving each load down to its store.
template <int NB, int NTHR, class F>
__device__ __forceinline__ void lds_stage_batched(double* dst, int n, int tid, F load) {
    for (int k0 = 0; k0 < n; k0 += NB * NTHR) {
        double v[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int k = k0 + tid + u * NTHR;
            v[u] = load(k < n ? k : n - 1);
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) asm volatile("" : "+v"(v[u]));
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int k = k0 + tid + u * NTHR;
            if (k < n) dst[k] = v[u];
        }
    }
}

// phase 1 (dense): the element (b_s, η_s) of every segment.  It is linear in the segment's observations with per-offset
// constant maps (host tables Ψ_i, Θ_i: see build_dense_tables), so instead of a sequential recursion per segment it is ONE
// product  [2d × L·dy] · [L·dy × S]  on the matrix cores, split over K-chunks of agg_oc offsets for parallelism:
//   kd_agg_gemm   grid (segment blocks of 16, K-chunks, chains): partial sums, v_mfma_f64_16x16x4_f64, operands straight from
//                 L2 (the table is shared by all segments; y is read once)
//   kd_agg_finish grid (S, chains): fixed-order sum of the partials, the data-dependent halves of the boundary scan
//                 (w_s = b_s + M2_s η_s, w_s' = η_s − N2_s b_s) and B'Q⁻¹y_t of every step of the segment
//                 (handed to the forward kernels in the record).
// The full segments 0 … S−2 share table set 0; the last segment (length Llast ≤ L) has its own set and its own block.
template <int NT>
__global__ void __launch_bounds__(64 * NT) kd_agg_gemm(DenseParams p) {
    constexpr int D = 16 * NT;
    typedef double v4d __attribute__((ext_vector_type(4)));
    const int dy = p.dy, dyp = (dy + 3) & ~3, tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int jl = lane & 15, kq = lane >> 4;
    const int nblk = gridDim.x, sb = blockIdx.x, kc = blockIdx.y;
    const long long chain = blockIdx.z + p.chain0;
    const DenseModel M = dense_model(p, chain);  // this chain's model tables
    const bool lastblk = sb == nblk - 1;
    const int seg = lastblk ? p.S - 1 : 16 * sb + jl;                      // this lane's B-operand column
    const bool segok = lastblk ? jl == 0 : seg < p.S - 1;
    const long long len = lastblk ? p.Llast : p.L;
    const double* tab = M.tab + (lastblk ? (size_t)p.L * dyp * 2 * D : 0);
    long long o0 = (long long)kc * p.agg_oc, o1 = o0 + p.agg_oc;
    if (o1 > len) o1 = len;
    v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    const int r0 = 16 * w + jl, r1 = 16 * (w + NT) + jl;                   // A-operand rows of this wave's two tiles
    // flattened (offset, k-step) loop, AG steps per trip with all their loads issued first
    const int kpo = dyp / 4;
    const long long nq = (o1 > o0 ? o1 - o0 : 0) * kpo;
    const double* ybase = p.y + ((1 + (long long)seg * p.L) * p.n_chains + chain) * dy;
    const long long ystride = p.n_chains * (long long)dy;
    long long o = o0;
    int kk = 0;
    constexpr int AG = 16;  // k-steps per trip: 3·AG loads in flight (the operands come straight from L2)
    for (long long q = 0; q < nq; q += AG) {
        double a0[AG], a1[AG], b[AG];
#pragma unroll
        for (int u = 0; u < AG; ++u) {
            const bool in = q + u < nq;
            const int k = 4 * kk + kq;
            const double* tb = tab + ((size_t)o * dyp + k) * 2 * D;
            a0[u] = in ? tb[r0] : 0.0;
            a1[u] = in ? tb[r1] : 0.0;
            b[u] = (in && segok && k < dy) ? ybase[o * ystride + k] : 0.0;
            if (++kk == kpo) { kk = 0; ++o; }
        }
#pragma unroll
        for (int u = 0; u < AG; ++u) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b[u], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b[u], acc1, 0, 0, 0);
        }
    }
    // accumulator element (row = 16·tile + kq + 4r, column = segment jl)
    if (segok) {
        double* out = p.aggpart + (((size_t)chain * p.agg_kc + kc) * p.S + seg) * 2 * D;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            out[16 * w + kq + 4 * r] = acc0[r];
            out[16 * (w + NT) + kq + 4 * r] = acc1[r];
        }
    }
}

template <int NT>
__global__ void __launch_bounds__(64 * NT) kd_agg_finish(DenseParams p) {
    constexpr int D = 16 * NT;
    constexpr int TS = 16;  // steps per tile of the B'Q⁻¹y pass
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int dy = p.dy, tid = threadIdx.x;
    const int dm = ((D > dy ? D : dy) + 1) & ~1;  // even: keeps every LDS carve 16-byte aligned
    double* m = smem;               // b_s
    double* eta = m + dm;           // η_s
    double* red = eta + dm;         // [4][dm] partial sums
    double* GTs = red + 4 * dm;     // (B'Q⁻¹)' [dy][D]
    double* Ys = GTs + D * dy;      // [TS][dy]
    const long long seg = blockIdx.x, chain = blockIdx.y + p.chain0;
    const DenseModel M = dense_model(p, chain);  // this chain's model tables
    const DenseCst c = DenseCst::make(D, dy);
    const double* cst = M.cst;
    const long long b0 = 1 + seg * p.L;
    long long b1 = b0 + p.L;
    if (b1 > p.T) b1 = p.T;
    const long long len = b1 - b0;
    const int part = tid / D, i = tid - part * D;
    const int half = part & 1;
    if (tid < 2 * D) {   // fixed-order sum of the K-chunk partials, every load in flight at once
        double s = 0.0;
        for (int kc0 = 0; kc0 < p.agg_kc; kc0 += 12) {   // (dense_schedule_ints: at most 12 chunks)
            double v[12];
#pragma unroll
            for (int u = 0; u < 12; ++u) {
                const int kc = kc0 + u < p.agg_kc ? kc0 + u : p.agg_kc - 1;
                v[u] = p.aggpart[(((size_t)chain * p.agg_kc + kc) * p.S + seg) * 2 * D + tid];
            }
#pragma unroll
            for (int u = 0; u < 12; ++u) asm volatile("" : "+v"(v[u]));
#pragma unroll
            for (int u = 0; u < 12; ++u) s += kc0 + u < p.agg_kc ? v[u] : 0.0;
        }
        if (tid < D) m[tid] = s;
        else eta[tid - D] = s;
    }
    lds_stage_batched<16, 64 * NT>(GTs, D * dy, tid, [&](int q) { return cst[c.oGT + q]; });
    lds_barrier();
    // The boundary scan carries v <- w_s + M1_s v (prefix) and ξ <- w_s' + N1_s ξ (suffix); the parts that do not depend on
    // the carried vector are formed here, in parallel over segments:  w_s = b_s + M2_s η_s,  w_s' = η_s − N2_s b_s.
    {
        const size_t MM = (size_t)D * D;
        const double* Mt = scan_mat(M, p.S, seg, part < 2 ? 1 : 4, MM);  // transposed maps: [k][i]
        const double* x = part < 2 ? eta : m;
        const int k0 = half * (D / 2);
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int k = 0; k < D / 2; k += 16) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = (k + u < D / 2) ? Mt[(size_t)(k0 + k + u) * D + i] : 0.0;
#pragma unroll
            for (int u = 0; u < 16; u += 2)
                if (k + u < D / 2) {  // D/2 is even: pairs never straddle the bound
                    s0 += v[u] * x[k0 + k + u];
                    s1 += v[u + 1] * x[k0 + k + u + 1];
                }
        }
        red[part * dm + i] = s0 + s1;
        lds_barrier();
        if (tid < D) {
            double* o = p.elem + ((chain * p.S + seg) * 2) * D;
            o[tid] = m[tid] + (red[tid] + red[dm + tid]);                     // w_s
            o[D + tid] = eta[tid] - (red[2 * dm + tid] + red[3 * dm + tid]);  // w_s'
        }
    }
    // B'Q⁻¹ y_t for every step of the segment, TS steps per pass: thread (i, part) forms row i for steps part, part + 4, …
    for (long long s0 = 0; s0 < len; s0 += TS) {
        lds_barrier();
        lds_stage_batched<4, 64 * NT>(Ys, TS * dy, tid, [&](int q) {   // (steps past the segment: the last step again; never stored to the record)
            const int st = q / dy, j = q - st * dy;
            const long long t = b0 + (s0 + st < len ? s0 + st : len - 1);
            return p.y[(t * p.n_chains + chain) * dy + j];
        });
        lds_barrier();
        double g[TS / 4];
#pragma unroll
        for (int u = 0; u < TS / 4; ++u) g[u] = 0.0;
#pragma unroll 4
        for (int k = 0; k < dy; ++k) {
            const double gt = GTs[k * D + i];
#pragma unroll
            for (int u = 0; u < TS / 4; ++u) g[u] += gt * Ys[(part + 4 * u) * dy + k];
        }
#pragma unroll
        for (int u = 0; u < TS / 4; ++u) {
            const long long st = s0 + part + 4 * u;
            if (st < len) p.filt[(chain * p.T + (b0 + st)) * DenseCfg<NT>::REC + D + i] = g[u];
        }
    }
}

// ---- boundary scan, two levels --------------------------------------------------------------------------------------
// The scan carries a d-vector across segment boundaries: x_{st+1} = w_st + Map_st x_st, st = 0 … S−2 (prefix: segments
// ascending, map 0 and w_s of the aggregation kernel, x = filtered mean at the segment start; suffix: segments descending,
// map 3 and w_s', x = ξβ at the segment end).  The maps are per-model constants, so products of maps are too: the host
// composes, for every q = st + 1, Q_q = Map_st ⋯ Map_{first step of q's group} (groups of sg ≈ √S steps).  Then
//   level 1 (kd_scan_local, one workgroup per group): the recursion from x = 0 at every group start -> l_q
//   level 2 (kd_scan_fix, every workgroup redundantly): X_{j+1} = l_{(j+1)sg} + Q_{(j+1)sg} X_j over the groups before its own
//   level 3 (kd_scan_fix):                              x_q = l_q + Q_q X_j   for the q of its group
// — a sequential depth of ≈ 3√S matvec rounds instead of S (250 rounds of ≈1.1 µs were 16 % of a d = 64 sweep).
// One round: y = w + Mt'x with thread group `part` summing a quarter of the k range; the maps of the next PD rounds are kept
// in flight in registers (a cold 32 KB map read costs ≈2.6 µs; the loads do not depend on x).
//   CARRY: x <- y after every round (recursion), else x stays (independent applications).
// Round 3: the additive vectors w of a chunk of rounds are staged in LDS up front and the results leave through LDS at its end
// (`stage`: 2·SCAN_CHUNK·D doubles) — inside the chain of rounds a global load or store is a VMEM wait per round (and in a
// converged stretch, where the canonical map stays in registers, the only one): 1.0 -> ≈0.5 µs per round.
constexpr int SCAN_CHUNK = 32;
// One workgroup barrier per round: wave w owns rows 16w … 16w + 15 of the product — lane (r = lane & 15, kq = lane >> 4) sums a
// quarter of the k range of row 16w + r, two shuffles add the four quarters, the 16 lanes with kq = 0 publish x — instead of four
// thread groups writing partial sums through LDS for wave 0 to combine behind a second barrier.
template <int NT, bool CARRY, class MapF, class WF, class OutF>
__device__ __forceinline__ void dense_affine_rounds(int nrounds, MapF map_of, WF w_of, OutF out, double* v0, double* v0alt, double* stage, int tid) {
    // v0 | v0alt: the carried vector, double-buffered (a round reads one copy and publishes into the other: ONE barrier per round);
    // on return the state is in v0 again
    constexpr int D = 16 * NT, KP = D / 4, PD = 4;
    const int w = tid >> 6, lane = tid & 63, row = 16 * w + (lane & 15), k0 = (lane >> 4) * KP;
    const bool pub = lane < 16;
    if (nrounds <= 0) return;
    double* wst = stage;                     // [SCAN_CHUNK][D]
    double* ost = stage + SCAN_CHUNK * D;    // [SCAN_CHUNK][D]
    // the map of every round of a chunk, resolved up front: map_of() goes through the canonical-index table in global memory, and a
    // dependent load per round in front of a uniform branch was the whole round time (≈1 µs)
    const double** mst = reinterpret_cast<const double**>(stage + 2 * SCAN_CHUNK * D);   // [SCAN_CHUNK + PD]
    double buf[PD][KP];
    const double* cur[PD];   // the map each slot holds (uniform over the workgroup): a canonical map that is still there is not fetched again
#pragma unroll
    for (int q = 0; q < PD; ++q) cur[q] = nullptr;
    auto fetch = [&](double (&dst)[KP], const double*& have, const double* Mt0) {
        const double* Mt = as_global_v(Mt0);   // (through LDS the pointer lost its address space: flat loads count as LDS traffic too)
        if (Mt != have) {
#pragma unroll
            for (int u = 0; u < KP; ++u) dst[u] = Mt[(size_t)(k0 + u) * D + row];
            have = Mt;
        }
    };
    for (int c0 = 0; c0 < nrounds; c0 += SCAN_CHUNK) {
        const int cn = nrounds - c0 < SCAN_CHUNK ? nrounds - c0 : SCAN_CHUNK;
        if (tid < cn + PD) {   // maps of this chunk and of the PD rounds fetched ahead of the next one (clamped)
            const int r = c0 + tid < nrounds ? c0 + tid : nrounds - 1;
            mst[tid] = map_of(r);
        }
        {   // the chunk's additive vectors: SCAN_CHUNK·D elements over the 4·D threads, EVERY load issued before the first store (the
            // round index is clamped, not tested: a test per load makes it load -> wait -> store, one memory round trip per round)
            constexpr int PER = SCAN_CHUNK / 4;
            double wv[PER];
            const int part = tid / D, i = tid - part * D;
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int rr = part + 4 * u;
                wv[u] = w_of(c0 + (rr < cn ? rr : cn - 1))[i];
            }
#pragma unroll
            for (int u = 0; u < PER; ++u) wst[(part + 4 * u) * D + i] = wv[u];
        }
        lds_barrier();
        if (c0 == 0) {
#pragma unroll
            for (int q = 0; q < PD; ++q) fetch(buf[q], cur[q], mst[q]);
        }
        for (int r0 = c0; r0 < c0 + cn; r0 += PD) {   // SCAN_CHUNK is a multiple of PD: slot q always holds round r0 + q
#pragma unroll
            for (int q = 0; q < PD; ++q) {
                const int r = r0 + q;
                if (r >= c0 + cn) break;
                const double* vin = (CARRY && (r & 1)) ? v0alt : v0;
                double* vout = (r & 1) ? v0 : v0alt;
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int u = 0; u < KP; u += 2) {
                    s0 += buf[q][u] * vin[k0 + u];
                    s1 += buf[q][u + 1] * vin[k0 + u + 1];
                }
                fetch(buf[q], cur[q], mst[r - c0 + PD]);
                double sm = s0 + s1;
                sm += __shfl_xor(sm, 16);
                sm += __shfl_xor(sm, 32);
                const double x = wst[(r - c0) * D + row] + sm;
                if (pub) {
                    if (CARRY) vout[row] = x;
                    ost[(r - c0) * D + row] = x;
                }
                if (CARRY) lds_barrier();   // the copy this round read is free again only after the NEXT round's barrier: two copies
            }
        }
        lds_barrier();
        if (tid < D)
            for (int rr = 0; rr < cn; ++rr) out(c0 + rr, ost[rr * D + tid]);
        lds_barrier();
    }
    if (CARRY && (nrounds & 1)) {   // an odd number of rounds leaves the state in the second copy
        if (tid < D) v0[tid] = v0alt[tid];
        lds_barrier();
    }
}

// level 1.  grid (2·ng + 1, chains): blockIdx.x = dir·ng + group;  dir 0 = prefix, 1 = suffix.  The last workgroup performs the
// t = 1 update (filtered belief of the first observation, evidence term of filtering runs).
template <int NT, bool FE>
__global__ void __launch_bounds__(64 * NT) kd_scan_local(DenseParams p) {
    constexpr int D = 16 * NT;
    using C = DenseCfg<NT>;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int dy = p.dy, tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int dm = ((D > dy ? D : dy) + 1) & ~1;  // even: keeps every LDS carve 16-byte aligned
    double* v0 = smem;
    double* v1 = v0 + dm;
    double* v2 = v1 + dm;
    double* yv = v2 + dm;
    double* red = yv + dm;
    const long long chain = blockIdx.y + p.chain0;
    const DenseModel M = dense_model(p, chain);  // this chain's model tables
    const DenseCst c = DenseCst::make(D, dy);
    const double* cst = M.cst;
    const int S = p.S, n = S - 1;
    const size_t MM = (size_t)D * D;
    if (blockIdx.x == gridDim.x - 1) {
        // The extra workgroup: filtered belief at t = 1: ξf = V1⁻¹m1 + G y;  mf = c1 + K1 y — beside the scans, not in front of the
        // first group's (its loops over dy were eight memory round trips at the head of the kernel's longest workgroup).  The four
        // thread groups of D sum a quarter of the k range each.
        if (tid < dy) yv[tid] = p.y[(0 * p.n_chains + chain) * dy + tid];
        lds_barrier();
        {
            double* pst = red + 3 * 64 * NT + 16;   // [3][4][dm] partial sums (the scan's staging area, unused here)
            const int part = tid / D, i = tid - part * D, kb = (dy + 3) / 4, k0 = part * kb, k1 = k0 + kb < dy ? k0 + kb : dy;
            double xs = 0.0, ms = 0.0, qs = 0.0;
#pragma unroll 8
            for (int k = k0; k < k1; ++k) {
                xs += cst[c.oGT + (long long)k * D + i] * yv[k];
                ms += cst[c.oK1T + (long long)k * D + i] * yv[k];
            }
            const bool qsplit = dy <= D;   // rows of Q⁻¹y fit a thread group (else: one thread per row, the whole k range)
            if (qsplit ? i < dy : tid < dy) {
                const int r = qsplit ? i : tid, q0 = qsplit ? k0 : 0, q1 = qsplit ? k1 : dy;
#pragma unroll 8
                for (int k = q0; k < q1; ++k) qs += cst[c.oQI + (long long)k * dy + r] * yv[k];   // Q⁻¹ is symmetric: column r, coalesced
            }
            pst[(0 * 4 + part) * dm + i] = xs;
            pst[(1 * 4 + part) * dm + i] = ms;
            if (qsplit) { if (i < dy) pst[(2 * 4 + part) * dm + i] = qs; }
            else if (tid < dy) { pst[8 * dm + tid] = qs; pst[9 * dm + tid] = 0.0; pst[10 * dm + tid] = 0.0; pst[11 * dm + tid] = 0.0; }
            lds_barrier();
            if (tid < D) {
                v1[tid] = cst[c.oX1 + tid] + ((pst[0 * dm + tid] + pst[1 * dm + tid]) + (pst[2 * dm + tid] + pst[3 * dm + tid]));   // ξf
                v0[tid] = cst[c.oC1 + tid] + ((pst[4 * dm + tid] + pst[5 * dm + tid]) + (pst[6 * dm + tid] + pst[7 * dm + tid]));   // mf
            }
            if (tid < dy) v2[tid] = (pst[8 * dm + tid] + pst[9 * dm + tid]) + (pst[10 * dm + tid] + pst[11 * dm + tid]);             // Q⁻¹ y
        }
        lds_barrier();
        double* rec = p.filt + (chain * p.T + 0) * C::REC;
        if (tid < D) {
            rec[tid] = v0[tid];
            if (S > 0) p.fstart_m[(chain * S + 0) * D + tid] = v0[tid];  // x_0 of the prefix scan
        }
        {
            Acc<NT> a;
            acc_load<NT>(a, cst + c.oVF1, D, w, lane);
            if (p.T == 1 || p.filter) {
                if (tid < D) dense_store_mean(p, 0, chain, tid, v0[tid]);
                dense_store_cov<NT>(p, a, 0, chain, w, lane);
            }
        }
        if (FE) {
            double dots[3], dot1[3] = {0.0, 0.0, 0.0};
            block_dot3(v2, yv, dy, v1, v0, D, v1, v0, 0, red, tid, 64 * NT, dots);
            if (NT == 1 && p.pack == 2)   // the second chain's share of the two data-dependent dot products
                block_dot3(v2 + p.dy_sub, yv + p.dy_sub, dy - p.dy_sub, v1 + p.d_sub, v0 + p.d_sub, D - p.d_sub, v1, v0, 0, red, tid, 64 * NT, dot1);
            if (tid == 0) {
                const double dep1 = dot1[0] - dot1[1];
                dense_fe_write(p, 0, chain, cst[c.oC0] + cst[c.oS1] + cst[c.oLD1], (dots[0] - dots[1]) - dep1, dep1);
            }
        }
        return;
    }
    const int dir = blockIdx.x / p.ng, grpj = blockIdx.x - dir * p.ng;
    if (dir == 1 && grpj == 0 && tid < D && S > 0) p.beta_xi[(chain * (S + 1) + S) * D + tid] = 0.0;  // x_0 of the suffix scan
    if (n <= 0) return;
    const int st0 = grpj * p.sg;
    int st1 = st0 + p.sg;
    if (st1 > n) st1 = n;
    if (tid < D) v0[tid] = 0.0;
    lds_barrier();
    double* loc = p.loc + ((chain * 2 + dir) * (size_t)S) * D;
    auto seg_of = [&](int st) { return dir ? S - 1 - st : st; };
    dense_affine_rounds<NT, true>(
        st1 - st0,
        [&](int r) { return scan_mat(M, S, seg_of(st0 + r), dir ? 3 : 0, MM); },
        [&](int r) { return p.elem + ((chain * S + seg_of(st0 + r)) * 2 + dir) * D; },
        [&](int r, double x) { loc[(size_t)(st0 + r + 1) * D + tid] = x; }, v0, v1, red + 3 * 64 * NT + 16, tid);
}

// levels 2 and 3.  Same grid.  Output: prefix x_q -> fstart_m[q] (filtered mean at the start of segment q);
// suffix x_q -> beta_xi[S − q] (ξβ at the start boundary of segment S − q).
template <int NT>
__global__ void __launch_bounds__(64 * NT) kd_scan_fix(DenseParams p) {
    constexpr int D = 16 * NT;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int dy = p.dy, tid = threadIdx.x;
    const int dm = ((D > dy ? D : dy) + 1) & ~1;
    double* v0 = smem;
    double* red = v0 + 4 * dm;
    const long long chain = blockIdx.y + p.chain0;
    const DenseModel M = dense_model(p, chain);  // this chain's model tables
    const int S = p.S, n = S - 1;
    const int dir = blockIdx.x / p.ng, grpj = blockIdx.x - dir * p.ng;
    const size_t MM = (size_t)D * D;
    if (n <= 0) return;
    const double* loc = p.loc + ((chain * 2 + dir) * (size_t)S) * D;
    const double* qt = M.qtab + (size_t)dir * S * MM;
    const int* qc = M.canon ? M.canon + (2 + dir) * S : nullptr;   // canonical q of every composed map
    if (tid < D) v0[tid] = dir ? 0.0 : p.fstart_m[(chain * S + 0) * D + tid];
    lds_barrier();
    const int sg = p.sg;
    // level 2: state at the start of this group
    dense_affine_rounds<NT, true>(
        grpj, [&](int r) { const int q = (r + 1) * sg; return qt + (size_t)(qc ? qc[q] : q) * MM; }, [&](int r) { return loc + (size_t)((r + 1) * sg) * D; },
        [&](int, double) {}, v0, v0 + dm, red + 4 * D + 16, tid);
    // level 3: the states inside the group
    const int q0 = grpj * sg + 1;
    int q1 = q0 + sg;
    if (q1 > n + 1) q1 = n + 1;
    dense_affine_rounds<NT, false>(
        q1 - q0, [&](int r) { const int q = q0 + r; return qt + (size_t)(qc ? qc[q] : q) * MM; }, [&](int r) { return loc + (size_t)(q0 + r) * D; },
        [&](int r, double x) {
            const int q = q0 + r;
            if (dir) p.beta_xi[(chain * (S + 1) + (S - q)) * D + tid] = x;
            else p.fstart_m[(chain * S + q) * D + tid] = x;
        },
        v0, v0 + dm, red + 4 * D + 16, tid);
}

// Once per engine: the inverses at the segment boundaries that do not depend on the data —
//   bnd[s][0] = Λ_f(b_s) = V(b_s)⁻¹  (start of kd_forward_info),   bnd[s][1] = V_s(b_{s+1}) = (Λ_f(b_{s+1}) + Λβ(b_{s+1}))⁻¹  (start of
//   kd_backward_info, s < S − 1) — so that no sweep pays for them.  One workgroup per segment.
template <int NT>
__global__ void __launch_bounds__(64 * NT) kd_prepare_bnd(DenseParams p) {
    constexpr int D = 16 * NT;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* rowbuf = smem;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int seg = blockIdx.x;
    const size_t MM = (size_t)D * D;
    bool ok = true;
    LogProd lpd;
    Acc<NT> a;
    const DenseModel M{p.cst, p.tab, p.scanm, p.qtab, p.bnd, p.canon};
    acc_load<NT>(a, scan_mat(M, p.S, seg, 2, MM), D, w, lane);
    ok = spd_inverse<NT>(a, rowbuf, w, lane, lpd) && ok;
    acc_store<NT>(a, p.bnd + ((size_t)seg * 2 + 0) * MM, D, w, lane);
    if (seg + 1 < p.S) {
        acc_load<NT>(a, scan_mat(M, p.S, seg + 1, 2, MM), D, w, lane);
        ok = spd_inverse<NT>(a, rowbuf, w, lane, lpd) && ok;
        acc_add_mat<NT>(a, scan_mat(M, p.S, seg, 5, MM), D, w, lane, 1.0);
        ok = spd_inverse<NT>(a, rowbuf, w, lane, lpd) && ok;
        acc_store<NT>(a, p.bnd + ((size_t)seg * 2 + 1) * MM, D, w, lane);
    }
    if (!ok && tid == 0) atomicOr(p.status, ST_NOT_POSDEF);
}

// sum over the first n ≤ 64 lanes of wave 0 (valid in lane 0)
__device__ __forceinline__ double wave0_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}

// phase 3 (dense, FILTERING runs): covariance-form forward sweep of one segment — the filtered belief (m_f, V_f) of every
// time index is the marginal of the streaming one-step graph and is written out (two inverses per step: V_p⁻¹ and Λ_f⁻¹).
// Smoothing runs use the information-form kernels at the end of this file.  As there, everything that does not depend on the
// carried belief stays out of the chain: A is the A operand of both contractions (V_p = A (A V)' + P) and lives in registers,
// B'Q⁻¹y_t comes from the aggregation kernel, the matvecs are spread over the four thread groups and combined after barriers
// the step has anyway, the three dot products of the evidence term are wave-0 shuffles.
template <int NT, bool FE>
__global__ void __launch_bounds__(64 * NT) kd_forward(DenseParams p) {
    constexpr int D = 16 * NT;
    using C = DenseCfg<NT>;
    constexpr int LD = C::LD;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int dy = p.dy, tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int dm = ((D > dy ? D : dy) + 1) & ~1;  // even: keeps every LDS carve 16-byte aligned
    double* M0 = smem;          // V_f of the previous step, then of this step
    double* M1 = M0 + C::MAT;   // T = A V_f
    double* M2 = M1 + C::MAT;   // Λp
    double* vec = M2 + C::MAT;
    double* m = vec;
    double* mp = m + dm;
    double* xf = mp + dm;
    double* yv = xf + dm;
    double* qy = yv + dm;
    double* rowbuf = qy + dm;   // scratch of the SPD inverse
    double* red = rowbuf + (DenseLds<NT>::SCR > 8 * D ? DenseLds<NT>::SCR : 8 * D);  // [4][dm] partial sums
    double* red2 = red + 4 * dm;   // [4][dm]
    const long long seg = blockIdx.x, chain = blockIdx.y + p.chain0;
    const DenseModel M = dense_model(p, chain);  // this chain's model tables
    const DenseCst c = DenseCst::make(D, dy);
    const double* cst = M.cst;
    const int grp = tid / D, gi = tid - grp * D;  // four thread groups of D
    const size_t MM = (size_t)D * D;
    const long long b0 = 1 + seg * p.L;
    long long b1 = b0 + p.L;
    if (b1 > p.T) b1 = p.T;
    const long long len = b1 - b0, t0 = seg * p.L + 1;
    bool ok = true;
    double acc_quad = 0.0, acc_const = 0.0, acc_dep1 = 0.0;  // data-dependent sum | constants | second chain of a packed pair
    LogProd lp;
    double af[D / 4];  // A-operand fragments of the transition matrix
    {
        const double* A = cst + c.oA;
        const int i = 16 * w + (lane & 15), kq = lane >> 4;
#pragma unroll
        for (int kk = 0; kk < D / 4; ++kk) af[kk] = A[i * D + 4 * kk + kq];
    }
    // cacc += A Y (TY = false) or A Y' (TY = true), Y in LDS
    auto mm_a = [&](Acc<NT>& cacc, const double* Y, auto ty) {
        constexpr bool TY = decltype(ty)::value;
        const int jl = lane & 15, kq = lane >> 4;
        d4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (d4){cacc.v[t][0], cacc.v[t][1], cacc.v[t][2], cacc.v[t][3]};
#pragma unroll
        for (int kk = 0; kk < D / 4; ++kk) {
            const int k = 4 * kk + kq;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int j = 16 * t + jl;
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk], TY ? Y[j * LD + k] : Y[k * LD + j], acc[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            cacc.v[t][0] = acc[t][0];
            cacc.v[t][1] = acc[t][1];
            cacc.v[t][2] = acc[t][2];
            cacc.v[t][3] = acc[t][3];
        }
    };
    // out-partial of a matvec with a row-major LDS matrix: group grp sums a quarter of the k range for row gi
    auto part_lds = [&](double* dst, const double* M, const double* x) {
        const int k0 = grp * (D / 4);
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int k = 0; k < D / 4; k += 2) {
            s0 += M[gi * LD + k0 + k] * x[k0 + k];
            s1 += M[gi * LD + k0 + k + 1] * x[k0 + k + 1];
        }
        dst[grp * dm + gi] = s0 + s1;
    };
    if (tid < D) m[tid] = p.fstart_m[(chain * p.S + seg) * D + tid];
    Acc<NT> a, lam;
    acc_load<NT>(a, scan_mat(M, p.S, seg, 2, MM), D, w, lane);
    acc_store<NT>(a, M0, LD, w, lane);
    // y_t and B'Q⁻¹y_t (record of t, second header slot: kd_agg_finish), fetched one step ahead
    double yn = (tid < dy && len > 0) ? p.y[(t0 * p.n_chains + chain) * dy + tid] : 0.0;
    double gyn = (tid < D && len > 0) ? p.filt[(chain * p.T + t0) * C::REC + D + tid] : 0.0;
    if (tid < dy) yv[tid] = yn;
    lds_barrier();
    for (long long i = 0; i < len; ++i) {
        const long long t = t0 + i, tn = i + 1 < len ? t + 1 : t;
        const double gyc = gyn;
        if (tid < dy) yn = p.y[(tn * p.n_chains + chain) * dy + tid];
        if (tid < D) gyn = p.filt[(chain * p.T + tn) * C::REC + D + tid];
        // partial sums of  mp = A m  (`*`_A(:out) mean) and, for the evidence term, Q⁻¹y  — constant maps from L2, coalesced
        {
            const double* AT = cst + c.oAT;
            const int k0 = grp * (D / 4);
            double v[D / 4];
#pragma unroll
            for (int u = 0; u < D / 4; ++u) v[u] = AT[(size_t)(k0 + u) * D + gi];
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int u = 0; u < D / 4; u += 2) {
                s0 += v[u] * m[k0 + u];
                s1 += v[u + 1] * m[k0 + u + 1];
            }
            red[grp * dm + gi] = s0 + s1;
            if (FE) {
                const double* QI = cst + c.oQI;
                const int kq4 = (dy + 3) / 4, q0 = grp * kq4, q1 = (q0 + kq4 < dy) ? q0 + kq4 : dy;
                for (int r = gi; r < dy; r += D) {
                    double sq = 0.0;
#pragma unroll 4
                    for (int k = q0; k < q1; ++k) sq += QI[(size_t)k * dy + r] * yv[k];  // Q⁻¹ symmetric
                    red2[grp * dm + r] = sq;
                }
            }
        }
        // `*`_A(:out): T = A V ; Vp = A T' + P
        acc_load<NT>(lam, cst + c.oP, D, w, lane);
        acc_zero<NT>(a);
        mm_a(a, M0, std::false_type{});
        acc_store<NT>(a, M1, LD, w, lane);
        lds_barrier();
        if (tid < D) mp[tid] = (red[tid] + red[dm + tid]) + (red[2 * dm + tid] + red[3 * dm + tid]);
        if (FE && tid < dy) qy[tid] = (red2[tid] + red2[dm + tid]) + (red2[2 * dm + tid] + red2[3 * dm + tid]);
        mm_a(lam, M1, std::true_type{});
        // weightedmean_precision of the forward message: Λp = Vp⁻¹
        ok = spd_inverse<NT>(lam, rowbuf, w, lane, lp) && ok;
        acc_store<NT>(lam, M2, LD, w, lane);
        lds_barrier();
        // product with the `*`_B(:in) message: Λf = Λp + B'Q⁻¹B, ξf = Λp mp + G y   (partials; combined after the inverse)
        part_lds(red, M2, mp);
        acc_add_mat<NT>(lam, cst + c.oLOBS, D, w, lane, 1.0);
        // mean_cov of the product: Vf = Λf⁻¹, mf = Vf ξf
        ok = spd_inverse<NT>(lam, rowbuf, w, lane, lp) && ok;
        double sx = 0.0, xfr = 0.0;
        if (tid < D) {
            sx = (red[tid] + red[dm + tid]) + (red[2 * dm + tid] + red[3 * dm + tid]);  // Λp mp
            xfr = sx + gyc;                                                               // ξf = Λp mp + B'Q⁻¹ y
            xf[tid] = xfr;
        }
        acc_store<NT>(lam, M0, LD, w, lane);
        lds_barrier();
        part_lds(red, M0, xf);
        // q(x_t | y_1..t) is the marginal of the one-step graph
        dense_store_cov<NT>(p, lam, t, chain, w, lane);
        lds_barrier();
        double mnew = 0.0;
        if (tid < D) {
            mnew = (red[tid] + red[dm + tid]) + (red[2 * dm + tid] + red[3 * dm + tid]);
            dense_store_mean(p, t, chain, tid, mnew);
        }
        if (FE && w == 0) {  // −2 log p(y_t | y_<t) − log-dets = y'Q⁻¹y − ξf'mf + (Λp mp)'mp + const   (D, dy ≤ 64: one lane per element)
            double d0 = lane < dy ? qy[lane] * yv[lane] : 0.0;
            double d1 = lane < D ? xfr * mnew : 0.0;
            double d2 = lane < D ? sx * mp[lane] : 0.0;
            if (NT == 1 && p.pack == 2) {  // the second chain's share: its elements of the same three dot products
                double e0 = (lane >= p.dy_sub && lane < dy) ? d0 : 0.0, e12 = (lane >= p.d_sub && lane < D) ? d2 - d1 : 0.0;
                e0 = wave0_sum(e0 + e12);
                if (lane == 0) acc_dep1 += e0;
            }
            d0 = wave0_sum(d0); d1 = wave0_sum(d1); d2 = wave0_sum(d2);
            if (lane == 0) { acc_quad += d0 - d1 + d2; acc_const += cst[c.oC0]; }
        }
        lds_barrier();  // every reader of m, yv, mp of this step is done
        if (tid < D) m[tid] = mnew;
        if (tid < dy) yv[tid] = yn;
        lds_barrier();
    }
    if (FE && tid == 0) dense_fe_write(p, seg + 1, chain, acc_const + lp.value(), acc_quad - acc_dep1, acc_dep1);
    if (!ok && tid == 0) atomicOr(p.status, ST_NOT_POSDEF);
}

// =====================================================================================================================
// Information-form smoother (smoothing runs).  The covariance-form forward step above needs two d×d inverses
// (V_p⁻¹ and Λ_f⁻¹); carrying the filtered belief as (ξ_f, Λ_f) needs ONE:
//     M_t = Λ_f(t) + A'P⁻¹A,   C_t = M_t⁻¹,   G_t = C_t (P⁻¹A)',   Λ_p(t+1) = P⁻¹ − (P⁻¹A) G_t,   ξ_p(t+1) = (P⁻¹A) C_t ξ_f(t)
// and C_t, G_t are exactly the residual V_f − G A V_f and the smoother gain V_f A'V_p⁻¹ the backward sweep wants
// (matrix inversion lemma), with  m_s(t) = C_t ξ_f(t) + G_t m_s(t+1),  V_s(t) = C_t + G_t V_s(t+1) G_t'.
// The Bethe free energy of the tree, −log p(y), is evaluated at the smoothed means x̂ (for a Gaussian posterior
// log p(y) = log p(x̂, y) − log q(x̂)):
//     F = ½[log|V1| + (T−1) log|P| + T(d_y log 2π + log|Q|)]                     (constant, host)
//       + ½[Σ_{t<T} log|M_t| + log|Λ_f(T)|]                                       (pivots of the inverses, forward / last init)
//       + ½[(x̂_1−m1)'V1⁻¹(x̂_1−m1) + Σ r_x'P⁻¹r_x + Σ r_y'Q⁻¹r_y]                  (residuals at the smoothed means, backward)
// Record of time index t: ξ_f(t) | (spare) | C_t (lower tiles) | G_t' (accumulator order).  vend: Λ_f at the segment end.
// fe_part slots (negated contributions): 0 and S+s: backward of segment 0 / s ≥ 1;  1+s: forward of segment s.
// STEPM: per-step constants (masked schedule only) — the constants of the transition INTO step t and of the observation at t come from
// block step_model[t]: M_t = [P⁻¹ + B′Q⁻¹B]_t + [A′P⁻¹A]_{t+1} − K_t G_{t−1}.  A template flag: the fixed-model kernel keeps its registers.
#ifdef RXHIP_TEST_SEEDCOUNT   // per-phase cycle counters of kd_forward_info (debug build only), lane 0 of every wave
#define RXHIP_PH(i) do { if constexpr (SEEDED) { const long long tn_ = __builtin_readcyclecounter(); if (lane == 0) cpp[4 * dm + w * DiagSeed::LDS_DOUBLES + 260 + (i)] += (double)(tn_ - tph_); tph_ = tn_; } } while (0)
#else
#define RXHIP_PH(i) do { } while (0)
#endif
template <int NT, bool FE, bool STEPM = false>
#ifndef RXHIP_FWD_WAVES
#define RXHIP_FWD_WAVES 2
#endif
__global__ void __launch_bounds__(64 * NT, RXHIP_FWD_WAVES) kd_forward_info(DenseParams p) {  // ≥ 2 waves per SIMD: ≤ 256 registers
    constexpr int D = 16 * NT;
    using C = DenseCfg<NT>;
    constexpr int LD = C::LD;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int dy = p.dy, tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int dm = ((D > dy ? D : dy) + 1) & ~1;
    // Two LDS matrices only (75 KB at d = 64): TWO workgroups share a CU, so that one's publish → barrier → cofactor → MFMA
    // latency chain runs under the other's (the kernel holds 255 registers: two waves per SIMD fit).  The constant
    // PLW = P⁻¹ + B'Q⁻¹B + A'P⁻¹A is re-read from L2 into dead registers a whole contraction before its use.
    double* S0 = smem;          // scratch of the inverse, then −G_t
    double* S1 = S0 + C::MAT;   // C_t, then M_{t+1} for the symmetrisation
    double* vec = S1 + C::MAT;
    double* xi = vec;           // ξ_f
    double* u = xi + dm;
    double* xpp = u + dm;       // [4][D] partial sums of ξ_p
    double* cpp = xpp + 4 * dm; // [4][D] partial sums of C_t ξ_f(t) (handed to the backward kernel in the record)
    // scratch of the SPD inverse: S0 is dead while it runs (−G_{t−1}: every reader is behind the barrier that precedes the
    // store of M), so at d ≥ 48 the panel buffers live there and two workgroups still share a CU
    double* rowbuf = DenseLds<NT>::ALIAS ? S0 : cpp + 4 * dm;
    const long long seg = blockIdx.x, chain = blockIdx.y + p.chain0;
    const DenseModel M = dense_model(p, chain);  // this chain's model tables
    const DenseCst c = DenseCst::make(D, dy);
    const double* cst = M.cst;
    const size_t MM = (size_t)D * D;
    const int grp = tid / D, gi = tid - grp * D;
    const long long b0 = 1 + seg * p.L;
    long long b1 = b0 + p.L;
    if (b1 > p.T) b1 = p.T;
    const long long len = b1 - b0, t0 = seg * p.L + 1;
    bool ok = true;
    LogProd lp;
    Acc<NT> lam, a;
    constexpr bool SEEDED = DenseLds<NT>::SEEDED && !STEPM;   // (per-step constants: every step's tile is another matrix)
    using SeedT = typename std::conditional<SEEDED, DiagSeed, NoSeed>::type;
    SeedT seed;
    if constexpr (SEEDED) {
        seed.x0 = cpp + 4 * dm + w * DiagSeed::LDS_DOUBLES;
        seed.use = p.mseg == 0;   // (masked sweeps: the same)
#ifdef RXHIP_TEST_SEEDCOUNT
        if (lane == 0) for (int q = 257; q < 264; ++q) seed.x0[q] = 0.0;
        if (lane == 0) for (int q = 0; q < 20; ++q) seed.x0[260 + q] = 0.0;
#endif
    }
    // K = P⁻¹A is the A operand of both contractions of a step.  Its fragments (16 doubles per thread at d = 64) are re-read from
    // L2 after every inverse (KF_RELOAD): held across the panel inverse they pushed the kernel over 256 registers, and every
    // scratch reload is a VMEM wait that also drains the record stores in flight.  The address is laundered per step so that
    // the loads are not hoisted out of the loop again.
    constexpr bool KF_RELOAD = RXHIP_KF_RELOAD && NT >= 3;
    double kf[D / 4];
    const double* cst_a = cst;   // STEPM: the block of the step being entered (set at the top of every iteration)
    auto step_cst = [&](long long t) { return p.cst + (size_t)p.step_model[t < p.T ? t : p.T - 1] * (size_t)p.cst_stride; };
    auto load_kf = [&]() {
        const double* K = (STEPM ? cst_a : cst) + c.oK + (16 * w + (lane & 15)) * D + (lane >> 4);
        if (KF_RELOAD) asm volatile("" : "+v"(K));
#pragma unroll
        for (int kk = 0; kk < D / 4; ++kk) kf[kk] = K[4 * kk];
    };
    if (!KF_RELOAD && !STEPM) load_kf();
    // symmetric contraction M_{t+1} = PLW − K G: slots of this wave (tile (w, (w + sl) mod NT), sl < nsw)
    constexpr int NS = NT / 2 + 1;
    const int ws = __builtin_amdgcn_readfirstlane(w);
    const int nsw = (NT % 2 == 0 && NT > 1 && ws >= NT / 2) ? NS - 1 : NS;
    auto slot_tile = [&](int sl) { const int t = ws + sl; return t >= NT ? t - NT : t; };
    // the equilibration exponents of the next inverse are published where M is stored (one barrier less per step)
    constexpr bool PREPUB = RXHIP_INV_BLOCKED != 0;
    auto mm_k = [&](Acc<NT>& cacc, const double* Y) {  // cacc += K Y   (Y in LDS, leading dimension LD)
        const int jl = lane & 15, kq = lane >> 4;
        d4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (d4){cacc.v[t][0], cacc.v[t][1], cacc.v[t][2], cacc.v[t][3]};
#pragma unroll
        for (int kk = 0; kk < D / 4; ++kk) {
            const int k = 4 * kk + kq;
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(kf[kk], Y[k * LD + 16 * t + jl], acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            cacc.v[t][0] = acc[t][0];
            cacc.v[t][1] = acc[t][1];
            cacc.v[t][2] = acc[t][2];
            cacc.v[t][3] = acc[t][3];
        }
    };
    // belief at the segment start in information form: Λ_f = V(b_s)⁻¹ (data-independent: host table), ξ_f = Λ_f m(b_s)
    acc_load<NT>(lam, p.mseg ? p.mbnd + (((size_t)chain * p.S + seg) * 2 + 0) * MM : M.bnd + ((size_t)seg * 2 + 0) * MM, D, w, lane);
    acc_store<NT>(lam, S1, LD, w, lane);
    if (tid < D) u[tid] = p.fstart_m[(chain * p.S + seg) * D + tid];
    lds_barrier();
    if (p.mseg == 2) { if (tid < D) xi[tid] = u[tid]; }   // masked sweeps hand over ξ_f(b_s) itself (dense_mseg_kernels.hpp)
    else matvec_lds(xi, S1, LD, D, D, u, nullptr, 0.0, tid);
    const double* cst_w = STEPM ? step_cst(t0) : cst;     // whose A′P⁻¹A sits in lam at the moment
    acc_add_mat<NT>(lam, cst_w + c.oW, D, w, lane, 1.0);  // lam carries M_t = Λ_f(t−1) + A'P⁻¹A from here on
    if (PREPUB) {  // equilibration exponents of the first inverse (later ones: where M is stored)
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (t == ws) blk_publish_exponents<NT>(lam.v[t], rowbuf, ws, lane);
    }
    // B'Q⁻¹ y_t comes from the aggregation kernel (record of t, second header slot), fetched one step ahead
    double gyn = (tid < D && len > 0) ? p.filt[(chain * p.T + t0) * C::REC + D + tid] : 0.0;
    double obn = (p.mseg && len > 0) ? p.obs[chain * p.T + t0] : 1.0;   // is y_t observed (masked sweeps), one step ahead like gyn
    d4 plw_r[NT <= 2 ? NS : 1], plwm_r[NT <= 2 ? NS : 1];
    if constexpr (NT <= 2 && !STEPM) {
        const int lq0 = lane >> 4, lj0 = lane & 15;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            const double* s0 = cst + c.oPLW + (size_t)(16 * ws + lq0) * D + 16 * slot_tile(sl) + lj0;
            const double* s1 = cst + c.oPLWM + (size_t)(16 * ws + lq0) * D + 16 * slot_tile(sl) + lj0;
            plw_r[sl] = (d4){s0[0], s0[4 * D], s0[8 * D], s0[12 * D]};
            plwm_r[sl] = (d4){s1[0], s1[4 * D], s1[8 * D], s1[12 * D]};
        }
    }
    lds_barrier();
#ifdef RXHIP_TEST_SEEDCOUNT
    long long tph_ = __builtin_readcyclecounter();
#endif
    // FROZEN (one set of constants, no masks): once M_{t+1} = M_t to rounding every matrix of a step repeats (C, G′, M, the determinants), and the
    // rest of the segment is the loop behind this one: vectors and record stores only.  Two tests, both passed on the same step (RXHIP_FZ_CONFIRM):
    //  (a) at the end of a step each LANE forms two weighted sums of its 4·NT entries of M_{t+1} as the next inverse is about to see them — equilibrated by
    //      the exact powers of two of M's diagonal, M_ij · 2^(h_i + h_j) with |·| ≲ 2 whatever the scales of the state's components — and compares
    //      them with its own sums of the step before: unchanged to 1.4e-14 (absolute, on that scale) in EVERY lane of the workgroup (256 pairs of functionals
    //      over 16 entries each: an entry that still moves by more than ≈ 3e-14 · sqrt(M_ii M_jj) per step is seen unless the other 15 entries of its lane cancel
    //      it in both sums);
    //  (b) two plain sums over the tiles of M_{t+1} unchanged to 2 ulp (rounds 3–4's only test: it sees what moves the largest entries, and nothing
    //      else — tests/test_fixed_point_adversarial_gpu.py).
    // (The first comparison of an interior segment is between the boundary table's M and the kernel's first iterate — two different summation orders — and
    //  usually fails (a); the second one, between two iterates of this loop, passes.)
    // At a contraction rate ρ of the recursion a frozen entry is within ≈ 3e-14 / (1 − ρ) · sqrt(M_ii M_jj) of its fixed point (include/rxhip.h
    // "Fixed-point exits").  Interior segments start on the fixed point (boundary table) and leave after two steps.
    constexpr bool FROZEN = RXHIP_FWD_FROZEN && SEEDED;
    // for the backward sweep: the first time index of this segment whose record holds the repeated matrices (default: beyond the segment) — in a
    // slot of the segment's first record that nothing else touches: tile (1, 0) of C, which no wave owns in the symmetric pairing (NT ≥ 3)
    constexpr int FZ_SLOT = C::HDR + (NT * 4) * 64;
    if (FROZEN && tid == 0 && p.mseg == 0 && len > 0) p.filt[(chain * p.T + t0) * C::REC + FZ_SLOT] = (double)(t0 + len);
    double* fz = cpp + 4 * dm + DenseLds<NT>::FWD_TAIL - DenseLds<NT>::FWD_FROZEN;   // [2·w], [2·w + 1]: this wave's functionals; [8], [9]: lp in front of the inverse
    double fzp1 = 0.0, fzp2 = 0.0, lzp1 = 0.0, lzp2 = 0.0;
    int fz_same = 0;
    long long i_frozen = len;
    for (long long i = 0; i < len; ++i) {
        const long long t = t0 + i;
        if constexpr (STEPM) {
            cst_a = step_cst(t);
            if (!KF_RELOAD) load_kf();
        }
        double* rec = p.filt + (chain * p.T + (t - 1)) * C::REC;
        RXHIP_PH(11);
        const double gyc = gyn;
        if (tid < D) {
            rec[tid] = xi[tid];  // ξ_f(t − 1)
            gyn = p.filt[(chain * p.T + (i + 1 < len ? t + 1 : t)) * C::REC + D + tid];
        }
        if constexpr (FROZEN && FE) {
            if (tid == 0) { fz[8] = lp.mant; fz[9] = (double)lp.expo; }   // the step's determinant factor = what the inverse below multiplies in
        }
        // C = (Λ_f + A'P⁻¹A)⁻¹.  The inverse's scratch is S0 (or its own carve): the next writer of S0 is the store of −G below,
        // behind a workgroup barrier, so the inverse ends without one.
#ifndef RXHIP_TEST_NOINV
        if (KF_RELOAD && RXHIP_KF_RELOAD == 1) ok = spd_inverse<NT, decltype(load_kf), false, !PREPUB, SeedT>(lam, rowbuf, w, lane, lp, load_kf, &seed) && ok;
        else ok = spd_inverse<NT, NoPrefetch, false, !PREPUB, SeedT>(lam, rowbuf, w, lane, lp, NoPrefetch(), &seed) && ok;
#endif
        if (KF_RELOAD && RXHIP_KF_RELOAD == 2) load_kf();
        RXHIP_PH(0);   // the inverse
        // C_t goes to the record as the backward kernel reads it: the tiles this wave owns in the symmetric pairing (below) — the
        // accumulator of V_s(t) = C_t + H G' there is computed for the same tiles only (10 of 16 at d = 64)
#pragma unroll
        for (int t2 = 0; t2 < NT; ++t2) {
            int dist = t2 - ws;
            dist = dist < 0 ? dist + NT : dist;
            if (dist < NS - 1 || (dist == NS - 1 && dist < nsw)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) rec[C::HDR + ((w * NT + t2) * 4 + r) * 64 + lane] = lam.v[t2][r];
            }
        }
        acc_store<NT>(lam, S1, LD, w, lane);   // every reader of S1 (the symmetrisation) is behind the inverse's barriers
        int ln = lane;
        asm volatile("" : "+v"(ln));   // addresses below are recomputed per step, not hoisted into (spilled) registers
        const int lq = ln >> 4, lj = ln & 15;
        // PLW tiles of this wave's share of M_{t+1} (below): the L2 latency hides under G' = K C
        // y_t missing: no `*`_B(:in) message, Λ_f(t) = Λ_p(t) — the constant without B'Q⁻¹B (a second table, selected by a flag that
        // was fetched a step ahead: subtracting the tiles here was eight dependent round trips on every missing step)
        const bool miss = obn == 0.0;
        if (p.mseg) obn = p.obs[chain * p.T + (i + 1 < len ? t + 1 : t)];
        d4 macc[NS];
        if constexpr (STEPM) {     // [P⁻¹ + B′Q⁻¹B + A′P⁻¹A]_t − [A′P⁻¹A]_t + [A′P⁻¹A]_{t+1}
            const double* cb = step_cst(t + 1);
            const double* plw = cst_a + (miss ? c.oPLWM : c.oPLW);
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) {
                const size_t o = (size_t)(16 * ws + lq) * D + 16 * slot_tile(sl) + lj;
                const double *s0 = plw + o, *s1 = cst_a + c.oW + o, *s2 = cb + c.oW + o;
                macc[sl] = (d4){s0[0] - s1[0] + s2[0], s0[4 * D] - s1[4 * D] + s2[4 * D], s0[8 * D] - s1[8 * D] + s2[8 * D], s0[12 * D] - s1[12 * D] + s2[12 * D]};
            }
            cst_w = cb;
        } else if constexpr (NT <= 2) {   // narrow models: both constants live in registers (G' = K C is too short to hide an L2 round trip per step)
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) macc[sl] = miss ? plwm_r[sl] : plw_r[sl];
        } else {
            const double* plw = cst + (miss ? c.oPLWM : c.oPLW);
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) {
                const double* src = plw + (size_t)(16 * ws + lq) * D + 16 * slot_tile(sl) + lj;
                macc[sl] = (d4){src[0], src[4 * D], src[8 * D], src[12 * D]};
            }
        }
        lds_barrier();
        // G' = K C
        acc_zero<NT>(a);
        RXHIP_PH(1);   // record stores of C, C -> S1, PLW loads, barrier
        mm_k(a, S1);
        acc_store_full<NT>(a, rec + C::HDR + D * D, w, lane);
        RXHIP_PH(2);   // G' = K C + its record stores (issue)
#pragma unroll
        for (int q = 0; q < NT; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) a.v[q][r] = -a.v[q][r];
        acc_store_T<NT>(a, S0, LD, w, lane);  // S0 = −G
        lds_barrier();
        RXHIP_PH(3);   // -G -> S0, barrier
        // ξ_p = K C ξ_f = G' ξ_f: column sums of S0, a quarter of the range per thread group
        {
            double s0 = 0.0, s1 = 0.0, c0 = 0.0, c1 = 0.0;
            const int k0 = grp * (D / 4);
#pragma unroll
            for (int k = 0; k < D / 4; k += 2) {
                s0 += S0[(k0 + k) * LD + gi] * xi[k0 + k];
                s1 += S0[(k0 + k + 1) * LD + gi] * xi[k0 + k + 1];
                c0 += S1[(k0 + k) * LD + gi] * xi[k0 + k];          // C is symmetric: column sums are conflict-free
                c1 += S1[(k0 + k + 1) * LD + gi] * xi[k0 + k + 1];
            }
            xpp[grp * D + gi] = s0 + s1;
            cpp[grp * D + gi] = c0 + c1;
        }
        RXHIP_PH(4);   // column sums
        // M_{t+1} = Λ_f(t) + A'P⁻¹A = PLW − K G is symmetric: every unordered pair of tile indices is computed ONCE, by the wave
        // that owns the pair (tile (w, t) with (t − w) mod NT ≤ NT/2 — 3, 3, 2, 2 tiles per wave at d = 64 instead of 4), and the
        // mirror image is read back transposed below: 48 instead of 64 MFMAs on the critical wave, exact symmetry for free
        // (only the diagonal tiles still average with their own transpose).
        if (nsw == NS) {
#pragma unroll
            for (int kk = 0; kk < D / 4; ++kk)
#pragma unroll
                for (int sl = 0; sl < NS; ++sl)
                    macc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(kf[kk], S0[(4 * kk + lq) * LD + 16 * slot_tile(sl) + lj], macc[sl], 0, 0, 0);
        } else {
#pragma unroll
            for (int kk = 0; kk < D / 4; ++kk)
#pragma unroll
                for (int sl = 0; sl < NS - 1; ++sl)
                    macc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(kf[kk], S0[(4 * kk + lq) * LD + 16 * slot_tile(sl) + lj], macc[sl], 0, 0, 0);
        }
        lds_barrier();
        RXHIP_PH(5);   // M' contraction, barrier
        if (tid < D) {
            rec[2 * D + tid] = (cpp[tid] + cpp[D + tid]) + (cpp[2 * D + tid] + cpp[3 * D + tid]);      // C_{t−1} ξ_f(t−1)
            xi[tid] = gyc - ((xpp[tid] + xpp[D + tid]) + (xpp[2 * D + tid] + xpp[3 * D + tid]));       // ξ_f(t)
        }
        // C is no longer needed: S1 carries the computed tiles of M; the exponents of its diagonal (equilibration of the next
        // inverse) share the barrier
#pragma unroll
        for (int sl = 0; sl < NS; ++sl)
            if (sl < nsw) {
                double* dst = S1 + (16 * ws + lq) * LD + 16 * slot_tile(sl) + lj;
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[4 * r * LD] = macc[sl][r];
            }
        if (PREPUB) {
            const double dtile[4] = {macc[0][0], macc[0][1], macc[0][2], macc[0][3]};
            blk_publish_exponents<NT>(dtile, rowbuf, ws, ln);
        }
        if constexpr (FROZEN) {
            if (p.mseg == 0) {
                double f1 = 0.0, f2 = 0.0;
#pragma unroll
                for (int sl = 0; sl < NS; ++sl)
                    if (sl < nsw) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            f1 += macc[sl][r];
                            f2 += (1.0 + 0.37 * (4 * sl + r) + 0.011 * ln) * macc[sl][r];
                        }
                    }
                f1 = row16_sum(f1);
                f2 = row16_sum(f2);
                f1 = (rd_lane(f1, 0) + rd_lane(f1, 16)) + (rd_lane(f1, 32) + rd_lane(f1, 48));
                f2 = (rd_lane(f2, 0) + rd_lane(f2, 16)) + (rd_lane(f2, 32) + rd_lane(f2, 48));
                if (ln == 0) { fz[2 * ws] = f1; fz[2 * ws + 1] = f2; }
            }
        }
        lds_barrier();
        RXHIP_PH(6);   // xi, M' -> S1, exponents, barrier
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            int dist = t - ws;
            dist = dist < 0 ? dist + NT : dist;
            const bool owned = dist < NS - 1 || (dist == NS - 1 && dist < nsw);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ws + lq + 4 * r, col = 16 * t + lj;
                const double v = S1[owned ? row * LD + col : col * LD + row];
                lam.v[t][r] = (t == ws) ? 0.5 * (v + S1[col * LD + row]) : v;
            }
        }
        RXHIP_PH(7);   // symmetrisation reads
        if constexpr (FROZEN) {
            if (p.mseg == 0) {   // test (a) on M_{t+1} (`lam`, symmetrised): the exponents of its diagonal were published for the next inverse (natural order | row q + 4r at 4q + r)
                const int* sc = reinterpret_cast<const int*>(rowbuf + 2 * blk_half_doubles(NT));
                const int4 hr = *reinterpret_cast<const int4*>(sc + 16 * NT + 16 * ws + 4 * (lane >> 4));
                double f1 = 0.0, f2 = 0.0;
#pragma unroll
                for (int t2 = 0; t2 < NT; ++t2) {
                    const int hc = sc[16 * t2 + (lane & 15)];
                    const double s0 = __builtin_ldexp(lam.v[t2][0], hr.x + hc), s1 = __builtin_ldexp(lam.v[t2][1], hr.y + hc),
                                 s2 = __builtin_ldexp(lam.v[t2][2], hr.z + hc), s3 = __builtin_ldexp(lam.v[t2][3], hr.w + hc);
                    f1 += (s0 + s1) + (s2 + s3);
                    f2 = fma(fma(fma(fma(f2, 1.37, s0), 1.37, s1), 1.37, s2), 1.37, s3);   // weights 1.37^k: ONE constant in a register, not sixteen
                }
                // (absolute bounds: the equilibrated entries live on the scale of a diagonal of order one, and so does their rounding noise — a lane off the
                //  diagonal sums sixteen entries of either sign to next to nothing, and a bound relative to that sum never passes)
                const bool lane_same = fabs(f1 - lzp1) <= RXHIP_FZ_LANE_TOL && fabs(f2 - lzp2) <= 26.0 * RXHIP_FZ_LANE_TOL;   // Σ 1.37^k = 26 · 16
                lzp1 = f1;
                lzp2 = f2;
                const bool wave_same = __all(lane_same) != 0;
                if (lane == 0) fz[10 + ws] = wave_same ? 1.0 : 0.0;   // (an invalid exponent gives inf / NaN: never "same")
                lds_barrier();
            }
        }
        if constexpr (FROZEN) {
            if (p.mseg == 0) {
                double g1 = 0.0, g2 = 0.0;
#pragma unroll
                for (int q = 0; q < NT; ++q) {
                    g1 += fz[2 * q];
                    g2 += fz[2 * q + 1];
                }
                const bool same = fabs(g1 - fzp1) <= 4.5e-16 * fabs(g1) && fabs(g2 - fzp2) <= 4.5e-16 * fabs(g2) && (fz[10] + fz[11]) + (fz[12] + fz[13]) == (double)NT && !p.no_frozen;
                fzp1 = g1;
                fzp2 = g2;
                fz_same = same ? fz_same + 1 : 0;
                if (fz_same >= RXHIP_FZ_CONFIRM && i + 1 < len) {   // (the same numbers in every thread of the workgroup)
                    i_frozen = i + 1;
                    break;
                }
            }
        }
    }
    if constexpr (FROZEN) {
        if (i_frozen < len) {
            // The matrices of step i_frozen − 1, from its record (L2): C in the owned tiles, G′ whole; S1 gets C back, S0 −G (M sits in `lam`)
            const long long tl = t0 + i_frozen - 1;
            const double* recl = p.filt + (chain * p.T + (tl - 1)) * C::REC;
            if (tid == 0) p.filt[(chain * p.T + t0) * C::REC + FZ_SLOT] = (double)(tl - 1);
            double ct[NS][4], gp[NT][4];
#pragma unroll
            for (int sl = 0; sl < NS; ++sl)
                if (sl < nsw) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) ct[sl][r] = recl[C::HDR + ((w * NT + slot_tile(sl)) * 4 + r) * 64 + lane];
                }
#pragma unroll
            for (int t2 = 0; t2 < NT; ++t2)
#pragma unroll
                for (int r = 0; r < 4; ++r) gp[t2][r] = recl[C::HDR + D * D + ((w * NT + t2) * 4 + r) * 64 + lane];
            const int lq = lane >> 4, lj = lane & 15;
            lds_barrier();   // the symmetrisation reads of S1 are done
#pragma unroll
            for (int sl = 0; sl < NS; ++sl)
                if (sl < nsw) {
                    const int tt = slot_tile(sl);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * ws + lq + 4 * r, col = 16 * tt + lj;
                        S1[row * LD + col] = ct[sl][r];
                        S1[col * LD + row] = ct[sl][r];   // the mirror image (a diagonal tile: its own transpose, the same values to rounding)
                    }
                }
            {   // … and −G again: the exponents of the next inverse were published into S0's lines (the scratch of the inverse lives there)
                Acc<NT> ng;
#pragma unroll
                for (int t2 = 0; t2 < NT; ++t2)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ng.v[t2][r] = -gp[t2][r];
                acc_store_T<NT>(ng, S0, LD, w, lane);
            }
            // the determinant factor of a step, and this wave's share of the seeded traces per step
            double fm = 1.0;
            long long fe2 = 0;
            if constexpr (FE) {
                if (tid == 0) {
                    fm = lp.mant / fz[8];
                    fe2 = lp.expo - (long long)fz[9];
                }
            }
            lds_barrier();
            for (long long i = i_frozen; i < len; ++i) {
                const long long t = t0 + i;
                double* rec = p.filt + (chain * p.T + (t - 1)) * C::REC;
                const double gyc = gyn;
                if (tid < D) {
                    rec[tid] = xi[tid];
                    gyn = p.filt[(chain * p.T + (i + 1 < len ? t + 1 : t)) * C::REC + D + tid];
                }
                // (the records of a frozen stretch carry their VECTORS only: C and G′ are the ones of record FZ_SLOT, and every reader of a sweep's records —
                //  kd_backward_info, kd_cross_from_records — goes there for them; p.full_records: a reader that indexes records by time alone follows)
                if (p.full_records) {   // (workgroup-uniform)
#pragma unroll
                    for (int sl = 0; sl < NS; ++sl)
                        if (sl < nsw) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) rec[C::HDR + ((w * NT + slot_tile(sl)) * 4 + r) * 64 + lane] = ct[sl][r];
                        }
#pragma unroll
                    for (int t2 = 0; t2 < NT; ++t2)
#pragma unroll
                        for (int r = 0; r < 4; ++r) rec[C::HDR + D * D + ((w * NT + t2) * 4 + r) * 64 + lane] = gp[t2][r];
                }
                {
                    double s0 = 0.0, s1 = 0.0, c0 = 0.0, c1 = 0.0;
                    const int k0 = grp * (D / 4);
#pragma unroll
                    for (int k = 0; k < D / 4; k += 2) {
                        s0 += S0[(k0 + k) * LD + gi] * xi[k0 + k];
                        s1 += S0[(k0 + k + 1) * LD + gi] * xi[k0 + k + 1];
                        c0 += S1[(k0 + k) * LD + gi] * xi[k0 + k];
                        c1 += S1[(k0 + k + 1) * LD + gi] * xi[k0 + k + 1];
                    }
                    xpp[grp * D + gi] = s0 + s1;
                    cpp[grp * D + gi] = c0 + c1;
                }
                lds_barrier();
                if (tid < D) {
                    rec[2 * D + tid] = (cpp[tid] + cpp[D + tid]) + (cpp[2 * D + tid] + cpp[3 * D + tid]);
                    xi[tid] = gyc - ((xpp[tid] + xpp[D + tid]) + (xpp[2 * D + tid] + xpp[3 * D + tid]));
                }
                if constexpr (FE) {
                    if (tid == 0) {
                        lp.mul(fm);
                        lp.expo += fe2;
                    }
                    seed.a += seed.r;
                }
                lds_barrier();
            }
        }
    }
    acc_add_mat<NT>(lam, cst_w + c.oW, D, w, lane, -1.0);
    acc_store_tri<NT>(lam, p.vend + (chain * p.S + seg) * C::TRI, w, lane);  // Λ_f at the segment end
    if (seg == p.S - 1 && tid < D) p.filt[(chain * p.T + (t0 + len - 1)) * C::REC + tid] = xi[tid];  // ξ_f(T): no successor writes it
    double ldet = 0.0;
#ifdef RXHIP_TEST_SEEDCOUNT
    if constexpr (SEEDED) {
        if (lane == 0 && seg == 100) printf("  wave %d phases (cycles per step): top %.0f | inverse %.0f | C stores+barrier %.0f | G'=KC %.0f | -G store+barrier %.0f | colsums %.0f | M' contraction+barrier %.0f | xi, M' store, barrier %.0f | symmetrise %.0f\n", w,
            seed.x0[271] / len, seed.x0[260] / len, seed.x0[261] / len, seed.x0[262] / len, seed.x0[263] / len, seed.x0[264] / len, seed.x0[265] / len, seed.x0[266] / len, seed.x0[267] / len);
        if (lane == 0 && seg == 100) printf("  wave %d inverse (cycles per step): equilibrate %.0f | before the barrier (publish, own tile inverse; else nothing) %.0f | in the barrier %.0f | after it (trailing update) %.0f | scale back %.0f\n", w,
            seed.x0[272] / len, seed.x0[273] / len, seed.x0[274] / len, seed.x0[275] / len, seed.x0[276] / len);
        if (tid == 0 && (seg < 40 || seg % 50 == 0)) printf("seg %d: frozen from step %d of %d\n", (int)seg, (int)i_frozen, (int)len);
        if (lane == 0 && (seg % 50 == 0 || seg == 1 || seg == 2 || seg == 3)) printf("seg %d wave %d: seeded %d of %d steps, %.0f cycles per seeded tile inverse, %.0f per exact one\n", (int)seg, w, (int)seed.x0[257], (int)len, seed.x0[258] / fmax(1.0, seed.x0[257]), seed.x0[259] / fmax(1.0, (double)len - seed.x0[257]));
    }
#endif
    if constexpr (SEEDED && FE) {   // the seeded steps' share of Σ_t log det M_t: lane partials → one number per wave → thread 0
        double sa = row16_sum(seed.a);
        sa = (rd_lane(sa, 0) + rd_lane(sa, 16)) + (rd_lane(sa, 32) + rd_lane(sa, 48));
        lds_barrier();
        if (lane == 0) xpp[w] = sa;
        lds_barrier();
        if (tid == 0)
            for (int q = 0; q < NT; ++q) ldet += xpp[q];
    }
    if (FE && tid == 0) dense_fe_write(p, 1 + seg, chain, lp.value() + ldet, 0.0, 0.0);
    if (!ok && tid == 0) atomicOr(p.status, ST_NOT_POSDEF);
}

template <int NT, bool FE>
__global__ void __launch_bounds__(64 * NT, 2) kd_backward_info(DenseParams p) {  // ≥ 2 waves per SIMD: ≤ 256 registers
    constexpr int D = 16 * NT;
    using C = DenseCfg<NT>;
    constexpr int LD = C::LD;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int dy = p.dy, tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int dm = ((D > dy ? D : dy) + 1) & ~1;
    // Two LDS matrices (73 KB at d = 64), so that two workgroups share a CU as in the forward kernel:
    //   MV  V_s(t+1) as the B operand of H = G V_s; then every wave parks ITS OWN 16 rows of H there to re-read them as the A
    //       operand of V_s(t) = C + H G' (accumulator layout -> operand layout is a wave-local exchange);
    //   MG  G_t' exactly as the record holds it (MG[k][j] = G[j][k]): the B operand of the second contraction, the A operand
    //       of the first one (read transposed), and — by columns, conflict-free — the matvec G m_s.
    // C_t never touches LDS: the record carries it in accumulator order (it is the accumulator's initial value) together
    // with C_t ξ_f(t), formed in the forward kernel where both operands sit in LDS anyway.
    double* MV = smem;
    double* MG = MV + C::MAT;
    double* vec = MG + C::MAT;
    double* ms = vec;           // m_s(t+1), then m_s(t)
    double* xf = ms + dm;       // boundary: ξ_f + ξβ; in the loop: C_t ξ_f(t)
    double* rowbuf = xf + dm;   // the matvec partials (and, where no matrix can be lent, the scratch of the last segment's inverse)
    double* invbuf = DenseLds<NT>::ALIAS ? MG : rowbuf;  // MG is not in use before the first commit()
    const long long seg = blockIdx.x, chain = blockIdx.y + p.chain0;
    const DenseModel M = dense_model(p, chain);  // this chain's model tables
    const DenseCst c = DenseCst::make(D, dy);
    const double* cst = M.cst;
    const size_t MM = (size_t)D * D;
    const int grp = tid / D, gi = tid - grp * D;
    const long long b0 = 1 + seg * p.L;
    long long b1 = b0 + p.L;
    if (b1 > p.T) b1 = p.T;
    const long long len = b1 - b0, tb = seg * p.L, te = tb + len;
    bool ok = true;
    LogProd lpe;
    Acc<NT> a;
    // smoothed belief at the end boundary: V_s = (Λ_f + Λβ)⁻¹, m_s = V_s (ξ_f + ξβ).  Inner boundaries: V_s does not depend
    // on the data (host table).  Last segment: Λβ = 0, V_s = Λ_f(T)⁻¹ from the forward sweep; its pivots give |Λ_f(T)|.
    {
        if (tid < D) xf[tid] = p.filt[(chain * p.T + te) * C::REC + tid] + p.beta_xi[(chain * (p.S + 1) + seg + 1) * D + tid];
        if (seg == p.S - 1) {  // uniform over the workgroup
            tri_to_lds<NT>(p.vend + (chain * p.S + seg) * C::TRI, MV, LD, w, lane);
            lds_barrier();
            acc_load<NT>(a, MV, LD, w, lane);
            lds_barrier();  // every wave has its rows in registers before MV is overwritten below
            ok = spd_inverse<NT>(a, invbuf, w, lane, lpe) && ok;
            if (p.vlast && chain == 0) acc_store<NT>(a, p.vlast, D, w, lane);
        } else
            acc_load<NT>(a, p.mseg ? p.mbnd + (((size_t)chain * p.S + seg) * 2 + 1) * MM : M.bnd + ((size_t)seg * 2 + 1) * MM, D, w, lane);
        acc_store<NT>(a, MV, LD, w, lane);
        lds_barrier();
        matvec_lds(ms, MV, LD, D, D, xf, nullptr, 0.0, tid);
        lds_barrier();
        if (seg == p.S - 1) {
            if (tid < D) dense_store_mean(p, te, chain, tid, ms[tid]);
            dense_store_cov<NT>(p, a, te, chain, w, lane);
        }
    }
    Acc<NT> gN, cc;
    double cxN = 0.0;
    // V_s(t) = C_t + H G' is symmetric: every unordered pair of tile indices is computed once, by the wave that owns it (the pairing
    // of kd_forward_info: 3, 3, 2, 2 tiles per wave at d = 64 — 48 instead of 64 MFMAs on the critical wave, and only those tiles
    // of C_t are read from the record: −19 % of the record traffic of a kernel that moves 3.9 TB/s), written to MV with its mirror
    // image; every wave then reads its full tile row back for the posterior store.
    constexpr int NS = NT / 2 + 1;
    const int ws = __builtin_amdgcn_readfirstlane(w);
    const int nsw = (NT % 2 == 0 && NT > 1 && ws >= NT / 2) ? NS - 1 : NS;
    auto slot_tile = [&](int sl) { const int t2 = ws + sl; return t2 >= NT ? t2 - NT : t2; };
    // the record that carries the MATRICES of time index tt: its own, or — behind the point where the forward sweep found its matrices repeating
    // (tfz, set below) — the one record of the segment that holds them (kd_forward_info's frozen loop writes vectors only)
    long long tfz = te + 1;
    auto mrec = [&](long long tt) { return p.filt + (chain * p.T + (tt > tfz ? tfz : tt)) * C::REC; };
    auto prefetch = [&](long long tt) {   // G_tt' and C_tt ξ_f(tt) of the NEXT step travel under the current one
        acc_load_full<NT>(gN, mrec(tt) + C::HDR + D * D, w, lane);
        if (tid < D) cxN = p.filt[(chain * p.T + tt) * C::REC + 2 * D + tid];
    };
    if constexpr (RXHIP_FWD_FROZEN && DenseLds<NT>::ALIAS) {
        if (p.mseg == 0 && !p.step_model && len > 0) tfz = (long long)p.filt[(chain * p.T + (tb + 1)) * C::REC + C::HDR + (NT * 4) * 64];   // (kd_forward_info: FZ_SLOT)
    }
    auto commit = [&]() {
        acc_store<NT>(gN, MG, LD, w, lane);
        if (tid < D) xf[tid] = cxN;
    };
    if (te - 1 >= tb) {
        prefetch(te - 1);
        commit();
    }
    lds_barrier();
    bool pending = false;       // the posterior of the previous step is still in MV / ms: stored at the top of the next iteration
    long long tprev = 0;
    auto flush_posterior = [&]() {
        acc_load<NT>(cc, MV, LD, w, lane);
        if (tid < D) dense_store_mean(p, tprev, chain, tid, ms[tid]);
        dense_store_cov<NT>(p, cc, tprev, chain, w, lane);
    };
    // BFROZEN (d ≥ 48, one set of constants, no masks): where the forward sweep found its matrices repeating (records tfz … te − 1 hold the same C and G′)
    // and V_s has stopped moving as well, a step is the mean recursion and the stores: m_s(t) = C ξ_f(t) + G m_s(t+1) against the G′ that sits in MG,
    // V_s(t) = the tile row already in registers.  "Stopped moving" has two stages: two functionals of V_s's tiles unchanged to 2 ulp (every step; a
    // filter — plain sums see only what moves the largest entries), then on the following step V_s(t) against V_s(t + 1) ENTRY BY ENTRY,
    // |ΔV_ij| ≤ RXHIP_FZ_TOL · sqrt(V_ii V_jj): the new matrix from MV, the old one from the posterior array, where this very lane stored these entries
    // at the top of the step.  One failing entry in any wave and the sweep goes on in full steps.
    constexpr bool BFROZEN = RXHIP_FWD_FROZEN && DenseLds<NT>::ALIAS;
    double* bz = rowbuf + 4 * D;   // [2·w], [2·w + 1]: this wave's functionals of V_s
    double bzp1 = 0.0, bzp2 = 0.0;
    int bz_same = 0;
    bool bz_verify = false, bz_ok = false;   // (workgroup-uniform) this step ends with the entrywise comparison / it has passed
    const long long tfz_b = p.no_frozen ? te + 1 : tfz;   // where THIS sweep may stop recomputing V_s (test hook: nowhere)
    for (long long t = te - 1; t >= tb; --t) {
        if constexpr (BFROZEN) {
            if (bz_ok && t >= tfz_b && t > tb) {   // (workgroup-uniform) a frozen stretch: t … max(tfz, tb + 1)
                const long long tstop = tfz_b > tb + 1 ? tfz_b : tb + 1;
                // MV holds V_s (complete since the barrier that closed the last step), MG the G′ committed for step t, xf = C ξ_f(t)
                acc_load<NT>(cc, MV, LD, w, lane);
                if (pending) {
                    if (tid < D) dense_store_mean(p, tprev, chain, tid, ms[tid]);
                    dense_store_cov<NT>(p, cc, tprev, chain, w, lane);
                    pending = false;
                }
                double cxn = 0.0;
                for (; t >= tstop; --t) {
                    if (tid < D) cxn = p.filt[(chain * p.T + (t - 1)) * C::REC + 2 * D + tid];   // C ξ_f(t − 1) for the next step
                    {
                        const int k0 = grp * (D / 4);
                        double s0 = 0.0, s1 = 0.0;
#pragma unroll
                        for (int k = 0; k < D / 4; k += 2) {
                            s0 += MG[(k0 + k) * LD + gi] * ms[k0 + k];
                            s1 += MG[(k0 + k + 1) * LD + gi] * ms[k0 + k + 1];
                        }
                        rowbuf[grp * D + gi] = s0 + s1;
                    }
                    lds_barrier();
                    if (tid < D) {
                        const double mnew = xf[tid] + ((rowbuf[tid] + rowbuf[D + tid]) + (rowbuf[2 * D + tid] + rowbuf[3 * D + tid]));
                        ms[tid] = mnew;
                        xf[tid] = cxn;
                        dense_store_mean(p, t, chain, tid, mnew);
                    }
                    dense_store_cov<NT>(p, cc, t, chain, w, lane);
                    lds_barrier();
                }
                // back to full steps at t = tstop − 1 (≥ tb): its G′ into MG (xf already holds C ξ_f of that step)
                bz_same = 0;
                bz_ok = bz_verify = false;
                acc_load_full<NT>(gN, mrec(t) + C::HDR + D * D, w, lane);
                acc_store<NT>(gN, MG, LD, w, lane);
                lds_barrier();
            }
        }
        prefetch(t - 1 >= tb ? t - 1 : tb);
        // C_t (owned tiles): the accumulators of the second contraction; their L2 / HBM latency hides under the first one
        d4 vacc[NS];
        {
            const double* rec = mrec(t) + C::HDR;
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) {
                const double* src = rec + ((w * NT + slot_tile(sl)) * 4) * 64 + lane;
                vacc[sl] = (sl < nsw) ? (d4){src[0], src[64], src[128], src[192]} : (d4){0.0, 0.0, 0.0, 0.0};
            }
        }
        if (pending) flush_posterior();   // V_s(t+1), m_s(t+1): complete in MV / ms since the barrier that closed the last iteration
        // G m_s(t+1): thread group `grp` sums a quarter of the k range down the columns of MG (consecutive threads read
        // consecutive addresses)
        {
            const int k0 = grp * (D / 4);
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int k = 0; k < D / 4; k += 2) {
                s0 += MG[(k0 + k) * LD + gi] * ms[k0 + k];
                s1 += MG[(k0 + k + 1) * LD + gi] * ms[k0 + k + 1];
            }
            rowbuf[grp * D + gi] = s0 + s1;
        }
        // H = G V_s
        acc_zero<NT>(a);
        mm_acc<NT, true, false>(a, MG, LD, MV, LD, w, lane);
        lds_barrier();  // every wave is through with V_s (and has its rows of it in registers for the store); the matvec partials are complete
        double mnew = 0.0;
        if (tid < D) mnew = xf[tid] + ((rowbuf[tid] + rowbuf[D + tid]) + (rowbuf[2 * D + tid] + rowbuf[3 * D + tid]));
        acc_store<NT>(a, MV, LD, w, lane);  // this wave's rows of H; only this wave reads them back (LDS is in order per wave)
        // V_s(t) = C + H G'  (owned tiles)
        {
            const int i = 16 * w + (lane & 15), kq = lane >> 4, jl = lane & 15;
            if (nsw == NS) {
#pragma unroll
                for (int kk = 0; kk < D / 4; ++kk) {
                    const double av = MV[i * LD + 4 * kk + kq];
#pragma unroll
                    for (int sl = 0; sl < NS; ++sl)
                        vacc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, MG[(4 * kk + kq) * LD + 16 * slot_tile(sl) + jl], vacc[sl], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < D / 4; ++kk) {
                    const double av = MV[i * LD + 4 * kk + kq];
#pragma unroll
                    for (int sl = 0; sl < NS - 1; ++sl)
                        vacc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, MG[(4 * kk + kq) * LD + 16 * slot_tile(sl) + jl], vacc[sl], 0, 0, 0);
                }
            }
        }
        lds_barrier();  // every wave is through with G' (and with its rows of H)
        // V_s(t) into MV: the owned tiles and their mirror images
#pragma unroll
        for (int sl = 0; sl < NS; ++sl)
            if (sl < nsw) {
                const int ts = slot_tile(sl);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * ws + (lane >> 4) + 4 * r, col = 16 * ts + (lane & 15);
                    MV[row * LD + col] = vacc[sl][r];
                    if (ts != ws) MV[col * LD + row] = vacc[sl][r];
                }
            }
        if constexpr (BFROZEN) {
            if (t > tfz_b) {   // (a frozen stretch can only cover steps t − 1 ≥ tfz: below it the tests have nothing to decide)
                double f1 = 0.0, f2 = 0.0;
#pragma unroll
                for (int sl = 0; sl < NS; ++sl)
                    if (sl < nsw) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            f1 += vacc[sl][r];
                            f2 += (1.0 + 0.37 * (4 * sl + r) + 0.011 * lane) * vacc[sl][r];
                        }
                    }
                f1 = row16_sum(f1);
                f2 = row16_sum(f2);
                f1 = (rd_lane(f1, 0) + rd_lane(f1, 16)) + (rd_lane(f1, 32) + rd_lane(f1, 48));
                f2 = (rd_lane(f2, 0) + rd_lane(f2, 16)) + (rd_lane(f2, 32) + rd_lane(f2, 48));
                if (lane == 0) { bz[2 * ws] = f1; bz[2 * ws + 1] = f2; }
            }
        }
        if (tid < D) ms[tid] = mnew;
        commit();
        lds_barrier();
        if constexpr (BFROZEN) {
            if (t > tfz_b) {   // (a frozen stretch can only cover steps t − 1 ≥ tfz: below it the tests have nothing to decide)
                double g1 = 0.0, g2 = 0.0;
#pragma unroll
                for (int q = 0; q < NT; ++q) {
                    g1 += bz[2 * q];
                    g2 += bz[2 * q + 1];
                }
                const bool same = fabs(g1 - bzp1) <= 4.5e-16 * fabs(g1) && fabs(g2 - bzp2) <= 4.5e-16 * fabs(g2);
                bzp1 = g1;
                bzp2 = g2;
                bz_same = same ? bz_same + 1 : 0;
                bz_ok = false;
                if (bz_verify) {
                    // MV holds V_s(t) whole (the barrier above); the posterior array holds V_s(t + 1): entry (16w + lane/16 + 4r, 16·t2 + lane%16) was
                    // stored by this lane (flush_posterior at the top of the step).  Padding dimensions (≥ d_out) are decoupled constants.
                    const double* vold = p.cov + (tprev * p.n_chains + chain) * ((size_t)p.d_out * p.d_out);
                    bool bad = false;
#pragma unroll 1
                    for (int k = 0; k < 4 * NT; ++k) {   // rolled, no register arrays: once or twice per segment, and the steps have no register to give
                        const int row = acc_row<NT>(w, lane, k & 3), col = acc_col<NT>(lane, k >> 2);
                        if (row < p.d_out && col < p.d_out) {
                            const double sc = sqrt(fabs(MV[row * LD + row] * MV[col * LD + col]));
                            bad = bad || !(fabs(MV[row * LD + col] - vold[(size_t)row * p.d_out + col]) <= RXHIP_FZ_TOL * sc);
                        }
                    }
                    const bool wave_bad = __any(bad) != 0;
                    if (lane == 0) bz[8 + ws] = wave_bad ? 1.0 : 0.0;
                    lds_barrier();
                    double nbad = 0.0;
#pragma unroll
                    for (int q = 0; q < NT; ++q) nbad += bz[8 + q];
                    bz_verify = false;
                    bz_ok = same && nbad == 0.0;
                    if (!bz_ok) bz_same = 0;
                } else if (bz_same >= RXHIP_FZ_CONFIRM - 1)
                    bz_verify = true;
            }
        }
        pending = true;
        tprev = t;
    }
    if (pending) flush_posterior();
    if (FE && tid == 0) {  // the residual quadratic forms are kd_fe_resid's
        double f = 0.0;
        if (seg == p.S - 1) f += lpe.value();        // log|Λ_f(T)|
        if (seg == 0) {
            if (p.step_model) f += p.fe_const[chain];   // per-step constants: the sum over the chain's steps (km_feconst)
            else {
                f += 2.0 * cst[c.oFEC];
                if (p.mseg) f -= ((double)p.T - p.nobs[chain]) * cst[c.oC0];   // dy log 2π + log|Q| of the observed time indices only
            }
        }
        dense_fe_write(p, seg == 0 ? 0 : p.S + seg, chain, f, 0.0, 0.0);
    }
    if (!ok && tid == 0) atomicOr(p.status, ST_NOT_POSDEF);
}

// Residual quadratic forms of the Bethe free energy at the smoothed means x̂ (information-form smoothing runs):
//   (x̂_1 − m1)'V1⁻¹(x̂_1 − m1) + Σ_{t<T} r_x(t)'P⁻¹r_x(t) + Σ_t r_y(t)'Q⁻¹r_y(t),   r_x = x̂_{t+1} − A x̂_t,  r_y = y_t − B x̂_t
// — the average energies of the MvNormalMeanCovariance nodes minus the part the entropies cancel (a7/a8).  Every term
// depends on the smoothed means of at most two neighbouring steps, so it is NOT part of the sequential backward sweep:
// it runs over all steps in parallel after it, reading the means the sweep stored (inside the sweep the four matvecs and
// their exchange points cost 0.3 ms of 0.67 ms at d = 64 and spilled registers).  One workgroup per FR_STEPS consecutive
// steps of one chain; the four constant maps are staged in LDS once per workgroup (≤ 128 KB), thread (i, wave g) forms row
// i of the matvecs for four steps at a time (one matrix element read feeds four FMAs; the means are LDS broadcasts).
// Fixed summation order -> deterministic.  Partial written (negated, as every fe_part slot) to slot `slot0 + blockIdx.x`.
// Small tiles (d, dy ≤ 16 / ≤ 32) would leave three quarters / half of every wavefront idle with one row per lane, so the 64
// lanes of a wavefront split into G = 4 / 2 lane groups that work on different steps: a pass covers 16·G steps.
__host__ __device__ inline int fe_resid_groups(int D, int dy) { const int m = D > dy ? D : dy; return m <= 16 ? 4 : m <= 32 ? 2 : 1; }
// steps per workgroup: 3 passes of the VALU form; wide tiles (d or dy > 32) take TWO 16-step tiles per workgroup — a single long chain
// (T = 10⁴: 313 workgroups) then fills the chip in one round of two workgroups per CU
__host__ __device__ inline int fe_resid_steps(int D, int dy) { const int g = fe_resid_groups(D, dy); return g == 1 ? 32 : 48 * g; }
// tiles of 16 steps per round of kd_fe_resid_mfma: 8 (tile, item) units over its four waves, at most four observation tiles staged
__host__ __device__ inline int fe_resid_tiles(int D, int dy) {
    const int nty = (dy + 15) / 16, items = D / 16 + nty;
    if (D >= 48) return 1;            // (the kernel treats the tile index as a constant there: registers)
    int t = 8 / items;
    if (t > 4 / nty) t = 4 / nty;
    if (t > 2 && D > 16) t = 2;       // steps per workgroup (fe_resid_steps) stay a multiple of the round: 192 = 3·64, 96 = 3·32, 32 = 1·32
    return t < 1 ? 1 : t;
}
inline size_t fe_resid_lds_bytes(int D, int dy) {
    const size_t pass = 16 * (size_t)fe_resid_groups(D, dy);
    return sizeof(double) * ((size_t)2 * D * D + (size_t)D * dy + (size_t)dy * dy + (2 * pass + 1) * D + pass * dy + 16);
}
static __global__ void __launch_bounds__(256) kd_fe_resid(DenseParams p, int slot0) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int D = p.d, dy = p.dy, tid = threadIdx.x, lane = tid & 63, g = tid >> 6;
    const int G = fe_resid_groups(D, dy), DL = 64 / G, q = lane / DL, i = lane - q * DL;  // lane group q, row i
    const int PASS = 16 * G, STEPS = fe_resid_steps(D, dy);
    const bool pk = p.pack == 2;
    const bool x1 = pk && i >= p.d_sub, y1 = pk && i >= p.dy_sub;  // row i of the state / observation terms belongs to the pair's second chain
    double acc1 = 0.0;
    const long long chain = blockIdx.y + p.chain0, t00 = (long long)blockIdx.x * STEPS;
    const DenseModel M = dense_model(p, chain);  // this chain's model tables
    const DenseCst c = DenseCst::make(D, dy);
    double* AT = smem;                     // [D][D]   A'
    double* PI = AT + (size_t)D * D;       // [D][D]   P⁻¹
    double* BT = PI + (size_t)D * D;       // [D][dy]  B'
    double* QI = BT + (size_t)D * dy;      // [dy][dy] Q⁻¹
    double* mb = QI + (((size_t)dy * dy + 1) & ~(size_t)1);  // [PASS + 1][D], 16-byte aligned  x̂_t of the pass (+ the step after it)
    double* rx = mb + (size_t)(PASS + 1) * D;  // [PASS][D]
    double* ry = rx + (size_t)PASS * D;        // [PASS][dy]
    double* red = ry + (((size_t)PASS * dy + 1) & ~(size_t)1);  // [8]
    auto stage = [&](double* dst, const double* src, int n) {  // eight loads in flight per thread
        int k = tid;
        for (; k + 7 * 256 < n; k += 8 * 256) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[k + u * 256];
#pragma unroll
            for (int u = 0; u < 8; ++u) dst[k + u * 256] = v[u];
        }
        for (; k < n; k += 256) dst[k] = src[k];
    };
    stage(AT, M.cst + c.oAT, D * D);
    stage(PI, M.cst + c.oPI, D * D);
    stage(BT, M.cst + c.oBT, D * dy);
    stage(QI, M.cst + c.oQI, dy * dy);
    double acc = 0.0;
    for (int ps = 0; ps * PASS < STEPS; ++ps) {
        const long long t0 = t00 + (long long)ps * PASS;
        if (t0 >= p.T) break;  // uniform over the workgroup
        for (int k = tid; k < (PASS + 1) * D; k += 256) {
            const int s = k / D, j = k - s * D;
            const long long t = t0 + s;
            mb[k] = t < p.T ? dense_load_mean(p, t, chain, j) : 0.0;
        }
        for (int k = tid; k < PASS * dy; k += 256) {
            const int s = k / dy, j = k - s * dy;
            const long long t = t0 + s;
            ry[k] = t < p.T ? p.y[(t * p.n_chains + chain) * dy + j] : 0.0;
        }
        __syncthreads();
        if (t0 == 0 && g == 0 && q == 0) {  // prior of the first state (constant map read from L2 once per chain)
            if (i < D) {
                const double* V1I = M.cst + c.oV1I;
                const double* m1 = M.cst + c.oM1;
                double u = 0.0;
                for (int k = 0; k < D; k += 16) {  // D is a multiple of 16; sixteen loads in flight
                    double v[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = V1I[(size_t)(k + r) * D + i];
#pragma unroll
                    for (int r = 0; r < 16; ++r) u += v[r] * (mb[k + r] - m1[k + r]);
                }
                acc += (mb[i] - m1[i]) * u;
                if (x1) acc1 += (mb[i] - m1[i]) * u;
            }
        }
        const int s0 = (g * G + q) * 4;  // this thread's four steps of the pass
        {
            double ax[4] = {0.0, 0.0, 0.0, 0.0}, bx[4] = {0.0, 0.0, 0.0, 0.0};
            const bool ix = i < D, iy = i < dy;
#pragma unroll 4
            for (int k = 0; k < D; k += 2) {  // D is even (a multiple of 16)
                const double a0 = ix ? AT[(size_t)k * D + i] : 0.0, a1 = ix ? AT[(size_t)(k + 1) * D + i] : 0.0;
                const double b0 = iy ? BT[(size_t)k * dy + i] : 0.0, b1 = iy ? BT[(size_t)(k + 1) * dy + i] : 0.0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double2 m2 = *reinterpret_cast<const double2*>(mb + (size_t)(s0 + u) * D + k);
                    ax[u] += a0 * m2.x + a1 * m2.y;
                    bx[u] += b0 * m2.x + b1 * m2.y;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long t = t0 + s0 + u;
                if (ix) rx[(size_t)(s0 + u) * D + i] = (t + 1 < p.T) ? mb[(size_t)(s0 + u + 1) * D + i] - ax[u] : 0.0;
                if (iy) ry[(size_t)(s0 + u) * dy + i] = (t < p.T) ? ry[(size_t)(s0 + u) * dy + i] - bx[u] : 0.0;
            }
        }
        __syncthreads();
        {
            double ux[4] = {0.0, 0.0, 0.0, 0.0}, uy[4] = {0.0, 0.0, 0.0, 0.0};
            if (i < D) {
#pragma unroll 4
                for (int k = 0; k < D; k += 2) {
                    const double a0 = PI[(size_t)k * D + i], a1 = PI[(size_t)(k + 1) * D + i];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const double2 r2 = *reinterpret_cast<const double2*>(rx + (size_t)(s0 + u) * D + k);
                        ux[u] += a0 * r2.x + a1 * r2.y;
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc += rx[(size_t)(s0 + u) * D + i] * ux[u];
                    if (x1) acc1 += rx[(size_t)(s0 + u) * D + i] * ux[u];
                }
            }
            if (i < dy) {
#pragma unroll 4
                for (int k = 0; k < dy; ++k) {
                    const double a0 = QI[(size_t)k * dy + i];
#pragma unroll
                    for (int u = 0; u < 4; ++u) uy[u] += a0 * ry[(size_t)(s0 + u) * dy + k];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc += ry[(size_t)(s0 + u) * dy + i] * uy[u];
                    if (y1) acc1 += ry[(size_t)(s0 + u) * dy + i] * uy[u];
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        acc += __shfl_down(acc, off);
        acc1 += __shfl_down(acc1, off);
    }
    if (lane == 0) { red[g] = acc; red[4 + g] = acc1; }
    __syncthreads();
    if (tid == 0) {
        const double tot = ((red[0] + red[1]) + red[2]) + red[3], sec = ((red[4] + red[5]) + red[6]) + red[7];
        dense_fe_write(p, slot0 + blockIdx.x, chain, 0.0, tot - sec, sec);
    }
}

// The same residual forms on the matrix cores (round 3).  With the whitening maps of DenseCst (oLPX, oLQX) every term is the
// squared norm of ONE map applied to stacked means,  ρ_x(t) = [L_P⁻¹ | −L_P⁻¹A]·[x̂_{t+1}; x̂_t],  ρ_y(t) = [L_Q⁻¹ | −L_Q⁻¹B]·[y_t; x̂_t],
// i.e. two GEMMs with the time steps as columns: the map rows are the A operand, a 16-step tile of means / observations the
// B operand, and the accumulator (rows = components, columns = steps) is squared and summed in place.  No intermediate
// vector, no LDS exchange between the two stages of r'P⁻¹r.  Same grid and slots as kd_fe_resid.
template <int NT>
inline size_t fe_resid_mfma_lds_bytes(int dy) {
    constexpr int D = 16 * NT;
    const size_t rs = (size_t)16 * fe_resid_tiles(D, dy);
    return sizeof(double) * ((rs + 1) * (D + 2) + rs * (((dy + 7) & ~7) + 2) + 16 + D + 2);
}
// The kernel has 3 % of the sweep's arithmetic and was 5 % of its time (d = 64, T = 10⁴: 44 µs) until time stamps inside it
// (wall_clock64 at six points of four workgroups) showed where: none of it was arithmetic.
//   * maps: straight from L2 into registers in the A-operand layout, once per wave (at most two row tiles: 64 doubles per lane at
//     d = dy = 64), 16 bytes per lane and load with a permuted contraction index (see load_item);
//   * means / observations of a tile: coalesced loads, ALL issued before the first LDS store, the next tile's issued before this
//     tile's products (a load -> wait -> store loop was nine round trips per tile);
//   * B operands: unconditional 16-byte LDS reads in batches (a predicated read per product serialised read -> wait -> product);
//   * the model pointers: values, not a vector load through the kernel-argument segment (dense_model);
//   * the prior term: one round trip spread over the four waves at the end, not four round trips of one wave in the first tile.
// 44 -> 21 µs at d = 64, T = 10⁴ (what is left: 4 µs until the maps are in registers, 2 × 3 µs of products with two workgroups
// per CU on the same matrix pipes, 2 µs of staging per tile).  17 KB of LDS per workgroup.
template <int NT>
__global__ void __launch_bounds__(256, 2) kd_fe_resid_mfma(DenseParams p, int slot0) {
    constexpr int D = 16 * NT, KX = 2 * D, LDB = D + 2;
    typedef double v4d __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int dy = p.dy, dy4 = (dy + 3) & ~3, dy8 = (dy + 7) & ~7, nty = (dy + 15) / 16, KY = dy4 + D, LDY = dy8 + 2;
    const int tid = threadIdx.x, lane = tid & 63, g = tid >> 6, j = lane & 15, kq = lane >> 4;
    // row-tile items: [x-tiles 0 … NT−1 | y-tiles 0 … nty−1].  A wave takes at most two (tile, item) units per round, so a round
    // covers TPB = 8 / items tiles of 16 steps (narrow models: d = 16 has two items — four tiles at once instead of two idle waves)
    const int nitems = NT + nty, TPB = NT >= 3 ? 1 : fe_resid_tiles(D, dy), nunits = TPB * nitems, RS = 16 * TPB;   // (d ≥ 48: one tile — a constant)
    double* xs = smem;                 // [RS + 1][LDB]  x̂ of the round's steps and of the step after them
    double* ys = xs + (RS + 1) * LDB;  // [RS][LDY]
    double* red = ys + RS * LDY;       // [8]
    double* dm = red + 8;              // [D]  x̂_1 − m1 (first workgroup of a chain)
    const int DUMP = (int)(dm + D - xs), DUMPY = (int)(dm + D - ys);   // one slot behind everything: where out-of-tile stores go
    const long long chain = blockIdx.y + p.chain0;
    DenseModel M = dense_model(p, chain);
    if (p.step_model) M.cst = p.cst + (size_t)p.model_sel * (size_t)p.cst_stride;   // per-step constants: ONE model per launch, the columns of the others masked
    const DenseCst c = DenseCst::make(D, dy);
    const int STEPS = fe_resid_steps(D, dy), NTT = STEPS / RS;   // rounds per workgroup
    const long long t00 = (long long)blockIdx.x * STEPS;
    double s0 = 0.0, s1 = 0.0;   // first / second chain of a packed pair (unpacked: everything in s0)
    const bool pk = p.pack == 2;
    // units of this wave: g and g + 4 of the round's TPB·items;  unit q = (tile q / items, item q mod items)
    const int tl0 = NT >= 3 ? 0 : g / nitems, tl1 = NT >= 3 ? 0 : (g + 4) / nitems;
    const int it0 = g < nunits ? g - tl0 * nitems : nitems, it1 = g + 4 < nunits ? g + 4 - tl1 * nitems : nitems;
    // A operands of an item: an x item holds [L_P⁻¹ | −L_P⁻¹A] (2·D/4 k-steps), a y item holds L_Q⁻¹ padded to 16 k-steps (zeros
    // behind dy) followed by −L_Q⁻¹B (D/4 k-steps) — every register index is a compile-time constant.  The contraction index is
    // PERMUTED so that memory is read in 16-byte pieces: k-steps 2m and 2m + 1 of lane quarter kq are elements 8m + 2kq and
    // 8m + 2kq + 1 of the row — one 16-byte load per lane and pair of steps, 64 contiguous bytes per row and instruction (half
    // the cache-line visits of an 8-byte gather: the map loads of the 2 × 4 waves of a CU were 7 of the kernel's 17 µs), and the
    // B operand of the pair is one 16-byte LDS read of the same two elements.
    constexpr int KA = (64 + D) / 4 > KX / 4 ? (64 + D) / 4 : KX / 4;   // even
    typedef double v2d __attribute__((ext_vector_type(2)));
    double a0[KA], a1[KA];
    auto load_item = [&](double (&a)[KA], int it) {
        const bool isx = it < NT, live = it < nitems;
        const int rt = isx ? it : it - NT, K = isx ? KX : KY;
        const double* row = (isx ? M.cst + c.oLPX : M.cst + c.oLQX) + (live ? (size_t)(16 * rt + j) * K + 2 * kq : 0);
#pragma unroll
        for (int m = 0; m < KA / 2; ++m) {
            // x item: pairs m < D/8 the first half, the next D/8 the second;  y item: pairs m < 8 the y part (elements below dy4), then the x̂ part
            const int e = isx ? (m < D / 8 ? 8 * m : D + 8 * (m - D / 8)) : (m < 8 ? 8 * m : dy4 + 8 * (m - 8));
            const bool ex = live && (isx ? m < D / 4 : (m < 8 ? 8 * m + 2 * kq < dy4 : 8 * (m - 8) < D));
            v2d v = {0.0, 0.0};
            if (ex) v = *reinterpret_cast<const v2d*>(row + e);   // (a skipped load, no wait in between)
            a[2 * m] = v.x;
            a[2 * m + 1] = v.y;
        }
    };
    load_item(a0, it0);
    load_item(a1, it1);
    // The B operands are read unconditionally (a batch of LDS reads ahead of a batch of products; a predicated read per product
    // serialises read -> wait -> product: 14 µs per tile at d = 64 instead of 2); what must not count — the transition out of the
    // last step, the observation term of a missing step — is a COLUMN of the product and this lane's own column: masked at the end.
    auto run_item = [&](const double (&a)[KA], int it, int tile, bool vx, bool ob) {
        if (it >= nitems) return;   // uniform over the wave
        const bool isx = it < NT;
        const int rt = isx ? it : it - NT;
        v4d acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};   // two independent chains on the matrix pipe
        const double* xr = xs + (16 * tile + j) * LDB + 2 * kq;
        if (isx) {   // ρ_x = L_P⁻¹ x̂_{t+1} − (L_P⁻¹A) x̂_t
#pragma unroll
            for (int m = 0; m < D / 8; ++m) {
                const v2d b1 = *reinterpret_cast<const v2d*>(xr + LDB + 8 * m), b0 = *reinterpret_cast<const v2d*>(xr + 8 * m);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2 * m], b1.x, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[D / 4 + 2 * m], b0.x, acc2, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2 * m + 1], b1.y, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[D / 4 + 2 * m + 1], b0.y, acc2, 0, 0, 0);
                if (m % 2 == 1) __builtin_amdgcn_sched_barrier(0);   // four reads ahead of eight products, not all of them (registers)
            }
        } else {     // ρ_y = L_Q⁻¹ y_t − (L_Q⁻¹B) x̂_t
            const double* yr = ys + (16 * tile + j) * LDY + 2 * kq;
#pragma unroll
            for (int m = 0; m < 8; ++m)
                if (8 * m < dy4) {   // uniform; the staged row is zero-filled up to dy8, the operand is zero behind dy4
                    const v2d b = *reinterpret_cast<const v2d*>(yr + 8 * m);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2 * m], b.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2 * m + 1], b.y, acc, 0, 0, 0);
                }
#pragma unroll
            for (int m = 0; m < D / 8; ++m) {
                const v2d b = *reinterpret_cast<const v2d*>(xr + 8 * m);
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[16 + 2 * m], b.x, acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[16 + 2 * m + 1], b.y, acc2, 0, 0, 0);
                if (m % 4 == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        acc += acc2;
        const int sub = isx ? p.d_sub : p.dy_sub;
        const bool count = isx ? vx : ob;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double e = count ? acc[r] * acc[r] : 0.0;
            if (pk && 16 * rt + kq + 4 * r >= sub) s1 += e;
            else s0 += e;
        }
    };
    // A tile's loads are all issued first and stored to LDS later: a load -> wait -> store loop pays one memory round trip per pass
    // (nine per tile: 40 µs of a 45 µs kernel at d = 64, whatever the rest of the kernel did), and the NEXT tile's loads are issued
    // before this tile's products so that they arrive under them.  The existence flags go through an empty asm before they
    // select (otherwise the compiler moves each load under its flag, and the wait with it); elements past the tile are stored
    // to a dump slot so that the stores are unconditional too.
    constexpr int XV = 5;   // (RS + 1)·D / 256 passes at most (TPB·NT ≤ 4);  y: TPB·dy8 / 16 ≤ 4
    double xv[XV], yv[4], obv0 = 1.0, obv1 = 1.0, m1v = 0.0;
    int xe[XV], ye[4];
    auto issue = [&](long long tb) {
#pragma unroll
        for (int u = 0; u < XV; ++u) {   // coalesced: consecutive threads, consecutive components of one step
            const int k = tid + 256 * u, sidx = k / D, comp = k - sidx * D;
            const long long t = tb + sidx;
            bool ex;
            const long long off = dense_mean_offset(p, t < p.T ? t : p.T - 1, chain, comp, ex);
            xv[u] = p.mean[off];
            xe[u] = ex && k < (RS + 1) * D && t < p.T;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = tid + 256 * u, sidx = k / dy8, comp = k - sidx * dy8;
            const long long t = tb + sidx;
            const bool ex = k < RS * dy8 && t < p.T && comp < dy;
            yv[u] = p.y[ex ? (t * p.n_chains + chain) * dy + comp : 0];
            ye[u] = ex;
        }
        // lane (j, ·): B-operand column j of tile tl = time step tb + 16·tl + j
        obv0 = p.mseg ? p.obs[tb + 16 * tl0 + j < p.T ? chain * p.T + tb + 16 * tl0 + j : 0] : 1.0;
        obv1 = p.mseg ? p.obs[tb + 16 * tl1 + j < p.T ? chain * p.T + tb + 16 * tl1 + j : 0] : 1.0;
        if (tb == 0) m1v = M.cst[c.oM1 + (tid < D ? tid : 0)];
    };
    auto deposit = [&](long long tb) {
#pragma unroll
        for (int u = 0; u < XV; ++u) {
            const int k = tid + 256 * u, sidx = k / D, comp = k - sidx * D;
            asm volatile("" : "+v"(xe[u]));
            xv[u] = xe[u] ? xv[u] : 0.0;
            xs[k < (RS + 1) * D ? sidx * LDB + comp : DUMP] = xv[u];
        }
        if (tb == 0 && tid < D) dm[tid] = xv[0] - m1v;   // k = tid < D: step 0, component tid
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = tid + 256 * u, sidx = k / dy8, comp = k - sidx * dy8;
            asm volatile("" : "+v"(ye[u]));
            ys[k < RS * dy8 ? sidx * LDY + comp : DUMPY] = ye[u] ? yv[u] : 0.0;
        }
    };
    if (t00 < p.T) issue(t00);
    for (int tt = 0; tt < NTT; ++tt) {
        const long long tb = t00 + (long long)RS * tt;
        if (tb >= p.T) break;   // uniform over the workgroup
        __syncthreads();        // the previous round's readers are done
        deposit(tb);
        const long long tj0 = tb + 16 * tl0 + j, tj1 = tb + 16 * tl1 + j;
        bool vx0 = tj0 + 1 < p.T, vx1 = tj1 + 1 < p.T;
        bool ob0 = tj0 < p.T && obv0 != 0.0, ob1 = tj1 < p.T && obv1 != 0.0;   // `missing`: no observation node energy at this time index
        if (p.step_model) {   // the transition into t + 1 belongs to model step_model[t + 1], the observation at t to step_model[t]
            vx0 = vx0 && p.step_model[tj0 + 1] == p.model_sel;
            vx1 = vx1 && p.step_model[tj1 + 1] == p.model_sel;
            ob0 = ob0 && p.step_model[tj0] == p.model_sel;
            ob1 = ob1 && p.step_model[tj1] == p.model_sel;
        }
        __syncthreads();
        if (tt + 1 < NTT && tb + RS < p.T) issue(tb + RS);
        run_item(a0, it0, tl0, vx0, ob0);
        run_item(a1, it1, tl1, vx1, ob1);
    }
    if (t00 == 0 && lane < D && (!p.step_model || p.step_model[0] == p.model_sel)) {   // prior of the first state: (x̂_1 − m1)'V1⁻¹(x̂_1 − m1) = Σ_waves dm_i Σ_{k in the wave's quarter} V1⁻¹[k][i] dm_k
        const double* V1I = M.cst + c.oV1I + lane;   // row i = lane; every load independent (a uniform address per term would be a
        double u = 0.0;                              // chain of scalar loads), one round trip for the whole term
#pragma unroll
        for (int k = g * (D / 4); k < (g + 1) * (D / 4); ++k) u += V1I[(size_t)k * D] * dm[k];
        const double e = dm[lane] * u;
        if (pk && lane >= p.d_sub) s1 += e;
        else s0 += e;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s0 += __shfl_down(s0, off);
        s1 += __shfl_down(s1, off);
    }
    __syncthreads();
    if (lane == 0) { red[g] = s0; red[4 + g] = s1; }
    __syncthreads();
    if (tid == 0) {
        const double first = ((red[0] + red[1]) + red[2]) + red[3], sec = ((red[4] + red[5]) + red[6]) + red[7];
        dense_fe_write(p, slot0 + blockIdx.x, chain, 0.0, first, sec);
    }
}

// Node-local joints q(x[t], x[t+1] | y) need Cov(x[t], x[t+1] | y) = G_t V_s(t+1) beyond the posteriors (rxhip_get_node_marginals).  After a
// smoothing run of the information-form sweep the gain is still in the records (G_t′ = K C_t, accumulator order) and V_s(t+1) in the
// posterior array: one workgroup per (t, chain), one product — instead of re-running the whole chain on the sequential kernels.
// One chain per tile only (pack = 1).  cross: [T − 1][chain][d_out][d_out].
template <int NT>
__global__ void __launch_bounds__(64 * NT) kd_cross_from_records(DenseParams p, double* cross) {
    constexpr int D = 16 * NT;
    using C = DenseCfg<NT>;
    constexpr int LD = C::LD;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Gt = smem;            // G_t′: [k][i]
    double* Vs = Gt + C::MAT;     // V_s(t + 1), zero-padded
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, dout = p.d_out;
    const long long row = blockIdx.x, t = row / p.n_chains, chain = row - t * p.n_chains;
    Acc<NT> g;
    long long tm = t;   // the record that carries G_t′: its own, or the one record of a frozen stretch that holds the matrices (kd_forward_info: FZ_SLOT)
    if constexpr (RXHIP_FWD_FROZEN && DenseLds<NT>::ALIAS) {
        if (p.mseg == 0 && !p.step_model) {
            const long long tb = (t / p.L) * p.L;
            const long long tfz = (long long)p.filt[(chain * p.T + (tb + 1)) * C::REC + C::HDR + (NT * 4) * 64];
            if (t > tfz) tm = tfz;
        }
    }
    acc_load_full<NT>(g, p.filt + (chain * p.T + tm) * C::REC + C::HDR + D * D, w, lane);
    acc_store<NT>(g, Gt, LD, w, lane);
    const double* vs = p.cov + ((t + 1) * p.n_chains + chain) * (size_t)dout * dout;
    for (int k = tid; k < D * D; k += 64 * NT) {
        const int i = k / D, j = k - i * D;
        Vs[i * LD + j] = (i < dout && j < dout) ? vs[(size_t)i * dout + j] : 0.0;
    }
    __syncthreads();
    Acc<NT> a;
    acc_zero<NT>(a);
    mm_acc<NT, true, false>(a, Gt, LD, Vs, LD, w, lane);   // (G_t′)′ V_s(t + 1)
    acc_store_out<NT>(a, cross + row * (size_t)dout * dout, dout, w, lane);
}

// Per-step constants with MANY models (a fully time-varying model has one per step): a pass of kd_fe_resid_mfma per model would be a
// launch per model.  Here every wavefront takes one time step at a time — lane i forms row i of the two whitened residuals with the maps of
// THAT step's models (rows read along k: uncoalesced, but the work is tiny: 4·d² multiply-adds per step) — 64 steps per workgroup, one
// partial slot each.  Same terms, same masks as the MFMA form.
constexpr int FE_STEPS_BLOCK = 64;
static __global__ void __launch_bounds__(256) kd_fe_resid_steps(DenseParams p, int slot0) {
    __shared__ double xb[4][3][64];
    __shared__ double red[4];
    const int D = p.d, dy = p.dy, dy4 = (dy + 3) & ~3, KY = dy4 + D;
    const int tid = threadIdx.x, lane = tid & 63, g = tid >> 6;
    const long long chain = blockIdx.y + p.chain0, t00 = (long long)blockIdx.x * FE_STEPS_BLOCK;
    const DenseCst c = DenseCst::make(D, dy);
    double* x0 = xb[g][0];
    double* x1 = xb[g][1];
    double* yv = xb[g][2];
    double acc = 0.0;
    for (int q = g; q < FE_STEPS_BLOCK; q += 4) {
        const long long t = t00 + q;
        if (t >= p.T) break;   // uniform over the wavefront
        const bool hasx = t + 1 < p.T, ob = p.obs[chain * p.T + t] != 0.0;
        x0[lane] = lane < D ? dense_load_mean(p, t, chain, lane) : 0.0;
        x1[lane] = (hasx && lane < D) ? dense_load_mean(p, t + 1, chain, lane) : 0.0;
        yv[lane] = (ob && lane < dy) ? p.y[(t * p.n_chains + chain) * dy + lane] : 0.0;
        wave_lds_fence();
        if (hasx && lane < D) {   // ρ_x = L_P⁻¹ x̂_{t+1} − (L_P⁻¹A) x̂_t with the transition INTO t + 1
            const double* mp = p.cst + (size_t)p.step_model[t + 1] * (size_t)p.cst_stride + c.oLPX + (size_t)lane * 2 * D;
            double r0 = 0.0, r1 = 0.0;
#pragma unroll 8
            for (int k = 0; k < D; k += 2) {
                r0 += mp[k] * x1[k] + mp[D + k] * x0[k];
                r1 += mp[k + 1] * x1[k + 1] + mp[D + k + 1] * x0[k + 1];
            }
            acc += (r0 + r1) * (r0 + r1);
        }
        if (ob && lane < dy) {   // ρ_y = L_Q⁻¹ y_t − (L_Q⁻¹B) x̂_t
            const double* mq = p.cst + (size_t)p.step_model[t] * (size_t)p.cst_stride + c.oLQX + (size_t)lane * KY;
            double r = 0.0;
            for (int k = 0; k < dy; ++k) r += mq[k] * yv[k];
            for (int k = 0; k < D; ++k) r += mq[dy4 + k] * x0[k];
            acc += r * r;
        }
        if (t == 0 && lane < D) {   // prior of the first state
            const double* cm = p.cst + (size_t)p.step_model[0] * (size_t)p.cst_stride;
            double u = 0.0;
            for (int k = 0; k < D; ++k) u += cm[c.oV1I + (size_t)k * D + lane] * (x0[k] - cm[c.oM1 + k]);
            acc += (x0[lane] - cm[c.oM1 + lane]) * u;
        }
        wave_lds_fence();
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if (lane == 0) red[g] = acc;
    __syncthreads();
    if (tid == 0) dense_fe_write(p, slot0 + blockIdx.x, chain, 0.0, ((red[0] + red[1]) + red[2]) + red[3], 0.0);
}

}  // namespace rxhip
