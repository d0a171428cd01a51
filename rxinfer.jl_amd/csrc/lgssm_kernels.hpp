// lgssm_kernels.hpp — hand-written HIP kernels (gfx950 / CDNA4, fp64) for the Gaussian
// sum-product hot path of RxInfer on linear Gaussian state-space factor graphs.
//
// What these kernels replace (reference = RxInfer.jl checkout; the rule bodies themselves live
// in the un-vendored ReactiveMP.jl / ExponentialFamily.jl):
//   a3  @rule MvNormalMeanCovariance(:out|:μ)      selected at src/model/graphppl.jl:372-376
//   a4  @rule typeof(*)(:out|:in) with constant A   created for `A * x[t-1]` (benchmarks notebook cell 4)
//   a5  message product at a variable               src/model/plugins/reactivemp_inference.jl:365-374,432-447
//   a6  marginal computation + mean_cov             reactivemp_inference.jl:440-447,626-629
//   a7  Bethe free energy                           src/model/plugins/reactivemp_free_energy.jl:51-126
//
// Design: one lane owns one (chain, time-segment).  All d×d algebra of a
// factor node lives in that lane's registers as fully unrolled fp64 FMAs (packed-symmetric
// where the matrix is symmetric); adjacent lanes are adjacent chains so every global access of a
// wave is a contiguous run.  Time is cut into S segments per chain so that 1024 chains × S
// segments fill 256 CUs; segment boundaries are made exact (not approximate) with Kalman
// "elements" (Särkkä & García-Fernández 2021, temporal parallelisation of Bayesian smoothers):
//   phase 1  k_seg_aggregate : per (chain, segment) the data-dependent part (b, η) of the
//                              segment's element, using per-model gain tables (reads y)
//   phase 2  k_boundary_scan : prefix scan -> filtered belief at every segment start,
//                              suffix scan -> backward message at every segment end
//   phase 3  k_forward       : forward messages inside each segment (`*`(:out), MvN(:out), product
//                              with the observation message) + evidence terms of the Bethe free
//                              energy; stores the filtered message packed-symmetric
//   phase 4  k_backward      : backward messages + marginals inside each segment (MvN(:μ),
//                              `*`(:in), 3-way product, mean_cov) in Rauch–Tung–Striebel form
//   k_fe_reduce              : deterministic reduction of the free-energy partials
// The result equals the reference's sequential schedule up to fp64 rounding (tests: 1e-6
// relative on posteriors, 1e-8 relative on free energy — the north_star tolerances).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rxhip {

// ------------------------------------------------------------------------------------------
// device status bits (OR-ed into Params::status)
constexpr int ST_NOT_POSDEF = 1;
constexpr int ST_NONFINITE = 2;
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also fences global memory, i.e. it drains every global
// load / store in flight (s_waitcnt vmcnt(0)) — fatal for kernels that keep prefetches or large posterior / record stores in
// flight across their LDS exchange points.  Use where threads communicate through LDS alone.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// finiteness from the bit pattern (x − x == 0 is not safe under -ffp-contract=fast when x is a product)
__device__ __forceinline__ bool is_finite(double x) { return (__double2hiint(x) & 0x7ff00000) != 0x7ff00000; }

// the 16-byte posterior stores of the table-driven smoother (written once, never read again by the sweep).  RXHIP_NT_STORES=1 issues them
// with the non-temporal hint (`global_store_dwordx4 … nt`): an A/B switch — the plain store is what ships, see DESIGN §6f for the measurement
#ifndef RXHIP_NT_STORES
#define RXHIP_NT_STORES 0
#endif
typedef double rx_d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void stream_store(double2* p, double a, double b) {
#if RXHIP_NT_STORES
    rx_d2v v = {a, b};
    __builtin_nontemporal_store(v, reinterpret_cast<rx_d2v*>(p));
#else
    *p = make_double2(a, b);
#endif
}

template <int D>
struct Dim {
    static constexpr int NS = D * (D + 1) / 2;  // packed symmetric size
    static constexpr int NP = D + NS;           // Gaussian record: vector + packed matrix
    static constexpr int NP2 = (NP + 1) / 2;    // record size in 16-byte pairs
};

__host__ __device__ constexpr int sidx(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

template <int D>
struct Sym {
    double v[D * (D + 1) / 2];
    __device__ __forceinline__ double& operator()(int i, int j) { return v[sidx(i, j)]; }
    __device__ __forceinline__ const double& operator()(int i, int j) const { return v[sidx(i, j)]; }
};

// Per-model constant block (doubles), built on the host at create time.
template <int D, int DY>
struct CstLayout {
    static constexpr int NS = D * (D + 1) / 2;
    static constexpr int NSY = DY * (DY + 1) / 2;
    static constexpr int A = 0;               // [D][D]      transition matrix
    static constexpr int P = A + D * D;       // [NS]        state-noise covariance (packed)
    static constexpr int LOBS = P + NS;       // [NS]        B' Q^-1 B  (precision of the `*`_B(:in) message)
    static constexpr int G = LOBS + NS;       // [D][DY]     B' Q^-1    (its weighted mean is G y)
    static constexpr int QI = G + D * DY;     // [NSY]       Q^-1 (packed)
    static constexpr int C0 = QI + NSY;       // [1]         dy log 2π + logdet Q
    static constexpr int M1 = C0 + 1;         // [D]         mean of the message toward x[1]
    static constexpr int V1 = M1 + D;         // [NS]        its covariance
    static constexpr int HF = V1 + NS;        // [DY][D]     B A
    static constexpr int SIZE = ((HF + DY * D + 7) / 8) * 8;
};
// Per-model, per-offset gain table entry (phase 1): K_i [D][DY], U_i [D][DY]
template <int D, int DY>
struct TabLayout {
    static constexpr int K = 0;
    static constexpr int U = D * DY;
    static constexpr int SIZE = 2 * D * DY;
};
// Shared-model smoothing runs make ONE pass over the observations (k_forward0): per position in a segment the gains of
// TabLayout plus the inverse innovation covariance of the known-start filter (evidence), and per TIME INDEX the map M_t that
// turns the known-start quantities into the true filtered mean up to the term in the segment's start mean (see k_forward0).
template <int D, int DY>
struct F0Layout {
    static constexpr int K = 0;                   // [D][DY]
    static constexpr int U = D * DY;              // [D][DY]
    static constexpr int SI = 2 * D * DY;         // [NSY]  (S⁰_i)⁻¹ packed
    static constexpr int SIZE = ((SI + DY * (DY + 1) / 2 + 1) / 2) * 2;
};
template <int D>
struct TimeTab {
    static constexpr int MT = ((D * D + 1) / 2) * 2;  // a row of mtab / ntab: [D][D], padded to whole 16-byte pieces
};
// per position in a segment, data-independent (input of k_time_tables): Π_i, J_i, C_i
template <int D>
struct PosLayout {
    static constexpr int NS = D * (D + 1) / 2;
    static constexpr int PI = 0;
    static constexpr int J = D * D;
    static constexpr int C = J + NS;
    static constexpr int SIZE = C + NS;
};
// per segment, data-independent: the quadratic form of the segment's evidence in (m_s, η_s)  (k_fe_seg)
template <int D>
struct FeSegLayout {
    static constexpr int NS = D * (D + 1) / 2;
    static constexpr int A1 = 0;          // [NS]   J W V_s⁻¹ = (V_s + J⁻¹)⁻¹
    static constexpr int A2 = NS;         // [D][D] W V_s⁻¹
    static constexpr int W = A2 + D * D;  // [NS]   (V_s⁻¹ + J)⁻¹
    static constexpr int SIZE = W + NS;
};
// Per-model, per-length matrix part of a segment element (phase 2)
template <int D>
struct AggLayout {
    static constexpr int NS = D * (D + 1) / 2;
    static constexpr int PI = 0;          // [D][D]  Π  (product of closed-loop matrices)
    static constexpr int C = PI + D * D;  // [NS]    C  (covariance of x_end | x_start, y_seg)
    static constexpr int J = C + NS;      // [NS]    J  (information about x_start in y_seg)
    static constexpr int CI = J + NS;     // [NS]    C^-1
    static constexpr int X = CI + NS;     // [D][D]  C^-1 Π
    static constexpr int JJ = X + D * D;  // [NS]    J + Π' C^-1 Π
    static constexpr int SIZE = ((JJ + NS + 7) / 8) * 8;
};

// Per-model, per-segment matrices of the boundary scan (shared-model batches): the covariance at every
// segment start, the precision of the backward message at every segment end and the d×d maps that carry the
// data-dependent vectors across a segment do not depend on the observations; they are built on the host.
//   m(b_{s+1}) = b_s + M1_s m(b_s) + M2_s η_s          ξβ(b_s) = η_s + N1_s ξβ(b_{s+1}) − N2_s b_s
template <int D>
struct ScanLayout {
    static constexpr int NS = D * (D + 1) / 2;
    static constexpr int M1 = 0;             // [D][D]
    static constexpr int M2 = M1 + D * D;    // [D][D]
    static constexpr int VB = M2 + D * D;    // [NS]  V(b_s)
    static constexpr int N1 = VB + NS;       // [D][D]
    static constexpr int N2 = N1 + D * D;    // [D][D]
    static constexpr int LB = N2 + D * D;    // [NS]  Λβ(b_{s+1})
    static constexpr int SIZE = ((LB + NS + 7) / 8) * 8;
};

struct Params {
    // problem
    long long T;
    long long n_chains;
    int S;           // segments
    long long L;     // segment length (last one may be shorter)
    int n_models;
    // device buffers
    const double* y;        // [T][chain][DY]
    double* filt;           // [T][chain/64][NP2][64][2]   filtered message (m_f, V_f packed), wave-blocked
    long long nb64;         // ceil(n_chains / 64)
    double* vtab;           // [T][NS]  forward-message covariance V_f(t), ONE copy per model (shared-model batches)
    const double* scan;     // [S][ScanLayout::SIZE]  data-independent part of the boundary scan (shared-model batches)
    double* mean;           // [T][chain][D]
    double* cov;            // [T][chain][D][D]
    const double* cst;      // [n_models][CstLayout::SIZE]
    const double* tab;      // [n_models][L][TabLayout::SIZE]
    const double* agg;      // [n_models][2][AggLayout::SIZE]   (0: length L, 1: length of the last segment)
    const int* chain_model; // [chain] or nullptr
    double* elem;           // [S][2D][chain]       (b, η) of every segment element
    double* fstart;         // [S][NP][chain]       filtered belief at every segment start
    double* beta;           // [S+1][NP][chain]     backward message (ξ, Λ packed) at every boundary
    double* fe_part;        // [S+1][chain]         Σ log p(y_t | y_<t) partials
    double* fe_chain;       // [chain]
    double* fe_total;       // [iterations]
    int iteration;
    int filter;        // 1: filtering run — forward pass only, q(x_t | y_1..t) written as the marginals
    double fe_scale;   // 1 (smoothing: Bethe free energy of the chain) or 1/T (filtering: mean over observations)
    int* status;
    int masked;        // 1: NaN observations are `missing` (per-chain records, one segment: rxhip_lgssm_desc.allow_missing)
    // shared-model smoothing, one pass over the observations (null / 0: the two-pass schedule of a filtering run)
    const double* ftab;     // [L][F0Layout::SIZE]
    const double* mtab;     // [T][TimeTab::MT]   M_t
    const double* ntab;     // [T][TimeTab::MT]   N_t = M_t V_s⁻¹
    const double* fseg;     // [S][FeSegLayout::SIZE]
    double fe_const;        // Σ_segments of the data-independent evidence terms
    double* elemx;          // [S][D² + 2·NS][chain] or null: matrix part (Π, J, C) of every segment element, per chain — the
                            // masked / per-step-constant schedules, where it depends on the chain's data pattern and the
                            // time index (k_seg_elements); null: per-model tables `agg`
    const int* step_model;  // [T] or null: the model of time index t (transition INTO x[t] and observation of y[t]);
                            // one segment, per-chain records (rxhip_lgssm_desc.step_model)
    // engines with an unknown observation-noise precision (noise_kernels.hpp): the backward sweep accumulates the residual second moments
    // Σ_t [(y_t − B m_t)(y_t − B m_t)′ + B V_t B′] of its segment while it holds (m_t, V_t) — part[seg][NS_y][chain]; null otherwise
    const double* noise_B;  // [DY][D]
    double* noise_part;
    int skip_marginals;     // unknown-noise engines, VMP iterations before the last of a run: the sweep leaves its residual moments and free-energy terms,
                            // not the 160 B/U of posteriors that the next iteration overwrites unread (rxhip_get_marginals returns the LAST iteration's)
    int tinv_records;       // per-chain, time-invariant models on long segments: k_forward_tinv / k_backward_tinv (mean-only records behind the fixed point)
    int elem_full;          // test hook: k_seg_elements runs the full recursion to the end of every segment (no frozen tail)
};

// ------------------------------------------------------------------------------------------
// small dense algebra, everything unrolled so that all indices are compile-time constants and
// the operands stay in VGPRs (checked with -Rpass-analysis=kernel-resource-usage: no scratch).

// 1/x for a positive, normal x: v_rcp_f64 seed + two Newton steps (5 instructions instead of the
// ~11 of the IEEE-exact division expansion; result within 1–2 ulp, far inside the 1e-6 / 1e-8
// parity budget — the pivots it is applied to are already rounded results).
__device__ __forceinline__ double rcp_pos(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}

// SPD inverse through LDL' (no square roots).  `det` receives det(a) (product of pivots).
// Restates FastCholesky.cholinv for the small blocks on the path; a non-positive pivot
// reports ST_NOT_POSDEF (the reference throws PosDefException).
template <int D>
__device__ __forceinline__ bool spd_inv(const Sym<D>& a, Sym<D>& out, double& det) {
    double L[D][D], W[D][D], r[D];
    bool ok = true;
    det = 1.0;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        double s = a(j, j);
#pragma unroll
        for (int k = 0; k < j; ++k) s -= W[j][k] * L[j][k];
        ok = ok && (s > 0.0);
        det *= s;
        r[j] = rcp_pos(s);
#pragma unroll
        for (int i = j + 1; i < D; ++i) {
            double t = a(i, j);
#pragma unroll
            for (int k = 0; k < j; ++k) t -= W[i][k] * L[j][k];
            W[i][j] = t;
            L[i][j] = t * r[j];
        }
    }
    // M = L^-1 (unit lower)
    double M[D][D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
#pragma unroll
        for (int i = j + 1; i < D; ++i) {
            double s = -L[i][j];
#pragma unroll
            for (int k = j + 1; k < i; ++k) s -= L[i][k] * M[k][j];
            M[i][j] = s;
        }
    }
    // out = M' D^-1 M
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            // k = i term: M(i,i) = 1
            double s = (i == j) ? r[i] : r[i] * M[i][j];
#pragma unroll
            for (int k = i + 1; k < D; ++k) s += (M[k][i] * r[k]) * M[k][j];
            out(i, j) = s;
        }
    }
    return ok;
}

// y = S x  (S symmetric packed)
// Scale-free linear functionals of a symmetric matrix, for the fixed-point exits of the time-invariant recursions (k_seg_elements<TINV>,
// k_boundary_scan<TS>, k_forward_tinv).  Every entry is first scaled by an EXACT power of two, S_ij = M_ij · 2^−⌊(e_i + e_j)/2⌋ with e_i the binary
// exponent of |M_ii| — the entry on the scale of its own row and column, |S_ij| ≤ 2 for a definite matrix — and the functionals are weighted sums
// of the S_ij.  A test "f unchanged to 2 ulp" then sees a change of ANY entry above ≈ 1e-14 · sqrt(M_ii M_jj), however many decades lie between
// the state's components (the plain sums of rounds 3–4 saw only what moved the largest entries: tests/test_fixed_point_adversarial_gpu.py).
// A diagonal entry that crosses a power of two changes its scaled value by a factor of two: the test reads "moved" for that step, never "same".
// Cost: D exponent extractions and D(D+1)/2 integer adds + ldexp per step.
template <int D>
__device__ __forceinline__ void fixpoint_functionals(const Sym<D>& M, double& f1, double& f2) {
    int e[D];
#pragma unroll
    for (int i = 0; i < D; ++i) e[i] = __builtin_amdgcn_frexp_exp(fabs(M(i, i)));
    f1 = 0.0;
    f2 = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            const double sv = __builtin_amdgcn_ldexp(M.v[sidx(i, j)], -((e[i] + e[j]) >> 1));
            f1 += sv;
            f2 += (1.0 + 0.37 * sidx(i, j)) * sv;
        }
}
template <int D>
__device__ __forceinline__ void symv(const Sym<D>& S, const double (&x)[D], double (&y)[D]) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) s += S(i, k) * x[k];
        y[i] = s;
    }
}

// constant access: uniform (scalar loads) or per-lane pointer — same code
struct CPtr {
    const double* p;
    __device__ __forceinline__ double operator[](int i) const { return p[i]; }
};
struct LdsLanePtr {  // constants parked in LDS, [k][lane of a 64-thread workgroup]
    const double* p;
    __device__ __forceinline__ double operator[](int i) const { return p[i * 64]; }
};
// When every chain uses the same model the constant block travels BY VALUE in the kernel
// argument segment: kernarg reads are scalar loads (s_load_*) that in-loop global stores can
// never alias, and the values feed v_fma_f64 straight from SGPRs.  (Read through the global
// pointer instead, the compiler must assume the stores clobber them and re-fetches all ~60
// constants with vector loads every step — measured: 58 % of k_forward's wave cycles in
// s_waitcnt.)  With per-chain models the kernels fall back to per-lane global loads.
template <int N>
struct CstArg {
    double v[N];
};
template <bool UNI, int N>
using CstArgFor = CstArg<UNI ? N : 1>;

// Vp = A V A' + P ; also returns T = A V (needed by the smoother gain)
template <int D, class PT = CPtr>
__device__ __forceinline__ void predict_cov(const CPtr A, const PT P, const Sym<D>& V, double (&T)[D][D],
                                            Sym<D>& Vp) {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) s += A[i * D + k] * V(k, j);
            T[i][j] = s;
        }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double s = P[sidx(i, j)];
#pragma unroll
            for (int k = 0; k < D; ++k) s += T[i][k] * A[j * D + k];
            Vp(i, j) = s;
        }
}

template <int D>
__device__ __forceinline__ void matvec_c(const CPtr A, const double (&x)[D], double (&y)[D]) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) s += A[i * D + k] * x[k];
        y[i] = s;
    }
}

// Observation update in information form (product of the forward message with the `*`_B(:in)
// message) and the evidence term.  In: predicted (mp, Vp), y.  Out: filtered (m, V).
// When FE: log p(y_t | y_<t) = −½[quad + log(detprod)], returned in two parts so that the caller
// can take ONE logarithm per segment (running product with exponent extraction) instead of one
// per step:  quad = c0 + y'Q⁻¹y − ξf'm_f + m_p'Λ_p m_p,   detprod = det Λf · det Vp.
// The constants of the observation side (B'Q⁻¹B packed, B'Q⁻¹, Q⁻¹ packed, c0) as plain arrays: k_forward keeps them in
// VECTOR registers for shared-model batches (see there), every other caller reads them through the constant pointer.
template <int D, int DY>
struct ObsCst {
    double lobs[D * (D + 1) / 2], g[D * DY], qi[DY * (DY + 1) / 2], c0;
    __device__ __forceinline__ void load(const double* p) {
        using CL = CstLayout<D, DY>;
#pragma unroll
        for (int i = 0; i < D * (D + 1) / 2; ++i) lobs[i] = p[CL::LOBS + i];
#pragma unroll
        for (int i = 0; i < D * DY; ++i) g[i] = p[CL::G + i];
#pragma unroll
        for (int i = 0; i < DY * (DY + 1) / 2; ++i) qi[i] = p[CL::QI + i];
        c0 = p[CL::C0];
    }
};
// A `missing` observation (docs/src/manuals/inference/static.md:98-123; the host marks it with NaNs): no message arrives from
// the observation branch, the product is the forward message itself and the step contributes no evidence term.
template <int DY>
__device__ __forceinline__ bool obs_missing(const double (&y)[DY]) {
    bool miss = false;
#pragma unroll
    for (int k = 0; k < DY; ++k) miss = miss || (y[k] != y[k]);
    return miss;
}
template <int D, int DY, bool FE, bool MASKED = false>
__device__ __forceinline__ void obs_update(const ObsCst<D, DY>& oc, const double (&mp)[D], const Sym<D>& Vp,
                                           const double (&yin)[DY], double (&m)[D], Sym<D>& V, bool& ok,
                                           double& quad, double& detprod, bool miss = false) {
    Sym<D> Lp, Lf;
    double detp, detl;
    ok = spd_inv<D>(Vp, Lp, detp) && ok;  // weightedmean_precision of the forward message
    double xp[D], xf[D], y[DY];
    const double wgt = (MASKED && miss) ? 0.0 : 1.0;
#pragma unroll
    for (int k = 0; k < DY; ++k) y[k] = (MASKED && miss) ? 0.0 : yin[k];
    symv<D>(Lp, mp, xp);
#pragma unroll
    for (int i = 0; i < D * (D + 1) / 2; ++i) Lf.v[i] = MASKED ? Lp.v[i] + wgt * oc.lobs[i] : Lp.v[i] + oc.lobs[i];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double s = xp[i];
#pragma unroll
        for (int k = 0; k < DY; ++k) s += oc.g[i * DY + k] * y[k];
        xf[i] = s;
    }
    ok = spd_inv<D>(Lf, V, detl) && ok;  // mean_cov of the product
    symv<D>(V, xf, m);
    if (!FE) return;
    double q = 0.0;
#pragma unroll
    for (int i = 0; i < DY; ++i) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < DY; ++k) s += oc.qi[sidx(i, k)] * y[k];
        q += s * y[i];
    }
    double a1 = 0.0, a2 = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        a1 += xf[i] * m[i];
        a2 += xp[i] * mp[i];
    }
    quad = (MASKED ? wgt * oc.c0 : oc.c0) + q - a1 + a2;
    detprod = detl * detp;
}
template <int D, int DY, bool FE, bool MASKED = false>
__device__ __forceinline__ void obs_update(const CPtr c, const double (&mp)[D], const Sym<D>& Vp,
                                           const double (&y)[DY], double (&m)[D], Sym<D>& V, bool& ok,
                                           double& quad, double& detprod, bool miss = false) {
    ObsCst<D, DY> oc;
    oc.load(c.p);
    obs_update<D, DY, FE, MASKED>(oc, mp, Vp, y, m, V, ok, quad, detprod, miss);
}

// running Σ log(x_t) as log(Π x_t): mantissa product renormalised every step (v_frexp_*), exponents
// summed exactly; one log at the end.
struct LogProd {
    double mant = 1.0;
    long long expo = 0;
    __device__ __forceinline__ void mul(double x) {
        double t = mant * x;
        int e = __builtin_amdgcn_frexp_exp(t);
        mant = __builtin_amdgcn_frexp_mant(t);
        expo += e;
    }
    __device__ __forceinline__ double value() const { return log(mant) + 0.6931471805599453094 * (double)expo; }
};

// record I/O.  A Gaussian record is NP = D + NS doubles: vector, then packed lower triangle.
// filt layout [T][chain/64][NP2][64][2]: lane = chain % 64; every 16-byte access of a wave is one
// contiguous 1 KiB run and the NP2 accesses of a wave-step cover one contiguous NP2 KiB block.
// Shared-model batches: V_f(t) does not depend on the data, so it is bitwise identical in every chain of a
// model (same instruction sequence, same inputs).  Every chain still COMPUTES it, but only the mean part of
// the forward message is stored per chain ([T][chain/64][MP2][64][2]); the covariance is stored once per model
// (written by chain 0) and read back as a broadcast — 32 instead of 112 B per (chain, step) each way at d = 4.
template <int D>
struct DimM {
    static constexpr int MP2 = (D + 1) / 2;  // mean part in 16-byte pairs
};
template <int D>
__device__ __forceinline__ void store_filt_sh(const Params& p, long long t, long long chain, const double (&m)[D],
                                              const Sym<D>& V) {
    constexpr int MP2 = DimM<D>::MP2;
    double r[2 * MP2];
#pragma unroll
    for (int i = 0; i < D; ++i) r[i] = m[i];
    if (D < 2 * MP2) r[2 * MP2 - 1] = 0.0;
    double2* base = reinterpret_cast<double2*>(p.filt) + ((t * p.nb64 + (chain >> 6)) * MP2) * 64 + (chain & 63);
#pragma unroll
    for (int k = 0; k < MP2; ++k) base[k * 64] = make_double2(r[2 * k], r[2 * k + 1]);
    if (chain == 0) {
        double* v = p.vtab + t * Dim<D>::NS;
#pragma unroll
        for (int i = 0; i < Dim<D>::NS; ++i) v[i] = V.v[i];
    }
}
template <int D>
__device__ __forceinline__ void load_filt_m_sh(const Params& p, long long t, long long chain, double2 (&r)[DimM<D>::MP2]) {
    constexpr int MP2 = DimM<D>::MP2;
    const double2* base = reinterpret_cast<const double2*>(p.filt) + ((t * p.nb64 + (chain >> 6)) * MP2) * 64 + (chain & 63);
#pragma unroll
    for (int k = 0; k < MP2; ++k) r[k] = base[k * 64];
}
template <int D>
__device__ __forceinline__ void unpack_m_sh(const double2 (&r)[DimM<D>::MP2], double (&m)[D]) {
    double f[2 * DimM<D>::MP2];
#pragma unroll
    for (int k = 0; k < DimM<D>::MP2; ++k) {
        f[2 * k] = r[k].x;
        f[2 * k + 1] = r[k].y;
    }
#pragma unroll
    for (int i = 0; i < D; ++i) m[i] = f[i];
}
template <int D>
__device__ __forceinline__ void load_v_sh(const Params& p, long long t, Sym<D>& V) {
    const double* v = p.vtab + t * Dim<D>::NS;  // wave-uniform address
#pragma unroll
    for (int i = 0; i < Dim<D>::NS; ++i) V.v[i] = v[i];
}

template <int D>
__device__ __forceinline__ void store_filt(double* filt, long long t, long long n_chains, long long chain,
                                           const double (&m)[D], const Sym<D>& V) {
    constexpr int NP = Dim<D>::NP, NP2 = Dim<D>::NP2;
    double r[2 * NP2];
#pragma unroll
    for (int i = 0; i < D; ++i) r[i] = m[i];
#pragma unroll
    for (int i = 0; i < Dim<D>::NS; ++i) r[D + i] = V.v[i];
    if (NP < 2 * NP2) r[2 * NP2 - 1] = 0.0;
    const long long nb64 = (n_chains + 63) >> 6;
    double2* base = reinterpret_cast<double2*>(filt) + ((t * nb64 + (chain >> 6)) * NP2) * 64 + (chain & 63);
#pragma unroll
    for (int k = 0; k < NP2; ++k) base[k * 64] = make_double2(r[2 * k], r[2 * k + 1]);
}
// the rows of a record that hold the mean (MP2 = ⌈D/2⌉ of NP2; odd D: the last of them also carries V[0]) — what the sweeps of a
// time-invariant per-chain model write and read once the filter covariance has reached its fixed point (TINV variants below)
template <int D>
__device__ __forceinline__ void store_filt_mean(double* filt, long long t, long long n_chains, long long chain, const double (&m)[D], const Sym<D>& V) {
    constexpr int NP2 = Dim<D>::NP2, MP2 = (D + 1) / 2;
    double r[2 * MP2];
#pragma unroll
    for (int i = 0; i < D; ++i) r[i] = m[i];
    if (D < 2 * MP2) r[D] = V.v[0];
    const long long nb64 = (n_chains + 63) >> 6;
    double2* base = reinterpret_cast<double2*>(filt) + ((t * nb64 + (chain >> 6)) * NP2) * 64 + (chain & 63);
#pragma unroll
    for (int k = 0; k < MP2; ++k) base[k * 64] = make_double2(r[2 * k], r[2 * k + 1]);
}
template <int D>
__device__ __forceinline__ void load_filt_mean(const double* filt, long long t, long long n_chains, long long chain, double2 (&r)[(D + 1) / 2]) {
    constexpr int NP2 = Dim<D>::NP2, MP2 = (D + 1) / 2;
    const long long nb64 = (n_chains + 63) >> 6;
    const double2* base = reinterpret_cast<const double2*>(filt) + ((t * nb64 + (chain >> 6)) * NP2) * 64 + (chain & 63);
#pragma unroll
    for (int k = 0; k < MP2; ++k) r[k] = base[k * 64];
}
template <int D>
__device__ __forceinline__ void load_filt_raw(const double* filt, long long t, long long n_chains,
                                              long long chain, double2 (&r)[Dim<D>::NP2]) {
    constexpr int NP2 = Dim<D>::NP2;
    const long long nb64 = (n_chains + 63) >> 6;
    const double2* base = reinterpret_cast<const double2*>(filt) + ((t * nb64 + (chain >> 6)) * NP2) * 64 + (chain & 63);
#pragma unroll
    for (int k = 0; k < NP2; ++k) r[k] = base[k * 64];
}
template <int D>
__device__ __forceinline__ void unpack_rec(const double2 (&r)[Dim<D>::NP2], double (&m)[D], Sym<D>& V) {
    constexpr int NP = Dim<D>::NP;
    double f[2 * Dim<D>::NP2];
#pragma unroll
    for (int k = 0; k < Dim<D>::NP2; ++k) {
        f[2 * k] = r[k].x;
        f[2 * k + 1] = r[k].y;
    }
#pragma unroll
    for (int i = 0; i < D; ++i) m[i] = f[i];
#pragma unroll
    for (int i = 0; i < NP - D; ++i) V.v[i] = f[D + i];
}
// SoA boundary records [slot][NP][chain]
template <int D>
__device__ __forceinline__ void store_soa(double* buf, long long slot, long long n_chains, long long chain,
                                          const double (&m)[D], const Sym<D>& V) {
    double* b = buf + (slot * Dim<D>::NP) * n_chains + chain;
#pragma unroll
    for (int i = 0; i < D; ++i) b[i * n_chains] = m[i];
#pragma unroll
    for (int i = 0; i < Dim<D>::NS; ++i) b[(D + i) * n_chains] = V.v[i];
}
template <int D>
__device__ __forceinline__ void load_soa(const double* buf, long long slot, long long n_chains, long long chain,
                                         double (&m)[D], Sym<D>& V) {
    const double* b = buf + (slot * Dim<D>::NP) * n_chains + chain;
#pragma unroll
    for (int i = 0; i < D; ++i) m[i] = b[i * n_chains];
#pragma unroll
    for (int i = 0; i < Dim<D>::NS; ++i) V.v[i] = b[(D + i) * n_chains];
}
template <int DY>
__device__ __forceinline__ void load_y(const double* y, long long t, long long n_chains, long long chain,
                                       double (&v)[DY]) {
    const double* p = y + (t * n_chains + chain) * DY;
    if constexpr (DY % 2 == 0) {
        const double2* p2 = reinterpret_cast<const double2*>(p);
#pragma unroll
        for (int k = 0; k < DY / 2; ++k) {
            double2 q = p2[k];
            v[2 * k] = q.x;
            v[2 * k + 1] = q.y;
        }
    } else {
#pragma unroll
        for (int k = 0; k < DY; ++k) v[k] = p[k];
    }
}

template <bool UNI>
__device__ __forceinline__ int model_of(const Params& p, long long chain) {
    if (UNI) return 0;
    return p.chain_model ? p.chain_model[chain] : 0;
}
// time-varying constants A_t, P_t, B_t, Q_t: the model of time index t
template <bool UNI>
__device__ __forceinline__ int model_at(const Params& p, int chain_mdl, long long t) {
    if constexpr (UNI) return 0;
    else return p.step_model ? p.step_model[t] : chain_mdl;
}

// segment s covers times (1-based) b_s+1 .. b_{s+1}, b_s = 1 + s·L, b_S = T
__device__ __forceinline__ long long seg_len(const Params& p, long long s) {
    long long b0 = 1 + s * p.L;
    long long b1 = b0 + p.L;
    if (b1 > p.T) b1 = p.T;
    return b1 - b0;
}

// ------------------------------------------------------------------------------------------
// phase 1: data-dependent part (b, η) of each segment element.
//   e_i = y_i − (BA) m_{i−1};  η += U_i e_i;  m_i = A m_{i−1} + K_i e_i       (m_0 = 0)
// K_i, U_i: per-model gain tables (Kalman gain of the filter started from an exactly known
// state, and the sensitivity of the innovations to that state).  64 FMAs / step at d = dy = 4.
// (The phases of the four-phase schedule are written as wave-level bodies — `g`: the (chain, segment) lane of the batch, `lane`: the lane of
// the wavefront, LDS handed in by the caller — so that k_small_sweep below can run all of them in ONE launch; the kernels proper wrap them.)
template <int D, int DY>
struct AggStage {
    static constexpr int U = 4;                                   // steps per chunk
    static constexpr int NPC = U * TabLayout<D, DY>::SIZE / 2;    // 16-byte pieces of gain table per chunk
};
template <int D, int DY, bool UNI>
__device__ __forceinline__ void seg_aggregate_body(const Params& p, const CstArgFor<UNI, CstLayout<D, DY>::SIZE>& cb, const long long g, const int lane,
                                                   double2* __restrict__ tbuf_flat) {
    using CL = CstLayout<D, DY>;
    using TL = TabLayout<D, DY>;
    constexpr int U = AggStage<D, DY>::U;
    constexpr int NPC = AggStage<D, DY>::NPC;
    constexpr int PPL = (NPC + 63) / 64;  // pieces per lane
    // The per-offset gains are the same for every lane of the wave (one model): the wave streams
    // them cooperatively global -> registers -> LDS one chunk ahead (double buffered) and reads
    // them back with broadcast ds_reads.  (Scalar loads of the table stall the wave on every
    // step: SMEM returns out of order, so each s_load needs lgkmcnt(0) before first use.)
    double2 (*tbuf)[UNI ? NPC : 1] = reinterpret_cast<double2 (*)[UNI ? NPC : 1]>(tbuf_flat);   // [2][NPC], private to the wavefront
    const long long total = p.n_chains * (long long)p.S;
    const bool live = g < total;
    const long long seg = live ? g / p.n_chains : 0;
    const long long chain = live ? g - seg * p.n_chains : 0;
    const long long len = live ? seg_len(p, seg) : 0;
    const int mdl = model_of<UNI>(p, chain);
    const CPtr c{UNI ? cb.v : p.cst + (long long)mdl * CL::SIZE};
    const double* tab = p.tab + (long long)mdl * p.L * TL::SIZE;
    const long long t0 = seg * p.L + 1;  // zero-based index of the first observation of the segment
    const long long tab_pieces = p.L * TL::SIZE / 2;

    double2 tr[PPL];
    auto fetch = [&](long long i0) {
        const double2* t2 = reinterpret_cast<const double2*>(p.tab);
#pragma unroll
        for (int k = 0; k < PPL; ++k) {
            const int idx = k * 64 + lane;
            const long long gi = i0 * (TL::SIZE / 2) + idx;
            tr[k] = (idx < NPC && gi < tab_pieces) ? t2[gi] : make_double2(0.0, 0.0);
        }
    };
    auto stash = [&](int b) {
#pragma unroll
        for (int k = 0; k < PPL; ++k) {
            const int idx = k * 64 + lane;
            if (idx < NPC) tbuf[b][idx] = tr[k];
        }
    };
    if (UNI) {
        fetch(0);
        stash(0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    double m[D], eta[D];
#pragma unroll
    for (int i = 0; i < D; ++i) m[i] = eta[i] = 0.0;
    // observations are prefetched one chunk (U steps) ahead: they do not depend on the recursion
    double yb[U][DY], yn[U][DY];
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (u < len) load_y<DY>(p.y, t0 + u, p.n_chains, chain, yn[u]);
    int b = 0;
    for (long long i0 = 0; i0 < p.L; i0 += U, b ^= 1) {  // uniform trip count
        if (UNI && i0 + U < p.L) fetch(i0 + U);
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int k = 0; k < DY; ++k) yb[u][k] = yn[u][k];
            if (i0 + U + u < len) load_y<DY>(p.y, t0 + i0 + U + u, p.n_chains, chain, yn[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long i = i0 + u;
            if (i < len) {
                const CPtr tb{UNI ? reinterpret_cast<const double*>(&tbuf[b][0]) + u * TL::SIZE : tab + i * TL::SIZE};
                double e[DY];
#pragma unroll
                for (int a = 0; a < DY; ++a) {
                    double s = yb[u][a];
#pragma unroll
                    for (int k = 0; k < D; ++k) s -= c[CL::HF + a * D + k] * m[k];
                    e[a] = s;
                }
                double mn[D];
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    double s = 0.0, uu = eta[a];
#pragma unroll
                    for (int k = 0; k < D; ++k) s += c[CL::A + a * D + k] * m[k];
#pragma unroll
                    for (int k = 0; k < DY; ++k) {
                        s += tb[TL::K + a * DY + k] * e[k];
                        uu += tb[TL::U + a * DY + k] * e[k];
                    }
                    mn[a] = s;
                    eta[a] = uu;
                }
#pragma unroll
                for (int a = 0; a < D; ++a) m[a] = mn[a];
            }
        }
        if (UNI) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (i0 + U < p.L) stash(b ^ 1);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (live) {
        double* o = p.elem + (seg * 2 * D) * p.n_chains + chain;
#pragma unroll
        for (int a = 0; a < D; ++a) {
            o[a * p.n_chains] = m[a];
            o[(D + a) * p.n_chains] = eta[a];
        }
    }
}
template <int D, int DY, bool UNI>
__global__ void __launch_bounds__(64) k_seg_aggregate(Params p, const CstArgFor<UNI, CstLayout<D, DY>::SIZE> cb) {
    __shared__ double2 tbuf[2 * (UNI ? AggStage<D, DY>::NPC : 1)];
    seg_aggregate_body<D, DY, UNI>(p, cb, (long long)blockIdx.x * blockDim.x + threadIdx.x, (int)threadIdx.x, tbuf);
}

// ------------------------------------------------------------------------------------------
// phase 1 without tables: the WHOLE segment element (Π, b, C, η, J), computed in the lane.
//
// With `missing` observations the gains of the known-start filter depend on the chain's own pattern of observed steps, with
// per-step constants on the time index: no per-position table exists.  The lane then runs the known-start filter itself —
// the information-form update of k_forward from (m, V) = (0, 0) — and carries the dependence on the unknown start state
// along:  m_i(x_s) = Π_i x_s + b_i.  With Z = A_i Π_{i−1}, Λp = V_p⁻¹, Y = Λp Z and the filtered covariance V_i:
//     Π_i = V_i Y,    b_i = V_i (Λp A b_{i−1} + B′Q⁻¹ y_i),    C = V_L,
//     η += Y′ (b_i − A b_{i−1}),          J += Z′ (Y − Λp Π_i)
// (the innovation form  η += (B Z)′S⁻¹e,  J += (B Z)′S⁻¹(B Z)  with S⁻¹ = Q⁻¹ − Q⁻¹B V B′Q⁻¹ pushed through B: neither B nor Q
// is needed, only the constants k_forward uses).  A missing step has V_i = V_p, b_i = A b_{i−1}: both increments vanish.
template <int D>
struct ElemX {
    static constexpr int NS = D * (D + 1) / 2;
    static constexpr int PI = 0, J = D * D, C = D * D + NS, SIZE = D * D + 2 * NS;
};
template <int D, int DY, bool TINV = false>   // TINV: time-invariant models (no masks, no per-step constants) — the variant with the frozen tail
__global__ void __launch_bounds__(64, 2) k_seg_elements(Params p) {  // two wavefronts per SIMD: the batch is sized for that
    using CL = CstLayout<D, DY>;
    using EX = ElemX<D>;
    constexpr int NS = Dim<D>::NS;
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= p.n_chains * (long long)p.S) return;
    const long long seg = g / p.n_chains, chain = g - seg * p.n_chains;
    const long long len = seg_len(p, seg);
    const long long t0 = seg * p.L + 1;
    const int cmdl = model_of<false>(p, chain);
    bool ok = true;
    double b[D], eta[D], Pi[D][D];
    Sym<D> V, J;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        b[i] = eta[i] = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) Pi[i][j] = i == j ? 1.0 : 0.0;
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) V.v[i] = J.v[i] = 0.0;
    // the ≈50 constants of a step stay with the lane for the whole segment (per-step constants reload them every step): A and P in registers; at
    // D = 4 the 10 + 4·DY doubles of Λ_obs and G — each read once per step — in LDS, [k][lane]: the state of the recursion (44 doubles), the
    // constants and a step's temporaries do not fit 256 registers at two wavefronts per SIMD, and LDS is the cheaper place to spill to
    constexpr bool CLDS = D == 4;
    constexpr int LB = !CLDS ? CL::QI : TINV ? CL::P : CL::LOBS;   // first constant kept in LDS (the time-invariant variant parks P there too)
    __shared__ double cl[CLDS ? (CL::QI - LB + (TINV ? 3 : 0)) * 64 : 1];
    double cr[LB];  // A (| P | LOBS | G)
    auto load_c = [&](const double* src) {
#pragma unroll
        for (int k = 0; k < LB; ++k) cr[k] = src[k];
        if constexpr (CLDS) {
#pragma unroll
            for (int k = 0; k < CL::QI - LB; ++k) cl[k * 64 + threadIdx.x] = src[LB + k];
        }
    };
    int lo = (int)threadIdx.x;   // the lane's column of the LDS block (laundered once per step: see the loop)
    auto c_lobs = [&](int q) -> double {
        if constexpr (CLDS) return cl[(CL::LOBS - LB + q) * 64 + lo];
        else return cr[CL::LOBS + q];
    };
    auto c_g = [&](int k) -> double {
        if constexpr (CLDS) return cl[(CL::G - LB + k) * 64 + lo];
        else return cr[CL::G + k];
    };
    auto c_P = [&]() {
        if constexpr (CLDS && LB == CL::P) return LdsLanePtr{cl + lo};
        else return CPtr{cr + CL::P};
    };
    load_c(p.cst + (long long)cmdl * CL::SIZE);
    if constexpr (CLDS && TINV) {
#pragma unroll
        for (int k = 0; k < 3; ++k) cl[(CL::QI - LB + k) * 64 + threadIdx.x] = 0.0;
    }
    // Time-invariant models (no masks, no per-step constants): the covariance recursion of the known-start filter reaches its fixed point after
    // the filter's mixing time (≈ 250 of a segment's 782 steps at the BASELINE model) — from there on V, Λp and the gains are CONSTANT, and what
    // still moves is Π ← F Π (F = V Λp A, the closed loop), b, and the two accumulations that are bilinear in Π (they die out with it).  The loop
    // below stops where V and J have stopped moving (every lane of the wavefront) and a second kernel runs the frozen recursion over the rest of
    // the segment: 150 instead of 640 FMAs per step until Π is below 10⁻²⁰, then 36.  Same numbers to rounding (tests/test_converged_elements_gpu.py).
    constexpr bool tinv = TINV;   // (a kernel of its own: in one kernel the tail's registers cost the masked sweeps 30 % — 148 → 276 B of scratch)
    double cf1 = 0.0, cf2 = 0.0, cf3 = 0.0;
    int nsame = 0;
    long long i = 0;
    bool go = true;   // TINV: until the fixed point (the step that finds it is counted: the tail starts behind it)
    for (; i < len && go; ++i) {
        const long long t = t0 + i;
        // (without this the compiler reads the next step's G out of LDS at the end of this one — and spills it over the back edge)
        if constexpr (CLDS && TINV) asm volatile("" : "+v"(lo));
        if (!TINV && p.step_model) load_c(p.cst + (long long)p.step_model[t] * CL::SIZE);
        const double* ct = cr;
        double yv[DY];
        load_y<DY>(p.y, t, p.n_chains, chain, yv);
        const bool miss = !TINV && p.masked && obs_missing<DY>(yv);
        double mp[D], T[D][D], Z[D][D];
        Sym<D> Vp, Lp, Lf, Vn;
        matvec_c<D>(CPtr{ct + CL::A}, b, mp);
        predict_cov<D>(CPtr{ct + CL::A}, c_P(), V, T, Vp);
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
            for (int c = 0; c < D; ++c) {
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) acc += ct[CL::A + a * D + k] * Pi[k][c];
                Z[a][c] = acc;
            }
        double det;
        ok = spd_inv<D>(Vp, Lp, det) && ok;
        const double wgt = miss ? 0.0 : 1.0;
#pragma unroll
        for (int q = 0; q < NS; ++q) Lf.v[q] = Lp.v[q] + wgt * c_lobs(q);
        ok = spd_inv<D>(Lf, Vn, det) && ok;
        double xf[D], bn[D];
        symv<D>(Lp, mp, xf);
        // (G is read HERE, behind Λp·mp: read at the top of the step with the other constants it waits in scratch for this line)
        if constexpr (CLDS && TINV) asm volatile("" : "+v"(lo), "+v"(xf[0]));
#pragma unroll
        for (int a = 0; a < D; ++a) {
            double acc = xf[a];
#pragma unroll
            for (int k = 0; k < DY; ++k) acc += c_g(a * DY + k) * (miss ? 0.0 : yv[k]);
            xf[a] = acc;
        }
        symv<D>(Vn, xf, bn);
        double Y[D][D], Pn[D][D];
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
            for (int c = 0; c < D; ++c) {
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) acc += Lp(a, k) * Z[k][c];
                Y[a][c] = acc;
            }
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
            for (int c = 0; c < D; ++c) {
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) acc += Vn(a, k) * Y[k][c];
                Pn[a][c] = acc;
            }
        double db[D];
#pragma unroll
        for (int a = 0; a < D; ++a) db[a] = bn[a] - mp[a];
#pragma unroll
        for (int a = 0; a < D; ++a) {
            double acc = eta[a];
#pragma unroll
            for (int k = 0; k < D; ++k) acc += Y[k][a] * db[k];
            eta[a] = acc;
        }
        // J += Z′ (Y − Λp Π_i)   (symmetric: lower triangle; one row of R = Y − Λp Π_i at a time — the whole of R does not fit the register file)
#pragma unroll
        for (int k = 0; k < D; ++k) {
            double Rk[D];
#pragma unroll
            for (int c = 0; c < D; ++c) {
                double acc = Y[k][c];
#pragma unroll
                for (int q = 0; q < D; ++q) acc -= Lp(k, q) * Pn[q][c];
                Rk[c] = acc;
            }
#pragma unroll
            for (int a = 0; a < D; ++a)
#pragma unroll
                for (int c = 0; c <= a; ++c) J(a, c) += Z[k][a] * Rk[c];
            if constexpr (D == 4) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int a = 0; a < D; ++a) {
            b[a] = bn[a];
#pragma unroll
            for (int c = 0; c < D; ++c) Pi[a][c] = Pn[a][c];
        }
        V = Vn;
        if constexpr (tinv) {
            // fixed point of the covariance recursion: two independent linear functionals of V unchanged to 2 ulp on two steps in a row (an
            // entrywise comparison would keep the previous V alive through the whole step — 120 more bytes of scratch in a kernel that has none to give)
            double f1, f2, f3, f3b;
            fixpoint_functionals<D>(V, f1, f2);
            fixpoint_functionals<D>(J, f3b, f3);   // J stops moving later than V (its increments are quadratic in Π): the tail does not touch it
            f3 += 0.21 * f3b;
            bool same;
            if constexpr (CLDS) {   // the previous step's functionals wait in LDS as well: six registers the step does not have
                double* cf = cl + (CL::QI - LB) * 64 + lo;
                const double g1 = cf[0], g2 = cf[64], g3 = cf[128];   // (no short circuit: three reads, no branches)
                same = (int)(fabs(f1 - g1) <= 4.5e-16 * fabs(f1)) & (int)(fabs(f2 - g2) <= 4.5e-16 * fabs(f2)) & (int)(fabs(f3 - g3) <= 2.3e-16 * fabs(f3));
                cf[0] = f1;
                cf[64] = f2;
                cf[128] = f3;
            } else {
                same = fabs(f1 - cf1) <= 4.5e-16 * fabs(f1) && fabs(f2 - cf2) <= 4.5e-16 * fabs(f2) && fabs(f3 - cf3) <= 2.3e-16 * fabs(f3);
                cf1 = f1;
                cf2 = f2;
                cf3 = f3;
            }
            nsame = same ? nsame + 1 : 0;
            go = !__all(nsame >= 2);
        }
    }
    if constexpr (TINV) p.fstart[(seg * Dim<D>::NP) * p.n_chains + chain] = (double)i;   // where the tail kernel takes over (the scan overwrites the slot later)
    double* o = p.elem + (seg * 2 * D) * p.n_chains + chain;
    asm volatile("" : "+v"(o));
#pragma unroll
    for (int a = 0; a < D; ++a) {
        o[a * p.n_chains] = b[a];
        o[(D + a) * p.n_chains] = eta[a];
    }
    double* x = p.elemx + (seg * EX::SIZE) * p.n_chains + chain;
    asm volatile("" : "+v"(x));   // (the 36 addresses below are formed here, not hoisted above the loop)
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int c = 0; c < D; ++c) x[(EX::PI + a * D + c) * p.n_chains] = Pi[a][c];
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        x[(EX::J + q) * p.n_chains] = J.v[q];
        x[(EX::C + q) * p.n_chains] = V.v[q];
    }
    if (!ok) atomicOr(p.status, ST_NOT_POSDEF);
}
// The tail of a segment behind the fixed point of its covariance recursion (k_seg_elements<…, TINV = true> stops there and leaves its state in
// the element arrays and the step index in the first slot of the segment's boundary record): a kernel of its own, because its few registers let
// many wavefronts hide what is now a chain of dependent loads and short products.
template <int D, int DY>
__global__ void __launch_bounds__(64, 2) k_seg_elements_tail(Params p) {
    using CL = CstLayout<D, DY>;
    using EX = ElemX<D>;
    constexpr int NS = Dim<D>::NS;
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= p.n_chains * (long long)p.S) return;
    const long long seg = g / p.n_chains, chain = g - seg * p.n_chains;
    const long long len = seg_len(p, seg);
    const long long t0 = seg * p.L + 1;
    long long i = (long long)p.fstart[(seg * Dim<D>::NP) * p.n_chains + chain];
    if (i >= len) return;
    bool ok = true;
    double cr[CL::M1];
    {
        const double* src = p.cst + (long long)model_of<false>(p, chain) * CL::SIZE;
#pragma unroll
        for (int k = 0; k < CL::M1; ++k) cr[k] = src[k];
    }
    double b[D], eta[D], Pi[D][D];
    Sym<D> V;
    double* o = p.elem + (seg * 2 * D) * p.n_chains + chain;
    double* x = p.elemx + (seg * EX::SIZE) * p.n_chains + chain;
#pragma unroll
    for (int a = 0; a < D; ++a) {
        b[a] = o[a * p.n_chains];
        eta[a] = o[(D + a) * p.n_chains];
#pragma unroll
        for (int c = 0; c < D; ++c) Pi[a][c] = x[(EX::PI + a * D + c) * p.n_chains];
    }
#pragma unroll
    for (int q = 0; q < NS; ++q) V.v[q] = x[(EX::C + q) * p.n_chains];
    if (i < len) {   // (wave-uniform: every lane of the wavefront left the full recursion at the same step)
        const double* ct = cr;
        double T[D][D], LA[D][D], F[D][D], Gy[D][DY];
        Sym<D> Vp, Lp, Lf, Vn;
        double det;
        predict_cov<D>(CPtr{ct + CL::A}, CPtr{ct + CL::P}, V, T, Vp);
        ok = spd_inv<D>(Vp, Lp, det) && ok;
#pragma unroll
        for (int q = 0; q < NS; ++q) Lf.v[q] = Lp.v[q] + ct[CL::LOBS + q];
        ok = spd_inv<D>(Lf, Vn, det) && ok;
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
            for (int c = 0; c < D; ++c) {
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) acc += Lp(a, k) * ct[CL::A + k * D + c];
                LA[a][c] = acc;   // Λp A
            }
#pragma unroll
        for (int a = 0; a < D; ++a) {
#pragma unroll
            for (int c = 0; c < D; ++c) {
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) acc += Vn(a, k) * LA[k][c];
                F[a][c] = acc;    // V Λp A
            }
#pragma unroll
            for (int c = 0; c < DY; ++c) {
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) acc += Vn(a, k) * ct[CL::G + k * DY + c];
                Gy[a][c] = acc;   // V B′Q⁻¹
            }
        }
        bool pilive = true;
        // the tail is a short dependent chain per step: the observations travel two steps ahead of it
        double y1[DY], y2[DY];
        load_y<DY>(p.y, t0 + i, p.n_chains, chain, y1);
        load_y<DY>(p.y, t0 + (i + 1 < len ? i + 1 : i), p.n_chains, chain, y2);
        for (; i < len; ++i) {
            double yv[DY], bn[D];
#pragma unroll
            for (int k = 0; k < DY; ++k) {
                yv[k] = y1[k];
                y1[k] = y2[k];
            }
            load_y<DY>(p.y, t0 + (i + 2 < len ? i + 2 : len - 1), p.n_chains, chain, y2);
#pragma unroll
            for (int a = 0; a < D; ++a) {
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) acc += F[a][k] * b[k];
#pragma unroll
                for (int k = 0; k < DY; ++k) acc += Gy[a][k] * yv[k];
                bn[a] = acc;
            }
            if (pilive) {
                double db[D], w[D];
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    double acc = bn[a];
#pragma unroll
                    for (int k = 0; k < D; ++k) acc -= ct[CL::A + a * D + k] * b[k];
                    db[a] = acc;
                }
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    double acc = 0.0;
#pragma unroll
                    for (int a = 0; a < D; ++a) acc += LA[a][k] * db[a];
                    w[k] = acc;
                }
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    double acc = eta[c];
#pragma unroll
                    for (int k = 0; k < D; ++k) acc += Pi[k][c] * w[k];
                    eta[c] = acc;
                }
                double Pn[D][D], pmax = 0.0;
#pragma unroll
                for (int a = 0; a < D; ++a)
#pragma unroll
                    for (int c = 0; c < D; ++c) {
                        double acc = 0.0;
#pragma unroll
                        for (int k = 0; k < D; ++k) acc += F[a][k] * Pi[k][c];
                        Pn[a][c] = acc;
                        pmax = fmax(pmax, fabs(acc));
                    }
#pragma unroll
                for (int a = 0; a < D; ++a)
#pragma unroll
                    for (int c = 0; c < D; ++c) Pi[a][c] = Pn[a][c];
                pilive = __any(pmax > 1e-20);
            }
#pragma unroll
            for (int a = 0; a < D; ++a) b[a] = bn[a];
        }
    }
    // (the addresses are formed again here: kept from the loads above, the 24 of them would sit in 48 registers through the whole loop)
    asm volatile("" : "+v"(o), "+v"(x));
#pragma unroll
    for (int a = 0; a < D; ++a) {
        o[a * p.n_chains] = b[a];
        o[(D + a) * p.n_chains] = eta[a];
#pragma unroll
        for (int c = 0; c < D; ++c) x[(EX::PI + a * D + c) * p.n_chains] = Pi[a][c];
    }
    if (!ok) atomicOr(p.status, ST_NOT_POSDEF);
}
// the element matrices of (chain, segment) in AggLayout order; `derived`: also C⁻¹, C⁻¹Π, J + Π′C⁻¹Π (suffix scan)
template <int D>
__device__ __forceinline__ bool load_elemx(const Params& p, long long s, long long chain, bool derived, double (&al)[AggLayout<D>::SIZE]) {
    using AL = AggLayout<D>;
    using EX = ElemX<D>;
    constexpr int NS = Dim<D>::NS;
    const double* x = p.elemx + (s * EX::SIZE) * p.n_chains + chain;
#pragma unroll
    for (int q = 0; q < D * D; ++q) al[AL::PI + q] = x[(EX::PI + q) * p.n_chains];
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        al[AL::J + q] = x[(EX::J + q) * p.n_chains];
        al[AL::C + q] = x[(EX::C + q) * p.n_chains];
    }
    if (!derived) return true;
    Sym<D> C, Ci;
    double det;
#pragma unroll
    for (int q = 0; q < NS; ++q) C.v[q] = al[AL::C + q];
    const bool ok = spd_inv<D>(C, Ci, det);
#pragma unroll
    for (int q = 0; q < NS; ++q) al[AL::CI + q] = Ci.v[q];
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int c = 0; c < D; ++c) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) acc += Ci(a, k) * al[AL::PI + k * D + c];
            al[AL::X + a * D + c] = acc;
        }
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int c = 0; c <= a; ++c) {
            double acc = al[AL::J + sidx(a, c)];
#pragma unroll
            for (int k = 0; k < D; ++k) acc += al[AL::PI + k * D + a] * al[AL::X + k * D + c];
            al[AL::JJ + sidx(a, c)] = acc;
        }
    return ok;
}

// ------------------------------------------------------------------------------------------
// Shared-model smoothing runs: phases 1 and 3 in ONE pass over the observations.
//
// With one model for every chain all covariances are data-independent, and the true filtered mean inside segment s
// (start belief N(m_s, V_s) at boundary b_s) is an affine function of the known-start quantities this phase already tracks:
//     m_f(b_s + i) = b_i + Π_i W_{s,i} (V_s⁻¹ m_s + η_i),      W_{s,i} = (V_s⁻¹ + J_i)⁻¹
// (the element formula of the boundary scan, applied to the prefix of length i).  With the data-independent, per-time-index
// matrices  M_t = Π_i W_{s,i}  and  N_t = M_t V_s⁻¹  (k_time_tables, once per engine)
//     m_f(t) = z_t + N_t m_s,        z_t = b_i + M_t η_i.
// z_t is stored as the forward-message mean record; the backward kernels add N_t m_s, with m_s from the boundary scan that
// runs in between.  The observations are read once and the data-independent inverses of k_forward are not recomputed in
// every lane.  The segment's evidence comes from the same quantities:
//     −2 log p(y_seg | y_before) = const_s + q0 + [m_s'A1 m_s − 2 η'A2 m_s − η'W η],   q0 = Σ_i e_i'(S⁰_i)⁻¹e_i
// with e_i the known-start innovations (k_fe_seg evaluates the bracket, const_s is summed on the host).
template <int D, int DY, bool FE>
__global__ void __launch_bounds__(64) k_forward0(Params p, const CstArg<CstLayout<D, DY>::SIZE> cb) {
    using CL = CstLayout<D, DY>;
    using FL = F0Layout<D, DY>;
    constexpr int MT = TimeTab<D>::MT;
    constexpr int MP2 = DimM<D>::MP2;
    constexpr int U = 4;                    // steps per chunk
    constexpr int NPF = U * FL::SIZE / 2;   // 16-byte pieces of the position table per chunk
    constexpr int NPM = U * MT / 2;         // … of the time table
    constexpr int NPC = NPF + NPM;
    constexpr int PPL = (NPC + 63) / 64;    // pieces per lane
    __shared__ double2 tbuf[2][NPC];
    const int lane = threadIdx.x;
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = p.n_chains * (long long)p.S;
    const bool live = g < total;
    // every wave holds 64 consecutive (segment, chain) lanes; when the batch is not a multiple of 64 a wave may straddle two
    // segments — the time-table stream follows lane 0's segment, so such waves read their own M_t rows from memory
    const long long seg = live ? g / p.n_chains : 0;
    const long long chain = live ? g - seg * p.n_chains : 0;
    const long long len = live ? seg_len(p, seg) : 0;
    const CPtr c{cb.v};
    const long long t0 = seg * p.L + 1;
    const long long seg0 = __builtin_amdgcn_readfirstlane((int)seg);
    const bool straddle = __builtin_amdgcn_ballot_w64(live && seg != seg0) != 0;
    const long long tw0 = seg0 * p.L + 1;   // first time index of the wave's table stream
    const long long f_pieces = p.L * (FL::SIZE / 2);
    const long long m_pieces = p.T * (MT / 2);
    const double2* f2 = reinterpret_cast<const double2*>(p.ftab);
    const double2* m2 = reinterpret_cast<const double2*>(p.mtab);

    double2 tr[PPL];
    auto fetch = [&](long long i0) {
#pragma unroll
        for (int k = 0; k < PPL; ++k) {
            const int idx = k * 64 + lane;
            double2 v = make_double2(0.0, 0.0);
            if (idx < NPF) {
                const long long gi = i0 * (FL::SIZE / 2) + idx;
                if (gi < f_pieces) v = f2[gi];
            } else if (idx < NPC) {
                const long long gi = (tw0 + i0) * (MT / 2) + (idx - NPF);
                if (gi < m_pieces) v = m2[gi];
            }
            tr[k] = v;
        }
    };
    auto stash = [&](int b) {
#pragma unroll
        for (int k = 0; k < PPL; ++k) {
            const int idx = k * 64 + lane;
            if (idx < NPC) tbuf[b][idx] = tr[k];
        }
    };
    fetch(0);
    stash(0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    double m[D], eta[D], q0 = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) m[i] = eta[i] = 0.0;
    double yb[U][DY], yn[U][DY];
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (u < len) load_y<DY>(p.y, t0 + u, p.n_chains, chain, yn[u]);
    int b = 0;
    for (long long i0 = 0; i0 < p.L; i0 += U, b ^= 1) {  // uniform trip count
        if (i0 + U < p.L) fetch(i0 + U);
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int k = 0; k < DY; ++k) yb[u][k] = yn[u][k];
            if (i0 + U + u < len) load_y<DY>(p.y, t0 + i0 + U + u, p.n_chains, chain, yn[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long i = i0 + u;
            if (i < len) {
                const double* tb = reinterpret_cast<const double*>(&tbuf[b][0]) + u * FL::SIZE;
                const double* mt = reinterpret_cast<const double*>(&tbuf[b][NPF]) + u * MT;
                double e[DY];
#pragma unroll
                for (int a = 0; a < DY; ++a) {
                    double s = yb[u][a];
#pragma unroll
                    for (int k = 0; k < D; ++k) s -= c[CL::HF + a * D + k] * m[k];
                    e[a] = s;
                }
                if (FE) {
#pragma unroll
                    for (int a = 0; a < DY; ++a) {
                        double s = 0.0;
#pragma unroll
                        for (int k = 0; k < DY; ++k) s += tb[FL::SI + sidx(a, k)] * e[k];
                        q0 += s * e[a];
                    }
                }
                double mn[D];
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    double s = 0.0, uu = eta[a];
#pragma unroll
                    for (int k = 0; k < D; ++k) s += c[CL::A + a * D + k] * m[k];
#pragma unroll
                    for (int k = 0; k < DY; ++k) {
                        s += tb[FL::K + a * DY + k] * e[k];
                        uu += tb[FL::U + a * DY + k] * e[k];
                    }
                    mn[a] = s;
                    eta[a] = uu;
                }
#pragma unroll
                for (int a = 0; a < D; ++a) m[a] = mn[a];
                // z_t = b_i + M_t η_i  (values, not pointers, select between LDS and memory: a pointer that may be either
                // becomes a generic pointer, which this compiler cannot lower here)
                double z[2 * MP2], mrow[D * D];
                if (straddle) {
                    const double* gm = p.mtab + (t0 + i) * MT;
#pragma unroll
                    for (int k = 0; k < D * D; ++k) mrow[k] = gm[k];
                } else {
#pragma unroll
                    for (int k = 0; k < D * D; ++k) mrow[k] = mt[k];
                }
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    double s = m[a];
#pragma unroll
                    for (int k = 0; k < D; ++k) s += mrow[a * D + k] * eta[k];
                    z[a] = s;
                }
                if (D < 2 * MP2) z[2 * MP2 - 1] = 0.0;
                double2* base = reinterpret_cast<double2*>(p.filt) + (((t0 + i) * p.nb64 + (chain >> 6)) * MP2) * 64 + (chain & 63);
#pragma unroll
                for (int k = 0; k < MP2; ++k) base[k * 64] = make_double2(z[2 * k], z[2 * k + 1]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (i0 + U < p.L) stash(b ^ 1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (live) {
        double* o = p.elem + (seg * 2 * D) * p.n_chains + chain;
#pragma unroll
        for (int a = 0; a < D; ++a) {
            o[a * p.n_chains] = m[a];
            o[(D + a) * p.n_chains] = eta[a];
        }
        if (FE) p.fe_part[(seg + 1) * p.n_chains + chain] = -0.5 * q0;
    }
}

// Data-independent per-time-index tables of the one-pass schedule, once per engine: one lane per time index.
struct TimeTabParams {
    long long T, L;
    const double* pos;   // [L][PosLayout::SIZE]
    const double* scan;  // [S][ScanLayout::SIZE]  (VB = V(b_s))
    double* mtab;        // [T][MT]
    double* ntab;        // [T][MT]
    double* vtab;        // [T][NS]   V_f(t)
    int* status;
};
template <int D>
__global__ void __launch_bounds__(64) k_time_tables(TimeTabParams q) {
    using PL = PosLayout<D>;
    using SL = ScanLayout<D>;
    constexpr int NS = Dim<D>::NS;
    constexpr int MT = TimeTab<D>::MT;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= q.T) return;
    if (t == 0) {  // the filtered covariance at x[1] is the start covariance of segment 0; rows 0 of M, N are never used
#pragma unroll
        for (int k = 0; k < NS; ++k) q.vtab[k] = q.scan[SL::VB + k];
#pragma unroll
        for (int k = 0; k < MT; ++k) q.mtab[k] = q.ntab[k] = 0.0;
        return;
    }
    const long long s = (t - 1) / q.L, i = t - s * q.L;  // position 1..L inside segment s
    const double* ps = q.pos + (i - 1) * PL::SIZE;
    const double* sc = q.scan + s * SL::SIZE;
    Sym<D> Vs, Vi, T1, W;
    double det;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < NS; ++k) Vs.v[k] = sc[SL::VB + k];
    ok = spd_inv<D>(Vs, Vi, det) && ok;
#pragma unroll
    for (int k = 0; k < NS; ++k) T1.v[k] = Vi.v[k] + ps[PL::J + k];
    ok = spd_inv<D>(T1, W, det) && ok;
    double M[D][D], N[D][D];
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) acc += ps[PL::PI + a * D + k] * W(k, b);
            M[a][b] = acc;
        }
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) acc += M[a][k] * Vi(k, b);
            N[a][b] = acc;
        }
    double* mo = q.mtab + t * MT;
    double* no = q.ntab + t * MT;
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) {
            mo[a * D + b] = M[a][b];
            no[a * D + b] = N[a][b];
        }
    if (D * D < MT) mo[MT - 1] = no[MT - 1] = 0.0;
    double* vo = q.vtab + t * NS;  // V_f(t) = C_i + Π_i W Π_i'
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) {
            double acc = ps[PL::C + sidx(a, b)];
#pragma unroll
            for (int k = 0; k < D; ++k) acc += M[a][k] * ps[PL::PI + b * D + k];
            vo[sidx(a, b)] = acc;
        }
    if (!ok) atomicOr(q.status, ST_NOT_POSDEF);
}

// The data-dependent quadratic form of every segment's evidence, after the boundary scan: one lane per (chain, segment).
template <int D>
__global__ void __launch_bounds__(64) k_fe_seg(Params p) {
    using FS = FeSegLayout<D>;
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= p.n_chains * (long long)p.S) return;
    const long long seg = g / p.n_chains, chain = g - seg * p.n_chains;
    const double* f = p.fseg + seg * FS::SIZE;
    const double* ms = p.fstart + (seg * Dim<D>::NP) * p.n_chains + chain;
    const double* el = p.elem + (seg * 2 * D) * p.n_chains + chain;
    double m[D], eta[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        m[i] = ms[i * p.n_chains];
        eta[i] = el[(D + i) * p.n_chains];
    }
    double v = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double a1 = 0.0, a2 = 0.0, w = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            a1 += f[FS::A1 + sidx(i, k)] * m[k];
            a2 += f[FS::A2 + i * D + k] * m[k];
            w += f[FS::W + sidx(i, k)] * eta[k];
        }
        v += m[i] * a1 - eta[i] * (2.0 * a2 + w);
    }
    double* slot = p.fe_part + (seg + 1) * p.n_chains + chain;
    *slot += -0.5 * (v + (seg == 0 ? p.fe_const : 0.0));
}

// ------------------------------------------------------------------------------------------
// phase 2: scans over segment boundaries, one lane per (chain, role).
//   role 0 (prefix): filtered belief at x[1] (prior ⊗ observation), then
//        f(b_{s+1}) = element_s applied to f(b_s):  W = (V⁻¹ + J)⁻¹,
//        m' = Π W (V⁻¹ m + η) + b,  V' = Π W Π' + C
//   role 1 (suffix): backward message β(b_S) = (0, 0), then
//        β(b_s): W = (C⁻¹ + Λ)⁻¹,  ξ' = η + X' W (ξ − Λ b),  Λ' = JJ − X' W X
template <int D, int DY, bool UNI, bool FE, bool TS = false>   // TS: time-invariant per-chain models — the recursions stop at their fixed points (below)
__global__ void __launch_bounds__(64) k_boundary_scan(Params p, const CstArgFor<UNI, CstLayout<D, DY>::SIZE> cb) {
    static_assert(!TS || !UNI, "shared-model batches scan vectors only (k_boundary_scan_tab)");
    using CL = CstLayout<D, DY>;
    using AL = AggLayout<D>;
    constexpr int NS = Dim<D>::NS;
    const long long chain = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (chain >= p.n_chains) return;
    const int role = blockIdx.y;
    const int mdl = model_at<UNI>(p, model_of<UNI>(p, chain), 0);  // role 0 takes the step at t = 0; the scans below run all-observed, time-invariant segments
    const CPtr c{UNI ? cb.v : p.cst + (long long)mdl * CL::SIZE};
    const double* aggm = p.agg + (long long)mdl * 2 * AL::SIZE;
    const int S = p.S;
    bool ok = true;
    if (role == 0) {
        double mp[D], m[D], yv[DY];
        Sym<D> Vp, V;
#pragma unroll
        for (int i = 0; i < D; ++i) mp[i] = c[CL::M1 + i];
#pragma unroll
        for (int i = 0; i < NS; ++i) Vp.v[i] = c[CL::V1 + i];
        load_y<DY>(p.y, 0, p.n_chains, chain, yv);
        double quad = 0.0, detprod = 1.0;
        obs_update<D, DY, FE, !UNI>(c, mp, Vp, yv, m, V, ok, quad, detprod, !UNI && p.masked && obs_missing<DY>(yv));
        if (UNI) store_filt_sh<D>(p, 0, chain, m, V);
        else store_filt<D>(p.filt, 0, p.n_chains, chain, m, V);
        if (FE) p.fe_part[chain] = -0.5 * (quad + log(detprod));
        if (p.T == 1 || p.filter) {  // single observation: the filtered belief is the posterior
            double* om = p.mean + chain * D;
            double* oc = p.cov + chain * D * D;
#pragma unroll
            for (int i = 0; i < D; ++i) om[i] = m[i];
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) oc[i * D + j] = V(i, j);
        }
        // per-chain models: the element matrices go through registers — the model's table (interior segments always have the full length L:
        // fetched ONCE, in front of the recursion — re-read per segment it was an L2 round trip on every step of a chain of dependent steps)
        // or, on the masked / per-step schedules, this (chain, segment)'s own.  The element vectors travel one segment ahead.
        double al[(!UNI) ? AL::SIZE : 1];
        if constexpr (!UNI) {
            if (!p.elemx) {
#pragma unroll
                for (int q = 0; q < AL::SIZE; ++q) al[q] = aggm[q];
            }
        }
        double eln[2 * D];
        {
            const double* el0 = p.elem + chain;
#pragma unroll
            for (int i = 0; i < 2 * D; ++i) eln[i] = S > 1 ? el0[i * p.n_chains] : 0.0;
        }
        // Time-invariant per-chain models (!UNI, no masks, no per-step constants): the interior segments' element MATRICES are identical, so the boundary
        // covariance V(b_s) is a Riccati recursion over s that reaches its fixed point after a few segments — from there on a segment costs the two
        // matrix–vector products of the mean, m′ = M1 m + M2 η_s + b_s with M2 = Π W, M1 = M2 V⁻¹, instead of two 4×4 inverses and three products
        // (the same test as in k_seg_elements; every lane of the wavefront)
        constexpr bool tinv_scan = TS;   // (an instantiation of its own: masked sweeps keep the plain kernel and its registers)
        double cf1 = 0.0, cf2 = 0.0, M1[TS ? D : 1][TS ? D : 1], M2[TS ? D : 1][TS ? D : 1];
        int nsame = 0;
        bool frozen = false;
        for (int s = 0; s < S; ++s) {
            store_soa<D>(p.fstart, s, p.n_chains, chain, m, V);
            if (s == S - 1) break;
            if constexpr (!UNI) {
                if (p.elemx && !frozen) load_elemx<D>(p, s, chain, false, al);
            }
            const CPtr a{UNI ? aggm : al};
            double elc[2 * D];
#pragma unroll
            for (int i = 0; i < 2 * D; ++i) elc[i] = eln[i];
            if (s + 2 < S) {   // (segment S − 1's element is not used by this role)
                const double* eq = p.elem + ((long long)(s + 1) * 2 * D) * p.n_chains + chain;
#pragma unroll
                for (int i = 0; i < 2 * D; ++i) eln[i] = eq[i * p.n_chains];
            }
            if constexpr (TS) {
                if (frozen) {   // (wave-uniform)
                    double mn[D];
#pragma unroll
                    for (int i = 0; i < D; ++i) {
                        double sacc = elc[i];
#pragma unroll
                        for (int k = 0; k < D; ++k) sacc += M1[i][k] * m[k] + M2[i][k] * elc[D + k];
                        mn[i] = sacc;
                    }
#pragma unroll
                    for (int i = 0; i < D; ++i) m[i] = mn[i];
                    continue;
                }
            }
            Sym<D> Vi, W, T1;
            double det;
            ok = spd_inv<D>(V, Vi, det) && ok;
#pragma unroll
            for (int i = 0; i < NS; ++i) T1.v[i] = Vi.v[i] + a[AL::J + i];
            ok = spd_inv<D>(T1, W, det) && ok;
            double u[D], w[D];
            symv<D>(Vi, m, u);
#pragma unroll
            for (int i = 0; i < D; ++i) u[i] += elc[D + i];
            symv<D>(W, u, w);
            // PW = Π W
            double PW[D][D];
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    double sacc = 0.0;
#pragma unroll
                    for (int k = 0; k < D; ++k) sacc += a[AL::PI + i * D + k] * W(k, j);
                    PW[i][j] = sacc;
                }
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double sacc = elc[i];
#pragma unroll
                for (int k = 0; k < D; ++k) sacc += a[AL::PI + i * D + k] * w[k];
                m[i] = sacc;
            }
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) {
                    double sacc = a[AL::C + sidx(i, j)];
#pragma unroll
                    for (int k = 0; k < D; ++k) sacc += PW[i][k] * a[AL::PI + j * D + k];
                    V(i, j) = sacc;
                }
            if constexpr (TS) {
                if (tinv_scan) {
                    double f1, f2;
                    fixpoint_functionals<D>(V, f1, f2);
                    const bool same = fabs(f1 - cf1) <= 4.5e-16 * fabs(f1) && fabs(f2 - cf2) <= 4.5e-16 * fabs(f2);
                    cf1 = f1;
                    cf2 = f2;
                    nsame = same ? nsame + 1 : 0;
                    if (__all(nsame >= 2)) {   // M2 = Π W, M1 = Π W V⁻¹ of the step just taken (V in = V out to 2 ulp)
                        frozen = true;
#pragma unroll
                        for (int i = 0; i < D; ++i)
#pragma unroll
                            for (int j = 0; j < D; ++j) {
                                M2[i][j] = PW[i][j];
                                double sacc = 0.0;
#pragma unroll
                                for (int k = 0; k < D; ++k) sacc += PW[i][k] * Vi(k, j);
                                M1[i][j] = sacc;
                            }
                    }
                }
            }
        }
    } else {
        double xi[D];
        Sym<D> Lm;
#pragma unroll
        for (int i = 0; i < D; ++i) xi[i] = 0.0;
#pragma unroll
        for (int i = 0; i < NS; ++i) Lm.v[i] = 0.0;
        store_soa<D>(p.beta, S, p.n_chains, chain, xi, Lm);
        double al[(!UNI) ? AL::SIZE : 1];
        double eln[2 * D];
        if (S > 1) {
            const double* eq = p.elem + ((long long)(S - 1) * 2 * D) * p.n_chains + chain;
#pragma unroll
            for (int i = 0; i < 2 * D; ++i) eln[i] = eq[i * p.n_chains];
        }
        // (the same shortcut as in the prefix role: Λ(b_s) reaches its fixed point a few interior segments behind the last one; then
        //  ξ′ = η_s + N1 ξ − N2 b_s with N1 = X′W, N2 = N1 Λ)
        constexpr bool tinv_scan = TS;   // (an instantiation of its own: masked sweeps keep the plain kernel and its registers)
        double cf1 = 0.0, cf2 = 0.0, N1[TS ? D : 1][TS ? D : 1], N2[TS ? D : 1][TS ? D : 1];
        int nsame = 0;
        bool frozen = false;
        for (int s = S - 1; s >= 1; --s) {
            const double* am = aggm + ((s == S - 1) ? AL::SIZE : 0);
            if constexpr (!UNI) {
                if (frozen) {
                } else if (p.elemx) ok = load_elemx<D>(p, s, chain, true, al) && ok;
                else if (s >= S - 2) {   // the last segment's table, then the interior one: two fetches for the whole recursion
#pragma unroll
                    for (int q = 0; q < AL::SIZE; ++q) al[q] = am[q];
                }
            }
            const CPtr a{UNI ? am : al};
            double b[D], eta[D];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                b[i] = eln[i];
                eta[i] = eln[D + i];
            }
            if (s > 1) {
                const double* eq = p.elem + ((long long)(s - 1) * 2 * D) * p.n_chains + chain;
#pragma unroll
                for (int i = 0; i < 2 * D; ++i) eln[i] = eq[i * p.n_chains];
            }
            if constexpr (TS) {
                if (frozen) {   // (wave-uniform)
                    double xn[D];
#pragma unroll
                    for (int i = 0; i < D; ++i) {
                        double sacc = eta[i];
#pragma unroll
                        for (int k = 0; k < D; ++k) sacc += N1[i][k] * xi[k] - N2[i][k] * b[k];
                        xn[i] = sacc;
                    }
#pragma unroll
                    for (int i = 0; i < D; ++i) xi[i] = xn[i];
                    store_soa<D>(p.beta, s, p.n_chains, chain, xi, Lm);
                    continue;
                }
            }
            Sym<D> T1, W;
            double det;
#pragma unroll
            for (int i = 0; i < NS; ++i) T1.v[i] = a[AL::CI + i] + Lm.v[i];
            ok = spd_inv<D>(T1, W, det) && ok;
            double v[D], w[D];
            symv<D>(Lm, b, v);
#pragma unroll
            for (int i = 0; i < D; ++i) v[i] = xi[i] - v[i];
            symv<D>(W, v, w);
            // WX = W X
            double WX[D][D];
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    double sacc = 0.0;
#pragma unroll
                    for (int k = 0; k < D; ++k) sacc += W(i, k) * a[AL::X + k * D + j];
                    WX[i][j] = sacc;
                }
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double sacc = eta[i];
#pragma unroll
                for (int k = 0; k < D; ++k) sacc += a[AL::X + k * D + i] * w[k];
                xi[i] = sacc;
            }
            Sym<D> Lin = Lm;   // Λ the step started from (N2 below)
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) {
                    double sacc = a[AL::JJ + sidx(i, j)];
#pragma unroll
                    for (int k = 0; k < D; ++k) sacc -= a[AL::X + k * D + i] * WX[k][j];
                    Lm(i, j) = sacc;
                }
            store_soa<D>(p.beta, s, p.n_chains, chain, xi, Lm);
            if constexpr (TS) {
                if (tinv_scan && s < S - 1) {   // (interior elements only: the last segment has its own)
                    double f1, f2;
                    fixpoint_functionals<D>(Lm, f1, f2);
                    const bool same = fabs(f1 - cf1) <= 4.5e-16 * fabs(f1) && fabs(f2 - cf2) <= 4.5e-16 * fabs(f2);
                    cf1 = f1;
                    cf2 = f2;
                    nsame = same ? nsame + 1 : 0;
                    if (__all(nsame >= 2)) {
                        frozen = true;
#pragma unroll
                        for (int i = 0; i < D; ++i)
#pragma unroll
                            for (int j = 0; j < D; ++j) {
                                N1[i][j] = WX[j][i];   // (X′W)[i][j] = (W X)[j][i], W symmetric
                            }
#pragma unroll
                        for (int i = 0; i < D; ++i)
#pragma unroll
                            for (int j = 0; j < D; ++j) {
                                double sacc = 0.0;
#pragma unroll
                                for (int k = 0; k < D; ++k) sacc += N1[i][k] * Lin(k, j);
                                N2[i][j] = sacc;
                            }
                    }
                }
            }
        }
    }
    if (!ok) atomicOr(p.status, ST_NOT_POSDEF);
}

// phase 2 for shared-model batches: only the data-dependent vectors are scanned on the device
// (two 4×4 matvecs per segment and role); matrices come from the per-model ScanLayout table, staged through
// LDS in chunks of 64 segments (the recursion is latency-bound: a scalar or global load per step would
// cost more than the arithmetic), and the per-chain element vectors are prefetched one segment ahead.
constexpr int SCAN_TAB_CHUNK = 64;  // segments per LDS chunk
// one wavefront = 64 chains in one role; `tbl` (SCAN_TAB_CHUNK · ScanLayout::SIZE doubles) is private to it: the staging is ordered by
// wave-level fences (LDS executes a wave's accesses in order)
template <int D, int DY, bool FE, int CH = SCAN_TAB_CHUNK>
__device__ __forceinline__ void boundary_scan_tab_body(const Params& p, const CstArg<CstLayout<D, DY>::SIZE>& cb, const long long chain_raw, const int role,
                                                       const int lane, double2* __restrict__ tbl) {
    using CL = CstLayout<D, DY>;
    using SL = ScanLayout<D>;
    constexpr int NS = Dim<D>::NS;
    const bool live = chain_raw < p.n_chains;
    const long long chain = live ? chain_raw : 0;
    const CPtr c{cb.v};
    const int S = p.S;
    bool ok = true;
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto stage = [&](int s0, int n) {  // segments [s0, s0+n) -> LDS
        wave_sync();
        const double2* src = reinterpret_cast<const double2*>(p.scan + (long long)s0 * SL::SIZE);
        for (int q = lane; q < n * SL::SIZE / 2; q += 64) tbl[q] = src[q];
        wave_sync();
    };
    auto load_el = [&](int s, double (&b)[D], double (&eta)[D]) {
        const double* el = p.elem + ((long long)s * 2 * D) * p.n_chains + chain;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            b[i] = el[i * p.n_chains];
            eta[i] = el[(D + i) * p.n_chains];
        }
    };
    if (role == 0) {
        double mp[D], m[D], yv[DY];
        Sym<D> Vp, V;
#pragma unroll
        for (int i = 0; i < D; ++i) mp[i] = c[CL::M1 + i];
#pragma unroll
        for (int i = 0; i < NS; ++i) Vp.v[i] = c[CL::V1 + i];
        load_y<DY>(p.y, 0, p.n_chains, chain, yv);
        double quad = 0.0, detprod = 1.0;
        obs_update<D, DY, FE>(c, mp, Vp, yv, m, V, ok, quad, detprod);
        if (live) {
            store_filt_sh<D>(p, 0, chain, m, V);
            if (FE) p.fe_part[chain] = -0.5 * (quad + log(detprod));
            if (p.T == 1 || p.filter) {
                double* om = p.mean + chain * D;
                double* oc = p.cov + chain * D * D;
#pragma unroll
                for (int i = 0; i < D; ++i) om[i] = m[i];
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) oc[i * D + j] = V(i, j);
            }
        }
        double bn[D], en[D];
        if (S > 1) load_el(0, bn, en);
        for (int s0 = 0; s0 < S; s0 += CH) {
            const int n = (S - s0 < CH) ? S - s0 : CH;
            stage(s0, n);
            for (int q = 0; q < n; ++q) {
                const int s = s0 + q;
                const double* t = reinterpret_cast<const double*>(tbl) + q * SL::SIZE;
                Sym<D> Vb;
#pragma unroll
                for (int i = 0; i < NS; ++i) Vb.v[i] = t[SL::VB + i];
                if (live) store_soa<D>(p.fstart, s, p.n_chains, chain, m, Vb);
                if (s == S - 1) break;
                double b[D], eta[D];
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    b[i] = bn[i];
                    eta[i] = en[i];
                }
                if (s + 1 < S - 1) load_el(s + 1, bn, en);
                double mn[D];
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    double acc = b[i];
#pragma unroll
                    for (int k = 0; k < D; ++k) acc += t[SL::M1 + i * D + k] * m[k] + t[SL::M2 + i * D + k] * eta[k];
                    mn[i] = acc;
                }
#pragma unroll
                for (int i = 0; i < D; ++i) m[i] = mn[i];
            }
        }
    } else {
        double xi[D];
        Sym<D> Lm;
#pragma unroll
        for (int i = 0; i < D; ++i) xi[i] = 0.0;
#pragma unroll
        for (int i = 0; i < NS; ++i) Lm.v[i] = 0.0;
        if (live) store_soa<D>(p.beta, S, p.n_chains, chain, xi, Lm);
        double bn[D], en[D];
        if (S > 1) load_el(S - 1, bn, en);
        // segments S−1 … 1 from the top; step s needs table[s] (N1, N2) and table[s−1].LB = Λβ(b_s), so
        // consecutive chunks overlap by one entry
        int hi = S;
        while (hi > 1) {
            const int s0 = (hi - CH > 0) ? hi - CH : 0;
            stage(s0, hi - s0);
            for (int s = hi - 1; s >= s0 + 1; --s) {
                const double* t = reinterpret_cast<const double*>(tbl) + (s - s0) * SL::SIZE;
                const double* tp = reinterpret_cast<const double*>(tbl) + (s - 1 - s0) * SL::SIZE;
                double b[D], eta[D];
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    b[i] = bn[i];
                    eta[i] = en[i];
                }
                if (s - 1 >= 1) load_el(s - 1, bn, en);
                double xn[D];
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    double acc = eta[i];
#pragma unroll
                    for (int k = 0; k < D; ++k) acc += t[SL::N1 + i * D + k] * xi[k] - t[SL::N2 + i * D + k] * b[k];
                    xn[i] = acc;
                }
#pragma unroll
                for (int i = 0; i < D; ++i) xi[i] = xn[i];
#pragma unroll
                for (int i = 0; i < NS; ++i) Lm.v[i] = tp[SL::LB + i];
                if (live) store_soa<D>(p.beta, s, p.n_chains, chain, xi, Lm);
            }
            hi = s0 + 1;
        }
    }
    if (!ok && live) atomicOr(p.status, ST_NOT_POSDEF);
}
template <int D, int DY, bool FE>
__global__ void __launch_bounds__(64) k_boundary_scan_tab(Params p, const CstArg<CstLayout<D, DY>::SIZE> cb) {
    __shared__ double2 tbl[SCAN_TAB_CHUNK * ScanLayout<D>::SIZE / 2];
    boundary_scan_tab_body<D, DY, FE>(p, cb, (long long)blockIdx.x * blockDim.x + threadIdx.x, (int)blockIdx.y, (int)threadIdx.x, tbl);
}

// ------------------------------------------------------------------------------------------
// phase 3: forward sweep inside each segment.  Per step (reference rule names):
//   `*`_A(:out)           N(A m, A V A')                       predict_cov / matvec_c
//   MvN_x(:out)           + P
//   MvN_y(:μ), `*`_B(:in) observation message (G y, B'Q⁻¹B)     constants LOBS, G
//   product at x[t]       information-form sum, then mean_cov   obs_update
//   Bethe FE terms        telescoped to log p(y_t | y_<t)       obs_update<FE>
// FILT: filtering run (the streaming driver of src/inference/streaming.jl:349-407 with `@autoupdates` posterior ->
// prior feedback, notebook model `linear_gaussian_ssm_filtering`): the filtered belief IS the marginal of x_t and is
// written to the output arrays instead of the forward-message store.
template <int D>
__device__ __forceinline__ void write_marginal(const Params& p, long long t, long long chain, const double (&m)[D],
                                               const Sym<D>& V);
template <int D>
struct OutTile;
template <int D>
__device__ __forceinline__ void write_marginal_wave(const Params& p, double2* tile, int lane, long long t,
                                                    long long chain0, const double (&m)[D], const Sym<D>& V);
template <int D, int DY, bool UNI, bool FE, bool FILT = false, bool TINV = false>
__device__ __forceinline__ void forward_body(const Params& p, const CstArgFor<UNI, CstLayout<D, DY>::SIZE>& cb, const long long g, const int lane,
                                             double2* __restrict__ tile) {   // tile: the wave's output transpose buffer (FILT), or null
    static_assert(!TINV || (!UNI && !FILT), "the frozen tail belongs to smoothing runs of per-chain, time-invariant models");
    using CL = CstLayout<D, DY>;
    constexpr bool CAN_TILE = FILT && (D % 2 == 0);
    const bool tiled = CAN_TILE && tile != nullptr && (p.n_chains % 64 == 0);
    const long long total = p.n_chains * (long long)p.S;
    const bool live = g < total;
    const long long seg = live ? g / p.n_chains : 0;
    const long long chain = live ? g - seg * p.n_chains : 0;
    const long long len = live ? seg_len(p, seg) : 0;
    const int mdl = model_of<UNI>(p, chain);
    const CPtr c{UNI ? cb.v : p.cst + (long long)mdl * CL::SIZE};
    const long long t0 = seg * p.L + 1;  // zero-based time index of the segment's first step

    double m[D];
    Sym<D> V;
    if (live) load_soa<D>(p.fstart, seg, p.n_chains, chain, m, V);
    else {
#pragma unroll
        for (int i = 0; i < D; ++i) m[i] = 0.0;
#pragma unroll
        for (int i = 0; i < Dim<D>::NS; ++i) V.v[i] = (sidx(0, 0) == i || sidx(D - 1, D - 1) == i) ? 1.0 : 0.0;
    }
    bool ok = true;
    double acc = 0.0;
    LogProd lp;
    // The ≈60 constants of a step do not fit the scalar register file (measured: 70 SGPRs spilled to VGPR lanes and 50
    // v_readlane restores per step, 14 % of the issue slots of this VALU-bound kernel).  The observation-side constants are
    // therefore loaded once through an address the compiler cannot prove uniform, which keeps them in VECTOR registers —
    // free here: the batch occupies two wavefronts per SIMD whatever the register count.
    ObsCst<D, DY> oc;
    if constexpr (UNI) {
        int z;
        asm volatile("v_mov_b32 %0, 0" : "=v"(z));
        oc.load(p.cst + z);
    } else
        oc.load(c.p);
    // per-chain models: the transition constants live in registers for the whole segment (read through the global pointer
    // the compiler must assume the record stores alias them and re-fetches all 26 every step)
    double Ar[UNI ? 1 : D * D], Pr[UNI ? 1 : Dim<D>::NS];
    auto load_AP = [&](const double* ct) {
        if constexpr (!UNI) {
#pragma unroll
            for (int k = 0; k < D * D; ++k) Ar[k] = ct[CL::A + k];
#pragma unroll
            for (int k = 0; k < Dim<D>::NS; ++k) Pr[k] = ct[CL::P + k];
        }
    };
    load_AP(c.p);
    double yv[DY], yn[DY];
    if (len > 0) load_y<DY>(p.y, t0, p.n_chains, chain, yn);
    double cf1 = 0.0, cf2 = 0.0;      // TINV: the fixed-point test of V_f (as in k_seg_elements)
    int nsame = 0;
    long long i_frozen = len;         // first step of the frozen tail
    for (long long i = 0; i < len; ++i) {
#pragma unroll
        for (int k = 0; k < DY; ++k) yv[k] = yn[k];
        if (i + 1 < len) load_y<DY>(p.y, t0 + i + 1, p.n_chains, chain, yn);
        double mp[D], T[D][D];
        Sym<D> Vp;
        if constexpr (!UNI) {
            if (p.step_model) {  // A_t, P_t and the observation constants of this time index
                const double* ct = p.cst + (long long)p.step_model[t0 + i] * CL::SIZE;
                oc.load(ct);
                load_AP(ct);
            }
        }
        const CPtr Ac{UNI ? c.p + CL::A : Ar}, Pc{UNI ? c.p + CL::P : Pr};
        matvec_c<D>(Ac, m, mp);
        predict_cov<D>(Ac, Pc, V, T, Vp);
        double quad = 0.0, detprod = 1.0;
        obs_update<D, DY, FE, !UNI>(oc, mp, Vp, yv, m, V, ok, quad, detprod, !UNI && p.masked && obs_missing<DY>(yv));
        if (FE) {
            acc += quad;
            lp.mul(detprod);
        }
        if constexpr (FILT) {
            if constexpr (CAN_TILE) {
                if (tiled) write_marginal_wave<D>(p, tile, lane, t0 + i, chain - lane, m, V);
                else if (live) write_marginal<D>(p, t0 + i, chain, m, V);
            } else if (live)
                write_marginal<D>(p, t0 + i, chain, m, V);
        } else {
            if (UNI) store_filt_sh<D>(p, t0 + i, chain, m, V);
            else store_filt<D>(p.filt, t0 + i, p.n_chains, chain, m, V);
        }
        if constexpr (TINV) {
            // Time-invariant model: V_f stops moving after the filter's mixing time (interior segments start ON the fixed point).  From there on
            // the records carry the mean only (32 instead of 112 B per step at d = 4, each way) and a step costs no inverse: the loop below.
            double f1, f2;
            fixpoint_functionals<D>(V, f1, f2);
            const bool same = fabs(f1 - cf1) <= 4.5e-16 * fabs(f1) && fabs(f2 - cf2) <= 4.5e-16 * fabs(f2);
            cf1 = f1;
            cf2 = f2;
            nsame = same ? nsame + 1 : 0;
            if (__all(nsame >= 2 || !live)) {
                i_frozen = i + 1;
                break;
            }
        }
    }
    if constexpr (TINV) {
        // first time index whose record carries the mean only (the segment's last record never does) — for the backward sweep, in a slot the scan has left
        if (live) p.elem[(seg * 2 * D) * p.n_chains + chain] = (double)(t0 + i_frozen);
        if (i_frozen < len) {   // wave-uniform
            const CPtr Ac{Ar}, Pc{Pr};
            double T[D][D];
            Sym<D> Vp, Lp, Lf;
            double detp, detl;
            predict_cov<D>(Ac, Pc, V, T, Vp);
            ok = spd_inv<D>(Vp, Lp, detp) && ok;
#pragma unroll
            for (int q = 0; q < Dim<D>::NS; ++q) Lf.v[q] = Lp.v[q] + oc.lobs[q];
            ok = spd_inv<D>(Lf, V, detl) && ok;   // V = the fixed point, recomputed from itself
            const double detc = detl * detp;
            for (long long i = i_frozen; i < len; ++i) {
#pragma unroll
                for (int k = 0; k < DY; ++k) yv[k] = yn[k];
                if (i + 1 < len) load_y<DY>(p.y, t0 + i + 1, p.n_chains, chain, yn);
                double mp[D], xp[D], xf[D];
                matvec_c<D>(Ac, m, mp);
                symv<D>(Lp, mp, xp);
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    double sacc = xp[a];
#pragma unroll
                    for (int k = 0; k < DY; ++k) sacc += oc.g[a * DY + k] * yv[k];
                    xf[a] = sacc;
                }
                symv<D>(V, xf, m);
                if (FE) {
                    double q = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
                    for (int a = 0; a < DY; ++a) {
                        double sacc = 0.0;
#pragma unroll
                        for (int k = 0; k < DY; ++k) sacc += oc.qi[sidx(a, k)] * yv[k];
                        q += sacc * yv[a];
                    }
#pragma unroll
                    for (int a = 0; a < D; ++a) {
                        a1 += xf[a] * m[a];
                        a2 += xp[a] * mp[a];
                    }
                    acc += oc.c0 + q - a1 + a2;
                    lp.mul(detc);
                }
                if (i + 1 < len) store_filt_mean<D>(p.filt, t0 + i, p.n_chains, chain, m, V);
                else store_filt<D>(p.filt, t0 + i, p.n_chains, chain, m, V);   // the segment's last record is the next segment's (and the end boundary's) full one
            }
        }
    }
    if (FE && live) p.fe_part[(seg + 1) * p.n_chains + chain] = -0.5 * (acc + lp.value());
    if (!ok) atomicOr(p.status, ST_NOT_POSDEF);
}
template <int D, int DY, bool UNI, bool FE, bool FILT = false>
__global__ void __launch_bounds__(64) k_forward(Params p, const CstArgFor<UNI, CstLayout<D, DY>::SIZE> cb) {
    constexpr bool CAN_TILE = FILT && (D % 2 == 0);
    __shared__ double2 tile[CAN_TILE ? 64 * (((D + D * D) / 2) | 1) : 1];
    forward_body<D, DY, UNI, FE, FILT>(p, cb, (long long)blockIdx.x * blockDim.x + threadIdx.x, (int)threadIdx.x, CAN_TILE ? tile : nullptr);
}
// smoothing runs of per-chain, time-invariant models (whole wavefronts of one segment: n_chains a multiple of 64): mean-only records behind the
// fixed point of V_f
template <int D, int DY, bool FE>
__global__ void __launch_bounds__(64) k_forward_tinv(Params p) {   // (capped at 256 registers it spills 900 bytes and takes 5.9 instead of 1.6 ms)
    forward_body<D, DY, false, FE, false, true>(p, CstArg<1>{}, (long long)blockIdx.x * blockDim.x + threadIdx.x, (int)threadIdx.x, nullptr);
}

// ------------------------------------------------------------------------------------------
// phase 4: backward sweep + marginals inside each segment, Rauch–Tung–Striebel form of the
// reference's backward schedule (MvN_x(:μ) -> `*`_A(:in) -> 3-way product -> mean_cov):
//   Vp = A V_f A' + P,  G = V_f A' Vp⁻¹,
//   m_s(t) = m_f + G (m_s(t+1) − A m_f),  V_s(t) = V_f + G (V_s(t+1) − Vp) G'
// The smoothed belief at the segment's end is (filtered ⊗ β) with β from phase 2.
template <int D>
__device__ __forceinline__ void write_marginal(const Params& p, long long t, long long chain, const double (&m)[D],
                                               const Sym<D>& V) {
    double* om = p.mean + (t * p.n_chains + chain) * D;
    double* oc = p.cov + (t * p.n_chains + chain) * D * D;
    if constexpr (D % 2 == 0) {
        double2* om2 = reinterpret_cast<double2*>(om);
#pragma unroll
        for (int i = 0; i < D / 2; ++i) om2[i] = make_double2(m[2 * i], m[2 * i + 1]);
        double2* oc2 = reinterpret_cast<double2*>(oc);
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D / 2; ++j) oc2[i * (D / 2) + j] = make_double2(V(i, 2 * j), V(i, 2 * j + 1));
    } else {
#pragma unroll
        for (int i = 0; i < D; ++i) om[i] = m[i];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) oc[i * D + j] = V(i, j);
    }
}

// Posterior stores, coalesced.  A lane owns one chain, so written directly each 16-byte store
// of a wave lands in 64 different 128-byte lines (lane stride d²·8 B): the stores become
// request-rate bound (measured: SQ_WAIT_INST_ANY = 74 % of k_backward's wave cycles).  Instead
// the wave transposes its 64 × (d + d²) doubles through LDS so that every global_store_dwordx4
// writes one contiguous 1 KiB run of the [T][chain][d] / [T][chain][d][d] arrays.
// Row stride is an odd number of 16-byte chunks: the per-lane row writes (ds_write_b128) are conflict-free; the transposed
// reads walk rows of 2 (mean) / 8 (covariance) chunks at that stride and collide two-way on part of the banks (PMC:
// SQ_LDS_BANK_CONFLICT ≈ 1.2·10⁸ cycles per C2 launch, profiles/r01/rocprof_summary_v2.txt) — hidden: the kernel is bound by
// the HBM writes these reads feed.
// One wave per workgroup and LDS executes a wave's accesses in order, so no s_barrier is needed.
template <int D>
struct OutTile {
    static constexpr int NOUT = D + D * D;           // doubles per chain
    static constexpr int CH = NOUT / 2;              // 16-byte chunks per chain
    static constexpr int STRIDE = CH | 1;            // odd
    static constexpr int CH_MEAN = D / 2, CH_COV = D * D / 2;
};
template <int D>
__device__ __forceinline__ void write_marginal_wave(const Params& p, double2* tile, int lane, long long t,
                                                    long long chain0, const double (&m)[D], const Sym<D>& V) {
    using OT = OutTile<D>;
    double2* row = tile + lane * OT::STRIDE;
#pragma unroll
    for (int i = 0; i < D / 2; ++i) row[i] = make_double2(m[2 * i], m[2 * i + 1]);
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D / 2; ++j) row[OT::CH_MEAN + i * (D / 2) + j] = make_double2(V(i, 2 * j), V(i, 2 * j + 1));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double2* om = reinterpret_cast<double2*>(p.mean + (t * p.n_chains + chain0) * D);
    double2* oc = reinterpret_cast<double2*>(p.cov + (t * p.n_chains + chain0) * D * D);
#pragma unroll
    for (int k = 0; k < OT::CH_MEAN; ++k) {
        const int q = k * 64 + lane;
        om[q] = tile[(q / OT::CH_MEAN) * OT::STRIDE + (q % OT::CH_MEAN)];
    }
#pragma unroll
    for (int k = 0; k < OT::CH_COV; ++k) {
        const int q = k * 64 + lane;
        oc[q] = tile[(q / OT::CH_COV) * OT::STRIDE + OT::CH_MEAN + (q % OT::CH_COV)];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int D, int DY, bool UNI, bool FUSED = false, bool NOISE = false, bool TINV = false>
__device__ __forceinline__ void backward_body(const Params& p, const CstArgFor<UNI, CstLayout<D, DY>::SIZE>& cb, const long long g, const int lane,
                                              double2* __restrict__ tile) {   // tile: the wave's output transpose buffer, or null
    static_assert(UNI || !FUSED, "the one-pass schedule exists for shared-model batches only");
    static_assert(!NOISE || !UNI, "residual moments are accumulated on the per-chain-model sweep");
    static_assert(!TINV || !UNI, "mean-only records: per-chain, time-invariant models (k_forward_tinv)");
    using CL = CstLayout<D, DY>;
    constexpr int NS = Dim<D>::NS;
    constexpr int NP2 = Dim<D>::NP2;
    const long long total = p.n_chains * (long long)p.S;
    // coalesced-store path: every wave holds 64 consecutive chains of ONE segment
    constexpr bool CAN_TILE = (D % 2 == 0);
    const bool tiled = CAN_TILE && tile != nullptr && (p.n_chains % 64 == 0);
    if (g >= total) return;
    const long long seg = g / p.n_chains;
    const long long chain = g - seg * p.n_chains;
    const long long len = seg_len(p, seg);
    const int mdl = model_of<UNI>(p, chain);
    const CPtr c{UNI ? cb.v : p.cst + (long long)mdl * CL::SIZE};
    const long long tb = seg * p.L;   // zero-based time index of boundary b_seg
    const long long te = tb + len;    // zero-based time index of boundary b_{seg+1}
    bool ok = true;

    // one-pass runs (k_forward0): the mean records hold z_t, the filtered mean is z_t + N_t m_seg
    constexpr bool fused = FUSED;
    double mseg[D];
    if (fused) {
        const double* q = p.fstart + (seg * Dim<D>::NP) * p.n_chains + chain;
#pragma unroll
        for (int i = 0; i < D; ++i) mseg[i] = q[i * p.n_chains];
    }
    double Nn[FUSED ? D * D : 1];  // N_t of the NEXT step to process, loaded one step ahead (wave-uniform address)
    auto load_N = [&](long long t) {
        if constexpr (FUSED) {
            const double* N = p.ntab + t * TimeTab<D>::MT;
#pragma unroll
            for (int k = 0; k < D * D; ++k) Nn[k] = N[k];
        }
    };
    auto add_start = [&](double (&mfv)[D]) {
        if constexpr (FUSED) {
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double s = mfv[i];
#pragma unroll
                for (int k = 0; k < D; ++k) s += Nn[i * D + k] * mseg[k];
                mfv[i] = s;
            }
        }
    };
    // NOISE: Σ_t [(y_t − B m_t)(y_t − B m_t)′ + B V_t B′] over the marginals this lane writes (a separate pass over the posteriors was 2 GB
    // of reads per VMP iteration at d = 4 × 1024 chains × T = 10⁴: here they are in registers)
    constexpr int NSY = DY * (DY + 1) / 2;
    double nacc[NOISE ? NSY : 1];
    if constexpr (NOISE) {
#pragma unroll
        for (int k = 0; k < NSY; ++k) nacc[k] = 0.0;
    }
    auto noise_add = [&](long long t, const double (&m)[D], const Sym<D>& V) {
        if constexpr (NOISE) {
            double yv[DY], r[DY], BV[DY][D];
            load_y<DY>(p.y, t, p.n_chains, chain, yv);
            const double* Bm = p.noise_B;   // wave-uniform
#pragma unroll
            for (int a = 0; a < DY; ++a) {
                double sacc = yv[a];
#pragma unroll
                for (int k = 0; k < D; ++k) sacc -= Bm[a * D + k] * m[k];
                r[a] = sacc;
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    double v = 0.0;
#pragma unroll
                    for (int l = 0; l < D; ++l) v += Bm[a * D + l] * V(l, k);
                    BV[a][k] = v;
                }
            }
#pragma unroll
            for (int a = 0; a < DY; ++a)
#pragma unroll
                for (int b = 0; b <= a; ++b) {
                    double v = r[a] * r[b];
#pragma unroll
                    for (int k = 0; k < D; ++k) v += BV[a][k] * Bm[b * D + k];
                    nacc[sidx(a, b)] += v;
                }
        }
    };
    // smoothed belief at the end boundary: filtered(te) ⊗ β(b_{seg+1})
    double ms[D], mf[D];
    Sym<D> Vs, Vf;
    {
        if (UNI) {
            double2 r[DimM<D>::MP2];
            load_filt_m_sh<D>(p, te, chain, r);
            unpack_m_sh<D>(r, mf);
            load_v_sh<D>(p, te, Vf);
            if constexpr (FUSED) {
                if (len > 0) {
                    load_N(te);
                    add_start(mf);
                } else {
#pragma unroll
                    for (int i = 0; i < D; ++i) mf[i] = mseg[i];
                }
            }
        } else {
            double2 r[NP2];
            load_filt_raw<D>(p.filt, te, p.n_chains, chain, r);
            unpack_rec<D>(r, mf, Vf);
        }
        double xb[D];
        Sym<D> Lb, Vi, Ls;
        load_soa<D>(p.beta, seg + 1, p.n_chains, chain, xb, Lb);
        double det;
        ok = spd_inv<D>(Vf, Vi, det) && ok;
        double u[D];
        symv<D>(Vi, mf, u);
#pragma unroll
        for (int i = 0; i < D; ++i) u[i] += xb[i];
#pragma unroll
        for (int i = 0; i < NS; ++i) Ls.v[i] = Vi.v[i] + Lb.v[i];
        ok = spd_inv<D>(Ls, Vs, det) && ok;
        symv<D>(Vs, u, ms);
        if (seg == p.S - 1) {
            if (!(NOISE && p.skip_marginals)) write_marginal<D>(p, te, chain, ms, Vs);
            noise_add(te, ms, Vs);
        }
    }
    double2 rn[UNI ? DimM<D>::MP2 : NP2];
    // TINV: records tc … te − 1 of this segment carry the mean only (k_forward_tinv) — their covariance is the one of record te, already in Vf
    long long tc = te + 1;
    if constexpr (TINV) tc = (long long)p.elem[(seg * 2 * D) * p.n_chains + chain];
    auto prefetch = [&](long long tt) {
        if constexpr (UNI) load_filt_m_sh<D>(p, tt, chain, rn);
        else if constexpr (TINV) {
            if (tt >= tc && tt < te) {   // (wave-uniform)
                const long long nb64 = (p.n_chains + 63) >> 6;
                const double2* base = reinterpret_cast<const double2*>(p.filt) + ((tt * nb64 + (chain >> 6)) * NP2) * 64 + (chain & 63);
#pragma unroll
                for (int k = 0; k < (D + 1) / 2; ++k) rn[k] = base[k * 64];
            } else
                load_filt_raw<D>(p.filt, tt, p.n_chains, chain, rn);
        } else
            load_filt_raw<D>(p.filt, tt, p.n_chains, chain, rn);
    };
    // per-chain models: transition constants in registers for the whole segment (see k_forward)
    double Ar[UNI ? 1 : D * D], Pr[UNI ? 1 : NS];
    auto load_AP = [&](const double* ct) {
        if constexpr (!UNI) {
#pragma unroll
            for (int k = 0; k < D * D; ++k) Ar[k] = ct[CL::A + k];
#pragma unroll
            for (int k = 0; k < NS; ++k) Pr[k] = ct[CL::P + k];
        }
    };
    load_AP(c.p);
    if (len > 0) prefetch(te - 1);
    if (fused && len > 0) load_N(te - 1);
    long long tstart = te - 1;
    if constexpr (TINV) {
        if (tc < te) {   // (wave-uniform) the mean-only records tc … te − 1: V_f is the one of record te, so Vp, its inverse and the smoother gain are constants
            const CPtr Ac{Ar}, Pc{Pr};
            double T[D][D], G[D][D], det;
            Sym<D> Vp, Lp;
            predict_cov<D>(Ac, Pc, Vf, T, Vp);
            ok = spd_inv<D>(Vp, Lp, det) && ok;
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    double sacc = 0.0;
#pragma unroll
                    for (int k = 0; k < D; ++k) sacc += T[k][i] * Lp(k, j);
                    G[i][j] = sacc;
                }
            for (long long t = te - 1; t >= tc; --t) {
                {
                    double f[2 * ((D + 1) / 2)];
#pragma unroll
                    for (int k = 0; k < (D + 1) / 2; ++k) {
                        f[2 * k] = rn[k].x;
                        f[2 * k + 1] = rn[k].y;
                    }
#pragma unroll
                    for (int a = 0; a < D; ++a) mf[a] = f[a];
                }
                prefetch(t - 1);   // (tc − 1 > tb: the full record the loop below starts with)
                double mp[D], dm[D], H[D][D];
                matvec_c<D>(Ac, mf, mp);
#pragma unroll
                for (int i = 0; i < D; ++i) dm[i] = ms[i] - mp[i];
                Sym<D> Dm;
#pragma unroll
                for (int i = 0; i < NS; ++i) Dm.v[i] = Vs.v[i] - Vp.v[i];
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        double sacc = 0.0;
#pragma unroll
                        for (int k = 0; k < D; ++k) sacc += G[i][k] * Dm(k, j);
                        H[i][j] = sacc;
                    }
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    double sacc = mf[i];
#pragma unroll
                    for (int k = 0; k < D; ++k) sacc += G[i][k] * dm[k];
                    ms[i] = sacc;
                }
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j <= i; ++j) {
                        double sacc = Vf(i, j);
#pragma unroll
                        for (int k = 0; k < D; ++k) sacc += H[i][k] * G[j][k];
                        Vs(i, j) = sacc;
                    }
                if (!(NOISE && p.skip_marginals)) {   // (wave-uniform)
                    if constexpr (CAN_TILE) {
                        if (tiled) write_marginal_wave<D>(p, tile, lane, t, chain - lane, ms, Vs);
                        else write_marginal<D>(p, t, chain, ms, Vs);
                    } else
                        write_marginal<D>(p, t, chain, ms, Vs);
                }
                noise_add(t, ms, Vs);
            }
            tstart = tc - 1;
        }
    }
    for (long long t = tstart; t >= tb; --t) {
        if constexpr (UNI) {
            unpack_m_sh<D>(rn, mf);
            load_v_sh<D>(p, t, Vf);
            if constexpr (FUSED) {
                double zf[D];
#pragma unroll
                for (int i = 0; i < D; ++i) zf[i] = mf[i];
                add_start(zf);
                // at the segment's start boundary the filtered mean is the scan's own result
#pragma unroll
                for (int i = 0; i < D; ++i) mf[i] = (t > tb) ? zf[i] : mseg[i];
                load_N(t > 0 ? t - 1 : 0);  // unconditional: behind a branch the compiler copies the 16 values every step and waits
            }
        } else
            unpack_rec<D>(rn, mf, Vf);
        if (t > tb) prefetch(t - 1);
        double mp[D], T[D][D];
        Sym<D> Vp, Lp;
        if constexpr (!UNI) {
            if (p.step_model) load_AP(p.cst + (long long)p.step_model[t + 1] * CL::SIZE);  // the transition into x[t+1]
        }
        const CPtr Ac{UNI ? c.p + CL::A : Ar}, Pc{UNI ? c.p + CL::P : Pr};
        matvec_c<D>(Ac, mf, mp);
        predict_cov<D>(Ac, Pc, Vf, T, Vp);
        double det;
        ok = spd_inv<D>(Vp, Lp, det) && ok;
        // G = T' Lp   (T = A V_f)
        double G[D][D];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) s += T[k][i] * Lp(k, j);
                G[i][j] = s;
            }
        double dm[D];
#pragma unroll
        for (int i = 0; i < D; ++i) dm[i] = ms[i] - mp[i];
        Sym<D> Dm;
#pragma unroll
        for (int i = 0; i < NS; ++i) Dm.v[i] = Vs.v[i] - Vp.v[i];
        double H[D][D];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) s += G[i][k] * Dm(k, j);
                H[i][j] = s;
            }
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double s = mf[i];
#pragma unroll
            for (int k = 0; k < D; ++k) s += G[i][k] * dm[k];
            ms[i] = s;
        }
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                double s = Vf(i, j);
#pragma unroll
                for (int k = 0; k < D; ++k) s += H[i][k] * G[j][k];
                Vs(i, j) = s;
            }
        if (!(NOISE && p.skip_marginals)) {
            if constexpr (CAN_TILE) {
                if (tiled) write_marginal_wave<D>(p, tile, lane, t, chain - lane, ms, Vs);
                else write_marginal<D>(p, t, chain, ms, Vs);
            } else
                write_marginal<D>(p, t, chain, ms, Vs);
        }
        noise_add(t, ms, Vs);
    }
    if constexpr (NOISE) {
#pragma unroll
        for (int k = 0; k < NSY; ++k) p.noise_part[((long long)seg * NSY + k) * p.n_chains + chain] = nacc[k];
    }
    if (!ok) atomicOr(p.status, ST_NOT_POSDEF);
}
template <int D, int DY, bool UNI, bool FUSED = false>
__global__ void __launch_bounds__(64) k_backward(Params p, const CstArgFor<UNI, CstLayout<D, DY>::SIZE> cb) {
    constexpr bool CAN_TILE = (D % 2 == 0);
    __shared__ double2 tile[CAN_TILE ? 64 * OutTile<CAN_TILE ? D : 2>::STRIDE : 1];
    backward_body<D, DY, UNI, FUSED>(p, cb, (long long)blockIdx.x * blockDim.x + threadIdx.x, (int)threadIdx.x, CAN_TILE ? tile : nullptr);
}
// the same sweep behind k_forward_tinv: mean-only records where the filter covariance had reached its fixed point
template <int D, int DY>
__global__ void __launch_bounds__(64) k_backward_tinv(Params p) {
    constexpr bool CAN_TILE = (D % 2 == 0);
    __shared__ double2 tile[CAN_TILE ? 64 * OutTile<CAN_TILE ? D : 2>::STRIDE : 1];
    backward_body<D, DY, false, false, false, true>(p, CstArg<1>{}, (long long)blockIdx.x * blockDim.x + threadIdx.x, (int)threadIdx.x, CAN_TILE ? tile : nullptr);
}
// the same sweep of an engine with an unknown observation-noise precision (per-chain constants): + the residual second moments per (segment, chain)
template <int D, int DY, bool TINV = false>
__global__ void __launch_bounds__(64) k_backward_noise(Params p) {
    constexpr bool CAN_TILE = (D % 2 == 0);
    __shared__ double2 tile[CAN_TILE ? 64 * OutTile<CAN_TILE ? D : 2>::STRIDE : 1];
    backward_body<D, DY, false, false, true, TINV>(p, CstArg<1>{}, (long long)blockIdx.x * blockDim.x + threadIdx.x, (int)threadIdx.x, CAN_TILE ? tile : nullptr);
}

// ------------------------------------------------------------------------------------------
// Bethe free energy reduction: fe_chain[c] = −Σ_s fe_part[s][c] and the batch total Σ_c fe_chain[c],
// both with a fixed summation shape (deterministic run to run: needed for the 1e-8 tolerance).
// Restates the global sum of src/model/plugins/reactivemp_free_energy.jl:101-123
// (`sumreduce`, src/helpers.jl:21); NaN/Inf check mirrors src/score/diagnostics.jl:19-51.
static __global__ void __launch_bounds__(256) k_fe_chain(Params p, double* block_part) {
    // 64 chains per block; the 4 waves each sum a contiguous quarter of the S+1 partials of
    // their chain (independent loads in flight), combined in fixed order.
    __shared__ double sh[4][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const long long ch = (long long)blockIdx.x * 64 + lane;
    const int n = p.S + 1;
    const int per = (n + 3) / 4;
    const int k0 = q * per, k1 = (k0 + per < n) ? k0 + per : n;
    double s = 0.0;
    if (ch < p.n_chains) {
#pragma unroll 8
        for (int k = k0; k < k1; ++k) s += p.fe_part[(long long)k * p.n_chains + ch];
    }
    sh[q][lane] = s;
    __syncthreads();
    if (q == 0) {
        double f = -(((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane]) * p.fe_scale;
        bool bad = false;
        if (ch < p.n_chains) {
            p.fe_chain[ch] = f;
            bad = !is_finite(f);
        } else
            f = 0.0;
        if (bad) atomicOr(p.status, ST_NONFINITE);
        sh[0][lane] = f;
    }
    __syncthreads();
    if (threadIdx.x < 32) sh[0][threadIdx.x] += sh[0][threadIdx.x + 32];
    __syncthreads();
    for (int w = 16; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[0][threadIdx.x] += sh[0][threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_part[blockIdx.x] = sh[0][0];
}
// Few chains (≤ 16): k_fe_chain leaves all but one lane of each wave idle and walks the partials one by one.  Here one
// workgroup sums the p.S + 1 partials of every chain with all 256 threads (fixed stride and tree shape: deterministic) and
// finishes with the total — chain and total reduction in a single launch.
__device__ __forceinline__ void fe_few_body(const Params& p, double* __restrict__ sh) {   // 256 threads, sh[256]
    const int n = p.S + 1;
    double total = 0.0;
    bool bad = false;
    for (long long ch = 0; ch < p.n_chains; ++ch) {
        double s = 0.0;
        for (int k = threadIdx.x; k < n; k += 256) s += p.fe_part[(long long)k * p.n_chains + ch];
        sh[threadIdx.x] = s;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
            __syncthreads();
        }
        const double f = -sh[0] * p.fe_scale;
        __syncthreads();
        if (threadIdx.x == 0) p.fe_chain[ch] = f;
        bad = bad || !is_finite(f);
        total += f;
    }
    if (threadIdx.x == 0) {
        p.fe_total[p.iteration] = total;
        if (bad) atomicOr(p.status, ST_NONFINITE);
    }
}
static __global__ void __launch_bounds__(256) k_fe_few(Params p) {
    __shared__ double sh[256];
    fe_few_body(p, sh);
}

// ------------------------------------------------------------------------------------------
// Small problems — a few chains, a short series: BASELINE config 1 and the reference's own benchmark sizes (benchmarks/…ipynb cells 12 / 24,
// T = 50 … 5000, one chain) — in ONE launch.  The four-phase schedule of such a problem is a latency chain (L steps per phase, S sequential
// boundary steps) that no kernel boundary shortens; five dependent launches cost more than the arithmetic between them (66 µs per sweep at
// T = 1000, of which ≈ 25 µs are the gaps).  One workgroup of four wavefronts runs the phases back to back with a workgroup barrier between
// them; every (chain, segment) pair is one lane, exactly as in the kernels above (same bodies, same arithmetic, bit-identical results):
//   seg_aggregate_body  all lanes   |  boundary_scan_tab_body  wave 0: prefix role, wave 1: suffix role, lanes = chains
//   forward_body        all lanes   |  backward_body           all lanes   |  fe_few_body  all 256 threads
// Needs n_chains · S ≤ 256 and n_chains ≤ 16 (the reduction of k_fe_few; rxhip.hip picks S accordingly), one model, a smoothing run.
// LDS: the phases alias one block.
// The boundary recursion of k_small_sweep in LOG depth.  Both recursions of boundary_scan_tab_body are affine in the data-dependent vectors
// with data-independent matrices (the per-segment table):
//     prefix  m(b_{s+1}) = M1_s m(b_s) + (b_s + M2_s η_s)            suffix  ξβ(b_s) = N1_s ξβ(b_{s+1}) + (η_s − N2_s b_s),  ξβ(b_S) = 0
// so the maps (A, c): x ↦ A x + c compose associatively, (A2, c2)∘(A1, c1) = (A2 A1, A2 c1 + c2), and an inclusive Hillis–Steele scan over the
// S − 1 maps of a chain — lane = (chain, segment), ⌈log₂(S − 1)⌉ rounds of one D×D product and one matrix–vector product per lane, the partner's
// map through LDS — yields every boundary state at once.  S sequential steps of ≈ 0.6 µs (one wavefront, broadcast table reads) were a third of
// the sweep of BASELINE config 1 and what kept its segments long (S = 56, L = 18); with the recursion at ≈ 8 × 0.4 µs the schedule takes
// S ≈ 250, L = 4.  The compositions are NOT the sequential recursion's operations: results agree with it to rounding, not bit for bit.
// LDS: (D² + D) doubles per lane, [component][lane] (conflict-free): 41 KB at d = 4.
template <int D>
struct ScanPar {
    static constexpr int NE = D * D + D;                         // doubles per map
    static constexpr int LDS_DOUBLES = NE * 256 + 16 * D;        // the maps + m(b_0) of up to 16 chains
};
template <int D, int DY, bool FE>
__device__ __forceinline__ void boundary_scan_par_body(const Params& p, const CstArg<CstLayout<D, DY>::SIZE>& cb, const int tid, double* __restrict__ lds) {
    using CL = CstLayout<D, DY>;
    using SL = ScanLayout<D>;
    constexpr int NS = Dim<D>::NS, NE = ScanPar<D>::NE;
    const CPtr c{cb.v};
    const int S = p.S, C = (int)p.n_chains, n = S - 1;   // n maps per chain and direction
    const int seg = tid / C, chain = tid - seg * C;
    const bool live = seg < S;
    double* x0 = lds + NE * 256;   // [chain][D]: m(b_0)
    bool ok = true;
    // the belief after the first observation (t = 0): the prefix recursion starts from its mean
    if (tid < C) {
        double mp[D], m[D], yv[DY];
        Sym<D> Vp, V;
#pragma unroll
        for (int i = 0; i < D; ++i) mp[i] = c[CL::M1 + i];
#pragma unroll
        for (int i = 0; i < NS; ++i) Vp.v[i] = c[CL::V1 + i];
        load_y<DY>(p.y, 0, p.n_chains, tid, yv);
        double quad = 0.0, detprod = 1.0;
        obs_update<D, DY, FE>(c, mp, Vp, yv, m, V, ok, quad, detprod);
        store_filt_sh<D>(p, 0, tid, m, V);
        if (FE) p.fe_part[tid] = -0.5 * (quad + log(detprod));
        if (p.filter) write_marginal<D>(p, 0, tid, m, V);   // a filtering run: the filtered belief is the marginal
#pragma unroll
        for (int i = 0; i < D; ++i) x0[tid * D + i] = m[i];
    }
    const int ndir = p.filter ? 1 : 2;   // a filtering run needs the prefix direction only
    for (int dir = 0; dir < ndir; ++dir) {
        // this lane's map: prefix index seg (segment seg), suffix index seg ↔ segment S − 1 − seg
        const int sidx_seg = dir == 0 ? seg : S - 1 - seg;
        const bool has = live && seg < n;
        double A[D][D], v[D];
        if (has) {
            const double* t = p.scan + (long long)sidx_seg * SL::SIZE;
            const double* el = p.elem + ((long long)sidx_seg * 2 * D) * p.n_chains + chain;
            double b[D], eta[D];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                b[i] = el[i * p.n_chains];
                eta[i] = el[(D + i) * p.n_chains];
            }
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double acc = dir == 0 ? b[i] : eta[i];
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    A[i][k] = t[(dir == 0 ? SL::M1 : SL::N1) + i * D + k];
                    acc += dir == 0 ? t[SL::M2 + i * D + k] * eta[k] : -t[SL::N2 + i * D + k] * b[k];
                }
                v[i] = acc;
            }
        } else {
#pragma unroll
            for (int i = 0; i < D; ++i) {
                v[i] = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) A[i][k] = i == k ? 1.0 : 0.0;
            }
        }
        for (int h = 1; h < n; h <<= 1) {
            double* cur = lds;
            __syncthreads();   // the previous round's reads (and the first round: x0, the previous direction's reads) are done
#pragma unroll
            for (int i = 0; i < D; ++i) {
#pragma unroll
                for (int k = 0; k < D; ++k) cur[(i * D + k) * 256 + tid] = A[i][k];
                cur[(D * D + i) * 256 + tid] = v[i];
            }
            __syncthreads();
            if (has && seg >= h) {   // compose with the map that ends h positions earlier (applied first)
                const int pl = tid - h * C;
                double Ap[D][D], vp[D];
#pragma unroll
                for (int i = 0; i < D; ++i) {
#pragma unroll
                    for (int k = 0; k < D; ++k) Ap[i][k] = cur[(i * D + k) * 256 + pl];
                    vp[i] = cur[(D * D + i) * 256 + pl];
                }
                double An[D][D], vn[D];
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    double acc = v[i];
#pragma unroll
                    for (int k = 0; k < D; ++k) acc += A[i][k] * vp[k];
                    vn[i] = acc;
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        double a2 = 0.0;
#pragma unroll
                        for (int k = 0; k < D; ++k) a2 += A[i][k] * Ap[k][j];
                        An[i][j] = a2;
                    }
                }
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    v[i] = vn[i];
#pragma unroll
                    for (int j = 0; j < D; ++j) A[i][j] = An[i][j];
                }
            }
        }
        __syncthreads();
        if (dir == 0) {
            // m(b_{seg+1}) = A m(b_0) + v;  the covariances at the boundaries come from the table
            if (live) {
                double m[D];
                if (seg == 0) {
#pragma unroll
                    for (int i = 0; i < D; ++i) m[i] = x0[chain * D + i];
                    Sym<D> Vb;
#pragma unroll
                    for (int i = 0; i < NS; ++i) Vb.v[i] = p.scan[SL::VB + i];
                    store_soa<D>(p.fstart, 0, p.n_chains, chain, m, Vb);
                }
                if (has) {
#pragma unroll
                    for (int i = 0; i < D; ++i) {
                        double acc = v[i];
#pragma unroll
                        for (int k = 0; k < D; ++k) acc += A[i][k] * x0[chain * D + k];
                        m[i] = acc;
                    }
                    Sym<D> Vb;
#pragma unroll
                    for (int i = 0; i < NS; ++i) Vb.v[i] = p.scan[(long long)(seg + 1) * SL::SIZE + SL::VB + i];
                    store_soa<D>(p.fstart, seg + 1, p.n_chains, chain, m, Vb);
                }
            }
        } else if (live) {
            // ξβ(b_{S−1−seg}) = v (the recursion starts from ξβ(b_S) = 0);  Λβ(b_s) = table[s − 1].LB
            if (seg == 0) {
                double z[D];
                Sym<D> Z;
#pragma unroll
                for (int i = 0; i < D; ++i) z[i] = 0.0;
#pragma unroll
                for (int i = 0; i < NS; ++i) Z.v[i] = 0.0;
                store_soa<D>(p.beta, S, p.n_chains, chain, z, Z);
            }
            if (has) {
                const int sb = S - 1 - seg;   // ≥ 1
                Sym<D> Lm;
#pragma unroll
                for (int i = 0; i < NS; ++i) Lm.v[i] = p.scan[(long long)(sb - 1) * SL::SIZE + SL::LB + i];
                store_soa<D>(p.beta, sb, p.n_chains, chain, v, Lm);
            }
        }
    }
    if (!ok) atomicOr(p.status, ST_NOT_POSDEF);
}

constexpr int SMALL_SWEEP_THREADS = 256;
constexpr int SMALL_SWEEP_SCAN_CHUNK = 16;
template <int D, int DY>
struct SmallSweepLds {
    static constexpr int AGG = 4 * 2 * AggStage<D, DY>::NPC;                               // double2 per workgroup: one [2][NPC] block per wave
    static constexpr int SCAN = 2 * SMALL_SWEEP_SCAN_CHUNK * ScanLayout<D>::SIZE / 2;       // two roles
    static constexpr int FE = 256 / 2;
    static constexpr int N = AGG > SCAN ? (AGG > FE ? AGG : FE) : (SCAN > FE ? SCAN : FE);
};
template <int D, int DY>
__host__ __device__ constexpr int small_sweep_lds_doubles() {
    return 2 * SmallSweepLds<D, DY>::N > ScanPar<D>::LDS_DOUBLES ? 2 * SmallSweepLds<D, DY>::N : ScanPar<D>::LDS_DOUBLES;
}
// par_scan: the boundary recursion in log depth (boundary_scan_par_body) instead of the sequential one — the host asks for it from 24 segments on.
// FILT: a filtering run (rxhip_run_filter): prefix direction only, the filtered beliefs are the marginals, no backward phase.
template <int D, int DY, bool FE, bool FILT>
__global__ void __launch_bounds__(SMALL_SWEEP_THREADS) k_small_sweep(Params p, const CstArg<CstLayout<D, DY>::SIZE> cb, int par_scan) {
    extern __shared__ __attribute__((aligned(16))) double small_lds[];
    double2* lds = reinterpret_cast<double2*>(small_lds);
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    seg_aggregate_body<D, DY, true>(p, cb, tid, lane, lds + w * 2 * AggStage<D, DY>::NPC);
    __syncthreads();
    if (par_scan) boundary_scan_par_body<D, DY, FE>(p, cb, tid, small_lds);
    else if (w < (FILT ? 1 : 2)) boundary_scan_tab_body<D, DY, FE, SMALL_SWEEP_SCAN_CHUNK>(p, cb, lane, w, lane, lds + w * (SMALL_SWEEP_SCAN_CHUNK * ScanLayout<D>::SIZE / 2));
    __syncthreads();
    forward_body<D, DY, true, FE, FILT>(p, cb, tid, lane, nullptr);
    if constexpr (!FILT) {
        __syncthreads();
        backward_body<D, DY, true, false>(p, cb, tid, lane, nullptr);
    }
    if (FE) {
        __syncthreads();
        fe_few_body(p, small_lds);
    }
}
static __global__ void __launch_bounds__(256) k_fe_total(Params p, const double* block_part, int nblocks) {
    __shared__ double sh[256];
    double local = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) local += block_part[b];
    sh[threadIdx.x] = local;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) p.fe_total[p.iteration] = sh[0];
}


// ------------------------------------------------------------------------------------------
// phase 4 for shared-model batches whose size is a multiple of 64: table-driven backward sweep.
//
// With one model for every chain the smoother gain G_t and the smoothed covariance V_s(t) do not depend on the data either.
// The k_smooth_tab_* kernels build them once per engine (the recursion of k_backward on covariances only) together
// with  E_t = I − G_t A  and  F_t = E_t N_t,  so that the per-chain work of a step is three small matrix–vector products,
//     m_s(t) = E_t z_t + F_t m_seg + G_t m_s(t+1)
// (m_s(t) = m_f + G (m_s(t+1) − A m_f) with m_f = z_t + N_t m_seg), and the posterior covariance is written from the table:
// the 64·d² doubles a wavefront owes for one time index are contiguous in memory and periodic with period d², so every lane
// keeps writing the same one or two table entries and every store instruction covers a contiguous 1 KiB run — no transposition
// through LDS, no inverse, ≈50 instead of ≈600 VALU instructions per step.  The means go through a 2 KiB LDS tile.
template <int D>
struct SmoothTab {  // one row per time index
    static constexpr int E = 0, F = D * D, G = 2 * D * D, VS = 3 * D * D;  // VS: V_s(t), full row-major [D][D]
    static constexpr int SIZE = 4 * D * D;
};
template <int D>
struct SegEndTab {  // per segment: m_s(te) = H1 m_f(te) + H2 ξβ
    static constexpr int H1 = 0, H2 = D * D;  // V_s(te) V_f(te)⁻¹, V_s(te)
    static constexpr int SIZE = 2 * D * D;
};
// The tables are built in four data-parallel stages (the first version walked a whole segment in one lane: 5.6 ms at
// T = 10⁵ — longer than the sweep it serves):
//   steps      one lane per time index: G_t, E_t, F_t and  C_t = V_f − G_t V_p G_t'  — everything that needs V_f(t) only;
//   compose    one lane per block of SMOOTH_LB steps: the map  V_s(u0) = Ĝ V_s(u1) Ĝ' + Ĉ  over the block
//              ((G1, C1)∘(G2, C2) = (G1 G2, C1 + G1 C2 G1'): V ↦ G V G' + C is closed under composition);
//   chain      one lane per segment: V_s at the segment end from the boundary scan, then over the segment's blocks;
//   apply      one lane per block: V_s(t) = C_t + G_t V_s(t+1) G_t' from the block's end value.
constexpr int SMOOTH_LB = 32;
template <int D>
struct SmoothBlk {  // per block of SMOOTH_LB steps
    static constexpr int GA = 0, CA = D * D, VE = 2 * D * D;  // Ĝ, Ĉ, V_s at the block's end (all full row-major)
    static constexpr int SIZE = 3 * D * D;
};
struct SmoothTabParams {
    long long T, L;
    int S;
    const double* vtab;  // [T][NS]
    const double* ntab;  // [T][MT]
    const double* scan;  // [S][ScanLayout::SIZE]  (LB = Λβ(b_{s+1}))
    double* gtab;        // [T][SmoothTab::SIZE]
    double* segend;      // [S][SegEndTab::SIZE]
    double* blk;         // [S][ceil(L / SMOOTH_LB)][SmoothBlk::SIZE]
    int* status;
};
__host__ __device__ inline long long smooth_blocks_per_segment(long long L) { return (L + SMOOTH_LB - 1) / SMOOTH_LB; }

template <int D>
__device__ __forceinline__ void congruence(const double (&G)[D][D], const double (&V)[D][D], const double* C, double (&out)[D][D]) {
    double H[D][D];  // out = C + G V G'  (V, C, out symmetric)
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) acc += G[a][k] * V[k][b];
            H[a][b] = acc;
        }
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) {
            double acc = C[a * D + b];
#pragma unroll
            for (int k = 0; k < D; ++k) acc += H[a][k] * G[b][k];
            out[a][b] = acc;
            out[b][a] = acc;
        }
}

template <int D, int DY>
__global__ void __launch_bounds__(64) k_smooth_tab_steps(SmoothTabParams q, const CstArg<CstLayout<D, DY>::SIZE> cb) {
    using CL = CstLayout<D, DY>;
    using ST = SmoothTab<D>;
    constexpr int NS = Dim<D>::NS;
    constexpr int MT = TimeTab<D>::MT;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= q.T - 1) return;
    const CPtr c{cb.v};
    Sym<D> Vf, Vp, Lp;
    double det;
#pragma unroll
    for (int k = 0; k < NS; ++k) Vf.v[k] = q.vtab[t * NS + k];
    double AV[D][D], G[D][D], E[D][D];
    predict_cov<D>(CPtr{c.p + CL::A}, CPtr{c.p + CL::P}, Vf, AV, Vp);   // AV = A V_f
    if (!spd_inv<D>(Vp, Lp, det)) atomicOr(q.status, ST_NOT_POSDEF);
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) acc += AV[k][a] * Lp(k, b);
            G[a][b] = acc;
        }
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) {
            double acc = (a == b) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) acc -= G[a][k] * c[CL::A + k * D + b];
            E[a][b] = acc;
        }
    double* row = q.gtab + t * ST::SIZE;
    const double* N = q.ntab + t * MT;
    const bool at_start = t % q.L == 0;  // at the segment's own start boundary the filtered mean is m_seg itself: no N term
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) {
            double f = 0.0;
            if (!at_start) {
#pragma unroll
                for (int k = 0; k < D; ++k) f += E[a][k] * N[k * D + b];
            }
            double cc = Vf(a, b);  // C_t = V_f − G V_p G',  G V_p = (A V_f)'
#pragma unroll
            for (int k = 0; k < D; ++k) cc -= AV[k][a] * G[b][k];
            row[ST::E + a * D + b] = E[a][b];
            row[ST::F + a * D + b] = f;
            row[ST::G + a * D + b] = G[a][b];
            row[ST::VS + a * D + b] = cc;   // replaced by V_s(t) in the apply stage
        }
}

// block (s, j) covers the time indices [u0, u1) of segment s; false when it is empty
__device__ __forceinline__ bool smooth_block_range(const SmoothTabParams& q, long long id, long long& u0, long long& u1) {
    const long long nb = smooth_blocks_per_segment(q.L);
    const long long s = id / nb, j = id - s * nb;
    if (s >= q.S) return false;
    const long long tb = s * q.L;
    long long te = tb + q.L;
    if (te > q.T - 1) te = q.T - 1;
    u0 = tb + j * SMOOTH_LB;
    u1 = u0 + SMOOTH_LB;
    if (u1 > te) u1 = te;
    return u0 < u1;
}

template <int D>
__global__ void __launch_bounds__(64) k_smooth_tab_compose(SmoothTabParams q) {
    using ST = SmoothTab<D>;
    using SB = SmoothBlk<D>;
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long u0, u1;
    if (!smooth_block_range(q, id, u0, u1)) return;
    double Ga[D][D], Ca[D][D];
    {
        const double* row = q.gtab + (u1 - 1) * ST::SIZE;
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
            for (int b = 0; b < D; ++b) {
                Ga[a][b] = row[ST::G + a * D + b];
                Ca[a][b] = row[ST::VS + a * D + b];
            }
    }
    for (long long t = u1 - 2; t >= u0; --t) {  // prepend the map of step t
        const double* row = q.gtab + t * ST::SIZE;
        double G[D][D], Gn[D][D], Cn[D][D];
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
            for (int b = 0; b < D; ++b) G[a][b] = row[ST::G + a * D + b];
        congruence<D>(G, Ca, row + ST::VS, Cn);
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
            for (int b = 0; b < D; ++b) {
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) acc += G[a][k] * Ga[k][b];
                Gn[a][b] = acc;
            }
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
            for (int b = 0; b < D; ++b) {
                Ga[a][b] = Gn[a][b];
                Ca[a][b] = Cn[a][b];
            }
    }
    double* o = q.blk + id * SB::SIZE;
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) {
            o[SB::GA + a * D + b] = Ga[a][b];
            o[SB::CA + a * D + b] = Ca[a][b];
        }
}

template <int D>
__global__ void __launch_bounds__(64) k_smooth_tab_chain(SmoothTabParams q) {
    using SL = ScanLayout<D>;
    using ST = SmoothTab<D>;
    using SB = SmoothBlk<D>;
    constexpr int NS = Dim<D>::NS;
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= q.S) return;
    const long long tb = s * q.L;
    long long te = tb + q.L;
    if (te > q.T - 1) te = q.T - 1;
    bool ok = true;
    Sym<D> Vf, Vi, Ls, Vs;
    double det;
#pragma unroll
    for (int k = 0; k < NS; ++k) Vf.v[k] = q.vtab[te * NS + k];
    ok = spd_inv<D>(Vf, Vi, det) && ok;
#pragma unroll
    for (int k = 0; k < NS; ++k) Ls.v[k] = Vi.v[k] + q.scan[s * SL::SIZE + SL::LB + k];
    ok = spd_inv<D>(Ls, Vs, det) && ok;
    double* se = q.segend + s * SegEndTab<D>::SIZE;
    double V[D][D];
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) acc += Vs(a, k) * Vi(k, b);
            se[SegEndTab<D>::H1 + a * D + b] = acc;
            se[SegEndTab<D>::H2 + a * D + b] = Vs(a, b);
            V[a][b] = Vs(a, b);
        }
    if (s == q.S - 1) {  // the last time index is written by the last segment only
        double* row = q.gtab + te * ST::SIZE;
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
            for (int b = 0; b < D; ++b) {
                row[ST::E + a * D + b] = row[ST::F + a * D + b] = row[ST::G + a * D + b] = 0.0;
                row[ST::VS + a * D + b] = Vs(a, b);
            }
    }
    const long long nb = smooth_blocks_per_segment(q.L);
    const long long used = te > tb ? (te - tb + SMOOTH_LB - 1) / SMOOTH_LB : 0;
    for (long long j = used - 1; j >= 0; --j) {
        double* o = q.blk + (s * nb + j) * SB::SIZE;
        double G[D][D], Vn[D][D];
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
            for (int b = 0; b < D; ++b) {
                o[SB::VE + a * D + b] = V[a][b];
                G[a][b] = o[SB::GA + a * D + b];
            }
        congruence<D>(G, V, o + SB::CA, Vn);
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
            for (int b = 0; b < D; ++b) V[a][b] = Vn[a][b];
    }
    if (!ok) atomicOr(q.status, ST_NOT_POSDEF);
}

template <int D>
__global__ void __launch_bounds__(64) k_smooth_tab_apply(SmoothTabParams q) {
    using ST = SmoothTab<D>;
    using SB = SmoothBlk<D>;
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long u0, u1;
    if (!smooth_block_range(q, id, u0, u1)) return;
    double V[D][D];
    const double* o = q.blk + id * SB::SIZE;
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) V[a][b] = o[SB::VE + a * D + b];
    for (long long t = u1 - 1; t >= u0; --t) {
        double* row = q.gtab + t * ST::SIZE;
        double G[D][D], Vn[D][D];
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
            for (int b = 0; b < D; ++b) G[a][b] = row[ST::G + a * D + b];
        congruence<D>(G, V, row + ST::VS, Vn);
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
            for (int b = 0; b < D; ++b) {
                V[a][b] = Vn[a][b];
                row[ST::VS + a * D + b] = Vn[a][b];
            }
    }
}

template <int D>
__global__ void __launch_bounds__(64) k_backward_sh(Params p, const double* __restrict__ gtab, const double* __restrict__ segend) {
    using ST = SmoothTab<D>;
    constexpr int MP2 = DimM<D>::MP2;
    constexpr int MT = TimeTab<D>::MT;
    constexpr int U = 4;                     // steps per table chunk
    constexpr int RP = ST::SIZE / 2;         // 16-byte pieces per row
    constexpr int NPC = U * RP;
    constexpr int PPL = (NPC + 63) / 64;
    constexpr int NMP = 32 * D;              // 16-byte pieces of the 64 means of a time index
    constexpr int NCP = 32 * D * D;          // … of the 64 covariances
    __shared__ double2 tbuf[2][NPC];
    __shared__ double mtile[64 * D];
    const int lane = threadIdx.x;
    const long long g0 = (long long)blockIdx.x * 64;  // n_chains % 64 == 0: the wave holds 64 chains of ONE segment
    const long long seg = g0 / p.n_chains;
    const long long chain0 = g0 - seg * p.n_chains;
    const long long chain = chain0 + lane;
    const long long len = seg_len(p, seg);
    const long long tb = seg * p.L, te = tb + len;

    double mseg[D], ms[D];
    {
        const double* q = p.fstart + (seg * Dim<D>::NP) * p.n_chains + chain;
        const double* bq = p.beta + ((seg + 1) * Dim<D>::NP) * p.n_chains + chain;
        double mf[D], xb[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            mseg[i] = q[i * p.n_chains];
            xb[i] = bq[i * p.n_chains];
        }
        if (len > 0) {
            double2 r[MP2];
            load_filt_m_sh<D>(p, te, chain, r);
            unpack_m_sh<D>(r, mf);
            const double* N = p.ntab + te * MT;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double s = mf[i];
#pragma unroll
                for (int k = 0; k < D; ++k) s += N[i * D + k] * mseg[k];
                mf[i] = s;
            }
        } else {
#pragma unroll
            for (int i = 0; i < D; ++i) mf[i] = mseg[i];
        }
        const double* se = segend + seg * SegEndTab<D>::SIZE;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) s += se[SegEndTab<D>::H1 + i * D + k] * mf[k] + se[SegEndTab<D>::H2 + i * D + k] * xb[k];
            ms[i] = s;
        }
    }
    // one time index of output: means through the LDS tile, covariances straight from table row `vs`
    auto write_out = [&](long long t, const double* vs) {
#pragma unroll
        for (int i = 0; i < D; ++i) mtile[lane * D + i] = ms[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double2* om = reinterpret_cast<double2*>(p.mean + (t * p.n_chains + chain0) * D);
        double2* oc = reinterpret_cast<double2*>(p.cov + (t * p.n_chains + chain0) * D * D);
#pragma unroll
        for (int k = 0; k < (NMP + 63) / 64; ++k) {
            const int q = k * 64 + lane;
            if (q < NMP) stream_store(om + q, mtile[2 * q], mtile[2 * q + 1]);
        }
#pragma unroll
        for (int k = 0; k < (NCP + 63) / 64; ++k) {
            const int q = k * 64 + lane;
            if (q < NCP) stream_store(oc + q, vs[(2 * q) % (D * D)], vs[(2 * q + 1) % (D * D)]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    if (seg == p.S - 1) write_out(te, gtab + te * ST::SIZE + ST::VS);

    // table rows te−1, te−2, … stream through LDS one chunk ahead (the gains are the same for every lane of the wave)
    const double2* g2 = reinterpret_cast<const double2*>(gtab);
    double2 tr[PPL];
    auto fetch = [&](long long i0) {
#pragma unroll
        for (int k = 0; k < PPL; ++k) {
            const int idx = k * 64 + lane;
            const long long r = i0 + idx / RP;  // step index
            tr[k] = (idx < NPC && r < len) ? g2[(te - 1 - r) * RP + idx % RP] : make_double2(0.0, 0.0);
        }
    };
    auto stash = [&](int b) {
#pragma unroll
        for (int k = 0; k < PPL; ++k) {
            const int idx = k * 64 + lane;
            if (idx < NPC) tbuf[b][idx] = tr[k];
        }
    };
    fetch(0);
    stash(0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double2 rn[MP2];
    if (len > 0) load_filt_m_sh<D>(p, te - 1, chain, rn);
    int b = 0;
    for (long long i0 = 0; i0 < len; i0 += U, b ^= 1) {  // `len` is uniform across the wave
        if (i0 + U < len) fetch(i0 + U);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long i = i0 + u;
            if (i < len) {
                const long long t = te - 1 - i;
                const double* row = reinterpret_cast<const double*>(&tbuf[b][0]) + u * ST::SIZE;
                double z[D];
                unpack_m_sh<D>(rn, z);
                if (t > tb) load_filt_m_sh<D>(p, t - 1, chain, rn);
                else {
#pragma unroll
                    for (int k = 0; k < D; ++k) z[k] = mseg[k];  // the start boundary: the scan's own filtered mean (F row is zero)
                }
                double mn[D];
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    double s0 = 0.0, s1 = 0.0;
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        s0 += row[ST::E + a * D + k] * z[k] + row[ST::F + a * D + k] * mseg[k];
                        s1 += row[ST::G + a * D + k] * ms[k];
                    }
                    mn[a] = s0 + s1;
                }
#pragma unroll
                for (int a = 0; a < D; ++a) ms[a] = mn[a];
                write_out(t, row + ST::VS);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (i0 + U < len) stash(b ^ 1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace rxhip
