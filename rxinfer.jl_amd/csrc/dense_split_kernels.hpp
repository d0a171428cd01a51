// dense_split_kernels.hpp — batches that SHARE one model on the MFMA path (4 < d ≤ 64): model pass once, data pass per sweep.
//
// In the information-form smoother of dense_kernels.hpp everything that is a matrix — C_t = (Λ_f(t) + A′P⁻¹A)⁻¹, the smoother
// gain G_t = C_t (P⁻¹A)′, the posterior covariance V_s(t), the log-determinants of the free energy — depends on the model and
// the time index only.  A batch of chains with one model used to recompute all of it in every chain's workgroup (one d×d
// inverse and four d×d×d contractions per step and chain).  Here kd_forward_info / kd_backward_info run ONCE per engine on a
// single chain (the model pass: their records are turned into plain row-major tables by kd_split_tables), and a sweep is
//     ξ_f(t) = B′Q⁻¹y_t + G′_{t−1} ξ_f(t−1),      c_t = C_t ξ_f(t)                    (kd_split_forward)
//     m_s(t) = c_t + G_t m_s(t+1)                                                      (kd_split_backward)
// per chain — three matrix–vector products per step against tables every chain of the batch reads from L2 — plus a broadcast
// of V_s(t) into the posterior arrays and the residual quadratic forms of the free energy (kd_fe_resid, unchanged).  The
// segment boundaries (start belief, backward message at the segment end) come from the same aggregation and scan kernels as
// before: they were vector-only already.  Same idea as lgssm_kernels.hpp's one-pass schedule for d ≤ 4.
//
// One wavefront serves 64 / D (chain, segment) units: lane i of a unit owns component i of every vector; a matrix–vector
// product reads the matrix TRANSPOSED row by row (consecutive lanes, consecutive addresses; the units of a wavefront belong to
// the same segment and read the same rows) and gets x[k] by a cross-lane read.
#pragma once
#include "dense_kernels.hpp"

namespace rxhip {

struct SplitTab {  // per time index: three D×D row-major matrices
    // CT: C_t (symmetric);  AT: (G_t′)′ = G_t, i.e. [k][i] = G′[i][k] — the operand of  out_i = Σ_k G′[i][k] x_k;
    // A: G_t′ itself, [k][i] = G′[k][i] — the operand of  out_i = Σ_k G[i][k] x_k
    __host__ __device__ static size_t size(int D) { return 3 * (size_t)D * D; }
};
struct SplitParams {
    DenseParams p;         // as the sweep's (n_chains = workgroup chains: a chain, or a packed pair)
    int D;                 // kernel dimension 16·NT
    int rec;               // doubles per record (DenseCfg<NT>::REC)
    double* dtab;          // [T][3][D][D]
    double* vlast;         // [D][D]  V_s(T−1) of the model pass (row-major)
    double* vstab;         // [T][d_out][d_out]  posterior covariance of a (user) chain
    double* fe_const;      // [2S]  data-independent free-energy slots of one user chain
};

// records of workgroup chain 0 (accumulator order) -> row-major tables; one workgroup per time index
__global__ void __launch_bounds__(256) kd_split_tables(SplitParams q) {
    const int D = q.D, NT = D / 16, tid = threadIdx.x;
    const long long t = blockIdx.x;
    const double* rec = q.p.filt + t * q.rec + 3 * D;   // chain 0: C_t | G_t′, both in accumulator order
    double* tab = q.dtab + (size_t)t * SplitTab::size(D);
    for (int e = tid; e < D * D; e += blockDim.x) {
        // accumulator order: index ((w·NT + tile)·4 + r)·64 + lane  <->  (row 16w + (lane >> 4) + 4r, col 16·tile + (lane & 15))
        const int lane = e & 63, r = (e >> 6) & 3, wt = e >> 8, w = wt / NT, tile = wt - w * NT;
        const int row = 16 * w + (lane >> 4) + 4 * r, col = 16 * tile + (lane & 15);
        const double c = rec[e], a = rec[D * D + e];
        tab[row * D + col] = c;
        tab[D * D + col * D + row] = a;       // AT[k = col][i = row] = G′[row][col]
        tab[2 * D * D + row * D + col] = a;   // A[k = row][i = col]  = G′[row][col]
    }
}

// after the model pass: the posterior covariances and the constant free-energy slots of user chain 0
__global__ void __launch_bounds__(256) kd_split_save(SplitParams q, long long user_chains) {
    const long long dd = (long long)q.p.d_out * q.p.d_out, total = q.p.T * dd;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long t = e / dd, k = e - t * dd;
        q.vstab[e] = q.p.cov[(t * user_chains) * dd + k];
    }
    if (blockIdx.x == 0)
        for (int s = threadIdx.x; s < 2 * q.p.S; s += blockDim.x) q.fe_const[s] = q.p.fe_part[(long long)s * user_chains];
}

// every sweep: V_s(t) into the posterior array of every chain, the constant free-energy slots into every chain's column
__global__ void __launch_bounds__(256) kd_split_broadcast(SplitParams q, long long user_chains, int want_fe) {
    const long long dd = (long long)q.p.d_out * q.p.d_out, row = user_chains * dd, total = q.p.T * row;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long t = e / row, k = (e - t * row) % dd;
        q.p.cov[e] = q.vstab[t * dd + k];
    }
    if (want_fe) {
        const long long n = 2LL * q.p.S * user_chains;
        for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x)
            q.p.fe_part[e] = q.fe_const[e / user_chains];
    }
}

struct SplitUnit {
    long long seg, chain;
    int i, base;
    bool live;
};
template <int D>
__device__ __forceinline__ SplitUnit split_unit(const SplitParams& q) {
    constexpr int gpw = 64 / D;
    const int lane = threadIdx.x;
    SplitUnit u;
    const int g = lane / D;
    u.i = lane - g * D;
    u.base = g * D;
    const long long unit = (long long)blockIdx.x * gpw + (g < gpw ? g : 0);
    u.seg = unit / q.p.n_chains;        // neighbouring units are neighbouring chains of one segment: they read the same table rows
    u.chain = unit - u.seg * q.p.n_chains;
    u.live = g < gpw && u.seg < q.p.S;
    if (!u.live) { u.seg = 0; u.chain = 0; }  // idle lanes follow unit 0 (uniform control flow, no stores)
    return u;
}

// column i of a D×D row-major table matrix (the lane's operand of out_i = Σ_k Mt[k][i] x_k): D independent loads in flight
template <int D>
__device__ __forceinline__ void split_load_col(double (&c)[D], const double* __restrict__ Mt, int i) {
#pragma unroll
    for (int k = 0; k < D; ++k) c[k] = Mt[k * D + i];
}
// lane k of this lane's 16-lane row, without the LDS crossbar: v_mov_b32_dpp row_newbcast:k per half (a ds_bpermute pair costs
// ≈100 cycles of latency and two LDS-pipe issues per element; the data pass is exactly this chain, D times per product)
template <int K>
__device__ __forceinline__ double row_bcast(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x150 + K, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0x150 + K, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int K0, int D>
struct RowDot {  // s[k & 3] += c[KOFF + k] · (lane k of the row), k = K0 … 15
    template <int KOFF>
    static __device__ __forceinline__ void run(const double (&c)[D], double x, double (&s)[4]) {
        s[K0 & 3] += c[KOFF + K0] * row_bcast<K0>(x);
        RowDot<K0 + 1, D>::template run<KOFF>(c, x, s);
    }
};
template <int D>
struct RowDot<16, D> {
    template <int KOFF>
    static __device__ __forceinline__ void run(const double (&)[D], double, double (&)[4]) {}
};
// Σ_k c[k] · x_k for a unit of D = 16 or 32 lanes (x_k lives in lane base + k; four partial sums)
template <int D>
__device__ __forceinline__ double split_dot(const double (&c)[D], int base, double x) {
    static_assert(D == 16 || D == 32, "one or two DPP rows");
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    if (D == 16) {
        RowDot<0, D>::template run<0>(c, x, s);
    } else {
        const double xo = __shfl_xor(x, 16);          // the unit's other row
        const bool hi = (threadIdx.x & 16) != 0;       // this lane sits in the unit's second row: its own row holds x_16 … x_31
        const double x0 = hi ? xo : x, x1 = hi ? x : xo;
        RowDot<0, D>::template run<0>(c, x0, s);
        RowDot<0, D>::template run<(D == 32 ? 16 : 0)>(c, x1, s);
    }
    (void)base;
    return (s[0] + s[1]) + (s[2] + s[3]);
}
// without registers for a whole column (D = 48, 64): sixteen rows of the table at a time
template <int D>
__device__ __forceinline__ double split_matvec(const double* __restrict__ Mt, int i, int base, double x) {
    double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k0 = 0; k0 < D; k0 += 16) {
        double c[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) c[k] = Mt[(k0 + k) * D + i];
#pragma unroll
        for (int k = 0; k < 16; ++k) s[k & 3] += c[k] * __shfl(x, base + k0 + k);
    }
    return (s[0] + s[1]) + (s[2] + s[3]);
}

// PF: the table columns of the NEXT step travel in registers under the current step's cross-lane reads (D ≤ 32: 4·D doubles)
template <int D>
__global__ void __launch_bounds__(64) kd_split_forward(SplitParams q) {
    constexpr bool PF = D <= 32;
    const DenseParams& p = q.p;
    const SplitUnit u = split_unit<D>(q);
    const int i = u.i;
    constexpr size_t DD = (size_t)D * D, TS = 3 * DD;
    const long long b0 = 1 + u.seg * p.L;
    long long b1 = b0 + p.L;
    if (b1 > p.T) b1 = p.T;
    const long long len = b1 - b0, t0 = u.seg * p.L + 1;
    double* filt = p.filt + u.chain * p.T * q.rec;
    // belief at the segment start in information form: ξ_f = Λ_f(b_s) m(b_s)
    double xi = split_matvec<D>(p.bnd + ((size_t)u.seg * 2 + 0) * DD, i, u.base, p.fstart_m[(u.chain * p.S + u.seg) * D + i]);
    double gyn = len > 0 ? filt[t0 * q.rec + D + i] : 0.0;
    if constexpr (PF) {
        // two column sets in registers, used alternately: the step that works on one set loads the other for the next step
        double c0[D], a0[D], c1[D], a1[D];
        if (len > 0) {
            split_load_col<D>(c0, q.dtab + (size_t)(t0 - 1) * TS, i);
            split_load_col<D>(a0, q.dtab + (size_t)(t0 - 1) * TS + DD, i);
        }
        auto step = [&](long long s, const double (&cc)[D], const double (&ca)[D], double (&nc)[D], double (&na)[D]) {
            const long long t = t0 + s;
            double* rec = filt + (t - 1) * q.rec;
            const double gyc = gyn;
            const long long tn = s + 1 < len ? t + 1 : t;   // unconditional prefetch (the last one re-reads this step's rows)
            gyn = filt[tn * q.rec + D + i];
            split_load_col<D>(nc, q.dtab + (size_t)(tn - 1) * TS, i);
            split_load_col<D>(na, q.dtab + (size_t)(tn - 1) * TS + DD, i);
            const double cxi = split_dot<D>(cc, u.base, xi);                // C_{t−1} ξ_f(t−1)
            const double axi = split_dot<D>(ca, u.base, xi);                // G′_{t−1} ξ_f(t−1)
            if (u.live) {
                rec[i] = xi;
                rec[2 * D + i] = cxi;
            }
            xi = gyc + axi;                                                 // ξ_f(t)
        };
        long long s = 0;
        for (; s + 1 < len; s += 2) {
            step(s, c0, a0, c1, a1);
            step(s + 1, c1, a1, c0, a0);
        }
        if (s < len) step(s, c0, a0, c1, a1);
    } else {
        for (long long s = 0; s < len; ++s) {
            const long long t = t0 + s;
            double* rec = filt + (t - 1) * q.rec;
            const double gyc = gyn;
            gyn = filt[(s + 1 < len ? t + 1 : t) * q.rec + D + i];
            const double* tab = q.dtab + (size_t)(t - 1) * TS;
            const double cxi = split_matvec<D>(tab, i, u.base, xi);
            const double axi = split_matvec<D>(tab + DD, i, u.base, xi);
            if (u.live) {
                rec[i] = xi;
                rec[2 * D + i] = cxi;
            }
            xi = gyc + axi;
        }
    }
    if (u.live && u.seg == p.S - 1) filt[(t0 + len - 1) * q.rec + i] = xi;  // ξ_f(T−1): no successor writes it
}

template <int D>
__global__ void __launch_bounds__(64) kd_split_backward(SplitParams q) {
    constexpr bool PF = D <= 32;
    const DenseParams& p = q.p;
    const SplitUnit u = split_unit<D>(q);
    const int i = u.i;
    constexpr size_t DD = (size_t)D * D, TS = 3 * DD;
    const long long b0 = 1 + u.seg * p.L;
    long long b1 = b0 + p.L;
    if (b1 > p.T) b1 = p.T;
    const long long len = b1 - b0, tb = u.seg * p.L, te = tb + len;
    const double* filt = p.filt + u.chain * p.T * q.rec;
    // smoothed mean at the end boundary: m_s = V_s (ξ_f + ξβ); V_s of an inner boundary from the table, of the last index from
    // the model pass
    const double xf = filt[te * q.rec + i] + p.beta_xi[(u.chain * (p.S + 1) + u.seg + 1) * D + i];
    const bool last = u.seg == p.S - 1;
    double ms = split_matvec<D>(last ? q.vlast : p.bnd + ((size_t)u.seg * 2 + 1) * DD, i, u.base, xf);
    if (u.live && last) dense_store_mean(p, te, u.chain, i, ms);
    double cxn = len > 0 ? filt[(te - 1) * q.rec + 2 * D + i] : 0.0;
    if constexpr (PF) {
        double g0[D], g1[D];
        if (len > 0) split_load_col<D>(g0, q.dtab + (size_t)(te - 1) * TS + 2 * DD, i);
        auto step = [&](long long t, const double (&cg)[D], double (&ng)[D]) {
            const double cx = cxn;
            const long long tn = t - 1 >= tb ? t - 1 : tb;
            cxn = filt[tn * q.rec + 2 * D + i];
            split_load_col<D>(ng, q.dtab + (size_t)tn * TS + 2 * DD, i);
            ms = cx + split_dot<D>(cg, u.base, ms);                         // C_t ξ_f(t) + G_t m_s(t+1)
            if (u.live) dense_store_mean(p, t, u.chain, i, ms);
        };
        long long t = te - 1;
        for (; t - 1 >= tb; t -= 2) {
            step(t, g0, g1);
            step(t - 1, g1, g0);
        }
        if (t >= tb) step(t, g0, g1);
    } else {
        for (long long t = te - 1; t >= tb; --t) {
            const double cx = cxn;
            cxn = filt[(t - 1 >= tb ? t - 1 : tb) * q.rec + 2 * D + i];
            ms = cx + split_matvec<D>(q.dtab + (size_t)t * TS + 2 * DD, i, u.base, ms);
            if (u.live) dense_store_mean(p, t, u.chain, i, ms);
        }
    }
}

// ---- D = 48, 64: the table rows of a step go through LDS ------------------------------------------------------------------
// At these sizes a lane cannot hold the columns of the next step in registers, and reading them when they are needed made every
// step a chain of L2 round trips (26 µs per step at D = 64).  Here a workgroup of four wavefronts serves four chains of ONE
// segment: the 2·D² (forward) or D² (backward) table doubles of the next step are fetched by all 256 threads (coalesced, in
// flight under the current step) and parked in the other half of a double buffer; the products read LDS rows (consecutive
// lanes, consecutive addresses) and get x_k by v_readlane.  One barrier per step.
__device__ __forceinline__ double wave_bcast(double v, int k) {  // lane k of the wavefront (k uniform)
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), k), hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
    return __hiloint2double(hi, lo);
}
template <int D>
__device__ __forceinline__ double lds_matvec(const double* Mt, int i, double x) {  // Σ_k Mt[k·D + i] x_k, Mt in LDS
    double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 16
    for (int k = 0; k < D; ++k) s[k & 3] += Mt[k * D + i] * wave_bcast(x, k);
    return (s[0] + s[1]) + (s[2] + s[3]);
}
template <int N>   // N doubles per thread of a contiguous block of 256·N doubles (a wavefront instruction covers 512 contiguous bytes)
struct StageRegs {
    double v[N];
    __device__ __forceinline__ void load(const double* __restrict__ src, int tid) {
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = src[j * 256 + tid];
    }
    __device__ __forceinline__ void store(double* dst, int tid) const {
#pragma unroll
        for (int j = 0; j < N; ++j) dst[j * 256 + tid] = v[j];
    }
};
__host__ __device__ inline size_t split_lds_bytes(int D, int matrices) { return sizeof(double) * 2 * (size_t)matrices * D * D; }

template <int D>
__global__ void __launch_bounds__(256) kd_split_forward_lds(SplitParams q) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr size_t DD = (size_t)D * D, TS = 3 * DD;
    constexpr int NPT = (int)(2 * DD / 256);   // doubles per thread and step: C_t | G_t (contiguous in the table row)
    static_assert((2 * DD) % 256 == 0, "whole doubles over 256 threads");
    const DenseParams& p = q.p;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, i = lane < D ? lane : D - 1;
    const long long seg = blockIdx.y;
    long long chain = (long long)blockIdx.x * 4 + w;
    const bool live = chain < p.n_chains && lane < D;
    if (chain >= p.n_chains) chain = p.n_chains - 1;   // idle wavefronts follow the last chain (uniform control flow, no stores)
    const long long b0 = 1 + seg * p.L;
    long long b1 = b0 + p.L;
    if (b1 > p.T) b1 = p.T;
    const long long len = b1 - b0, t0 = seg * p.L + 1;
    double* filt = p.filt + chain * p.T * q.rec;
    double xi = split_matvec<D>(p.bnd + ((size_t)seg * 2 + 0) * DD, i, 0, p.fstart_m[(chain * p.S + seg) * D + i]);
    double gyn = len > 0 ? filt[t0 * q.rec + D + i] : 0.0;
    StageRegs<NPT> st;
    if (len > 0) {
        st.load(q.dtab + (size_t)(t0 - 1) * TS, tid);
        st.store(smem, tid);
    }
    __syncthreads();
    for (long long s = 0; s < len; ++s) {
        const long long t = t0 + s;
        const double* buf = smem + (s & 1) * 2 * DD;
        const long long tn = s + 1 < len ? t + 1 : t;            // unconditional prefetch (the last one re-reads this step's rows)
        st.load(q.dtab + (size_t)(tn - 1) * TS, tid);
        double* rec = filt + (t - 1) * q.rec;
        const double gyc = gyn;
        gyn = filt[tn * q.rec + D + i];
        const double cxi = lds_matvec<D>(buf, i, xi);             // C_{t−1} ξ_f(t−1)
        const double axi = lds_matvec<D>(buf + DD, i, xi);        // G′_{t−1} ξ_f(t−1)
        if (live) {
            rec[i] = xi;
            rec[2 * D + i] = cxi;
        }
        xi = gyc + axi;                                           // ξ_f(t)
        st.store(smem + ((s + 1) & 1) * 2 * DD, tid);             // nobody reads that half before the barrier below
        __syncthreads();
    }
    if (live && seg == p.S - 1) filt[(t0 + len - 1) * q.rec + i] = xi;
}

template <int D>
__global__ void __launch_bounds__(256) kd_split_backward_lds(SplitParams q) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr size_t DD = (size_t)D * D, TS = 3 * DD;
    constexpr int NPT = (int)(DD / 256);
    static_assert(DD % 256 == 0, "whole doubles over 256 threads");
    const DenseParams& p = q.p;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, i = lane < D ? lane : D - 1;
    const long long seg = blockIdx.y;
    long long chain = (long long)blockIdx.x * 4 + w;
    const bool live = chain < p.n_chains && lane < D;
    if (chain >= p.n_chains) chain = p.n_chains - 1;
    const long long b0 = 1 + seg * p.L;
    long long b1 = b0 + p.L;
    if (b1 > p.T) b1 = p.T;
    const long long len = b1 - b0, tb = seg * p.L, te = tb + len;
    const double* filt = p.filt + chain * p.T * q.rec;
    const double xf = filt[te * q.rec + i] + p.beta_xi[(chain * (p.S + 1) + seg + 1) * D + i];
    const bool last = seg == p.S - 1;
    double ms = split_matvec<D>(last ? q.vlast : p.bnd + ((size_t)seg * 2 + 1) * DD, i, 0, xf);
    if (live && last) dense_store_mean(p, te, chain, i, ms);
    double cxn = len > 0 ? filt[(te - 1) * q.rec + 2 * D + i] : 0.0;
    StageRegs<NPT> st;
    if (len > 0) {
        st.load(q.dtab + (size_t)(te - 1) * TS + 2 * DD, tid);
        st.store(smem, tid);
    }
    __syncthreads();
    long long n = 0;
    for (long long t = te - 1; t >= tb; --t, ++n) {
        const double* buf = smem + (n & 1) * DD;
        const long long tn = t - 1 >= tb ? t - 1 : tb;
        st.load(q.dtab + (size_t)tn * TS + 2 * DD, tid);
        const double cx = cxn;
        cxn = filt[tn * q.rec + 2 * D + i];
        ms = cx + lds_matvec<D>(buf, i, ms);                      // C_t ξ_f(t) + G_t m_s(t+1)
        if (live) dense_store_mean(p, t, chain, i, ms);
        st.store(smem + ((n + 1) & 1) * DD, tid);
        __syncthreads();
    }
}

}  // namespace rxhip
