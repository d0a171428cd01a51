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
static __global__ void __launch_bounds__(256) kd_split_tables(SplitParams q) {
    const int D = q.D, NT = D / 16, tid = threadIdx.x;
    const long long t = blockIdx.x;
    const double* rec = q.p.filt + t * q.rec + 3 * D;   // chain 0: C_t | G_t′, both in accumulator order
    double* tab = q.dtab + (size_t)t * SplitTab::size(D);
    for (int e = tid; e < D * D; e += blockDim.x) {
        // accumulator order: index ((w·NT + tile)·4 + r)·64 + lane  <->  (row 16w + (lane >> 4) + 4r, col 16·tile + (lane & 15))
        const int lane = e & 63, r = (e >> 6) & 3, wt = e >> 8, w = wt / NT, tile = wt - w * NT;
        const int row = 16 * w + (lane >> 4) + 4 * r, col = 16 * tile + (lane & 15);
        // C_t is symmetric and the record holds the owned tiles only (dense_owned_tile): the others are the mirror images
        double c;
        if (dense_owned_tile(NT, w, tile)) c = rec[e];
        else {
            const int j = lane & 15, rp = j >> 2, lp = (j & 3) * 16 + (lane >> 4) + 4 * r;   // element (col, row) in tile (tile, w)
            c = rec[((tile * NT + w) * 4 + rp) * 64 + lp];
        }
        const double a = rec[D * D + e];
        tab[row * D + col] = c;
        tab[D * D + col * D + row] = a;       // AT[k = col][i = row] = G′[row][col]
        tab[2 * D * D + row * D + col] = a;   // A[k = row][i = col]  = G′[row][col]
    }
}

// after the model pass: the posterior covariances and the constant free-energy slots of user chain 0
static __global__ void __launch_bounds__(256) kd_split_save(SplitParams q, long long user_chains) {
    const long long dd = (long long)q.p.d_out * q.p.d_out, total = q.p.T * dd;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long t = e / dd, k = e - t * dd;
        q.vstab[e] = q.p.cov[(t * user_chains) * dd + k];
    }
    if (blockIdx.x == 0)
        for (int s = threadIdx.x; s < 2 * q.p.S; s += blockDim.x) q.fe_const[s] = q.p.fe_part[(long long)s * user_chains];
}

// every sweep: V_s(t) into the posterior array of every chain, the constant free-energy slots into every chain's column
static __global__ void __launch_bounds__(256) kd_split_broadcast(SplitParams q, long long user_chains, int want_fe, int want_cov) {
    const long long dd = (long long)q.p.d_out * q.p.d_out, row = user_chains * dd, total = want_cov ? q.p.T * row : 0;
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

// lane k of this lane's 16-lane row, without the LDS crossbar: v_mov_b32_dpp row_newbcast:k per half (a ds_bpermute pair costs
// ≈100 cycles of latency and two LDS-pipe issues per element; the data pass is exactly this chain, D times per product)
template <int K>
__device__ __forceinline__ double row_bcast(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x150 + K, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0x150 + K, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// Σ_k Mt[k·D + i] x_k with Mt in global memory (segment prologues: once per segment), sixteen rows of the table at a time
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

// ---- the table rows of a step go through LDS ----------------------------------------------------------------------------------
// Reading the table columns when they are needed made every step a chain of L2 round trips (26 µs per step at D = 64; at D ≤ 32
// a register prefetch of the next step's columns hid one trip, 1.9 µs per step were left).  A workgroup of four wavefronts
// serves 4·(64 / D) chains of ONE segment: the 2·D² (forward) or D² (backward) table doubles of the next step are fetched by
// all 256 threads (coalesced, in flight under the current step) and parked in the other half of a double buffer; the products
// read LDS rows (consecutive lanes, consecutive addresses) and get x_k by DPP row broadcasts (D = 16, 32) or v_readlane.  One
// barrier per step.
__device__ __forceinline__ double wave_bcast(double v, int k) {  // lane k of the wavefront (k uniform)
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), k), hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
    return __hiloint2double(hi, lo);
}
template <int K0, int D>
struct LdsRowDot {  // s[k & 3] += Mt[(KOFF + k)·D + i] · (lane k of this lane's 16-lane row), k = K0 … 15
    template <int KOFF>
    static __device__ __forceinline__ void run(const double* Mt, int i, double x, double (&s)[4]) {
        s[K0 & 3] += Mt[(KOFF + K0) * D + i] * row_bcast<K0>(x);
        LdsRowDot<K0 + 1, D>::template run<KOFF>(Mt, i, x, s);
    }
};
template <int D>
struct LdsRowDot<16, D> {
    template <int KOFF>
    static __device__ __forceinline__ void run(const double*, int, double, double (&)[4]) {}
};
// Σ_k Mt[k·D + i] x_k with Mt in LDS; x_k lives in lane k of the unit (a 16- or 32-lane group, or the whole wavefront)
template <int D>
__device__ __forceinline__ double lds_matvec(const double* Mt, int i, double x) {
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    if constexpr (D == 16) {
        LdsRowDot<0, D>::template run<0>(Mt, i, x, s);
    } else if constexpr (D == 32) {
        const double xo = __shfl_xor(x, 16);          // the unit's other row
        const bool hi = (threadIdx.x & 16) != 0;
        const double x0 = hi ? xo : x, x1 = hi ? x : xo;
        LdsRowDot<0, D>::template run<0>(Mt, i, x0, s);
        LdsRowDot<0, D>::template run<16>(Mt, i, x1, s);
    } else {
#pragma unroll 16
        for (int k = 0; k < D; ++k) s[k & 3] += Mt[k * D + i] * wave_bcast(x, k);
    }
    return (s[0] + s[1]) + (s[2] + s[3]);
}
// (chain, component) of this lane in a workgroup of four wavefronts that serves 4·(64 / D) chains of ONE segment
template <int D>
struct LdsUnit {
    long long chain;
    int i, base;
    bool live;
    __device__ __forceinline__ LdsUnit(const DenseParams& p) {
        constexpr int gpw = 64 / D;
        const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, g = lane / D;
        i = lane - g * D;
        base = g * D;
        chain = ((long long)blockIdx.x * 4 + w) * gpw + (g < gpw ? g : 0);
        live = g < gpw && chain < p.n_chains;
        if (g >= gpw) i = D - 1;                          // D = 48: lanes 48 … 63 idle
        if (chain >= p.n_chains) chain = p.n_chains - 1;  // idle units follow the last chain (uniform control flow, no stores)
    }
};
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
    const int tid = threadIdx.x;
    const LdsUnit<D> u(p);
    const int i = u.i;
    const long long seg = blockIdx.y, chain = u.chain;
    const bool live = u.live;
    const long long b0 = 1 + seg * p.L;
    long long b1 = b0 + p.L;
    if (b1 > p.T) b1 = p.T;
    const long long len = b1 - b0, t0 = seg * p.L + 1;
    double* filt = p.filt + chain * p.T * q.rec;
    double xi = split_matvec<D>(p.bnd + ((size_t)seg * 2 + 0) * DD, i, u.base, p.fstart_m[(chain * p.S + seg) * D + i]);
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
    const int tid = threadIdx.x;
    const LdsUnit<D> u(p);
    const int i = u.i;
    const long long seg = blockIdx.y, chain = u.chain;
    const bool live = u.live;
    const long long b0 = 1 + seg * p.L;
    long long b1 = b0 + p.L;
    if (b1 > p.T) b1 = p.T;
    const long long len = b1 - b0, tb = seg * p.L, te = tb + len;
    const double* filt = p.filt + chain * p.T * q.rec;
    const double xf = filt[te * q.rec + i] + p.beta_xi[(chain * (p.S + 1) + seg + 1) * D + i];
    const bool last = seg == p.S - 1;
    double ms = split_matvec<D>(last ? q.vlast : p.bnd + ((size_t)seg * 2 + 1) * DD, i, u.base, xf);
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
