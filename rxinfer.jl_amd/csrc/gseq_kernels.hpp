// gseq_kernels.hpp — `missing` observations anywhere in the data and per-step constants A[t], P[t], B[t], Q[t] for ANY state
// dimension (d, dy ≤ 64).
//
// Both features make the covariances of a chain depend on the time index in a way no table of the time-parallel MFMA schedule
// survives (dense_kernels.hpp builds its segment boundaries from ONE model and a fully observed chain).  What remains is the
// message schedule of the reference itself (src/inference/batch.jl:391-430 on the chain of test/models/statespace/
// mlgssm_test.jl:9-26): forward messages in time order, backward messages in reverse, one product per variable — sequential
// in t, parallel over chains.  One workgroup per chain, all matrices in LDS, runtime dimensions:
//
//   k_gseq_forward    `*`_A(:out) -> MvN_x(:out) -> product with MvN_y(:μ)∘`*`_B(:in) when y[t] is not `missing`
//                     (covariance form; a missing y[t] sends no message, docs/src/manuals/inference/static.md:98-123);
//                     writes the filtered belief to the output arrays and −log p(y_t | y_<t) terms to the free energy
//   k_gseq_backward   MvN_x(:μ) -> `*`_A(:in) backward messages folded into the marginal (RTS form), in place on the output
//                     arrays:  G = V_f A′ V_p⁻¹,  m_s = m_f + G (m_s⁺ − A m_f),  V_s = V_f − G (A V_f) + (G V_s⁺) G′
//
// Matrix products run on 4×4 register tiles over LDS (leading dimension n | 1: the four rows of a tile fall into different
// banks), the d_y×d_y and d×d inverses are in-place Gauss–Jordan sweeps (two barriers per pivot).  This is the coverage path
// for these model classes at d > 4 — the d, dy ≤ 4 kernels (lgssm_kernels.hpp) stay the fast path for small states.
#pragma once
#ifndef RXHIP_GSEQ_HOST_EMULATION  // tests/emu/gseq_emu.cpp compiles this file as plain C++ (one-thread workgroup) for the CPU tests
#include "generic_kernels.hpp"
#define RXHIP_GSEQ_EXTERN_SHARED(name) extern __shared__ double name[];
#endif

namespace rxhip {

struct GseqParams {
    long long T, n_chains;
    int d, dy, ptt, fe;
    const double* y;         // [T][chain][dy]; a NaN entry: y[t] is missing
    double* mean;            // [T(+H)][chain][d]     filtered beliefs after the forward kernel, marginals after the backward kernel
    double* cov;             // [T(+H)][chain][d][d]
    const double* user;      // [n_models][A | P | B | Q | Q⁻¹]  (generic_kernels.hpp)
    const double* prior;     // [n_models][m0 | V0]
    const int* chain_model;  // model of a chain, or null
    const int* step_model;   // model of a time index (transition INTO x[t], observation of y[t]), or null
    double* fe_part;         // [chain]  log p(y) of the chain (the reduction kernels negate and scale)
    int* status;
    double* cross;           // null, or [T−1][chain][d][d]: Cov(x[t], x[t+1] | y) = G_t V_s(t+1), kept by the backward kernel for the
                             // node-local joints (k_joint_generic)
};
__device__ __forceinline__ size_t gseq_model_index(const GseqParams& p, long long chain, long long t) {
    return p.step_model ? (size_t)p.step_model[t] : p.chain_model ? (size_t)p.chain_model[chain] : 0;
}
__device__ __forceinline__ GenericModel gseq_model(const GseqParams& p, size_t idx) {
    const size_t sz = 2 * (size_t)p.d * p.d + (size_t)p.dy * p.d + 2 * (size_t)p.dy * p.dy;
    const double* u = p.user + idx * sz;
    GenericModel m;
    m.A = u;
    m.P = m.A + (size_t)p.d * p.d;
    m.B = m.P + (size_t)p.d * p.d;
    m.Q = m.B + (size_t)p.dy * p.d;
    m.Qi = m.Q + (size_t)p.dy * p.dy;
    return m;
}

__host__ __device__ inline int gseq_ld(int d, int dy) { return (d > dy ? d : dy) | 1; }
__host__ __device__ inline size_t gseq_lds_bytes(int d, int dy) {
    const size_t n = (size_t)(d > dy ? d : dy), ld = (size_t)gseq_ld(d, dy);
    return sizeof(double) * (4 * n * ld + 8 * n + 8);
}

// out(i, j) = [out(i, j) +] alpha · Σ_k X(i, k) Y(k, j),   i < ni, j < nj, k < nk
// with X(i, k) = X[i·sxi + k·sxk], Y(k, j) = Y[k·syk + j·syj], out(i, j) = out[i·ldo + j]; one 4×4 tile per thread and pass
__device__ __forceinline__ void tile_gemm(double* out, int ldo, int ni, int nj, int nk, const double* X, int sxi, int sxk,
                                          const double* Y, int syk, int syj, double alpha, bool accumulate, int tid, int nthreads) {
    const int ti_n = (ni + 3) >> 2, tj_n = (nj + 3) >> 2;
    for (int tile = tid; tile < ti_n * tj_n; tile += nthreads) {
        const int ti = tile / tj_n, tj = tile - ti * tj_n;
        const int i0 = 4 * ti, j0 = 4 * tj;
        int xi[4], yj[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // rows / columns past the edge repeat the last valid one (never stored)
            xi[u] = (i0 + u < ni ? i0 + u : ni - 1) * sxi;
            yj[u] = (j0 + u < nj ? j0 + u : nj - 1) * syj;
        }
        double acc[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[u][v] = 0.0;
#pragma unroll 2
        for (int k = 0; k < nk; ++k) {
            double xv[4], yv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                xv[u] = X[xi[u] + k * sxk];
                yv[u] = Y[k * syk + yj[u]];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[u][v] += xv[u] * yv[v];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                if (i0 + u < ni && j0 + v < nj) {
                    double* o = out + (i0 + u) * ldo + j0 + v;
                    *o = accumulate ? *o + alpha * acc[u][v] : alpha * acc[u][v];
                }
    }
}

// thread -> (column j0, row group i0) for element-wise passes over an n-column matrix: `lanes` (a power of two ≥ n, at most a
// wavefront) consecutive threads walk one row, so that no index needs a division by the runtime dimension
struct Map2 {
    int lanes, groups, j0, i0;
    __device__ __forceinline__ Map2(int n, int tid, int nthreads) {
        int sh = 0;
        while ((1 << sh) < n && (1 << sh) < 64 && (2 << sh) <= nthreads) ++sh;
        lanes = 1 << sh;
        groups = nthreads >> sh;
        j0 = tid & (lanes - 1);
        i0 = tid >> sh;
    }
};
#define RXHIP_FOR_2D(mp, ni, nj, i, j) for (int i = (mp).i0; i < (ni); i += (mp).groups) for (int j = (mp).j0; j < (nj); j += (mp).lanes)

// M <- ½(M + M′) + ½(C + C′) in place (M: n×n in LDS, leading dimension ld; C: n×n row-major, may be null): every thread owns
// whole (i, j) / (j, i) pairs
__device__ __forceinline__ void sym_add(double* M, int ld, int n, const double* C, int tid, int nthreads) {
    const Map2 mp(n, tid, nthreads);
    RXHIP_FOR_2D(mp, n, n, i, j) {
        if (j > i) continue;
        double v = 0.5 * (M[i * ld + j] + M[j * ld + i]);
        if (C) v += 0.5 * (C[i * n + j] + C[j * n + i]);
        M[i * ld + j] = v;
        M[j * ld + i] = v;
    }
}

// in-place inverse of a symmetric positive definite n×n matrix in LDS (Gauss–Jordan without pivoting: the pivots are those of
// the LDL′ factorisation, their product is the determinant); buf: 2n doubles.  Returns false when a pivot is not positive;
// *logdet (thread 0 only is meaningful) = log|M|.  Ends with a barrier.
__device__ __forceinline__ bool lds_gj_inverse(double* M, int ld, int n, double* buf, double* logdet, int tid, int nthreads) {
    bool ok = true;
    double ld_acc = 0.0;
    double* rowk = buf;
    double* colk = buf + n;
    const Map2 mp(n, tid, nthreads);
    for (int k = 0; k < n; ++k) {
        for (int i = tid; i < n; i += nthreads) {
            rowk[i] = M[k * ld + i];
            colk[i] = M[i * ld + k];
        }
        __syncthreads();
        const double piv = rowk[k];
        ok = ok && piv > 0.0;
        const double r = 1.0 / piv;
        if (logdet && tid == 0) ld_acc += log(piv);
        for (int j = mp.j0; j < n; j += mp.lanes) {
            const double rj = rowk[j] * r;
#pragma unroll 4
            for (int i = mp.i0; i < n; i += mp.groups) {  // independent elements: their LDS reads overlap
                const double ci = colk[i], mij = M[i * ld + j];
                double v = mij - ci * rj;
                v = (j == k) ? -ci * r : v;
                v = (i == k) ? ((j == k) ? r : rj) : v;
                M[i * ld + j] = v;
            }
        }
        __syncthreads();
    }
    if (logdet) *logdet = ld_acc;
    return ok;
}

struct GseqLds {  // the dynamic LDS block of the forward kernels
    double *X0, *X1, *X2, *X3;   // X0: V_f(t−1) -> V_p(t) -> V_f(t);  X1: A V, then B V_p;  X2: products / S and S⁻¹;  X3: A, B, S⁻¹ B V_p
    double *m, *mp, *r, *sr;     // filtered mean, predicted mean, innovation, S⁻¹ r
    double *buf, *yv;            // 2n: pivot row / column;  n: y[t] (known observation offsets already taken out)
    int ld;
};
__device__ __forceinline__ GseqLds gseq_lds(double* sm, int d, int dy) {
    const int n = d > dy ? d : dy;
    GseqLds l;
    l.ld = gseq_ld(d, dy);
    l.X0 = sm;
    l.X1 = l.X0 + n * l.ld;
    l.X2 = l.X1 + n * l.ld;
    l.X3 = l.X2 + n * l.ld;
    l.m = l.X3 + n * l.ld;
    l.mp = l.m + n;
    l.r = l.mp + n;
    l.sr = l.r + n;
    l.buf = l.sr + n;
    l.yv = l.buf + 2 * n;
    return l;
}

// One time index of the forward pass on the belief (l.m, l.X0): `*`_A(:out) -> MvN_x(:out) when `predict` (cx: known input c[t]
// of this step or null), then the product with the observation branch unless l.yv holds a NaN.  Every thread of the workgroup
// calls it; starts and ends with the belief consistent across the workgroup (barrier at the end).  *logev -= the step's
// −log p(y_t | y_<t) on thread 0 when `fe`.  Returns false when S lost positive definiteness.
__device__ __forceinline__ bool gseq_forward_step(const GseqLds& l, const GenericModel& M, int d, int dy, bool predict, const double* cx,
                                                  const Map2& md, int* s_obs, bool fe, double* logev, int tid, int nt) {
    double *X0 = l.X0, *X1 = l.X1, *X2 = l.X2, *X3 = l.X3, *m = l.m, *mp = l.mp, *r = l.r, *sr = l.sr, *yv = l.yv;
    const int ld = l.ld;
    bool ok = true;
    if (predict) {  // A is staged in X3 (free until the observation update): an operand read from global memory costs an L2
        RXHIP_FOR_2D(md, d, d, i, j) X3[i * ld + j] = M.A[i * d + j];  // round trip per k step
        __syncthreads();
        tile_gemm(X1, ld, d, d, d, X3, ld, 1, X0, ld, 1, 1.0, false, tid, nt);       // A V
        for (int i = tid; i < d; i += nt) {
            double s = cx ? cx[i] : 0.0;
            for (int k = 0; k < d; ++k) s += X3[i * ld + k] * m[k];
            mp[i] = s;
        }
        __syncthreads();
        tile_gemm(X2, ld, d, d, d, X1, ld, 1, X3, 1, ld, 1.0, false, tid, nt);       // (A V) A′
        __syncthreads();
        RXHIP_FOR_2D(md, d, d, i, j) X0[i * ld + j] = 0.5 * (X2[i * ld + j] + X2[j * ld + i]) + 0.5 * (M.P[i * d + j] + M.P[j * d + i]);
    } else
        for (int i = tid; i < d; i += nt) mp[i] = m[i];
    RXHIP_FOR_2D(md, dy, d, a, k) X3[a * ld + k] = M.B[a * d + k];   // B, until S⁻¹ B V_p takes the buffer
    __syncthreads();
    if (tid == 0) {
        int obs = 1;
        for (int k = 0; k < dy; ++k) obs = obs && (yv[k] == yv[k]);
        *s_obs = obs;
    }
    __syncthreads();
    if (*s_obs) {  // uniform over the workgroup
        tile_gemm(X1, ld, dy, d, d, X3, ld, 1, X0, ld, 1, 1.0, false, tid, nt);      // B V_p
        for (int a = tid; a < dy; a += nt) {
            double s = yv[a];
            for (int k = 0; k < d; ++k) s -= X3[a * ld + k] * mp[k];
            r[a] = s;
        }
        __syncthreads();
        tile_gemm(X2, ld, dy, dy, d, X1, ld, 1, X3, 1, ld, 1.0, false, tid, nt);     // (B V_p) B′
        __syncthreads();
        sym_add(X2, ld, dy, M.Q, tid, nt);                                            // S
        __syncthreads();
        double logdet = 0.0;
        ok = lds_gj_inverse(X2, ld, dy, l.buf, &logdet, tid, nt);
        tile_gemm(X3, ld, dy, d, dy, X2, ld, 1, X1, ld, 1, 1.0, false, tid, nt);     // S⁻¹ B V_p
        for (int a = tid; a < dy; a += nt) {
            double s = 0.0;
            for (int k = 0; k < dy; ++k) s += X2[a * ld + k] * r[k];
            sr[a] = s;
        }
        __syncthreads();
        tile_gemm(X2, ld, d, d, dy, X1, 1, ld, X3, ld, 1, 1.0, false, tid, nt);      // (B V_p)′ S⁻¹ (B V_p)
        for (int i = tid; i < d; i += nt) {
            double s = mp[i];
            for (int a = 0; a < dy; ++a) s += X1[a * ld + i] * sr[a];
            m[i] = s;
        }
        if (fe && tid == 0) {
            double q = 0.0;
            for (int a = 0; a < dy; ++a) q += r[a] * sr[a];
            *logev -= 0.5 * ((double)dy * 1.8378770664093453 + logdet + q);
        }
        __syncthreads();
        RXHIP_FOR_2D(md, d, d, i, j) {
            if (j > i) continue;
            const double v = 0.5 * (X0[i * ld + j] + X0[j * ld + i]) - 0.5 * (X2[i * ld + j] + X2[j * ld + i]);
            X0[i * ld + j] = v;
            X0[j * ld + i] = v;
        }
    } else
        for (int i = tid; i < d; i += nt) m[i] = mp[i];
    __syncthreads();
    return ok;
}

__global__ void __launch_bounds__(256, 4) k_gseq_forward(GseqParams p) {
    RXHIP_GSEQ_EXTERN_SHARED(sm)
    const int d = p.d, dy = p.dy, tid = threadIdx.x, nt = blockDim.x;
    const GseqLds l = gseq_lds(sm, d, dy);
    const Map2 md(d, tid, nt);
    __shared__ int s_obs;
    const long long c = blockIdx.x;
    bool ok = true;
    double logev = 0.0;              // thread 0: Σ log p(y_t | y_<t)
    {
        const size_t idx = gseq_model_index(p, c, 0);
        const double* pr = p.prior + idx * ((size_t)d + (size_t)d * d);
        for (int i = tid; i < d; i += nt) l.m[i] = pr[i];
        RXHIP_FOR_2D(md, d, d, i, j) l.X0[i * l.ld + j] = pr[d + i * d + j];
    }
    __syncthreads();
    for (long long t = 0; t < p.T; ++t) {
        const GenericModel M = gseq_model(p, gseq_model_index(p, c, t));
        const long long row = t * p.n_chains + c;
        for (int a = tid; a < dy; a += nt) l.yv[a] = p.y[row * dy + a];
        ok = gseq_forward_step(l, M, d, dy, t > 0 || p.ptt, nullptr, md, &s_obs, p.fe != 0, &logev, tid, nt) && ok;
        for (int i = tid; i < d; i += nt) p.mean[row * d + i] = l.m[i];
        RXHIP_FOR_2D(md, d, d, i, j) p.cov[row * d * d + i * d + j] = l.X0[i * l.ld + j];
        __syncthreads();
    }
    if (tid == 0) {
        if (p.fe) p.fe_part[c] = logev;
        if (!ok) atomicOr(p.status, ST_NOT_POSDEF);
    }
}

// The streaming driver one observation at a time at any d, dy ≤ 64 (rxhip_filter_step; src/inference/streaming.jl:349-407):
// the belief of every chain stays in `state`, one call pushes one observation through the one-step graph.
struct GseqStreamParams {
    long long n_chains, k;   // k: number of observations seen so far (indexes step_model and the known inputs)
    int d, dy, ptt, first;   // first: the belief is the prior (through its transition when ptt)
    const double* y;         // [chain][dy]
    double* state;           // [chain][d + d²]  belief q(x) after the last step
    const double* user;
    const double* prior;
    const int* chain_model;
    const int* step_model;
    const double *cx, *cy;   // null, or known inputs [·][d], [·][dy] of observation k ([·][chain][·] when off_chain)
    int off_chain;
    double* mean;            // [chain][d]
    double* cov;             // [chain][d][d]
    double* fe;              // [chain]  −log p(y_k | y_<k)
    int* status;
};
__global__ void __launch_bounds__(256, 4) k_gseq_stream_step(GseqStreamParams p) {
    RXHIP_GSEQ_EXTERN_SHARED(sm)
    const int d = p.d, dy = p.dy, tid = threadIdx.x, nt = blockDim.x;
    const GseqLds l = gseq_lds(sm, d, dy);
    const Map2 md(d, tid, nt);
    __shared__ int s_obs;
    const long long c = blockIdx.x;
    const size_t idx = p.step_model ? (size_t)p.step_model[p.k] : p.chain_model ? (size_t)p.chain_model[c] : 0;
    GseqParams q{};
    q.d = d; q.dy = dy; q.user = p.user;
    const GenericModel M = gseq_model(q, idx);
    double* st = p.state + c * ((size_t)d + (size_t)d * d);
    const double* src = p.first ? p.prior + idx * ((size_t)d + (size_t)d * d) : st;
    for (int i = tid; i < d; i += nt) l.m[i] = src[i];
    RXHIP_FOR_2D(md, d, d, i, j) l.X0[i * l.ld + j] = src[d + i * d + j];
    const long long orow = p.k * (p.off_chain ? p.n_chains : 1) + (p.off_chain ? c : 0);
    for (int a = tid; a < dy; a += nt) l.yv[a] = p.y[c * dy + a] - (p.cy ? p.cy[orow * dy + a] : 0.0);
    __syncthreads();
    double logev = 0.0;
    const bool ok = gseq_forward_step(l, M, d, dy, !p.first || p.ptt, p.cx ? p.cx + orow * d : nullptr, md, &s_obs, true, &logev, tid, nt);
    for (int i = tid; i < d; i += nt) {
        st[i] = l.m[i];
        p.mean[c * d + i] = l.m[i];
    }
    RXHIP_FOR_2D(md, d, d, i, j) {
        st[d + i * d + j] = l.X0[i * l.ld + j];
        p.cov[(c * d + i) * d + j] = l.X0[i * l.ld + j];
    }
    if (tid == 0) {
        if (p.fe) p.fe[c] = -logev;
        if (!ok) atomicOr(p.status, ST_NOT_POSDEF);
    }
}

__global__ void __launch_bounds__(256, 4) k_gseq_backward(GseqParams p) {
    RXHIP_GSEQ_EXTERN_SHARED(sm)
    const int d = p.d, dy = p.dy, tid = threadIdx.x, nt = blockDim.x;
    const int n = d > dy ? d : dy, ld = gseq_ld(d, dy);
    const Map2 md(d, tid, nt);
    double* X0 = sm;                 // V_f(t), then V_p(t+1) and its inverse, then G V_s(t+1)
    double* X1 = X0 + n * ld;        // V_s(t+1), then V_s(t)
    double* X2 = X1 + n * ld;        // A V_f
    double* X3 = X2 + n * ld;        // A, then G
    double* ms = X3 + n * ld;        // m_s(t+1), then m_s(t)
    double* mf = ms + n;             // m_f(t)
    double* dm = mf + n;             // m_s(t+1) − A m_f
    double* buf = dm + n;            // 2n
    const long long c = blockIdx.x;
    bool ok = true;
    {
        const long long row = (p.T - 1) * p.n_chains + c;
        for (int i = tid; i < d; i += nt) ms[i] = p.mean[row * d + i];
        RXHIP_FOR_2D(md, d, d, i, j) X1[i * ld + j] = p.cov[row * d * d + i * d + j];
    }
    __syncthreads();
    for (long long t = p.T - 2; t >= 0; --t) {
        const GenericModel M = gseq_model(p, gseq_model_index(p, c, t + 1));  // the transition into x[t+1]
        const long long row = t * p.n_chains + c;
        const double* Vf = p.cov + row * d * d;
        for (int i = tid; i < d; i += nt) mf[i] = p.mean[row * d + i];
        RXHIP_FOR_2D(md, d, d, i, j) {
            X0[i * ld + j] = Vf[i * d + j];
            X3[i * ld + j] = M.A[i * d + j];   // A staged in LDS for the two products that use it
        }
        __syncthreads();
        tile_gemm(X2, ld, d, d, d, X3, ld, 1, X0, ld, 1, 1.0, false, tid, nt);           // A V_f
        for (int i = tid; i < d; i += nt) {
            double s = 0.0;
            for (int k = 0; k < d; ++k) s += X3[i * ld + k] * mf[k];
            dm[i] = ms[i] - s;
        }
        __syncthreads();
        tile_gemm(X0, ld, d, d, d, X2, ld, 1, X3, 1, ld, 1.0, false, tid, nt);           // (A V_f) A′   (V_f is re-read from memory below)
        __syncthreads();
        sym_add(X0, ld, d, M.P, tid, nt);                                                 // V_p(t+1)
        __syncthreads();
        ok = lds_gj_inverse(X0, ld, d, buf, nullptr, tid, nt) && ok;
        tile_gemm(X3, ld, d, d, d, X2, 1, ld, X0, ld, 1, 1.0, false, tid, nt);           // G = (A V_f)′ V_p⁻¹
        __syncthreads();
        tile_gemm(X0, ld, d, d, d, X3, ld, 1, X1, ld, 1, 1.0, false, tid, nt);           // G V_s(t+1)
        for (int i = tid; i < d; i += nt) {
            double s = mf[i];
            for (int k = 0; k < d; ++k) s += X3[i * ld + k] * dm[k];
            mf[i] = s;                                                                    // m_s(t): thread i owns entry i
        }
        __syncthreads();
        if (p.cross) RXHIP_FOR_2D(md, d, d, i, j) p.cross[row * d * d + i * d + j] = X0[i * ld + j];
        tile_gemm(X1, ld, d, d, d, X0, ld, 1, X3, 1, ld, 1.0, false, tid, nt);           // (G V_s⁺) G′
        tile_gemm(X1, ld, d, d, d, X3, ld, 1, X2, ld, 1, -1.0, true, tid, nt);           // − G (A V_f): same tile owner, no barrier
        __syncthreads();
        RXHIP_FOR_2D(md, d, d, i, j) {
            if (j > i) continue;
            const double v = 0.5 * (X1[i * ld + j] + X1[j * ld + i]) + 0.5 * (Vf[i * d + j] + Vf[j * d + i]);
            X1[i * ld + j] = v;
            X1[j * ld + i] = v;
        }
        for (int i = tid; i < d; i += nt) ms[i] = mf[i];
        __syncthreads();  // every read of the filtered row is done before it is overwritten
        for (int i = tid; i < d; i += nt) p.mean[row * d + i] = ms[i];
        RXHIP_FOR_2D(md, d, d, i, j) p.cov[row * d * d + i * d + j] = X1[i * ld + j];
        __syncthreads();
    }
    if (!ok && tid == 0) atomicOr(p.status, ST_NOT_POSDEF);
}

// Node-local joint marginal of every transition node MvNormalMeanCovariance(out = x[k+1], μ = A x[k] (+ c[k+1]), Σ = P) at any
// d ≤ 64 (predict_kernels.hpp k_joint is the d ≤ 4 form):
//     q(out, μ) = N([m_s(k+1); A m_s(k) + c], [[V_s(k+1), (A X)′], [A X, A V_s(k) A′]]),   X = Cov(x[k], x[k+1] | y) = `cross`.
// One workgroup per (node, chain); A, X and V_s(k) are staged in LDS, the two products run on the register tiles above.
struct JointParams {
    long long T, n_chains;
    int d, dy;
    const double* mean;      // [T][chain][d]   posteriors of the last smoothing run
    const double* cov;       // [T][chain][d][d]
    const double* cross;     // [T−1][chain][d][d]
    const double* user;
    const int* chain_model;
    const int* step_model;
    const double* cx;        // null, or known inputs c[t] ([T][d], or [T][chain][d] when off_chain)
    int off_chain;
    double* jmean;           // [T−1][chain][2d]
    double* jcov;            // [T−1][chain][2d][2d]
};
__host__ __device__ inline size_t joint_lds_bytes(int d) { return sizeof(double) * (4 * (size_t)d * (d | 1) + 2 * (size_t)d); }
__global__ void __launch_bounds__(256, 4) k_joint_generic(JointParams p) {
    RXHIP_GSEQ_EXTERN_SHARED(sm)
    const int d = p.d, ld = d | 1, tid = threadIdx.x, nt = blockDim.x;
    const Map2 md(d, tid, nt);
    double* LA = sm;                  // A
    double* LX = LA + d * ld;         // X, then A V_s(k)
    double* LV = LX + d * ld;         // V_s(k), then A V_s(k) A′
    double* LT = LV + d * ld;         // A X
    double* m0 = LT + d * ld;         // m_s(k)
    const long long g = blockIdx.x, k = g / p.n_chains, c = g - k * p.n_chains;
    GseqParams q{};
    q.d = d; q.dy = p.dy; q.user = p.user;
    const GenericModel M = gseq_model(q, p.step_model ? (size_t)p.step_model[k + 1] : p.chain_model ? (size_t)p.chain_model[c] : 0);
    const long long r0 = k * p.n_chains + c, r1 = (k + 1) * p.n_chains + c;
    RXHIP_FOR_2D(md, d, d, i, j) {
        LA[i * ld + j] = M.A[i * d + j];
        LX[i * ld + j] = p.cross[r0 * d * d + i * d + j];
        LV[i * ld + j] = p.cov[r0 * d * d + i * d + j];
    }
    for (int i = tid; i < d; i += nt) m0[i] = p.mean[r0 * d + i];
    __syncthreads();
    tile_gemm(LT, ld, d, d, d, LA, ld, 1, LX, ld, 1, 1.0, false, tid, nt);               // A X
    __syncthreads();
    tile_gemm(LX, ld, d, d, d, LA, ld, 1, LV, ld, 1, 1.0, false, tid, nt);               // A V_s(k)
    __syncthreads();
    tile_gemm(LV, ld, d, d, d, LX, ld, 1, LA, 1, ld, 1.0, false, tid, nt);               // (A V_s(k)) A′
    __syncthreads();
    double* jm = p.jmean + g * 2 * d;
    double* jc = p.jcov + g * 4 * d * d;
    for (int i = tid; i < d; i += nt) {
        double s = p.cx ? p.cx[((k + 1) * (p.off_chain ? p.n_chains : 1) + (p.off_chain ? c : 0)) * d + i] : 0.0;
        for (int j = 0; j < d; ++j) s += LA[i * ld + j] * m0[j];
        jm[i] = p.mean[r1 * d + i];
        jm[d + i] = s;
    }
    RXHIP_FOR_2D(md, d, d, i, j) {
        jc[(size_t)i * 2 * d + j] = p.cov[r1 * d * d + i * d + j];                        // (out, out) = V_s(k+1)
        jc[(size_t)i * 2 * d + d + j] = LT[j * ld + i];                                   // (out, μ) = (A X)′
        jc[(size_t)(d + i) * 2 * d + j] = LT[i * ld + j];                                 // (μ, out) = A X
        jc[(size_t)(d + i) * 2 * d + d + j] = 0.5 * (LV[i * ld + j] + LV[j * ld + i]);    // (μ, μ) = A V_s(k) A′
    }
}

}  // namespace rxhip
