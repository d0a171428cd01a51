// dense_mseg_kernels.hpp — `missing` observations on the MFMA path, PARALLEL IN TIME (round 3).
//
// The reference treats `missing` as first class (docs/src/manuals/inference/static.md:98-123, src/inference/batch.jl:221-227): the
// observation node of such a step sends no message.  On the device that makes every covariance of a chain depend on the chain
// and the time index, so none of the per-model tables of the time-parallel schedule survives (dense_kernels.hpp builds its
// segment boundaries from ONE model and a fully observed chain).  Rounds 1–2 ran these engines sequentially in time
// (gseq_kernels.hpp: one workgroup per chain — 686 ms for d = 64, T = 2000, against 0.40 ms for the fully observed chain).
// Here the segment boundaries are built per (chain, segment) ON THE DEVICE, with the mask applied per step:
//   km_mask      obs[chain][t] = 1 when y[t] of the chain is fully observed (a partly missing vector counts as missing), n_obs per chain
//   km_elements  one workgroup per (chain, segment): the known-start filter over the segment with the mask — the element (Π, C, J, b, η)
//                of Särkkä & García-Fernández (2021) and what the backward scan needs of it (C⁻¹, C⁻¹Π, J + Π'C⁻¹Π); B'Q⁻¹y_t of the
//                observed steps goes into the records for the sweep kernel
//   km_scan      per chain, two workgroups: prefix — the filtered belief at every segment start (V' = Π(V⁻¹ + J)⁻¹Π' + C, two SPD
//                inverses per segment); suffix — the backward message (Λβ, ξβ) at every segment end
//   km_bnd       (V(b_{s+1})⁻¹ + Λβ(b_{s+1}))⁻¹ for every inner boundary (parallel)
// and the sweep itself is kd_forward_info / kd_backward_info with per-chain boundaries and the observation precision B'Q⁻¹B left
// out of M_{t+1} at missing steps (DenseParams::mseg).  The free energy is evaluated as on the fully observed path (at the smoothed
// means), with the observation constants and residuals counted for observed steps only.
// The matrix products are the non-inlined blocks of dense_tab_kernels.hpp (operands in L2): ≈70 µs per step at d = 64 — a coverage
// path that is two orders of magnitude faster than the sequential one, not a roofline path: the boundary recursion is sequential
// over segments (S ≈ √(2T) balances it against the segment length), a tree scan over the elements is the next step.
// Scope: one model per engine (chain_model / step_model engines keep the sequential schedule), smoothing runs.
#pragma once
#include "dense_tab_kernels.hpp"

namespace rxhip {

struct MsegParams {
    int d, dy, dy_user, ptt;   // kernel-level dims (d = 16·NT, dy ≤ d); dy_user: length of an observation vector in y
    long long T, L, n_chains;
    int S;
    const double* y;        // [T][chain][dy_user]
    const double* in;       // padded model: A | P | V0 | B | Q | m0
    const double* cw;       // workspace of kt_consts (TabWs slots: Q⁻¹, P⁻¹, G, …, V1, V1⁻¹, V_f(1))
    double* ws;             // [chain·S + seg][MSEG_WS] d×d scratch matrices of km_elements; km_scan uses the first 2·chains blocks
    double* obs;            // [chain][T]
    double* nobs;           // [chain]
    double* mel;            // [chain][S][6][d][d]   Π, C, J, C⁻¹, C⁻¹Π, J + Π'C⁻¹Π
    double* mvec;           // [chain][S][2][d]      b, η
    double* mbnd;           // [chain][S][2][d][d]   Λ_f(b_s) | V_s(b_{s+1})
    double* mlb;            // [chain][S][d][d]      Λβ(b_{s+1})
    double* fstart_m;       // [chain][S][d]
    double* beta_xi;        // [chain][S+1][d]
    double* filt;           // records [chain][T][REC]: slot 1 of the header receives B'Q⁻¹y_t
    int rec;                // REC
    int* status;
};
constexpr int MSEG_WS = 14;

__global__ void km_mask(MsegParams p) {
    __shared__ double red[256];
    const long long chain = blockIdx.x;
    double cnt = 0.0;
    for (long long t = threadIdx.x; t < p.T; t += blockDim.x) {
        const double* yt = p.y + (t * p.n_chains + chain) * p.dy_user;
        bool ok = true;
        for (int k = 0; k < p.dy_user; ++k) ok = ok && yt[k] == yt[k];   // NaN = missing
        p.obs[chain * p.T + t] = ok ? 1.0 : 0.0;
        cnt += ok ? 1.0 : 0.0;
    }
    red[threadIdx.x] = cnt;
    __syncthreads();
    for (int n = blockDim.x >> 1; n > 0; n >>= 1) {
        if ((int)threadIdx.x < n) red[threadIdx.x] += red[threadIdx.x + n];
        __syncthreads();
    }
    if (threadIdx.x == 0) p.nobs[chain] = red[0];
}

// out[i] = Σ_k M[i][k] x[k] (+ add[i]); M row-major d×d in global memory, x / out in LDS; thread i < D
template <int D>
__device__ __forceinline__ double tab_row_dot(const double* M, int i, const double* x) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
    for (int k = 0; k < D; k += 2) {
        s0 += M[i * D + k] * x[k];
        s1 += M[i * D + k + 1] * x[k + 1];
    }
    return s0 + s1;
}
template <int D>
__device__ __forceinline__ double tab_col_dot(const double* M, int i, const double* x) {   // Σ_k M[k][i] x[k]
    double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
    for (int k = 0; k < D; k += 2) {
        s0 += M[k * D + i] * x[k];
        s1 += M[(k + 1) * D + i] * x[k + 1];
    }
    return s0 + s1;
}

template <int NT>
__global__ void __launch_bounds__(64 * NT) km_elements(MsegParams p) {
    constexpr int D = 16 * NT, MM = D * D;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    TabOps<NT> o{(int)threadIdx.x, (int)threadIdx.x >> 6, (int)threadIdx.x & 63, smem};
    double* vec = smem + blk_scratch_doubles(NT) + 2 * 64 * NT;   // m | am | e | eta | yv
    double *m = vec, *am = vec + D, *ev = vec + 2 * D, *eta = vec + 3 * D, *yv = vec + 4 * D;
    const int tid = o.tid, dyu = p.dy_user;
    const long long seg = blockIdx.x, chain = blockIdx.y;
    const double *A = p.in, *P = p.in + MM, *B = p.in + 3 * MM, *Q = p.in + 4 * MM;
    auto CW = [&](int slot) { return p.cw + (size_t)slot * MM; };
    const double *HF = CW(TabWs::HF), *G = CW(TabWs::G);
    double* W = p.ws + ((size_t)chain * p.S + seg) * MSEG_WS * MM;
    double *V = W, *Pi = W + MM, *J = W + 2 * MM, *Vp = W + 3 * MM, *Si = W + 4 * MM, *K = W + 5 * MM, *U = W + 6 * MM,
           *HFPi = W + 7 * MM, *T1 = W + 8 * MM, *T2 = W + 9 * MM, *Phi = W + 10 * MM, *T3 = W + 11 * MM;
    const long long t0 = seg * p.L;   // boundary b_s: state index of the known start
    long long t1 = t0 + p.L;
    if (t1 > p.T - 1) t1 = p.T - 1;
    bool ok = true;
    o.eye(V, 0.0);
    o.eye(J, 0.0);
    o.eye(Pi, 1.0);
    if (tid < D) { m[tid] = 0.0; eta[tid] = 0.0; }
    o.sync();
    for (long long t = t0 + 1; t <= t1; ++t) {
        const bool ob = p.obs[chain * p.T + t] != 0.0;   // uniform
        double* rec = p.filt + (chain * p.T + t) * p.rec;
        o.template mm<false, false>(T1, A, V);
        o.template mm<false, true>(Vp, T1, A, 1.0, P, 1.0);              // V_p = A V A' + P
        if (tid < D) am[tid] = tab_row_dot<D>(A, tid, m);               // A m
        if (ob) {
            if (tid < D) yv[tid] = tid < dyu ? p.y[(t * p.n_chains + chain) * dyu + tid] : 0.0;
            o.template mm<false, false>(T1, B, Vp);                      // B V_p  (also a barrier: am, yv are visible)
            o.template mm<false, true>(T2, T1, B, 1.0, Q, 1.0);          // S = B V_p B' + Q
            ok = o.inv(Si, T2, nullptr) && ok;
            o.template mm<true, false>(K, T1, Si);                       // K = V_p B' S⁻¹
            o.template mm<false, false>(HFPi, HF, Pi);                   // (BA) Π
            o.template mm<true, false>(U, HFPi, Si);                     // U = ((BA)Π)' S⁻¹
            o.template mm<false, false>(J, U, HFPi, 1.0, J, 1.0);        // J += U (BA)Π
            o.template mm<false, false>(T2, K, T1, -1.0, Vp, 1.0);       // V_p − K B V_p
            o.sym(V, T2);
            o.template mm<false, false>(Phi, K, HF, -1.0, A, 1.0);       // Φ = A − K (BA)
            if (tid < D) {
                ev[tid] = yv[tid] - tab_row_dot<D>(B, tid, am);          // e = y − B A m   (rows ≥ dy: 0 − 0)
                rec[D + tid] = tab_row_dot<D>(G, tid, yv);               // B'Q⁻¹y_t for the sweep kernel
            }
            o.sync();
            if (tid < D) {
                eta[tid] += tab_row_dot<D>(U, tid, ev);
                m[tid] = am[tid] + tab_row_dot<D>(K, tid, ev);
            }
            o.template mm<false, false>(T3, Phi, Pi);
        } else {
            o.lin(V, 1.0, Vp);                                           // no message from the observation node: filtered = predicted
            if (tid < D) { m[tid] = am[tid]; rec[D + tid] = 0.0; }
            o.template mm<false, false>(T3, A, Pi);                      // Φ = A
        }
        o.lin(Pi, 1.0, T3);
    }
    double* g = p.mel + ((size_t)chain * p.S + seg) * 6 * MM;   // Π, C, J, C⁻¹, X = C⁻¹Π, JJ = J + Π'X
    o.lin(g, 1.0, Pi);
    o.lin(g + MM, 1.0, V);
    o.sym(g + 2 * MM, J);
    if (t1 > t0) {
        ok = o.inv(g + 3 * MM, V, nullptr) && ok;
        o.template mm<false, false>(g + 4 * MM, g + 3 * MM, Pi);
        o.template mm<true, false>(g + 5 * MM, Pi, g + 4 * MM, 1.0, g + 2 * MM, 1.0);
    }
    if (tid < D) {
        double* v = p.mvec + ((size_t)chain * p.S + seg) * 2 * D;
        v[tid] = m[tid];
        v[D + tid] = eta[tid];
    }
    if (!ok && tid == 0) atomicOr(p.status, ST_NOT_POSDEF);
}

// One segment per chain (batches with enough chains to fill the machine: mseg_setup): no element is needed, only what km_elements hands
// to the sweep kernel besides it — B'Q⁻¹y_t of the observed steps, 0 for the missing ones.  One thread per (chain, t ≥ 1, component).
__global__ void __launch_bounds__(256) km_gy(MsegParams p) {
    const int D = p.d, dyu = p.dy_user;
    const long long chain = blockIdx.y, idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long t = 1 + idx / D;
    const int i = (int)(idx - (t - 1) * D);
    if (t >= p.T) return;
    const double* G = p.cw + (size_t)TabWs::G * D * D + (size_t)i * D;
    const double* yt = p.y + (t * p.n_chains + chain) * dyu;
    double s = 0.0;
    if (p.obs[chain * p.T + t] != 0.0)
        for (int k = 0; k < dyu; ++k) s += G[k] * yt[k];
    p.filt[(chain * p.T + t) * p.rec + D + i] = s;
}

// boundary recursion over the segments of one chain: blockIdx.x = 0 prefix, 1 suffix
template <int NT>
__global__ void __launch_bounds__(64 * NT) km_scan(MsegParams p) {
    constexpr int D = 16 * NT, MM = D * D;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    TabOps<NT> o{(int)threadIdx.x, (int)threadIdx.x >> 6, (int)threadIdx.x & 63, smem};
    double* vec = smem + blk_scratch_doubles(NT) + 2 * 64 * NT;
    double *m = vec, *u = vec + D, *xi = vec + 2 * D, *tv = vec + 3 * D;
    const int tid = o.tid, dir = blockIdx.x, S = p.S, dyu = p.dy_user;
    const long long chain = blockIdx.y;
    auto CW = [&](int slot) { return p.cw + (size_t)slot * MM; };
    double* W = p.ws + ((size_t)chain * 2 + dir) * MSEG_WS * MM;   // km_elements is done with the workspace: reuse it
    double *cur = W, *tt = W + 2 * MM, *Wm = W + 3 * MM, *M1 = W + 4 * MM, *M2 = W + 5 * MM, *nxt = W + 6 * MM;
    bool ok = true;
    if (dir == 0) {
        // belief at t = 0: prior ⊗ observation message (if y_0 is observed)
        const bool ob0 = p.obs[chain * p.T] != 0.0;
        const double* m1v = p.in + 5 * MM;   // m0; through the transition when the prior sits on x_0 of the reference's other spelling
        if (tid < D) {
            double mm1 = m1v[tid];
            if (p.ptt) mm1 = tab_row_dot<D>(p.in, tid, m1v);           // A m0
            u[tid] = mm1;
            tv[tid] = tid < dyu ? (ob0 ? p.y[(0 * p.n_chains + chain) * dyu + tid] : 0.0) : 0.0;
        }
        o.sync();
        if (ob0) {
            o.lin(cur, 1.0, CW(TabWs::VF1));
            if (tid < D) xi[tid] = tab_row_dot<D>(CW(TabWs::V1I), tid, u) + tab_row_dot<D>(CW(TabWs::G), tid, tv);   // V1⁻¹m1 + G y
            o.sync();
            if (tid < D) m[tid] = tab_row_dot<D>(CW(TabWs::VF1), tid, xi);
        } else {
            o.lin(cur, 1.0, CW(TabWs::V1));
            if (tid < D) m[tid] = u[tid];
        }
        o.sync();
        for (int s = 0; s < S; ++s) {
            if (tid < D) p.fstart_m[((size_t)chain * S + s) * D + tid] = m[tid];
            double* bn = p.mbnd + ((size_t)chain * S + s) * 2 * MM;
            ok = o.inv(bn, cur, nullptr) && ok;                           // Λ_f(b_s) = V(b_s)⁻¹
            if (s == S - 1) break;
            const double* g = p.mel + ((size_t)chain * S + s) * 6 * MM;
            const double* gv = p.mvec + ((size_t)chain * S + s) * 2 * D;
            o.lin(tt, 1.0, bn, 1.0, g + 2 * MM);                          // V⁻¹ + J
            ok = o.inv(Wm, tt, nullptr) && ok;
            o.template mm<false, false>(M2, g, Wm);                       // M2 = Π W
            if (tid < D) u[tid] = tab_row_dot<D>(bn, tid, m) + gv[D + tid];   // V⁻¹m + η
            o.template mm<false, true>(tt, M2, g);                        // M2 Π'   (barrier: u visible)
            if (tid < D) tv[tid] = tab_row_dot<D>(M2, tid, u) + gv[tid];  // m' = M2 (V⁻¹m + η) + b
            o.lin(nxt, 0.5, tt, 0.5, tt, true);
            o.lin(cur, 1.0, nxt, 1.0, g + MM);                            // V' = sym(M2 Π') + C
            if (tid < D) m[tid] = tv[tid];
            o.sync();
        }
    } else {
        o.eye(cur, 0.0);   // Λβ(b_S) = 0
        if (tid < D) xi[tid] = 0.0;
        o.sync();
        for (int s = S - 1; s >= 0; --s) {
            // message at the END boundary of segment s
            o.lin(p.mlb + ((size_t)chain * S + s) * MM, 1.0, cur);
            if (tid < D) p.beta_xi[((size_t)chain * (S + 1) + s + 1) * D + tid] = xi[tid];
            if (s == 0) break;
            const double* g = p.mel + ((size_t)chain * S + s) * 6 * MM;   // C⁻¹ = g+3MM, X = g+4MM, JJ = g+5MM
            const double* gv = p.mvec + ((size_t)chain * S + s) * 2 * D;
            o.lin(tt, 1.0, g + 3 * MM, 1.0, cur);
            ok = o.inv(Wm, tt, nullptr) && ok;                            // (C⁻¹ + Λβ)⁻¹
            o.template mm<true, false>(M1, g + 4 * MM, Wm);               // N1 = X'W
            o.template mm<false, false>(M2, M1, cur);                     // N2 = N1 Λβ
            o.template mm<false, false>(tt, M1, g + 4 * MM);              // N1 X
            if (tid < D) tv[tid] = gv[D + tid] - tab_row_dot<D>(M2, tid, gv) + tab_row_dot<D>(M1, tid, xi);   // η − N2 b + N1 ξβ
            o.lin(nxt, -0.5, tt, -0.5, tt, true);
            o.lin(cur, 1.0, nxt, 1.0, g + 5 * MM);                        // JJ − sym(N1 X)
            if (tid < D) xi[tid] = tv[tid];
            o.sync();
        }
    }
    if (!ok && tid == 0) atomicOr(p.status, ST_NOT_POSDEF);
}

// smoothed covariance at the inner boundaries: V_s(b_{s+1}) = (Λ_f(b_{s+1}) + Λβ(b_{s+1}))⁻¹
template <int NT>
__global__ void __launch_bounds__(64 * NT) km_bnd(MsegParams p) {
    constexpr int D = 16 * NT, MM = D * D;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const long long seg = blockIdx.x, chain = blockIdx.y;
    if (seg + 1 >= p.S) return;
    Acc<NT> a;
    acc_load<NT>(a, p.mbnd + (((size_t)chain * p.S + seg + 1) * 2 + 0) * MM, D, w, lane);
    acc_add_mat<NT>(a, p.mlb + ((size_t)chain * p.S + seg) * MM, D, w, lane, 1.0);
    LogProd lp;
    const bool ok = blk_inverse<NT>(a, smem, w, lane, lp);
    acc_store<NT>(a, p.mbnd + (((size_t)chain * p.S + seg) * 2 + 1) * MM, D, w, lane);
    if (!ok && tid == 0) atomicOr(p.status, ST_NOT_POSDEF);
}

}  // namespace rxhip
