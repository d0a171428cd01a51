// predict_kernels.hpp — predictions of the data variables and the unobserved tail of a state-space chain (d, dy ≤ 4).
//
// Replaces `obtain_prediction(ref)` (src/model/plugins/reactivemp_inference.jl:619-624): the stream of the message toward a
// data variable.  For y[t] that message is MvN_y(:out)(m_μ, q_Σ) = N(B m, B V B' + Q), where (m, V) is the message
// x[t] -> `*`_B: the product of the OTHER messages into x[t] (forward ⊗ backward), its own observation excluded.  The
// sweep has already formed the full product q(x[t]) = fwd ⊗ obs ⊗ bwd, so the leave-one-out belief is that posterior with
// the observation's information taken out again:
//     Λ = V_s⁻¹ − B'Q⁻¹B,    ξ = V_s⁻¹ m_s − B'Q⁻¹ y_t,    (m, V) = mean_cov(ξ, Λ)
// — independent for every (chain, t): one thread each, no recursion.  Time indices beyond the last observation
// (`missing` observations, rxhip_lgssm_desc.horizon) carry forward messages only: k_forecast runs
// `*`_A(:out) -> MvN_x(:out) from the last filtered belief, one thread per chain, and their predictions are N(B m, B V B' + Q)
// of those posteriors directly.
#pragma once
#include "lgssm_kernels.hpp"

namespace rxhip {

struct PredictParams {
    long long T, H, n_chains;
    const double* y;      // [T][chain][DY]
    double* mean;         // [T+H][chain][D]      posteriors (rows ≥ T written by k_forecast)
    double* cov;          // [T+H][chain][D][D]
    const double* cst;    // [n_models][CstLayout::SIZE]
    const double* bq;     // [n_models][DY·D + DY·DY]   B | Q (row-major)
    const int* chain_model;
    const int* step_model; // [T+H] or null: time-varying constants (Params::step_model)
    // node-local joints (k_joint): the forward message's covariance of every step, as the sweep left it
    const double* filt;   // per-chain records [T][chain/64][NP2][64][2] (per-chain models, masked / per-step schedules) …
    const double* vtab;   // … or [T][NS], one copy for a shared-model batch (then `filt` is null)
    const double* tinv_tc; // null, or Params::elem after a sweep of k_forward_tinv: first mean-only record of (segment, chain) at [(seg·2D)·chains + chain]
    long long L;           // … with the segment length of that sweep: the records tc … te − 1 of a segment have the covariance of record te
    double* jmean;        // [T-1][chain][2D]
    double* jcov;         // [T-1][chain][2D][2D]
    const double* mu;     // known inputs (null: none): μ[t] [T+H][D] — the sweep ran on x − μ, y − ν; posteriors are back in x
    const double* nu;     // ν[t] = B μ[t] + d[t] [T+H][DY]
    const double* cx;     // c[t] [T+H][D]
    int off_chain;        // 1: the three arrays carry a chain axis, [T+H][chain][·] (inputs that are data of every chain)
    double* pmean;        // [T+H][chain][DY]
    double* pcov;         // [T+H][chain][DY][DY]
    int* status;
};

template <int D, int DY>
__global__ __launch_bounds__(64) void k_forecast(PredictParams p) {
    using CL = CstLayout<D, DY>;
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= p.n_chains) return;
    const double* cst = p.cst + (size_t)(p.chain_model ? p.chain_model[c] : 0) * CL::SIZE;
    double m[D];
    Sym<D> V;
    const long long r0 = (p.T - 1) * p.n_chains + c;
#pragma unroll
    for (int i = 0; i < D; ++i) m[i] = p.mean[r0 * D + i];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) V(i, j) = p.cov[r0 * D * D + i * D + j];
    for (long long h = 0; h < p.H; ++h) {
        double mn[D], Tm[D][D];
        Sym<D> Vn;
        if (p.step_model) cst = p.cst + (size_t)p.step_model[p.T + h] * CL::SIZE;
        const CPtr A{cst + CL::A}, P{cst + CL::P};
        matvec_c<D>(A, m, mn);             // `*`_A(:out): N(A m, A V A')
        if (p.cx) {                        // `+` with the known input of this time index
#pragma unroll
            for (int i = 0; i < D; ++i) mn[i] += p.cx[((p.T + h) * (p.off_chain ? p.n_chains : 1) + (p.off_chain ? c : 0)) * D + i];
        }
        predict_cov<D>(A, P, V, Tm, Vn);   // MvN_x(:out): + P
        const long long r = (p.T + h) * p.n_chains + c;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            m[i] = mn[i];
            p.mean[r * D + i] = mn[i];
        }
        V = Vn;
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) p.cov[r * D * D + i * D + j] = Vn(i, j);
    }
}

template <int D, int DY>
__global__ __launch_bounds__(256) void k_predict(PredictParams p) {
    using CL = CstLayout<D, DY>;
    const long long total = (p.T + p.H) * p.n_chains;
    bool ok = true;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const long long t = g / p.n_chains, c = g - t * p.n_chains;
        const int mdl = p.step_model ? p.step_model[t] : p.chain_model ? p.chain_model[c] : 0;
        const double* cst = p.cst + (size_t)mdl * CL::SIZE;
        const double* B = p.bq + (size_t)mdl * (DY * D + DY * DY);
        const double* Q = B + DY * D;
        double m[D];
        Sym<D> V;
#pragma unroll
        for (int i = 0; i < D; ++i) m[i] = p.mean[g * D + i] - (p.mu ? p.mu[(p.off_chain ? g : t) * D + i] : 0.0);
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) V(i, j) = p.cov[g * D * D + i * D + j];
        double yv[DY];
        bool observed = t < p.T;
        if (observed) {
#pragma unroll
            for (int k = 0; k < DY; ++k) yv[k] = p.y[g * DY + k];
            observed = !obs_missing<DY>(yv);  // a `missing` y[t] sent no message: the posterior IS the leave-one-out belief
        }
        if (observed) {  // take this step's observation message out of the posterior again
            Sym<D> Ls, L;
            double det, xi[D];
            ok = spd_inv<D>(V, Ls, det) && ok;
            symv<D>(Ls, m, xi);
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double s = xi[i];
#pragma unroll
                for (int k = 0; k < DY; ++k) s -= cst[CL::G + i * DY + k] * yv[k];
                xi[i] = s;
            }
#pragma unroll
            for (int q = 0; q < D * (D + 1) / 2; ++q) L.v[q] = Ls.v[q] - cst[CL::LOBS + q];
            ok = spd_inv<D>(L, V, det) && ok;
            symv<D>(V, xi, m);
        }
        // `*`_B(:out) = N(B m, B V B'), MvN_y(:out) = + Q
        double BV[DY][D];
#pragma unroll
        for (int a = 0; a < DY; ++a) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) s += B[a * D + k] * m[k];
            p.pmean[g * DY + a] = s + (p.nu ? p.nu[(p.off_chain ? g : t) * DY + a] : 0.0);
#pragma unroll
            for (int j = 0; j < D; ++j) {
                double v = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) v += B[a * D + k] * V(k, j);
                BV[a][j] = v;
            }
        }
#pragma unroll
        for (int a = 0; a < DY; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) {
                double s = 0.5 * (Q[a * DY + b] + Q[b * DY + a]);
#pragma unroll
                for (int k = 0; k < D; ++k) s += BV[a][k] * B[b * D + k];
                p.pcov[g * DY * DY + a * DY + b] = s;
                p.pcov[g * DY * DY + b * DY + a] = s;
            }
    }
    if (!ok) atomicOr(p.status, ST_NOT_POSDEF);
}


// Node-local joint marginal of every transition node MvNormalMeanCovariance(out = x[t], μ = A x[t-1], Σ = P), t = 2 … T:
// what `@marginalrule MvNormalMeanCovariance(:out_μ)` forms in the reference (the joint the Bethe energy of the node
// is taken over, src/model/plugins/reactivemp_force_marginal_computation_plugin.jl:52-98 for the deterministic neighbours).
// On a chain it follows from the sweep's own results:  Cov(x[t-1], x[t] | y) = G V_s(t),  G = V_f(t-1) A′ (A V_f(t-1) A′ + P)⁻¹,
// so  q(out, μ) = N([m_s(t); A m_s(t-1)], [[V_s(t), (A X)′], [A X, A V_s(t-1) A′]])  with X = G V_s(t).  One thread per (chain, node).
template <int D, int DY>
__global__ __launch_bounds__(256) void k_joint(PredictParams p) {
    using CL = CstLayout<D, DY>;
    constexpr int NS = Dim<D>::NS, NP2 = Dim<D>::NP2;
    const long long total = (p.T - 1) * p.n_chains;
    bool ok = true;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const long long k = g / p.n_chains, c = g - k * p.n_chains;  // node k joins x[k] (μ side) and x[k+1] (out), zero-based
        const int mdl = p.step_model ? p.step_model[k + 1] : p.chain_model ? p.chain_model[c] : 0;
        const double* cst = p.cst + (size_t)mdl * CL::SIZE;
        Sym<D> Vf, Vp, Lp;
        if (p.filt) {
            double2 r[NP2];
            double mf[D];
            long long kr = k;
            if (p.tinv_tc && k >= 1) {
                const long long seg = (k - 1) / p.L;
                long long te = (seg + 1) * p.L;
                te = te < p.T - 1 ? te : p.T - 1;
                const long long tc = (long long)p.tinv_tc[(seg * 2 * D) * p.n_chains + c];
                if (k >= tc && k < te) kr = te;   // a mean-only record: the segment's last record carries its covariance
            }
            load_filt_raw<D>(p.filt, kr, p.n_chains, c, r);
            unpack_rec<D>(r, mf, Vf);
        } else {
#pragma unroll
            for (int q = 0; q < NS; ++q) Vf.v[q] = p.vtab[k * NS + q];
        }
        double T[D][D], G[D][D], X[D][D], AX[D][D], det;
        predict_cov<D>(CPtr{cst + CL::A}, CPtr{cst + CL::P}, Vf, T, Vp);
        ok = spd_inv<D>(Vp, Lp, det) && ok;
        const double* v0 = p.cov + (k * p.n_chains + c) * D * D;        // V_s(k)
        const double* v1 = p.cov + ((k + 1) * p.n_chains + c) * D * D;  // V_s(k+1)
        const double* m0 = p.mean + (k * p.n_chains + c) * D;
        const double* m1 = p.mean + ((k + 1) * p.n_chains + c) * D;
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
            for (int b = 0; b < D; ++b) {
                double s = 0.0;
#pragma unroll
                for (int q = 0; q < D; ++q) s += T[q][a] * Lp(q, b);  // G = (A V_f)′ V_p⁻¹
                G[a][b] = s;
            }
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
            for (int b = 0; b < D; ++b) {
                double s = 0.0;
#pragma unroll
                for (int q = 0; q < D; ++q) s += G[a][q] * v1[q * D + b];
                X[a][b] = s;
            }
        double AV[D][D];
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
            for (int b = 0; b < D; ++b) {
                double s = 0.0, w = 0.0;
#pragma unroll
                for (int q = 0; q < D; ++q) {
                    s += cst[CL::A + a * D + q] * X[q][b];
                    w += cst[CL::A + a * D + q] * v0[q * D + b];
                }
                AX[a][b] = s;
                AV[a][b] = w;
            }
        double* jm = p.jmean + g * 2 * D;
        double* jc = p.jcov + g * 4 * D * D;
#pragma unroll
        for (int a = 0; a < D; ++a) {
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < D; ++q) s += cst[CL::A + a * D + q] * m0[q];
            jm[a] = m1[a];
            jm[D + a] = s + (p.cx ? p.cx[((k + 1) * (p.off_chain ? p.n_chains : 1) + (p.off_chain ? c : 0)) * D + a] : 0.0);
        }
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
            for (int b = 0; b < D; ++b) {
                double s = 0.0;
#pragma unroll
                for (int q = 0; q < D; ++q) s += AV[a][q] * cst[CL::A + b * D + q];  // A V_s(k) A′
                jc[a * 2 * D + b] = v1[a * D + b];
                jc[a * 2 * D + D + b] = AX[b][a];
                jc[(D + a) * 2 * D + b] = AX[a][b];
                jc[(D + a) * 2 * D + D + b] = s;
            }
    }
    if (!ok) atomicOr(p.status, ST_NOT_POSDEF);
}


// One step of the streaming driver (src/inference/streaming.jl:349-407 with `@autoupdates`: every new observation fires the
// one-step graph and the posterior becomes the next prior): the belief of every chain lives in `state`, one thread per chain.
//   prior (first step) or `*`_A(:out) -> [+ c_k] -> MvN_x(:out), then the product with the observation message of y (missing: none)
// `k` counts the observations of the stream; per-step constants and known inputs are indexed by it.
struct StreamParams {
    long long n_chains, k;
    int ptt, first;          // first: the belief is the prior (through its transition when ptt)
    const double* y;         // [chain][DY]  (device)
    double* state;           // [chain][D + NS]  belief q(x) after the last step
    const double* cst;
    const int* chain_model;
    const int* step_model;   // null, or model of observation k (k < its length)
    const double *cx, *cy;   // null, or known inputs [·][D], [·][DY] of observation k ([·][chain][·] when off_chain)
    int off_chain;
    double* mean;            // [chain][D]
    double* cov;             // [chain][D][D]
    double* fe;              // [chain]  −log p(y_k | y_<k), or null
    int* status;
};
template <int D, int DY>
__global__ __launch_bounds__(64) void k_stream_step(StreamParams p) {
    using CL = CstLayout<D, DY>;
    constexpr int NS = Dim<D>::NS;
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= p.n_chains) return;
    const int mdl = p.step_model ? p.step_model[p.k] : p.chain_model ? p.chain_model[c] : 0;
    const double* cst = p.cst + (size_t)mdl * CL::SIZE;
    double m[D], mp[D], yv[DY];
    Sym<D> V, Vp;
    double* st = p.state + c * (D + NS);
    if (p.first) {  // CL::M1 / V1 hold the prior already pushed through its transition when ptt
#pragma unroll
        for (int i = 0; i < D; ++i) mp[i] = cst[CL::M1 + i] + ((p.ptt && p.cx) ? p.cx[(p.k * (p.off_chain ? p.n_chains : 1) + (p.off_chain ? c : 0)) * D + i] : 0.0);
#pragma unroll
        for (int i = 0; i < NS; ++i) Vp.v[i] = cst[CL::V1 + i];
    } else {
#pragma unroll
        for (int i = 0; i < D; ++i) m[i] = st[i];
#pragma unroll
        for (int i = 0; i < NS; ++i) V.v[i] = st[D + i];
        double T[D][D];
        matvec_c<D>(CPtr{cst + CL::A}, m, mp);
        if (p.cx) {
#pragma unroll
            for (int i = 0; i < D; ++i) mp[i] += p.cx[(p.k * (p.off_chain ? p.n_chains : 1) + (p.off_chain ? c : 0)) * D + i];
        }
        predict_cov<D>(CPtr{cst + CL::A}, CPtr{cst + CL::P}, V, T, Vp);
    }
    bool miss = false;
#pragma unroll
    for (int a = 0; a < DY; ++a) {
        yv[a] = p.y[c * DY + a];
        miss = miss || (yv[a] != yv[a]);
        if (p.cy) yv[a] -= p.cy[(p.k * (p.off_chain ? p.n_chains : 1) + (p.off_chain ? c : 0)) * DY + a];
    }
    bool ok = true;
    double quad = 0.0, detprod = 1.0;
    obs_update<D, DY, true, true>(CPtr{cst}, mp, Vp, yv, m, V, ok, quad, detprod, miss);
#pragma unroll
    for (int i = 0; i < D; ++i) {
        st[i] = m[i];
        p.mean[c * D + i] = m[i];
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) st[D + i] = V.v[i];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) p.cov[(c * D + i) * D + j] = V(i, j);
    if (p.fe) p.fe[c] = miss ? 0.0 : 0.5 * (quad + log(detprod));
    if (!ok) atomicOr(p.status, ST_NOT_POSDEF);
}

}  // namespace rxhip
