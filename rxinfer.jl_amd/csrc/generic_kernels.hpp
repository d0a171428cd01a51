// generic_kernels.hpp — predictions and the unobserved tail for ANY state dimension (the MFMA path, d, dy ≤ 64).
//
// Not on the sweep's critical path: these run once per `rxhip_get_predictions` / per sweep with a forecast horizon, on the
// dense posteriors the sweep has written.  One workgroup per (chain, time index), matrices in LDS, plain loops over runtime
// dimensions.
//
// Prediction of y[t] (`obtain_prediction`, src/model/plugins/reactivemp_inference.jl:619-624): the message MvN_y(:out) =
// N(B m, B V B′ + Q) with (m, V) the posterior of x[t] with its own observation taken out again.  predict_kernels.hpp does
// that in the state space (Λ = V_s⁻¹ − B′Q⁻¹B: two d×d inverses).  Pushed through B it needs the observation space only:
// with Σ = B V_s B′, R = (Q − Σ)⁻¹ and u = B m_s − Σ Q⁻¹ y  (Woodbury on Λ⁻¹ = V_s + V_s B′ R B V_s)
//     mean = Q R u,        cov = Q + Σ R Q
// — one dy×dy inverse, no inverse of the posterior covariance.  A `missing` y[t] (t ≥ T) sent no message: N(B m_s, Σ + Q).
#pragma once
#include "lgssm_kernels.hpp"

namespace rxhip {

struct GenericModel {  // user-level constants of one model, row-major, in one device block
    const double *A, *P, *B, *Q, *Qi;
};
struct GenericParams {
    long long T, H, n_chains;
    int d, dy;
    const double* y;     // [T][chain][dy]
    double* mean;        // [T+H][chain][d]
    double* cov;         // [T+H][chain][d][d]
    const double* user;  // [n_models][2d² + dy·d + 2dy²]  A | P | B | Q | Q⁻¹
    const int* chain_model;
    const int* step_model;       // [T+H] model of a time index (per-step constants), or null
    const double *mu, *nu, *cx;  // known inputs (PredictParams): μ[t] [T+H][d], ν[t] [T+H][dy], c[t] [T+H][d]; null: none
    int off_chain;               // 1: with a chain axis, [T+H][chain][·]
    double* pmean;       // [T+H][chain][dy]
    double* pcov;        // [T+H][chain][dy][dy]
    int* status;
};
__device__ __forceinline__ GenericModel generic_model(const GenericParams& p, long long chain, long long t) {
    const size_t sz = 2 * (size_t)p.d * p.d + (size_t)p.dy * p.d + 2 * (size_t)p.dy * p.dy;
    const double* u = p.user + (p.step_model ? p.step_model[t] : p.chain_model ? p.chain_model[chain] : 0) * sz;
    GenericModel m;
    m.A = u;
    m.P = m.A + (size_t)p.d * p.d;
    m.B = m.P + (size_t)p.d * p.d;
    m.Q = m.B + (size_t)p.dy * p.d;
    m.Qi = m.Q + (size_t)p.dy * p.dy;
    return m;
}

// in-place inverse of a symmetric positive definite n×n matrix in LDS (Gauss–Jordan, no pivoting); false: a pivot ≤ 0
__device__ __forceinline__ bool lds_spd_inverse(double* M, int n, double* colbuf, int tid, int nthreads) {
    bool ok = true;
    for (int k = 0; k < n; ++k) {
        const double piv = M[k * n + k];
        ok = ok && piv > 0.0;
        const double r = 1.0 / piv;
        __syncthreads();
        for (int i = tid; i < n; i += nthreads) colbuf[i] = M[i * n + k];  // column k before it is overwritten
        __syncthreads();
        for (int e = tid; e < n * n; e += nthreads) {
            const int i = e / n, j = e - i * n;
            double v;
            if (i == k) v = (j == k) ? r : M[k * n + j] * r;
            else if (j == k) v = -colbuf[i] * r;
            else v = M[e] - colbuf[i] * M[k * n + j] * r;
            // every element reads row k and its own value: write after all reads of this sweep step
            colbuf[n + e] = v;
        }
        __syncthreads();
        for (int e = tid; e < n * n; e += nthreads) M[e] = colbuf[n + e];
        __syncthreads();  // the next pivot is another thread's element
    }
    return ok;
}

// LDS: BV dy·d | Sg dy² | Mx dy² | scratch (dy + dy²) | vectors 3·dy + d      (d = dy = 64: 133 KB)
__host__ __device__ inline size_t generic_predict_lds(int d, int dy) {
    return sizeof(double) * ((size_t)dy * d + 3 * (size_t)dy * dy + 4 * (size_t)dy + d);
}
__global__ void __launch_bounds__(256) k_predict_generic(GenericParams p) {
    extern __shared__ double sm[];
    const int d = p.d, dy = p.dy, tid = threadIdx.x, nt = blockDim.x;
    double* BV = sm;
    double* Sg = BV + dy * d;
    double* Mx = Sg + dy * dy;
    double* scr = Mx + dy * dy;          // dy + dy²
    double* u = scr + dy + dy * dy;      // dy
    double* w = u + dy;                  // dy
    double* qy = w + dy;                 // dy
    double* ms = qy + dy;                // d
    const long long g = blockIdx.x;      // row (t, chain)
    const long long t = g / p.n_chains, c = g - t * p.n_chains;
    const GenericModel M = generic_model(p, c, t);
    const double* Vs = p.cov + g * d * d;  // read straight from memory (L2): used once, in B V_s
    for (int e = tid; e < d; e += nt) ms[e] = p.mean[g * d + e] - (p.mu ? p.mu[(p.off_chain ? g : g / p.n_chains) * d + e] : 0.0);
    __shared__ int s_obs;
    if (tid == 0) {
        int obs = t < p.T;
        if (obs)
            for (int k = 0; k < dy; ++k) obs = obs && (p.y[g * dy + k] == p.y[g * dy + k]);
        s_obs = obs;
    }
    __syncthreads();
    const bool observed = s_obs != 0;
    for (int e = tid; e < dy * d; e += nt) {  // BV = B V_s
        const int a = e / d, j = e - a * d;
        double s = 0.0;
        for (int k = 0; k < d; ++k) s += M.B[a * d + k] * Vs[k * d + j];
        BV[e] = s;
    }
    __syncthreads();
    for (int e = tid; e < dy * dy; e += nt) {  // Σ = B V_s B′
        const int a = e / dy, b = e - a * dy;
        double s = 0.0;
        for (int k = 0; k < d; ++k) s += BV[a * d + k] * M.B[b * d + k];
        Sg[e] = s;
    }
    for (int a = tid; a < dy; a += nt) {  // B m_s and Q⁻¹ y
        double s = 0.0, q = 0.0;
        for (int k = 0; k < d; ++k) s += M.B[a * d + k] * ms[k];
        if (observed)
            for (int k = 0; k < dy; ++k) q += M.Qi[a * dy + k] * p.y[g * dy + k];
        u[a] = s;
        qy[a] = q;
    }
    __syncthreads();
    if (!observed) {
        for (int a = tid; a < dy; a += nt) p.pmean[g * dy + a] = u[a] + (p.nu ? p.nu[(p.off_chain ? g : t) * dy + a] : 0.0);
        for (int e = tid; e < dy * dy; e += nt) {
            const int a = e / dy, b = e - a * dy;
            p.pcov[g * dy * dy + e] = 0.5 * (Sg[a * dy + b] + Sg[b * dy + a]) + 0.5 * (M.Q[a * dy + b] + M.Q[b * dy + a]);
        }
        return;
    }
    for (int e = tid; e < dy * dy; e += nt) {  // Q − Σ, symmetrised
        const int a = e / dy, b = e - a * dy;
        Mx[e] = 0.5 * (M.Q[a * dy + b] + M.Q[b * dy + a]) - 0.5 * (Sg[a * dy + b] + Sg[b * dy + a]);
    }
    for (int a = tid; a < dy; a += nt) {  // u = B m_s − Σ Q⁻¹ y
        double s = u[a];
        for (int k = 0; k < dy; ++k) s -= Sg[a * dy + k] * qy[k];
        w[a] = s;
    }
    __syncthreads();
    const bool ok = lds_spd_inverse(Mx, dy, scr, tid, nt);  // R
    for (int a = tid; a < dy; a += nt) {  // R u
        double s = 0.0;
        for (int k = 0; k < dy; ++k) s += Mx[a * dy + k] * w[k];
        u[a] = s;
    }
    for (int e = tid; e < dy * dy; e += nt) {  // R Q  -> scratch
        const int a = e / dy, b = e - a * dy;
        double s = 0.0;
        for (int k = 0; k < dy; ++k) s += Mx[a * dy + k] * M.Q[k * dy + b];
        scr[dy + e] = s;
    }
    __syncthreads();
    for (int a = tid; a < dy; a += nt) {  // mean = Q R u
        double s = 0.0;
        for (int k = 0; k < dy; ++k) s += M.Q[a * dy + k] * u[k];
        p.pmean[g * dy + a] = s + (p.nu ? p.nu[(p.off_chain ? g : t) * dy + a] : 0.0);
    }
    for (int e = tid; e < dy * dy; e += nt) {  // cov = Q + Σ (R Q): symmetric in exact arithmetic, stored symmetrised
        const int a = e / dy, b = e - a * dy;
        double s = 0.0, s2 = 0.0;
        for (int k = 0; k < dy; ++k) {
            s += Sg[a * dy + k] * scr[dy + k * dy + b];
            s2 += Sg[b * dy + k] * scr[dy + k * dy + a];
        }
        p.pcov[g * dy * dy + e] = 0.5 * (M.Q[a * dy + b] + M.Q[b * dy + a]) + 0.5 * (s + s2);
    }
    if (!ok && tid == 0) atomicOr(p.status, ST_NOT_POSDEF);
}

// The unobserved tail (rxhip_lgssm_desc.horizon): `*`_A(:out) -> MvN_x(:out) from the last posterior, one workgroup per chain.
__host__ __device__ inline size_t generic_forecast_lds(int d) { return sizeof(double) * (2 * (size_t)d * d + 2 * (size_t)d); }
__global__ void __launch_bounds__(256) k_forecast_generic(GenericParams p) {
    extern __shared__ double sm[];
    const int d = p.d, tid = threadIdx.x, nt = blockDim.x;
    double* V = sm;
    double* AV = V + d * d;
    double* m = AV + d * d;
    double* mn = m + d;
    const long long c = blockIdx.x;
    const long long r0 = (p.T - 1) * p.n_chains + c;
    for (int e = tid; e < d * d; e += nt) V[e] = p.cov[r0 * d * d + e];
    for (int e = tid; e < d; e += nt) m[e] = p.mean[r0 * d + e];
    __syncthreads();
    for (long long h = 0; h < p.H; ++h) {
        const long long r = (p.T + h) * p.n_chains + c;
        const GenericModel M = generic_model(p, c, p.T + h);  // the transition into x[T+h]
        for (int e = tid; e < d * d; e += nt) {
            const int i = e / d, j = e - i * d;
            double s = 0.0;
            for (int k = 0; k < d; ++k) s += M.A[i * d + k] * V[k * d + j];
            AV[e] = s;
        }
        for (int i = tid; i < d; i += nt) {
            double s = 0.0;
            for (int k = 0; k < d; ++k) s += M.A[i * d + k] * m[k];
            mn[i] = s + (p.cx ? p.cx[((p.T + h) * (p.off_chain ? p.n_chains : 1) + (p.off_chain ? c : 0)) * d + i] : 0.0);
        }
        __syncthreads();
        for (int e = tid; e < d * d; e += nt) {
            const int i = e / d, j = e - i * d;
            double s = 0.0, s2 = 0.0;
            for (int k = 0; k < d; ++k) {
                s += AV[i * d + k] * M.A[j * d + k];
                s2 += AV[j * d + k] * M.A[i * d + k];
            }
            const double v = 0.5 * (s + s2) + 0.5 * (M.P[i * d + j] + M.P[j * d + i]);
            V[e] = v;
            p.cov[r * d * d + e] = v;
        }
        for (int i = tid; i < d; i += nt) {
            m[i] = mn[i];
            p.mean[r * d + i] = mn[i];
        }
        __syncthreads();
    }
}


// Known inputs that are DATA of every chain (`B_u * u[t]` with u a datavar): μ[t] = A μ[t-1] + c[t], ν[t] = B μ[t] + d[t] per
// chain, one thread each (sequential in t, four steps of inputs in flight).  cx, cy: [T+H][chain][·]; ab: A | B of every model.
struct MuParams {
    long long To, n_chains;
    int d, dy, ptt, n_models;
    const double *cx, *cy, *ab;
    const int* step_model;
    double *mu, *nu;
};
__global__ void __launch_bounds__(64) k_mu_recursion(MuParams p) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= p.n_chains) return;
    const int d = p.d, dy = p.dy;
    double m[64], mn[64];
    for (int i = 0; i < d; ++i) m[i] = 0.0;
    for (long long t = 0; t < p.To; ++t) {
        const size_t mdl = p.step_model ? (size_t)p.step_model[t] : 0;
        const double* A = p.ab + mdl * ((size_t)d * d + (size_t)dy * d);
        const double* B = A + (size_t)d * d;
        const long long r = t * p.n_chains + c;
        if (t > 0 || p.ptt) {
            for (int i = 0; i < d; ++i) {
                double s = p.cx ? p.cx[r * d + i] : 0.0;
                if (t > 0)
                    for (int k = 0; k < d; ++k) s += A[i * d + k] * m[k];
                mn[i] = s;
            }
            for (int i = 0; i < d; ++i) m[i] = mn[i];
        }
        for (int i = 0; i < d; ++i) p.mu[r * d + i] = m[i];
        for (int a = 0; a < dy; ++a) {
            double s = p.cy ? p.cy[r * dy + a] : 0.0;
            for (int k = 0; k < d; ++k) s += B[a * d + k] * m[k];
            p.nu[r * dy + a] = s;
        }
    }
}

}  // namespace rxhip
