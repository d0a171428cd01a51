// drift_kernels.hpp — sum-product on the noise-free drift chain of test/models/statespace/ulgssm_tests.jl:8-15
//     x_prior ~ Normal(μ = m0, v = v0);   x[t] ~ x[t-1] + c;   y[t] ~ Normal(μ = x[t], v = obs_var)
// Every transition is a deterministic `+` node, so the chain has ONE degree of freedom: x[t] = x0 + off_t with
// off_t = (t + ptt)·c (t = 0…T−1; ptt = 1 when the prior sits on x_prior, one `+` before the first observed state).
// The reference's forward / backward sweep through the `+`(:out) / `+`(:in1) rules and the products at every x[t]
// collapse to a reduction over time: Λ = 1/v0 + T/v, ξ = m0/v0 + Σ (y_t − off_t)/v; q(x[t]) = N(ξ/Λ + off_t, 1/Λ).
// Bethe free energy of the tree (node energies and entropies telescope):
//     F = ½ [ T log 2πv + Σ e_t²/v + (x̂0 − m0)²/v0 + log(v0 Λ) ],   e_t = y_t − off_t − x̂0.
// One workgroup per chain; fixed-shape reductions (bit-identical from run to run).
#pragma once
#include <hip/hip_runtime.h>

namespace rxhip {

struct DriftParams {
    long long T, n_chains;
    const double* y;   // [T][chain]
    double* mean;      // [T][chain]
    double* var;       // [T][chain]
    double* fe_chain;  // [chain]
    double m0, v0, c, obs_var;
    int ptt;
    int* status;
};

__device__ inline double drift_block_sum(double v, double* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) sh[w] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += sh[i];  // every thread: same order, same value
    return s;
}

template <bool FE>
__global__ __launch_bounds__(256) void k_drift_chain(DriftParams p) {
    __shared__ double sh[4];
    const long long ch = blockIdx.x, C = p.n_chains;
    const double iv = 1.0 / p.obs_var;
    double s = 0.0;
    for (long long t = threadIdx.x; t < p.T; t += blockDim.x) s += p.y[t * C + ch] - (double)(t + p.ptt) * p.c;
    s = drift_block_sum(s, sh);
    const double lam = 1.0 / p.v0 + (double)p.T * iv;
    const double V = 1.0 / lam;
    const double x0 = (p.m0 / p.v0 + s * iv) * V;
    double q = 0.0;
    for (long long t = threadIdx.x; t < p.T; t += blockDim.x) {
        const double off = (double)(t + p.ptt) * p.c;
        p.mean[t * C + ch] = x0 + off;
        p.var[t * C + ch] = V;
        if (FE) {
            const double e = p.y[t * C + ch] - off - x0;
            q += e * e;
        }
    }
    if (FE) {
        q = drift_block_sum(q, sh);
        if (threadIdx.x == 0) {
            const double dm = x0 - p.m0;
            const double f = 0.5 * ((double)p.T * log(6.283185307179586476925286766559 * p.obs_var) + q * iv + dm * dm / p.v0 + log(p.v0 * lam));
            p.fe_chain[ch] = f;
            if (!(f == f) || f - f != 0.0) atomicOr(p.status, 2);  // ST_NONFINITE_FE
        }
    }
}

// out[0] = Σ in[0..n) in a fixed order (one workgroup)
__global__ __launch_bounds__(256) void k_sum_fixed(const double* __restrict__ in, long long n, double* __restrict__ out) {
    __shared__ double sh[4];
    double s = 0.0;
    for (long long i = threadIdx.x; i < n; i += blockDim.x) s += in[i];
    s = drift_block_sum(s, sh);
    if (threadIdx.x == 0) out[0] = s;
}

}  // namespace rxhip
