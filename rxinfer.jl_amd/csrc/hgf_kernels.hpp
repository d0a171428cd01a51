// hgf_kernels.hpp — hierarchical Gaussian filter (GCV node, BASELINE config 4) on gfx950.
//
// Reference rules replaced (bodies in the un-vendored ReactiveMP.jl):
//   a11 GCV(:y | :x | :z) rules, @marginalrule GCV(:y_x), the GCV average energy (verbatim in-tree at
//       test/inference/inference_tests.jl:594-606) and GaussHermiteCubature moment matching
//       (meta at test/models/statespace/hgf_tests.jl:37-40)
//   a9  NormalMeanVariance prior / transition / observation nodes of the one-step graph (hgf_tests.jl:9-31)
//   the streaming driver's per-observation VMP loop with @autoupdates posterior -> prior feedback
//       (src/inference/streaming.jl:349-407, src/inference/autoupdates.jl:640-659), kept on the device
// One series is sequential in time; series are independent.  A series is owned by a 32-lane half-wave: the
// scalar algebra of an iteration is done redundantly by all lanes, the n_gh ≤ 32 cubature points of the
// z-message are evaluated one per lane (2 exp each) and reduced with half-wave xor shuffles.  4096 series
// fill 2048 wavefronts.  Compute-bound (transcendentals), ≈48 B of HBM traffic per (series, observation).
#pragma once
#include <hip/hip_runtime.h>

#include "gmm_kernels.hpp"

namespace rxhip {

struct HgfParams {
    long long T, n_series;
    const double* y;   // [T][series]
    double *zm, *zv, *xm, *xv;  // [T][series] posteriors after the last iteration of every observation
    double* fe_series;  // [iters][series]  Σ_t FE_t,iter / T
    const double* gh;   // [2][32]: nodes, weights / sqrt(pi)
    double kappa, omega, z_variance, y_variance, z0m, z0v, x0m, x0v;
    int iters, n_gh;
    int* status;
};

__device__ __forceinline__ double half_sum(double v) {
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 32);
    return v;  // identical in all 32 lanes (xor butterfly: fixed order)
}

template <bool FE>
__global__ void __launch_bounds__(64) k_hgf_filter(HgfParams p) {
    const int j = threadIdx.x & 31;
    const long long s = (long long)blockIdx.x * 2 + (threadIdx.x >> 5);
    const bool live = s < p.n_series;
    const long long sc_ = live ? s : 0;
    const double gx = j < p.n_gh ? p.gh[j] : 0.0;
    const double gw = j < p.n_gh ? p.gh[32 + j] : 0.0;
    const double kappa = p.kappa, omega = p.omega, zvar = p.z_variance, yvar = p.y_variance;
    const double A = exp(-omega);
    const double pe = 1.4142135623730951 * gx;  // cubature points against N(0, 1)
    double qzm = p.z0m, qzv = p.z0v, qxm = p.x0m, qxv = p.x0v;
    bool bad = false;
    double yn = p.y[sc_];
    for (long long t = 0; t < p.T; ++t) {
        const double yt = yn;
        if (t + 1 < p.T) yn = p.y[(t + 1) * p.n_series + sc_];
        const double zm = qzm, zv = qzv, xm = qxm, xv = qxv;  // @autoupdates
        const double fzv = zv + zvar, sc = sqrt(2.0 * fzv);
        const double pt = zm + sc * gx;
        for (int n = 0; n < p.iters; ++n) {
            const double B = exp(-kappa * qzm + 0.5 * kappa * kappa * qzv);
            const double g = A * B;
            const double l11 = 1.0 / yvar + g, l22 = 1.0 / xv + g, l12 = -g;
            const double det = l11 * l22 - l12 * l12;
            const double id = 1.0 / det;
            const double v11 = l22 * id, v22 = l11 * id, v12 = -l12 * id;
            const double x1 = yt / yvar, x2 = xm / xv;
            const double m1 = v11 * x1 + v12 * x2, m2 = v12 * x1 + v22 * x2;
            const double psi = (m1 - m2) * (m1 - m2) + v11 + v22 - 2.0 * v12;
            const double b = psi * A;
            // one cubature point per lane
            const double gv = exp(-0.5 * (kappa * pt + b * exp(-kappa * pt)));
            const double cv = gw * gv;
            const double norm = half_sum(cv);
            const double mean = half_sum(pt * cv) / norm;
            const double dv = pt - mean;
            const double var = half_sum(cv * dv * dv) / norm;
            bad = bad || !(det > 0.0) || !(var > 0.0) || !is_finite(mean);
            qzm = mean; qzv = var; qxm = m1; qxv = v11;
            if (FE) {
                const double Bn = exp(-kappa * qzm + 0.5 * kappa * kappa * qzv);
                // the transition node's joint q(zt, zt_min) and the message toward zt_min see the z-message through its
                // Gaussian moments: mean_var(ExponentialLinearQuadratic) = cubature of pdf(z)·exp(z²/2) against N(0, 1)
                const double ecv = gw * exp(-0.5 * (kappa * pe + b * exp(-kappa * pe)) + 0.5 * pe * pe);
                const double en = half_sum(ecv);
                const double em = half_sum(pe * ecv) / en;
                const double ed = pe - em;
                const double ev = half_sum(ecv * ed * ed) / en;
                bad = bad || !(ev > 0.0) || !is_finite(em);
                const double wb = 1.0 / zvar, w00 = 1.0 / ev + wb, w11 = 1.0 / zv + wb;
                const double dW = w00 * w11 - wb * wb;
                const double idw = 1.0 / dW;
                const double s00 = w11 * idw, s11 = w00 * idw, s01 = wb * idw;
                const double j0 = s00 * (em / ev) + s01 * (zm / zv), j1 = s01 * (em / ev) + s11 * (zm / zv);
                const double mu_m = j1, var_m = s11;
                const double e2 = (j0 - j1) * (j0 - j1) + s00 + s11 - 2.0 * s01;
                double F = 0.0;
                F += 0.5 * (kLog2Pi + log(zv) + ((mu_m - zm) * (mu_m - zm) + var_m) / zv);
                F += 0.5 * (kLog2Pi + log(xv) + ((m2 - xm) * (m2 - xm) + v22) / xv);
                F += 0.5 * (kLog2Pi + log(zvar) + e2 / zvar);
                F -= 0.5 * (2.0 * (kLog2Pi + 1.0) - log(dW));
                F += 0.5 * (kLog2Pi + (qzm * kappa + omega) + psi * A * Bn);
                F -= 0.5 * (2.0 * (kLog2Pi + 1.0) + log(v11 * v22 - v12 * v12));
                F += 0.5 * (kLog2Pi + log(yvar) + ((yt - m1) * (yt - m1) + v11) / yvar);
                if (live && j == 0) p.fe_series[(long long)n * p.n_series + s] += F;
            }
        }
        if (live && j == 0) {
            const long long o = t * p.n_series + s;
            p.zm[o] = qzm; p.zv[o] = qzv; p.xm[o] = qxm; p.xv[o] = qxv;
        }
    }
    if (bad && live) atomicOr(p.status, ST_NONFINITE);
}

// per-iteration totals: fe_total[n] = Σ_series fe_series[n][s] / T   (fixed order), also normalises fe_series
__global__ void __launch_bounds__(256) k_hgf_fe(HgfParams p, double* fe_total) {
    __shared__ double sh[256];
    const int n = blockIdx.x;
    double s = 0.0;
    for (long long q = threadIdx.x; q < p.n_series; q += 256) {
        const double v = p.fe_series[(long long)n * p.n_series + q] / (double)p.T;
        p.fe_series[(long long)n * p.n_series + q] = v;
        s += v;
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int wd = 128; wd > 0; wd >>= 1) {
        if ((int)threadIdx.x < wd) sh[threadIdx.x] += sh[threadIdx.x + wd];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        fe_total[n] = sh[0];
        if (!is_finite(sh[0])) atomicOr(p.status, ST_NONFINITE);
    }
}

}  // namespace rxhip
