// hgf_kernels.hpp — hierarchical Gaussian filter (GCV node, BASELINE config 4) on gfx950.
//
// Reference rules replaced (bodies in the un-vendored ReactiveMP.jl):
//   a11 GCV(:y | :x | :z) rules, @marginalrule GCV(:y_x), the GCV average energy (verbatim in-tree at
//       test/inference/inference_tests.jl:594-606) and GaussHermiteCubature moment matching
//       (meta at test/models/statespace/hgf_tests.jl:37-40)
//   a9  NormalMeanVariance prior / transition / observation nodes of the one-step graph (hgf_tests.jl:9-31)
//   the streaming driver's per-observation VMP loop with @autoupdates posterior -> prior feedback
//       (src/inference/streaming.jl:349-407, src/inference/autoupdates.jl:640-659), kept on the device
// One series is sequential in time; series are independent.  A series is owned by a 16-lane DPP row: the
// scalar algebra of an iteration is done redundantly by its lanes, the n_gh ≤ 32 cubature points of the
// z-message are evaluated two per lane (2 exp each) and reduced with row-local DPP adds.  4096 series
// fill 1024 wavefronts (one per SIMD).  Compute-bound (transcendentals), ≈48 B of HBM traffic per (series, observation).
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

// ---- 16-lane sums without the LDS crossbar: four DPP stages inside a 16-lane row.  The cubature sums are the latency
// chain of an iteration; ds_bpermute shuffles cost ≈100 cycles per stage, DPP moves a few.  Fixed order -> deterministic;
// every lane of the row gets the sum.
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    // every lane of these permutations has a source lane inside its row: no `old` value is needed (bound_ctrl form, one
    // v_mov_b32_dpp per half instead of a copy + a DPP move)
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int N>
__device__ __forceinline__ void row_sums(double (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_mov<0xB1>(v[i]);   // quad_perm [1,0,3,2]
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_mov<0x4E>(v[i]);   // quad_perm [2,3,0,1]
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_mov<0x141>(v[i]);  // row_half_mirror
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_mov<0x140>(v[i]);  // row_mirror
}

// One series per 16-lane DPP row (four series per wavefront): the scalar algebra of an iteration is shared by 16 lanes
// instead of 32, every lane carries two of the n_gh ≤ 32 cubature points, and the reductions stay inside a row (no
// cross-row permute).  Measured against the 32-lane form: ≈half the VALU instructions per (series, iteration).
constexpr int HGF_LANES = 16, HGF_SERIES_PER_WAVE = 64 / HGF_LANES;
template <bool FE>
__global__ void __launch_bounds__(64) k_hgf_filter(HgfParams p) {
    const int j = threadIdx.x & (HGF_LANES - 1);
    const long long s = (long long)blockIdx.x * HGF_SERIES_PER_WAVE + (threadIdx.x / HGF_LANES);
    const bool live = s < p.n_series;
    const long long sc_ = live ? s : 0;
    double gx[2], gw[2], pe[2], hpe2[2], epe[2], lpe[2];
    const double kappa = p.kappa, omega = p.omega, zvar = p.z_variance, yvar = p.y_variance;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int q = j + HGF_LANES * u;
        gx[u] = q < p.n_gh ? p.gh[q] : 0.0;
        gw[u] = q < p.n_gh ? p.gh[32 + q] : 0.0;
        pe[u] = 1.4142135623730951 * gx[u];  // cubature points against N(0, 1)
        hpe2[u] = 0.5 * pe[u] * pe[u];
        epe[u] = exp_bounded(-kappa * pe[u]);
        lpe[u] = -0.5 * kappa * pe[u] + hpe2[u];
    }
    const double A = exp_bounded(-omega);
    const double iyvar = 1.0 / yvar, wb = 1.0 / zvar, lzvar = log(zvar), lyvar = log(yvar);
    double qzm = p.z0m, qzv = p.z0v, qxm = p.x0m, qxv = p.x0v;
    bool bad = false;
    double yn = p.y[sc_];
    double fe_acc = 0.0;  // lane n of the row accumulates the free energy of VMP iteration n (n < 16; beyond: global)
    double Bcur = exp_bounded(-kappa * qzm + 0.5 * kappa * kappa * qzv);  // exp_bounded(−κ E z + ½κ² var z) of the current q(z)
    for (long long t = 0; t < p.T; ++t) {
        const double yt = yn;
        if (t + 1 < p.T) yn = p.y[(t + 1) * p.n_series + sc_];
        const double zm = qzm, zv = qzv, xm = qxm, xv = qxv;  // @autoupdates
        const double fzv = zv + zvar, sc = sqrt(2.0 * fzv);
        const double ixv = rcp_pos(xv), izv = rcp_pos(zv);
        const double x1 = yt * iyvar, x2 = xm * ixv, zx = zm * izv;
        // point-wise constants of the two cubatures: exp_bounded(−½κ·point) factors out of the iteration loop
        double dx[2], ept[2], lpt[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            dx[u] = sc * gx[u];  // cubature points against the forward message N(zm, fzv)
            const double pt = zm + dx[u];
            ept[u] = exp_bounded(-kappa * pt);
            lpt[u] = -0.5 * kappa * pt;
        }
        double fe_t_const = 0.0;
        if (FE) fe_t_const = 0.5 * (kLog2Pi + log(zv)) + 0.5 * (kLog2Pi + log(xv)) + 0.5 * (kLog2Pi + lzvar) + 0.5 * (kLog2Pi + lyvar);
        // One VMP iteration = a dependent chain (B → joint (x, x_min) → b → cubature → reductions → new q(z)) followed by the
        // free-energy terms of the iteration, which nothing downstream waits for.  The loop is software-pipelined by hand:
        // the free energy of iteration n − 1 is evaluated inside iteration n, so its ≈250 instructions fill the latency gaps
        // of the chain instead of extending it.  exp_bounded(−κ E z + ½κ² var z) of the new q(z) is both the GCV energy's factor of
        // this iteration and B of the next one: computed once (Bcur is carried across iterations and observations).
        struct FeIn { double Bn, ev, em, m1, m2, v11, v22, psi, det, qzm; };
        auto chain = [&](FeIn& f) {
            const double B = Bcur;
            const double g = A * B;
            const double l11 = iyvar + g, l22 = ixv + g;
            const double det = l11 * l22 - g * g;
            const double id = rcp_pos(det);
            const double v11 = l22 * id, v22 = l11 * id, v12 = g * id;
            const double m1 = v11 * x1 + v12 * x2, m2 = v12 * x1 + v22 * x2;
            const double psi = (m1 - m2) * (m1 - m2) + v11 + v22 - 2.0 * v12;
            const double b = psi * A;
            // two cubature points per lane: z-message pdf exp_bounded(−½(κz + b·exp(−κz))) at the forward-message points and (for
            // the free energy) times exp_bounded(z²/2) at the N(0,1) points; first moments taken about zm / 0
            // Two reduction rounds, exactly the reference's approximate_meancov: (norm, first moment), then the second
            // moment about the mean — when a message's mode leaves the cubature range the variance is pure rounding
            // residue and only the same formula reproduces the reference there.
            double r[FE ? 4 : 2];
            double cv[2], ecv[2] = {0.0, 0.0};
#pragma unroll
            for (int u = 0; u < 2; ++u) cv[u] = gw[u] * exp_bounded(lpt[u] - 0.5 * b * ept[u]);
            r[0] = cv[0] + cv[1];
            r[1] = cv[0] * dx[0] + cv[1] * dx[1];
            if (FE) {
#pragma unroll
                for (int u = 0; u < 2; ++u) ecv[u] = gw[u] * exp_bounded(lpe[u] - 0.5 * b * epe[u]);
                r[2] = ecv[0] + ecv[1];
                r[3] = ecv[0] * pe[0] + ecv[1] * pe[1];
            }
            row_sums(r);
            const double in0 = rcp_pos(r[0]);
            const double dmean = r[1] * in0;
            const double mean = zm + dmean;
            double em = 0.0;
            double ie0 = 0.0;
            if (FE) {
                ie0 = rcp_pos(r[2]);
                em = r[3] * ie0;
            }
            double q[FE ? 2 : 1];
            q[0] = cv[0] * (dx[0] - dmean) * (dx[0] - dmean) + cv[1] * (dx[1] - dmean) * (dx[1] - dmean);
            if (FE) q[1] = ecv[0] * (pe[0] - em) * (pe[0] - em) + ecv[1] * (pe[1] - em) * (pe[1] - em);
            row_sums(q);
            const double var = q[0] * in0;
            bad = bad || !(det > 0.0) || !(var > 0.0) || !is_finite(mean);
            qzm = mean; qzv = var; qxm = m1; qxv = v11;
            Bcur = exp_bounded(-kappa * qzm + 0.5 * kappa * kappa * qzv);
            if (FE) {
                f.Bn = Bcur; f.ev = q[1] * ie0; f.em = em; f.m1 = m1; f.m2 = m2; f.v11 = v11; f.v22 = v22; f.psi = psi; f.det = det;
                f.qzm = mean;
            }
        };
        auto fe_eval = [&](const FeIn& f, int n) {
            // the transition node's joint q(zt, zt_min) and the message toward zt_min see the z-message through its
            // Gaussian moments: mean_var(ExponentialLinearQuadratic) = cubature of pdf(z)·exp(z²/2) against N(0, 1)
            bad = bad || !(f.ev > 0.0) || !is_finite(f.em);
            const double iev = rcp_pos(f.ev);
            const double w00 = iev + wb, w11 = izv + wb;
            const double dW = w00 * w11 - wb * wb;
            const double idw = rcp_pos(dW);
            const double s00 = w11 * idw, s11 = w00 * idw, s01 = wb * idw;
            const double xo = f.em * iev;
            const double j0 = s00 * xo + s01 * zx, j1 = s01 * xo + s11 * zx;
            const double e2 = (j0 - j1) * (j0 - j1) + s00 + s11 - 2.0 * s01;
            double F = fe_t_const;
            F += 0.5 * ((j1 - zm) * (j1 - zm) + s11) * izv;                      // prior zt_min
            F += 0.5 * ((f.m2 - xm) * (f.m2 - xm) + f.v22) * ixv;                // prior xt_min
            F += 0.5 * e2 * wb;                                                  // transition
            F += 0.5 * (kLog2Pi + (f.qzm * kappa + omega) + f.psi * A * f.Bn);   // GCV average energy
            // −H[zt, zt_min] − H[xt, xt_min] = −2(log 2π + 1) + ½ log(dW / det Σ_x),  det Σ_x = 1 / det
            F += -2.0 * (kLog2Pi + 1.0) + 0.5 * log(dW * f.det);
            F += 0.5 * ((yt - f.m1) * (yt - f.m1) + f.v11) * iyvar;              // observation
            if (n < HGF_LANES) fe_acc += (j == n) ? F : 0.0;
            else if (live && j == 0) p.fe_series[(long long)n * p.n_series + s] += F;
        };
        FeIn pend, cur;
        chain(pend);
        for (int n = 1; n < p.iters; ++n) {
            chain(cur);
            if (FE) fe_eval(pend, n - 1);
            pend = cur;
        }
        if (FE) fe_eval(pend, p.iters - 1);
        if (live && j == 0) {
            const long long o = t * p.n_series + s;
            p.zm[o] = qzm; p.zv[o] = qzv; p.xm[o] = qxm; p.xv[o] = qxv;
        }
    }
    if (FE && live && j < p.iters) p.fe_series[(long long)j * p.n_series + s] += fe_acc;
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
