// gj_inverse_legacy.hpp — the rank-4 sweep inverse of rounds 1–2 (Sweep4 / gj_inverse), kept for scripts/inv_micro.hip only: on badly scaled
// input it loses every digit (profiles/r03/inv_micro.txt), the product uses the 16×16-panel inverse (csrc/dense_kernels.hpp blk_inverse).
#pragma once
#include "../rxinfer.jl_amd/csrc/dense_kernels.hpp"
namespace rxhip {
template <int NT, int Q>
struct Sweep4 {
    static __device__ __forceinline__ void run(Acc<NT>& a, double* rowbuf, int pb, int w, int lane, bool& ok, LogProd& lp) {
        typedef double v4d __attribute__((ext_vector_type(4)));
        constexpr int D = 16 * NT;
        double* rb = rowbuf + ((pb * 4 + Q) & 1) * 4 * D;
        const int jl = lane & 15, vl = lane >> 4;
        const bool rowown = (w == pb) && (vl == Q);
        if (rowown) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                double* dst = rb + (16 * t + jl) * 4;
                reinterpret_cast<double2*>(dst)[0] = make_double2(a.v[t][0], a.v[t][1]);
                reinterpret_cast<double2*>(dst)[1] = make_double2(a.v[t][2], a.v[t][3]);
            }
        }
        lds_barrier();
        // The whole sweep step as ONE rank-4 MFMA per 16×16 tile, A ← A + X·Y with D4⁻¹ on the X side:
        //   X[i][v] = −(R'D4⁻¹)[i][v] + [i = K_u]·D4⁻¹[u][v]      (16 rows of this wave × 4)
        //   Y[v][j] = R[v][j] − [j = K_v]                           (4 × 16 columns of tile t)
        // which yields  A_JJ − R_J'D4⁻¹R_J,  A_KJ = D4⁻¹R_J,  A_JK = (D4⁻¹R_J)'  and  A_KK = 2I − D4⁻¹  (the 2I is removed
        // once, at the end of gj_inverse): −D4⁻¹ on the pivot block, as the sweep operator requires.  The Y operands are plain LDS values that do
        // not wait for D4⁻¹; a thread applies D4⁻¹ to ONE column of R (its row's X entry: four FMAs) instead of one per tile.
        double yb[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) yb[t] = rb[(16 * t + jl) * 4 + vl];   // R[vl][16t + jl]
        double ri[4];                                                       // R[0..3][row i of this lane's X entry]
        {
            const double2* src = reinterpret_cast<const double2*>(rb + (16 * w + jl) * 4);
            const double2 x0 = src[0], x1 = src[1];
            ri[0] = x0.x; ri[1] = x0.y; ri[2] = x1.x; ri[3] = x1.y;
        }
        // pivot block D4[u][v] = R[u][K_v]  (lower triangle), inverse — redundantly in every thread (cheaper than a barrier)
        Sym<4> d4, di;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const double2* src = reinterpret_cast<const double2*>(rb + (16 * pb + Q + 4 * v) * 4);
            const double2 x0 = src[0], x1 = src[1];
            const double col[4] = {x0.x, x0.y, x1.x, x1.y};
#pragma unroll
            for (int u = v; u < 4; ++u) d4(u, v) = col[u];
        }
        double det, idet;
        ok = spd_adj4_cof(d4, di, det, idet) && ok;  // di = adj(D4)
        if (w == 0 && lane == 0) lp.mul(det);
        // column vl of D4⁻¹ (= row vl: symmetric) — this lane's k index in the MFMA operand layout — as adj(D4)·e_vl / det with
        // the unit vector as DATA (loop-invariant registers): FMAs instead of a select tree per entry
        const double e0 = vl == 0 ? 1.0 : 0.0, e1 = vl == 1 ? 1.0 : 0.0, e2 = vl == 2 ? 1.0 : 0.0, e3 = vl == 3 ? 1.0 : 0.0;
        double dv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) dv[u] = (di(u, 0) * e0 + di(u, 1) * e1 + di(u, 2) * e2 + di(u, 3) * e3) * idet;
        const bool pcol = (jl & 3) == Q;  // position jl of tile pb is a pivot index K_c, c = jl >> 2
        const int c = jl >> 2;
        double dvc = dv[0];
        dvc = c == 1 ? dv[1] : dvc;
        dvc = c == 2 ? dv[2] : dvc;
        dvc = c == 3 ? dv[3] : dvc;
        double xa = -(ri[0] * dv[0] + ri[1] * dv[1] + ri[2] * dv[2] + ri[3] * dv[3]);
        xa += (w == pb && pcol) ? dvc : 0.0;                 // row 16w + jl is the pivot row K_c
        const bool ycol = jl == Q + 4 * vl;                  // column jl of tile pb is K_vl
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            v4d acc = {a.v[t][0], a.v[t][1], a.v[t][2], a.v[t][3]};
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, yb[t] - ((ycol && t == pb) ? 1.0 : 0.0), acc, 0, 0, 0);
            a.v[t][0] = acc[0]; a.v[t][1] = acc[1]; a.v[t][2] = acc[2]; a.v[t][3] = acc[3];
        }
        Sweep4<NT, Q + 1>::run(a, rowbuf, pb, w, lane, ok, lp);
    }
};
template <int NT>
struct Sweep4<NT, 4> {
    static __device__ __forceinline__ void run(Acc<NT>&, double*, int, int, int, bool&, LogProd&) {}
};
template <int NT>
__device__ __forceinline__ bool gj_inverse(Acc<NT>& a, double* rowbuf, double* /*unused*/, int w, int lane, LogProd& lp) {
    bool ok = true;
#pragma unroll 1
    for (int pb = 0; pb < NT; ++pb) Sweep4<NT, 0>::run(a, rowbuf, pb, w, lane, ok, lp);
    // The 2I of the pivot blocks (see Sweep4) comes off here, once: a pivot block's diagonal is never read again after its own
    // round (later rounds publish OTHER rows and only add to it), so the correction commutes with every later update — inside
    // the round it sat in the owner wave's publish path, behind the matrix pipe's result latency (11.3 instead of 16.7 µs per
    // 64×64 inverse, scripts/dense_micro.hip).  Diagonal element of this lane: tile t = w, register r with (lane & 15) = (lane >> 4) + 4r.
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) a.v[t][r] = ((t == w && (lane & 15) == (lane >> 4) + 4 * r) ? 2.0 : 0.0) - a.v[t][r];
    lds_barrier();
    return ok;
}

}  // namespace rxhip
