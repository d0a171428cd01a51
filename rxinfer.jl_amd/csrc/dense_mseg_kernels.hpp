// dense_mseg_kernels.hpp — `missing` observations on the MFMA path, PARALLEL IN TIME (round 3).
//
// The reference treats `missing` as first class (docs/src/manuals/inference/static.md:98-123, src/inference/batch.jl:221-227): the
// observation node of such a step sends no message.  On the device that makes every covariance of a chain depend on the chain
// and the time index, so none of the per-model tables of the time-parallel schedule survives (dense_kernels.hpp builds its
// segment boundaries from ONE model and a fully observed chain).  Rounds 1–2 ran these engines sequentially in time
// (gseq_kernels.hpp: one workgroup per chain — 686 ms for d = 64, T = 2000, against 0.40 ms for the fully observed chain).
// Here the segment boundaries are built per (chain, segment) ON THE DEVICE, with the mask applied per step, in information form:
//   km_mask      obs[chain][t] = 1 when y[t] of the chain is fully observed (a partly missing vector counts as missing), n_obs per chain
//   km_elements  one workgroup per (chain, segment): the joint information of (state at the segment start, state at its end) that the
//                segment's transitions and observed steps define — precision [[Ĵ, −Ψ′], [−Ψ, Λ]], vector [η̂, ξ]; one inverse and five
//                products per step (the step of kd_forward_info plus three products); B′Q⁻¹y_t of the observed steps goes into the
//                records for the sweep kernel
//   km_compose   the boundary recursion in LOG DEPTH: composition of elements is associative, so all prefix compositions E_0 ∘ … ∘ E_j and all
//                suffix compositions E_j ∘ … ∘ E_{S−1} come out of ⌈log₂ S⌉ rounds of pairwise compositions, 2·S workgroups per round
//   km_apply     every boundary state at once: the belief at t = 0 through the prefix composition in front of a segment (filtered belief at
//                its start), the empty message through the suffix composition behind it (backward message at its end)
//   km_fold,     segments finer than the entries of the rounds: km_fold composes groups of g segments into the entries, km_inner carries the
//   km_inner     boundary states from the group edges to the segments inside (g − 1 steps; all groups at once)
//   km_group,    the same states from SEQUENTIAL recursions (few segments, or RXHIP_MSEG_SCAN=sequential — each kind is the other's checker):
//   km_scan      prefix / suffix steps over the elements, one inverse and two products per element and direction; from 16 segments on in
//                two levels — km_group folds the ≈√S segments of a group into one element, km_scan level 2 runs over the group elements,
//                level 3 inside every group in parallel (level 0: all segments in one run)
//   km_bnd       (Λ_f(b_{s+1}) + Λβ(b_{s+1}))⁻¹ for every inner boundary (parallel)
//   km_gy        one segment per chain (batches that fill the chip on their own): B′Q⁻¹y_t only — no elements, no recursion
// and the sweep itself is kd_forward_info / kd_backward_info with per-chain boundaries (information vector ξ_f at the segment start:
// DenseParams::mseg = 2) and the observation precision B′Q⁻¹B left out of M_{t+1} at missing steps (DenseCst::oPLWM).  The free energy
// is evaluated as on the fully observed path (at the smoothed means), with the observation constants and residuals counted for observed
// steps only.  mseg_setup (rxhip.hip) picks the number of segments and the kind of recursion from a cost model over the measured step times.
// km_elements, km_compose and km_apply are FUSED (mseg_compose_fused, mseg_absorb_fused and the step of km_elements below): the inverse leaves
// its result in the accumulators, every product takes its second operand from one staging matrix in LDS and its first from register fragments
// loaded once, nothing between a kernel's inputs and outputs goes through memory — built from the non-inlined blocks of dense_tab_kernels.hpp
// (km_group / km_scan still are) a round of 450 compositions moved 330 MB, at the memory system's rate.  d = 64, T = 2000, one chain, 10 %
// missing: 0.80 ms per sweep (690 on the sequential schedule of rounds 1–2, 2.5 with the sequential recursion on building blocks; the fully
// observed chain: 0.29).  The algebra and the bookkeeping of the rounds are restated in numpy in tests/test_mseg_information_form.py.
// Per-step constants (desc.step_model): the same kernels with the constant block of model step_model[t] per step (MsegParams::step_model,
// kd_forward_info<…, STEPM>, one residual pass per model, km_feconst).
// Scope: one model per engine, per-step constants shared by all chains, or one model per chain; smoothing runs and filtering runs
// (km_filter_out); the tiles are
// 16·⌈max(d, dy)/16⌉ wide (MsegParams::d).
#pragma once
#include "dense_tab_kernels.hpp"

namespace rxhip {

struct MsegParams {
    int d, dy, dy_user, ptt;   // kernel-level dims (d = 16·NT, dy ≤ d); dy_user: length of an observation vector in y
    long long T, L, n_chains;
    long long chain0;       // first chain of this launch: grid.y holds 65 535 blocks, so a launch covers a slice of at most 32 768 chains
    int S;
    const double* y;        // [T][chain][dy_user]
    const double* in;       // padded model: A | P | V0 | B | Q | m0
    const double* cw;       // workspace of kt_consts (TabWs slots: Q⁻¹, P⁻¹, G, …, V1, V1⁻¹, V_f(1))
    double* ws;             // [chain·S + seg][MSEG_WS] d×d scratch matrices of km_elements; km_scan uses the first 2·chains blocks
    double* obs;            // [chain][T]
    double* nobs;           // [chain]
    double* mel;            // [chain][S][3][d][d]   Λ, Ψ, Ĵ of the segment (information form, km_elements)
    double* mvec;           // [chain][S][2][d]      ξ, η̂
    double* mbnd;           // [chain][S][2][d][d]   Λ_f(b_s) | V_s(b_{s+1})
    double* mlb;            // [chain][S][d][d]      Λβ(b_{s+1})
    double* fstart_m;       // [chain][S][d]         ξ_f(b_s) (the information vector: DenseParams::mseg = 2)
    double* beta_xi;        // [chain][S+1][d]
    double* filt;           // records [chain][T][REC]: slot 1 of the header receives B'Q⁻¹y_t
    int rec;                // REC
    int* status;
    // two-level boundary recursion (km_group + km_scan levels 2 / 3): groups of sg segments, ng groups; 0: one level over all segments
    int sg, ng;
    double* mgrp;           // [chain][ng][3][d][d]  Λ, Ψ, Ĵ of a whole group (km_group)
    double* mgvec;          // [chain][ng][2][d]     ξ, η̂
    // log-depth boundary recursion (km_compose + km_apply): hs_rounds > 0 (or S ≤ 2) selects it; two generations of running compositions
    // per direction — hs [dir][gen parity][chain][S][3][d][d], hsv [dir][gen parity][chain][S][2][d]
    int hs, hs_rounds;
    double *hsel, *hsvec;
    // what the rounds run over: hs_n entries of hs_g segments each (hs_g = 1: the segments themselves; else km_fold composes every group of
    // hs_g segments into mgrp / mgvec first, and km_inner carries the boundary states from the group edges to the segments inside)
    int hs_n, hs_g;
    // per-step constants (desc.step_model; null: one model): time index t uses block step_model[t] of `in`, `cw` and of the constant
    // blocks `cst` (strides in doubles); fe_const[chain]: the data-independent part of the free energy, summed over the chain's steps
    const int* step_model;
    const int* chain_model; // [chain] one model per chain (exclusive with step_model)
    long long in_stride, cw_stride, cst_stride;
    const double* cst;
    double* fe_const;
    int oC0, oLDP;          // DenseCst offsets of dy log 2π + log|Q| and of log|P|, log|V1|
};
// the constant block of (chain, time index): per-step constants shared by all chains, or one model per chain (desc.chain_model), or block 0
__device__ __forceinline__ int mseg_model(const MsegParams& p, long long chain, long long t) {
    if (p.step_model) return p.step_model[t < p.T ? t : p.T - 1];
    return p.chain_model ? p.chain_model[chain] : 0;
}
constexpr int MSEG_WS = 14;
// LDS of the km kernels (doubles): scratch of the inverse | 2·64·NT | 8 vectors | 16, then — kernels with inlined blocks — the staging
// matrix of the fused blocks
__host__ __device__ constexpr int mseg_stage_offset(int NT) { return blk_scratch_doubles(NT) + 2 * 64 * NT + 8 * 16 * NT + 16; }
__host__ __device__ constexpr int mseg_lds_doubles(int NT, bool staged) { return mseg_stage_offset(NT) + (staged ? 16 * NT * tab_stage_ld(NT) : 0); }

// grid (blocks over time, chains); nobs[chain] must be zero on entry (mseg_launch clears it): exact integer counts, any order.
// Sixteen lanes share an observation vector (coalesced reads of y), 16 time steps per 256-thread pass.
static __global__ void __launch_bounds__(256) km_mask(MsegParams p) {
    __shared__ double red[256];
    const long long chain = blockIdx.y + p.chain0;
    const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
    double cnt = 0.0;
    for (long long t0 = (long long)blockIdx.x * 16; t0 < p.T; t0 += (long long)gridDim.x * 16) {
        const long long t = t0 + grp;
        int ok = 1;
        if (t < p.T) {
            const double* yt = p.y + (t * p.n_chains + chain) * p.dy_user;
            for (int k = sub; k < p.dy_user; k += 16) ok &= yt[k] == yt[k] ? 1 : 0;   // NaN = missing
        }
#pragma unroll
        for (int m = 8; m > 0; m >>= 1) ok &= __shfl_xor(ok, m, 16);
        if (t < p.T && sub == 0) {
            p.obs[chain * p.T + t] = ok ? 1.0 : 0.0;
            cnt += ok ? 1.0 : 0.0;
        }
    }
    red[threadIdx.x] = cnt;
    __syncthreads();
    for (int n = blockDim.x >> 1; n > 0; n >>= 1) {
        if ((int)threadIdx.x < n) red[threadIdx.x] += red[threadIdx.x + n];
        __syncthreads();
    }
    if (threadIdx.x == 0 && red[0] != 0.0) atomicAdd(p.nobs + chain, red[0]);   // (sums of small integers: exact in any order)
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

// ---- fused blocks: operands of the short kernels in registers and LDS -----------------------------------------------------------------
// The composition rounds and the element pass run hundreds of workgroups at once, each with a working set of ≈0.5 MB of d×d matrices:
// built from the blocks of dense_tab_kernels.hpp (every intermediate through global memory) a round of 450 compositions moved 330 MB in
// 62 µs — the memory system's rate, not a latency chain.  Here the intermediates stay on the CU: the inverse leaves its result in the
// accumulators, every product takes its second operand from ONE staging matrix in LDS (written from the accumulators of the block
// before), first operands are register fragments loaded once, and the symmetrisations read the staging matrix transposed.
template <int NT, bool TA>
__device__ __forceinline__ void mfrag_load(double (&av)[4 * NT], const double* a, int w, int lane) {   // A-operand fragments of op(a), rows 16w … 16w + 15
    constexpr int D = 16 * NT;
    typedef double v2d __attribute__((ext_vector_type(2)));
    const int i = 16 * w + (lane & 15), kq = lane >> 4;
#pragma unroll
    for (int m = 0; m < 2 * NT; ++m) {
        const int k = 8 * m + 2 * kq;                     // the permuted contraction index of tab_mm
        if (TA) { av[2 * m] = a[k * D + i]; av[2 * m + 1] = a[(k + 1) * D + i]; }
        else { const v2d v = *reinterpret_cast<const v2d*>(a + i * D + k); av[2 * m] = v.x; av[2 * m + 1] = v.y; }
    }
}
template <int NT>
__device__ __forceinline__ void macc_to_stage(const d4 (&acc)[NT], double* stage, int w, int lane) {   // accumulator tiles -> row-major staging matrix
    constexpr int LD = tab_stage_ld(NT);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) stage[acc_row<NT>(w, lane, r) * LD + acc_col<NT>(lane, t)] = acc[t][r];
}
// acc_k += op(a_k) · op(B), B staged row-major (TB: its transpose is the operand)
template <int NT, bool TB, bool TWO>
__device__ __forceinline__ void mstaged_mma(d4 (&acc1)[NT], d4 (&acc2)[NT], const double (&av1)[4 * NT], const double (&av2)[4 * NT], const double* stage, int lane) {
    constexpr int LD = tab_stage_ld(NT);
    typedef double v2d __attribute__((ext_vector_type(2)));
    const int il = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int m = 0; m < 2 * NT; ++m) {
        const int k = 8 * m + 2 * kq;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int j = 16 * t + il;
            double b0v, b1v;
            if (TB) { const v2d v = *reinterpret_cast<const v2d*>(stage + j * LD + k); b0v = v.x; b1v = v.y; }
            else { b0v = stage[k * LD + j]; b1v = stage[(k + 1) * LD + j]; }
            acc1[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av1[2 * m], b0v, acc1[t], 0, 0, 0);
            if (TWO) acc2[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av2[2 * m], b0v, acc2[t], 0, 0, 0);
            acc1[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av1[2 * m + 1], b1v, acc1[t], 0, 0, 0);
            if (TWO) acc2[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av2[2 * m + 1], b1v, acc2[t], 0, 0, 0);
        }
    }
}
template <int NT>
__device__ __forceinline__ void macc_zero(d4 (&acc)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
}
// rd[r] = Σ_j acc[row r of this lane][j] · x[j]: the row sums of a product against a vector in LDS, complete in the lanes with (lane & 15) == 0
template <int NT>
__device__ __forceinline__ void macc_rowdot(double (&rd)[4], const d4 (&acc)[NT], const double* x, int lane) {
    const int il = lane & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) rd[r] = 0.0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const double xj = x[16 * t + il];
#pragma unroll
        for (int r = 0; r < 4; ++r) rd[r] += acc[t][r] * xj;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) rd[r] = row16_sum(rd[r]);
}
// Two segments in a row as one (the formulas of km_group), fused: reads Λ1, Ĵ2, Ψ1, Ψ2, Ĵ1, Λ2 once each, writes Λ, Ψ, Ĵ — 9 matrices of
// traffic where the block version moved 23.  scr: scratch of the inverse, stage: the staging matrix,
// u: D doubles of LDS.
template <int NT>
__device__ __forceinline__ bool mseg_compose_fused(const double* e1_, const double* e2_, const double* v1, const double* v2, double* eo_, double* vo,
                                                   double* scr, double* stage, double* u, int w, int lane_) {
    constexpr int D = 16 * NT, MM = D * D, LD = tab_stage_ld(NT);
    const double *e1 = as_global(e1_), *e2 = as_global(e2_);
    double* eo = as_global(eo_);
    int lane = lane_;
    asm volatile("" : "+v"(lane));
    const int tid = 64 * w + lane, il = lane & 15;
    // T = Λ1 + Ĵ2.  Every Λ of this schedule is symmetric by construction (a symmetrised register tile, written once); Ĵ = Ĵ1 − Ψ1′T⁻¹Ψ1 is
    // symmetric up to the rounding of its products, which ⌈log₂ S⌉ rounds do not amplify (‖T⁻¹Ĵ‖ ≤ 1) — one coalesced read each instead
    // of a second, transposed (uncoalesced) one for a symmetrisation on the way in
    Acc<NT> T;
    {
        const double *L1 = e1, *J2 = e2 + 2 * MM;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = acc_row<NT>(w, lane, r), j = acc_col<NT>(lane, t);
                T.v[t][r] = L1[i * D + j] + J2[i * D + j];
            }
    }
    if (tid < D) u[tid] = v1[tid] + v2[D + tid];                      // ξ1 + η̂2
    LogProd lp;
    const bool ok = blk_inverse<NT>(T, scr, w, lane, lp);             // T⁻¹ in the accumulators (ends with a barrier: u is visible)
    double av1[4 * NT], av2[4 * NT];
    mfrag_load<NT, true>(av1, e1 + MM, w, lane);                      // Ψ1′ and Ψ2: first operands of every product below
    mfrag_load<NT, false>(av2, e2 + MM, w, lane);
    d4 a1[NT], a2[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) a1[t] = (d4){T.v[t][0], T.v[t][1], T.v[t][2], T.v[t][3]};
    macc_to_stage<NT>(a1, stage, w, lane);
    lds_barrier();
    macc_zero<NT>(a1); macc_zero<NT>(a2);
    mstaged_mma<NT, false, true>(a1, a2, av1, av2, stage, lane);      // A′ = Ψ1′T⁻¹,  B = Ψ2 T⁻¹
    {
        double r1[4], r2[4];
        macc_rowdot<NT>(r1, a1, u, lane);
        macc_rowdot<NT>(r2, a2, u, lane);
        if (il == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = acc_row<NT>(w, lane, r);
                vo[D + i] = v1[D + i] + r1[r];                        // η̂ = η̂1 + A′(ξ1 + η̂2)
                vo[i] = v2[i] + r2[r];                                // ξ = ξ2 + B(ξ1 + η̂2)
            }
        }
    }
    double cv[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) cv[t][r] = e1[2 * MM + acc_row<NT>(w, lane, r) * D + acc_col<NT>(lane, t)];   // Ĵ1
    lds_barrier();                                                  // every wave is done with T⁻¹
    macc_to_stage<NT>(a1, stage, w, lane);                            // A′
    lds_barrier();
    d4 jj[NT], pp[NT];
    macc_zero<NT>(jj); macc_zero<NT>(pp);
    mstaged_mma<NT, true, true>(jj, pp, av1, av2, stage, lane);       // Ψ1′A,  Ψ2 A
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = acc_row<NT>(w, lane, r) * D + acc_col<NT>(lane, t);
            eo[2 * MM + o] = cv[t][r] - jj[t][r];                     // Ĵ = Ĵ1 − Ψ1′A
            eo[MM + o] = pp[t][r];                                    // Ψ = Ψ2 A
            cv[t][r] = e2[o];                                         // Λ2 (for the last block)
        }
    lds_barrier();
    macc_to_stage<NT>(a2, stage, w, lane);                            // B
    lds_barrier();
    macc_zero<NT>(jj);
    mstaged_mma<NT, true, false>(jj, pp, av2, av2, stage, lane);      // Ψ2 B′ = Ψ2 T⁻¹Ψ2′
    lds_barrier();
    macc_to_stage<NT>(jj, stage, w, lane);
    lds_barrier();
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = acc_row<NT>(w, lane, r), j = acc_col<NT>(lane, t);
            eo[i * D + j] = cv[t][r] - 0.5 * (jj[t][r] + stage[j * LD + i]);   // Λ = Λ2 − sym(Ψ2 B′)
        }
    return ok;
}

// One boundary step (the formulas of km_scan), fused: with T⁻¹ in the accumulators (T = carried information + the element's corner),
//   TA = false (prefix):  N = Ψ T⁻¹,   out_vec = base_vec + N u,  out = Mbase − sym(Ψ N′)      (Λ_f′ = Λ − ΨT⁻¹Ψ′,  ξ_f′ = ξ + ΨT⁻¹(ξ_f + η̂))
//   TA = true  (suffix):  N = Ψ′T⁻¹,  out_vec = base_vec + N u,  out = Mbase − sym(Ψ′N′)     (Λβ′ = Ĵ − Ψ′T⁻¹Ψ,  ξβ′ = η̂ + Ψ′T⁻¹(ξ + ξβ))
template <int NT, bool TA>
__device__ __forceinline__ void mseg_absorb_fused(const Acc<NT>& Ti, const double* psi_, const double* base_vec, const double* u, double* out_vec,
                                                  const double* Mbase_, double* out_, double* stage, int w, int lane_) {
    constexpr int D = 16 * NT, LD = tab_stage_ld(NT);
    const double *psi = as_global(psi_), *Mbase = as_global(Mbase_);
    double* out = as_global(out_);
    int lane = lane_;
    asm volatile("" : "+v"(lane));
    const int il = lane & 15;
    double av[4 * NT];
    mfrag_load<NT, TA>(av, psi, w, lane);
    d4 n[NT], t2[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) n[t] = (d4){Ti.v[t][0], Ti.v[t][1], Ti.v[t][2], Ti.v[t][3]};
    macc_to_stage<NT>(n, stage, w, lane);
    lds_barrier();
    macc_zero<NT>(n);
    mstaged_mma<NT, false, false>(n, t2, av, av, stage, lane);        // N = op(Ψ) T⁻¹
    {
        double rd[4];
        macc_rowdot<NT>(rd, n, u, lane);
        if (il == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = acc_row<NT>(w, lane, r);
                out_vec[i] = base_vec[i] + rd[r];
            }
        }
    }
    double cv[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) cv[t][r] = Mbase[acc_row<NT>(w, lane, r) * D + acc_col<NT>(lane, t)];
    lds_barrier();
    macc_to_stage<NT>(n, stage, w, lane);
    lds_barrier();
    macc_zero<NT>(t2);
    mstaged_mma<NT, true, false>(t2, n, av, av, stage, lane);         // op(Ψ) N′
    lds_barrier();
    macc_to_stage<NT>(t2, stage, w, lane);
    lds_barrier();
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = acc_row<NT>(w, lane, r), j = acc_col<NT>(lane, t);
            out[i * D + j] = cv[t][r] - 0.5 * (t2[t][r] + stage[j * LD + i]);
        }
}
// The element of a segment in INFORMATION form.  Given the state x_b at the segment start, the segment's transitions and observations
// define a joint Gaussian over (x_b, x_e) with precision [[Ĵ, −Ψ′], [−Ψ, Λ]] and information vector [η̂, ξ] — in the notation of Särkkä &
// García-Fernández (2021): Λ = C⁻¹, Ψ = C⁻¹Π, Ĵ = J + Π′C⁻¹Π, ξ = C⁻¹b, η̂ = η − Π′C⁻¹b.  Moving the end of the segment one step on is
// adding the transition factor and eliminating x_t (a Schur complement), then adding the observation's information:
//     C_t = (Λ + A′P⁻¹A)⁻¹,  Λp = P⁻¹ − K C_t K′ (K = P⁻¹A),  Y = C_t Ψ,  Ψ ← K Y,  Ĵ ← Ĵ − Ψ′Y,  c = C_t ξ,  η̂ ← η̂ + Ψ′c,  ξ ← K c + B′Q⁻¹y,
//     Λ ← Λp + B′Q⁻¹B   (observed steps only)
// — the forward step of kd_forward_info plus three products, ONE inverse and five products per step where the covariance form (the
// first version of this file) needed an inverse and eleven (products that share an operand run as one pass over the staged operand); and the boundary recursions below need exactly (Λ, Ψ, Ĵ, ξ, η̂), nothing else.
// Out of the known start through the first transition: Λp = P⁻¹, Ψ = K, Ĵ = A′P⁻¹A, ξ = η̂ = 0.
template <int NT>
__global__ void __launch_bounds__(64 * NT, 2) km_elements(MsegParams p) {
    constexpr int D = 16 * NT, MM = D * D, LD = tab_stage_ld(NT);
    extern __shared__ __attribute__((aligned(16))) double smem[];
    TabOps<NT> o{(int)threadIdx.x, (int)threadIdx.x >> 6, (int)threadIdx.x & 63, smem};
    double* vec = smem + blk_scratch_doubles(NT) + 2 * 64 * NT;   // ξ (two copies) | η̂ | y | B′Q⁻¹y
    double *xi = vec, *xn = vec + D, *eta = vec + 2 * D, *yv = vec + 3 * D, *gyv = vec + 4 * D;
    double* stage = smem + mseg_stage_offset(NT);
    const int tid = o.tid, dyu = p.dy_user, w = o.w;
    const long long seg = blockIdx.x, chain = blockIdx.y + p.chain0;
    auto CW = [&](int m, int slot) { return as_global(p.cw + (size_t)m * (size_t)p.cw_stride + (size_t)slot * MM); };   // constants of model m
    double* W = p.ws + ((size_t)chain * p.S + seg) * MSEG_WS * MM;
    double* g = as_global(p.mel + ((size_t)chain * p.S + seg) * 3 * MM);   // Λ | Ψ | Ĵ of this segment
    double *Pc = g + MM, *Pn = as_global(W), *Jh = g + 2 * MM;             // Ψ alternates between its slot and a scratch matrix (a step reads the old one while it writes the new one)
    const long long t0 = seg * p.L;   // boundary b_s: state index of the known start
    long long t1 = t0 + p.L;
    if (t1 > p.T - 1) t1 = p.T - 1;
    bool ok = true, ob = true;
    // The running Λp stays in the accumulator registers, symmetrised; Ψ and Ĵ live in global memory (read and written once per step); C, K C,
    // Y′ = Ψ′C and the unsymmetrised Λp pass through the staging matrix.  Step t's transition and observation use the constants of model
    // step_model[t] (one model: block 0).
    double lam[NT][4];
    const double* LOp = nullptr;   // B′Q⁻¹B of the previous step's model when that step was observed
    for (long long t = t0 + 1; t <= t1; ++t) {
        const int a = mseg_model(p, chain, t);
        // (P⁻¹, A′P⁻¹A, B′Q⁻¹B: the exactly symmetric copies kt_consts leaves for this kernel — one coalesced read each)
        const double *PI = CW(a, TabWs::SPINV), *KC = CW(a, TabWs::KC), *WC = CW(a, TabWs::SWC), *LO = CW(a, TabWs::SLOBS), *G = CW(a, TabWs::G);
        ob = p.obs[chain * p.T + t] != 0.0;   // uniform
        double* rec = p.filt + (chain * p.T + t) * p.rec;
        int lane = o.lane;
        asm volatile("" : "+v"(lane));        // (index arithmetic per step, not hoisted into spilled registers)
        const int il = lane & 15;
        if (tid < D) yv[tid] = (ob && tid < dyu) ? p.y[(t * p.n_chains + chain) * dyu + tid] : 0.0;
        lds_barrier();
        if (tid < D) {
            const double gy = ob ? tab_row_dot<D>(G, tid, yv) : 0.0;
            gyv[tid] = gy;
            rec[D + tid] = gy;                                    // B′Q⁻¹y_t for the sweep kernel
        }
        if (t == t0 + 1) {   // out of the known start through the first transition: Λp = P⁻¹, Ψ = K, Ĵ = A′P⁻¹A, ξ = B′Q⁻¹y, η̂ = 0
            o.lin(Pc, 1.0, KC);
            o.lin(Jh, 1.0, WC);
            if (tid < D) { xi[tid] = gyv[tid]; eta[tid] = 0.0; }
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = acc_row<NT>(w, lane, r), j = acc_col<NT>(lane, tt);
                    lam[tt][r] = PI[i * D + j];
                }
            __syncthreads();
            LOp = ob ? LO : nullptr;
            continue;
        }
        Acc<NT> T;   // C = (Λ(t − 1) + A′P⁻¹A)⁻¹,  Λ(t − 1) = Λp [+ B′Q⁻¹B]
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = acc_row<NT>(w, lane, r), j = acc_col<NT>(lane, tt);
                double v = lam[tt][r] + WC[i * D + j];
                if (LOp) v += LOp[i * D + j];
                T.v[tt][r] = v;
            }
        LogProd lp;
        ok = blk_inverse<NT>(T, smem, w, lane, lp) && ok;         // (ends with a barrier: gyv is visible, the staging matrix is free)
        double avk[4 * NT], avp[4 * NT];
        mfrag_load<NT, false>(avk, KC, w, lane);                  // K and Ψ′: first operands of the products below
        mfrag_load<NT, true>(avp, Pc, w, lane);
        d4 gt[NT], yt[NT];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) gt[tt] = (d4){T.v[tt][0], T.v[tt][1], T.v[tt][2], T.v[tt][3]};
        macc_to_stage<NT>(gt, stage, w, lane);
        lds_barrier();
        macc_zero<NT>(gt); macc_zero<NT>(yt);
        mstaged_mma<NT, false, true>(gt, yt, avk, avp, stage, lane);      // K C,  Y′ = Ψ′C
        {   // ξ = K C ξ + B′Q⁻¹y,  η̂ += Ψ′C ξ: row sums of the two products against the old ξ
            double r1[4], r2[4];
            macc_rowdot<NT>(r1, gt, xi, lane);
            macc_rowdot<NT>(r2, yt, xi, lane);
            if (il == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = acc_row<NT>(w, lane, r);
                    xn[i] = r1[r] + gyv[i];
                    eta[i] += r2[r];
                }
            }
        }
        double cv[NT][4];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) cv[tt][r] = Jh[acc_row<NT>(w, lane, r) * D + acc_col<NT>(lane, tt)];
        lds_barrier();                                                  // every wave is done with C
        macc_to_stage<NT>(yt, stage, w, lane);
        lds_barrier();
        d4 jj[NT], pn[NT];
        macc_zero<NT>(jj); macc_zero<NT>(pn);
        mstaged_mma<NT, true, true>(jj, pn, avp, avk, stage, lane);       // Ψ′Y,  K Y
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = acc_row<NT>(w, lane, r) * D + acc_col<NT>(lane, tt);
                Jh[q] = cv[tt][r] - jj[tt][r];                            // Ĵ −= Ψ′Y
                Pn[q] = pn[tt][r];                                        // Ψ = K Y   (into the other copy)
                cv[tt][r] = PI[q];
            }
        lds_barrier();
        macc_to_stage<NT>(gt, stage, w, lane);                            // K C
        lds_barrier();
        macc_zero<NT>(jj);
        mstaged_mma<NT, true, false>(jj, pn, avk, avk, stage, lane);      // K (K C)′ = K C K′
        lds_barrier();
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) jj[tt][r] = cv[tt][r] - jj[tt][r];   // Λp = P⁻¹ − K C K′
        macc_to_stage<NT>(jj, stage, w, lane);
        lds_barrier();
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = acc_row<NT>(w, lane, r), j = acc_col<NT>(lane, tt);
                lam[tt][r] = 0.5 * (jj[tt][r] + stage[j * LD + i]);
            }
        LOp = ob ? LO : nullptr;
        { double* sw = Pc; Pc = Pn; Pn = sw; }
        { double* sw = xi; xi = xn; xn = sw; }
        __syncthreads();                                                  // the staging matrix and yv / gyv are free for the next step
    }
    {   // Λ at the segment end = Λp [+ B′Q⁻¹B]
        const int lane = o.lane;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = acc_row<NT>(w, lane, r), j = acc_col<NT>(lane, tt);
                double v = lam[tt][r];
                if (LOp) v += LOp[i * D + j];
                g[i * D + j] = v;
            }
    }
    __syncthreads();
    if (Pc != g + MM) o.lin(g + MM, 1.0, Pc);
    if (tid < D) {
        double* v = p.mvec + ((size_t)chain * p.S + seg) * 2 * D;
        v[tid] = xi[tid];
        v[D + tid] = eta[tid];
    }
    if (!ok && tid == 0) atomicOr(p.status, ST_NOT_POSDEF);
}

// One segment per chain (batches with enough chains to fill the machine: mseg_setup): no element is needed, only what km_elements hands
// to the sweep kernel besides it — B'Q⁻¹y_t of the observed steps, 0 for the missing ones.  One thread per (chain, t ≥ 1, component).
static __global__ void __launch_bounds__(256) km_gy(MsegParams p) {
    const int D = p.d, dyu = p.dy_user;
    const long long chain = blockIdx.y + p.chain0, idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long t = 1 + idx / D;
    const int i = (int)(idx - (t - 1) * D);
    if (t >= p.T) return;
    const double* G = p.cw + (size_t)mseg_model(p, chain, t) * (size_t)p.cw_stride + (size_t)TabWs::G * D * D + (size_t)i * D;
    const double* yt = p.y + (t * p.n_chains + chain) * dyu;
    double s = 0.0;
    if (p.obs[chain * p.T + t] != 0.0)
        for (int k = 0; k < dyu; ++k) s += G[k] * yt[k];
    p.filt[(chain * p.T + t) * p.rec + D + i] = s;
}

// per-step constants: what the free energy of a chain owes to constants alone, summed over its steps —
// log|V1| + Σ_{t ≥ 1} log|P_t| + Σ_{t observed} (dy log 2π + log|Q_t|).  One workgroup per chain, fixed summation order.
static __global__ void __launch_bounds__(256) km_feconst(MsegParams p) {
    __shared__ double red[256];
    const long long chain = blockIdx.x + p.chain0;
    double acc = 0.0;
    for (long long t = threadIdx.x; t < p.T; t += 256) {
        const double* cm = p.cst + (size_t)mseg_model(p, chain, t) * (size_t)p.cst_stride;
        if (t >= 1) acc += cm[p.oLDP];
        if (p.obs[chain * p.T + t] != 0.0) acc += cm[p.oC0];
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int n = 128; n > 0; n >>= 1) {
        if ((int)threadIdx.x < n) red[threadIdx.x] += red[threadIdx.x + n];
        __syncthreads();
    }
    if (threadIdx.x == 0) p.fe_const[chain] = red[0] + (p.cst + (size_t)mseg_model(p, chain, 0) * (size_t)p.cst_stride)[p.oLDP + 1];
}

// Two segments in a row are one segment: stack the joints of (x_a, x_b) and (x_b, x_c) and eliminate x_b —
//     T = Λ1 + Ĵ2,  Ĵ = Ĵ1 − Ψ1′T⁻¹Ψ1,  Ψ = Ψ2 T⁻¹Ψ1,  Λ = Λ2 − Ψ2 T⁻¹Ψ2′,  η̂ = η̂1 + Ψ1′T⁻¹(ξ1 + η̂2),  ξ = ξ2 + Ψ2 T⁻¹(ξ1 + η̂2)
// (one inverse, five products).  km_group folds the sg segments of a group into one element, in time order; the boundary recursion then
// has two levels (km_scan): over the ng group elements, and — all groups in parallel — over the segments of a group from the state at its
// edge: a sequential depth of 2·sg + ng steps instead of S.
template <int NT>
__global__ void __launch_bounds__(64 * NT) km_group(MsegParams p) {
    constexpr int D = 16 * NT, MM = D * D;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    TabOps<NT> o{(int)threadIdx.x, (int)threadIdx.x >> 6, (int)threadIdx.x & 63, smem};
    double* vec = smem + blk_scratch_doubles(NT) + 2 * 64 * NT;
    double *xi = vec, *eta = vec + D, *u = vec + 2 * D;
    const int tid = o.tid, S = p.S;
    const long long grp = blockIdx.x, chain = blockIdx.y + p.chain0;
    const int s0 = (int)grp * p.sg, s1 = s0 + p.sg < S ? s0 + p.sg : S;
    double* W = p.ws + ((size_t)chain * S + s0) * MSEG_WS * MM;   // the first segment's scratch (km_elements is done with it)
    double *Ti = W, *Am = W + MM, *Bm = W + 2 * MM, *T2 = W + 3 * MM, *Ps2 = W + 4 * MM;
    double* G = p.mgrp + ((size_t)chain * p.ng + grp) * 3 * MM;    // Λ | Ψ | Ĵ of the group
    double *Lc = G, *Pc = G + MM, *Pn = Ps2, *Jc = G + 2 * MM;
    bool ok = true;
    {
        const double* g = p.mel + ((size_t)chain * S + s0) * 3 * MM;
        const double* gv = p.mvec + ((size_t)chain * S + s0) * 2 * D;
        o.lin(Lc, 1.0, g);
        o.lin(Pc, 1.0, g + MM);
        o.lin(Jc, 1.0, g + 2 * MM);
        if (tid < D) { xi[tid] = gv[tid]; eta[tid] = gv[D + tid]; }
        o.sync();
    }
    for (int s = s0 + 1; s < s1; ++s) {
        const double* g = p.mel + ((size_t)chain * S + s) * 3 * MM;     // element 2: Λ2 | Ψ2 | Ĵ2
        const double* gv = p.mvec + ((size_t)chain * S + s) * 2 * D;
        ok = o.inv_symadd(Ti, 1.0, Lc, 1.0, g + 2 * MM) && ok;          // T⁻¹, T = Λ1 + Ĵ2
        if (tid < D) u[tid] = xi[tid] + gv[D + tid];                    // ξ1 + η̂2
        o.template mm<false, false, false>(Am, Ti, Pc);                 // A = T⁻¹Ψ1
        o.template mm<false, true>(Bm, Ti, g + MM);                     // B′ = T⁻¹Ψ2′   (barrier: A is stored, u is visible)
        if (tid < D) {
            eta[tid] += tab_col_dot<D>(Am, tid, u);                     // η̂ = η̂1 + A′(ξ1 + η̂2)
            xi[tid] = gv[tid] + tab_col_dot<D>(Bm, tid, u);             // ξ = ξ2 + Ψ2 T⁻¹(ξ1 + η̂2)   (u was read from the old ξ before the barrier)
        }
        o.template mm<true, false, false>(Jc, Pc, Am, -1.0, Jc, 1.0);   // Ĵ = Ĵ1 − Ψ1′A
        o.template mm<false, false, false>(Pn, g + MM, Am);             // Ψ = Ψ2 A   (into the other copy)
        o.template mm<false, false>(T2, g + MM, Bm);                    // Ψ2 T⁻¹Ψ2′
        o.symadd(Lc, -1.0, T2, 1.0, g);                                 // Λ = Λ2 − sym(B Ψ2′)
        double* sw = Pc; Pc = Pn; Pn = sw;
    }
    if (Pc != G + MM) o.lin(G + MM, 1.0, Pc);
    if (tid < D) {
        double* v = p.mgvec + ((size_t)chain * p.ng + grp) * 2 * D;
        v[tid] = xi[tid];
        v[D + tid] = eta[tid];
    }
    if (!ok && tid == 0) atomicOr(p.status, ST_NOT_POSDEF);
}

// boundary recursion.  Both directions are the same elimination on a joint with the carried information added to one corner:
//   prefix (filtered belief (Λ_f, ξ_f) at a segment start -> at its end):  T = Λ_f + Ĵ,  Λ_f′ = Λ − Ψ T⁻¹Ψ′,  ξ_f′ = ξ + Ψ T⁻¹(ξ_f + η̂)
//   suffix (backward message (Λβ, ξβ) at a segment end -> at its start):    T = Λ + Λβ,   Λβ′ = Ĵ − Ψ′T⁻¹Ψ,  ξβ′ = η̂ + Ψ′T⁻¹(ξ + ξβ)
// — one SPD inverse and two products per element and direction.  The carried matrix lives in the array that hands it to the sweep kernels
// (mbnd / mlb): every step writes the next slot, none copies.
//   level 0: grid (2, chains): every segment of the chain, one after the other (few segments)
//   level 2: grid (2, chains): over the GROUP elements — the states at the group edges
//   level 3: grid (2·ng, chains): blockIdx.x = dir·ng + group: the segments inside a group, from the state at its edge
template <int NT>
__global__ void __launch_bounds__(64 * NT) km_scan(MsegParams p, int level) {
    constexpr int D = 16 * NT, MM = D * D;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    TabOps<NT> o{(int)threadIdx.x, (int)threadIdx.x >> 6, (int)threadIdx.x & 63, smem};
    double* vec = smem + blk_scratch_doubles(NT) + 2 * 64 * NT;
    double *xi = vec, *u = vec + D, *tv = vec + 2 * D;
    const int tid = o.tid, S = p.S, dyu = p.dy_user;
    const int dir = level == 3 ? (int)blockIdx.x / p.ng : (int)blockIdx.x, grp = level == 3 ? (int)blockIdx.x - dir * p.ng : 0;
    const long long chain = blockIdx.y + p.chain0;
    const int m0i = mseg_model(p, chain, 0);   // the prior and the first observation use the constants of step 0's model
    auto CW = [&](int slot) { return p.cw + (size_t)m0i * (size_t)p.cw_stride + (size_t)slot * MM; };
    const double* in0 = p.in + (size_t)m0i * (size_t)p.in_stride;
    // scratch: slots 8 … 13 of a block nobody else touches now — levels 0 / 2: block (2·chain + dir); level 3: the block of the group's first
    // segment, three slots per direction (the two directions of a group run side by side)
    double* W = p.ws + (level == 3 ? ((size_t)chain * S + (size_t)grp * p.sg) * MSEG_WS + 8 + 3 * dir : ((size_t)chain * 2 + dir) * MSEG_WS + 8) * MM;
    double *Wm = W, *N1 = W + MM, *T2 = W + 2 * MM;
    bool ok = true;
    // the run of this workgroup: elements e = 0 … n − 1 in time order, element e ↔ segment (or group) first + e
    const int first = level == 3 ? grp * p.sg : 0;
    const int n = level == 3 ? (first + p.sg < S ? p.sg : S - first) : level == 2 ? p.ng : S;
    const int stride = level == 2 ? p.sg : 1;                     // segments per element
    auto el = [&](int e) { return level == 2 ? p.mgrp + ((size_t)chain * p.ng + e) * 3 * MM : p.mel + ((size_t)chain * S + first + e) * 3 * MM; };
    auto elv = [&](int e) { return level == 2 ? p.mgvec + ((size_t)chain * p.ng + e) * 2 * D : p.mvec + ((size_t)chain * S + first + e) * 2 * D; };
    auto seg_of = [&](int e) { const int sgm = first + e * stride; return sgm < S ? sgm : S; };   // first segment of element e (S: past the end)
    if (dir == 0) {
        // state k = filtered belief at the START of element k: Λ_f -> mbnd[seg_of(k)][0], ξ_f -> fstart_m[seg_of(k)]
        auto slot = [&](int k) { return p.mbnd + ((size_t)chain * S + seg_of(k)) * 2 * MM; };
        if (level != 3) {
            // belief at t = 0: prior ⊗ observation message (if y_0 is observed)
            const bool ob0 = p.obs[chain * p.T] != 0.0;
            const double* m1v = in0 + 5 * MM;   // m0; through the transition when the prior sits on x_0 of the reference's other spelling
            if (tid < D) {
                double mm1 = m1v[tid];
                if (p.ptt) mm1 = tab_row_dot<D>(in0, tid, m1v);           // A m0
                u[tid] = mm1;
                tv[tid] = (ob0 && tid < dyu) ? p.y[(0 * p.n_chains + chain) * dyu + tid] : 0.0;
            }
            o.lin(slot(0), 1.0, CW(TabWs::V1I), ob0 ? 1.0 : 0.0, CW(TabWs::LOBS));   // Λ_f(0) = V1⁻¹ [+ B′Q⁻¹B]   (barriers: u, tv visible)
            if (tid < D) xi[tid] = tab_col_dot<D>(CW(TabWs::V1I), tid, u) + (ob0 ? tab_row_dot<D>(CW(TabWs::G), tid, tv) : 0.0);   // V1⁻¹m1 [+ B′Q⁻¹y]
            o.sync();
        } else {
            if (tid < D) xi[tid] = p.fstart_m[((size_t)chain * S + first) * D + tid];   // level 2 left the state at this group's start
            o.sync();
        }
        for (int k = 0; k < n; ++k) {
            if (tid < D && (level != 3 || k > 0)) p.fstart_m[((size_t)chain * S + seg_of(k)) * D + tid] = xi[tid];   // ξ_f  (DenseParams::mseg = 2)
            if (k == n - 1) break;                                        // the state behind the last element is the next run's (or nobody's)
            const double* g = el(k);
            const double* gv = elv(k);
            ok = o.inv_symadd(Wm, 1.0, slot(k), 1.0, g + 2 * MM) && ok;   // T⁻¹, T = Λ_f + Ĵ
            o.template mm<false, true>(N1, Wm, g + MM);                   // N1′ = T⁻¹Ψ′   (the transpose: its columns are coalesced for the vector below)
            if (tid < D) u[tid] = xi[tid] + gv[D + tid];                  // ξ_f + η̂
            o.template mm<false, false>(T2, g + MM, N1);                  // Ψ T⁻¹Ψ′   (its barrier: u is visible)
            if (tid < D) tv[tid] = gv[tid] + tab_col_dot<D>(N1, tid, u);  // ξ_f′ = ξ + Ψ T⁻¹(ξ_f + η̂)
            o.symadd(slot(k + 1), -1.0, T2, 1.0, g);                      // Λ_f′ = Λ − sym(N1 Ψ′)
            if (tid < D) xi[tid] = tv[tid];
            o.sync();
        }
    } else {
        // state k = backward message at the END of element k: Λβ -> mlb[last segment of element k], ξβ -> beta_xi[that segment + 1]
        auto last = [&](int k) { const int e1 = seg_of(k + 1); return e1 - 1; };
        auto slot = [&](int k) { return p.mlb + ((size_t)chain * S + last(k)) * MM; };
        if (level != 3 || last(n - 1) == S - 1) {
            if (level != 3) { o.eye(slot(n - 1), 0.0); }   // Λβ(b_S) = 0   (level 3 finds it there)
            if (tid < D) xi[tid] = level != 3 ? 0.0 : p.beta_xi[((size_t)chain * (S + 1) + last(n - 1) + 1) * D + tid];
            o.sync();
        } else {
            if (tid < D) xi[tid] = p.beta_xi[((size_t)chain * (S + 1) + last(n - 1) + 1) * D + tid];
            o.sync();
        }
        for (int k = n - 1; k >= 0; --k) {
            if (tid < D && (level != 3 || k < n - 1)) p.beta_xi[((size_t)chain * (S + 1) + last(k) + 1) * D + tid] = xi[tid];
            if (k == 0) break;                                            // the message in front of the first element is the previous run's
            const double* g = el(k);
            const double* gv = elv(k);
            ok = o.inv_symadd(Wm, 1.0, slot(k), 1.0, g) && ok;            // T⁻¹, T = Λβ + Λ
            o.template mm<false, false>(N1, Wm, g + MM);                  // N1′ = T⁻¹Ψ
            if (tid < D) u[tid] = gv[tid] + xi[tid];                      // ξ + ξβ
            o.template mm<true, false>(T2, g + MM, N1);                   // Ψ′T⁻¹Ψ   (its barrier: u is visible)
            if (tid < D) tv[tid] = gv[D + tid] + tab_col_dot<D>(N1, tid, u);   // ξβ′ = η̂ + Ψ′T⁻¹(ξ + ξβ)
            o.symadd(slot(k - 1), -1.0, T2, 1.0, g + 2 * MM);             // Λβ′ = Ĵ − sym(N1 Ψ)
            if (tid < D) xi[tid] = tv[tid];
            o.sync();
        }
    }
    if (!ok && tid == 0) atomicOr(p.status, ST_NOT_POSDEF);
}

// Log-depth boundary recursion (few chains, many segments: the sequential depth of the two-level recursion, 2·√S + √S element steps, is what
// a single long chain waits for).  Composition of elements is associative, so all prefix compositions P[j] = E_0 ∘ … ∘ E_j and all suffix
// compositions Q[j] = E_j ∘ … ∘ E_{S−1} come out of ⌈log₂ S⌉ rounds of pairwise compositions (round r: P[j] ← P[j − 2^r] ∘ P[j],
// Q[j] ← Q[j] ∘ Q[j + 2^r]), 2·S workgroups per round, each the km_group step (one inverse, five products).  An entry at distance i from
// the origin of its scan takes part in rounds 0 … ⌊log₂ i⌋; generation 0 is the element itself (mel / mvec), generation g ≥ 1 lives in buffer
// (g − 1) mod 2: a round reads generation r (and the finished generation of its partner) and writes r + 1 into the other buffer, nobody
// copies.  km_apply then sends the belief at t = 0 through P[s − 1] (filtered belief at the start of segment s) and the empty message
// through Q[s + 1] (backward message at the end of segment s): one step of km_scan each, all segments in parallel.
__device__ __forceinline__ int hs_generations(int i) { return i == 0 ? 0 : 32 - __clz(i); }
__device__ __forceinline__ size_t hs_slot(const MsegParams& p, int dir, int gen, long long chain, int idx) {
    return ((size_t)(dir * 2 + ((gen - 1) & 1)) * (size_t)p.n_chains + (size_t)chain) * (size_t)p.hs_n + (size_t)idx;
}
// generation 0 of entry idx: the segment element, or the composition of its group of segments
__device__ __forceinline__ const double* hs_gen0(const MsegParams& p, long long chain, int idx, int MM) {
    return p.hs_g > 1 ? p.mgrp + ((size_t)chain * p.hs_n + idx) * 3 * MM : p.mel + ((size_t)chain * p.S + idx) * 3 * MM;
}
__device__ __forceinline__ const double* hs_gen0v(const MsegParams& p, long long chain, int idx, int D) {
    return p.hs_g > 1 ? p.mgvec + ((size_t)chain * p.hs_n + idx) * 2 * D : p.mvec + ((size_t)chain * p.S + idx) * 2 * D;
}
template <int NT>
__global__ void __launch_bounds__(64 * NT, 2) km_compose(MsegParams p, int r) {   // ≤ 256 registers: two workgroups per CU at d ≥ 48
    constexpr int D = 16 * NT, MM = D * D;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    TabOps<NT> o{(int)threadIdx.x, (int)threadIdx.x >> 6, (int)threadIdx.x & 63, smem};
    double* u = smem + blk_scratch_doubles(NT) + 2 * 64 * NT;
    const int tid = o.tid, S = p.hs_n;               // entries of the scan
    const int dir = (int)blockIdx.x / S, j = (int)blockIdx.x - dir * S, h = 1 << r;
    const long long chain = blockIdx.y + p.chain0;
    const int i = dir == 0 ? j : S - 1 - j;          // distance from the origin of the scan
    if (i < h || i == S - 1) return;                 // finished — or the whole chain's composition, which nobody reads
    const int jp = dir == 0 ? j - h : j + h, gp = hs_generations(i - h) < r ? hs_generations(i - h) : r;
    auto el = [&](int idx, int gen) { return gen == 0 ? hs_gen0(p, chain, idx, MM) : p.hsel + hs_slot(p, dir, gen, chain, idx) * 3 * MM; };
    auto elv = [&](int idx, int gen) { return gen == 0 ? hs_gen0v(p, chain, idx, D) : p.hsvec + hs_slot(p, dir, gen, chain, idx) * 2 * D; };
    // element 1 comes first in time
    const double *e1 = dir == 0 ? el(jp, gp) : el(j, r), *e2 = dir == 0 ? el(j, r) : el(jp, gp);
    const double *v1 = dir == 0 ? elv(jp, gp) : elv(j, r), *v2 = dir == 0 ? elv(j, r) : elv(jp, gp);
    double* eo = p.hsel + hs_slot(p, dir, r + 1, chain, j) * 3 * MM;
    double* vo = p.hsvec + hs_slot(p, dir, r + 1, chain, j) * 2 * D;
    const bool ok = mseg_compose_fused<NT>(e1, e2, v1, v2, eo, vo, smem, smem + mseg_stage_offset(NT), u, o.w, o.lane);
    if (!ok && tid == 0) atomicOr(p.status, ST_NOT_POSDEF);
}
template <int NT>
__global__ void __launch_bounds__(64 * NT, 2) km_apply(MsegParams p) {
    constexpr int D = 16 * NT, MM = D * D;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    TabOps<NT> o{(int)threadIdx.x, (int)threadIdx.x >> 6, (int)threadIdx.x & 63, smem};
    double* vec = smem + blk_scratch_doubles(NT) + 2 * 64 * NT;
    double *xi = vec, *u = vec + D, *tv = vec + 2 * D;
    const int tid = o.tid, S = p.hs_n, dyu = p.dy_user;                      // S: entries of the scan; entry s = segments hs_g·s … (seg0 … seg1)
    const int dir = (int)blockIdx.x / S, s = (int)blockIdx.x - dir * S;
    const long long chain = blockIdx.y + p.chain0;
    const int seg0 = s * p.hs_g, seg1 = (seg0 + p.hs_g < p.S ? seg0 + p.hs_g : p.S) - 1;
    bool ok = true;
    auto fin = [&](int dr, int idx, const double*& g, const double*& gv) {   // the finished composition of entry idx
        const int gen = hs_generations(dr == 0 ? idx : S - 1 - idx);
        g = gen == 0 ? hs_gen0(p, chain, idx, MM) : p.hsel + hs_slot(p, dr, gen, chain, idx) * 3 * MM;
        gv = gen == 0 ? hs_gen0v(p, chain, idx, D) : p.hsvec + hs_slot(p, dr, gen, chain, idx) * 2 * D;
    };
    if (dir == 0) {
        // belief at t = 0: prior ⊗ observation message (if y_0 is observed) — as km_scan forms it
        const int m0i = mseg_model(p, chain, 0);
        auto CW = [&](int slot) { return p.cw + (size_t)m0i * (size_t)p.cw_stride + (size_t)slot * MM; };
        const double* in0 = p.in + (size_t)m0i * (size_t)p.in_stride;
        const bool ob0 = p.obs[chain * p.T] != 0.0;
        const double* m1v = in0 + 5 * MM;
        if (tid < D) {
            double mm1 = m1v[tid];
            if (p.ptt) mm1 = tab_row_dot<D>(in0, tid, m1v);
            u[tid] = mm1;
            tv[tid] = (ob0 && tid < dyu) ? p.y[(0 * p.n_chains + chain) * dyu + tid] : 0.0;
        }
        o.sync();
        if (tid < D) xi[tid] = tab_col_dot<D>(CW(TabWs::V1I), tid, u) + (ob0 ? tab_row_dot<D>(CW(TabWs::G), tid, tv) : 0.0);
        o.sync();
        double* out = p.mbnd + ((size_t)chain * p.S + seg0) * 2 * MM;
        if (s == 0) {
            o.lin(out, 1.0, CW(TabWs::V1I), ob0 ? 1.0 : 0.0, CW(TabWs::LOBS));
            if (tid < D) p.fstart_m[((size_t)chain * p.S) * D + tid] = xi[tid];
        } else {
            const double *g, *gv;
            fin(0, s - 1, g, gv);
            Acc<NT> T;                                                        // T = Λ_f(0) + Ĵ, symmetrised
            {
                const double *V1 = as_global(CW(TabWs::V1I)), *LO = as_global(CW(TabWs::LOBS)), *J = as_global(g + 2 * MM);
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = acc_row<NT>(o.w, o.lane, r), j = acc_col<NT>(o.lane, t);
                        double v = 0.5 * (V1[i * D + j] + V1[j * D + i]) + 0.5 * (J[i * D + j] + J[j * D + i]);
                        if (ob0) v += 0.5 * (LO[i * D + j] + LO[j * D + i]);
                        T.v[t][r] = v;
                    }
            }
            if (tid < D) u[tid] = xi[tid] + gv[D + tid];                      // ξ_f(0) + η̂   (every reader of the old u is behind a barrier)
            LogProd lp;
            ok = blk_inverse<NT>(T, smem, o.w, o.lane, lp);                   // (ends with a barrier: u is visible)
            mseg_absorb_fused<NT, false>(T, g + MM, gv, u, p.fstart_m + ((size_t)chain * p.S + seg0) * D, g, out, smem + mseg_stage_offset(NT), o.w, o.lane);
        }
    } else {
        double* out = p.mlb + ((size_t)chain * p.S + seg1) * MM;
        double* xo = p.beta_xi + ((size_t)chain * (p.S + 1) + seg1 + 1) * D;
        if (s == S - 1) {
            o.eye(out, 0.0);                                                  // nothing behind the last step
            if (tid < D) xo[tid] = 0.0;
        } else {
            const double *g, *gv;
            fin(1, s + 1, g, gv);
            Acc<NT> T;                                                        // T = Λ (+ Λβ = 0)
            {
                const double* L = as_global(g);
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = acc_row<NT>(o.w, o.lane, r), j = acc_col<NT>(o.lane, t);
                        T.v[t][r] = 0.5 * (L[i * D + j] + L[j * D + i]);
                    }
            }
            if (tid < D) u[tid] = gv[tid];                                    // ξ (+ ξβ = 0)
            LogProd lp;
            ok = blk_inverse<NT>(T, smem, o.w, o.lane, lp);
            mseg_absorb_fused<NT, true>(T, g + MM, gv + D, u, xo, g + 2 * MM, out, smem + mseg_stage_offset(NT), o.w, o.lane);   // ξβ = η̂ + Ψ′T⁻¹ξ,  Λβ = Ĵ − Ψ′T⁻¹Ψ
        }
    }
    if (!ok && tid == 0) atomicOr(p.status, ST_NOT_POSDEF);
}

// Segments finer than the entries of the scan (hs_g > 1): the element pass and the sweep kernels gain from short segments, the rounds pay per
// entry.  km_fold composes the hs_g segments of a group into one entry (hs_g − 1 fused compositions in a row, one workgroup per group);
// km_inner, behind km_apply, carries the filtered belief from the start of a group to the starts of the segments inside it and the backward
// message from its end to their ends (hs_g − 1 fused boundary steps per direction, all groups at once) — the levels of km_group / km_scan,
// with the log-depth rounds in the middle.
template <int NT>
__global__ void __launch_bounds__(64 * NT, 2) km_fold(MsegParams p) {
    constexpr int D = 16 * NT, MM = D * D;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    TabOps<NT> o{(int)threadIdx.x, (int)threadIdx.x >> 6, (int)threadIdx.x & 63, smem};
    double* u = smem + blk_scratch_doubles(NT) + 2 * 64 * NT;
    const int tid = o.tid, S = p.S;
    const long long grp = blockIdx.x, chain = blockIdx.y + p.chain0;
    const int s0 = (int)grp * p.hs_g, s1 = s0 + p.hs_g < S ? s0 + p.hs_g : S;
    double* G = p.mgrp + ((size_t)chain * p.hs_n + grp) * 3 * MM;
    double* Gv = p.mgvec + ((size_t)chain * p.hs_n + grp) * 2 * D;
    auto el = [&](int sgm) { return p.mel + ((size_t)chain * S + sgm) * 3 * MM; };
    auto elv = [&](int sgm) { return p.mvec + ((size_t)chain * S + sgm) * 2 * D; };
    bool ok = true;
    if (s1 - s0 == 1) {   // a ragged last group of one segment
        o.lin(G, 1.0, el(s0));
        o.lin(G + MM, 1.0, el(s0) + MM);
        o.lin(G + 2 * MM, 1.0, el(s0) + 2 * MM);
        if (tid < 2 * D) Gv[tid] = elv(s0)[tid];
        return;
    }
    for (int sgm = s0 + 1; sgm < s1; ++sgm) {
        // running composition ∘ next segment; from the second step on the running one is read from, and written back to, the group's slot
        // (every wave has its fragments of Ψ1 and its tiles of Λ1, Ĵ1 in registers before the first output is stored)
        const double* e1 = sgm == s0 + 1 ? el(s0) : G;
        const double* v1 = sgm == s0 + 1 ? elv(s0) : Gv;
        ok = mseg_compose_fused<NT>(e1, el(sgm), v1, elv(sgm), G, Gv, smem, smem + mseg_stage_offset(NT), u, o.w, o.lane) && ok;
        __syncthreads();
    }
    if (!ok && tid == 0) atomicOr(p.status, ST_NOT_POSDEF);
}
template <int NT>
__global__ void __launch_bounds__(64 * NT, 2) km_inner(MsegParams p) {
    constexpr int D = 16 * NT, MM = D * D;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* u = smem + blk_scratch_doubles(NT) + 2 * 64 * NT;
    double* stage = smem + mseg_stage_offset(NT);
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, S = p.S;
    const int dir = (int)blockIdx.x / p.hs_n, grp = (int)blockIdx.x - dir * p.hs_n;
    const long long chain = blockIdx.y + p.chain0;
    const int s0 = grp * p.hs_g, s1 = s0 + p.hs_g < S ? s0 + p.hs_g : S;   // segments s0 … s1 − 1
    bool ok = true;
    auto load_sum = [&](Acc<NT>& T, const double* a_, const double* b_) {   // T = a + b (both symmetric up to rounding: see mseg_compose_fused)
        const double *a = as_global(a_), *b = as_global(b_);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = acc_row<NT>(w, lane, r) * D + acc_col<NT>(lane, t);
                T.v[t][r] = a[q] + b[q];
            }
    };
    if (dir == 0) {
        for (int sgm = s0; sgm + 1 < s1; ++sgm) {   // filtered belief at the start of sgm -> at the start of sgm + 1
            const double* g = p.mel + ((size_t)chain * S + sgm) * 3 * MM;
            const double* gv = p.mvec + ((size_t)chain * S + sgm) * 2 * D;
            const double* lf = p.mbnd + ((size_t)chain * S + sgm) * 2 * MM;
            const double* xf = p.fstart_m + ((size_t)chain * S + sgm) * D;
            Acc<NT> T;
            load_sum(T, lf, g + 2 * MM);                                      // T = Λ_f + Ĵ
            if (tid < D) u[tid] = xf[tid] + gv[D + tid];                      // ξ_f + η̂
            LogProd lp;
            ok = blk_inverse<NT>(T, smem, w, lane, lp) && ok;
            mseg_absorb_fused<NT, false>(T, g + MM, gv, u, p.fstart_m + ((size_t)chain * S + sgm + 1) * D, g,
                                         p.mbnd + ((size_t)chain * S + sgm + 1) * 2 * MM, stage, w, lane);
            __syncthreads();
        }
    } else {
        for (int sgm = s1 - 1; sgm > s0; --sgm) {   // backward message at the end of sgm -> at the end of sgm − 1
            const double* g = p.mel + ((size_t)chain * S + sgm) * 3 * MM;
            const double* gv = p.mvec + ((size_t)chain * S + sgm) * 2 * D;
            const double* lb = p.mlb + ((size_t)chain * S + sgm) * MM;
            const double* xb = p.beta_xi + ((size_t)chain * (S + 1) + sgm + 1) * D;
            Acc<NT> T;
            load_sum(T, g, lb);                                               // T = Λ + Λβ
            if (tid < D) u[tid] = gv[tid] + xb[tid];                          // ξ + ξβ
            LogProd lp;
            ok = blk_inverse<NT>(T, smem, w, lane, lp) && ok;
            mseg_absorb_fused<NT, true>(T, g + MM, gv + D, u, p.beta_xi + ((size_t)chain * (S + 1) + sgm) * D, g + 2 * MM,
                                        p.mlb + ((size_t)chain * S + sgm - 1) * MM, stage, w, lane);
            __syncthreads();
        }
    }
    if (!ok && tid == 0) atomicOr(p.status, ST_NOT_POSDEF);
}

// Filtering runs (rxhip_run_filter) of masked / per-step engines: the posteriors asked for are q(x_t | y_1..t).  The forward sweep left
// them in its records in information form — record t holds ξ_f(t) and C_t = (Λ_f(t) + A′P⁻¹A)⁻¹ (the owned tiles), the last step's Λ_f(T−1)
// is the `vend` of the last segment — so one workgroup per (time index, chain) turns them into moments: Λ_f = C_t⁻¹ − [A′P⁻¹A]_{t+1},
// V_f = Λ_f⁻¹, m_f = V_f ξ_f.  Two inverses per time index, all in parallel.  (The free energy of such a run is −log p(y) / T: the
// smoothing pipeline's value, scaled by the reduction.)
template <int NT>
__global__ void __launch_bounds__(64 * NT) km_filter_out(MsegParams p, DenseParams q) {
    constexpr int D = 16 * NT;
    using C = DenseCfg<NT>;
    constexpr int LD = C::LD;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Sm = smem;                  // one matrix
    double* xi = Sm + C::MAT;           // ξ_f
    double* red = xi + D;               // [4][D] partial sums
    double* scr = red + 4 * D;          // scratch of the inverse
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const long long t = blockIdx.x, chain = blockIdx.y + p.chain0;
    const double* rec = p.filt + (chain * p.T + t) * p.rec;
    if (tid < D) xi[tid] = rec[tid];
    Acc<NT> a;
    LogProd lp;
    bool ok = true;
    if (t == p.T - 1) {   // Λ_f(T − 1): the end of the last segment
        tri_to_lds<NT>(q.vend + (chain * p.S + (p.S - 1)) * C::TRI, Sm, LD, w, lane);
        __syncthreads();
        acc_load<NT>(a, Sm, LD, w, lane);
        __syncthreads();
    } else {
        // C_t: the tiles every wave owns in the symmetric pairing of the sweep kernels, mirrored through LDS
        constexpr int NS = NT / 2 + 1;
        const int nsw = (NT % 2 == 0 && NT > 1 && w >= NT / 2) ? NS - 1 : NS;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl)
            if (sl < nsw) {
                const int t2 = w + sl >= NT ? w + sl - NT : w + sl;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double v = rec[C::HDR + ((w * NT + t2) * 4 + r) * 64 + lane];
                    const int row = 16 * w + (lane >> 4) + 4 * r, col = 16 * t2 + (lane & 15);
                    Sm[row * LD + col] = v;
                    if (t2 != w) Sm[col * LD + row] = v;
                }
            }
        __syncthreads();
        acc_load<NT>(a, Sm, LD, w, lane);
        __syncthreads();
        ok = blk_inverse<NT>(a, scr, w, lane, lp) && ok;                    // M_t = C_t⁻¹
        const double* cm = p.cst + (size_t)mseg_model(p, chain, t + 1) * (size_t)p.cst_stride;
        acc_add_mat<NT>(a, cm + q.oW_off, D, w, lane, -1.0);                 // Λ_f(t) = M_t − [A′P⁻¹A]_{t+1}
    }
    ok = blk_inverse<NT>(a, scr, w, lane, lp) && ok;                        // V_f = Λ_f⁻¹
    dense_store_cov<NT>(q, a, t, chain, w, lane);
    acc_store<NT>(a, Sm, LD, w, lane);
    __syncthreads();
    {   // m_f = V_f ξ_f: thread group g sums a quarter of the k range (V_f is symmetric: columns)
        const int g = tid / D, i = tid - g * D, k0 = g * (D / 4);
        double sacc = 0.0;
#pragma unroll
        for (int k = 0; k < D / 4; ++k) sacc += Sm[(k0 + k) * LD + i] * xi[k0 + k];
        red[g * D + i] = sacc;
    }
    __syncthreads();
    if (tid < D) dense_store_mean(q, t, chain, tid, (red[tid] + red[D + tid]) + (red[2 * D + tid] + red[3 * D + tid]));
    if (!ok && tid == 0) atomicOr(p.status, ST_NOT_POSDEF);
}

// smoothed covariance at the inner boundaries: V_s(b_{s+1}) = (Λ_f(b_{s+1}) + Λβ(b_{s+1}))⁻¹
template <int NT>
__global__ void __launch_bounds__(64 * NT) km_bnd(MsegParams p) {
    constexpr int D = 16 * NT, MM = D * D;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const long long seg = blockIdx.x, chain = blockIdx.y + p.chain0;
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
