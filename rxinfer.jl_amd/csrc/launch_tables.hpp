// launch_tables.hpp — the host-side seam between the runtime (rxhip.hip) and the kernel translation units.
//
// librxhip.so is linked from several HIP translation units, each with its own gfx950 code object:
//   rxhip.hip       the runtime, the C ABI and the kernels that are not templated on a dimension (mixtures, HGF, drift chain,
//                   sequential d > 4 kernels, layout helpers)
//   tu_lgssm.hip    the d, dy ≤ 4 state-space kernels of ONE state dimension (compiled four times: -DRXHIP_TU_D=1…4)
//   tu_dense.hip    the MFMA path of ONE tile count (compiled four times: -DRXHIP_TU_NT=1…4: d ≤ 16, 32, 48, 64)
// The HIP runtime loads a code object when the first kernel of its translation unit is launched, so an engine pays for the kernels of
// its own dimension only (round 3 shipped ONE object with 674 kernels), and `make -j` builds the units side by side.
// The runtime reaches a unit through a table of plain function pointers; nothing below is templated.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "lgssm_kernels.hpp"
#include "predict_kernels.hpp"
#include "noise_kernels.hpp"
#ifndef RXHIP_LAUNCH_LGSSM_ONLY   // tu_lgssm.hip: the d ≤ 4 units do not parse the MFMA headers
#include "dense_kernels.hpp"
#include "dense_tab_kernels.hpp"
#include "dense_mseg_kernels.hpp"
#include "dense_split_kernels.hpp"
#include "dense8_kernels.hpp"
#endif

// convergence of the boundary-table recursions (host builders in rxhip.hip, device builder in dense_tab_kernels.hpp): two consecutive iterates agree
// entry by entry, |Δ_ij| ≤ TOL · sqrt(a_ii a_jj)
#ifndef RXHIP_TAB_SAME_TOL
#define RXHIP_TAB_SAME_TOL 5e-15   // measured on the BASELINE d = 64 chain (1000 segments of 10 steps that start on the table's fixed point): 1e-13 -> sweep 0.82 ms (the
                                   // segments spend steps converging from a loosely converged boundary), 2e-14 -> 0.65, 5e-15 -> 0.59, 1e-15 -> 0.65 (the recursion never
                                   // repeats that closely: every boundary its own matrix); round 4's max-norm test: 0.56, blind to badly scaled blocks
#endif

namespace rxhip {

// ------------------------------------------------------------------------------------------
// per-(d, dy) dispatch table of the d, dy ≤ 4 kernels
struct LgssmVtbl {
    int d, dy;
    int cst_size, tab_size, agg_size;
    // layout offsets (host table builder writes through these)
    int oA, oP, oLOBS, oG, oQI, oC0, oM1, oV1, oHF;
    int tK, tU;
    int aPI, aC, aJ, aCI, aX, aJJ;
    int scan_size, sM1, sM2, sVB, sN1, sN2, sLB;
    int f0_size, fK, fU, fSI, pos_size, pPI, pJ, pC, fs_size, fsA1, fsA2, fsW, mt_row;  // one-pass schedule (k_forward0)
    void (*forward0)(const Params&, const double*, bool, hipStream_t);
    void (*time_tables)(const TimeTabParams&, hipStream_t);
    void (*fe_seg)(const Params&, hipStream_t);
    int gt_row, se_size;  // SmoothTab / SegEndTab
    void (*smooth_tables)(const SmoothTabParams&, const double*, hipStream_t);
    void (*backward_sh)(const Params&, const double*, const double*, hipStream_t);
    void (*boundary_scan_tab)(const Params&, const double*, bool, hipStream_t);
    void (*seg_aggregate)(const Params&, const double*, bool, hipStream_t);
    int ex_size;  // ElemX
    void (*seg_elements)(const Params&, hipStream_t);
    void (*boundary_scan)(const Params&, const double*, bool, bool, hipStream_t);
    void (*forward)(const Params&, const double*, bool, bool, hipStream_t);  // p.filter selects the filtering variant
    void (*backward)(const Params&, const double*, bool, hipStream_t);
    void (*forecast)(const PredictParams&, hipStream_t);
    void (*predict)(const PredictParams&, hipStream_t);
    void (*joint)(const PredictParams&, hipStream_t);
    void (*stream_step)(const StreamParams&, hipStream_t);
    void (*small_sweep)(const Params&, const double*, bool, hipStream_t);   // the four phases + free energy in ONE launch (k_small_sweep)
    // unknown observation-noise precision (noise_kernels.hpp): q(W) ← initial marginal / the Wishart update after a sweep
    int noise_prior_size;
    void (*noise_reset)(const NoiseParams&, hipStream_t);
    void (*noise_update)(const NoiseParams&, hipStream_t);
};
// tu_lgssm.hip, one definition per state dimension: fills out[0..3] (dy = 1…4)
void lgssm_vtbls_d1(LgssmVtbl* out);
void lgssm_vtbls_d2(LgssmVtbl* out);
void lgssm_vtbls_d3(LgssmVtbl* out);
void lgssm_vtbls_d4(LgssmVtbl* out);

inline unsigned nblk(long long n, int b) { return (unsigned)((n + b - 1) / b); }

// Schedule switches (include/rxhip.h, "Environment"): every RXHIP_* variable that selects a schedule or a checker path is a TEST HOOK and is read
// only when RXHIP_TEST_HOOKS=1 is set as well — the environment of a host process cannot change the schedule behind a result by accident.
inline const char* hook_env(const char* name) {
    const char* on = std::getenv("RXHIP_TEST_HOOKS");
    return (on && on[0] == '1' && on[1] == 0) ? std::getenv(name) : nullptr;
}

#ifndef RXHIP_LAUNCH_LGSSM_ONLY
// ------------------------------------------------------------------------------------------
// per-tile-count dispatch table of the MFMA path (d padded to 16·nt)
constexpr int FE_RESID_MAX_PASSES = 8;
struct DenseVtbl {
    int nt;
    hipError_t (*prepare)();   // dynamic-LDS ceilings of the sweep kernels (per function, per device: call under once_per_device)
    void (*prepare_bnd)(const DenseParams&, hipStream_t);
    void (*seg_aggregate)(const DenseParams&, hipStream_t);
    void (*boundary_scan)(const DenseParams&, bool fe, hipStream_t);
    void (*forward)(const DenseParams&, bool fe, hipStream_t);
    void (*forward_info)(const DenseParams&, bool fe, hipStream_t, long long chains);
    void (*backward_info)(const DenseParams&, bool fe, hipStream_t, long long chains);
    void (*fe_resid)(const DenseParams&, hipStream_t, int passes);
    // model tables on the device (dense_tab_kernels.hpp)
    hipError_t (*tab_prepare)();
    hipError_t (*tab_build)(const TabParams&, hipStream_t);
    void (*tab_consts)(const TabParams&, unsigned models, size_t lds, hipStream_t);
    // `missing` observations / per-step constants, parallel in time (dense_mseg_kernels.hpp)
    hipError_t (*mseg_prepare)();
    void (*mseg_sweep)(const MsegParams&, const DenseParams&, bool fe, bool filter, hipStream_t);
    void (*mseg_filter_out)(const MsegParams&, const DenseParams&, hipStream_t);
    // shared-model batches on the model / data split (dense_split_kernels.hpp)
    hipError_t (*split_prepare)();
    void (*split_forward)(const SplitParams&, dim3 grid, hipStream_t);
    void (*split_backward)(const SplitParams&, dim3 grid, hipStream_t);
    void (*cross_from_records)(const DenseParams&, double* cross, dim3 grid, hipStream_t);
};
const DenseVtbl* dense_vtbl_nt1();
const DenseVtbl* dense_vtbl_nt2();
const DenseVtbl* dense_vtbl_nt3();
const DenseVtbl* dense_vtbl_nt4();
#endif

}  // namespace rxhip
