// tu_dense.hip — the MFMA path of ONE tile count NT (state dimension padded to 16·NT) and its launchers (see launch_tables.hpp):
// sweep kernels (dense_kernels.hpp), model tables on the device (dense_tab_kernels.hpp), the masked / per-step-constant schedule
// (dense_mseg_kernels.hpp), the model / data split of shared-model batches (dense_split_kernels.hpp).
// Compiled once per tile count: -DRXHIP_TU_NT=1…4; each object carries its own gfx950 code object.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "launch_tables.hpp"

#ifndef RXHIP_TU_NT
#error "compile with -DRXHIP_TU_NT=1..4"
#endif

namespace rxhip {
namespace {

constexpr int NT = RXHIP_TU_NT;

struct DenseLaunchNT {
    static size_t lds_bytes(int d, int dy) { return DenseLds<NT>::bytes(((d > dy ? d : dy) + 1) & ~1); }
    // The dynamic-LDS ceiling is a per-FUNCTION attribute shared by every engine of the process: it is raised to the
    // hardware limit once, never to one engine's need (a later, smaller engine would otherwise lower it under a live one).
    static hipError_t prepare() {
        const int bytes = 160 * 1024;
        hipError_t e;
        for (const void* f : {(const void*)kd_agg_finish<NT>, (const void*)kd_scan_local<NT, true>, (const void*)kd_scan_local<NT, false>,
                              (const void*)kd_scan_fix<NT>, (const void*)kd_prepare_bnd<NT>, (const void*)kd_forward<NT, true>, (const void*)kd_forward<NT, false>,
                              (const void*)kd_forward_info<NT, true>, (const void*)kd_forward_info<NT, false>,
                              (const void*)kd_forward_info<NT, true, true>, (const void*)kd_forward_info<NT, false, true>,
                              (const void*)kd_backward_info<NT, true>, (const void*)kd_backward_info<NT, false>, (const void*)kd_fe_resid,
                              (const void*)kd_fe_resid_mfma<NT>})
            if ((e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes))) return e;
        return hipSuccess;
    }
    // one launch per slice of at most 32 768 workgroup chains (grid.y / grid.z hold 65 535 blocks)
    template <class F>
    static void slices(const DenseParams& p, long long chains, F launch) {
        for (long long c0 = 0; c0 < chains; c0 += 32768) {
            DenseParams q = p;
            q.chain0 = p.chain0 + c0;
            launch(q, (unsigned)(chains - c0 < 32768 ? chains - c0 : 32768));
        }
    }
    static void prepare_bnd(const DenseParams& p, hipStream_t s) {
        hipLaunchKernelGGL((kd_prepare_bnd<NT>), dim3(p.S), dim3(64 * NT), lds_bytes(p.d, p.dy), s, p);
    }
    static void seg_aggregate(const DenseParams& p, hipStream_t s) {
        const unsigned sb = (unsigned)((p.S - 1 + 15) / 16 + 1);  // blocks of 16 full segments + the last segment on its own
        slices(p, p.n_chains, [&](const DenseParams& q, unsigned nc) {
            hipLaunchKernelGGL((kd_agg_gemm<NT>), dim3(sb, (unsigned)q.agg_kc, nc), dim3(64 * NT), 0, s, q);
            hipLaunchKernelGGL((kd_agg_finish<NT>), dim3(q.S, nc), dim3(64 * NT), DenseLds<NT>::agg_bytes(q.dy), s, q);
        });
    }
    static void boundary_scan(const DenseParams& p, bool fe, hipStream_t s) {
        slices(p, p.n_chains, [&](const DenseParams& q, unsigned nc) {
            dim3 g((q.filter ? 1 : 2) * q.ng, nc);  // filtering runs need the prefix direction only
            dim3 g1(g.x + 1, nc);                   // + the workgroup of the t = 1 update
            if (fe) hipLaunchKernelGGL((kd_scan_local<NT, true>), g1, dim3(64 * NT), lds_bytes(q.d, q.dy), s, q);
            else hipLaunchKernelGGL((kd_scan_local<NT, false>), g1, dim3(64 * NT), lds_bytes(q.d, q.dy), s, q);
            if (q.S > 1) hipLaunchKernelGGL((kd_scan_fix<NT>), g, dim3(64 * NT), lds_bytes(q.d, q.dy), s, q);
        });
    }
    static void forward(const DenseParams& p, bool fe, hipStream_t s) {
        slices(p, p.n_chains, [&](const DenseParams& q, unsigned nc) {
            dim3 g(q.S, nc);
            const size_t lds = DenseLds<NT>::fwd_bytes(((q.d > q.dy ? q.d : q.dy) + 1) & ~1);
            if (fe) hipLaunchKernelGGL((kd_forward<NT, true>), g, dim3(64 * NT), lds, s, q);
            else hipLaunchKernelGGL((kd_forward<NT, false>), g, dim3(64 * NT), lds, s, q);
        });
    }
    // information-form smoother (one inverse per step; free energy at the smoothed means)
    static void forward_info_stepm(const DenseParams& p, bool fe, hipStream_t s, long long chains = -1) {   // per-step constants (masked schedule)
        slices(p, chains < 0 ? p.n_chains : chains, [&](const DenseParams& q, unsigned nc) {
            dim3 g(q.S, nc);
            const size_t lds = DenseLds<NT>::fwd_info_bytes(((q.d > q.dy ? q.d : q.dy) + 1) & ~1);
            if (fe) hipLaunchKernelGGL((kd_forward_info<NT, true, true>), g, dim3(64 * NT), lds, s, q);
            else hipLaunchKernelGGL((kd_forward_info<NT, false, true>), g, dim3(64 * NT), lds, s, q);
        });
    }
    static void forward_info(const DenseParams& p, bool fe, hipStream_t s, long long chains = -1) {
        const size_t lds = DenseLds<NT>::fwd_info_bytes(((p.d > p.dy ? p.d : p.dy) + 1) & ~1);
        slices(p, chains < 0 ? p.n_chains : chains, [&](const DenseParams& q, unsigned nc) {
            dim3 g(q.S, nc);
            if (fe) hipLaunchKernelGGL((kd_forward_info<NT, true>), g, dim3(64 * NT), lds, s, q);
            else hipLaunchKernelGGL((kd_forward_info<NT, false>), g, dim3(64 * NT), lds, s, q);
        });
    }
    static void backward_info(const DenseParams& p, bool fe, hipStream_t s, long long chains = -1) {
        const size_t lds = DenseLds<NT>::bwd_info_bytes(((p.d > p.dy ? p.d : p.dy) + 1) & ~1);
        slices(p, chains < 0 ? p.n_chains : chains, [&](const DenseParams& q, unsigned nc) {
            dim3 g(q.S, nc);
            if (fe) hipLaunchKernelGGL((kd_backward_info<NT, true>), g, dim3(64 * NT), lds, s, q);
            else hipLaunchKernelGGL((kd_backward_info<NT, false>), g, dim3(64 * NT), lds, s, q);
        });
    }
};
// free-energy residual terms of an information-form smoothing run: one workgroup per FR_STEPS steps, partial slots 2S…
static int fe_resid_blocks(long long T, int d, int dy) { const int st = fe_resid_steps(d, dy); return (int)((T + st - 1) / st); }
// passes > 1: per-step constants — one launch per model, each with its own partial slots, the columns of the other models masked
static void launch_fe_resid(const DenseParams& p, hipStream_t s, int passes) {
    static const bool valu_env = hook_env("RXHIP_FE_RESID_VALU") != nullptr;  // the round-2 form (vector FMAs), kept as a cross-check
    const bool valu = valu_env && !p.step_model;
    if (p.step_model && passes > FE_RESID_MAX_PASSES) {   // many models: one step per wavefront instead of one launch per model
        for (long long c0 = 0; c0 < p.n_chains; c0 += 32768) {
            DenseParams q = p;
            q.chain0 = c0;
            const unsigned nc = (unsigned)(p.n_chains - c0 < 32768 ? p.n_chains - c0 : 32768);
            hipLaunchKernelGGL(kd_fe_resid_steps, dim3((unsigned)((p.T + FE_STEPS_BLOCK - 1) / FE_STEPS_BLOCK), nc), dim3(256), 0, s, q, 2 * p.S);
        }
        return;
    }
    for (int pass = 0; pass < passes; ++pass)
    for (long long c0 = 0; c0 < p.n_chains; c0 += 32768) {  // grid.y holds 65 535 blocks
        DenseParams q = p;
        q.chain0 = c0;
        q.model_sel = pass;
        const unsigned nc = (unsigned)(p.n_chains - c0 < 32768 ? p.n_chains - c0 : 32768);
        const dim3 g(fe_resid_blocks(p.T, p.d, p.dy), nc);
        const int slot0 = 2 * p.S + pass * (int)g.x;
        if (valu) { hipLaunchKernelGGL(kd_fe_resid, g, dim3(256), fe_resid_lds_bytes(p.d, p.dy), s, q, slot0); continue; }
        hipLaunchKernelGGL(kd_fe_resid_mfma<NT>, g, dim3(256), fe_resid_mfma_lds_bytes<NT>(p.dy), s, q, slot0);   // p.d == 16 NT: the caller picked this unit by p.d
    }
}
// The per-model tables built on the device (dense_tab_kernels.hpp): the host pads the model (copies only) and launches six small
// kernels; nothing is uploaded but 6 d×d matrices.
static hipError_t tab_prepare() {
    hipError_t err;
    for (const void* f : {(const void*)kt_consts<NT>, (const void*)kt_gains<NT>, (const void*)kt_agg<NT>, (const void*)kt_scan<NT>, (const void*)kt_qtab<NT>})
        if ((err = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))) return err;
    return hipSuccess;
}
static hipError_t launch_dense_tab(const TabParams& tp, hipStream_t s) {   // tab_prepare() first
    constexpr int D = 16 * NT;
    const size_t lds = sizeof(double) * (size_t)(blk_scratch_doubles(NT) + 2 * 64 * NT + 16);
    const size_t lds_c = lds + sizeof(double) * (size_t)D * (D + 1);
    // RXHIP_TRACE=1: device time of every builder kernel (HIP events on the engine's stream; costs a synchronisation — measurement aid)
    static const bool trace = std::getenv("RXHIP_TRACE") != nullptr;
    hipEvent_t ev[7];
    int nev = 0;
    const char* names[6] = {"kt_consts", "kt_gains", "kt_agg", "kt_scan", "kt_qcanon", "kt_qtab"};
    auto mark = [&]() {
        if (trace && hipEventCreate(&ev[nev]) == hipSuccess) { (void)hipEventRecord(ev[nev], s); ++nev; }
    };
    mark();
    hipLaunchKernelGGL((kt_consts<NT>), dim3(1), dim3(64 * NT), lds_c, s, tp);
    mark();
    if (tp.S > 0) {
        hipLaunchKernelGGL((kt_gains<NT>), dim3(1), dim3(64 * NT), lds, s, tp);
        mark();
        hipLaunchKernelGGL((kt_agg<NT>), dim3(2), dim3(64 * NT), lds, s, tp);
        mark();
        hipLaunchKernelGGL((kt_scan<NT>), dim3(2), dim3(64 * NT), lds, s, tp);
        mark();
        hipLaunchKernelGGL(kt_qcanon, dim3(1), dim3(64), 0, s, tp);
        mark();
        if (tp.S > 1) hipLaunchKernelGGL((kt_qtab<NT>), dim3((unsigned)tp.ng, 2), dim3(64 * NT), lds, s, tp);
        mark();
    }
    const hipError_t err = hipGetLastError();
    if (trace && nev > 1) {
        (void)hipEventSynchronize(ev[nev - 1]);
        for (int i = 0; i + 1 < nev; ++i) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            std::fprintf(stderr, "[rxhip]   device time %-10s %8.3f ms\n", names[i], ms);
        }
    }
    for (int i = 0; i < nev; ++i) (void)hipEventDestroy(ev[i]);
    return err;
}
static void tab_consts(const TabParams& tp, unsigned models, size_t lds, hipStream_t s) {   // one workgroup per model (masked schedule)
    hipLaunchKernelGGL((kt_consts<NT>), dim3(models), dim3(64 * NT), lds, s, tp);
}
static hipError_t mseg_prepare_kernels() {
    hipError_t err;
    for (const void* f : {(const void*)km_elements<NT>, (const void*)km_scan<NT>, (const void*)km_group<NT>, (const void*)km_compose<NT>, (const void*)km_apply<NT>, (const void*)km_fold<NT>, (const void*)km_inner<NT>, (const void*)km_bnd<NT>, (const void*)km_filter_out<NT>,
                          (const void*)kt_consts<NT>})
        if ((err = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))) return err;
    return DenseLaunchNT::prepare();
}
// One launch covers at most 32 768 chains (grid.y holds 65 535 blocks; the chains are independent, so a slice runs the whole schedule before
// the next one starts — the sweep kernels slice themselves the same way, DenseLaunchNT::slices).
template <class F>
static void mseg_slices(const MsegParams& mp, const DenseParams& dp, F run) {
    for (long long c0 = 0; c0 < mp.n_chains; c0 += 32768) {
        MsegParams m = mp;
        DenseParams q = dp;
        m.chain0 = c0;
        q.chain0 = dp.chain0 + c0;
        run(m, q, (unsigned)(mp.n_chains - c0 < 32768 ? mp.n_chains - c0 : 32768));
    }
}
static void mseg_launch(const MsegParams& mp_all, const DenseParams& dp_all, bool fe, bool filter, hipStream_t s) {
    const size_t lds = sizeof(double) * (size_t)mseg_lds_doubles(NT, false), lds_s = sizeof(double) * (size_t)mseg_lds_doubles(NT, true);
    (void)hipMemsetAsync(mp_all.nobs, 0, sizeof(double) * (size_t)mp_all.n_chains, s);
    mseg_slices(mp_all, dp_all, [&](const MsegParams& mp, const DenseParams& dp, unsigned nc) {
        hipLaunchKernelGGL(km_mask, dim3((unsigned)std::min<long long>(nc >= 64 ? 16 : 256, (mp.T + 15) / 16), nc), dim3(256), 0, s, mp);
        if (mp.S == 1) hipLaunchKernelGGL(km_gy, dim3((unsigned)(((mp.T - 1) * mp.d + 255) / 256), nc), dim3(256), 0, s, mp);
        else hipLaunchKernelGGL((km_elements<NT>), dim3((unsigned)mp.S, nc), dim3(64 * NT), lds_s, s, mp);
        if (mp.hs) {       // log-depth: all prefix / suffix compositions in ⌈log₂ S⌉ rounds, then every boundary state at once
            const dim3 g1((unsigned)mp.hs_n, nc), g2(2 * (unsigned)mp.hs_n, nc);
            if (mp.hs_g > 1) hipLaunchKernelGGL((km_fold<NT>), g1, dim3(64 * NT), lds_s, s, mp);
            for (int r = 0; r < mp.hs_rounds; ++r) hipLaunchKernelGGL((km_compose<NT>), g2, dim3(64 * NT), lds_s, s, mp, r);
            hipLaunchKernelGGL((km_apply<NT>), g2, dim3(64 * NT), lds_s, s, mp);
            if (mp.hs_g > 1) hipLaunchKernelGGL((km_inner<NT>), g2, dim3(64 * NT), lds_s, s, mp);
        } else if (mp.ng > 0) {   // two levels: group elements, the states at the group edges, then every group on its own
            hipLaunchKernelGGL((km_group<NT>), dim3((unsigned)mp.ng, nc), dim3(64 * NT), lds, s, mp);
            hipLaunchKernelGGL((km_scan<NT>), dim3(2, nc), dim3(64 * NT), lds, s, mp, 2);
            hipLaunchKernelGGL((km_scan<NT>), dim3(2 * (unsigned)mp.ng, nc), dim3(64 * NT), lds, s, mp, 3);
        } else
            hipLaunchKernelGGL((km_scan<NT>), dim3(2, nc), dim3(64 * NT), lds, s, mp, 0);
        if (mp.S > 1) hipLaunchKernelGGL((km_bnd<NT>), dim3((unsigned)(mp.S - 1), nc), dim3(64 * NT), sizeof(double) * blk_scratch_doubles(NT), s, mp);
        if constexpr (NT == 1) if (dp.wave8) {   // one segment per chain, d ≤ 8, one model: the sweep inside a wavefront (dense8_kernels.hpp)
            // two chains per wavefront where the chip is oversubscribed anyway (≥ 4 wavefronts per SIMD: 5.9 against 6.2 ms at 4096 chains × T = 1000);
            // a batch that gives every SIMD at most one wavefront is faster with one chain each (1024 chains: 2.3 against 3.1 ms)
            const bool two = nc % 2 == 0 && nc >= 4096;
            const dim3 g8(two ? nc / 2 : nc);
            if (fe && two) {
                hipLaunchKernelGGL((k8_forward<true, 2>), g8, dim3(64), 0, s, dp);
                hipLaunchKernelGGL((k8_backward<true, 2>), g8, dim3(64), 0, s, dp);
            } else if (fe) {
                hipLaunchKernelGGL((k8_forward<true, 1>), g8, dim3(64), 0, s, dp);
                hipLaunchKernelGGL((k8_backward<true, 1>), g8, dim3(64), 0, s, dp);
            } else if (two) {
                hipLaunchKernelGGL((k8_forward<false, 2>), g8, dim3(64), 0, s, dp);
                hipLaunchKernelGGL((k8_backward<false, 2>), g8, dim3(64), 0, s, dp);
            } else {
                hipLaunchKernelGGL((k8_forward<false, 1>), g8, dim3(64), 0, s, dp);
                hipLaunchKernelGGL((k8_backward<false, 1>), g8, dim3(64), 0, s, dp);
            }
            return;
        }
        if (mp.step_model) {
            if (fe) hipLaunchKernelGGL(km_feconst, dim3(nc), dim3(256), 0, s, mp);   // reads the mask of this slice
            DenseLaunchNT::forward_info_stepm(dp, fe, s, nc);
        } else
            DenseLaunchNT::forward_info(dp, fe, s, nc);
        if (!filter || fe) DenseLaunchNT::backward_info(dp, fe, s, nc);   // a filtering run needs the backward sweep for its free energy only
    });
}
static void mseg_filter_out(const MsegParams& mp_all, const DenseParams& dp_all, hipStream_t s) {
    const size_t ldf = sizeof(double) * (size_t)(DenseCfg<NT>::MAT + 5 * 16 * NT + blk_scratch_doubles(NT) + 16);
    mseg_slices(mp_all, dp_all, [&](const MsegParams& mp, const DenseParams& dp, unsigned nc) {
        hipLaunchKernelGGL((km_filter_out<NT>), dim3((unsigned)mp.T, nc), dim3(64 * NT), ldf, s, mp, dp);
    });
}
static hipError_t split_prepare() {
    hipError_t err;
    for (const void* f : {(const void*)kd_split_forward_lds<16 * NT>, (const void*)kd_split_backward_lds<16 * NT>})
        if ((err = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024))) return err;
    return hipSuccess;
}
static void split_forward(const SplitParams& sq, dim3 grid, hipStream_t s) {
    hipLaunchKernelGGL(kd_split_forward_lds<16 * NT>, grid, dim3(256), split_lds_bytes(16 * NT, 2), s, sq);
}
static void split_backward(const SplitParams& sq, dim3 grid, hipStream_t s) {
    hipLaunchKernelGGL(kd_split_backward_lds<16 * NT>, grid, dim3(256), split_lds_bytes(16 * NT, 1), s, sq);
}
static void cross_from_records(const DenseParams& cp, double* cross, dim3 grid, hipStream_t s) {
    hipLaunchKernelGGL(kd_cross_from_records<NT>, grid, dim3(64 * NT), sizeof(double) * 2 * DenseCfg<NT>::MAT, s, cp, cross);
}
static void forward_info_v(const DenseParams& p, bool fe, hipStream_t s, long long chains) { DenseLaunchNT::forward_info(p, fe, s, chains); }
static void backward_info_v(const DenseParams& p, bool fe, hipStream_t s, long long chains) { DenseLaunchNT::backward_info(p, fe, s, chains); }

}  // namespace

#define RXHIP_CAT_(a, b) a##b
#define RXHIP_CAT(a, b) RXHIP_CAT_(a, b)
const DenseVtbl* RXHIP_CAT(dense_vtbl_nt, RXHIP_TU_NT)() {
    static const DenseVtbl v = {
        NT,
        &DenseLaunchNT::prepare, &DenseLaunchNT::prepare_bnd, &DenseLaunchNT::seg_aggregate, &DenseLaunchNT::boundary_scan, &DenseLaunchNT::forward,
        &forward_info_v, &backward_info_v, &launch_fe_resid,
        &tab_prepare, &launch_dense_tab, &tab_consts,
        &mseg_prepare_kernels, &mseg_launch, &mseg_filter_out,
        &split_prepare, &split_forward, &split_backward, &cross_from_records,
    };
    return &v;
}

}  // namespace rxhip
