// tu_lgssm.hip — the d, dy ≤ 4 state-space kernels of ONE state dimension and their launchers (see launch_tables.hpp).
// Compiled once per state dimension: -DRXHIP_TU_D=1…4; each object carries its own gfx950 code object, loaded by the HIP runtime when an
// engine of that dimension launches its first kernel.
#include <cstring>

#define RXHIP_LAUNCH_LGSSM_ONLY
#include "launch_tables.hpp"

#ifndef RXHIP_TU_D
#error "compile with -DRXHIP_TU_D=1..4"
#endif

namespace rxhip {
namespace {

template <int D, int DY>
struct Launch {
    using CL = CstLayout<D, DY>;
    // `hc`: host copy of model 0's constant block, passed by value when all chains share it
    static CstArg<CL::SIZE> carg(const double* hc) {
        CstArg<CL::SIZE> a;
        std::memcpy(a.v, hc, sizeof(double) * CL::SIZE);
        return a;
    }
    static void seg_aggregate(const Params& p, const double* hc, bool uni, hipStream_t s) {
        const long long total = p.n_chains * (long long)p.S;
        // shared-model batches only (filtering runs, small smoothing runs); per-chain models compute their elements in the lane
        if (uni) hipLaunchKernelGGL((k_seg_aggregate<D, DY, true>), dim3(nblk(total, 64)), dim3(64), 0, s, p, carg(hc));
    }
    static void seg_elements(const Params& p, hipStream_t s) {
        if (!p.masked && !p.step_model && !p.elem_full && p.L >= 384) {   // (short segments end before the recursion settles: the plain kernel is the faster one there)
            hipLaunchKernelGGL((k_seg_elements<D, DY, true>), dim3(nblk(p.n_chains * (long long)p.S, 64)), dim3(64), 0, s, p);
            hipLaunchKernelGGL((k_seg_elements_tail<D, DY>), dim3(nblk(p.n_chains * (long long)p.S, 64)), dim3(64), 0, s, p);
        } else hipLaunchKernelGGL((k_seg_elements<D, DY, false>), dim3(nblk(p.n_chains * (long long)p.S, 64)), dim3(64), 0, s, p);
    }
    static void boundary_scan(const Params& p, const double* hc, bool uni, bool fe, hipStream_t s) {
        dim3 grid(nblk(p.n_chains, 64), p.filter ? 1 : 2);  // a filtering run needs the prefix role only
        if (uni) {
            if (fe) hipLaunchKernelGGL((k_boundary_scan<D, DY, true, true>), grid, dim3(64), 0, s, p, carg(hc));
            else hipLaunchKernelGGL((k_boundary_scan<D, DY, true, false>), grid, dim3(64), 0, s, p, carg(hc));
        } else if (!p.masked && !p.step_model && !p.elem_full) {   // time-invariant per-chain models
            if (fe) hipLaunchKernelGGL((k_boundary_scan<D, DY, false, true, true>), grid, dim3(64), 0, s, p, CstArg<1>{});
            else hipLaunchKernelGGL((k_boundary_scan<D, DY, false, false, true>), grid, dim3(64), 0, s, p, CstArg<1>{});
        } else {
            if (fe) hipLaunchKernelGGL((k_boundary_scan<D, DY, false, true>), grid, dim3(64), 0, s, p, CstArg<1>{});
            else hipLaunchKernelGGL((k_boundary_scan<D, DY, false, false>), grid, dim3(64), 0, s, p, CstArg<1>{});
        }
    }
    static void boundary_scan_tab(const Params& p, const double* hc, bool fe, hipStream_t s) {
        dim3 grid(nblk(p.n_chains, 64), p.filter ? 1 : 2);
        if (fe) hipLaunchKernelGGL((k_boundary_scan_tab<D, DY, true>), grid, dim3(64), 0, s, p, carg(hc));
        else hipLaunchKernelGGL((k_boundary_scan_tab<D, DY, false>), grid, dim3(64), 0, s, p, carg(hc));
    }
    template <bool FILT>
    static void forward_t(const Params& p, const double* hc, bool uni, bool fe, hipStream_t s) {
        const long long total = p.n_chains * (long long)p.S;
        dim3 grid(nblk(total, 64));
        if (uni) {
            if (fe) hipLaunchKernelGGL((k_forward<D, DY, true, true, FILT>), grid, dim3(64), 0, s, p, carg(hc));
            else hipLaunchKernelGGL((k_forward<D, DY, true, false, FILT>), grid, dim3(64), 0, s, p, carg(hc));
        } else if (!FILT && p.tinv_records) {
            if (fe) hipLaunchKernelGGL((k_forward_tinv<D, DY, true>), grid, dim3(64), 0, s, p);
            else hipLaunchKernelGGL((k_forward_tinv<D, DY, false>), grid, dim3(64), 0, s, p);
        } else {
            if (fe) hipLaunchKernelGGL((k_forward<D, DY, false, true, FILT>), grid, dim3(64), 0, s, p, CstArg<1>{});
            else hipLaunchKernelGGL((k_forward<D, DY, false, false, FILT>), grid, dim3(64), 0, s, p, CstArg<1>{});
        }
    }
    static void forward(const Params& p, const double* hc, bool uni, bool fe, hipStream_t s) {
        if (p.filter) forward_t<true>(p, hc, uni, fe, s);
        else forward_t<false>(p, hc, uni, fe, s);
    }
    static void backward(const Params& p, const double* hc, bool uni, hipStream_t s) {
        const long long total = p.n_chains * (long long)p.S;
        dim3 grid(nblk(total, 64));
        if (uni && p.ntab) hipLaunchKernelGGL((k_backward<D, DY, true, true>), grid, dim3(64), 0, s, p, carg(hc));  // one-pass run
        else if (uni) hipLaunchKernelGGL((k_backward<D, DY, true>), grid, dim3(64), 0, s, p, carg(hc));
        else if (p.noise_part && p.tinv_records) hipLaunchKernelGGL((k_backward_noise<D, DY, true>), grid, dim3(64), 0, s, p);
        else if (p.noise_part) hipLaunchKernelGGL((k_backward_noise<D, DY>), grid, dim3(64), 0, s, p);
        else if (p.tinv_records) hipLaunchKernelGGL((k_backward_tinv<D, DY>), grid, dim3(64), 0, s, p);
        else hipLaunchKernelGGL((k_backward<D, DY, false>), grid, dim3(64), 0, s, p, CstArg<1>{});
    }
    static void forward0(const Params& p, const double* hc, bool fe, hipStream_t s) {
        const long long total = p.n_chains * (long long)p.S;
        if (fe) hipLaunchKernelGGL((k_forward0<D, DY, true>), dim3(nblk(total, 64)), dim3(64), 0, s, p, carg(hc));
        else hipLaunchKernelGGL((k_forward0<D, DY, false>), dim3(nblk(total, 64)), dim3(64), 0, s, p, carg(hc));
    }
    static void time_tables(const TimeTabParams& q, hipStream_t s) {
        hipLaunchKernelGGL((k_time_tables<D>), dim3(nblk(q.T, 64)), dim3(64), 0, s, q);
    }
    static void fe_seg(const Params& p, hipStream_t s) {
        hipLaunchKernelGGL((k_fe_seg<D>), dim3(nblk(p.n_chains * (long long)p.S, 64)), dim3(64), 0, s, p);
    }
    static void smooth_tables(const SmoothTabParams& q, const double* hc, hipStream_t s) {
        const long long nblocks = (long long)q.S * smooth_blocks_per_segment(q.L);
        if (q.T > 1) hipLaunchKernelGGL((k_smooth_tab_steps<D, DY>), dim3(nblk(q.T - 1, 64)), dim3(64), 0, s, q, carg(hc));
        hipLaunchKernelGGL((k_smooth_tab_compose<D>), dim3(nblk(nblocks, 64)), dim3(64), 0, s, q);
        hipLaunchKernelGGL((k_smooth_tab_chain<D>), dim3(nblk(q.S, 64)), dim3(64), 0, s, q);
        hipLaunchKernelGGL((k_smooth_tab_apply<D>), dim3(nblk(nblocks, 64)), dim3(64), 0, s, q);
    }
    static void backward_sh(const Params& p, const double* gtab, const double* segend, hipStream_t s) {
        hipLaunchKernelGGL((k_backward_sh<D>), dim3((unsigned)(p.n_chains / 64 * p.S)), dim3(64), 0, s, p, gtab, segend);
    }
    static void forecast(const PredictParams& p, hipStream_t s) {
        hipLaunchKernelGGL((k_forecast<D, DY>), dim3(nblk(p.n_chains, 64)), dim3(64), 0, s, p);
    }
    static void predict(const PredictParams& p, hipStream_t s) {
        const long long total = (p.T + p.H) * p.n_chains;
        const long long nb = (total + 255) / 256;
        hipLaunchKernelGGL((k_predict<D, DY>), dim3((unsigned)(nb < 8192 ? nb : 8192)), dim3(256), 0, s, p);
    }
    static void joint(const PredictParams& p, hipStream_t s) {
        const long long nb = ((p.T - 1) * p.n_chains + 255) / 256;
        hipLaunchKernelGGL((k_joint<D, DY>), dim3((unsigned)(nb < 8192 ? (nb > 0 ? nb : 1) : 8192)), dim3(256), 0, s, p);
    }
    static void small_sweep(const Params& p, const double* hc, bool fe, hipStream_t s) {   // n_chains · S ≤ 256, n_chains ≤ 16, one model
        const size_t lds = sizeof(double) * (size_t)small_sweep_lds_doubles<D, DY>();
        const int par = p.S >= 24 ? 1 : 0;   // log-depth boundary recursion from 24 segments on (below, ⌈log₂ S⌉ rounds cost what S steps do)
        if (p.filter) {
            if (fe) hipLaunchKernelGGL((k_small_sweep<D, DY, true, true>), dim3(1), dim3(SMALL_SWEEP_THREADS), lds, s, p, carg(hc), par);
            else hipLaunchKernelGGL((k_small_sweep<D, DY, false, true>), dim3(1), dim3(SMALL_SWEEP_THREADS), lds, s, p, carg(hc), par);
        } else if (fe) hipLaunchKernelGGL((k_small_sweep<D, DY, true, false>), dim3(1), dim3(SMALL_SWEEP_THREADS), lds, s, p, carg(hc), par);
        else hipLaunchKernelGGL((k_small_sweep<D, DY, false, false>), dim3(1), dim3(SMALL_SWEEP_THREADS), lds, s, p, carg(hc), par);
    }
    static void noise_reset(const NoiseParams& p, hipStream_t s) {
        hipLaunchKernelGGL((k_noise_reset<D, DY>), dim3(nblk(p.n_chains, 64)), dim3(64), 0, s, p);
    }
    static void noise_update(const NoiseParams& p, hipStream_t s) {
        if (!p.moments_in_sweep) hipLaunchKernelGGL((k_noise_moments<D, DY>), dim3(nblk(p.n_chains, 64), (unsigned)p.slices), dim3(64), 0, s, p);
        hipLaunchKernelGGL((k_noise_update<D, DY>), dim3(nblk(p.n_chains, 64)), dim3(64), 0, s, p);
    }
    static void stream_step(const StreamParams& p, hipStream_t s) {
        hipLaunchKernelGGL((k_stream_step<D, DY>), dim3(nblk(p.n_chains, 64)), dim3(64), 0, s, p);
    }
    static LgssmVtbl vtbl() {
        using TL = TabLayout<D, DY>;
        using AL = AggLayout<D>;
        LgssmVtbl v;
        v.d = D; v.dy = DY;
        v.cst_size = CL::SIZE; v.tab_size = TL::SIZE; v.agg_size = AL::SIZE;
        v.oA = CL::A; v.oP = CL::P; v.oLOBS = CL::LOBS; v.oG = CL::G; v.oQI = CL::QI; v.oC0 = CL::C0;
        v.oM1 = CL::M1; v.oV1 = CL::V1; v.oHF = CL::HF;
        v.tK = TL::K; v.tU = TL::U;
        v.aPI = AL::PI; v.aC = AL::C; v.aJ = AL::J; v.aCI = AL::CI; v.aX = AL::X; v.aJJ = AL::JJ;
        using SL = ScanLayout<D>;
        v.scan_size = SL::SIZE; v.sM1 = SL::M1; v.sM2 = SL::M2; v.sVB = SL::VB; v.sN1 = SL::N1; v.sN2 = SL::N2; v.sLB = SL::LB;
        using FL = F0Layout<D, DY>;
        using PL = PosLayout<D>;
        using FS = FeSegLayout<D>;
        v.f0_size = FL::SIZE; v.fK = FL::K; v.fU = FL::U; v.fSI = FL::SI;
        v.pos_size = PL::SIZE; v.pPI = PL::PI; v.pJ = PL::J; v.pC = PL::C;
        v.fs_size = FS::SIZE; v.fsA1 = FS::A1; v.fsA2 = FS::A2; v.fsW = FS::W; v.mt_row = TimeTab<D>::MT;
        v.forward0 = &Launch::forward0;
        v.time_tables = &Launch::time_tables;
        v.fe_seg = &Launch::fe_seg;
        v.gt_row = SmoothTab<D>::SIZE; v.se_size = SegEndTab<D>::SIZE;
        v.smooth_tables = &Launch::smooth_tables;
        v.backward_sh = &Launch::backward_sh;
        v.boundary_scan_tab = &Launch::boundary_scan_tab;
        v.seg_aggregate = &Launch::seg_aggregate;
        v.ex_size = ElemX<D>::SIZE;
        v.seg_elements = &Launch::seg_elements;
        v.boundary_scan = &Launch::boundary_scan;
        v.forward = &Launch::forward;
        v.backward = &Launch::backward;
        v.forecast = &Launch::forecast;
        v.predict = &Launch::predict;
        v.joint = &Launch::joint;
        v.stream_step = &Launch::stream_step;
        v.small_sweep = &Launch::small_sweep;
        v.noise_prior_size = NoisePrior<DY>::SIZE;
        v.noise_reset = &Launch::noise_reset;
        v.noise_update = &Launch::noise_update;
        return v;
    }
};

}  // namespace

#define RXHIP_CAT_(a, b) a##b
#define RXHIP_CAT(a, b) RXHIP_CAT_(a, b)
// every observation dimension 1..4 of this state dimension (the reference is dimension-generic; d = 5 … 64 takes the MFMA path)
void RXHIP_CAT(lgssm_vtbls_d, RXHIP_TU_D)(LgssmVtbl* out) {
    out[0] = Launch<RXHIP_TU_D, 1>::vtbl();
    out[1] = Launch<RXHIP_TU_D, 2>::vtbl();
    out[2] = Launch<RXHIP_TU_D, 3>::vtbl();
    out[3] = Launch<RXHIP_TU_D, 4>::vtbl();
}

}  // namespace rxhip
