// engine.hpp — what every translation unit behind the C ABI shares: the engine object, the error path, the device guards.
//   rxhip.hip       runtime (pools, arenas), the state-space / mixture / HGF engines and the entry points common to all engines
//   graph_abi.hip   the graph entry points: host-only lowering (graph_lowering.hpp), rxhip_create, the node-array executor's rxhip_tree_* wrappers
//   rccl_abi.hip    the RCCL exchange entry points (rxhip_comm_*, rxhip_allreduce_free_energy, rxhip_gmm_allreduce_statistics)
//   tree_engine.hip the node-array executor itself (behind tree_engine.hpp)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/rxhip.h"
#include "tree_engine.hpp"

namespace rxhip {
struct LgssmVtbl;    // launch_tables.hpp
struct DenseModel;   // dense_kernels.hpp
}
struct DenseTables;  // rxhip.hip: shared per-model tables of the MFMA path
using rxhip::DenseModel;
using rxhip::LgssmVtbl;

// ------------------------------------------------------------------------------------------
// What a handle's OWNER changes after creation — data, runs, modes, counters.  One value-initialised sub-object: rxhip_lgssm_create hands a parked engine
// (engine pool, rxhip.hip) to its next owner by assigning a fresh rxhip_engine_life, so a field added here can never leak from one life to the next.
// Everything that describes the MODEL and its device tables stays in rxhip_engine proper.
struct rxhip_engine_life {
    bool have_data = false;
    long long stream_k = 0;        // rxhip_filter_step: observations seen
    bool have_inputs = false;      // engines with data inputs u[t] (du > 0): rxhip_set_data(RXHIP_VAR_U) has been called
    double stage_ms[4] = {0.0, 0.0, 0.0, 0.0};  // creation stages (rxhip_get_create_stages): tables host | tables device | upload | alloc (zero: nothing was built for this owner)
    bool records_hold_gains = false; // the last run was a smoothing sweep of kd_forward_info / kd_backward_info with one chain per tile: d_filt holds G_t′
    bool records_tinv = false;       // the last smoothing run left mean-only forward records behind the fixed point of V_f (Params::tinv_records)
    bool m_wave8_last = false;       // the last masked sweep ran on the in-wave d ≤ 8 kernels
    int cov_mode = 0;                // rxhip_set_covariance_mode (see below)
    bool cov_pending = false, cov_current = false;
    bool noise_continue = false;     // rxhip_lgssm_noise_continue: runs go on from the current q(W) (iteration-at-a-time drivers)
    bool full_recursions = false;    // rxhip_set_fixed_point_exits(e, 0): every recursion of a sweep in full (no frozen stretches, full records)
    // results bookkeeping
    int last_iterations = 0;
    bool last_want_fe = false;
    bool ran = false, last_filter = false;
    uint64_t rule_calls = 0, products = 0, marginals = 0;
    // per-kernel device times (rxhip_set_profiling)
    double k_ms[RXHIP_K_COUNT] = {};
    uint64_t k_n[RXHIP_K_COUNT] = {};
};
struct rxhip_engine : rxhip_engine_life {
    // one device allocation holds every buffer of a state-space engine (creation / destruction cost two driver calls
    // instead of ≈40: 2.9 ms -> see DESIGN §6c); pointers inside it are never freed individually
    char* arena = nullptr;
    size_t arena_bytes = 0;
    bool in_arena(const void* q) const { return arena && (const char*)q >= arena && (const char*)q < arena + arena_bytes; }
    // description
    int d = 0, dy = 0;
    int dpad = 0;  // dense path: d rounded up to a multiple of 16 (kernel dimension); == d otherwise
    long long T = 0, n_chains = 0;
    long long H = 0;     // time indices without an observation after the T observed ones (rxhip_lgssm_desc.horizon)
    long long Tout() const { return T + H; }  // rows of the posterior / prediction arrays
    std::vector<double> h_bq;  // per model B | Q (row-major): the prediction kernel needs them, the sweep does not
    double* d_bq = nullptr;
    int n_models = 1;
    int ptt = 0;
    int S = 0;
    long long L = 0, Llast = 0;
    bool uniform = true;
    const LgssmVtbl* vt = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // device memory
    double* d_y = nullptr;
    bool own_y = false;
    double *d_filt = nullptr, *d_mean = nullptr, *d_cov = nullptr, *d_cst = nullptr, *d_tab = nullptr,
           *d_vtab = nullptr, *d_scan = nullptr, *d_agg = nullptr, *d_elem = nullptr, *d_fstart = nullptr, *d_beta = nullptr, *d_fe_part = nullptr,
           *d_fe_chain = nullptr, *d_fe_total = nullptr;
    int* d_chain_model = nullptr;
    int* d_status = nullptr;
    double* d_fe_blocks = nullptr;
    // Gaussian-mixture VMP engine (kind == 1)
    int kind = 0;  // 0: LGSSM, 1: GMM, 2: HGF, 3: noise-free drift chain, 5: the level-scheduled node-array executor (tree_engine.hip; everything lives behind `tree`)
    rxhip::tree::Engine* tree = nullptr;
    struct Drift { double m0 = 0, v0 = 1, c = 0, obs_var = 1; } dr;
    struct Hgf {
        rxhip_hgf_desc ds;
        double *d_out = nullptr, *d_fe_series = nullptr, *d_gh = nullptr, *d_fe_total = nullptr;
        int fe_cap = 0;
    } h;
    struct Gmm {
        long long N = 0;
        int K = 0, KT = 0, materialize = 0, nblocks = 0, it = 0, iterations = 0, hist_cap = 0;
        int mvd = 0;          // 0: univariate engine; d ≥ 1: multivariate engine (mvgmm_kernels.hpp) of that dimension
        int nq = 0;           // statistics per iteration (the multi-GPU all-reduce payload)
        int hist_stride = 0;  // doubles of history per iteration
        int state_size = 0;   // doubles of the marginal block (d_par / d_init)
        double *d_resp = nullptr, *d_par = nullptr, *d_drv = nullptr, *d_prior = nullptr, *d_init = nullptr,
               *d_partial = nullptr, *d_totals = nullptr, *d_hist = nullptr, *d_fe = nullptr;
    } g;
    // dense (d = 16·NT) path
    bool dense = false;
    int nt = 0;
    double *d_scanm = nullptr, *d_fstart_m = nullptr, *d_beta_xi = nullptr, *d_vend = nullptr, *d_qtab = nullptr, *d_loc = nullptr, *d_bnd = nullptr;
    const int* d_canon = nullptr;  // canonical indices of the boundary maps (DenseParams::canon), inside the shared table block
    int scan_sg = 1, scan_ng = 1;  // two-level boundary scan of the dense path: group size, groups
    int agg_oc = 1, agg_kc = 1;    // dense aggregation product: offsets per K-chunk, K-chunks
    double* d_aggpart = nullptr;   // [chain][agg_kc][S][2·dpad] partial sums of kd_agg_gemm
    // shared-model smoothing in one pass over the observations (k_forward0): tables, see lgssm_kernels.hpp
    bool fused = false;
    double *d_ftab = nullptr, *d_mtab = nullptr, *d_ntab = nullptr, *d_pos = nullptr, *d_fseg = nullptr;
    double fe_const = 0.0;
    double *d_gtab = nullptr, *d_segend = nullptr, *d_sblk = nullptr;
    hipEvent_t ev_tab0 = nullptr, ev_tab1 = nullptr;  // around the once-per-engine table kernels (rxhip_get_model_tables_ms)  // table-driven backward sweep (k_backward_sh): batches of a multiple of 64 chains
    bool sequential = false;  // no per-position tables: missing observations / per-step constants (per-chain records; the segment
                              // elements are computed in the lane, k_seg_elements, or the chain is ONE segment)
    double* d_elemx = nullptr;
    double* h_io = nullptr;       // pinned staging of rxhip_lgssm_infer (pooled)
    size_t h_io_bytes = 0;
    double* h_stream = nullptr;   // its pinned host staging block
    double* d_stream = nullptr;   // rxhip_filter_step: belief per chain | staging of y, mean, cov, fe (allocated on first use)
    double *d_mu = nullptr, *d_nu = nullptr, *d_cx = nullptr, *d_cy_raw = nullptr;  // known inputs: μ[t] [Tout][d], ν[t] = B μ[t] + d[t] [Tout][dy], c[t], d[t]
    std::vector<double> h_mu, h_nu, h_cx, h_cy, h_offA, h_offB;
    std::vector<double> h_cx_const, h_cy_const;  // the offsets the engine was created with (graph constants), for RXHIP_VAR_U
    std::vector<int> h_offsm;
    int du = 0;                    // graph engines with data inputs `+ B_u * u[t]`: dimension of u, B_u [d][du], which steps have one
    std::vector<double> h_Bu;
    std::vector<char> h_umask;
    bool off_chain = false;        // the device offset arrays carry a chain axis (rxhip_lgssm_set_chain_offsets)
    double* d_off_chain = nullptr; // their block: μ | ν | c | d with a chain axis, and A | B of every model
    std::vector<double> h_user;  // MFMA path: user-level A | P | B | Q | Q⁻¹ of every model (generic_kernels.hpp)
    double* d_user = nullptr;
    int* d_step_model = nullptr;
    // MFMA path, a batch that shares one model: matrices once per engine, vectors per sweep (dense_split_kernels.hpp)
    bool split = false, split_ready = false;
    // rxhip_set_covariance_mode: 0 = every sweep writes the covariance of every chain; 1 = shared-model batches on the split schedule write
    // the per-chain array on request (the values do not depend on the data: one [T][d][d] table per model).  cov_pending: the last run left
    // the array to be materialised; cov_current: the array holds what a materialisation would write
    int m_dpad = 0, m_nt = 0;        // masked schedule: tile dimension 16·⌈max(d, dy)/16⌉ (the engine's own dpad pads d only)
    int m_sg = 0, m_ng = 0;          // masked schedule: groups of the two-level boundary recursion (0: one level)
    int m_models = 1;                // masked schedule: constant blocks (per-step constants: desc.n_models, else 1)
    bool m_stepm = false;            // per-step constants on the masked schedule
    bool m_chainm = false;           // one model per chain on the masked schedule
    DenseModel* m_modtab = nullptr;  // [m_models] constant-block pointers for the sweep kernels (one model per chain)
    double* m_feconst = nullptr;
    double *m_grp = nullptr, *m_gvec = nullptr;
    int m_hs = 0, m_hs_rounds = 0;   // masked schedule: log-depth boundary recursion (km_compose / km_apply)
    int m_hs_n = 0, m_hs_g = 1;      // … over m_hs_n entries of m_hs_g segments each (km_fold / km_inner when m_hs_g > 1)
    double *m_hsel = nullptr, *m_hsvec = nullptr;
    double *d_dtab = nullptr, *d_vlast = nullptr, *d_vstab = nullptr, *d_fe_const = nullptr;
    bool gseq = false;        // d > 4 with `missing` observations or per-step constants: sequential schedule (gseq_kernels.hpp)
    // … and, for `missing` observations under ONE model, the time-parallel schedule of dense_mseg_kernels.hpp for smoothing runs
    bool mseg = false;
    int mS = 0;               // its segments / segment length
    long long mL = 1;
    char* mseg_block = nullptr;   // one allocation: padded model | constants workspace | cst | obs | nobs | elements | boundaries | records | scratch
    double *m_in = nullptr, *m_cw = nullptr, *m_cst = nullptr, *m_obs = nullptr, *m_nobs = nullptr, *m_el = nullptr, *m_vec = nullptr,
           *m_bnd = nullptr, *m_lb = nullptr, *m_ws = nullptr, *m_fe_part = nullptr;
    double* d_prior = nullptr;  // gseq: [n_models][m0 | V0]
    bool masked = false;      // NaN observations are `missing` (rxhip_lgssm_desc.allow_missing): per-chain records, one segment
    int pack = 1;             // 2: pairs of chains share a 16×16 tile as a block-diagonal model (d ≤ 8), see dense_kernels.hpp
    long long wg_chains = 0;  // chains (or pairs) the kernels' grids run over
    int dyk = 0;              // observation dimension at kernel level (2·dy when packed)
    std::vector<struct DenseTables*> dts;  // shared per-model device tables of the MFMA path (d_cst, d_tab, … point into model 0's)
    struct DenseModel* d_models = nullptr;  // [n_models] table pointers on the device (several models per engine)
    std::vector<double> h_cst0;  // model 0's constant block (kernel argument when all chains share it)
    int fe_total_cap = 0;
    // profiling
    bool profiling = false;
    double* h_stage = nullptr;   // pinned staging block of the creation upload (arena_commit), kept until destruction
    size_t h_stage_bytes = 0;
    // unknown observation-noise precision (rxhip_lgssm_noise_create, noise_kernels.hpp): one block B | prior | state | history
    bool noise = false;
    char* noise_block = nullptr;
    double *n_B = nullptr, *n_prior = nullptr, *n_state = nullptr, *n_hist = nullptr, *n_part = nullptr;
    int n_hist_cap = 0, n_slices = 1;
    struct Pending { int k; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
    std::string err = "";
    std::string pool_key;   // non-empty: this engine may be parked by rxhip_destroy and handed out again by rxhip_lgssm_create (engine pool below)
    // scratch of the cross-GPU sums (rxhip_allreduce_free_energy / rxhip_gmm_allreduce_statistics): [nranks][n]
    double* d_coll = nullptr;
    size_t coll_cap = 0;
};

inline rxhip_status fail(rxhip_engine* e, rxhip_status s, const char* fmt, ...) {
    if (e) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        e->err = buf;
    }
    return s;
}
// entry points of the state-space / mixture engines, called on an engine of the node-array executor
#define TREE_GUARD(e) do { if ((e) && (e)->tree) return fail((e), RXHIP_ERR_UNSUPPORTED, "%s: not available on an engine of the node-array executor (rxhip_tree_* entry points)", __func__); } while (0)
// a temporary device block that is freed on every path out of its scope (early HIPCHK returns included)
struct DevTmp {
    void* p = nullptr;
    DevTmp() = default;
    DevTmp(const DevTmp&) = delete;
    DevTmp& operator=(const DevTmp&) = delete;
    ~DevTmp() { if (p) (void)hipFree(p); }
};
#define HIPCHK(e, call)                                                                              \
    do {                                                                                             \
        hipError_t _err = (call);                                                                    \
        if (_err != hipSuccess)                                                                      \
            return fail((e), RXHIP_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_err), \
                        __FILE__, __LINE__);                                                         \
    } while (0)


// Every entry point runs on the engine's device and leaves the CALLER's current device as it found it (a host that drives
// several GPUs from one thread — torch, the Julia shim — must not be left on another device after a destroy / getter).
struct DevGuard {
    int prev = -1;
    bool changed = false;
    hipError_t set(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev == dev) return hipSuccess;
        hipError_t err = hipSetDevice(dev);
        changed = err == hipSuccess;
        return err;
    }
    ~DevGuard() { if (changed && prev >= 0) (void)hipSetDevice(prev); }
};
#define SET_DEVICE(e) DevGuard _dev_guard; HIPCHK((e), _dev_guard.set((e)->device))

// what the translation units of the engines share with the runtime in rxhip.hip: an idle stream of the device (pooled: creating one costs milliseconds on this
// runtime), the HIP-event pair around a kernel of an engine that is being profiled, the HGF engine's run (vmp_engines.hip) as rxhip_run dispatches to it
namespace rxhip {
hipError_t stream_acquire(int device, hipStream_t* out);
rxhip_status prof_begin(rxhip_engine* e, int k);
rxhip_status prof_end(rxhip_engine* e);
rxhip_status hgf_run_async(rxhip_engine* e, int32_t iterations, int32_t want_fe);
bool host_chol_inv(int n, const double* A, double* out, double* logdet);   // host Cholesky inverse + log-determinant of an SPD matrix (rxhip.hip host::chol_inv)
}  // namespace rxhip

