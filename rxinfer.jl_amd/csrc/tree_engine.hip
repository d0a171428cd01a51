// tree_engine.hip — host side of the level-scheduled node-array executor (include/rxhip.h "The level-scheduled node-array executor").
//
// compile() — tree_compiler.hpp:  rxhip_graph_desc (struct-of-arrays dump of a materialised GraphPPL model)  ->  Program
//   1. classify the variables: constant, data, precision (random `out` of a Wishart / Gamma prior), derived-clamped (output of a deterministic node
//      whose inputs are all clamped: a PointMass message in the reference), Gaussian;
//   2. check the family (node types, constant third interfaces, dimensions) and that the Gaussian variables form a forest (union-find);
//   3. one message per (factor, Gaussian interface) and direction; structural analysis in dependency order (Kahn): which messages are uniform
//      (an unobserved leaf sends nothing), which variable → factor messages are aliases (degree 2), the form each message is stored in, its level;
//   4. ops sorted by level: rules, products, marginals (the sum-product sweep), then the Bethe terms / residual moments, the q(W) updates and the
//      fixed-order sums of the free energy;
//   5. slots of the replica-fastest device arrays, the constant pool (matrices, Σ | W | log|W| blocks inverted once on the host, priors).
// run(): per iteration either one launch per level over all (op, replica) items, or ONE launch in which a workgroup walks all levels for its
// replicas (deep narrow graphs: a chain is three levels per time step and a launch per level would cost more than the level).
#include "tree_engine.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <unordered_map>
#include <vector>

#include "graph_lowering.hpp"  // last_asymmetry
#include "launch_tables.hpp"   // hook_env
#include "tree_kernels.hpp"
#include "tree_wave.hpp"
#include "tree_compiler.hpp"   // Program, Compiler: the graph compiler (host only)

namespace rxhip {
namespace tree {

struct Engine {
    Program prog;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    long long R = 1, RS = 16;
    int mode = 0, mode_fe = 0, rb = 16, rb_fe = 16, wg = 256;   // schedule of the sweep phase / of the Bethe phase (tree_kernels.hpp: 0 a launch per level, 1 resident levels, 2 walk)
    int *d_ops = nullptr, *d_aux = nullptr, *d_lvl = nullptr, *d_status = nullptr, *d_sops = nullptr, *d_strands = nullptr;
    double *d_cpool = nullptr, *d_msg = nullptr, *d_marg = nullptr, *d_val = nullptr, *d_prec = nullptr, *d_term = nullptr, *d_stat = nullptr, *d_prec_init = nullptr, *d_marg_init = nullptr,
           *d_fe_rep = nullptr, *d_fe_hist = nullptr, *d_fe_part = nullptr;
    int fe_cap = 0;
    bool have_data = false, ran = false;
    bool cont = false;   // rxhip_tree_continue: later runs go on from the q(W) the previous run ended with
    bool allow_missing = false;   // created with rxhip_graph_desc.allow_missing: NaN in the data is `missing`
    // the launch-per-level schedule as a HIP graph: an iteration's launches (hundreds of short kernels at small batches — 5 µs each, most of it the launch) are captured
    // once per (free energy wanted, last level) and replayed with ONE hipGraphLaunch per iteration
    bool use_graph = false;
    hipGraphExec_t gexec[2] = {nullptr, nullptr};
    int g_lend[2] = {-1, -1};
    bool tiled = false;       // the wavefront-per-item kernels run this engine (dimensions above 8; 5 … 8: see create)
    bool elem_fast = false;   // … and with them: a replica's slots contiguous (TreeParams es = 1), so that a wavefront's loads of a message coalesce
    bool push_done = false;   // the image marginals (OP_MARG_PUSH, first level of the second phase) are those of the last sweep
    int last_iterations = 0, last_want_fe = 0;
    std::vector<char> data_set;
    uint64_t runs = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double last_iteration_ms = 0.0;
};

namespace {
struct DevScope {
    int prev = -1;
    bool changed = false;
    explicit DevScope(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) changed = hipSetDevice(dev) == hipSuccess;
    }
    ~DevScope() { if (changed && prev >= 0) (void)hipSetDevice(prev); }
};
#define TCHK(call)                                                                                                   \
    do {                                                                                                             \
        hipError_t _e = (call);                                                                                      \
        if (_e != hipSuccess) { err = std::string(#call) + " failed: " + hipGetErrorString(_e); return RXHIP_ERR_HIP; } \
    } while (0)

template <class T>
rxhip_status upload(T** dst, const std::vector<T>& src, std::string& err) {
    TCHK(hipMalloc(dst, sizeof(T) * std::max<size_t>(1, src.size())));
    if (!src.empty()) TCHK(hipMemcpy(*dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice));
    return RXHIP_OK;
}
rxhip_status zalloc(double** dst, long long doubles, std::string& err) {
    const size_t n = (size_t)std::max<long long>(1, doubles);
    TCHK(hipMalloc(dst, sizeof(double) * n));
    TCHK(hipMemset(*dst, 0, sizeof(double) * n));
    return RXHIP_OK;
}
// per-replica free energy = term[root]; total over replicas in a fixed order: workgroup b sums replicas [b·FE_CHUNK, (b + 1)·FE_CHUNK) (fixed stride per thread,
// pairwise tree over a fixed layout) into partial[b]; the last stage (one workgroup) sums the partials the same way — deterministic to the bit, and not one
// workgroup crawling over 65 536 values (87 µs of a 4.6 ms iteration)
constexpr int FE_CHUNK = 4096;
__global__ void __launch_bounds__(256) k_tree_fe_total(const double* __restrict__ src, long long stride, long long n, double* __restrict__ per_replica, double* __restrict__ out) {
    __shared__ double sh[256];
    const long long lo = (long long)blockIdx.x * FE_CHUNK, hi = lo + FE_CHUNK < n ? lo + FE_CHUNK : n;
    double s = 0.0;
    for (long long r = lo + threadIdx.x; r < hi; r += 256) {
        const double v = src[r * stride];
        if (per_replica) per_replica[r] = v;
        s += v;
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) sh[threadIdx.x] += sh[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = sh[0];
}

// state[k][r] = init[k] for every replica: a run starts from the `@initialization` marginals
__global__ void __launch_bounds__(256) k_tree_broadcast(double* __restrict__ dst, const double* __restrict__ init, long long n, long long RS, int elem_fast) {
    const long long total = n * RS;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) dst[i] = init[elem_fast ? i % n : i / RS];
}

// host layout <-> replica-fastest device layout, on the device (at 65 536 replicas the host loops these replace ran for seconds)
// data: dst[(k)·RS + r] = src[r·rows + col + k] for k < width — a 32×32 LDS tile so that both sides move whole lines
__global__ void __launch_bounds__(256) k_tree_scatter(double* __restrict__ dst, const double* __restrict__ src, long long R, long long es, long long rs, long long rows, long long col,
                                                      long long width, int* __restrict__ status) {
    __shared__ double tile[32][33];
    const long long r0 = (long long)blockIdx.x * 32, k0 = (long long)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 × 8
    for (int j = ty; j < 32; j += 8) {
        const long long r = r0 + j, k = k0 + tx;
        tile[j][tx] = (r < R && k < width) ? src[r * rows + col + k] : 0.0;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const long long k = k0 + j, r = r0 + tx;
        if (k < width && r < R) {
            const double x = tile[tx][j];
            dst[k * es + r * rs] = x;
            if (!(fabs(x) <= 1.79769313486231570815e308)) atomicOr(status, 2);   // NaN (`missing`) or ±Inf in the data: this executor has no rule for it
        }
    }
}
struct GatherVar { int off, d; long long mo, co; int clamped; };   // clamped: a data / derived value — reported as a point mass (mean = the value, zero covariance)
// marginals of the listed variables into [variable][replica][d] / [variable][replica][d][d] (the arrays rxhip_tree_get_marginals fills)
__global__ void __launch_bounds__(256) k_tree_gather(const double* __restrict__ marg, const double* __restrict__ val, const GatherVar* __restrict__ vars, int n_vars, long long R, long long es,
                                                     long long rs_marg, long long rs_val, double* __restrict__ mean, double* __restrict__ cov) {
    const long long rblocks = (R + 255) / 256;
    for (long long it = blockIdx.x; it < (long long)n_vars * rblocks; it += gridDim.x) {
        const int vi = (int)(it / rblocks);
        const long long r = (it - (long long)vi * rblocks) * 256 + threadIdx.x;
        if (r >= R) continue;
        const GatherVar g = vars[vi];
        const int d = g.d;
        if (g.clamped) {
            if (mean)
                for (int k = 0; k < d; ++k) mean[g.mo + r * d + k] = val[(long long)(g.off + k) * es + r * rs_val];
            if (cov)
                for (int k = 0; k < d * d; ++k) cov[g.co + r * d * d + k] = 0.0;
            continue;
        }
        if (mean)
            for (int k = 0; k < d; ++k) mean[g.mo + r * d + k] = marg[(long long)(g.off + k) * es + r * rs_marg];
        if (cov)
            for (int a = 0; a < d; ++a)
                for (int b = 0; b <= a; ++b) {
                    const double x = marg[(long long)(g.off + d + a * (a + 1) / 2 + b) * es + r * rs_marg];
                    cov[g.co + (r * d + a) * d + b] = x;
                    cov[g.co + (r * d + b) * d + a] = x;
                }
    }
}

TreeParams params_of(const Engine* e, int want_fe) {
    TreeParams p{};
    p.ops = e->d_ops; p.aux = e->d_aux; p.cpool = e->d_cpool; p.msg = e->d_msg; p.marg = e->d_marg; p.val = e->d_val; p.prec = e->d_prec;
    p.term = e->d_term; p.stat = e->d_stat; p.R = e->R; p.RS = e->RS; p.want_fe = want_fe; p.status = e->d_status;
    const Program& P = e->prog;
    p.es = e->elem_fast ? 1 : e->RS;
    p.rs_msg = e->elem_fast ? P.msg_doubles : 1; p.rs_marg = e->elem_fast ? P.marg_doubles : 1; p.rs_val = e->elem_fast ? P.val_doubles : 1;
    p.rs_prec = e->elem_fast ? P.prec_doubles : 1; p.rs_term = e->elem_fast ? P.term_slots : 1; p.rs_stat = e->elem_fast ? P.stat_doubles : 1;
    return p;
}
template <int N, int PHASE>
void launch_phase(const Engine* e, const TreeParams& p, int l0, int l1) {
    if (l1 <= l0) return;
    const int mode = PHASE == 0 ? e->mode : e->mode_fe;
    if (mode == 3 && PHASE == 0 && N <= 4) {   // the strand schedule: one launch per strand level (tree_kernels.hpp k_tree_strands); always the whole sweep
        const Program& P = e->prog;
        const long long nrb = (e->R + 63) / 64;
        for (size_t l = 0; l + 1 < P.slvl_ptr.size(); ++l) {
            const int s0 = P.slvl_ptr[l], s1 = P.slvl_ptr[l + 1];
            if (s1 == s0) continue;
            const unsigned blocks = (unsigned)std::min<long long>((long long)(s1 - s0) * nrb, 1 << 22);
            hipLaunchKernelGGL((k_tree_strands<(N <= 4 ? N : 4)>), dim3(blocks), dim3(64), 0, e->stream, p, (const int*)e->d_sops, (const int*)e->d_strands, s0, s1);
        }
    } else if (mode == 2 || mode == 3) {
        const unsigned blocks = (unsigned)((e->R + 63) / 64);
        hipLaunchKernelGGL((k_tree_walk<N, PHASE>), dim3(blocks), dim3(64), 0, e->stream, p, e->prog.lvl_ptr[l0], e->prog.lvl_ptr[l1]);
    } else if (mode == 1) {
        const int rb = PHASE == 0 ? e->rb : e->rb_fe, wg = PHASE == 0 ? e->wg : 256;   // (the Bethe phase's instance is built for 256 threads: at 512 it spills)
        const unsigned blocks = (unsigned)((e->R + rb - 1) / rb);
        hipLaunchKernelGGL((k_tree_levels<N, PHASE>), dim3(blocks), dim3(wg), 0, e->stream, p, e->d_lvl, l0, l1, rb);
    } else {
        for (int l = l0; l < l1; ++l) {
            const int o0 = e->prog.lvl_ptr[l], o1 = e->prog.lvl_ptr[l + 1];
            if (o1 == o0) continue;
            const long long items = (long long)(o1 - o0) * e->R;
            const unsigned blocks = (unsigned)std::min<long long>((items + 255) / 256, 1 << 20);
            hipLaunchKernelGGL((k_tree_ops<N, PHASE>), dim3(blocks), dim3(256), 0, e->stream, p, o0, o1);
        }
    }
}
// dimensions above 8 (tree_wave_kernels.hpp, a translation unit per dimension class): a wavefront per (op, replica) and level, or — mode 2 — a wavefront
// per replica over the whole range
void launch_wave_phase(const Engine* e, const TreeParams& p, int phase, int l0, int l1) {
    if (l1 <= l0) return;
    const int dmax = e->prog.dmax;
    const wave::WaveVtbl* vt = wave::wave_vt(dmax);
    if ((phase == 0 ? e->mode : e->mode_fe) == 2) {
        vt->walk(phase, p, e->prog.lvl_ptr[l0], e->prog.lvl_ptr[l1], dmax, (unsigned)std::min<long long>(e->R, 1 << 20), e->stream);
        return;
    }
    // the second phase on the register tiles: a level's ops are sorted by opcode — the ones without joint-marginal algebra or a q(W) update (the terms of observation
    // nodes, entropies, sums: most of a chain's second phase) go to the light kernel instance (phase code 2: half the registers, twice the wavefronts per SIMD)
    auto light = [](int op) { return op == OP_FE_NOISE0 || op == OP_FE_NOISE1 || op == OP_FE_ENT || op == OP_FE_NOISE_MF || op == OP_SUM_TERMS || op == OP_MARG_PUSH; };
    for (int l = l0; l < l1; ++l) {
        const int o0 = e->prog.lvl_ptr[l], o1 = e->prog.lvl_ptr[l + 1];
        if (o1 == o0) continue;
        if (phase == 0 || dmax > 32) {
            vt->ops(phase, p, o0, o1, dmax, (unsigned)std::min<long long>((long long)(o1 - o0) * e->R, 1 << 20), e->stream);
            continue;
        }
        auto cls = [&](int i) { const int op = e->prog.ops[(size_t)i * OP_WORDS + W_OP]; return light(op) ? 2 : op == OP_FE_NOISE2M ? 3 : 1; };   // (3: the joint term alone)
        for (int a = o0; a < o1;) {
            const int ca = cls(a);
            int b = a + 1;
            while (b < o1 && cls(b) == ca) ++b;
            vt->ops(ca, p, a, b, dmax, (unsigned)std::min<long long>((long long)(b - a) * e->R, 1 << 20), e->stream);
            a = b;
        }
    }
}
rxhip_status wave_attributes(int dmax, std::string& err) {
    TCHK(wave::wave_vt(dmax)->prepare(dmax));
    return RXHIP_OK;
}
template <int N>
void launch_levels(const Engine* e, const TreeParams& p, int l0, int l1) {   // the sweep, then the second phase
    const int lf = e->prog.fe_level;
    launch_phase<N, 0>(e, p, l0, std::min(l1, lf));
    if (N <= 4 && !e->prog.fe_heavy) launch_phase<N, (N <= 4 ? 2 : 1)>(e, p, std::max(l0, lf), l1);   // the light instance: no `+` of two random inputs, no q(W) update in this graph
    else launch_phase<N, 1>(e, p, std::max(l0, lf), l1);
}
void launch(const Engine* e, const TreeParams& p, int l0, int l1) {
    if (e->tiled) {
        const int lf = e->prog.fe_level;
        launch_wave_phase(e, p, 0, l0, std::min(l1, lf));
        launch_wave_phase(e, p, 1, std::max(l0, lf), l1);
        return;
    }
    switch (e->prog.dmax) {
    case 1: launch_levels<1>(e, p, l0, l1); break;
    case 2: launch_levels<2>(e, p, l0, l1); break;
    case 4: launch_levels<4>(e, p, l0, l1); break;
    default: launch_levels<8>(e, p, l0, l1); break;   // (registers for 4x4 blocks: the 8x8 instance spills — it exists so that such graphs run at all)
    }
}
}  // namespace

rxhip_status create(const rxhip_graph_desc* g, int device, void* stream, Engine** out, std::string& err) {
    if (!g || !out) { err = "null argument"; return RXHIP_ERR_BADARG; }
    *out = nullptr;
    Engine* e = new Engine();
    try {
        Compiler c(g, e->prog);
        c.compile();
    } catch (const Fail& f) {
        err = f.msg;
        delete e;
        return f.st;
    } catch (const std::exception& ex) {
        err = ex.what();
        delete e;
        return RXHIP_ERR_BADARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { err = "no HIP device visible: the executor has no CPU fallback"; delete e; return RXHIP_ERR_NO_DEVICE; }
    if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;
    if (device >= ndev) { err = "device ordinal out of range"; delete e; return RXHIP_ERR_BADARG; }
    e->device = device;
    DevScope ds(device);
    e->R = std::max<int64_t>(1, g->n_replicas);
    if (e->R > (long long)FE_CHUNK * FE_CHUNK) { err = "more than 16 777 216 replicas in one engine"; delete e; return RXHIP_ERR_UNSUPPORTED; }   // (two-stage free-energy sum)
    e->RS = (e->R + 15) / 16 * 16;
    const Program& P = e->prog;
    // Dimensions 5 … 8: the lane-per-item instance holds 8×8 blocks in one lane's registers (it spills, and a batch of 256 replicas is four wavefronts); a wavefront
    // per item on register tiles is 2 – 3 × faster up to ≈ 1 000 replicas (d = 8, T = 64, two branches: one replica 2.5 → 0.9 ms, 256: 3.1 → 1.2), the lanes win
    // once the replicas fill the device (4 096: 3.4 against 5.4 ms; 65 536: 16 against 83): profiles/r06/tree_tile.txt
    e->tiled = P.dmax > 8 || (P.dmax > 4 && e->R <= 1024);
    if (const char* t = hook_env("RXHIP_TREE_TILE")) e->tiled = P.dmax > 8 || (P.dmax > 4 && std::atoi(t) != 0);
    if (P.has_mix || P.has_valnoise || P.has_gcv) e->tiled = false;   // (the mixture ops exist in the lane-per-item kernels only; the compiler refused dimensions above 8)
    e->elem_fast = e->tiled;
    e->allow_missing = g->allow_missing != 0;
    // schedule: deep graphs walk their levels inside a workgroup (one launch per iteration); wide, shallow ones take a launch per level
    const double avg_width = (double)P.n_ops / std::max(1, P.n_levels);
    e->mode = (P.n_levels > 24 && (e->R >= 512 || avg_width * (double)e->R < 16384.0)) ? 1 : 0;
    if (e->R < 8192 && P.n_levels > 1) {
        // below the batches that fill the device: a cost model per level, from the widths of THIS schedule — resident levels pay a barrier and
        // ⌈width · 16 / 256⌉ rule evaluations per thread in a workgroup of 16 replicas, a launch per level pays the launch and ⌈width · R / (256 · 1024)⌉
        // (measured: 1.0 µs per barrier, 3.6 µs per launch, 2.5 µs per dependent rule evaluation; a tree of depth 7 with 87 ops per level at 16 … 4 096
        //  replicas is 2 – 3 × faster launched per level, a chain with 8 ops per level 1.4 × faster resident: profiles/r05/tree_small_batches.txt)
        double c0 = 0.0, c1 = 0.0;
        for (int l = 0; l < P.n_levels; ++l) {
            const double w = (double)(P.lvl_ptr[l + 1] - P.lvl_ptr[l]);
            if (w <= 0.0) continue;
            c1 += 1.0 + std::ceil(w * (double)std::min<long long>(16, e->R) / ((e->R >= 4096 && P.dmax <= 4) ? 512.0 : 256.0)) * 2.5;
            c0 += 3.6 + std::ceil(w * (double)e->R / (256.0 * 1024.0)) * 2.5;
        }
        e->mode = c1 < c0 ? 1 : 0;
    }
    // very large batches: a lane per replica walks the whole schedule (no barriers; 64 replicas per wavefront) — from two wavefronts per SIMD on it passes the
    // workgroup-resident schedule (131 072 replicas × T = 128: 19.6 against 20.3 ms; at 65 536 the resident schedule wins, 10.65 against 11.67: profiles/r05/tree_modes.txt)
    if (e->R >= 131072) e->mode = 2;
    // the Bethe phase is one wide level of independent terms and a short sum tree: there the walk is ahead from 65 536 replicas on (2.05 against 2.69 ms), the
    // sweep only from 131 072 (scripts/time_tree_phases.py)
    e->mode_fe = (e->mode == 1 && e->R >= 65536) ? 2 : e->mode;
    if (e->tiled) {
        // wavefront per item: a launch per level — a rule at d = 16 is ≈ 17 µs of dependent LDS round trips inside its wavefront, more than a launch, so
        // the walk (one wavefront per replica, ops in sequence) only pays once the replicas alone fill the device (measured: profiles/r05/tree_wave_modes.txt)
        // (round 6, items of 1 / 2 / 4 wavefronts: up to 16 a launch per level stays ahead at every batch — 28.3 against 35.3 ms at 4 096 replicas; above, the walk from
        //  2 048 replicas — d = 32: 33.8 against 37.1 ms; profiles/r06/tree_wave_modes.txt)
        // (register tiles up to 32 — tree_tile_kernels.hpp: d = 16: 1.5 against 2.9 ms at 256 replicas, 8.5 against 7.0 at 4 096; d = 32: 2.1 against 4.1 at 256, 10.8 against 9.7 at
        //  2 048; the LDS-staged class above: 7.9 against 7.3 at 256 — a workgroup per CU already)
        //  (crossover: d = 16: 2 048 replicas — 4.55 against 4.45 ms; d = 32: 1 024 — 5.8 against 4.9: profiles/r06/tree_tile.txt)
        e->mode = (e->R >= (P.dmax > 32 ? 256 : P.dmax > 16 ? 1024 : 2048)) ? 2 : 0;
    }
    // the second phase on the register tiles: a launch per level with the light instance for the ops without joint-marginal algebra (launch_wave_phase) — one wide level of
    // independent terms — beats the walk at every batch (d = 16 × 4 096: 7.2 → 6.4 ms, d = 32 × 2 048: 8.2 → 6.8; profiles/r06/tree_wave_modes.txt); the LDS-staged class keeps the sweep's
    if (e->tiled) e->mode_fe = P.dmax <= 32 ? 0 : e->mode;
    // the strand schedule (register hand-over along dependent ops, wide levels at full occupancy): from the batches at which two long strands fill the device
    // (at every batch: a T = 128 chain of ONE replica is 0.60 ms in strands — its two recursions are two lanes walking 382 dependent ops — against 1.9 ms for
    //  workgroup-resident levels and ≈ 1.5 ms for 390 launches; 256 replicas 0.63 against 1.86; 65 536: 2.3 against 3.9: profiles/r06/tree_strands.txt)
    if (P.dmax <= 4) e->mode = 3;
    // the second phase is one wide level of independent terms and a short sum tree: a launch per level up to 65 536 replicas (0.09 against 1.0 ms for the
    // walk at 4 096, 1.20 against 1.34 at 65 536), the walk above (2.04 against 2.44 at 131 072): profiles/r06/tree_strands.txt
    if (e->mode == 3 || (e->mode == 1 && e->R >= 4096)) e->mode_fe = e->R >= 131072 ? 2 : 0;
    if (e->mode == 3 && P.dmax > 4) e->mode = 1;
    if (const char* m = hook_env("RXHIP_TREE_MODE")) e->mode = e->mode_fe = std::max(0, std::min(3, std::atoi(m)));
    if (e->mode == 3 && P.dmax > 4) e->mode = 2;                  // (the strand kernel carries a message in registers: instances 1, 2, 4)
    if (e->mode_fe == 3) e->mode_fe = e->R >= 131072 ? 2 : 0;    // (the Bethe phase is one wide level of independent terms: no strands to speak of)
    if (const char* m = hook_env("RXHIP_TREE_MODE_FE")) e->mode_fe = std::max(0, std::min(2, std::atoi(m)));
    if (e->tiled && e->mode == 1) e->mode = e->mode_fe = 2;   // (no workgroup-resident schedule for the LDS-staged kernels)
    if (e->tiled && e->mode_fe == 1) e->mode_fe = 2;
    e->use_graph = e->mode == 0 && P.n_levels > 8;
    if (const char* q = hook_env("RXHIP_TREE_GRAPH")) e->use_graph = e->use_graph && std::atoi(q) != 0;
    {
        // Workgroups of the resident schedule.  Sweep phase, from 4 096 replicas: 512 threads owning R / 256 replicas (≤ 256) — one workgroup of eight wavefronts
        // per CU; below, and for the Bethe phase: 256 threads owning R / 512 replicas (≤ 128).  Measured optimum at every batch from 4 096 to 65 536 replicas
        // (profiles/r05/tree_modes.txt: 16 384 replicas 5.79 → 3.19 ms, 32 768: 7.47 → 5.1, 65 536: 11.7 → 9.2).
        e->wg = (e->R >= 4096 && P.dmax <= 4) ? 512 : 256;
        long long rb = (e->R / (e->wg == 512 ? 256 : 512)) / 16 * 16;
        e->rb = (int)std::min<long long>(e->wg == 512 ? 256 : 128, std::max<long long>(16, rb));
        e->rb_fe = (int)std::min<long long>(128, std::max<long long>(16, (e->R / 512) / 16 * 16));
        if (const char* q = hook_env("RXHIP_TREE_RB")) e->rb = e->rb_fe = std::max(16, std::min(256, std::atoi(q) / 16 * 16));
        if (const char* q = hook_env("RXHIP_TREE_WG")) e->wg = (std::atoi(q) >= 512 && P.dmax <= 4) ? 512 : 256;   // (the 8×8 instance is built for 256 threads)
    }
    if (const char* path = hook_env("RXHIP_TREE_DUMP")) {   // (test hook) the schedule, a line per level: the opcodes of its ops — read next to a kernel trace of mode 0
        if (FILE* fp = std::fopen(path, "w")) {
            for (int l = 0; l < P.n_levels; ++l) {
                std::fprintf(fp, "%d%s", l, l >= P.fe_level ? " fe" : "");
                for (int i = P.lvl_ptr[l]; i < P.lvl_ptr[l + 1]; ++i) std::fprintf(fp, " %d:%d", P.ops[(size_t)i * OP_WORDS + W_OP], P.ops[(size_t)i * OP_WORDS + W_FLAGS]);
                std::fprintf(fp, "\n");
            }
            std::fclose(fp);
        }
    }
    auto cleanup = [&](rxhip_status st) { destroy(e); return st; };
    if (stream) e->stream = (hipStream_t)stream;
    else {
        if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) { err = "hipStreamCreate failed"; return cleanup(RXHIP_ERR_HIP); }
        e->own_stream = true;
    }
    rxhip_status st;
    if (e->tiled && (st = wave_attributes(P.dmax, err))) return cleanup(st);
    if ((st = upload(&e->d_sops, P.sops, err)) || (st = upload(&e->d_strands, P.strands, err))) return cleanup(st);
    if ((st = upload(&e->d_ops, P.ops, err)) || (st = upload(&e->d_aux, P.aux, err)) || (st = upload(&e->d_lvl, P.lvl_ptr, err)) || (st = upload(&e->d_cpool, P.cpool, err)) ||
        (st = upload(&e->d_prec_init, P.prec_init, err)) || (st = upload(&e->d_marg_init, P.marg_init, err)) || (st = zalloc(&e->d_msg, P.msg_doubles * e->RS, err)) || (st = zalloc(&e->d_marg, P.marg_doubles * e->RS, err)) ||
        (st = zalloc(&e->d_val, P.val_doubles * e->RS, err)) || (st = zalloc(&e->d_prec, P.prec_doubles * e->RS, err)) || (st = zalloc(&e->d_term, P.term_slots * e->RS, err)) ||
        (st = zalloc(&e->d_stat, P.stat_doubles * e->RS, err)) || (st = zalloc(&e->d_fe_rep, e->RS, err)) || (st = zalloc(&e->d_fe_part, (e->R + FE_CHUNK - 1) / FE_CHUNK, err)))
        return cleanup(st);
    if (hipMalloc(&e->d_status, sizeof(int)) != hipSuccess || hipMemset(e->d_status, 0, sizeof(int)) != hipSuccess) { err = "hipMalloc failed"; return cleanup(RXHIP_ERR_HIP); }
    e->data_set.assign(P.data_vars.size(), 0);
    e->have_data = P.data_vars.empty();
    *out = e;
    return RXHIP_OK;
}

rxhip_status plan(const rxhip_graph_desc* g, rxhip_tree_info* out, uint64_t* rule_calls, uint64_t* products, uint64_t* marginals, std::string& err) {
    if (!g || !out) { err = "null argument"; return RXHIP_ERR_BADARG; }
    Engine e;
    try {
        Compiler c(g, e.prog);
        c.compile();
    } catch (const Fail& f) {
        err = f.msg;
        return f.st;
    } catch (const std::exception& ex) {
        err = ex.what();
        return RXHIP_ERR_BADARG;
    }
    e.mode = -1;   // (the schedule is chosen with the batch, at creation)
    info(&e, out);
    if (rule_calls) *rule_calls = e.prog.rule_calls;
    if (products) *products = e.prog.products;
    if (marginals) *marginals = e.prog.marginals;
    return RXHIP_OK;
}

void destroy(Engine* e) {
    if (!e) return;
    DevScope ds(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (void* q : {(void*)e->d_sops, (void*)e->d_strands, (void*)e->d_ops, (void*)e->d_aux, (void*)e->d_lvl, (void*)e->d_status, (void*)e->d_cpool, (void*)e->d_msg, (void*)e->d_marg, (void*)e->d_val, (void*)e->d_prec,
                    (void*)e->d_term, (void*)e->d_stat, (void*)e->d_prec_init, (void*)e->d_marg_init, (void*)e->d_fe_rep, (void*)e->d_fe_hist, (void*)e->d_fe_part})
        if (q) (void)hipFree(q);
    for (hipGraphExec_t x : e->gexec) if (x) (void)hipGraphExecDestroy(x);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

rxhip_status set_data(Engine* e, const int64_t* vars, int64_t n_vars, const double* host, std::string& err) {
    if (!e || !vars || !host || n_vars <= 0) { err = "set_data: null argument"; return RXHIP_ERR_BADARG; }
    DevScope ds(e->device);
    const Program& P = e->prog;
    long long rows = 0;
    for (int64_t i = 0; i < n_vars; ++i) {
        const int64_t v = vars[i];
        if (v < 0 || v >= (int64_t)P.vclass.size() || P.vclass[v] != VC_DATA) { err = "set_data: variable " + std::to_string(v) + " is not a data variable"; return RXHIP_ERR_BADARG; }
        rows += P.dim[v];
    }
    // the host rows go up as they are ([replica][Σ dims], in blocks of at most 256 MB) and are scattered into the replica-fastest slots on the device:
    // one kernel per run of adjacent slots
    TCHK(hipStreamSynchronize(e->stream));
    const long long blk = std::max<long long>(1, std::min<long long>(e->R, ((256ll << 20) / 8) / std::max<long long>(1, rows)));
    double* d_tmp = nullptr;
    TCHK(hipMalloc(&d_tmp, sizeof(double) * (size_t)blk * (size_t)rows));
    rxhip_status st_sc = RXHIP_OK;
    for (long long r0 = 0; r0 < e->R && !st_sc; r0 += blk) {
        const long long nr = std::min(blk, e->R - r0);
        if (hipMemcpy(d_tmp, host + (size_t)r0 * rows, sizeof(double) * (size_t)nr * rows, hipMemcpyHostToDevice) != hipSuccess) { st_sc = RXHIP_ERR_HIP; break; }
        long long col = 0;
        for (int64_t i = 0; i < n_vars;) {
            int64_t j = i;
            long long width = 0;
            while (j < n_vars && P.val_off[vars[j]] == P.val_off[vars[i]] + width) { width += P.dim[vars[j]]; ++j; }
            const dim3 grid((unsigned)((nr + 31) / 32), (unsigned)((width + 31) / 32));
            const long long es = e->elem_fast ? 1 : e->RS, rs = e->elem_fast ? P.val_doubles : 1;
            hipLaunchKernelGGL(k_tree_scatter, grid, dim3(256), 0, e->stream, e->d_val + (size_t)P.val_off[vars[i]] * es + (size_t)r0 * rs, (const double*)d_tmp, nr, es, rs, rows, col, width, e->d_status);
            col += width;
            i = j;
        }
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess) st_sc = RXHIP_ERR_HIP;
    }
    (void)hipFree(d_tmp);
    if (st_sc) { err = "set_data: copying the data to the device failed"; return st_sc; }
    {
        int flag = 0;
        TCHK(hipMemcpy(&flag, e->d_status, sizeof(int), hipMemcpyDeviceToHost));
        if ((flag & 2) && e->allow_missing) TCHK(hipMemset(e->d_status, 0, sizeof(int)));   // (NaN = `missing`: the leaves and Bethe terms of this engine look for it)
        else if (flag & 2) {
            TCHK(hipMemset(e->d_status, 0, sizeof(int)));
            err = "set_data: the data hold NaN or Inf — `missing` observations need an engine created with rxhip_graph_desc.allow_missing (the schedule keeps the data leaves in precision form then)";
            return RXHIP_ERR_BADARG;
        }
    }
    for (int64_t i = 0; i < n_vars; ++i) {
        const auto it = std::lower_bound(P.data_vars.begin(), P.data_vars.end(), vars[i]);
        e->data_set[it - P.data_vars.begin()] = 1;
    }
    e->have_data = std::all_of(e->data_set.begin(), e->data_set.end(), [](char c) { return c != 0; });
    return RXHIP_OK;
}

rxhip_status run(Engine* e, int iterations, int want_fe, std::string& err) {
    if (!e || iterations < 1) { err = "run: iterations must be positive"; return RXHIP_ERR_BADARG; }
    if (!e->have_data) { err = "run before every data variable has been set"; return RXHIP_ERR_STATE; }
    DevScope ds(e->device);
    const Program& P = e->prog;
    if (iterations > e->fe_cap) {
        TCHK(hipStreamSynchronize(e->stream));
        if (e->d_fe_hist) TCHK(hipFree(e->d_fe_hist));
        e->d_fe_hist = nullptr;
        TCHK(hipMalloc(&e->d_fe_hist, sizeof(double) * (size_t)iterations));
        e->fe_cap = iterations;
    }
    TCHK(hipMemsetAsync(e->d_status, 0, sizeof(int), e->stream));
    // a run starts from the @initialization marginals (iterations re-push the data: src/inference/batch.jl:391-430)
    // — unless the caller drives the loop one iteration per call (rxhip_tree_continue): then only the first run does
    if (P.prec_doubles > 0 && !(e->cont && e->ran)) {
        const long long total = P.prec_doubles * e->RS;
        hipLaunchKernelGGL(k_tree_broadcast, dim3((unsigned)std::min<long long>((total + 255) / 256, 4096)), dim3(256), 0, e->stream, e->d_prec, (const double*)e->d_prec_init,
                           (long long)P.prec_doubles, e->RS, e->elem_fast ? 1 : 0);
    }
    if (P.has_mf && !(e->cont && e->ran)) {   // the marginals the mean-field rules read in the first iteration
        const long long total = P.marg_doubles * e->RS;
        hipLaunchKernelGGL(k_tree_broadcast, dim3((unsigned)std::min<long long>((total + 255) / 256, 4096)), dim3(256), 0, e->stream, e->d_marg, (const double*)e->d_marg_init,
                           (long long)P.marg_doubles, e->RS, e->elem_fast ? 1 : 0);
    }
    const TreeParams p = params_of(e, want_fe);
    // without the free energy on a graph without precision variables the sweep ends with the marginals
    const int l_end = (!want_fe && P.prec_doubles == 0) ? P.fe_level : (P.lazy_level >= 0 ? P.lazy_level : P.n_levels);
    if (!e->ev0) { TCHK(hipEventCreate(&e->ev0)); TCHK(hipEventCreate(&e->ev1)); }
    hipGraphExec_t gx = nullptr;
    if (e->use_graph) {
        const int wf = want_fe ? 1 : 0;
        if (e->gexec[wf] && e->g_lend[wf] != l_end) { (void)hipGraphExecDestroy(e->gexec[wf]); e->gexec[wf] = nullptr; }
        if (!e->gexec[wf]) {
            hipGraph_t gr = nullptr;
            if (hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                launch(e, p, 0, l_end);
                if (hipStreamEndCapture(e->stream, &gr) == hipSuccess && gr && hipGraphInstantiate(&e->gexec[wf], gr, nullptr, nullptr, 0) == hipSuccess) e->g_lend[wf] = l_end;
                else e->gexec[wf] = nullptr;
                if (gr) (void)hipGraphDestroy(gr);
            }
            (void)hipGetLastError();
            if (!e->gexec[wf]) e->use_graph = false;   // (the direct launches below are the same kernels: nothing is lost but the launch overhead)
        }
        gx = e->gexec[wf];
    }
    TCHK(hipEventRecord(e->ev0, e->stream));
    for (int it = 0; it < iterations; ++it) {
        if (gx) TCHK(hipGraphLaunch(gx, e->stream));
        else launch(e, p, 0, l_end);
        if (want_fe) {   // Σ over replicas of term[root]: chunks of FE_CHUNK into partials, the partials (≤ FE_CHUNK of them: up to 16.7 M replicas) into the iteration's slot
            const unsigned nb = (unsigned)((e->R + FE_CHUNK - 1) / FE_CHUNK);
            const double* root = e->d_term + (size_t)P.fe_root * (e->elem_fast ? 1 : e->RS);
            const long long stride = e->elem_fast ? P.term_slots : 1;
            if (nb == 1)
                hipLaunchKernelGGL(k_tree_fe_total, dim3(1), dim3(256), 0, e->stream, root, stride, e->R, e->d_fe_rep, e->d_fe_hist + it);
            else {
                hipLaunchKernelGGL(k_tree_fe_total, dim3(nb), dim3(256), 0, e->stream, root, stride, e->R, e->d_fe_rep, e->d_fe_part);
                hipLaunchKernelGGL(k_tree_fe_total, dim3(1), dim3(256), 0, e->stream, (const double*)e->d_fe_part, 1ll, (long long)nb, (double*)nullptr, e->d_fe_hist + it);
            }
        }
    }
    TCHK(hipEventRecord(e->ev1, e->stream));
    TCHK(hipGetLastError());
    TCHK(hipStreamSynchronize(e->stream));
    {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, e->ev0, e->ev1) == hipSuccess) e->last_iteration_ms = (double)ms / iterations;
    }
    int status = 0;
    TCHK(hipMemcpy(&status, e->d_status, sizeof(int), hipMemcpyDeviceToHost));
    e->ran = true;
    e->push_done = false;
    e->last_iterations = iterations;
    e->last_want_fe = want_fe;
    ++e->runs;
    if (status & 1) { err = "a message or marginal precision was not positive definite"; return RXHIP_ERR_NOT_POSDEF; }
    if (want_fe) {
        std::vector<double> h((size_t)iterations);
        TCHK(hipMemcpy(h.data(), e->d_fe_hist, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
        for (double v : h)
            if (!std::isfinite(v)) { err = "free energy is not finite"; return RXHIP_ERR_NONFINITE_FE; }
    }
    return RXHIP_OK;
}

rxhip_status get_marginals(Engine* e, const int64_t* vars, int64_t n_vars, double* mean, double* cov, std::string& err) {
    if (!e || !vars || n_vars <= 0) { err = "get_marginals: null argument"; return RXHIP_ERR_BADARG; }
    if (!e->ran) { err = "get_marginals before run"; return RXHIP_ERR_STATE; }
    DevScope ds(e->device);
    const Program& P = e->prog;
    // gathered on the device into the caller's layout, chunks of variables of at most 256 MB of results, one copy per chunk and array
    std::vector<GatherVar> gv((size_t)n_vars);
    for (int64_t i = 0; i < n_vars; ++i) {
        const int64_t v = vars[i];
        // (a derived clamped value — `a + b` of two data variables, a random variable of the MODEL — is a point mass, as the reference publishes it; so is a data variable)
        const bool cl = v >= 0 && v < (int64_t)P.vclass.size() && (P.vclass[v] == VC_DERIVED || P.vclass[v] == VC_DATA);
        if (v < 0 || v >= (int64_t)P.vclass.size() || (P.vclass[v] != VC_GAUSS && !cl)) { err = "get_marginals: variable " + std::to_string(v) + " is not a random Gaussian variable"; return RXHIP_ERR_BADARG; }
        gv[i].off = cl ? P.val_off[v] : P.marg_off[v];
        gv[i].d = P.dim[v];
        gv[i].clamped = cl ? 1 : 0;
    }
    // a marginal that is the image of another one (the output of `A * x`) is formed in the second phase; a sweep that ran without it forms them now
    if (!e->push_done && P.lazy_level >= 0) {
        bool want = false;
        for (int64_t i = 0; i < n_vars; ++i) want = want || P.is_push[vars[i]];
        if (want) {
            launch(e, params_of(e, 0), P.lazy_level, P.lazy_level + 1);
            TCHK(hipGetLastError());
            e->push_done = true;
        }
    }
    TCHK(hipStreamSynchronize(e->stream));
    const size_t cap = (256u << 20) / 8;   // doubles per chunk and array
    size_t mo = 0, co = 0;
    for (int64_t i0 = 0; i0 < n_vars;) {
        int64_t i1 = i0;
        size_t nm = 0, nc = 0;
        while (i1 < n_vars) {
            const size_t d = (size_t)gv[i1].d, am = (size_t)e->R * d, ac = am * d;
            if (i1 > i0 && (nm + am > cap || nc + ac > cap)) break;
            gv[i1].mo = (long long)nm; gv[i1].co = (long long)nc;
            nm += am; nc += ac; ++i1;
        }
        GatherVar* d_gv = nullptr;
        double *d_m = nullptr, *d_c = nullptr;
        auto freeall = [&]() { for (void* q : {(void*)d_gv, (void*)d_m, (void*)d_c}) if (q) (void)hipFree(q); };
        hipError_t he = hipMalloc(&d_gv, sizeof(GatherVar) * (size_t)(i1 - i0));
        if (he == hipSuccess) he = hipMemcpy(d_gv, gv.data() + i0, sizeof(GatherVar) * (size_t)(i1 - i0), hipMemcpyHostToDevice);
        if (he == hipSuccess && mean) he = hipMalloc(&d_m, sizeof(double) * nm);
        if (he == hipSuccess && cov) he = hipMalloc(&d_c, sizeof(double) * nc);
        if (he == hipSuccess) {
            const long long items = (long long)(i1 - i0) * ((e->R + 255) / 256);
            hipLaunchKernelGGL(k_tree_gather, dim3((unsigned)std::min<long long>(items, 1 << 20)), dim3(256), 0, e->stream, (const double*)e->d_marg, (const double*)e->d_val, (const GatherVar*)d_gv, (int)(i1 - i0),
                               e->R, e->elem_fast ? 1ll : e->RS, e->elem_fast ? (long long)P.marg_doubles : 1ll, e->elem_fast ? (long long)P.val_doubles : 1ll, d_m, d_c);
            he = hipGetLastError();
        }
        if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
        if (he == hipSuccess && mean) he = hipMemcpy(mean + mo, d_m, sizeof(double) * nm, hipMemcpyDeviceToHost);
        if (he == hipSuccess && cov) he = hipMemcpy(cov + co, d_c, sizeof(double) * nc, hipMemcpyDeviceToHost);
        freeall();
        if (he != hipSuccess) { err = std::string("get_marginals: ") + hipGetErrorString(he); return RXHIP_ERR_HIP; }
        mo += nm; co += nc;
        i0 = i1;
    }
    return RXHIP_OK;
}

rxhip_status get_precision(Engine* e, int64_t var, double* nu, double* V, std::string& err) {
    if (!e) { err = "null engine"; return RXHIP_ERR_BADARG; }
    if (!e->ran) { err = "get_precision before run"; return RXHIP_ERR_STATE; }
    const Program& P = e->prog;
    if (var < 0 || var >= (int64_t)P.vclass.size() || P.vclass[var] != VC_PREC) { err = "not a precision variable"; return RXHIP_ERR_BADARG; }
    DevScope ds(e->device);
    const int d = P.dim[var], tri = d * (d + 1) / 2;
    std::vector<double> buf((size_t)(1 + tri) * e->RS);
    TCHK(hipStreamSynchronize(e->stream));
    long long bes = e->RS, brs = 1;   // the copy's layout: element k of replica r at buf[k·bes + r·brs]
    if (e->elem_fast) {   // a replica's slots are contiguous: the (1 + tri) doubles of this variable out of every replica's block
        bes = 1; brs = 1 + tri;
        TCHK(hipMemcpy2D(buf.data(), sizeof(double) * (1 + tri), e->d_prec + P.prec_off[var], sizeof(double) * (size_t)P.prec_doubles, sizeof(double) * (1 + tri), (size_t)e->R,
                         hipMemcpyDeviceToHost));
    } else
        TCHK(hipMemcpy(buf.data(), e->d_prec + (size_t)P.prec_off[var] * e->RS, sizeof(double) * buf.size(), hipMemcpyDeviceToHost));
    for (long long r = 0; r < e->R; ++r) {
        if (nu) nu[r] = buf[(size_t)r * brs];
        if (V)
            for (int a = 0; a < d; ++a)
                for (int b = 0; b <= a; ++b) {
                    const double x = buf[(size_t)(1 + a * (a + 1) / 2 + b) * bes + (size_t)r * brs];
                    V[((size_t)r * d + a) * d + b] = x;
                    V[((size_t)r * d + b) * d + a] = x;
                }
    }
    return RXHIP_OK;
}

// q(z) of a mixture node's switch (K probabilities) or the concentrations of q(s) of a probability vector: [replica][K]
rxhip_status get_discrete(Engine* e, int64_t var, double* out, int32_t* n_components, std::string& err) {
    if (!e) { err = "null engine"; return RXHIP_ERR_BADARG; }
    if (!e->ran) { err = "get_discrete before run"; return RXHIP_ERR_STATE; }
    const Program& P = e->prog;
    if (var < 0 || var >= (int64_t)P.vclass.size() || (P.vclass[var] != VC_CAT && P.vclass[var] != VC_DIR)) { err = "not the switch of a mixture node or its probability vector"; return RXHIP_ERR_BADARG; }
    const int K = P.discrete_k[var];
    if (n_components) *n_components = K;
    if (!out) return RXHIP_OK;
    DevScope ds(e->device);
    std::vector<double> buf((size_t)K * e->RS);
    TCHK(hipStreamSynchronize(e->stream));
    TCHK(hipMemcpy(buf.data(), e->d_prec + (size_t)P.prec_off[var] * e->RS, sizeof(double) * buf.size(), hipMemcpyDeviceToHost));   // (mixture graphs: lane-per-item kernels, replica-fastest storage)
    for (long long r = 0; r < e->R; ++r)
        for (int k = 0; k < K; ++k) out[(size_t)r * K + k] = buf[(size_t)k * e->RS + r];
    return RXHIP_OK;
}

rxhip_status get_free_energy(Engine* e, double* per_iteration, std::string& err) {
    if (!e || !per_iteration) { err = "null argument"; return RXHIP_ERR_BADARG; }
    if (!e->ran || !e->last_want_fe) { err = "no free energy: run with want_free_energy first"; return RXHIP_ERR_STATE; }
    DevScope ds(e->device);
    TCHK(hipMemcpy(per_iteration, e->d_fe_hist, sizeof(double) * (size_t)e->last_iterations, hipMemcpyDeviceToHost));
    return RXHIP_OK;
}
rxhip_status get_free_energy_per_replica(Engine* e, double* per_replica, std::string& err) {
    if (!e || !per_replica) { err = "null argument"; return RXHIP_ERR_BADARG; }
    if (!e->ran || !e->last_want_fe) { err = "no free energy: run with want_free_energy first"; return RXHIP_ERR_STATE; }
    DevScope ds(e->device);
    TCHK(hipMemcpy(per_replica, e->d_fe_rep, sizeof(double) * (size_t)e->R, hipMemcpyDeviceToHost));
    return RXHIP_OK;
}
void counters(Engine* e, uint64_t* rule_calls, uint64_t* products, uint64_t* marginals) {
    const uint64_t k = (uint64_t)e->R * (uint64_t)std::max(1, e->last_iterations);
    if (rule_calls) *rule_calls = e->prog.rule_calls * k;
    if (products) *products = e->prog.products * k;
    if (marginals) *marginals = e->prog.marginals * k;
}
void set_continue(Engine* e, bool on) { e->cont = on; }
void info(Engine* e, rxhip_tree_info* out) {
    const Program& P = e->prog;
    out->n_ops = P.n_ops; out->n_levels = P.n_levels; out->n_messages = P.n_messages;
    out->doubles_per_replica = P.msg_doubles + P.marg_doubles + P.val_doubles + P.prec_doubles + P.term_slots + P.stat_doubles;
    out->bytes_per_sweep = e->mode == 3 ? P.bytes_per_sweep_strands : P.bytes_per_sweep;
    out->io_bytes_per_sweep = P.io_bytes;
    out->n_strands = (int64_t)(P.strands.size() / 2);
    out->n_strand_levels = (int64_t)P.slvl_ptr.size() - 1;
    out->longest_strand = P.longest_strand;
    out->kernels = e->mode < 0 ? -1 : !e->tiled ? 0 : P.dmax <= 32 ? 1 : 2;
    out->strand_bytes_per_sweep = P.bytes_per_sweep_strands;
    out->fe_bytes_per_sweep = P.fe_bytes;
    out->dmax = P.dmax; out->mode = e->mode; out->replicas_per_workgroup = e->rb;
    int np = 0;
    for (int c : P.vclass) np += c == VC_PREC;
    out->n_precision_vars = np;
    out->last_iteration_ms = e->last_iteration_ms;
}
int device_of(Engine* e) { return e->device; }
double* free_energy_device(Engine* e, int* iterations) {
    const bool have = e->ran && e->last_want_fe;
    if (iterations) *iterations = have ? e->last_iterations : 0;
    return have ? e->d_fe_hist : nullptr;
}
void* stream_of(Engine* e) { return (void*)e->stream; }
rxhip_status sync(Engine* e, std::string& err) {
    DevScope ds(e->device);
    TCHK(hipStreamSynchronize(e->stream));
    return RXHIP_OK;
}

// ------------------------------------------------------------------------------------------
// one rule on a one-node schedule (include/rxhip.h rxhip_rule_eval)
rxhip_status rule_eval(const rxhip_rule_call* c, int device, std::string& err) {
    if (!c || c->n < 1 || !c->in_a || !c->in_B || !c->out_a || !c->out_B) { err = "rule_eval: null argument"; return RXHIP_ERR_BADARG; }
    const int t = c->node_type;
    const bool noise = t == RXHIP_NODE_MVNORMAL_MEAN_COV || t == RXHIP_NODE_NORMAL_MEAN_VARIANCE || t == RXHIP_NODE_MVNORMAL_MEAN_PRECISION || t == RXHIP_NODE_NORMAL_MEAN_PRECISION;
    if (!noise && t != RXHIP_NODE_MULTIPLY && t != RXHIP_NODE_ADD) { err = "rule_eval: node type without a device rule"; return RXHIP_ERR_UNSUPPORTED; }
    const int dout = c->d_out, din = t == RXHIP_NODE_MULTIPLY ? c->d_in : c->d_out;
    if (dout < 1 || din < 1 || dout > wave::DMAX_WAVE || din > wave::DMAX_WAVE) { err = "rule_eval: dimensions 1..64"; return RXHIP_ERR_UNSUPPORTED; }
    if ((noise && (c->iface < 0 || c->iface > 1)) || (t == RXHIP_NODE_MULTIPLY && c->iface != 0 && c->iface != 2) || (t == RXHIP_NODE_ADD && (c->iface < 0 || c->iface > 2))) {
        err = "rule_eval: no message leaves through that interface"; return RXHIP_ERR_BADARG;
    }
    if ((noise || t == RXHIP_NODE_MULTIPLY) && !c->constant) { err = "rule_eval: the node's constant is missing"; return RXHIP_ERR_BADARG; }
    if (t == RXHIP_NODE_ADD && (!c->in2_a || !c->in2_B)) { err = "rule_eval: `+` needs two inbound messages"; return RXHIP_ERR_BADARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { err = "no HIP device visible"; return RXHIP_ERR_NO_DEVICE; }
    if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;
    if (device >= ndev) { err = "rule_eval: device ordinal out of range"; return RXHIP_ERR_BADARG; }
    DevScope ds(device);
    // the dimension of the inbound message and of the result
    const int d_in_msg = t == RXHIP_NODE_MULTIPLY ? (c->iface == 0 ? din : dout) : dout;
    const int d_res = t == RXHIP_NODE_MULTIPLY ? (c->iface == 0 ? dout : din) : dout;
    const int dmax = std::max(dout, din), N = dmax <= 1 ? 1 : dmax <= 2 ? 2 : dmax <= 4 ? 4 : 8;
    auto msz = [](int d) { return d + d * (d + 1) / 2; };
    const long long R = c->n, RS = (R + 15) / 16 * 16;
    // slots: in0 | in1 | rule output | converted output (message) ; marginal slot for the moment form
    const int o_in0 = 0, o_in1 = o_in0 + msz(d_in_msg), o_rule = o_in1 + msz(dout), o_conv = o_rule + msz(d_res);
    std::vector<double> cpool;
    std::vector<int> ops, aux;
    auto op = [&](int code, int d0) { ops.resize(ops.size() + OP_WORDS, 0); int* w = &ops[ops.size() - OP_WORDS]; w[W_OP] = code; w[W_D0] = d0; w[W_IN0] = w[W_IN1] = w[W_IN2] = w[W_OUT] = w[W_PREC] = w[W_TERM] = -1; return w; };
    int out_wp;   // the form the rule itself leaves its result in
    if (noise) {
        std::vector<double> M(c->constant, c->constant + (size_t)dout * dout), Mi((size_t)dout * dout);
        double ld = 0.0;
        if (!host_chol_inv(dout, M.data(), Mi.data(), &ld)) { err = "rule_eval: the noise parameter is not positive definite"; return RXHIP_ERR_NOT_POSDEF; }
        const bool prec = t == RXHIP_NODE_MVNORMAL_MEAN_PRECISION || t == RXHIP_NODE_NORMAL_MEAN_PRECISION;
        const std::vector<double>&Sg = prec ? Mi : M, &W = prec ? M : Mi;
        cpool.insert(cpool.end(), Sg.begin(), Sg.end());
        cpool.insert(cpool.end(), W.begin(), W.end());
        cpool.push_back(prec ? ld : -ld);
        int* w = op(OP_NOISE, dout);
        w[W_IN0] = o_in0; w[W_OUT] = o_rule; w[W_C0] = 0;
        if (c->in_form) w[W_FLAGS] |= F_IN0_WP | F_OUT_WP;
        out_wp = c->in_form;
    } else if (t == RXHIP_NODE_MULTIPLY) {
        cpool.assign(c->constant, c->constant + (size_t)dout * din);
        int* w = op(c->iface == 0 ? OP_MUL_OUT : OP_MUL_IN, dout);
        w[W_D1] = din; w[W_IN0] = o_in0; w[W_OUT] = o_rule; w[W_C0] = 0;
        if (c->in_form) w[W_FLAGS] |= F_IN0_WP;
        out_wp = c->iface == 0 ? 0 : 1;
    } else {
        cpool.push_back(0.0);
        int* w = op(c->iface == 0 ? OP_ADD_OUT : OP_ADD_IN, dout);
        w[W_IN0] = o_in0; w[W_IN1] = o_in1; w[W_OUT] = o_rule;
        if (c->in_form) w[W_FLAGS] |= F_IN0_WP | F_IN1_WP;
        out_wp = (c->iface != 0 && c->in_form) ? 1 : 0;   // (:in) keeps the form of the message from `out`
    }
    {   // the result in the requested form: a one-message product (precision form) or marginal (mean, covariance)
        int* w = op(c->out_form ? OP_PRODUCT : OP_MARGINAL, d_res);
        w[W_OUT] = c->out_form ? o_conv : 0;
        w[W_LIST] = 0; w[W_N] = 1;
        aux = {o_rule, out_wp};
    }
    const long long msg_d = o_conv + msz(d_res), marg_d = msz(d_res) + 1;
    std::vector<double> img((size_t)msg_d * RS, 0.0);
    auto pack = [&](int off, int d, const double* a, const double* B, bool wp) {
        for (long long r = 0; r < RS; ++r) {
            const long long rr = r < R ? r : R - 1;   // (padding lanes never run)
            for (int k = 0; k < d; ++k) img[(size_t)(off + k) * RS + r] = a[(size_t)rr * d + k];
            for (int i = 0; i < d; ++i)
                for (int j = 0; j <= i; ++j) img[(size_t)(off + d + i * (i + 1) / 2 + j) * RS + r] = 0.5 * (B[((size_t)rr * d + i) * d + j] + B[((size_t)rr * d + j) * d + i]);
        }
        (void)wp;
    };
    pack(o_in0, d_in_msg, c->in_a, c->in_B, c->in_form);
    if (t == RXHIP_NODE_ADD) pack(o_in1, dout, c->in2_a, c->in2_B, c->in_form);
    int *d_ops = nullptr, *d_aux = nullptr, *d_status = nullptr;
    double *d_cp = nullptr, *d_msg = nullptr, *d_marg = nullptr;
    auto freeall = [&]() { for (void* q : {(void*)d_ops, (void*)d_aux, (void*)d_status, (void*)d_cp, (void*)d_msg, (void*)d_marg}) if (q) (void)hipFree(q); };
    rxhip_status st;
    if ((st = upload(&d_ops, ops, err)) || (st = upload(&d_aux, aux, err)) || (st = upload(&d_cp, cpool, err)) || (st = upload(&d_msg, img, err)) || (st = zalloc(&d_marg, marg_d * RS, err))) { freeall(); return st; }
    if (hipMalloc(&d_status, sizeof(int)) != hipSuccess || hipMemset(d_status, 0, sizeof(int)) != hipSuccess) { freeall(); err = "hipMalloc failed"; return RXHIP_ERR_HIP; }
    TreeParams p{};
    p.ops = d_ops; p.aux = d_aux; p.cpool = d_cp; p.msg = d_msg; p.marg = d_marg; p.R = R; p.RS = RS; p.status = d_status;
    p.es = RS; p.rs_msg = p.rs_marg = p.rs_val = p.rs_prec = p.rs_term = p.rs_stat = 1;   // (packed replica-fastest below, for either kernel family)
    const unsigned blocks = (unsigned)std::min<long long>((R + 255) / 256, 1 << 20);
    if (dmax > 8 && (st = wave_attributes(dmax, err))) { freeall(); return st; }
    for (int o = 0; o < 2; ++o) {
        if (dmax > 8) wave::wave_vt(dmax)->ops(0, p, o, o + 1, dmax, (unsigned)std::min<long long>(R, 1 << 20), nullptr);
        else if (N == 1) hipLaunchKernelGGL((k_tree_ops<1, 0>), dim3(blocks), dim3(256), 0, 0, p, o, o + 1);
        else if (N == 2) hipLaunchKernelGGL((k_tree_ops<2, 0>), dim3(blocks), dim3(256), 0, 0, p, o, o + 1);
        else if (N == 4) hipLaunchKernelGGL((k_tree_ops<4, 0>), dim3(blocks), dim3(256), 0, 0, p, o, o + 1);
        else hipLaunchKernelGGL((k_tree_ops<8, 0>), dim3(blocks), dim3(256), 0, 0, p, o, o + 1);
    }
    int status = 0;
    std::vector<double> res((size_t)(c->out_form ? msg_d : marg_d) * RS);
    hipError_t he = hipDeviceSynchronize();
    if (he == hipSuccess) he = hipMemcpy(&status, d_status, sizeof(int), hipMemcpyDeviceToHost);
    if (he == hipSuccess) he = hipMemcpy(res.data(), c->out_form ? d_msg : d_marg, sizeof(double) * res.size(), hipMemcpyDeviceToHost);
    freeall();
    if (he != hipSuccess) { err = std::string("rule_eval: ") + hipGetErrorString(he); return RXHIP_ERR_HIP; }
    if (status & 1) { err = "rule_eval: a matrix that must be positive definite was not"; return RXHIP_ERR_NOT_POSDEF; }
    const int ro = c->out_form ? o_conv : 0, d = d_res;
    for (long long r = 0; r < R; ++r) {
        for (int k = 0; k < d; ++k) c->out_a[(size_t)r * d + k] = res[(size_t)(ro + k) * RS + r];
        for (int i = 0; i < d; ++i)
            for (int j = 0; j <= i; ++j) {
                const double x = res[(size_t)(ro + d + i * (i + 1) / 2 + j) * RS + r];
                c->out_B[((size_t)r * d + i) * d + j] = x;
                c->out_B[((size_t)r * d + j) * d + i] = x;
            }
    }
    return RXHIP_OK;
}

}  // namespace tree
}  // namespace rxhip
