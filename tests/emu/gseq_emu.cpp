// Host emulation of rxinfer.jl_amd/csrc/gseq_kernels.hpp for the CPU test suite: the kernel SOURCE compiled as plain C++ with
// a workgroup of ONE thread (every `for (e = tid; e < n; e += nthreads)` loop becomes a full loop, barriers are no-ops).
// Checks the index arithmetic, the LDS layout and the formulas of the kernels against the oracle without a GPU; the
// synchronisation of the real 256-thread workgroup is what the `-m gpu` tests check.  Test infrastructure only.
#include <cmath>
#include <cstddef>
#include <cstring>
#include <vector>

#define RXHIP_GSEQ_HOST_EMULATION 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#define __syncthreads() ((void)0)

namespace rxhip {
struct GenericModel {
    const double *A, *P, *B, *Q, *Qi;
};
constexpr int ST_NOT_POSDEF = 1;
static inline void atomicOr(int* p, int v) { *p |= v; }
struct Idx { unsigned x; };
static Idx threadIdx{0}, blockDim{1}, blockIdx{0};
static double* sm = nullptr;   // the dynamic LDS block
}  // namespace rxhip
#define RXHIP_GSEQ_EXTERN_SHARED(name)   /* `extern __shared__ double sm[]` of the kernels: the namespace-level pointer above */

#include "../../rxinfer.jl_amd/csrc/gseq_kernels.hpp"

extern "C" int gseq_emu_run(long long T, long long n_chains, int d, int dy, int ptt, int n_models, const double* user, const double* prior,
                            const int* chain_model, const int* step_model, const double* y, int smooth, double* mean, double* cov,
                            double* logev) {
    using namespace rxhip;
    GseqParams p{};
    p.T = T; p.n_chains = n_chains; p.d = d; p.dy = dy; p.ptt = ptt; p.fe = 1; p.y = y; p.mean = mean; p.cov = cov; p.user = user;
    p.prior = prior; p.chain_model = chain_model; p.step_model = step_model; p.fe_part = logev;
    int status = 0;
    p.status = &status;
    std::vector<double> lds(gseq_lds_bytes(d, dy) / sizeof(double));
    sm = lds.data();
    for (long long c = 0; c < n_chains; ++c) {
        blockIdx.x = (unsigned)c;
        k_gseq_forward(p);
        if (smooth) k_gseq_backward(p);
    }
    (void)n_models;
    return status;
}

// rxhip_filter_step on the host: T calls of k_gseq_stream_step; y [T][chain][dy], outputs [T][chain][·]
extern "C" int gseq_emu_stream(long long T, long long n_chains, int d, int dy, int ptt, const double* user, const double* prior,
                               const int* chain_model, const int* step_model, const double* cx, const double* cy, const double* y,
                               double* mean, double* cov, double* fe) {
    using namespace rxhip;
    std::vector<double> state((size_t)n_chains * ((size_t)d + (size_t)d * d));
    std::vector<double> lds(gseq_lds_bytes(d, dy) / sizeof(double));
    sm = lds.data();
    int status = 0;
    for (long long k = 0; k < T; ++k) {
        GseqStreamParams p{};
        p.n_chains = n_chains; p.k = k; p.d = d; p.dy = dy; p.ptt = ptt; p.first = k == 0; p.y = y + (size_t)k * n_chains * dy;
        p.state = state.data(); p.user = user; p.prior = prior; p.chain_model = chain_model; p.step_model = step_model; p.cx = cx; p.cy = cy;
        p.off_chain = 0; p.mean = mean + (size_t)k * n_chains * d; p.cov = cov + (size_t)k * n_chains * d * d; p.fe = fe + (size_t)k * n_chains;
        p.status = &status;
        for (long long c = 0; c < n_chains; ++c) {
            blockIdx.x = (unsigned)c;
            k_gseq_stream_step(p);
        }
    }
    return status;
}

// rxhip_get_node_marginals at d > 4 on the host: forward + backward with the cross-covariances kept, then k_joint_generic
extern "C" int gseq_emu_joints(long long T, long long n_chains, int d, int dy, int ptt, const double* user, const double* prior,
                               const int* step_model, const double* y, double* jmean, double* jcov) {
    using namespace rxhip;
    std::vector<double> mean((size_t)T * n_chains * d), cov((size_t)T * n_chains * d * d), cross((size_t)(T > 1 ? T - 1 : 1) * n_chains * d * d),
        logev((size_t)n_chains);
    GseqParams p{};
    p.T = T; p.n_chains = n_chains; p.d = d; p.dy = dy; p.ptt = ptt; p.fe = 0; p.y = y; p.mean = mean.data(); p.cov = cov.data(); p.user = user;
    p.prior = prior; p.step_model = step_model; p.fe_part = logev.data(); p.cross = cross.data();
    int status = 0;
    p.status = &status;
    std::vector<double> lds((gseq_lds_bytes(d, dy) > joint_lds_bytes(d) ? gseq_lds_bytes(d, dy) : joint_lds_bytes(d)) / sizeof(double) + 8);
    sm = lds.data();
    for (long long c = 0; c < n_chains; ++c) {
        blockIdx.x = (unsigned)c;
        k_gseq_forward(p);
        k_gseq_backward(p);
    }
    JointParams j{};
    j.T = T; j.n_chains = n_chains; j.d = d; j.dy = dy; j.mean = mean.data(); j.cov = cov.data(); j.cross = cross.data(); j.user = user;
    j.step_model = step_model; j.jmean = jmean; j.jcov = jcov;
    for (long long g = 0; g < (T - 1) * n_chains; ++g) {
        blockIdx.x = (unsigned)g;
        k_joint_generic(j);
    }
    return status;
}
