// rccl_abi.hip — the exchange entry points of the C ABI for hosts without torch (the Julia shim): RCCL over xGMI for the path's one collective,
// the global Bethe free energy (all-gather + a sum in rank order: bit-identical on every rank and run), and the mixture's 3K + 1 statistics.
// librccl is opened on first use (dlopen): the copy that belongs to the HIP runtime the library runs on.
#include "engine.hpp"

#include <cstring>
#include <mutex>

using namespace rxhip;

// out[j] = Σ_r buf[r][j], ranks in ascending order: the cross-GPU sums are bit-identical on every rank and from run to run
// whatever algorithm the collective library picked for moving the bytes
__global__ void k_rank_sum(const double* __restrict__ buf, double* __restrict__ out, int nranks, int n) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    double s = 0.0;
    for (int r = 0; r < nranks; ++r) s += buf[(size_t)r * n + j];
    out[j] = s;
}

#include <dlfcn.h>
#include <rccl/rccl.h>
namespace {
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
    bool ok = false;
};
Rccl& rccl() {
    static Rccl* r = [] {
        Rccl* q = new Rccl;
        // The RCCL to use is the one that belongs to the HIP runtime THIS library runs on: streams and device pointers of
        // one ROCm installation mean nothing to the libraries of another, and a process may carry two (a pip-installed
        // torch bundles its own librccl / libhsa-runtime64 next to its libamdhip64).  So: the directory of the loaded
        // libamdhip64 (dladdr of a HIP entry point) first, then the usual names.  RTLD_DEEPBIND keeps a second RCCL copy in
        // the process from interposing this one's internal symbols.
        std::vector<std::string> names;
        Dl_info di;
        if (dladdr((const void*)&hipGetDeviceCount, &di) && di.dli_fname) {
            std::string dir(di.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) {
                dir.resize(slash + 1);
                names.push_back(dir + "librccl.so.1");
                names.push_back(dir + "librccl.so");
            }
        }
        names.push_back("librccl.so.1");
        names.push_back("librccl.so");
        names.push_back("/opt/rocm/lib/librccl.so.1");
        if (const char* forced = std::getenv("RXHIP_RCCL_LIB")) names.assign(1, forced);  // this copy or none (deployments with their own RCCL; tests)
        for (size_t i = 0; !q->h && i < names.size(); ++i) q->h = dlopen(names[i].c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
        if (!q->h) {
            const char* de = dlerror();  // ONE call: dlerror() clears the message it returns
            q->err = std::string("librccl not found: ") + (de ? de : "?");
            return q;
        }
        bool all = true;
        auto sym = [&](const char* n) { void* p = dlsym(q->h, n); if (!p) { all = false; q->err = std::string("librccl lacks ") + n; } return p; };
        q->GetUniqueId = (decltype(q->GetUniqueId))sym("ncclGetUniqueId");
        q->CommInitRank = (decltype(q->CommInitRank))sym("ncclCommInitRank");
        q->CommDestroy = (decltype(q->CommDestroy))sym("ncclCommDestroy");
        q->CommCount = (decltype(q->CommCount))sym("ncclCommCount");
        q->AllGather = (decltype(q->AllGather))sym("ncclAllGather");
        q->GetErrorString = (decltype(q->GetErrorString))sym("ncclGetErrorString");
        q->ok = all;
        return q;
    }();
    return *r;
}
thread_local std::string g_comm_err;
// all-gather `n` doubles of every rank into the engine's scratch, then sum in rank order into `inout` (in place)
rxhip_status ordered_allreduce(rxhip_engine* e, void* comm, double* inout, int n, const char* what) {
    Rccl& r = rccl();
    if (!r.ok) return fail(e, RXHIP_ERR_RCCL, "%s: %s", what, r.err.c_str());
    if (!comm) return fail(e, RXHIP_ERR_BADARG, "%s: null communicator", what);
    int nranks = 0;
    ncclResult_t rc = r.CommCount((ncclComm_t)comm, &nranks);
    if (rc != ncclSuccess) return fail(e, RXHIP_ERR_RCCL, "%s: ncclCommCount: %s", what, r.GetErrorString(rc));
    if (nranks <= 1) return RXHIP_OK;  // one rank: the local value is the global one, bit for bit
    SET_DEVICE(e);
    const size_t need = (size_t)nranks * (size_t)n;
    if (need > e->coll_cap) {
        HIPCHK(e, hipStreamSynchronize(e->stream));
        if (e->d_coll) HIPCHK(e, hipFree(e->d_coll));
        e->d_coll = nullptr;
        HIPCHK(e, hipMalloc(&e->d_coll, sizeof(double) * need));
        e->coll_cap = need;
    }
    (void)hipGetLastError();
    rc = r.AllGather(inout, e->d_coll, (size_t)n, ncclDouble, (ncclComm_t)comm, e->stream);
    if (rc != ncclSuccess) return fail(e, RXHIP_ERR_RCCL, "%s: ncclAllGather: %s", what, r.GetErrorString(rc));
    hipLaunchKernelGGL(k_rank_sum, dim3((n + 63) / 64), dim3(64), 0, e->stream, (const double*)e->d_coll, inout, nranks, n);
    HIPCHK(e, hipGetLastError());
    return RXHIP_OK;
}
}  // namespace
extern "C" {

const char* rxhip_comm_last_error(void) { return g_comm_err.c_str(); }

rxhip_status rxhip_comm_unique_id(char* id128) {
    if (!id128) return RXHIP_ERR_BADARG;
    Rccl& r = rccl();
    if (!r.ok) { g_comm_err = r.err; return RXHIP_ERR_RCCL; }
    ncclUniqueId id;
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    ncclResult_t rc = r.GetUniqueId(&id);
    if (rc != ncclSuccess) { g_comm_err = std::string("ncclGetUniqueId: ") + r.GetErrorString(rc); return RXHIP_ERR_RCCL; }
    std::memcpy(id128, &id, sizeof id);
    return RXHIP_OK;
}

rxhip_status rxhip_comm_init_rank(void** comm, int32_t nranks, const char* id128, int32_t rank, int32_t device) {
    if (!comm || !id128 || nranks <= 0 || rank < 0 || rank >= nranks) return RXHIP_ERR_BADARG;
    *comm = nullptr;
    Rccl& r = rccl();
    if (!r.ok) { g_comm_err = r.err; return RXHIP_ERR_RCCL; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_comm_err = "no HIP device"; return RXHIP_ERR_NO_DEVICE; }
    DevGuard dg;
    if (device >= 0) {
        if (device >= ndev || dg.set(device) != hipSuccess) { g_comm_err = "device out of range"; return RXHIP_ERR_BADARG; }
    }
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    ncclComm_t c = nullptr;
    (void)hipGetLastError();  // RCCL checks hipGetLastError() after its launches: a stale non-sticky error of the host process must not fail it
    ncclResult_t rc = r.CommInitRank(&c, nranks, id, rank);
    if (rc != ncclSuccess) { g_comm_err = std::string("ncclCommInitRank: ") + r.GetErrorString(rc); return RXHIP_ERR_RCCL; }
    *comm = (void*)c;
    return RXHIP_OK;
}

rxhip_status rxhip_comm_destroy(void* comm) {
    if (!comm) return RXHIP_OK;
    Rccl& r = rccl();
    if (!r.ok) { g_comm_err = r.err; return RXHIP_ERR_RCCL; }
    ncclResult_t rc = r.CommDestroy((ncclComm_t)comm);
    if (rc != ncclSuccess) { g_comm_err = std::string("ncclCommDestroy: ") + r.GetErrorString(rc); return RXHIP_ERR_RCCL; }
    return RXHIP_OK;
}

rxhip_status rxhip_allreduce_free_energy(rxhip_engine* e, void* rccl_comm) {
    if (!e) return RXHIP_ERR_BADARG;
    if (e->tree) {   // the node-array executor: replicas shard over ranks with no other exchange (engine.hpp: e->stream / e->device are the executor's)
        int its = 0;
        double* fe = rxhip::tree::free_energy_device(e->tree, &its);
        if (!fe) return fail(e, RXHIP_ERR_STATE, "free energy was not requested in the last run");
        return ordered_allreduce(e, rccl_comm, fe, its, "allreduce_free_energy");
    }
    if (!e->ran || !e->last_want_fe) return fail(e, RXHIP_ERR_STATE, "free energy was not requested in the last run");
    double* fe = e->kind == 1 ? e->g.d_fe : e->kind == 2 ? e->h.d_fe_total : e->d_fe_total;
    return ordered_allreduce(e, rccl_comm, fe, e->last_iterations, "allreduce_free_energy");
}

rxhip_status rxhip_gmm_allreduce_statistics(rxhip_engine* e, void* rccl_comm) {
    TREE_GUARD(e);
    if (!e || e->kind != 1) return RXHIP_ERR_BADARG;
    return ordered_allreduce(e, rccl_comm, e->g.d_totals, e->g.nq, "gmm_allreduce_statistics");
}

}  // extern "C"
