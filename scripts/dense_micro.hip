// dense_micro.hip — times the building blocks of the d = 64 path (256 workgroups, 40 repetitions each)
#include "gj_inverse_legacy.hpp"   // (includes csrc/dense_kernels.hpp)
#include <cstdio>
#include <vector>
using namespace rxhip;
constexpr int NT = 4, D = 64, LD = 66, REP = 100;
extern __shared__ __attribute__((aligned(16))) double smem[];
__global__ void __launch_bounds__(256) t_gj(const double* M, double* out) {
    double* rowbuf = smem;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    Acc<NT> a, b; acc_load<NT>(a, M, D, w, lane);
    LogProd lp; bool ok = true;
    for (int r = 0; r < REP; ++r) { b = a; ok = gj_inverse<NT>(b, rowbuf, rowbuf, w, lane, lp) && ok; a.v[0][0] += 1e-300 * b.v[1][1]; }
    acc_store<NT>(b, out + (size_t)blockIdx.x * D * D, D, w, lane);
    if (tid == 0) out[0] += lp.value() + (ok ? 0 : 1);
}
template <int MODE>
__global__ void __launch_bounds__(256) t_mm(const double* X, const double* Y, double* out) {
    double* M0 = smem; double* M1 = smem + D * LD;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    Acc<NT> a; acc_load<NT>(a, Y, D, w, lane); acc_store<NT>(a, M0, LD, w, lane); acc_store<NT>(a, M1, LD, w, lane);
    __syncthreads();
    acc_zero<NT>(a);
    for (int r = 0; r < REP; ++r) {
        if (MODE == 0) mm_acc<NT, false, false>(a, M1, LD, M0, LD, w, lane);        // both in LDS
        if (MODE == 1) mm_acc<NT, false, false>(a, X, D, M0, LD, w, lane);          // X from global
        if (MODE == 2) mm_acc<NT, false, true>(a, M1, LD, X, D, w, lane);           // Y' from global
        if (MODE == 3) mm_acc<NT, true, false>(a, M1, LD, M0, LD, w, lane);         // X' from LDS
        if (MODE == 4) { acc_store<NT>(a, M1, LD, w, lane); __syncthreads(); acc_load<NT>(a, M0, LD, w, lane); __syncthreads(); }
        if (MODE == 5) { matvec_gT(M1, X, D, D, M0, nullptr, 0.0, tid); __syncthreads(); }
        if (MODE == 6) { double d3[3]; block_dot3(M0, M1, D, M0, M1, D, M0, M1, D, M1 + 1024, tid, 256, d3); a.v[0][0] += d3[0]; }
        if (MODE == 7) { matvec_lds(M1, M0, LD, D, D, M0 + 70, nullptr, 0.0, tid); __syncthreads(); }
        if (MODE == 8) { acc_add_mat<NT>(a, X, D, w, lane, 1.0); }
        if (MODE == 9) { acc_store_tri<NT>(a, out + (size_t)blockIdx.x * 4096 + (size_t)(r & 1) * 0, w, lane); }
    }
    acc_store<NT>(a, out + (size_t)blockIdx.x * D * D, D, w, lane);
}
// mimic of the forward step: mm, mm, GJ, GJ
template <int WHAT>
__global__ void __launch_bounds__(256) t_step(const double* X, const double* Y, double* out) {
    double* M0 = smem; double* M1 = smem + D * LD; double* rowbuf = smem + 4 * D * LD + 384;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    Acc<NT> a; acc_load<NT>(a, Y, D, w, lane); acc_store<NT>(a, M0, LD, w, lane);
    __syncthreads();
    LogProd lp; bool ok = true;
    for (int r = 0; r < REP; ++r) {
        acc_zero<NT>(a);
        if (WHAT & 1) mm_acc<NT, false, false>(a, X, D, M0, LD, w, lane);
        acc_store<NT>(a, M1, LD, w, lane);
        __syncthreads();
        acc_load<NT>(a, Y, D, w, lane);
        if (WHAT & 1) mm_acc<NT, false, true>(a, M1, LD, X, D, w, lane);
        if (WHAT & 4) acc_load<NT>(a, Y, D, w, lane);
        if (WHAT & 2) ok = gj_inverse<NT>(a, rowbuf, rowbuf, w, lane, lp) && ok;
        acc_store<NT>(a, M1, LD, w, lane);
        __syncthreads();
        acc_add_mat<NT>(a, Y, D, w, lane, 1.0);
        if (WHAT & 4) acc_load<NT>(a, Y, D, w, lane);
        if (WHAT & 2) ok = gj_inverse<NT>(a, rowbuf, rowbuf, w, lane, lp) && ok;
        __syncthreads();
    }
    acc_store<NT>(a, out + (size_t)blockIdx.x * D * D, D, w, lane);
    if (tid == 0) out[0] += lp.value() + (ok ? 0 : 1);
}
int main() {
    std::vector<double> h(D * D);
    for (int i = 0; i < D; ++i) for (int j = 0; j < D; ++j) h[i * D + j] = (i == j ? 70.0 : 0.0) + 1.0 / (1 + i + j);
    double *M, *out; hipMalloc(&M, D * D * 8); hipMalloc(&out, 1024ull * D * D * 8 + 1024);
    hipMemcpy(M, h.data(), D * D * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = 150 * 1024;
    hipFuncSetAttribute((const void*)t_gj, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    auto run = [&](const char* name, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s %8.3f ms total  %8.2f us per op\n", name, ms, ms * 1e3 / REP);
    };
    run("gj_inverse 64x64", [&] { hipLaunchKernelGGL(t_gj, dim3(256), dim3(256), lds, 0, M, out); });
    // two / four workgroups per CU (registers allow two wavefronts per SIMD): what the sweep kernels see with two segments per CU
    run("gj_inverse 64x64, 512 workgroups", [&] { hipLaunchKernelGGL(t_gj, dim3(512), dim3(256), 16 * 1024, 0, M, out); });
    run("gj_inverse 64x64, 1024 workgroups", [&] { hipLaunchKernelGGL(t_gj, dim3(1024), dim3(256), 16 * 1024, 0, M, out); });
#define MM(mode, name) hipFuncSetAttribute((const void*)t_mm<mode>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
    run(name, [&] { hipLaunchKernelGGL(t_mm<mode>, dim3(256), dim3(256), lds, 0, M, M, out); });
    MM(0, "mm LDS x LDS"); MM(1, "mm global X x LDS"); MM(2, "mm LDS x global Y'"); MM(3, "mm LDS' x LDS");
    MM(4, "acc_store+load LDS + 2 barriers"); MM(5, "matvec_gT (global, coalesced)"); MM(6, "block_dot3");
    MM(7, "matvec_lds"); MM(8, "acc_add_mat (global)"); MM(9, "acc_store_tri (global)");
#define ST(what, name) hipFuncSetAttribute((const void*)t_step<what>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
    run(name, [&] { hipLaunchKernelGGL(t_step<what>, dim3(256), dim3(256), lds, 0, M, M, out); });
    ST(1, "step: 2 mm only"); ST(6, "step: 2 GJ on const input"); ST(7, "step: 2 mm + 2 GJ (const input)"); ST(3, "step: 2 mm + 2 GJ (chained data)");
    hipError_t err = hipGetLastError(); printf("last error: %s\n", hipGetErrorString(err));
    return 0;
}
