// membench.hip — calibrates what the MI355X memory system delivers for the access mixes of the
// LGSSM kernels (pure read, pure write, copy, read:write = 0.7:1), 16 B per lane, streaming.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_fill(double2* __restrict__ o, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) o[i] = make_double2(1.0, 2.0);
}
typedef double d2v __attribute__((ext_vector_type(2)));
__global__ void k_fill_nt(d2v* __restrict__ o, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    d2v v = {1.0, 2.0};
    for (; i < n; i += st) __builtin_nontemporal_store(v, o + i);
}
__global__ void k_mix_streams_nt(const d2v* __restrict__ a, d2v* __restrict__ o, size_t per_wave) {
    size_t w = blockIdx.x;
    const d2v* ap = a + w * per_wave * 7 * 64;
    d2v* op = o + w * per_wave * 10 * 64;
    for (size_t u = 0; u < per_wave; ++u) {
        d2v acc = {0, 0};
        for (int k = 0; k < 7; ++k) { d2v v = __builtin_nontemporal_load(ap + (u * 7 + k) * 64 + threadIdx.x); acc += v; }
        for (int k = 0; k < 10; ++k) __builtin_nontemporal_store(acc, op + (u * 10 + k) * 64 + threadIdx.x);
    }
}
__global__ void k_read(const double2* __restrict__ a, size_t n, double* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    double s = 0;
    for (; i < n; i += st) { double2 v = a[i]; s += v.x + v.y; }
    if (s == 12345.678) *out = s;
}
__global__ void k_copy(const double2* __restrict__ a, double2* __restrict__ o, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) o[i] = a[i];
}
// per block-iteration: read 7 chunks, write 10 chunks (k_backward's mix)
__global__ void k_mix(const double2* __restrict__ a, double2* __restrict__ o, size_t nunits) {
    size_t u = (size_t)blockIdx.x, st = gridDim.x;
    for (; u < nunits; u += st) {
        double2 acc = make_double2(0, 0);
        for (int k = 0; k < 7; ++k) { double2 v = a[(u * 7 + k) * 64 + threadIdx.x]; acc.x += v.x; acc.y += v.y; }
        for (int k = 0; k < 10; ++k) o[(u * 10 + k) * 64 + threadIdx.x] = acc;
    }
}
// the same with each wave walking its own far-apart stream (like one segment per wave)
__global__ void k_mix_streams(const double2* __restrict__ a, double2* __restrict__ o, size_t per_wave) {
    size_t w = blockIdx.x;
    const double2* ap = a + w * per_wave * 7 * 64;
    double2* op = o + w * per_wave * 10 * 64;
    for (size_t u = 0; u < per_wave; ++u) {
        double2 acc = make_double2(0, 0);
        for (int k = 0; k < 7; ++k) { double2 v = ap[(u * 7 + k) * 64 + threadIdx.x]; acc.x += v.x; acc.y += v.y; }
        for (int k = 0; k < 10; ++k) op[(u * 10 + k) * 64 + threadIdx.x] = acc;
    }
}
int main() {
    const size_t GB = 1ull << 30;
    size_t nbytes = 12 * GB, n = nbytes / 16;
    double2 *a, *b; double* out;
    CK(hipMalloc(&a, nbytes)); CK(hipMalloc(&b, nbytes * 10 / 7 + (1 << 20))); CK(hipMalloc(&out, 8));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto t = [&](const char* name, double bytes, auto f) {
        f(); hipDeviceSynchronize();
        float best = 1e9;
        for (int r = 0; r < 5; ++r) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
        printf("%-28s %8.3f ms  %7.1f GB/s\n", name, best, bytes / best / 1e6);
    };
    for (int blocks : {2048, 8192}) {
        printf("grid %d x 256\n", blocks);
        t("fill", (double)nbytes, [&] { hipLaunchKernelGGL(k_fill, dim3(blocks), dim3(256), 0, 0, a, n); });
        t("fill nontemporal", (double)nbytes, [&] { hipLaunchKernelGGL(k_fill_nt, dim3(blocks), dim3(256), 0, 0, (d2v*)a, n); });
        t("read", (double)nbytes, [&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, n, out); });
        t("copy (r+w)", 2.0 * nbytes, [&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, b, n); });
    }
    size_t nunits = n / (7 * 64);
    for (int blocks : {1024, 2048, 4096, 16384}) {
        char nm[64]; snprintf(nm, 64, "mix 7r:10w grid-stride %d", blocks);
        t(nm, (double)nunits * 17 * 1024, [&] { hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(64), 0, 0, a, b, nunits); });
    }
    for (int waves : {1024, 2048, 4096}) {
        size_t per = nunits / waves;
        char nm[64]; snprintf(nm, 64, "mix 7r:10w %d streams", waves);
        t(nm, (double)per * waves * 17 * 1024, [&] { hipLaunchKernelGGL(k_mix_streams, dim3(waves), dim3(64), 0, 0, a, b, per); });
    }
    for (int waves : {2048, 4096}) {
        size_t per = nunits / waves;
        char nm[64]; snprintf(nm, 64, "mix 7r:10w %d streams nt", waves);
        t(nm, (double)per * waves * 17 * 1024, [&] { hipLaunchKernelGGL(k_mix_streams_nt, dim3(waves), dim3(64), 0, 0, (const d2v*)a, (d2v*)b, per); });
    }
    return 0;
}
