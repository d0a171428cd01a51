// fetch_calib.hip — a known byte count through the access pattern of the node-array executor's register kernels (8 bytes per lane, unit stride across the
// lanes of a wavefront, a fresh 512-byte line group per instruction) and, for comparison, through the 16 B/lane streaming reads of the LGSSM kernels: run under
// `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`, the ratio true bytes ÷ (counter × 1024) is the factor that turns the counter into bytes for THAT pattern
// (the guide's ×2 correction on gfx950 is stated for 16 B/lane streams: MI355X_MICROARCH.md, HBM section — "other widths: calibrate").
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_calib_read8(const double* __restrict__ a, size_t n, double* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    double s = 0;
    for (; i < n; i += st) s += a[i];
    if (s == 12345.678) *out = s;
}
__global__ void k_calib_read16(const double2* __restrict__ a, size_t n, double* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    double s = 0;
    for (; i < n; i += st) { double2 v = a[i]; s += v.x + v.y; }
    if (s == 12345.678) *out = s;
}
__global__ void k_calib_write8(double* __restrict__ o, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) o[i] = 1.0;
}
// the executor's pattern proper: a lane walks K slots of its replica, slot k of replica r at (k·RS + r): every load instruction of a wavefront is 512 contiguous bytes
__global__ void k_calib_slots8(const double* __restrict__ a, size_t RS, int K, double* out) {
    size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double s = 0;
    if (r < RS)
        for (int k = 0; k < K; ++k) s += a[(size_t)k * RS + r];
    if (s == 12345.678) *out = s;
}
int main() {
    const size_t bytes = 4ull << 30, n = bytes / 8;
    double *a, *out;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) return 1;
    hipMemset(a, 0, bytes);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_calib_read8, dim3(4096), dim3(256), 0, 0, a, n, out);
        hipLaunchKernelGGL(k_calib_read16, dim3(4096), dim3(256), 0, 0, (const double2*)a, n / 2, out);
        hipLaunchKernelGGL(k_calib_write8, dim3(4096), dim3(256), 0, 0, a, n);
        hipLaunchKernelGGL(k_calib_slots8, dim3(65536 / 64), dim3(64), 0, 0, a, (size_t)65536, (int)(n / 65536), out);
    }
    hipDeviceSynchronize();
    printf("true bytes per launch: %zu (every kernel)\n", bytes);
    return 0;
}
