// scratch_first_touch.hip — what the FIRST launch of a kernel with a private segment costs on this runtime (the queue's scratch space is
// provisioned on demand), against the first launch of a kernel without one and against later launches.  Build: hipcc --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
template <int N>
__global__ void __launch_bounds__(256) k_scratch(double* out, int n) {
    double buf[N];   // indexed dynamically: lives in the private segment
    for (int i = 0; i < N; ++i) buf[i] = out[(threadIdx.x + i) % 64];
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += buf[(i * 7 + threadIdx.x) % N];
    out[64 + threadIdx.x % 64] = s;
}
__global__ void k_plain(double* out) { out[threadIdx.x % 64] = 1.0; }
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <class F>
static void timed(const char* what, F f) {
    const double t0 = now();
    f();
    (void)hipDeviceSynchronize();
    std::printf("%-44s %9.3f ms\n", what, now() - t0);
}
int main() {
    double* d;
    timed("hipMalloc (runtime init)", [&] { (void)hipMalloc(&d, 4096); (void)hipMemset(d, 0, 4096); });
    timed("plain kernel, first launch (code object)", [&] { hipLaunchKernelGGL(k_plain, dim3(1), dim3(64), 0, 0, d); });
    timed("plain kernel, second launch", [&] { hipLaunchKernelGGL(k_plain, dim3(1), dim3(64), 0, 0, d); });
    timed("32 doubles of scratch, 1 workgroup, first", [&] { hipLaunchKernelGGL(k_scratch<32>, dim3(1), dim3(256), 0, 0, d, 5); });
    timed("32 doubles of scratch, 1 workgroup, second", [&] { hipLaunchKernelGGL(k_scratch<32>, dim3(1), dim3(256), 0, 0, d, 5); });
    timed("40 doubles (320 B), 1 workgroup, first", [&] { hipLaunchKernelGGL(k_scratch<40>, dim3(1), dim3(256), 0, 0, d, 5); });
    timed("40 doubles (320 B), 1 workgroup, second", [&] { hipLaunchKernelGGL(k_scratch<40>, dim3(1), dim3(256), 0, 0, d, 5); });
    timed("40 doubles, 2048 workgroups, first", [&] { hipLaunchKernelGGL(k_scratch<40>, dim3(2048), dim3(256), 0, 0, d, 5); });
    timed("40 doubles, 2048 workgroups, second", [&] { hipLaunchKernelGGL(k_scratch<40>, dim3(2048), dim3(256), 0, 0, d, 5); });
    timed("130 doubles (1040 B), 1 workgroup, first", [&] { hipLaunchKernelGGL(k_scratch<130>, dim3(1), dim3(256), 0, 0, d, 5); });
    timed("130 doubles (1040 B), 1 workgroup, second", [&] { hipLaunchKernelGGL(k_scratch<130>, dim3(1), dim3(256), 0, 0, d, 5); });
    timed("40 doubles again after the larger one", [&] { hipLaunchKernelGGL(k_scratch<40>, dim3(1), dim3(256), 0, 0, d, 5); });
    hipStream_t s;
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    timed("40 doubles on a NEW stream, first", [&] { hipLaunchKernelGGL(k_scratch<40>, dim3(1), dim3(256), 0, s, d, 5); });
    timed("40 doubles on the new stream, second", [&] { hipLaunchKernelGGL(k_scratch<40>, dim3(1), dim3(256), 0, s, d, 5); });
    void* big;
    timed("hipMalloc 1 GiB", [&] { (void)hipMalloc(&big, 1ull << 30); });
    timed("hipMemset 1 GiB (first touch)", [&] { (void)hipMemset(big, 0, 1ull << 30); });
    timed("hipFree 1 GiB", [&] { (void)hipFree(big); });
    timed("hipMalloc 1 GiB again", [&] { (void)hipMalloc(&big, 1ull << 30); });
    timed("hipMalloc 4 B", [&] { void* q; (void)hipMalloc(&q, 4); (void)hipFree(q); });
    return 0;
}
