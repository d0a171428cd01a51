// readback_latency.hip — what it costs to learn a status word after a chain of short single-workgroup kernels (the shape of the model-table
// builders): stream sync alone, an async D2H copy into pageable / pinned host memory + stream sync, an event sync first, a kernel that
// writes the word into mapped pinned memory.  Build: hipcc --offload-arch=gfx950 -O2.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_spin(double* out, int iters, int* status) {
    double s = out[threadIdx.x];
    for (int i = 0; i < iters; ++i) s = s * 1.0000001 + 1e-9;
    out[threadIdx.x] = s;
    if (s < 0) atomicOr(status, 1);
}
__global__ void k_copy_word(const int* src, int* dst) { *dst = *src; }
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const bool big = argc > 1;   // first allocate and free 20 GB (what the bench process has done before its d = 64 engines)
    hipStream_t s;
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    double* d; int* dst; int* pinned; int* dpinned;
    (void)hipMalloc(&d, 4096); (void)hipMemset(d, 0, 4096); (void)hipMalloc(&dst, 4);
    (void)hipHostMalloc(&pinned, 4, hipHostMallocMapped); (void)hipHostGetDevicePointer((void**)&dpinned, pinned, 0);
    if (big) { void* b; (void)hipMalloc(&b, 20ull << 30); (void)hipMemset(b, 0, 20ull << 30); (void)hipDeviceSynchronize(); (void)hipFree(b); }
    std::vector<double> hin(20544, 1.0);
    double* din; (void)hipMalloc(&din, hin.size() * 8);
    auto chain = [&](bool upload) {
        (void)hipMemsetAsync(dst, 0, 4, s);
        if (upload) (void)hipMemcpyAsync(din, hin.data(), hin.size() * 8, hipMemcpyHostToDevice, s);   // pageable source
        for (int k = 0; k < 6; ++k) hipLaunchKernelGGL(k_spin, dim3(1), dim3(256), 0, s, d, 60000, dst);   // ≈ 0.8 ms each
    };
    for (int rep = 0; rep < 3; ++rep) {
        int h = 0;
        double t0;
        t0 = now(); chain(false); (void)hipStreamSynchronize(s); std::printf("kernels + stream sync                      %8.3f ms\n", now() - t0);
        t0 = now(); chain(true); (void)hipStreamSynchronize(s); std::printf("pageable upload + kernels + stream sync    %8.3f ms\n", now() - t0);
        t0 = now(); chain(false); (void)hipMemcpyAsync(&h, dst, 4, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s);
        std::printf("kernels + D2H to pageable + stream sync    %8.3f ms\n", now() - t0);
        t0 = now(); chain(true); (void)hipMemcpyAsync(&h, dst, 4, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s);
        std::printf("upload + kernels + D2H pageable + sync     %8.3f ms\n", now() - t0);
        t0 = now(); chain(false); (void)hipMemcpyAsync(pinned, dst, 4, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s);
        std::printf("kernels + D2H to pinned + stream sync      %8.3f ms\n", now() - t0);
        t0 = now(); chain(false); hipLaunchKernelGGL(k_copy_word, dim3(1), dim3(1), 0, s, dst, dpinned); (void)hipStreamSynchronize(s);
        std::printf("kernels + copy kernel to mapped + sync     %8.3f ms   (status %d)\n", now() - t0, *pinned);
        hipEvent_t ev; (void)hipEventCreate(&ev);
        t0 = now(); chain(false); (void)hipEventRecord(ev, s); (void)hipEventSynchronize(ev); (void)hipMemcpyAsync(&h, dst, 4, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s);
        std::printf("kernels + event sync + D2H pageable + sync %8.3f ms\n", now() - t0);
        t0 = now(); chain(false); (void)hipStreamSynchronize(s); (void)hipMemcpy(&h, dst, 4, hipMemcpyDeviceToHost);
        std::printf("kernels + stream sync + blocking hipMemcpy %8.3f ms\n", now() - t0);
        std::printf("--\n");
    }
    return 0;
}
