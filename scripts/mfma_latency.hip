// Latency of v_mfma_f64_16x16x4_f64 on gfx950: a chain of dependent products (SrcC = the previous result), two and four independent chains
// interleaved, and a dependent chain with a VALU read of the result in between (the s_nop 18 hazard).  One wavefront, s_memtime around 256 products.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/mfma_latency scripts/mfma_latency.hip && scripts/mfma_latency
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(double* out, long long* cyc, double a0, double b0) {
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < 64; ++i) {   // 256 dependent products
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    }
    asm volatile("s_nop 15\ns_nop 15" ::"v"(c0));
    long long t1 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < 32; ++i) {   // 256 products, two chains
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    }
    asm volatile("s_nop 15\ns_nop 15" ::"v"(c0), "v"(c1));
    long long t2 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < 16; ++i) {   // 256 products, four chains
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
    }
    asm volatile("s_nop 15\ns_nop 15" ::"v"(c0), "v"(c1), "v"(c2), "v"(c3));
    long long t3 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < 64; ++i) {   // 64 × (chain of 4, then a VALU op on the result that feeds the next chain's B operand)
        v4d c = {0, 0, 0, 0};
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        b = b * 0.5 + c[0] * 1e-30;
        c0 += c;
    }
    long long t4 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; }
    out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + b;
}
int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 4 * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, 1e-3, 1e-3);
    long long h[4]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    printf("s_memtime ticks per product (256 products each): dependent chain %.1f | two chains %.1f | four chains %.1f | chain of 4 + VALU round trip: %.1f per chain\n",
           h[0] / 256.0, h[1] / 256.0, h[2] / 256.0, h[3] / 64.0);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("(device clock %d kHz; __builtin_readcyclecounter = s_memtime)\n", clk);
    return 0;
}
