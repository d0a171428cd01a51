// inv_micro.hip — the SPD inverse of the MFMA path in isolation: blk_inverse (16×16 panels, round 3) against gj_inverse
// (rank-4 sweep, rounds 1–2).  Correctness against a long-double Gauss–Jordan on the host for d = 16, 32, 48, 64 — well scaled,
// scaled by 10⁻⁶ / 10⁶, and with twelve decades between the diagonal entries — detection of an indefinite matrix, and timing
// with one, two and four workgroups per CU.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -mllvm -amdgpu-mfma-vgpr-form -o scripts/inv_micro scripts/inv_micro.hip
#include "gj_inverse_legacy.hpp"   // (includes csrc/dense_kernels.hpp)
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
using namespace rxhip;
constexpr int REP = 100;
extern __shared__ __attribute__((aligned(16))) double smem[];

template <int NT, int WHICH>
__global__ void __launch_bounds__(64 * NT) t_inv(const double* M, double* out, double* ld, int rep) {
    constexpr int D = 16 * NT;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    Acc<NT> a, b;
    acc_load<NT>(a, M, D, w, lane);
    LogProd lp;
    bool ok = true;
    for (int r = 0; r < rep; ++r) {
        b = a;
        if (r > 0) lp = LogProd();
        ok = (WHICH ? blk_inverse<NT>(b, smem, w, lane, lp) : gj_inverse<NT>(b, smem, smem, w, lane, lp)) && ok;
        a.v[0][0] += 1e-300 * b.v[0][1];
    }
    acc_store<NT>(b, out + (size_t)blockIdx.x * D * D, D, w, lane);
    if (tid == 0) { ld[2 * blockIdx.x] = lp.value(); ld[2 * blockIdx.x + 1] = ok ? 1.0 : 0.0; }
}

static bool host_inv(int n, const std::vector<double>& A, std::vector<double>& out, double& logdet) {
    std::vector<long double> m((size_t)n * 2 * n, 0.0L);
    for (int i = 0; i < n; ++i) { for (int j = 0; j < n; ++j) m[(size_t)i * 2 * n + j] = A[(size_t)i * n + j]; m[(size_t)i * 2 * n + n + i] = 1.0L; }
    long double ld = 0.0L;
    for (int p = 0; p < n; ++p) {
        const long double pv = m[(size_t)p * 2 * n + p];
        if (!(pv > 0.0L)) return false;
        ld += logl(pv);
        for (int j = 0; j < 2 * n; ++j) m[(size_t)p * 2 * n + j] /= pv;
        for (int i = 0; i < n; ++i) if (i != p) { const long double f = m[(size_t)i * 2 * n + p]; if (f != 0.0L) for (int j = 0; j < 2 * n; ++j) m[(size_t)i * 2 * n + j] -= f * m[(size_t)p * 2 * n + j]; }
    }
    out.resize((size_t)n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) out[(size_t)i * n + j] = (double)m[(size_t)i * 2 * n + n + j];
    logdet = (double)ld;
    return true;
}

template <int NT>
static void run_nt(double* dM, double* dout, double* dld) {
    constexpr int D = 16 * NT;
    const size_t lds = sizeof(double) * (size_t)(blk_scratch_doubles(NT) > 8 * D ? blk_scratch_doubles(NT) : 8 * D);
    std::mt19937_64 g(1234 + NT);
    std::normal_distribution<double> nd;
    std::uniform_real_distribution<double> ud(-3.0, 3.0);
    hipFuncSetAttribute((const void*)t_inv<NT, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)t_inv<NT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    std::vector<double> keep;
    for (int cas = 0; cas < 5; ++cas) {
        std::vector<double> G((size_t)D * D), A((size_t)D * D), sv(D, 1.0);
        for (auto& x : G) x = nd(g);
        const double scale = cas == 1 ? 1e-6 : cas == 2 ? 1e6 : 1.0;
        if (cas == 3) for (auto& x : sv) x = pow(10.0, ud(g));
        for (int i = 0; i < D; ++i) for (int j = 0; j <= i; ++j) {
            double s = i == j ? (double)D : 0.0;
            for (int k = 0; k < D; ++k) s += G[(size_t)i * D + k] * G[(size_t)j * D + k];
            s *= scale * sv[i] * sv[j];
            A[(size_t)i * D + j] = A[(size_t)j * D + i] = s;
        }
        if (cas == 4) A[(size_t)(D - 3) * D + (D - 3)] = -1.0;  // indefinite
        if (cas == 0) keep = A;
        std::vector<double> ref;
        double ldref = 0.0;
        const bool spd = host_inv(D, A, ref, ldref);
        hipMemcpy(dM, A.data(), sizeof(double) * D * D, hipMemcpyHostToDevice);
        for (int which = 0; which < 2; ++which) {
            if (which) hipLaunchKernelGGL((t_inv<NT, 1>), dim3(1), dim3(64 * NT), lds, 0, dM, dout, dld, 1);
            else hipLaunchKernelGGL((t_inv<NT, 0>), dim3(1), dim3(64 * NT), lds, 0, dM, dout, dld, 1);
            std::vector<double> res((size_t)D * D);
            double l2[2];
            hipMemcpy(res.data(), dout, sizeof(double) * D * D, hipMemcpyDeviceToHost);
            hipMemcpy(l2, dld, sizeof(l2), hipMemcpyDeviceToHost);
            double err = 0.0, asym = 0.0;
            if (spd) for (int i = 0; i < D; ++i) for (int j = 0; j < D; ++j) {
                const double sc = sqrt(ref[(size_t)i * D + i] * ref[(size_t)j * D + j]);
                err = fmax(err, fabs(res[(size_t)i * D + j] - ref[(size_t)i * D + j]) / sc);
                asym = fmax(asym, fabs(res[(size_t)i * D + j] - res[(size_t)j * D + i]) / sc);
            }
            printf("d=%2d case %d (%s) %s: element-wise err %.2e  asym %.2e  logdet err %.2e  ok flag %g (expected %d)\n", D, cas,
                   cas == 0 ? "well scaled" : cas == 1 ? "x 1e-6" : cas == 2 ? "x 1e6" : cas == 3 ? "diag 1e-3..1e3" : "indefinite",
                   which ? "blk" : "gj ", err, asym, spd ? fabs(l2[0] - ldref) : 0.0, l2[1], (int)spd);
        }
    }
    hipMemcpy(dM, keep.data(), sizeof(double) * D * D, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs : {256, 512, 1024}) for (int which = 0; which < 2; ++which) {
        // LDS per workgroup sized so that `wgs / 256` workgroups fit a CU and no more (the sweep kernels' own occupancy)
        const size_t l = wgs == 256 ? 100 * 1024 : wgs == 512 ? 70 * 1024 : 36 * 1024;
        auto launch = [&] {
            if (which) hipLaunchKernelGGL((t_inv<NT, 1>), dim3(wgs), dim3(64 * NT), l, 0, dM, dout, dld, REP);
            else hipLaunchKernelGGL((t_inv<NT, 0>), dim3(wgs), dim3(64 * NT), l, 0, dM, dout, dld, REP);
        };
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("d=%2d %s  %4d workgroups: %8.2f us per inverse per workgroup  (%.2f us per inverse and CU)\n", D, which ? "blk" : "gj ", wgs, ms * 1e3 / REP,
               ms * 1e3 / REP / (wgs / 256.0));
    }
}

int main() {
    double *dM, *dout, *dld;
    hipMalloc(&dM, 64 * 64 * 8); hipMalloc(&dout, 1024ull * 64 * 64 * 8); hipMalloc(&dld, 1024 * 2 * 8);
    run_nt<1>(dM, dout, dld);
    run_nt<2>(dM, dout, dld);
    run_nt<3>(dM, dout, dld);
    run_nt<4>(dM, dout, dld);
    printf("last error: %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
