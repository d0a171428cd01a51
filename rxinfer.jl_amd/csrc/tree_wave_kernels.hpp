// tree_wave_kernels.hpp — the node-array executor for dimensions 33 … 64: a WORKGROUP of four wavefronts per (op, replica), matrices staged in LDS.
// (Round 5 wrote it for everything above 8, one wavefront per item; since round 6 dimensions up to 32 run on the register tiles of tree_tile_kernels.hpp — the
// items here turned out instruction-issue bound — and the 16 / 32 instances of this file are what the host emulation compiles, tests/test_tree_wave_host.py.)
//
// The lane-per-(op, replica) kernels of tree_kernels.hpp keep a rule's matrices in registers, which ends at 8×8 (and spills from 5×5 on).  Here the same op
// tables (same opcodes, same word layout, same replica-fastest storage: the host side does not know which kernel runs them) are evaluated by one wavefront
// per work item: a rule's operands are loaded into four d×d LDS tiles (leading dimension dmax + 1: odd, so that a column walk meets 32 different banks) and
// eight d-vectors, every primitive is a loop of the 64 lanes over the elements of its result, and a wavefront-scope fence orders a primitive's LDS writes
// before the next one's reads (one wavefront executes its LDS instructions in order; no s_barrier, so that workgroups of several wavefronts would be legal).
// The inverse is an in-place Gauss–Jordan sweep without pivoting (for a symmetric positive definite matrix the pivots are the Cholesky pivots squared:
// positive, and their logs sum to the log-determinant); products run on the matrix cores, one v_mfma_f64_16x16x4_f64 per 16×16 tile of the result and four
// columns of the left operand, operands read from the LDS tiles (mm_mfma; the 4×4 register tiles of mm_tiles are what the host emulation compiles).
// LDS per wavefront: (4·dmax·(dmax + 1) + 8·dmax)·8 bytes — 9.7 KB at d = 16, 35 KB at 32, 137 KB at 64 (one wavefront per CU: it runs, it is not fast).
//
// RXHIP_HOST_EMUL (tests/ only): the same rule bodies compiled for the host with a "wavefront" of one lane, so that every op can be checked against the
// register implementation of tree_kernels.hpp without a GPU (tests/test_tree_wave_host.py).  The product never defines it.
#pragma once
#include "tree_kernels.hpp"

namespace rxhip {
namespace tree {
namespace wave {

constexpr int NBUF = 4, NVEC = 8;
#ifdef RXHIP_HOST_EMUL
constexpr int DMAX_WAVE = 64;
#endif

#ifdef RXHIP_HOST_EMUL
constexpr int WL = 1;
static double wlds[NBUF * DMAX_WAVE * (DMAX_WAVE + 1) + NVEC * DMAX_WAVE];
__device__ __forceinline__ int w_lane() { return 0; }
__device__ __forceinline__ void w_sync() {}
__device__ __forceinline__ double w_sum(double x, int) { return x; }
#else
// NW wavefronts work on one item (a workgroup of 64·NW threads): one up to 16×16 — the LDS then holds 16 items per CU and a wavefront each keeps the SIMDs
// busy — four above: at d = 64 the LDS holds ONE item per CU, and a single wavefront walking 4096-element tiles 64 elements at a time with an LDS round trip
// per step left three SIMDs idle and the fourth waiting (≈ 100 µs per rule).  Element loops stride by the workgroup, a product's tile rows and the
// inverse's tile rows go one to a wavefront.
#ifdef RXHIP_TU_DC
constexpr int NW = RXHIP_TU_DC > 32 ? 4 : RXHIP_TU_DC > 16 ? 2 : 1;
#else
constexpr int NW = 1;
#endif
constexpr int WL = 64 * NW;
extern __shared__ double wlds[];
__device__ __forceinline__ int w_lane() { return (int)threadIdx.x; }
// orders the work item's earlier LDS / global accesses before its later ones as seen by all of its lanes (one wavefront: a fence; several: a barrier)
__device__ __forceinline__ void w_sync() {
    if (NW == 1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else
        __syncthreads();
}
__device__ __forceinline__ double w_sum(double x, int scratch) {   // the same value in every lane, summed in a fixed order (butterfly, then the wavefronts in order)
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) x += __shfl_xor(x, m, 64);
    if (NW == 1) return x;
    if ((threadIdx.x & 63) == 0) wlds[scratch + (threadIdx.x >> 6)] = x;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += wlds[scratch + w];
    __syncthreads();
    return s;
}
#endif

// (+ above 16: the two panels of the blocked inverse — 4 pivot rows of 16·NT + 1 and 16·NT rows of 5, NT = tiles per side of the dimension class)
inline int wave_nt(int dmax) { return dmax <= 16 ? 1 : dmax <= 32 ? 2 : 4; }
inline size_t lds_bytes(int dmax) {
    const size_t nt = (size_t)wave_nt(dmax);
    return sizeof(double) * ((size_t)NBUF * dmax * (dmax | 1) + (size_t)NVEC * dmax + 4 * (16 * nt + 1) + 16 * nt * 5);
}

// the wavefront's workspace: offsets (in doubles) into wlds
struct Ctx {
    int lane, LD, dmax, V;
    __device__ __forceinline__ int M(int k) const { return k * dmax * LD; }
    __device__ __forceinline__ int v(int k) const { return V + k * dmax; }
    __device__ __forceinline__ int S1() const { return V + NVEC * dmax; }             // blocked inverse: the pivot block's 4 rows (stride 16·NT + 1), then its 4 columns (stride 5)
};
__device__ __forceinline__ Ctx make_ctx(int dmax) {
    Ctx c;
    c.lane = w_lane();
    c.dmax = dmax;
    c.LD = dmax | 1;
    c.V = NBUF * dmax * c.LD;
    return c;
}

// f(i, j) for every element of a rows×cols block, element e = i·cols + j on lane e mod 64: ONE integer division per call — the (i, j) of a lane's next
// element follow from the last by adding (64 div cols, 64 mod cols) with a carry (a division per element was most of the VALU work of an inverse)
template <class F>
__device__ __forceinline__ void each(const Ctx& c, int rows, int cols, F f) {
    const int qi = WL / cols, qj = WL - qi * cols;
    int i = c.lane / cols, j = c.lane - i * cols;
    while (i < rows) {
        f(i, j);
        i += qi;
        j += qj;
        if (j >= cols) {
            j -= cols;
            ++i;
        }
    }
}
// f(i, j, e) for every element of a packed lower triangle (j <= i, e = i(i+1)/2 + j): the square's walk above, upper half skipped
template <class F>
__device__ __forceinline__ void each_tri(const Ctx& c, int d, F f) {
    each(c, d, d, [&](int i, int j) {
        if (j <= i) f(i, j, i * (i + 1) / 2 + j);
    });
}

__device__ __forceinline__ void l_vec(const Ctx& c, int v, const double* b, long long off, int d, long long RS, long long r) {
    for (int i = c.lane; i < d; i += WL) wlds[v + i] = b[(off + i) * RS + r];
}
__device__ __forceinline__ void l_cvec(const Ctx& c, int v, const double* cp, int d) {
    for (int i = c.lane; i < d; i += WL) wlds[v + i] = cp[i];
}
__device__ __forceinline__ void s_vec(const Ctx& c, double* b, long long off, int d, long long RS, long long r, int v) {
    for (int i = c.lane; i < d; i += WL) b[(off + i) * RS + r] = wlds[v + i];
}
#ifndef RXHIP_HOST_EMUL
// n elements, element e on lane e mod WL: ALL of a lane's loads first — independent, in flight together — then what is done with them.  (A loop that
// loads one element and stores it to LDS waits for the memory system once per iteration: at d = 64 a message was 9 … 33 dependent HBM round trips.)
template <int CH, class L, class S>
__device__ __forceinline__ void batch(const Ctx& c, int n, L load, S sink) {
    double x[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        const int e = c.lane + k * WL;
        x[k] = e < n ? load(e) : 0.0;
    }
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        const int e = c.lane + k * WL;
        if (e < n) sink(e, x[k]);
    }
}
__device__ __forceinline__ void tri_ij(int e, int& i, int& j) {   // packed lower triangle: e = i(i+1)/2 + j, j ≤ i
    i = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
    while (i * (i + 1) / 2 > e) --i;
    while ((i + 1) * (i + 2) / 2 <= e) ++i;
    j = e - i * (i + 1) / 2;
}
#endif
template <int DC>
__device__ __forceinline__ void l_sym(const Ctx& c, int M, const double* b, long long off, int d, long long RS, long long r) {
    const int LD = c.LD;
#ifndef RXHIP_HOST_EMUL
    batch<(DC * (DC + 1) / 2 + WL - 1) / WL>(c, d * (d + 1) / 2, [&](int e) { return b[(off + e) * RS + r]; }, [&](int e, double x) {
        int i, j;
        tri_ij(e, i, j);
        wlds[M + i * LD + j] = x;
        wlds[M + j * LD + i] = x;
    });
#else
    each_tri(c, d, [&](int i, int j, int e) {
        const double x = b[(off + e) * RS + r];
        wlds[M + i * LD + j] = x;
        wlds[M + j * LD + i] = x;
    });
#endif
}
template <int DC>
__device__ __forceinline__ void s_sym(const Ctx& c, double* b, long long off, int d, long long RS, long long r, int M) {
    const int LD = c.LD;
#ifndef RXHIP_HOST_EMUL
    batch<(DC * (DC + 1) / 2 + WL - 1) / WL>(c, d * (d + 1) / 2, [&](int e) {
        int i, j;
        tri_ij(e, i, j);
        return 0.5 * (wlds[M + i * LD + j] + wlds[M + j * LD + i]);
    }, [&](int e, double x) { b[(off + e) * RS + r] = x; });
#else
    each_tri(c, d, [&](int i, int j, int e) { b[(off + e) * RS + r] = 0.5 * (wlds[M + i * LD + j] + wlds[M + j * LD + i]); });
#endif
}
template <int DC>
__device__ __forceinline__ void l_full(const Ctx& c, int M, const double* b, long long off, int d, long long RS, long long r) {
    const int LD = c.LD;
#ifndef RXHIP_HOST_EMUL
    batch<(DC * DC + WL - 1) / WL>(c, d * d, [&](int e) { return b[(off + e) * RS + r]; }, [&](int e, double x) {
        const int i = e / d;
        wlds[M + i * LD + e - i * d] = x;
    });
#else
    each(c, d, d, [&](int i, int j) { wlds[M + i * LD + j] = b[(off + i * d + j) * RS + r]; });
#endif
}
__device__ __forceinline__ void s_full(const Ctx& c, double* b, long long off, int d, long long RS, long long r, int M, double scale) {
    const int LD = c.LD;
    each(c, d, d, [&](int i, int j) { b[(off + i * d + j) * RS + r] = scale * wlds[M + i * LD + j]; });
}
template <int DC>
__device__ __forceinline__ void l_cmat(const Ctx& c, int M, const double* cp, int rows, int cols) {
    const int LD = c.LD;
#ifndef RXHIP_HOST_EMUL
    batch<(DC * DC + WL - 1) / WL>(c, rows * cols, [&](int e) { return cp[e]; }, [&](int e, double x) {
        const int i = e / cols;
        wlds[M + i * LD + e - i * cols] = x;
    });
#else
    each(c, rows, cols, [&](int i, int j) { wlds[M + i * LD + j] = cp[i * cols + j]; });
#endif
}
__device__ __forceinline__ void zero_mat(const Ctx& c, int M, int d) {
    const int LD = c.LD;
    each(c, d, d, [&](int i, int j) { wlds[M + i * LD + j] = 0.0; });
}
// dst += sign · src
__device__ __forceinline__ void add_mat(const Ctx& c, int dst, int src, int d, double sign) {
    const int LD = c.LD;
    each(c, d, d, [&](int i, int j) { wlds[dst + i * LD + j] += sign * wlds[src + i * LD + j]; });
}
__device__ __forceinline__ void add_vec(const Ctx& c, int dst, int src, int d, double sign) {
    for (int i = c.lane; i < d; i += WL) wlds[dst + i] += sign * wlds[src + i];
}

// The sweeps below treat rows and columns differently (row k ← +a_k·/p, column k ← −a_·k/p), so the computed inverse X is symmetric only up to rounding — and the
// SKEW part of its error is of size cond·ε with no structure.  A reader that takes one triangle of X turns that skew part into a symmetric perturbation, and a
// second inversion of (X + Λ) multiplies it by the condition number again: 7e-6 sd at cond 8e6, d = 48, where the mean (X + Xᵀ)/2 gives 4e-10 — as the
// symmetric sweeps of the other two kernel families and LAPACK do (found by scripts/fuzz_executor.py, seed 3902; the model: scripts/sim_sweep_symmetry.py).
__device__ __forceinline__ void symmetrise(const Ctx& c, int A, int d) {
    const int LD = c.LD;
    each(c, d, d, [&](int i, int j) {
        if (i < j) {
            const double m = 0.5 * (wlds[A + i * LD + j] + wlds[A + j * LD + i]);
            wlds[A + i * LD + j] = m;
            wlds[A + j * LD + i] = m;
        }
    });
    w_sync();
}
// in place: A ← A⁻¹ of a symmetric positive definite d×d tile; log|A|; false: a pivot ≤ 0 or not finite.  Scratch: vectors 6 and 7
__device__ __forceinline__ bool spd_inv_lds(const Ctx& c, int A, int d, double& logdet) {
    const int LD = c.LD, rk = c.v(6), ck = c.v(7);
    bool ok = true;
    double ld = 0.0;
    for (int k = 0; k < d; ++k) {
        for (int i = c.lane; i < d; i += WL) {
            wlds[rk + i] = wlds[A + k * LD + i];
            wlds[ck + i] = wlds[A + i * LD + k];
        }
        w_sync();
        const double pv = wlds[rk + k];
        ok = ok && (pv > 0.0) && (pv < 1.0e300);
        ld += log(pv);
        const double ip = 1.0 / pv;
        each(c, d, d, [&](int i, int j) {
            double x;
            if (i == k) x = (j == k) ? ip : wlds[rk + j] * ip;
            else if (j == k) x = -wlds[ck + i] * ip;
            else x = wlds[A + i * LD + j] - wlds[ck + i] * wlds[rk + j] * ip;
            wlds[A + i * LD + j] = x;
        });
        w_sync();
    }
    symmetrise(c, A, d);
    logdet = ld;
    return ok;
}
#ifndef RXHIP_HOST_EMUL
// the same sweep with the matrix in REGISTERS: the 64 lanes form an 8×8 grid, lane (bi, bj) holds the B×B block at (bi·B, bj·B) of the matrix padded to
// 8B with the identity; per pivot the owners publish row k and column k (2·8B doubles of LDS traffic instead of the 3 reads and 1 write per ELEMENT of the
// sweep above — at d = 32 a wavefront is alone on its SIMD and the inverse was a chain of ≈ 64 dependent LDS instructions per pivot).  d ≤ 8B
template <int B>
__device__ __forceinline__ bool spd_inv_blk(const Ctx& c, int A, int d, double& logdet) {
    const int LD = c.LD, rk = c.v(6), ck = c.v(7);
    const int bi = c.lane >> 3, bj = c.lane & 7, i0 = bi * B, j0 = bj * B;
    double a[B][B];
#pragma unroll
    for (int p = 0; p < B; ++p)
#pragma unroll
        for (int q = 0; q < B; ++q) {
            const int i = i0 + p, j = j0 + q;
            a[p][q] = (i < d && j < d) ? wlds[A + i * LD + j] : (i == j ? 1.0 : 0.0);
        }
    bool ok = true;
    double mant = 1.0;   // log|A| = log(Π pivot mantissas) + ln 2 · Σ pivot exponents: ONE logarithm per inverse (a log per pivot was a third of its instructions);
    int expo = 0;        // the product of up to 64 mantissas in [½, 1) is ≥ 2⁻⁶⁴: no rescaling needed
    for (int k = 0; k < d; ++k) {
        const int kb = k / B, kl = k - kb * B;
        if (bi == kb) {
#pragma unroll
            for (int q = 0; q < B; ++q) {
                double v = a[0][q];
#pragma unroll
                for (int p = 1; p < B; ++p) v = (kl == p) ? a[p][q] : v;
                if (j0 + q < d) wlds[rk + j0 + q] = v;
            }
        }
        if (bj == kb) {
#pragma unroll
            for (int p = 0; p < B; ++p) {
                double v = a[p][0];
#pragma unroll
                for (int q = 1; q < B; ++q) v = (kl == q) ? a[p][q] : v;
                if (i0 + p < d) wlds[ck + i0 + p] = v;
            }
        }
        w_sync();
        const double pv = wlds[rk + k];
        ok = ok && (pv > 0.0) && (pv < 1.0e300);
        mant *= __builtin_amdgcn_frexp_mant(pv);
        expo += __builtin_amdgcn_frexp_exp(pv);
        double ip = __builtin_amdgcn_rcp(pv);   // v_rcp_f64 + two Newton steps (5 instructions; the IEEE division expands to ≈ 25)
        ip = fma(fma(-pv, ip, 1.0), ip, ip);
        ip = fma(fma(-pv, ip, 1.0), ip, ip);
        double rr[B], cc[B];
#pragma unroll
        for (int q = 0; q < B; ++q) {
            rr[q] = (j0 + q < d) ? wlds[rk + j0 + q] : 0.0;
            cc[q] = (i0 + q < d) ? wlds[ck + i0 + q] : 0.0;
        }
#pragma unroll
        for (int p = 0; p < B; ++p)
#pragma unroll
            for (int q = 0; q < B; ++q) {
                const int i = i0 + p, j = j0 + q;
                double x;
                if (i == k) x = (j == k) ? ip : rr[q] * ip;
                else if (j == k) x = -cc[p] * ip;
                else x = a[p][q] - cc[p] * rr[q] * ip;
                a[p][q] = x;
            }
        w_sync();   // (the next pivot's row and column overwrite the scratch vectors)
    }
#pragma unroll
    for (int p = 0; p < B; ++p)
#pragma unroll
        for (int q = 0; q < B; ++q)
            if (i0 + p < d && j0 + q < d) wlds[A + (i0 + p) * LD + j0 + q] = a[p][q];
    w_sync();
    symmetrise(c, A, d);
    logdet = log(mant) + 0.69314718055994530942 * (double)expo;
    return ok;
}
#endif
#ifndef RXHIP_HOST_EMUL
#ifndef RXHIP_VIF_GUARD
#define RXHIP_VIF_GUARD 1.0e3
#endif
template <int DC>
__device__ __forceinline__ bool spd_inv_blocked(const Ctx& c, int A, int d, double& logdet);   // (below, behind the matrix-core product it is made of)
#endif
// DC: the kernel instance's dimension class (16: dmax ≤ 16, 32: ≤ 32, 64) — one inverse per instance, so that the 2×2 blocks of the small class do not
// carry the register budget of the 4×4 ones (occupancy: 16 wavefronts per CU fit the LDS at d = 16)
template <int DC>
__device__ __forceinline__ bool spd_inv(const Ctx& c, int A, int d, double& logdet) {
#ifndef RXHIP_HOST_EMUL
#ifdef RXHIP_LDS_INVERSE_AB
    if (RXHIP_LDS_INVERSE_AB == 1) return spd_inv_lds(c, A, d, logdet);
    {   // 2: the symmetric scalar sweep (row k serves as column k)
        const int LD = c.LD, rk = c.v(6);
        bool ok = true;
        double ld = 0.0;
        for (int k = 0; k < d; ++k) {
            for (int i = c.lane; i < d; i += WL) wlds[rk + i] = wlds[A + k * LD + i];
            w_sync();
            const double pv = wlds[rk + k];
            ok = ok && (pv > 0.0) && (pv < 1.0e300);
            ld += log(pv);
            const double ip = 1.0 / pv;
            each(c, d, d, [&](int i, int j) {
                double x;
                if (i == k) x = (j == k) ? -ip : wlds[rk + j] * ip;
                else if (j == k) x = wlds[rk + i] * ip;
                else x = wlds[A + i * LD + j] - wlds[rk + i] * wlds[rk + j] * ip;
                wlds[A + i * LD + j] = x;
            });
            w_sync();
        }
        each(c, d, d, [&](int i, int j) { wlds[A + i * LD + j] = -wlds[A + i * LD + j]; });
        w_sync();
        logdet = ld;
        return ok;
    }
#endif
    if (DC <= 16) return spd_inv_blk<2>(c, A, d, logdet);   // (one wavefront; the blocked sweep at this size — four blocks of one MFMA — measured no faster: 31.3 against 33.2 ms)
    return spd_inv_blocked<DC>(c, A, d, logdet);
#else
    return spd_inv_lds(c, A, d, logdet);
#endif
}
// y = op(M) x: element (i, k) of op(M) at M + i·si + k·sk; y must not be x
__device__ __forceinline__ void matvec(const Ctx& c, int y, int M, int si, int sk, int x, int rows, int cols) {
    for (int i = c.lane; i < rows; i += WL) {
        double s = 0.0;
        for (int k = 0; k < cols; ++k) s += wlds[M + i * si + k * sk] * wlds[x + k];
        wlds[y + i] = s;
    }
    w_sync();
}
// C (m×n) = alpha·C + op(A) (m×kk) op(B) (kk×n), alpha ∈ {0, 1} and sign ∈ {+1, −1} on the product; C is neither A nor B
template <int TM, int TN>
__device__ __forceinline__ void mm_tiles(const Ctx& c, int C, int A, int sai, int sak, int B, int sbk, int sbj, int m, int kk, int n, bool acc_in, double sign) {
    const int LD = c.LD, tm = (m + TM - 1) / TM, tn = (n + TN - 1) / TN;
    for (int t = c.lane; t < tm * tn; t += WL) {
        const int ti = t / tn, i0 = ti * TM, j0 = (t - ti * tn) * TN;
        double acc[TM][TN];
        int ia[TM], jb[TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            ia[a] = A + (i0 + a < m ? i0 + a : m - 1) * sai;
#pragma unroll
            for (int b = 0; b < TN; ++b) acc[a][b] = 0.0;
        }
#pragma unroll
        for (int b = 0; b < TN; ++b) jb[b] = B + (j0 + b < n ? j0 + b : n - 1) * sbj;
        for (int k = 0; k < kk; ++k) {
            double av[TM], bv[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) av[a] = wlds[ia[a] + k * sak];
#pragma unroll
            for (int b = 0; b < TN; ++b) bv[b] = wlds[jb[b] + k * sbk];
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[a][b] = fma(av[a], bv[b], acc[a][b]);
        }
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
                if (i0 + a < m && j0 + b < n) {
                    const int ix = C + (i0 + a) * LD + j0 + b;
                    wlds[ix] = acc_in ? wlds[ix] + sign * acc[a][b] : sign * acc[a][b];
                }
    }
    w_sync();
}
#ifndef RXHIP_HOST_EMUL
// The same product on the matrix cores: one v_mfma_f64_16x16x4_f64 per (16×16 tile of C, 4 columns of op(A)) — lane l feeds op(A)[i0 + (l & 15)][k0 + (l >> 4)]
// and op(B)[k0 + (l >> 4)][j0 + (l & 15)] and holds C[i0 + (l >> 4) + 4r][j0 + (l & 15)], r = 0 … 3 (the layouts of dense_kernels.hpp).  A tile row of C at a
// time: the A operand of a k-step is read once from LDS for all NTJ column tiles (2 LDS reads per MFMA at NTJ = 1, 1.25 at 4, against 8 reads per 16 FMAs of
// the 4×4 register tiles above).  Elements beyond m / kk / n read as zero, whatever the LDS tile holds there.
typedef double mfma_d4 __attribute__((ext_vector_type(4)));
// ldc: row stride of C (default: the tiles'); skip ≥ 0: the tile row and the tile column of that index are left alone (the blocked inverse's rank-16 update)
template <int NTJ>
__device__ __forceinline__ void mm_mfma(const Ctx& c, int C, int A, int sai, int sak, int B, int sbk, int sbj, int m, int kk, int n, bool acc_in, double sign, int ldc = 0, int skip = -1) {
    const int LD = ldc ? ldc : c.LD, wv = c.lane >> 6, il = c.lane & 15, q = (c.lane >> 4) & 3;
    for (int i0 = 16 * wv; i0 < m; i0 += 16 * NW) {   // a tile row to a wavefront
        if (i0 == 16 * skip) continue;
        mfma_d4 acc[NTJ];
#pragma unroll
        for (int t = 0; t < NTJ; ++t) acc[t] = (mfma_d4){0.0, 0.0, 0.0, 0.0};
        const int i = i0 + il;
        const int ia = A + (i < m ? i : m - 1) * sai;
        for (int k0 = 0; k0 < kk; k0 += 4) {
            const int k = k0 + q;
            const bool kv = k < kk;
            const int kc = kv ? k : kk - 1;
            const double av = wlds[ia + kc * sak];
            const double a = (kv && i < m) ? av : 0.0;
#pragma unroll
            for (int t = 0; t < NTJ; ++t) {
                if (16 * t < n && t != skip) {   // (wavefront-uniform)
                    const int j = 16 * t + il;
                    const double bv = wlds[B + kc * sbk + (j < n ? j : n - 1) * sbj];
                    const double b = (kv && j < n) ? bv : 0.0;
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NTJ; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ii = i0 + q + 4 * r, j = 16 * t + il;
                if (ii < m && j < n && t != skip) {
                    const int ix = C + ii * LD + j;
                    wlds[ix] = acc_in ? wlds[ix] + sign * acc[t][r] : sign * acc[t][r];
                }
            }
    }
    w_sync();
}
#endif
#ifndef RXHIP_HOST_EMUL
// The inverse above 16: the same sweep, FOUR pivots at a time, with the matrix in REGISTERS in the accumulator layout of v_mfma_f64_16x16x4_f64 (NT × NT tiles,
// lane l holds rows (l >> 4) + 4r, column l & 15 of every tile; identity beyond d) — a block of four pivots is a rank-4 update of every tile: ONE matrix-core
// instruction per tile.  Per block K = {k0 … k0 + 3}: the owners publish the block's four rows and four columns (LDS), every lane inverts the 4×4 pivot block
// P = A_KK⁻¹ (register Cholesky of tree_kernels.hpp: its pivots are Cholesky pivots of the Schur complement — positive, their logs add up to log|A|), forms the
// operands it feeds — the old column panel A_IK and the new row panel R = P A_K: — and then
//     A_IJ −= A_IK R_J  (MFMA, all tiles),   A_K: ← R,   A_:K ← −A_:K P,   A_KK ← P
// in registers.  Four pivots per block keep the accuracy of the scalar sweep (sixteen lose two to four digits at condition numbers 1e6 … 1e11: measured on
// the host, scripts/sim_blk_inverse.py); d = 64: 16 blocks × 16 MFMA where the scalar sweep was 64 pivots of ≈ 700 VALU instructions and two LDS round trips.
template <int DC>
__device__ __forceinline__ bool spd_inv_blocked(const Ctx& c, int A, int d, double& logdet) {
    constexpr int NT = (DC + 15) / 16, SL = 16 * NT + 1, NTR = (NT + NW - 1) / NW;   // NTR: tile rows a wavefront holds (tile row ti = wv + NW·tl)
    const int LD = c.LD, wv = c.lane >> 6, il = c.lane & 15, q = (c.lane >> 4) & 3, S1 = c.S1(), S2 = S1 + 4 * SL;
    mfma_d4 a[NTR][NT];
#pragma unroll
    for (int tl = 0; tl < NTR; ++tl)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * (wv + NW * tl) + q + 4 * r, j = 16 * tj + il;
                a[tl][tj][r] = (i < d && j < d) ? wlds[A + i * LD + j] : (i == j ? 1.0 : 0.0);
            }
    mfma_d4 a0[NTR][NT];   // the matrix as it came (16 doubles a lane at NT = 4): what the guard below falls back on
#pragma unroll
    for (int tl = 0; tl < NTR; ++tl)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) a0[tl][tj] = a[tl][tj];
    bool ok = true;
    double ld = 0.0;
    auto sel4 = [](double x0, double x1, double x2, double x3, int k) { return k == 0 ? x0 : k == 1 ? x1 : k == 2 ? x2 : x3; };
    for (int k0 = 0; k0 < d; k0 += 4) {   // (a last block that reaches beyond d sweeps identity rows: pivots 1, nothing coupled)
        const int tK = k0 >> 4, m = (k0 & 15) >> 2;
        // the block's rows k0 + q (lane (q, il) of the wavefront that holds tile row tK has them in register m) and columns k0 + (il & 3) (lanes with il >> 2 == m)
#pragma unroll
        for (int tl = 0; tl < NTR; ++tl) {
            const int ti = wv + NW * tl;
            if (ti == tK) {
#pragma unroll
                for (int tj = 0; tj < NT; ++tj) wlds[S1 + q * SL + 16 * tj + il] = sel4(a[tl][tj][0], a[tl][tj][1], a[tl][tj][2], a[tl][tj][3], m);
            }
            if ((il >> 2) == m && ti < NT) {
#pragma unroll
                for (int tj = 0; tj < NT; ++tj)
                    if (tj == tK) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) wlds[S2 + (16 * ti + q + 4 * r) * 5 + (il & 3)] = a[tl][tj][r];
                    }
            }
        }
        w_sync();
        double Pk[4][4], Pi[4][4], l4;
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) Pk[x][y] = wlds[S1 + x * SL + k0 + y];
        ok = tree::spd_inv<4>(Pk, Pi, l4) && ok;
        ld += l4;
        // operands: the OLD column panel for this lane's (row il of its tile row, pivot q), negated; the NEW row panel R[q][·] for its (column il of the tile column, pivot q)
        double Pq[4], bop[NT];   // Pq: row q of P
#pragma unroll
        for (int y = 0; y < 4; ++y) Pq[y] = sel4(Pi[0][y], Pi[1][y], Pi[2][y], Pi[3][y], q);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            double s0 = 0.0;
#pragma unroll
            for (int y = 0; y < 4; ++y) s0 += Pq[y] * wlds[S1 + y * SL + 16 * t + il];
            bop[t] = s0;
        }
        const int y = il & 3;
        double Py[4];   // column y of P
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) Py[cc] = sel4(Pi[cc][0], Pi[cc][1], Pi[cc][2], Pi[cc][3], y);
#pragma unroll
        for (int tl = 0; tl < NTR; ++tl) {
            const int ti = wv + NW * tl;
            if (16 * ti >= d || ti >= NT) continue;   // (wavefront-uniform)
            const double aop = -wlds[S2 + (16 * ti + il) * 5 + q];
#pragma unroll
            for (int tj = 0; tj < NT; ++tj)
                if (16 * tj < d) a[tl][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop[tj], a[tl][tj], 0, 0, 0);
            // the block's own rows (row k0 + q of this lane ← R[q][16 tj + il]), then its columns and the corner
            if (ti == tK) {
#pragma unroll
                for (int tj = 0; tj < NT; ++tj) {
                    const double v = bop[tj];
                    if (m == 0) a[tl][tj][0] = v; else if (m == 1) a[tl][tj][1] = v; else if (m == 2) a[tl][tj][2] = v; else a[tl][tj][3] = v;
                }
            }
            if ((il >> 2) == m) {
#pragma unroll
                for (int tj = 0; tj < NT; ++tj)
                    if (tj == tK) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = 16 * ti + q + 4 * r;
                            double v;
                            if (i >= k0 && i < k0 + 4) v = sel4(Py[0], Py[1], Py[2], Py[3], i - k0);   // the corner: P[i − k0][y]
                            else {                                                                  // −Σ_c A_old[i][k0 + c] P[c][y]
                                v = 0.0;
#pragma unroll
                                for (int cc = 0; cc < 4; ++cc) v -= wlds[S2 + i * 5 + cc] * Py[cc];
                            }
                            a[tl][tj][r] = v;
                        }
                    }
            }
        }
        w_sync();   // (the next block's panels overwrite the scratch)
    }
    // Guard: the four-pivot sweep loses digits on ill-conditioned matrices that a second inversion then multiplies by the condition number again (the chain fuzz,
    // seeds 101839 / 119783: q(B x + c) behind a square B of condition 3e5 wrong by 0.3 sd where the one-pivot sweep is exact to 1e-10; the numpy model of the
    // blocked sweep does NOT show the loss, so the cause is not pinned — measured, not explained).  The scale-free measure of how hard the matrix is: the largest
    // variance inflation a_kk (A⁻¹)_kk = 1 / (1 − R²_k) (≤ the condition number of the correlation matrix; 1 for a diagonal matrix whatever its scaling).  Above
    // RXHIP_VIF_GUARD the inverse is redone from the kept copy with one pivot at a time (spd_inv_lds: ≈ 6× slower at d = 64; the benchmark graphs stay below 10²).
    double hard = 0.0;
#pragma unroll
    for (int tl = 0; tl < NTR; ++tl)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * (wv + NW * tl) + q + 4 * r, j = 16 * tj + il;
                if (i < d && j < d) wlds[A + i * LD + j] = a[tl][tj][r];
                if (i == j && i < d && !(a0[tl][tj][r] * a[tl][tj][r] <= RXHIP_VIF_GUARD)) hard = 1.0;
            }
    w_sync();
    if (ok && w_sum(hard, c.v(7)) > 0.0) {
#pragma unroll
        for (int tl = 0; tl < NTR; ++tl)
#pragma unroll
            for (int tj = 0; tj < NT; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * (wv + NW * tl) + q + 4 * r, j = 16 * tj + il;
                    if (i < d && j < d) wlds[A + i * LD + j] = a0[tl][tj][r];
                }
        w_sync();
        return spd_inv_lds(c, A, d, logdet);
    }
    symmetrise(c, A, d);
    logdet = ld;
    return ok;
}
#endif
template <int DC = 64>
__device__ __forceinline__ void matmul(const Ctx& c, int C, int A, bool ta, int B, bool tb, int m, int kk, int n, bool acc_in = false, double sign = 1.0) {
    const int LD = c.LD;
    const int sai = ta ? 1 : LD, sak = ta ? LD : 1, sbk = tb ? 1 : LD, sbj = tb ? LD : 1;
#ifndef RXHIP_HOST_EMUL
    mm_mfma<(DC + 15) / 16>(c, C, A, sai, sak, B, sbk, sbj, m, kk, n, acc_in, sign);
#else
    if (m * n >= 8 * WL) mm_tiles<4, 4>(c, C, A, sai, sak, B, sbk, sbj, m, kk, n, acc_in, sign);
    else mm_tiles<2, 2>(c, C, A, sai, sak, B, sbk, sbj, m, kk, n, acc_in, sign);
#endif
}
// tr(A B) of two d×d tiles
__device__ __forceinline__ double trace_prod(const Ctx& c, int A, int B, int d) {
    const int LD = c.LD;
    double s = 0.0;
    each(c, d, d, [&](int i, int k) { s += wlds[A + i * LD + k] * wlds[B + k * LD + i]; });
    return w_sum(s, c.v(7));
}

// a message in the form a rule wants, vector → v, matrix → M; a conversion is one inverse (in place) and one product (through vector 5)
template <int DC>
__device__ __forceinline__ bool load_msg(const Ctx& c, const TreeParams& p, int off, bool stored_wp, bool want_wp, int d, long long r, int v, int M) {
    l_vec(c, v, p.msg, off, d, p.es, r * p.rs_msg);
    l_sym<DC>(c, M, p.msg, off + d, d, p.es, r * p.rs_msg);
    w_sync();
    if (stored_wp == want_wp) return true;
    if (stored_wp) {   // the zero of the precision form (a `missing` observation) wanted as moments: tree_kernels.hpp load_msg
        double nz = 0.0;
        for (int i = c.lane; i < d; i += WL) nz += wlds[M + i * c.LD + i] != 0.0 ? 1.0 : 0.0;
        if (w_sum(nz, c.v(7)) == 0.0) {
            for (int i = c.lane; i < d; i += WL) {
                wlds[v + i] = 0.0;
                wlds[M + i * c.LD + i] = T_ABSENT_VARIANCE;
            }
            w_sync();
            return true;
        }
    }
    double ld;
    const bool ok = spd_inv<DC>(c, M, d, ld);
    const int t = c.v(5);
    matvec(c, t, M, c.LD, 1, v, d, d);
    for (int i = c.lane; i < d; i += WL) wlds[v + i] = wlds[t + i];
    w_sync();
    return ok;
}
template <int DC>
__device__ __forceinline__ void store_msg(const Ctx& c, const TreeParams& p, int off, int d, long long r, int v, int M) {
    s_vec(c, p.msg, off, d, p.es, r * p.rs_msg, v);
    s_sym<DC>(c, p.msg, off + d, d, p.es, r * p.rs_msg, M);
}
// Σ (want_sigma) or W = Σ⁻¹ of a Gaussian node into M; (E) log|W|
template <int DC>
__device__ __forceinline__ double load_noise(const Ctx& c, const TreeParams& p, const int* w, int d, long long r, bool want_sigma, int M) {
    const int ps = w[W_PREC];
    double el;
    if (ps >= 0) {
        const int tri = d * (d + 1) / 2;
        l_full<DC>(c, M, p.prec, ps + 1 + tri + (want_sigma ? d * d : 0), d, p.es, r * p.rs_prec);
        el = p.prec[(ps + 1 + tri + 2 * d * d) * p.es + r * p.rs_prec];
    } else {
        const double* cp = p.cpool + w[W_C0];
        l_cmat<DC>(c, M, cp + (want_sigma ? 0 : d * d), d, d);
        el = cp[2 * d * d];
    }
    w_sync();
    return el;
}
__device__ __forceinline__ void load_value(const Ctx& c, const TreeParams& p, int off, bool slot, int d, long long r, int v) {
    if (slot) l_vec(c, v, p.val, off, d, p.es, r * p.rs_val);
    else l_cvec(c, v, p.cpool + off, d);
    w_sync();
}

// A marginal as the second phase reads it — mean → vector vm, covariance → tile MV (want_cov), log|V| returned — of the marginal slot `off`, or (push) of the
// IMAGE of that marginal under the constant d × du matrix at cpool + aoff: the output of `A * x` has the marginal (A m, A V Aᵀ) of x's, so the sweep never forms
// the marginals of such (anonymous) variables from messages (tree_kernels.hpp load_marginal, op for op).  Scratch: vector vt; tiles TA, TB, TC (all four tiles of
// the item when an image's covariance is wanted: callers form images first).  A singular image: log|V| = −∞.
template <int DC>
__device__ __forceinline__ double load_marginal(const Ctx& c, const TreeParams& p, int off, bool push, int aoff, int du, int d, long long r, bool want_cov, int vm, int vt, int MV,
                                                int TA, int TB, int TC, int ldoff = -1) {
    const int LD = c.LD;
    if (!push) {
        l_vec(c, vm, p.marg, off, d, p.es, r * p.rs_marg);
        double ldV = 0.0;
        if (want_cov) {
            l_sym<DC>(c, MV, p.marg, off + d, d, p.es, r * p.rs_marg);
            ldV = p.marg[(long long)(off + d + d * (d + 1) / 2) * p.es + r * p.rs_marg];
        }
        w_sync();
        return ldV;
    }
    l_vec(c, vt, p.marg, off, du, p.es, r * p.rs_marg);
    l_cmat<DC>(c, TA, p.cpool + aoff, d, du);
    if (want_cov) l_sym<DC>(c, TB, p.marg, off + du, du, p.es, r * p.rs_marg);
    w_sync();
    matvec(c, vm, TA, LD, 1, vt, d, du);
    if (!want_cov) return 0.0;
    matmul<DC>(c, TC, TA, false, TB, false, d, du, du);   // A V
    matmul<DC>(c, MV, TC, false, TA, true, d, du, d);     // A V Aᵀ
    if (ldoff >= 0)   // a square map: log|A V Aᵀ| = log|V| + 2 log|det A| (the constant from the host) — no sweep
        return p.marg[(long long)(off + du + du * (du + 1) / 2) * p.es + r * p.rs_marg] + p.cpool[ldoff];
    each(c, d, d, [&](int i, int j) { wlds[TB + i * LD + j] = 0.5 * (wlds[MV + i * LD + j] + wlds[MV + j * LD + i]); });
    w_sync();
    double ld;
    const bool pd = spd_inv<DC>(c, TB, d, ld);
    return pd ? ld : -__builtin_huge_val();    // (the sweep inverts a copy of V; its pivots' logs are log|V|)
}

// the sweep (ops up to OP_MARGINAL): the rules of tree_kernels.hpp's eval_bp, op for op
template <int DC>
__device__ __forceinline__ void eval_bp(const Ctx& c, const TreeParams& p, const int* __restrict__ w, long long r) {
    const int op = w[W_OP], d = w[W_D0], fl = w[W_FLAGS], LD = c.LD;
    const int M0 = c.M(0), M1 = c.M(1), M2 = c.M(2), M3 = c.M(3);
    const int v0 = c.v(0), v1 = c.v(1), v2 = c.v(2);
    bool ok = true;
    switch (op) {
    case OP_DERIVE_MUL: {
        const int d1 = w[W_D1];
        l_cmat<DC>(c, M0, p.cpool + w[W_C0], d, d1);
        load_value(c, p, w[W_VAL], fl & F_VAL_SLOT, d1, r, v0);
        matvec(c, v1, M0, LD, 1, v0, d, d1);
        s_vec(c, p.val, w[W_OUT], d, p.es, r * p.rs_val, v1);
    } break;
    case OP_DERIVE_ADD: {
        load_value(c, p, w[W_VAL], fl & F_VAL_SLOT, d, r, v0);
        load_value(c, p, w[W_VAL2], fl & F_VAL2_SLOT, d, r, v1);
        add_vec(c, v0, v1, d, 1.0);
        w_sync();
        s_vec(c, p.val, w[W_OUT], d, p.es, r * p.rs_val, v0);
    } break;
    case OP_LEAF: {
        if (fl & F_VAL_MARG) {   // a Gaussian node under q(out) q(μ): the value is the MEAN of the other interface's marginal (of the previous iteration)
            l_vec(c, v0, p.marg, w[W_VAL], d, p.es, r * p.rs_marg);
            w_sync();
        } else
            load_value(c, p, w[W_VAL], fl & F_VAL_SLOT, d, r, v0);
        const bool wp = fl & F_OUT_WP;
        load_noise<DC>(c, p, w, d, r, !wp, M0);
        if (wp) {
            matvec(c, v1, M0, LD, 1, v0, d, d);
            if (fl & F_MAY_MISS) {   // a `missing` observation (NaN) sends nothing: the zero of the precision form
                bool miss = false;
                for (int i = 0; i < d; ++i) miss = miss || (wlds[v0 + i] != wlds[v0 + i]);
                if (miss) {
                    w_sync();
                    zero_mat(c, M0, d);
                    for (int i = c.lane; i < d; i += WL) wlds[v1 + i] = 0.0;
                    w_sync();
                }
            }
            store_msg<DC>(c, p, w[W_OUT], d, r, v1, M0);
        } else
            store_msg<DC>(c, p, w[W_OUT], d, r, v0, M0);
    } break;
    case OP_NOISE: {
        const bool wp = fl & F_IN0_WP;
        ok = load_msg<DC>(c, p, w[W_IN0], wp, wp, d, r, v0, M0);
        load_noise<DC>(c, p, w, d, r, !wp, M1);
        if (!wp) {
            add_mat(c, M0, M1, d, 1.0);
            w_sync();
            if (fl & F_OUT_WP) {   // converted once for all its readers (tree_kernels.hpp)
                double ld;
                ok = spd_inv<DC>(c, M0, d, ld) && ok;
                matvec(c, v1, M0, LD, 1, v0, d, d);
                store_msg<DC>(c, p, w[W_OUT], d, r, v1, M0);
            } else
                store_msg<DC>(c, p, w[W_OUT], d, r, v0, M0);
        } else {   // Λ' = Λ (Λ + W)⁻¹ W, ξ' = W (Λ + W)⁻¹ ξ
            each(c, d, d, [&](int i, int j) { wlds[M2 + i * LD + j] = wlds[M0 + i * LD + j] + wlds[M1 + i * LD + j]; });
            w_sync();
            double ld;
            ok = spd_inv<DC>(c, M2, d, ld) && ok;
            matvec(c, v1, M2, LD, 1, v0, d, d);
            matvec(c, v2, M1, LD, 1, v1, d, d);
            matmul<DC>(c, M3, M2, false, M1, false, d, d, d);   // (Λ + W)⁻¹ W
            matmul<DC>(c, M2, M0, false, M3, false, d, d, d);   // Λ (Λ + W)⁻¹ W
            store_msg<DC>(c, p, w[W_OUT], d, r, v2, M2);
        }
    } break;
    case OP_MUL_OUT: {
        const int d1 = w[W_D1];
        ok = load_msg<DC>(c, p, w[W_IN0], fl & F_IN0_WP, false, d1, r, v0, M0);
        l_cmat<DC>(c, M1, p.cpool + w[W_C0], d, d1);
        w_sync();
        matvec(c, v1, M1, LD, 1, v0, d, d1);
        matmul<DC>(c, M2, M1, false, M0, false, d, d1, d1);   // A V
        matmul<DC>(c, M3, M2, false, M1, true, d, d1, d);     // A V Aᵀ
        store_msg<DC>(c, p, w[W_OUT], d, r, v1, M3);
    } break;
    case OP_MUL_IN: {
        const int d1 = w[W_D1];
        ok = load_msg<DC>(c, p, w[W_IN0], fl & F_IN0_WP, true, d, r, v0, M0);
        l_cmat<DC>(c, M1, p.cpool + w[W_C0], d, d1);
        w_sync();
        matvec(c, v1, M1, 1, LD, v0, d1, d);              // Aᵀ ξ
        matmul<DC>(c, M2, M1, true, M0, false, d1, d, d);     // Aᵀ Λ
        matmul<DC>(c, M3, M2, false, M1, false, d1, d, d1);   // Aᵀ Λ A
        store_msg<DC>(c, p, w[W_OUT], d1, r, v1, M3);
    } break;
    case OP_ADD_OUT:
    case OP_ADD_IN: {
        if (op == OP_ADD_IN && (fl & F_IN0_WP)) {   // precision form in, precision form out (tree_kernels.hpp): Λ' = Λo (Λo + W2)⁻¹ W2, ξ' = W2 (Λo + W2)⁻¹ (ξo + ξ2) − ξ2
            ok = load_msg<DC>(c, p, w[W_IN0], true, true, d, r, v0, M0);
            ok = load_msg<DC>(c, p, w[W_IN1], fl & F_IN1_WP, true, d, r, v1, M1) && ok;
            each(c, d, d, [&](int i, int j) { wlds[M2 + i * LD + j] = wlds[M0 + i * LD + j] + wlds[M1 + i * LD + j]; });
            add_vec(c, v0, v1, d, 1.0);
            w_sync();
            double ld;
            ok = spd_inv<DC>(c, M2, d, ld) && ok;
            matvec(c, v2, M2, LD, 1, v0, d, d);
            matvec(c, v0, M1, LD, 1, v2, d, d);
            add_vec(c, v0, v1, d, -1.0);
            matmul<DC>(c, M3, M2, false, M1, false, d, d, d);   // (Λo + W2)⁻¹ W2
            matmul<DC>(c, M2, M0, false, M3, false, d, d, d);   // Λo (Λo + W2)⁻¹ W2
            store_msg<DC>(c, p, w[W_OUT], d, r, v0, M2);
            break;
        }
        ok = load_msg<DC>(c, p, w[W_IN0], fl & F_IN0_WP, false, d, r, v0, M0);
        ok = load_msg<DC>(c, p, w[W_IN1], fl & F_IN1_WP, false, d, r, v1, M1) && ok;
        add_vec(c, v0, v1, d, op == OP_ADD_OUT ? 1.0 : -1.0);
        add_mat(c, M0, M1, d, 1.0);
        w_sync();
        store_msg<DC>(c, p, w[W_OUT], d, r, v0, M0);
    } break;
    case OP_SHIFT: {
        const bool wp = fl & F_IN0_WP;
        ok = load_msg<DC>(c, p, w[W_IN0], wp, wp, d, r, v0, M0);
        load_value(c, p, w[W_VAL], fl & F_VAL_SLOT, d, r, v1);
        const double sg = (fl & F_NEG) ? -1.0 : 1.0;
        if (wp) {
            matvec(c, v2, M0, LD, 1, v1, d, d);
            add_vec(c, v0, v2, d, sg);
        } else
            add_vec(c, v0, v1, d, sg);
        w_sync();
        store_msg<DC>(c, p, w[W_OUT], d, r, v0, M0);
    } break;
    case OP_PRODUCT:
    case OP_MARGINAL: {
        zero_mat(c, M0, d);
        for (int i = c.lane; i < d; i += WL) wlds[v0 + i] = 0.0;
        w_sync();
        const int n = w[W_N];
        const int* lst = p.aux + w[W_LIST];
        bool single = op == OP_MARGINAL && n == 1 && lst[1] == 0;   // (tree_kernels.hpp: the marginal of one moment-form message is the message)
        int at = 0;
        if (op == OP_MARGINAL && (fl & F_MAY_MISS) && !single) {   // (… and of one moment-form message next to `missing` observations)
            int n_mv = 0;
            double info = 0.0;
            for (int q = 0; q < n; ++q) {
                if (lst[2 * q + 1] == 0) {
                    ++n_mv;
                    at = q;
                    continue;
                }
                for (int i = c.lane; i < d; i += WL) info += p.msg[(lst[2 * q] + d + i * (i + 1) / 2 + i) * p.es + r * p.rs_msg] != 0.0 ? 1.0 : 0.0;
            }
            single = n_mv == 1 && w_sum(info, c.v(7)) == 0.0;
        }
        for (int q = 0; q < n; ++q) {   // left to right, in factor order
            if (single && q != at) continue;
            ok = load_msg<DC>(c, p, lst[2 * q], lst[2 * q + 1] != 0, !single, d, r, v1, M1) && ok;
            add_vec(c, v0, v1, d, 1.0);
            add_mat(c, M0, M1, d, 1.0);
            w_sync();
        }
        if (op == OP_PRODUCT) {
            store_msg<DC>(c, p, w[W_OUT], d, r, v0, M0);
        } else {
            double ld;
            ok = spd_inv<DC>(c, M0, d, ld) && ok;
            if (single) {
                w_sync();
                load_msg<DC>(c, p, lst[2 * at], false, false, d, r, v2, M0);
                ld = -ld;
            } else
                matvec(c, v2, M0, LD, 1, v0, d, d);
            s_vec(c, p.marg, w[W_OUT], d, p.es, r * p.rs_marg, v2);
            s_sym<DC>(c, p.marg, w[W_OUT] + d, d, p.es, r * p.rs_marg, M0);
            if (c.lane == 0) p.marg[(w[W_OUT] + d + d * (d + 1) / 2) * p.es + r * p.rs_marg] = -ld;
        }
    } break;
    default: break;
    }
    if (!ok && c.lane == 0) atomicOr(p.status, 1);
}

// the second phase: Bethe terms, residual moments, q(W) updates — tree_kernels.hpp's eval_fe, op for op
template <int DC>
__device__ __forceinline__ void eval_fe(const Ctx& c, const TreeParams& p, const int* __restrict__ w, long long r) {
    const int op = w[W_OP], d = w[W_D0], fl = w[W_FLAGS], LD = c.LD;
    const int M0 = c.M(0), M1 = c.M(1), M2 = c.M(2), M3 = c.M(3);
    const int v0 = c.v(0), v1 = c.v(1), v2 = c.v(2), v3 = c.v(3), v4 = c.v(4);
    bool ok = true;
    switch (op) {
    case OP_FE_NOISE2: {
        // P = Λo + W → M0, S = Λμ + W − W P⁻¹ W → M1, W → M2, P⁻¹W → M3 (see eval_fe of tree_kernels.hpp for the algebra)
        if (w[W_IN0] >= 0) ok = load_msg<DC>(c, p, w[W_IN0], fl & F_IN0_WP, true, d, r, v0, M0);
        else {
            zero_mat(c, M0, d);
            for (int i = c.lane; i < d; i += WL) wlds[v0 + i] = 0.0;
        }
        if (w[W_IN1] >= 0) ok = load_msg<DC>(c, p, w[W_IN1], fl & F_IN1_WP, true, d, r, v1, M1) && ok;
        else {
            zero_mat(c, M1, d);
            for (int i = c.lane; i < d; i += WL) wlds[v1 + i] = 0.0;
        }
        const double el = load_noise<DC>(c, p, w, d, r, false, M2);
        add_mat(c, M0, M2, d, 1.0);
        add_mat(c, M1, M2, d, 1.0);
        w_sync();
        double ldP, ldS;
        ok = spd_inv<DC>(c, M0, d, ldP) && ok;
        matmul<DC>(c, M3, M0, false, M2, false, d, d, d);                // P⁻¹ W
        matmul<DC>(c, M1, M2, true, M3, false, d, d, d, true, -1.0);     // S −= Wᵀ P⁻¹ W
        ok = spd_inv<DC>(c, M1, d, ldS) && ok;
        // m_μ = S⁻¹ (ξ_μ + W P⁻¹ ξ_o) → v4, m_o = P⁻¹ (ξ_o + W m_μ) → v3
        matvec(c, v2, M0, LD, 1, v0, d, d);
        matvec(c, v3, M2, LD, 1, v2, d, d);
        add_vec(c, v3, v1, d, 1.0);
        w_sync();
        matvec(c, v4, M1, LD, 1, v3, d, d);
        matvec(c, v2, M2, LD, 1, v4, d, d);
        add_vec(c, v2, v0, d, 1.0);
        w_sync();
        matvec(c, v3, M0, LD, 1, v2, d, d);
        // E[rrᵀ] = P⁻¹ + (P⁻¹W − I) S⁻¹ (P⁻¹W − I)ᵀ + (m_o − m_μ)(m_o − m_μ)ᵀ, into M0
        for (int i = c.lane; i < d; i += WL) {
            wlds[M3 + i * LD + i] -= 1.0;
            wlds[v3 + i] -= wlds[v4 + i];
        }
        w_sync();
        matmul<DC>(c, M2, M3, false, M1, false, d, d, d);                // D S⁻¹
        matmul<DC>(c, M0, M2, false, M3, true, d, d, d, true, 1.0);      // P⁻¹ += D S⁻¹ Dᵀ
        each(c, d, d, [&](int i, int j) { wlds[M0 + i * LD + j] += wlds[v3 + i] * wlds[v3 + j]; });
        w_sync();
        double term = -0.5 * (2.0 * d * (T_LOG2PI + 1.0) - (ldP + ldS));
        if (fl & F_STAT) s_full(c, p.stat, w[W_C1], d, p.es, r * p.rs_stat, M0, 1.0);
        else {
            load_noise<DC>(c, p, w, d, r, false, M2);
            term += 0.5 * (d * T_LOG2PI - el + trace_prod(c, M2, M0, d));
        }
        if (c.lane == 0) p.term[(long long)w[W_TERM] * p.es + r * p.rs_term] = term;
    } break;
    case OP_MARG_PUSH: {   // the stored marginal of an `A * x` output, formed when a caller asks for it
        const double ldV = load_marginal<DC>(c, p, w[W_IN0], true, w[W_C0], w[W_D1], d, r, true, v0, v1, M0, M1, M2, M3, w[W_IN1]);
        s_vec(c, p.marg, w[W_OUT], d, p.es, r * p.rs_marg, v0);
        s_sym<DC>(c, p.marg, w[W_OUT] + d, d, p.es, r * p.rs_marg, M0);
        if (c.lane == 0) p.marg[(long long)(w[W_OUT] + d + d * (d + 1) / 2) * p.es + r * p.rs_marg] = ldV;
    } break;
    case OP_FE_NOISE2M: {
        // tree_kernels.hpp OP_FE_NOISE2M, op for op: the joint of a Gaussian node's two interfaces from ONE inbound message (side a) and the two marginals —
        // P = L_a + W, log|J| = log|P| − log|V_b|, Cov(a − b) = P⁻¹ + (P⁻¹W − I) V_b (P⁻¹W − I)ᵀ.  V_b → M2 first (an image needs every tile), then P⁻¹ → M0,
        // D = P⁻¹W − I → M3, D V_b → M1, E → M0.
        const double ldVb = load_marginal<DC>(c, p, w[W_VAL2], fl & F_PUSH_B, w[W_IN2], w[W_N], d, r, true, v2, v4, M2, M0, M1, M3, (fl & F_PUSH_B) ? w[W_D1] : -1);   // m_b → v2
        (void)load_marginal<DC>(c, p, w[W_VAL], fl & F_PUSH_A, w[W_IN1], w[W_LIST], d, r, false, v1, v4, M0, M0, M1, M3);             // m_a → v1
        if (w[W_IN0] >= 0) ok = load_msg<DC>(c, p, w[W_IN0], fl & F_IN0_WP, true, d, r, v0, M0);
        else zero_mat(c, M0, d);
        const double el = load_noise<DC>(c, p, w, d, r, false, M1);
        add_mat(c, M0, M1, d, 1.0);
        w_sync();
        double ldP;
        ok = spd_inv<DC>(c, M0, d, ldP) && ok;
        matmul<DC>(c, M3, M0, false, M1, false, d, d, d);                // P⁻¹ W
        for (int i = c.lane; i < d; i += WL) {
            wlds[M3 + i * LD + i] -= 1.0;
            wlds[v1 + i] -= wlds[v2 + i];
        }
        w_sync();
        matmul<DC>(c, M1, M3, false, M2, false, d, d, d);                // D V_b
        matmul<DC>(c, M0, M1, false, M3, true, d, d, d, true, 1.0);      // P⁻¹ += D V_b Dᵀ
        each(c, d, d, [&](int i, int j) { wlds[M0 + i * LD + j] += wlds[v1 + i] * wlds[v1 + j]; });
        w_sync();
        double term = -0.5 * (2.0 * d * (T_LOG2PI + 1.0) - (ldP - ldVb));
        if (fl & F_FOLD_ENT) term += (double)w[W_OUT] * 0.5 * (d * (T_LOG2PI + 1.0) + ldVb);
        if (fl & F_STAT) s_full(c, p.stat, w[W_C1], d, p.es, r * p.rs_stat, M0, 1.0);
        else {
            load_noise<DC>(c, p, w, d, r, false, M2);
            term += 0.5 * (d * T_LOG2PI - el + trace_prod(c, M2, M0, d));
        }
        if (c.lane == 0) p.term[(long long)w[W_TERM] * p.es + r * p.rs_term] = term;
    } break;
    case OP_FE_NOISE_MF: {   // a Gaussian node under q(out) q(μ): E[rrᵀ] = V_out + V_μ + (m_out − m_μ)(m_out − m_μ)ᵀ; the entropies go with the variables' terms
        (void)load_marginal<DC>(c, p, w[W_VAL], false, 0, 0, d, r, true, v0, v4, M0, M1, M2, M3);
        (void)load_marginal<DC>(c, p, w[W_VAL2], false, 0, 0, d, r, true, v1, v4, M2, M1, M1, M3);
        const double el = load_noise<DC>(c, p, w, d, r, false, M1);
        add_vec(c, v0, v1, d, -1.0);
        add_mat(c, M0, M2, d, 1.0);
        w_sync();
        each(c, d, d, [&](int i, int j) { wlds[M0 + i * LD + j] += wlds[v0 + i] * wlds[v0 + j]; });
        w_sync();
        double term = 0.0;
        if (fl & F_STAT) s_full(c, p.stat, w[W_C1], d, p.es, r * p.rs_stat, M0, 1.0);
        else term = 0.5 * (d * T_LOG2PI - el + trace_prod(c, M1, M0, d));
        if (c.lane == 0) p.term[(long long)w[W_TERM] * p.es + r * p.rs_term] = term;
    } break;
    case OP_FE_NOISE1:
    case OP_FE_NOISE0: {
        double H = 0.0;
        if (op == OP_FE_NOISE1) {   // (the marginal first: an image of another marginal needs every tile)
            const double ldV = load_marginal<DC>(c, p, w[W_IN0], fl & F_PUSH_A, w[W_IN1], w[W_D1], d, r, true, v0, v4, M0, M1, M2, M3, (fl & F_PUSH_A) ? w[W_IN2] : -1);
            H = 0.5 * (d * (T_LOG2PI + 1.0) + ldV);
            if (fl & F_FOLD_ENT) H *= (double)(1 - w[W_OUT]);
            load_value(c, p, w[W_VAL], fl & F_VAL_SLOT, d, r, v1);
        } else {
            zero_mat(c, M0, d);
            load_value(c, p, w[W_VAL], fl & F_VAL_SLOT, d, r, v0);
            load_value(c, p, w[W_VAL2], fl & F_VAL2_SLOT, d, r, v1);
        }
        const double el = load_noise<DC>(c, p, w, d, r, false, M1);
        add_vec(c, v0, v1, d, -1.0);
        w_sync();
        bool miss = false;   // a `missing` observation: energy and the predicted value's entropy cancel, −H of the random interface stays
        if (fl & F_MAY_MISS)
            for (int i = 0; i < d; ++i) miss = miss || (wlds[v0 + i] != wlds[v0 + i]);
        each(c, d, d, [&](int i, int j) { wlds[M0 + i * LD + j] += wlds[v0 + i] * wlds[v0 + j]; });
        w_sync();
        double term = -H;
        if (fl & F_STAT) s_full(c, p.stat, w[W_C1], d, p.es, r * p.rs_stat, M0, 1.0);
        else if (!miss) term += 0.5 * (d * T_LOG2PI - el + trace_prod(c, M1, M0, d));
        else (void)trace_prod(c, M1, M0, d);   // (every lane of the item runs the reduction's barriers)
        if (c.lane == 0) p.term[(long long)w[W_TERM] * p.es + r * p.rs_term] = term;
    } break;
    case OP_FE_ENT: {
        double ldV;
        if (fl & F_PUSH_A) ldV = load_marginal<DC>(c, p, w[W_IN0], true, w[W_C0], w[W_D1], d, r, true, v0, v4, M0, M1, M2, M3, w[W_IN1]);
        else ldV = p.marg[(w[W_IN0] + d + d * (d + 1) / 2) * p.es + r * p.rs_marg];
        if (c.lane == 0) p.term[(long long)w[W_TERM] * p.es + r * p.rs_term] = (double)w[W_N] * 0.5 * (d * (T_LOG2PI + 1.0) + ldV);
    } break;
    case OP_FE_ADD2: {   // P = Λ1 + Λo → M0, S = Λ2 + Λo − Λo P⁻¹ Λo → M1, Λo → M2
        if (w[W_IN0] >= 0) ok = load_msg<DC>(c, p, w[W_IN0], fl & F_IN0_WP, true, d, r, v0, M0);
        else zero_mat(c, M0, d);
        if (w[W_IN1] >= 0) ok = load_msg<DC>(c, p, w[W_IN1], fl & F_IN1_WP, true, d, r, v0, M1) && ok;
        else zero_mat(c, M1, d);
        if (w[W_IN2] >= 0) ok = load_msg<DC>(c, p, w[W_IN2], fl & F_IN2_WP, true, d, r, v0, M2) && ok;
        else zero_mat(c, M2, d);
        w_sync();
        add_mat(c, M0, M2, d, 1.0);
        add_mat(c, M1, M2, d, 1.0);
        w_sync();
        double ldP, ldS;
        ok = spd_inv<DC>(c, M0, d, ldP) && ok;
        matmul<DC>(c, M3, M0, false, M2, false, d, d, d);
        matmul<DC>(c, M1, M2, false, M3, false, d, d, d, true, -1.0);
        ok = spd_inv<DC>(c, M1, d, ldS) && ok;
        if (c.lane == 0) p.term[(long long)w[W_TERM] * p.es + r * p.rs_term] = -0.5 * (2.0 * d * (T_LOG2PI + 1.0) - (ldP + ldS));
    } break;
    case OP_SUM_TERMS: {
        if (c.lane == 0) {
            const int n = w[W_N];
            const int* lst = p.aux + w[W_LIST];
            double s = 0.0;
            for (int q = 0; q < n; ++q) s += p.term[(long long)lst[q] * p.es + r * p.rs_term];
            p.term[(long long)w[W_TERM] * p.es + r * p.rs_term] = s;
        }
    } break;
    case OP_PREC_UPDATE: {
        // prior block at c0: ν0 | S0⁻¹ (d²) | log|S0|;  Σ E[rrᵀ] symmetrised → M1, V⁻¹ = S0⁻¹ + that → M2, V → M3
        const double* cp = p.cpool + w[W_C0];
        const double nu0 = cp[0], ldS0 = cp[1 + d * d];
        zero_mat(c, M0, d);
        w_sync();
        const int n = w[W_N];
        const int* lst = p.aux + w[W_LIST];
        for (int q = 0; q < n; ++q) {
            const long long so = lst[q];
            each(c, d, d, [&](int i, int j) { wlds[M0 + i * LD + j] += p.stat[(so + i * d + j) * p.es + r * p.rs_stat]; });
        }
        w_sync();
        each(c, d, d, [&](int i, int j) {
            const double s = 0.5 * (wlds[M0 + i * LD + j] + wlds[M0 + j * LD + i]);
            wlds[M1 + i * LD + j] = s;
            const double vi = cp[1 + i * d + j] + s;
            wlds[M2 + i * LD + j] = vi;
            wlds[M3 + i * LD + j] = vi;
        });
        w_sync();
        double ldVi;
        ok = spd_inv<DC>(c, M3, d, ldVi);
        const double nu = nu0 + (double)n, ldV = -ldVi;
        const int ps = w[W_PREC], tri = d * (d + 1) / 2;
        const double elw = t_mvdigamma(0.5 * nu, d) + d * T_LOG2 + ldV;
        if (c.lane == 0) {
            p.prec[(long long)ps * p.es + r * p.rs_prec] = nu;
            p.prec[(long long)(ps + 1 + tri + 2 * d * d) * p.es + r * p.rs_prec] = elw;
        }
        s_sym<DC>(c, p.prec, ps + 1, d, p.es, r * p.rs_prec, M3);
        s_full(c, p.prec, ps + 1 + tri, d, p.es, r * p.rs_prec, M3, nu);
        s_full(c, p.prec, ps + 1 + tri + d * d, d, p.es, r * p.rs_prec, M2, 1.0 / nu);
        if (p.want_fe) {
            l_cmat<DC>(c, M0, cp + 1, d, d);
            w_sync();
            double F = 0.5 * ((double)n * (d * T_LOG2PI - elw) + nu * trace_prod(c, M3, M1, d));
            F += -(0.5 * (nu0 - d - 1.0) * elw - 0.5 * nu * trace_prod(c, M0, M3, d) - 0.5 * nu0 * d * T_LOG2 - 0.5 * nu0 * ldS0 - t_mvlgamma(0.5 * nu0, d));
            F -= 0.5 * (d + 1.0) * ldV + 0.5 * d * (d + 1.0) * T_LOG2 + t_mvlgamma(0.5 * nu, d) - 0.5 * (nu - d - 1.0) * t_mvdigamma(0.5 * nu, d) + 0.5 * nu * d;
            if (c.lane == 0) p.term[(long long)w[W_TERM] * p.es + r * p.rs_term] = F;
        }
    } break;
    default: break;
    }
    if (!ok && c.lane == 0) atomicOr(p.status, 1);
}

template <int PHASE, int DC>
__device__ __forceinline__ void eval_op(const Ctx& c, const TreeParams& p, const int* __restrict__ w, long long r) {
    if (PHASE == 0) eval_bp<DC>(c, p, w, r);
    else eval_fe<DC>(c, p, w, r);
}

#ifndef RXHIP_HOST_EMUL
// one launch per level: a workgroup (one wavefront) per item (op, replica).  Storage: element k of slot `off` of replica r at (off + k)·es + r·rs_<array>
// (TreeParams): the engines of this kernel store a replica's slots CONTIGUOUSLY (es = 1, rs = the array's doubles per replica), so that the 64 lanes of the
// wavefront that loads a message read 64 consecutive doubles — with the replica-fastest layout of the register kernels (es = RS, rs = 1; still what
// rxhip_rule_eval's one-node schedules use) every lane touched a 128-byte line of its own and 15/16 of the HBM traffic was other replicas' data
template <int PHASE, int DC>
__global__ void __launch_bounds__(WL, DC <= 16 ? 2 : 1) k_wave_ops(TreeParams p, int op0, int op1, int dmax) {
    const Ctx c = make_ctx(dmax);
    const long long total = (long long)(op1 - op0) * p.R;
    for (long long it = blockIdx.x; it < total; it += gridDim.x) {
        const long long o = it / p.R, r = it - o * p.R;
        eval_op<PHASE, DC>(c, p, p.ops + (size_t)(op0 + o) * OP_WORDS, r);
        __syncthreads();   // (one wavefront: a fence — the next item reuses the workspace)
    }
}
// a wavefront owns a replica and walks the ops of the range in order (every op's inputs were written by this wavefront or before the launch)
template <int PHASE, int DC>
__global__ void __launch_bounds__(WL, DC <= 16 ? 2 : 1) k_wave_walk(TreeParams p, int op0, int op1, int dmax) {
    const Ctx c = make_ctx(dmax);
    for (long long r = blockIdx.x; r < p.R; r += gridDim.x)
        for (int o = op0; o < op1; ++o) {
            eval_op<PHASE, DC>(c, p, p.ops + (size_t)o * OP_WORDS, r);
            __syncthreads();   // (a workgroup-scope fence: the op's stores to global memory before the next op's loads by other lanes)
        }
}
#endif

}  // namespace wave
}  // namespace tree
}  // namespace rxhip
