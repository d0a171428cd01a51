// tree_tile_kernels.hpp — the node-array executor for dimensions 9 … 32: a WAVEFRONT per (op, replica), matrices in REGISTERS in the accumulator layout of
// v_mfma_f64_16x16x4_f64.
//
// The LDS-staged kernels of tree_wave_kernels.hpp (still what runs above 32) spend their time issuing instructions, not waiting: per 16×16 rule ≈ 2 000 VALU and
// 1 200 SALU instructions, most of them index arithmetic around LDS tiles (packed-triangle decoding, divisions by the runtime dimension, bounds) and a
// Gauss–Jordan sweep that round-trips every element through LDS (profiles/r06/tree_wave_levels_d16.txt).  Here a matrix never leaves the register file:
//
//   layout   a matrix padded with ZEROS to 16·NT square is NT × NT tiles of four doubles per lane; lane l = (q, il) = (l >> 4, l & 15) holds
//            M[16 ti + q + 4r][16 tj + il], r = 0 … 3 — what the matrix cores write.  A vector is x[16 t + il] per lane ("I": one double per tile, the same in
//            the four q groups) or x[16 t + q + 4r] ("K": the row index of the matrix layout).
//   product  L·B on the matrix cores takes L as the layout of Lᵀ and B as its own layout, both straight from registers: lane (q, il) feeds
//            L[16 ti + il][16 tk + 4 kq + q] = Lᵀ's register kq of tile (tk, ti).  A symmetric left factor is its own transpose; for A V Aᵀ and Aᵀ Λ A the
//            constant is loaded in the one layout both products want; D V Dᵀ needs only Dᵀ = W P⁻¹ − I.  No operand is ever moved between lanes.
//   inverse  the symmetric sweep (pivot k: a_ij −= a_ik a_kj / a_kk, row and column k scaled, corner −1/a_kk; −A⁻¹ after the last pivot): row k is published
//            through 16·NT doubles of LDS, column k is row k; ≈ 40 instructions per pivot.  Pivots are the squared Cholesky pivots: positive for an SPD
//            matrix, their logs add up to log|A|.  Zero padding survives the sweep (pivots beyond d are not taken), so products need no masking.
//   memory   packed lower triangles and row-major constants are addressed per lane from indices computed once per kernel; every load of a matrix is in
//            flight before the first use.
//
// Same op tables, opcodes, flags and storage as tree_kernels.hpp / tree_wave_kernels.hpp (the host side does not know which kernel runs them); op for op the
// arithmetic of tree_wave_kernels.hpp's eval_bp / eval_fe, whose host emulation stays the differential check of the rule bodies (tests/test_tree_wave_host.py);
// this file is held against the oracle and against the other schedules on the device (tests/test_tree_wave_gpu.py, test_tree_engine_gpu.py).
#pragma once
#include "tree_kernels.hpp"

namespace rxhip {
namespace tree {
namespace tile {

typedef double d4 __attribute__((ext_vector_type(4)));

template <int NT>
struct Mat {
    d4 t[NT][NT];
};
template <int NT>
struct Vec {   // x[t] = v[16 t + il]
    double x[NT];
};
template <int NT>
struct VecK {  // k[t][r] = v[16 t + q + 4 r]
    double k[NT][4];
};
// what a lane knows about itself: its place in the tile and, per element it holds, the BYTE offset 8·e of the packed-triangle element e = hi (hi + 1) / 2 + lo
// (hi < d  ⇔  e < d (d + 1) / 2: the offset alone says whether the element exists)
template <int NT>
struct Lane {
    int q, il;
    unsigned e8[NT][NT][4];
};
template <int NT>
__device__ __forceinline__ Lane<NT> make_lane() {
    Lane<NT> L;
    L.q = (int)(threadIdx.x >> 4) & 3;
    L.il = (int)threadIdx.x & 15;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * ti + L.q + 4 * r, j = 16 * tj + L.il;
                const int hi = i > j ? i : j, lo = i > j ? j : i;
                L.e8[ti][tj][r] = 8u * (unsigned)(hi * (hi + 1) / 2 + lo);
            }
    return L;
}
// element at byte offset off8 of a slot: ES1 (the engines' element-fastest storage, and the constant pool) — scalar base + 32-bit lane offset, no address
// arithmetic; else element stride es (rxhip_rule_eval's replica-fastest one-node schedules)
template <bool ES1>
__device__ __forceinline__ double ldg(const double* b, long long es, unsigned off8) {
    return ES1 ? *(const double*)((const char*)b + off8) : b[(long long)(off8 >> 3) * es];
}
template <bool ES1>
__device__ __forceinline__ void stg(double* b, long long es, unsigned off8, double x) {
    if (ES1) *(double*)((char*)b + off8) = x;
    else b[(long long)(off8 >> 3) * es] = x;
}

// LDS: the row a pivot publishes, in natural order (R[j]) and with the row index of the register layout contiguous (R[16 t + q + 4 r] at 16 t + 4 q + r)
template <int NT>
struct Scratch {
    double nat[16 * NT];
    double perm[16 * NT];
};
__device__ __forceinline__ void w_fence() {   // one wavefront: its LDS instructions execute in order — this only keeps the compiler from moving them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double lane_read(double x, int lane) {   // lane: wavefront-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane), hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double x) {   // the same value in every lane, summed in a fixed order
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) x += __shfl_xor(x, m, 64);
    return x;
}

// ---- memory --------------------------------------------------------------------------------------------------------------------------------------------
// `b` points at element 0 of the slot for this replica
template <int NT, bool ES1>
__device__ __forceinline__ void load_sym(const Lane<NT>& L, const double* b, long long es, int d, Mat<NT>& m) {
    const unsigned n8 = 4u * (unsigned)(d * (d + 1));
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned e8 = L.e8[ti][tj][r];
                const double x = ldg<ES1>(b, es, e8 < n8 ? e8 : n8 - 8u);
                m.t[ti][tj][r] = e8 < n8 ? x : 0.0;
            }
}
// the lower triangle (j ≤ i): the stored message is symmetric by construction
template <int NT, bool ES1>
__device__ __forceinline__ void store_sym(const Lane<NT>& L, double* b, long long es, int d, const Mat<NT>& m) {
    const unsigned n8 = 4u * (unsigned)(d * (d + 1));
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj <= ti; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool low = tj < ti || L.il <= L.q + 4 * r;
                if (low && L.e8[ti][tj][r] < n8) stg<ES1>(b, es, L.e8[ti][tj][r], m.t[ti][tj][r]);
            }
}
// element (i, j), i < rows, j < cols, at element i · si + j · sj of the slot: a row-major rows × cols matrix with (si, sj) = (cols, 1); the transpose of a
// row-major n × rows matrix with (1, rows)
template <int NT, bool ES1>
__device__ __forceinline__ void load_mat(const Lane<NT>& L, const double* b, long long es, int rows, int cols, int si, int sj, Mat<NT>& m) {
    const unsigned last = 8u * (unsigned)((rows - 1) * si + (cols - 1) * sj);
    const unsigned qs = 8u * (unsigned)__mul24(L.q, si), js = 8u * (unsigned)__mul24(L.il, sj);
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * ti + L.q + 4 * r, j = 16 * tj + L.il;
                const bool in = i < rows && j < cols;
                const unsigned e8 = qs + js + 8u * (unsigned)((16 * ti + 4 * r) * si + 16 * tj * sj);
                const double x = ldg<ES1>(b, es, in ? e8 : last);
                m.t[ti][tj][r] = in ? x : 0.0;
            }
}
template <int NT, bool ES1>
__device__ __forceinline__ void store_full(const Lane<NT>& L, double* b, long long es, int d, const Mat<NT>& m, double scale) {
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * ti + L.q + 4 * r, j = 16 * tj + L.il;
                if (i < d && j < d) stg<ES1>(b, es, 8u * (unsigned)(i * d + j), scale * m.t[ti][tj][r]);
            }
}
template <int NT, bool ES1>
__device__ __forceinline__ void load_vec(const Lane<NT>& L, const double* b, long long es, int d, Vec<NT>& v) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int i = 16 * t + L.il;
        const double x = ldg<ES1>(b, es, 8u * (unsigned)(i < d ? i : d - 1));
        v.x[t] = i < d ? x : 0.0;
    }
}
template <int NT, bool ES1>
__device__ __forceinline__ void load_veck(const Lane<NT>& L, const double* b, long long es, int d, VecK<NT>& v) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * t + L.q + 4 * r;
            const double x = ldg<ES1>(b, es, 8u * (unsigned)(i < d ? i : d - 1));
            v.k[t][r] = i < d ? x : 0.0;
        }
}
template <int NT, bool ES1>
__device__ __forceinline__ void store_vec(const Lane<NT>& L, double* b, long long es, int d, const Vec<NT>& v) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int i = 16 * t + L.il;
        if (L.q == 0 && i < d) stg<ES1>(b, es, 8u * (unsigned)i, v.x[t]);
    }
}

// ---- arithmetic ----------------------------------------------------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void zero(Mat<NT>& m) {
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) m.t[ti][tj] = (d4){0.0, 0.0, 0.0, 0.0};
}
template <int NT>
__device__ __forceinline__ void zero(Vec<NT>& v) {
#pragma unroll
    for (int t = 0; t < NT; ++t) v.x[t] = 0.0;
}
template <int NT>
__device__ __forceinline__ void axpy(Mat<NT>& y, double a, const Mat<NT>& x) {   // y += a x
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) y.t[ti][tj] += a * x.t[ti][tj];
}
template <int NT>
__device__ __forceinline__ void axpy(Vec<NT>& y, double a, const Vec<NT>& x) {
#pragma unroll
    for (int t = 0; t < NT; ++t) y.x[t] += a * x.x[t];
}
// m[i][i] += a, i < d
template <int NT>
__device__ __forceinline__ void add_diag(const Lane<NT>& L, Mat<NT>& m, double a, int d) {
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool on = (L.q + 4 * r == L.il) && (16 * ti + L.il < d);
            m.t[ti][ti][r] += on ? a : 0.0;
        }
}
template <int NT>
__device__ __forceinline__ VecK<NT> to_k(const Lane<NT>& L, const Vec<NT>& v) {
    VecK<NT> k;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) k.k[t][r] = __shfl(v.x[t], L.q + 4 * r, 64);   // (lanes 0 … 15 are the group q = 0: lane j holds x[16 t + j])
    return k;
}
// y = Mᵀ x  (M in its own layout; a symmetric M: M x)
template <int NT>
__device__ __forceinline__ Vec<NT> matvec_t(const Mat<NT>& m, const VecK<NT>& x) {
    Vec<NT> y;
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) {
        double s = 0.0;
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) s = fma(m.t[ti][tj][r], x.k[ti][r], s);
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        y.x[tj] = s;
    }
    return y;
}
// E += u vᵀ (u in K, v in I)
template <int NT>
__device__ __forceinline__ void add_outer(Mat<NT>& e, const VecK<NT>& u, const Vec<NT>& v) {
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) e.t[ti][tj][r] = fma(u.k[ti][r], v.x[tj], e.t[ti][tj][r]);
}
// Σ_ij A_ij B_ij  (= tr(A B) when either is symmetric)
template <int NT>
__device__ __forceinline__ double dot(const Mat<NT>& a, const Mat<NT>& b) {
    double s = 0.0;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) s = fma(a.t[ti][tj][r], b.t[ti][tj][r], s);
    return wave_sum(s);
}
// c (+)= sign · L B, L given as the layout of Lᵀ (a symmetric L: itself).  n16: tiles per side that hold anything (wavefront-uniform)
template <int NT>
__device__ __forceinline__ void mul(Mat<NT>& c, const Mat<NT>& lt, const Mat<NT>& b, int n16, bool acc_in = false, double sign = 1.0) {
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
            d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
            if (ti < n16 && tj < n16) {
#pragma unroll
                for (int tk = 0; tk < NT; ++tk)
                    if (tk < n16) {
#pragma unroll
                        for (int kq = 0; kq < 4; ++kq) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(lt.t[tk][ti][kq], b.t[tk][tj][kq], acc, 0, 0, 0);
                    }
            }
            c.t[ti][tj] = acc_in ? c.t[ti][tj] + sign * acc : sign * acc;
        }
}

// in place: a ← a⁻¹ of a symmetric positive definite d × d matrix (zero beyond d, and zero beyond d it stays); log|a|; false: a pivot ≤ 0 or not finite
// (one copy per kernel — called, not inlined: the sweep is ≈ 500 instructions and a rule has up to three — with the matrix passed and returned BY VALUE, in registers)
template <int NT>
struct Inverse {
    Mat<NT> a;
    double logdet;
    int ok;
};
template <int NT>
__device__ __noinline__ Inverse<NT> spd_inv_call(int q, int il, Scratch<NT>* sp, Mat<NT> a, int d) {
    Scratch<NT>& s = *sp;
    struct { int q, il; } L{q, il};
    bool ok = true;
    double mant = 1.0;   // log|A| = log Π mantissas + ln 2 · Σ exponents: one logarithm per inverse (the product of ≤ 32 mantissas in [½, 1) stays normal)
    int expo = 0;
#pragma unroll
    for (int tk = 0; tk < NT; ++tk)
#pragma unroll
        for (int kr = 0; kr < 4; ++kr)
            for (int kq = 0; kq < 4; ++kq) {
                const int kc = 4 * kr + kq, k = 16 * tk + kc;   // the pivot: row (q = kq, register kr) of tile row tk, column il = kc of tile column tk
                if (k >= d) break;
                const int plo = __builtin_amdgcn_readlane(__double2loint(a.t[tk][tk][kr]), 16 * kq + kc), phi = __builtin_amdgcn_readlane(__double2hiint(a.t[tk][tk][kr]), 16 * kq + kc);
                const double pv = __hiloint2double(phi, plo);   // (wavefront-uniform: in scalar registers)
                if (L.q == kq) {
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj) {
                        const double v = a.t[tk][tj][kr];
                        s.nat[16 * tj + L.il] = v;
                        s.perm[16 * tj + 4 * (L.il & 3) + (L.il >> 2)] = v;
                    }
                }
                w_fence();
                double R[NT], C[NT][4];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    R[t] = s.nat[16 * t + L.il];
#pragma unroll
                    for (int r = 0; r < 4; ++r) C[t][r] = s.perm[16 * t + 4 * L.q + r];
                }
                w_fence();
                // sign and exponent of the pivot on the scalar unit: positive, normal, below 2^996 (≈ 1e300); mantissa in [½, 1) for the running product
                const unsigned se = (unsigned)phi >> 20;
                ok = ok && (se - 1u < 2018u);
                expo += (int)se - 1022;
                mant *= __hiloint2double((phi & (int)0x800fffff) | 0x3fe00000, plo);
                double ip = __builtin_amdgcn_rcp(pv);   // v_rcp_f64 + two Newton steps
                ip = fma(fma(-pv, ip, 1.0), ip, ip);
                ip = fma(fma(-pv, ip, 1.0), ip, ip);
                const bool colk = L.il == kc;
#pragma unroll
                for (int tj = 0; tj < NT; ++tj) {
                    const double t = R[tj] * ip;
#pragma unroll
                    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                        for (int r = 0; r < 4; ++r) a.t[ti][tj][r] = fma(-C[ti][r], t, a.t[ti][tj][r]);
                    double tt = t;
                    if (tj == tk && colk) {   // the four lanes of column k: a_ik / p, the corner −1/p (a branch, not eight selects in every lane)
                        __asm__ volatile("" ::: "memory");
#pragma unroll
                        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                            for (int r = 0; r < 4; ++r) a.t[ti][tj][r] = C[ti][r] * ip;
                        tt = -ip;
                    }
                    if (L.q == kq) a.t[tk][tj][kr] = tt;   // row k: R / p
                }
            }
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) a.t[ti][tj] = -a.t[ti][tj];
    Inverse<NT> out;
    out.a = a;
    out.logdet = log(mant) + 0.69314718055994530942 * (double)expo;
    out.ok = ok;
    return out;
}
template <int NT>
__device__ __forceinline__ bool spd_inv(const Lane<NT>& L, Scratch<NT>& s, Mat<NT>& a, int d, double& logdet) {
    const Inverse<NT> r = spd_inv_call<NT>(L.q, L.il, &s, a, d);
    a = r.a;
    logdet = r.logdet;
    return r.ok != 0;
}

// ---- what a rule reads ---------------------------------------------------------------------------------------------------------------------------------
template <int NT>
struct Env {
    const Lane<NT>& L;
    Scratch<NT>& s;
    const TreeParams& p;
    long long r;
    __device__ __forceinline__ const double* msg(int off) const { return p.msg + (long long)off * p.es + r * p.rs_msg; }
    __device__ __forceinline__ double* msg_w(int off) const { return p.msg + (long long)off * p.es + r * p.rs_msg; }
    __device__ __forceinline__ double* marg(int off) const { return p.marg + (long long)off * p.es + r * p.rs_marg; }
    __device__ __forceinline__ double* val(int off) const { return p.val + (long long)off * p.es + r * p.rs_val; }
    __device__ __forceinline__ double* prec(int off) const { return p.prec + (long long)off * p.es + r * p.rs_prec; }
    __device__ __forceinline__ double* stat(int off) const { return p.stat + (long long)off * p.es + r * p.rs_stat; }
    __device__ __forceinline__ double* term(int off) const { return p.term + (long long)off * p.es + r * p.rs_term; }
};
template <int NT>
__device__ __forceinline__ int tiles(int d) { return NT == 1 ? 1 : (d + 15) >> 4; }

// a message in the form a rule wants: a conversion is one inverse and one matrix-vector product
template <int NT, bool ES1>
__device__ __forceinline__ bool load_msg(const Env<NT>& E, int off, bool stored_wp, bool want_wp, int d, Vec<NT>& v, Mat<NT>& M) {
    const double* b = E.msg(off);
    load_vec<NT, ES1>(E.L, b, E.p.es, d, v);
    load_sym<NT, ES1>(E.L, b + (long long)d * E.p.es, E.p.es, d, M);
    if (stored_wp == want_wp) return true;
    if (stored_wp) {   // the zero of the precision form (a `missing` observation) wanted as moments: tree_kernels.hpp load_msg
        bool nz = false;
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) nz = nz || ((E.L.q + 4 * r == E.L.il) && M.t[ti][ti][r] != 0.0);
        if (!__any(nz)) {
            zero<NT>(v);
            zero<NT>(M);
            add_diag<NT>(E.L, M, T_ABSENT_VARIANCE, d);
            return true;
        }
    }
    double ld;
    const bool ok = spd_inv<NT>(E.L, E.s, M, d, ld);
    v = matvec_t<NT>(M, to_k<NT>(E.L, v));
    return ok;
}
template <int NT, bool ES1>
__device__ __forceinline__ void store_msg(const Env<NT>& E, int off, int d, const Vec<NT>& v, const Mat<NT>& M) {
    double* b = E.msg_w(off);
    store_vec<NT, ES1>(E.L, b, E.p.es, d, v);
    store_sym<NT, ES1>(E.L, b + (long long)d * E.p.es, E.p.es, d, M);
}
// Σ (want_sigma) or W = Σ⁻¹ of a Gaussian node; (E) log|W|
template <int NT, bool ES1>
__device__ __forceinline__ double load_noise(const Env<NT>& E, const int* w, int d, bool want_sigma, Mat<NT>& M) {
    const int ps = w[W_PREC];
    if (ps >= 0) {
        const int tri = d * (d + 1) / 2;
        const double* b = E.prec(ps);
        load_mat<NT, ES1>(E.L, b + (long long)(1 + tri + (want_sigma ? d * d : 0)) * E.p.es, E.p.es, d, d, d, 1, M);
        return b[(long long)(1 + tri + 2 * d * d) * E.p.es];
    }
    const double* cp = E.p.cpool + w[W_C0];
    load_mat<NT, true>(E.L, cp + (want_sigma ? 0 : d * d), 1, d, d, d, 1, M);
    return cp[2 * d * d];
}
template <int NT, bool ES1>
__device__ __forceinline__ void load_value(const Env<NT>& E, int off, bool slot, int d, Vec<NT>& v) {
    if (slot) load_vec<NT, ES1>(E.L, E.val(off), E.p.es, d, v);
    else load_vec<NT, true>(E.L, E.p.cpool + off, 1, d, v);
}
template <int NT>
__device__ __forceinline__ bool any_nan(const Vec<NT>& v) {
    bool bad = false;
#pragma unroll
    for (int t = 0; t < NT; ++t) bad = bad || (v.x[t] != v.x[t]);
    return __any(bad);
}

// A marginal as the second phase reads it (tree_wave_kernels.hpp load_marginal): mean, covariance (want_cov), log|V| — of the slot `off`, or (push) of its IMAGE
// under the constant d × du matrix at cpool + aoff: (A m, A V Aᵀ); ldoff ≥ 0: a square map, log|A V Aᵀ| = log|V| + cpool[ldoff]
template <int NT, bool ES1>
__device__ __forceinline__ double load_marginal(const Env<NT>& E, int off, bool push, int aoff, int du, int d, bool want_cov, Vec<NT>& m, Mat<NT>& V, int ldoff = -1) {
    const double* b = E.marg(off);
    const long long es = E.p.es;
    if (!push) {
        load_vec<NT, ES1>(E.L, b, es, d, m);
        if (!want_cov) return 0.0;
        load_sym<NT, ES1>(E.L, b + (long long)d * es, es, d, V);
        return b[(long long)(d + d * (d + 1) / 2) * es];
    }
    Mat<NT> at;   // Aᵀ (du × d): element (i, j) = A[j][i] at j · du + i
    load_mat<NT, true>(E.L, E.p.cpool + aoff, 1, du, d, 1, du, at);
    VecK<NT> mu;
    load_veck<NT, ES1>(E.L, b, es, du, mu);
    m = matvec_t<NT>(at, mu);
    if (!want_cov) return 0.0;
    Mat<NT> Vu, Y;
    load_sym<NT, ES1>(E.L, b + (long long)du * es, es, du, Vu);
    const int n16 = tiles<NT>(d > du ? d : du);
    mul<NT>(Y, Vu, at, n16);    // V Aᵀ
    mul<NT>(V, at, Y, n16);     // A (V Aᵀ)
    if (ldoff >= 0) return b[(long long)(du + du * (du + 1) / 2) * es] + E.p.cpool[ldoff];
    Mat<NT> T = V;
    double ld;
    const bool pd = spd_inv<NT>(E.L, E.s, T, d, ld);
    return pd ? ld : -__builtin_huge_val();
}

// the sweep (ops up to OP_MARGINAL): tree_wave_kernels.hpp eval_bp, op for op
template <int NT, bool ES1>
__device__ __forceinline__ void eval_bp(const Env<NT>& E, const int* __restrict__ w) {
    const int op = w[W_OP], d = w[W_D0], fl = w[W_FLAGS];
    const Lane<NT>& L = E.L;
    const long long es = E.p.es;
    bool ok = true;
    switch (op) {
    case OP_DERIVE_MUL: {
        const int d1 = w[W_D1];
        Mat<NT> at;
        load_mat<NT, true>(L, E.p.cpool + w[W_C0], 1, d1, d, 1, d1, at);
        VecK<NT> x;
        if (fl & F_VAL_SLOT) load_veck<NT, ES1>(L, E.val(w[W_VAL]), es, d1, x);
        else load_veck<NT, true>(L, E.p.cpool + w[W_VAL], 1, d1, x);
        store_vec<NT, ES1>(L, E.val(w[W_OUT]), es, d, matvec_t<NT>(at, x));
    } break;
    case OP_DERIVE_ADD: {
        Vec<NT> a, b;
        load_value<NT, ES1>(E, w[W_VAL], fl & F_VAL_SLOT, d, a);
        load_value<NT, ES1>(E, w[W_VAL2], fl & F_VAL2_SLOT, d, b);
        axpy<NT>(a, 1.0, b);
        store_vec<NT, ES1>(L, E.val(w[W_OUT]), es, d, a);
    } break;
    case OP_LEAF: {
        Vec<NT> v;
        if (fl & F_VAL_MARG) load_vec<NT, ES1>(L, E.marg(w[W_VAL]), es, d, v);   // q(out) q(μ): the MEAN of the other interface's marginal (of the previous iteration)
        else load_value<NT, ES1>(E, w[W_VAL], fl & F_VAL_SLOT, d, v);
        const bool wp = fl & F_OUT_WP;
        Mat<NT> M;
        load_noise<NT, ES1>(E, w, d, !wp, M);
        if (wp) {
            Vec<NT> y = matvec_t<NT>(M, to_k<NT>(L, v));
            if ((fl & F_MAY_MISS) && any_nan<NT>(v)) {   // a `missing` observation sends nothing: the zero of the precision form
                zero<NT>(M);
                zero<NT>(y);
            }
            store_msg<NT, ES1>(E, w[W_OUT], d, y, M);
        } else
            store_msg<NT, ES1>(E, w[W_OUT], d, v, M);
    } break;
    case OP_NOISE: {
        const bool wp = fl & F_IN0_WP;
        Vec<NT> v;
        Mat<NT> M, N;
        ok = load_msg<NT, ES1>(E, w[W_IN0], wp, wp, d, v, M);
        load_noise<NT, ES1>(E, w, d, !wp, N);
        if (!wp) {
            axpy<NT>(M, 1.0, N);
            if (fl & F_OUT_WP) {   // converted once for all its readers
                double ld;
                ok = spd_inv<NT>(L, E.s, M, d, ld) && ok;
                v = matvec_t<NT>(M, to_k<NT>(L, v));
            }
            store_msg<NT, ES1>(E, w[W_OUT], d, v, M);
        } else {   // Λ' = Λ (Λ + W)⁻¹ W, ξ' = W (Λ + W)⁻¹ ξ
            Mat<NT> G = M;
            axpy<NT>(G, 1.0, N);
            double ld;
            ok = spd_inv<NT>(L, E.s, G, d, ld) && ok;
            const Vec<NT> t = matvec_t<NT>(G, to_k<NT>(L, v));
            const Vec<NT> xo = matvec_t<NT>(N, to_k<NT>(L, t));
            Mat<NT> T1, Lo;
            const int n16 = tiles<NT>(d);
            mul<NT>(T1, G, N, n16);    // (Λ + W)⁻¹ W
            mul<NT>(Lo, M, T1, n16);   // Λ (Λ + W)⁻¹ W
            store_msg<NT, ES1>(E, w[W_OUT], d, xo, Lo);
        }
    } break;
    case OP_MUL_OUT: {   // N(A m, A V Aᵀ): in dimension d1, out dimension d
        const int d1 = w[W_D1];
        Vec<NT> v;
        Mat<NT> V, at, Y, Vo;
        ok = load_msg<NT, ES1>(E, w[W_IN0], fl & F_IN0_WP, false, d1, v, V);
        load_mat<NT, true>(L, E.p.cpool + w[W_C0], 1, d1, d, 1, d1, at);   // Aᵀ
        const Vec<NT> m = matvec_t<NT>(at, to_k<NT>(L, v));
        const int n16 = tiles<NT>(d > d1 ? d : d1);
        mul<NT>(Y, V, at, n16);     // V Aᵀ
        mul<NT>(Vo, at, Y, n16);    // A (V Aᵀ)
        store_msg<NT, ES1>(E, w[W_OUT], d, m, Vo);
    } break;
    case OP_MUL_IN: {    // (Aᵀ ξ, Aᵀ Λ A): in dimension d (the message toward `out`), out dimension d1
        const int d1 = w[W_D1];
        Vec<NT> v;
        Mat<NT> Lm, a, Y, Lo;
        ok = load_msg<NT, ES1>(E, w[W_IN0], fl & F_IN0_WP, true, d, v, Lm);
        load_mat<NT, true>(L, E.p.cpool + w[W_C0], 1, d, d1, d1, 1, a);   // A
        const Vec<NT> xo = matvec_t<NT>(a, to_k<NT>(L, v));
        const int n16 = tiles<NT>(d > d1 ? d : d1);
        mul<NT>(Y, Lm, a, n16);     // Λ A
        mul<NT>(Lo, a, Y, n16);     // Aᵀ (Λ A)
        store_msg<NT, ES1>(E, w[W_OUT], d1, xo, Lo);
    } break;
    case OP_ADD_OUT:
    case OP_ADD_IN: {
        Vec<NT> v0, v1;
        Mat<NT> M0, M1;
        if (op == OP_ADD_IN && (fl & F_IN0_WP)) {   // Λ' = Λo (Λo + W2)⁻¹ W2, ξ' = W2 (Λo + W2)⁻¹ (ξo + ξ2) − ξ2
            ok = load_msg<NT, ES1>(E, w[W_IN0], true, true, d, v0, M0);
            ok = load_msg<NT, ES1>(E, w[W_IN1], fl & F_IN1_WP, true, d, v1, M1) && ok;
            Mat<NT> G = M0;
            axpy<NT>(G, 1.0, M1);
            axpy<NT>(v0, 1.0, v1);
            double ld;
            ok = spd_inv<NT>(L, E.s, G, d, ld) && ok;
            const Vec<NT> t = matvec_t<NT>(G, to_k<NT>(L, v0));
            Vec<NT> xo = matvec_t<NT>(M1, to_k<NT>(L, t));
            axpy<NT>(xo, -1.0, v1);
            Mat<NT> T1, Lo;
            const int n16 = tiles<NT>(d);
            mul<NT>(T1, G, M1, n16);
            mul<NT>(Lo, M0, T1, n16);
            store_msg<NT, ES1>(E, w[W_OUT], d, xo, Lo);
            break;
        }
        ok = load_msg<NT, ES1>(E, w[W_IN0], fl & F_IN0_WP, false, d, v0, M0);
        ok = load_msg<NT, ES1>(E, w[W_IN1], fl & F_IN1_WP, false, d, v1, M1) && ok;
        axpy<NT>(v0, op == OP_ADD_OUT ? 1.0 : -1.0, v1);
        axpy<NT>(M0, 1.0, M1);
        store_msg<NT, ES1>(E, w[W_OUT], d, v0, M0);
    } break;
    case OP_SHIFT: {
        const bool wp = fl & F_IN0_WP;
        Vec<NT> v, x;
        Mat<NT> M;
        ok = load_msg<NT, ES1>(E, w[W_IN0], wp, wp, d, v, M);
        load_value<NT, ES1>(E, w[W_VAL], fl & F_VAL_SLOT, d, x);
        const double sg = (fl & F_NEG) ? -1.0 : 1.0;
        if (wp) axpy<NT>(v, sg, matvec_t<NT>(M, to_k<NT>(L, x)));
        else axpy<NT>(v, sg, x);
        store_msg<NT, ES1>(E, w[W_OUT], d, v, M);
    } break;
    case OP_PRODUCT:
    case OP_MARGINAL: {
        Vec<NT> v0, v1;
        Mat<NT> M0, M1;
        zero<NT>(M0);
        zero<NT>(v0);
        const int n = w[W_N];
        const int* lst = E.p.aux + w[W_LIST];
        bool single = op == OP_MARGINAL && n == 1 && lst[1] == 0;   // (tree_kernels.hpp: the marginal of one moment-form message is the message)
        int at = 0;
        if (op == OP_MARGINAL && (fl & F_MAY_MISS) && !single) {   // (… and of one moment-form message next to `missing` observations)
            int n_mv = 0;
            bool info = false;
            for (int q = 0; q < n; ++q) {
                if (lst[2 * q + 1] == 0) {
                    ++n_mv;
                    at = q;
                    continue;
                }
                const double* b = E.msg(lst[2 * q]);
                for (int i = threadIdx.x; i < d; i += 64) info = info || b[(long long)(d + i * (i + 1) / 2 + i) * es] != 0.0;
            }
            single = n_mv == 1 && !__any(info);
        }
        for (int q = 0; q < n; ++q) {   // left to right, in factor order
            if (single && q != at) continue;
            ok = load_msg<NT, ES1>(E, lst[2 * q], lst[2 * q + 1] != 0, !single, d, v1, M1) && ok;
            axpy<NT>(v0, 1.0, v1);
            axpy<NT>(M0, 1.0, M1);
        }
        if (op == OP_PRODUCT) store_msg<NT, ES1>(E, w[W_OUT], d, v0, M0);
        else {
            double ld;
            ok = spd_inv<NT>(L, E.s, M0, d, ld) && ok;
            Vec<NT> m = matvec_t<NT>(M0, to_k<NT>(L, v0));
            if (single) {
                load_msg<NT, ES1>(E, lst[2 * at], false, false, d, m, M0);
                ld = -ld;
            }
            double* b = E.marg(w[W_OUT]);
            store_vec<NT, ES1>(L, b, es, d, m);
            store_sym<NT, ES1>(L, b + (long long)d * es, es, d, M0);
            if (threadIdx.x == 0) b[(long long)(d + d * (d + 1) / 2) * es] = -ld;
        }
    } break;
    default: break;
    }
    if (!ok && threadIdx.x == 0) atomicOr(E.p.status, 1);
}

// the second phase: Bethe terms, residual moments, q(W) updates — tree_wave_kernels.hpp eval_fe, op for op
// HEAVY = false: the instance without the joint-marginal algebra (OP_FE_NOISE2M, OP_FE_ADD2) and the q(W) update — the terms of a chain's observation nodes, the
// entropies, the sums: half the registers, twice the wavefronts per SIMD (the host launches a level's ops by opcode class: tree_engine.hip launch_wave_phase)
template <int NT, bool ES1, int HEAVY = 1>   // 1: every op; 0: the light ops; 2: OP_FE_NOISE2M alone (the joint term of a chain's transitions: its own register budget)
__device__ __forceinline__ void eval_fe(const Env<NT>& E, const int* __restrict__ w) {
    const int op = w[W_OP], d = w[W_D0], fl = w[W_FLAGS];
    const Lane<NT>& L = E.L;
    const long long es = E.p.es;
    const int n16 = tiles<NT>(d);
    bool ok = true;
    switch (op) {
    case OP_MARG_PUSH: {   // the stored marginal of an `A * x` output, formed when a caller asks for it
        Vec<NT> m;
        Mat<NT> V;
        const double ldV = load_marginal<NT, ES1>(E, w[W_IN0], true, w[W_C0], w[W_D1], d, true, m, V, w[W_IN1]);
        double* b = E.marg(w[W_OUT]);
        store_vec<NT, ES1>(L, b, es, d, m);
        store_sym<NT, ES1>(L, b + (long long)d * es, es, d, V);
        if (threadIdx.x == 0) b[(long long)(d + d * (d + 1) / 2) * es] = ldV;
    } break;
    case OP_FE_NOISE2M: if (HEAVY) {
        // the joint of a Gaussian node's two interfaces from ONE inbound message (side a) and the two marginals: P = L_a + W, log|J| = log|P| − log|V_b|,
        // Cov(a − b) = P⁻¹ + D V_b Dᵀ with D = P⁻¹ W − I — formed from Dᵀ = W P⁻¹ − I alone: D (V_b Dᵀ)
        Vec<NT> mb, ma, v0;
        Mat<NT> Vb, P, W, Dt, Y, dummy;
        const double ldVb = load_marginal<NT, ES1>(E, w[W_VAL2], fl & F_PUSH_B, w[W_IN2], w[W_N], d, true, mb, Vb, (fl & F_PUSH_B) ? w[W_D1] : -1);
        (void)load_marginal<NT, ES1>(E, w[W_VAL], fl & F_PUSH_A, w[W_IN1], w[W_LIST], d, false, ma, dummy);
        if (w[W_IN0] >= 0) ok = load_msg<NT, ES1>(E, w[W_IN0], fl & F_IN0_WP, true, d, v0, P);
        else zero<NT>(P);
        const double el = load_noise<NT, ES1>(E, w, d, false, W);
        axpy<NT>(P, 1.0, W);
        double ldP;
        ok = spd_inv<NT>(L, E.s, P, d, ldP) && ok;
        mul<NT>(Dt, W, P, n16);           // W P⁻¹
        add_diag<NT>(L, Dt, -1.0, d);
        axpy<NT>(ma, -1.0, mb);
        mul<NT>(Y, Vb, Dt, n16);          // V_b Dᵀ
        mul<NT>(P, Dt, Y, n16, true);     // P⁻¹ += D (V_b Dᵀ)
        add_outer<NT>(P, to_k<NT>(L, ma), ma);
        double term = -0.5 * (2.0 * d * (T_LOG2PI + 1.0) - (ldP - ldVb));
        if (fl & F_FOLD_ENT) term += (double)w[W_OUT] * 0.5 * (d * (T_LOG2PI + 1.0) + ldVb);
        if (fl & F_STAT) store_full<NT, ES1>(L, E.stat(w[W_C1]), es, d, P, 1.0);
        else term += 0.5 * (d * T_LOG2PI - el + dot<NT>(W, P));
        if (threadIdx.x == 0) *E.term(w[W_TERM]) = term;
    } break;
    case OP_FE_NOISE_MF: {   // a Gaussian node under q(out) q(μ): E[rrᵀ] = V_out + V_μ + (m_out − m_μ)(m_out − m_μ)ᵀ
        Vec<NT> m0, m1;
        Mat<NT> V0, V1, W;
        (void)load_marginal<NT, ES1>(E, w[W_VAL], false, 0, 0, d, true, m0, V0);
        (void)load_marginal<NT, ES1>(E, w[W_VAL2], false, 0, 0, d, true, m1, V1);
        const double el = load_noise<NT, ES1>(E, w, d, false, W);
        axpy<NT>(m0, -1.0, m1);
        axpy<NT>(V0, 1.0, V1);
        add_outer<NT>(V0, to_k<NT>(L, m0), m0);
        double term = 0.0;
        if (fl & F_STAT) store_full<NT, ES1>(L, E.stat(w[W_C1]), es, d, V0, 1.0);
        else term = 0.5 * (d * T_LOG2PI - el + dot<NT>(W, V0));
        if (threadIdx.x == 0) *E.term(w[W_TERM]) = term;
    } break;
    case OP_FE_NOISE1:
    case OP_FE_NOISE0: {
        double H = 0.0;
        Vec<NT> v0, v1;
        Mat<NT> V, W;
        if (op == OP_FE_NOISE1) {
            const double ldV = load_marginal<NT, ES1>(E, w[W_IN0], fl & F_PUSH_A, w[W_IN1], w[W_D1], d, true, v0, V, (fl & F_PUSH_A) ? w[W_IN2] : -1);
            H = 0.5 * (d * (T_LOG2PI + 1.0) + ldV);
            if (fl & F_FOLD_ENT) H *= (double)(1 - w[W_OUT]);
            load_value<NT, ES1>(E, w[W_VAL], fl & F_VAL_SLOT, d, v1);
        } else {
            zero<NT>(V);
            load_value<NT, ES1>(E, w[W_VAL], fl & F_VAL_SLOT, d, v0);
            load_value<NT, ES1>(E, w[W_VAL2], fl & F_VAL2_SLOT, d, v1);
        }
        const double el = load_noise<NT, ES1>(E, w, d, false, W);
        axpy<NT>(v0, -1.0, v1);
        const bool miss = (fl & F_MAY_MISS) && any_nan<NT>(v0);   // a `missing` observation: energy and the predicted value's entropy cancel, −H stays
        add_outer<NT>(V, to_k<NT>(L, v0), v0);
        double term = -H;
        if (fl & F_STAT) store_full<NT, ES1>(L, E.stat(w[W_C1]), es, d, V, 1.0);
        else {
            const double tr = dot<NT>(W, V);
            if (!miss) term += 0.5 * (d * T_LOG2PI - el + tr);
        }
        if (threadIdx.x == 0) *E.term(w[W_TERM]) = term;
    } break;
    case OP_FE_ENT: {
        double ldV;
        if (fl & F_PUSH_A) {
            Vec<NT> m;
            Mat<NT> V;
            ldV = load_marginal<NT, ES1>(E, w[W_IN0], true, w[W_C0], w[W_D1], d, true, m, V, w[W_IN1]);
        } else
            ldV = E.marg(w[W_IN0])[(long long)(d + d * (d + 1) / 2) * es];
        if (threadIdx.x == 0) *E.term(w[W_TERM]) = (double)w[W_N] * 0.5 * (d * (T_LOG2PI + 1.0) + ldV);
    } break;
    case OP_FE_ADD2: if (HEAVY == 1) {   // P = Λ1 + Λo, S = Λ2 + Λo − Λo P⁻¹ Λo
        Vec<NT> v;
        Mat<NT> P, S, Lo, T;
        if (w[W_IN0] >= 0) ok = load_msg<NT, ES1>(E, w[W_IN0], fl & F_IN0_WP, true, d, v, P);
        else zero<NT>(P);
        if (w[W_IN1] >= 0) ok = load_msg<NT, ES1>(E, w[W_IN1], fl & F_IN1_WP, true, d, v, S) && ok;
        else zero<NT>(S);
        if (w[W_IN2] >= 0) ok = load_msg<NT, ES1>(E, w[W_IN2], fl & F_IN2_WP, true, d, v, Lo) && ok;
        else zero<NT>(Lo);
        axpy<NT>(P, 1.0, Lo);
        axpy<NT>(S, 1.0, Lo);
        double ldP, ldS;
        ok = spd_inv<NT>(L, E.s, P, d, ldP) && ok;
        mul<NT>(T, P, Lo, n16);              // P⁻¹ Λo
        mul<NT>(S, Lo, T, n16, true, -1.0);  // S −= Λo P⁻¹ Λo
        ok = spd_inv<NT>(L, E.s, S, d, ldS) && ok;
        if (threadIdx.x == 0) *E.term(w[W_TERM]) = -0.5 * (2.0 * d * (T_LOG2PI + 1.0) - (ldP + ldS));
    } break;
    case OP_SUM_TERMS: {
        if (threadIdx.x == 0) {
            const int n = w[W_N];
            const int* lst = E.p.aux + w[W_LIST];
            double s = 0.0;
            for (int q = 0; q < n; ++q) s += *E.term(lst[q]);
            *E.term(w[W_TERM]) = s;
        }
    } break;
    case OP_PREC_UPDATE: if (HEAVY == 1) {
        // prior block at c0: ν0 | S0⁻¹ (d²) | log|S0|;  S = Σ E[rrᵀ] symmetrised, V⁻¹ = S0⁻¹ + S, V
        const double* cp = E.p.cpool + w[W_C0];
        const double nu0 = cp[0], ldS0 = cp[1 + d * d];
        Mat<NT> S, St, Vi, V, S0i;
        zero<NT>(S);
        zero<NT>(St);
        const int n = w[W_N];
        const int* lst = E.p.aux + w[W_LIST];
        for (int q = 0; q < n; ++q) {
            Mat<NT> a, b;
            load_mat<NT, ES1>(L, E.stat(lst[q]), es, d, d, d, 1, a);
            load_mat<NT, ES1>(L, E.stat(lst[q]), es, d, d, 1, d, b);
            axpy<NT>(S, 1.0, a);
            axpy<NT>(St, 1.0, b);
        }
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) S.t[ti][tj] = 0.5 * (S.t[ti][tj] + St.t[ti][tj]);
        load_mat<NT, true>(L, cp + 1, 1, d, d, d, 1, S0i);
        Vi = S0i;
        axpy<NT>(Vi, 1.0, S);
        V = Vi;
        double ldVi;
        ok = spd_inv<NT>(L, E.s, V, d, ldVi);
        const double nu = nu0 + (double)n, ldV = -ldVi;
        const int ps = w[W_PREC], tri = d * (d + 1) / 2;
        const double elw = t_mvdigamma(0.5 * nu, d) + d * T_LOG2 + ldV;
        double* pb = E.prec(ps);
        if (threadIdx.x == 0) {
            pb[0] = nu;
            pb[(long long)(1 + tri + 2 * d * d) * es] = elw;
        }
        store_sym<NT, ES1>(L, pb + es, es, d, V);
        store_full<NT, ES1>(L, pb + (long long)(1 + tri) * es, es, d, V, nu);
        store_full<NT, ES1>(L, pb + (long long)(1 + tri + d * d) * es, es, d, Vi, 1.0 / nu);
        if (E.p.want_fe) {
            double F = 0.5 * ((double)n * (d * T_LOG2PI - elw) + nu * dot<NT>(V, S));
            F += -(0.5 * (nu0 - d - 1.0) * elw - 0.5 * nu * dot<NT>(S0i, V) - 0.5 * nu0 * d * T_LOG2 - 0.5 * nu0 * ldS0 - t_mvlgamma(0.5 * nu0, d));
            F -= 0.5 * (d + 1.0) * ldV + 0.5 * d * (d + 1.0) * T_LOG2 + t_mvlgamma(0.5 * nu, d) - 0.5 * (nu - d - 1.0) * t_mvdigamma(0.5 * nu, d) + 0.5 * nu * d;
            if (threadIdx.x == 0) *E.term(w[W_TERM]) = F;
        }
    } break;
    default: break;
    }
    if (!ok && threadIdx.x == 0) atomicOr(E.p.status, 1);
}

// One launch per level: a wavefront (a workgroup of 64) per item (op, replica).  Storage as TreeParams says: element k of slot `off` of replica r at
// (off + k)·es + r·rs_<array> — the engines above d = 8 store a replica's slots contiguously (es = 1), rxhip_rule_eval's one-node schedules replica-fastest.
template <int PHASE, int NT, bool ES1>
__global__ void __launch_bounds__(64) k_tile_ops(TreeParams p, int op0, int op1) {
    __shared__ Scratch<NT> scratch;
    const Lane<NT> L = make_lane<NT>();
    const long long total = (long long)(op1 - op0) * p.R;
    for (long long it = blockIdx.x; it < total; it += gridDim.x) {
        const long long o = it / p.R;
        const Env<NT> E{L, scratch, p, it - o * p.R};
        const int* w = p.ops + (size_t)(op0 + o) * OP_WORDS;
        if (PHASE == 0) eval_bp<NT, ES1>(E, w);
        else if (PHASE == 1) eval_fe<NT, ES1, 1>(E, w);
        else if (PHASE == 2) eval_fe<NT, ES1, 0>(E, w);
        else eval_fe<NT, ES1, 2>(E, w);
    }
}
// a wavefront owns a replica and walks the ops of the range in order (every op's inputs were written by this wavefront or before the launch)
template <int PHASE, int NT, bool ES1>
__global__ void __launch_bounds__(64) k_tile_walk(TreeParams p, int op0, int op1) {
    __shared__ Scratch<NT> scratch;
    const Lane<NT> L = make_lane<NT>();
    for (long long r = blockIdx.x; r < p.R; r += gridDim.x) {
        const Env<NT> E{L, scratch, p, r};
        for (int o = op0; o < op1; ++o) {
            const int* w = p.ops + (size_t)o * OP_WORDS;
            if (PHASE == 0) eval_bp<NT, ES1>(E, w);
            else eval_fe<NT, ES1>(E, w);
            __syncthreads();   // (one wavefront: a workgroup-scope fence — this op's stores before the next op's loads by other lanes)
        }
    }
}

}  // namespace tile
}  // namespace tree
}  // namespace rxhip
