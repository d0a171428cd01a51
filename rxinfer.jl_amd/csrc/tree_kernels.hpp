// tree_kernels.hpp — the level-scheduled node-array executor: ONE kernel that evaluates message rules over struct-of-arrays node tables.
//
// What it replaces: the reference builds `factornode(fform, interfaces, factorization)` for any graph and fires one @rule body per (node,
// interface) through reactive streams (src/model/plugins/reactivemp_inference.jl:490-540; products :432-447; Bethe terms
// reactivemp_free_energy.jl:51-126).  Here the host compiles the graph once (tree_engine.hip) into a list of OPS — one per message, product,
// marginal, free-energy term — sorted by dependency level, and a work item of this kernel is (op, replica): lane-per-node, replica-fastest
// storage, so the 64 lanes of a wavefront evaluate the same rule of the same node for 64 replicas with unit-stride loads, or (one replica) 64
// different nodes of one level.  All state dimensions ≤ DMAX (template: 1, 2, 4) live in registers; runtime dimensions are handled by padding
// (identity on the diagonal of whatever gets inverted, zeros elsewhere) under fully unrolled loops.
//
// Storage (doubles, replica-fastest: element k of slot `off` of replica r at (off + k)·RS + r, RS = replicas rounded up to 16 = a 128-byte line):
//   message   [d | d(d+1)/2]        moment form (m, V) or weighted-mean / precision form (ξ, Λ); lower triangle, row-major — 8·(d + d(d+1)/2) bytes
//   marginal  [d | d(d+1)/2 | 1]    mean, covariance, log det covariance
//   precision [ν | V tri | Ŵ d² | Ŵ⁻¹ d² | E log|W|]    q(W) = Wishart(ν, V) of a mean-field precision variable and what the rules read of it
//   term / stat: Bethe terms and residual second moments, summed in a fixed order (deterministic to the bit)
#pragma once
#include <hip/hip_runtime.h>

namespace rxhip {
namespace tree {

enum : int {
    OP_NOP = 0,
    OP_DERIVE_MUL = 1,   // clamped value through `A * x`
    OP_DERIVE_ADD = 2,   // clamped value through `a + b`
    OP_LEAF = 3,         // Gaussian node, other interface clamped:            N(value, Σ)  |  (W value, W)
    OP_NOISE = 4,        // Gaussian node toward out / μ, other interface random: the additive rule in the form the inbound message has
    OP_MUL_OUT = 5,      // typeof(*)(:out):  N(A m, A V Aᵀ)
    OP_MUL_IN = 6,       // typeof(*)(:in):   (Aᵀ ξ, Aᵀ Λ A)
    OP_ADD_OUT = 7,      // typeof(+)(:out), two random inputs:  N(m1 + m2, V1 + V2)
    OP_ADD_IN = 8,       // typeof(+)(:in1 | :in2), the other input random:  N(m_out − m2, V_out + V2) — in the form the message from `out` has
    OP_SHIFT = 9,        // typeof(+) with a clamped input: ± the value, in the inbound message's own form
    OP_PRODUCT = 10,     // product of inbound messages at a variable (outbound variable → factor message)
    OP_MARGINAL = 11,    // product of ALL inbound messages, as (mean, cov, log det)
    OP_FE_NOISE2 = 12,   // Bethe term of a Gaussian node with both interfaces random (node-local joint q(out, μ)); residual moments for q(W)
    OP_FE_NOISE1 = 13,   // … one interface random
    OP_FE_NOISE0 = 14,   // … both clamped
    OP_FE_ENT = 15,      // coef · H[q(v)] of a random variable: (degree − 1) minus the deterministic nodes whose only random input it is
    OP_FE_ADD2 = 16,     // −H[q(in1, in2)] of a `+` node with two random inputs
    OP_SUM_TERMS = 17,   // fixed-order partial sum of terms
    OP_PREC_UPDATE = 18, // q(W) ← Wishart(ν0 + n, (S0⁻¹ + Σ E[rrᵀ])⁻¹), its share of the Bethe sum
    OP_FE_NOISE2M = 19,  // OP_FE_NOISE2 from ONE inbound message and the two variables' marginals (register kernels): one inverse instead of three
    OP_MARG_PUSH = 20,   // marginal of the output of `A * x` as the image of x's marginal: (A m, A V Aᵀ, log|A V Aᵀ|) — second phase, register kernels
    OP_CAT_UPDATE = 22,  // q(z) of a NormalMixture node's switch: π_k ∝ exp(E log s_k − ½[d log 2π − E log|p_k| + tr(E[p_k] E[(out − m_k)(out − m_k)ᵀ])]) from the marginals of the
                         // previous iteration (lane-per-item kernels; the node itself is K weighted Gaussian nodes: F_WEIGHT)
    OP_DIR_UPDATE = 23,  // q(s) = Dirichlet(a + Σ_i π_i) of a probability vector with the terms of its switches: −Σ π E log s, −H[q(z)], the prior node U − H[q(s)]
    OP_GCV_Z = 24,       // GCV(y, x, z, κ, ω) toward z (scalars; q(y, x) q(z)): from the messages y → node, x → node and E[γ] the node-local joint and ψ = E[(y − x)²] (→ the slot W_C1); the
                         // message z and its neighbours see = the Gaussian moments of ExponentialLinearQuadratic(κ, ψ e^{−ω}, −κ) (cubature against N(0, 1)) → W_OUT.  Constants at W_C0: κ | ω | n | nodes | weights/√π
    OP_GCV_ZMARG = 26,   // … q(z): that ELQ times the product of all OTHER messages into z (W_IN0), moment-matched by cubature against the product; ψ from the slot OP_GCV_Z left it in
    OP_GCV_PREC = 25,    // the node's precision γ(z) = exp(−(κ z + ω)) under the new q(z): E γ, 1 / E γ, E log γ into the state slot W_PREC; the GCV average energy ½[log 2π − E log γ + E γ ψ]
    OP_FE_NOISE_MF = 21  // average energy of a Gaussian node under q(out) q(μ) (mean field between its Gaussian interfaces): E[rrᵀ] = V_out + V_μ + (m_out − m_μ)(…)ᵀ
};
constexpr int OP_WORDS = 16;
// word indices of an op descriptor
enum : int { W_OP = 0, W_D0, W_D1, W_OUT, W_IN0, W_IN1, W_IN2, W_FLAGS, W_C0, W_C1, W_VAL, W_VAL2, W_PREC, W_TERM, W_N, W_LIST };
// flags
enum : int {
    F_IN0_WP = 1, F_IN1_WP = 2, F_IN2_WP = 4, F_OUT_WP = 8,
    F_VAL_SLOT = 16,     // W_VAL names a per-replica value slot (data / derived), else the constant pool
    F_VAL2_SLOT = 32,
    F_NEG = 64,          // OP_SHIFT: subtract the value
    F_STAT = 128,        // FE_NOISE*: the node's precision is a random variable — write E[rrᵀ] to the stat slot W_C1
    F_RAND_IS_MU = 256,  // FE_NOISE1: the random interface is μ (r = value − μ)
    F_NO_STORE = 512,    // strand schedule: the only reader of this op's message is the next op of the strand (it takes it from registers)
    F_PUSH_A = 1024,     // Bethe terms: the marginal named by W_VAL (FE_NOISE2M side a) / W_IN0 (FE_NOISE1, FE_ENT) is the IMAGE of the stored one under a constant matrix
    F_PUSH_B = 2048,     // … the marginal named by W_VAL2 (FE_NOISE2M side b)
    F_FOLD_ENT = 4096,   // FE_NOISE2M / FE_NOISE1: W_OUT · H[q(v)] of the variable whose log|V| the op has at hand (side b / the random interface) is part of this term
    F_MAY_MISS = 16384,  // OP_LEAF / FE_NOISE1 / FE_NOISE0 on a DATA value of a graph created with allow_missing: NaN (`missing`) → no message / no energy term; OP_MARGINAL of such a graph
    F_WEIGHT = 32768,    // OP_LEAF / FE_NOISE0 / FE_NOISE1 / FE_NOISE_MF: a component of a mixture node — message, energy and residual moments × π_k, the double at p.prec[W_LIST];
                         // OP_PREC_UPDATE: the list holds (moments, weight | −1) pairs, ν = ν0 + Σ weights
    F_JOINT_MEAN = 65536,      // OP_FE_NOISE2M: the mean of side a from the node-local joint, P⁻¹(ξ_a + W m_b), not from the variable's marginal — side a is the volatility input of a
                               // GCV node, whose marginal is a cubature-matched product and not the product of the Gaussian messages the joint is made of
    F_JOINT_B = 524288,        // … and side b's moments as well (both interfaces are such inputs: a transition between two volatility states): W_IN1 names b's message,
                               // S = Λ_b + W − W P⁻¹ W, V_b = S⁻¹, m_b = V_b (ξ_b + W P⁻¹ ξ_a)
    F_NOISE_VAL = 131072,      // a SCALAR Gaussian node whose variance (F_NOISE_VAL_PREC: precision) is a DATA variable — `x ~ Normal(mean = m_prev, var = v_prev)` of a streaming
                               // model's @autoupdates: the value slot W_C0 instead of a constant block (lane-per-item kernels)
    F_NOISE_VAL_PREC = 262144,
    F_VAL_MARG = 8192    // OP_LEAF: the value is the MEAN of the marginal slot W_VAL — the rule of a Gaussian node under q(out) q(μ): N(E[μ], Σ) toward out, N(E[out], Σ) toward μ
};
// strand schedule: an input offset that names the message the previous op of the lane's strand left in registers
constexpr int OFF_REG = -2;
template <int N>
struct RegMsg {
    double a[N], B[N][N];
};

struct TreeParams {
    const int* ops;       // [n_ops][OP_WORDS]
    const int* aux;       // lists: (offset, form) pairs of OP_PRODUCT / OP_MARGINAL, offsets of OP_SUM_TERMS / OP_PREC_UPDATE
    const double* cpool;  // constants (replica-independent): matrices row-major; noise blocks Σ | W | log|W|; priors ν0 | S0⁻¹ | log|S0|
    double* msg;
    double* marg;
    double* val;          // data and derived values
    double* prec;
    double* term;
    double* stat;
    long long R, RS;      // replicas; replica stride (R rounded up to 16)
    // layout of the per-replica arrays as the LDS-staged kernels (tree_wave_kernels.hpp) address them: element k of slot `off` of replica r at
    // (off + k)·es + r·rs_<array>.  Replica-fastest (the register kernels of this file, hard-wired): es = RS, rs = 1.  Element-fastest (engines with a
    // dimension above 8): es = 1, rs = the array's doubles per replica.
    long long es, rs_msg, rs_marg, rs_val, rs_prec, rs_term, rs_stat;
    int want_fe;
    int* status;          // bit 0: a matrix that must be positive definite was not
};

constexpr double T_LOG2PI = 1.8378770664093454836;
constexpr double T_ABSENT_VARIANCE = 1.0e200;   // the moment form of "no message" (load_msg)
constexpr double T_LOG2 = 0.69314718055994530942;

template <int N>
__device__ __forceinline__ void ld_vec(const double* b, long long off, int d, long long RS, long long r, double (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = i < d ? b[(off + i) * RS + r] : 0.0;
}
template <int N>
__device__ __forceinline__ void st_vec(double* b, long long off, int d, long long RS, long long r, const double (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i)
        if (i < d) b[(off + i) * RS + r] = v[i];
}
// packed lower triangle -> full symmetric, `pad` on the diagonal beyond d
template <int N>
__device__ __forceinline__ void ld_sym(const double* b, long long off, int d, long long RS, long long r, double pad, double (&M)[N][N]) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            const double x = (i < d) ? b[(off + i * (i + 1) / 2 + j) * RS + r] : (i == j ? pad : 0.0);
            M[i][j] = x;
            M[j][i] = x;
        }
}
template <int N>
__device__ __forceinline__ void st_sym(double* b, long long off, int d, long long RS, long long r, const double (&M)[N][N]) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j)
            if (i < d) b[(off + i * (i + 1) / 2 + j) * RS + r] = 0.5 * (M[i][j] + M[j][i]);
}
// full d×d per-replica matrix (precision state)
template <int N>
__device__ __forceinline__ void ld_full(const double* b, long long off, int d, long long RS, long long r, double pad, double (&M)[N][N]) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) M[i][j] = (i < d && j < d) ? b[(off + i * d + j) * RS + r] : (i == j ? pad : 0.0);
}
template <int N>
__device__ __forceinline__ void st_full(double* b, long long off, int d, long long RS, long long r, const double (&M)[N][N]) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j)
            if (i < d && j < d) b[(off + i * d + j) * RS + r] = M[i][j];
}
// constant matrix rows×cols (row-major, the same for every replica), zero padding
template <int N>
__device__ __forceinline__ void ld_cmat(const double* c, int rows, int cols, double pad, double (&M)[N][N]) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) M[i][j] = (i < rows && j < cols) ? c[i * cols + j] : (i == j ? pad : 0.0);
}
template <int N>
__device__ __forceinline__ void ld_cvec(const double* c, int d, double (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = i < d ? c[i] : 0.0;
}

// inverse and log-determinant of a symmetric positive definite matrix (Cholesky, A = L Lᵀ, A⁻¹ = L⁻ᵀ L⁻¹); false: a pivot ≤ 0 or not finite.
// One reciprocal square root per pivot (1 / √s; √s = s · that) instead of a square root and a division, and ONE logarithm per call: the pivots' mantissas are
// multiplied (each in [½, 1): no under- or overflow for N ≤ 8 … 64 factors would need rescaling, N ≤ 8 here) and their binary exponents added —
// log|A| = log(Π mantissas) + ln 2 · Σ exponents.  (Four square roots, four divisions and four logarithms were ≈ 1000 of the ≈ 1600 cycles of a 4×4 call.)
template <int N>
__device__ __forceinline__ bool spd_inv(const double (&A)[N][N], double (&Ai)[N][N], double& logdet) {
    double L[N][N], Li[N][N];
    bool ok = true;
    double mant = 1.0;
    int expo = 0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double s = A[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) s -= L[j][k] * L[j][k];
        ok = ok && (s > 0.0) && (s < 1.0e300);
#ifdef RXHIP_HOST_EMUL
        const double rj = 1.0 / sqrt(s);
        int ex;
        mant *= frexp(s, &ex);
        expo += ex;
#else
        const double rj = rsqrt(s);
        mant *= __builtin_amdgcn_frexp_mant(s);
        expo += __builtin_amdgcn_frexp_exp(s);
#endif
        const double dj = s * rj;
        L[j][j] = dj;
        Li[j][j] = rj;
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            double t = A[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) t -= L[i][k] * L[j][k];
            L[i][j] = t * rj;
        }
    }
    const double ld = log(mant) + 0.69314718055994530942 * (double)expo;
    // L⁻¹ (lower): Li[i][j] = −(Σ_{k=j}^{i−1} L[i][k] Li[k][j]) / L[i][i]
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            double t = 0.0;
#pragma unroll
            for (int k = j; k < i; ++k) t += L[i][k] * Li[k][j];
            Li[i][j] = -t * Li[i][i];
        }
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double t = 0.0;
#pragma unroll
            for (int k = i; k < N; ++k) t += Li[k][i] * Li[k][j];
            Ai[i][j] = t;
            Ai[j][i] = t;
        }
    logdet = ld;
    return ok;
}
template <int N>
__device__ __forceinline__ void matvec(const double (&A)[N][N], const double (&x)[N], double (&y)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < N; ++k) s += A[i][k] * x[k];
        y[i] = s;
    }
}
template <int N>
__device__ __forceinline__ void matTvec(const double (&A)[N][N], const double (&x)[N], double (&y)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < N; ++k) s += A[k][i] * x[k];
        y[i] = s;
    }
}
template <int N>
__device__ __forceinline__ void matmul(const double (&A)[N][N], const double (&B)[N][N], double (&C)[N][N]) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < N; ++k) s += A[i][k] * B[k][j];
            C[i][j] = s;
        }
}
template <int N>
__device__ __forceinline__ void matmulT(const double (&A)[N][N], const double (&B)[N][N], double (&C)[N][N]) {   // A Bᵀ
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < N; ++k) s += A[i][k] * B[j][k];
            C[i][j] = s;
        }
}
template <int N>
__device__ __forceinline__ void matTmul(const double (&A)[N][N], const double (&B)[N][N], double (&C)[N][N]) {   // Aᵀ B
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < N; ++k) s += A[k][i] * B[k][j];
            C[i][j] = s;
        }
}
template <int N>
__device__ __forceinline__ double trace_prod(const double (&A)[N][N], const double (&B)[N][N], int d) {   // tr(A B) over the leading d×d block
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int k = 0; k < N; ++k)
            if (i < d && k < d) s += A[i][k] * B[k][i];
    return s;
}

// a message in the form a rule wants: stored (a, B) is converted by one inverse when the forms differ
template <int N, bool STRAND = false>
__device__ __forceinline__ bool load_msg(const TreeParams& p, int off, bool stored_wp, bool want_wp, int d, long long r, double (&a)[N], double (&B)[N][N], const RegMsg<N>* reg = nullptr) {
    if (STRAND && off == OFF_REG) {   // what the previous op of this strand produced: exactly the values a load of its stored message would return
#pragma unroll
        for (int i = 0; i < N; ++i) {
            a[i] = reg->a[i];
#pragma unroll
            for (int j = 0; j < N; ++j) B[i][j] = reg->B[i][j];
        }
    } else {
        ld_vec<N>(p.msg, off, d, p.RS, r, a);
        ld_sym<N>(p.msg, off + d, d, p.RS, r, 1.0, B);
    }
    if (stored_wp == want_wp) return true;
    if (stored_wp) {
        // The ZERO of the precision form — a `missing` observation, through whatever maps and shifts it went — has no moment form.  A rule that wants moments
        // (`*`(:out), `+`(:out)) gets a covariance so wide that what it passes on is nothing to fifteen digits beyond the exponent (the oracle drops the message).
        bool zero = true;
#pragma unroll
        for (int i = 0; i < N; ++i) zero = zero && (i >= d || B[i][i] == 0.0);
        if (zero) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                a[i] = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j) B[i][j] = i == j ? (i < d ? T_ABSENT_VARIANCE : 1.0) : 0.0;
            }
            return true;
        }
    }
    double Bi[N][N], t[N], ld;
    const bool ok = spd_inv<N>(B, Bi, ld);
    matvec<N>(Bi, a, t);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        a[i] = t[i];
#pragma unroll
        for (int j = 0; j < N; ++j) B[i][j] = Bi[i][j];
    }
    return ok;
}
template <int N, bool STRAND = false>
__device__ __forceinline__ void store_msg(const TreeParams& p, int off, int d, long long r, const double (&a)[N], const double (&B)[N][N], int fl = 0, RegMsg<N>* reg = nullptr) {
    if (STRAND) {   // the registers hold what ld_vec / ld_sym would read back: symmetrised, zero / identity padding beyond d
#pragma unroll
        for (int i = 0; i < N; ++i) {
            reg->a[i] = i < d ? a[i] : 0.0;
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                const double x = (i < d) ? 0.5 * (B[i][j] + B[j][i]) : (i == j ? 1.0 : 0.0);
                reg->B[i][j] = x;
                reg->B[j][i] = x;
            }
        }
        if (fl & F_NO_STORE) return;
    }
    st_vec<N>(p.msg, off, d, p.RS, r, a);
    st_sym<N>(p.msg, off + d, d, p.RS, r, B);
}
// noise of a Gaussian node: Σ, W = Σ⁻¹ and (E) log|W| — constants (block Σ | W | log|W| at c0) or the state of a precision variable
template <int N>
__device__ __forceinline__ void load_noise(const TreeParams& p, const int* w, int d, long long r, bool want_sigma, bool want_w, double (&Sg)[N][N], double (&Wm)[N][N], double& elogdet) {
    const int ps = w[W_PREC];
    if (w[W_FLAGS] & F_NOISE_VAL) {   // d = 1: the variance / precision of this replica from its value slot
        const double x = p.val[(long long)w[W_C0] * p.RS + r], xi = 1.0 / x;
        const bool isp = w[W_FLAGS] & F_NOISE_VAL_PREC;
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) {
                Sg[i][j] = i == j ? 1.0 : 0.0;
                Wm[i][j] = i == j ? 1.0 : 0.0;
            }
        Sg[0][0] = isp ? xi : x;
        Wm[0][0] = isp ? x : xi;
        elogdet = log(Wm[0][0]);
        return;
    }
    if (ps >= 0) {
        const int tri = d * (d + 1) / 2;
        if (want_w) ld_full<N>(p.prec, ps + 1 + tri, d, p.RS, r, 1.0, Wm);
        if (want_sigma) ld_full<N>(p.prec, ps + 1 + tri + d * d, d, p.RS, r, 1.0, Sg);
        elogdet = p.prec[(ps + 1 + tri + 2 * d * d) * p.RS + r];
    } else {
        const double* c = p.cpool + w[W_C0];
        if (want_sigma) ld_cmat<N>(c, d, d, 1.0, Sg);
        if (want_w) ld_cmat<N>(c + d * d, d, d, 1.0, Wm);
        elogdet = c[2 * d * d];
    }
}
template <int N>
__device__ __forceinline__ void load_value(const TreeParams& p, int off, bool slot, int d, long long r, double (&v)[N]) {
    if (slot) ld_vec<N>(p.val, off, d, p.RS, r, v);
    else ld_cvec<N>(p.cpool + off, d, v);
}

// log-determinant of a symmetric positive definite matrix (Cholesky pivots); false: a pivot ≤ 0
template <int N>
__device__ __forceinline__ bool spd_logdet(const double (&A)[N][N], double& logdet) {
    double L[N][N];
    bool ok = true;
    double mant = 1.0;
    int expo = 0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double s = A[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) s -= L[j][k] * L[j][k];
        ok = ok && (s > 0.0) && (s < 1.0e300);
#ifdef RXHIP_HOST_EMUL
        const double rj = 1.0 / sqrt(s);
        int ex;
        mant *= frexp(s, &ex);
        expo += ex;
#else
        const double rj = rsqrt(s);
        mant *= __builtin_amdgcn_frexp_mant(s);
        expo += __builtin_amdgcn_frexp_exp(s);
#endif
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            double t = A[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) t -= L[i][k] * L[j][k];
            L[i][j] = t * rj;
        }
    }
    logdet = log(mant) + 0.69314718055994530942 * (double)expo;
    return ok;
}
// A marginal as the Bethe terms read it: (mean, covariance, log|V|) of the slot `off` — or, `push`, of the image of that marginal under the constant d × du
// matrix at cpool + aoff: the output of `A * x` has the marginal (A m, A V Aᵀ) of x's (exact on a tree), so the marginals of such (anonymous) variables are
// never stored for the free energy's sake; a singular image (more rows than columns) has log|V| = −∞, as the entropy of the message route.
// ldoff ≥ 0: cpool[ldoff] = 2·log|det A| of a square map — log|A V Aᵀ| = log|V| + that, no Cholesky
template <int N>
__device__ __forceinline__ void load_marginal(const TreeParams& p, int off, bool push, int aoff, int du, int d, long long r, bool want_cov, double (&m)[N], double (&V)[N][N], double& ldV,
                                              int ldoff = -1) {
    if (!push) {
        ld_vec<N>(p.marg, off, d, p.RS, r, m);
        if (want_cov) {
            ld_sym<N>(p.marg, off + d, d, p.RS, r, 0.0, V);
            ldV = p.marg[(long long)(off + d + d * (d + 1) / 2) * p.RS + r];
        }
        return;
    }
    double mu[N], A[N][N];
    ld_vec<N>(p.marg, off, du, p.RS, r, mu);
    ld_cmat<N>(p.cpool + aoff, d, du, 0.0, A);
    matvec<N>(A, mu, m);
    if (!want_cov) return;
    double Vu[N][N], T1[N][N], Vp[N][N], ld;
    ld_sym<N>(p.marg, off + du, du, p.RS, r, 0.0, Vu);
    matmul<N>(A, Vu, T1);
    matmulT<N>(T1, A, V);
    if (ldoff >= 0) {
        ldV = p.marg[(long long)(off + du + du * (du + 1) / 2) * p.RS + r] + p.cpool[ldoff];
        return;
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) Vp[i][j] = (i < d && j < d) ? 0.5 * (V[i][j] + V[j][i]) : (i == j ? 1.0 : 0.0);
    const bool pd = spd_logdet<N>(Vp, ld);
    ldV = pd ? ld : -__builtin_huge_val();
}

__device__ __forceinline__ double t_digamma(double x) {   // x > 0
    double r = 0.0;
    while (x < 6.0) { r -= 1.0 / x; x += 1.0; }
    const double f = 1.0 / (x * x);
    return r + log(x) - 0.5 / x - f * (1.0 / 12.0 - f * (1.0 / 120.0 - f * (1.0 / 252.0 - f * (1.0 / 240.0 - f * (1.0 / 132.0)))));
}
__device__ __forceinline__ double t_mvdigamma(double a, int d) {
    double s = 0.0;
    for (int i = 0; i < d; ++i) s += t_digamma(a - 0.5 * i);
    return s;
}
__device__ __forceinline__ double t_mvlgamma(double a, int d) {
    double s = 0.25 * d * (d - 1) * 1.1447298858494001741;   // log π
    for (int i = 0; i < d; ++i) s += lgamma(a - 0.5 * i);
    return s;
}

// ------------------------------------------------------------------------------------------
// the sum-product sweep: rules, products, marginals (PHASE 0) — and the Bethe terms, residual moments and q(W) updates (PHASE 1), a kernel of their own:
// the log-gamma / digamma code and the joint-marginal algebra of the second phase would otherwise set the register budget of every rule
template <int N, bool STRAND = false>
__device__ __forceinline__ void eval_bp(const TreeParams& p, const int* __restrict__ w, long long r, RegMsg<N>* reg = nullptr) {
    const int op = w[W_OP], d = w[W_D0], fl = w[W_FLAGS];
    bool ok = true;
    switch (op) {
    case OP_DERIVE_MUL: {   // val[out] = A (d × d1) val[in]
        double A[N][N], x[N], y[N];
        ld_cmat<N>(p.cpool + w[W_C0], d, w[W_D1], 0.0, A);
        load_value<N>(p, w[W_VAL], fl & F_VAL_SLOT, w[W_D1], r, x);
        matvec<N>(A, x, y);
        st_vec<N>(p.val, w[W_OUT], d, p.RS, r, y);
    } break;
    case OP_DERIVE_ADD: {
        double x[N], y[N];
        load_value<N>(p, w[W_VAL], fl & F_VAL_SLOT, d, r, x);
        load_value<N>(p, w[W_VAL2], fl & F_VAL2_SLOT, d, r, y);
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] += y[i];
        st_vec<N>(p.val, w[W_OUT], d, p.RS, r, x);
    } break;
    case OP_GCV_Z: {
        double ay[N], By[N][N], ax[N], Bx[N][N];
        ok = load_msg<N, STRAND>(p, w[W_IN0], fl & F_IN0_WP, true, 1, r, ay, By, reg);
        ok = load_msg<N, STRAND>(p, w[W_IN1], fl & F_IN1_WP, true, 1, r, ax, Bx, reg) && ok;
        const double* c = p.cpool + w[W_C0];
        const double kappa = c[0], A = exp(-c[1]);
        const int n = (int)c[2];
        const double *gx = c + 3, *gw = c + 3 + n;
        const double gam = p.prec[(long long)(w[W_PREC] + 2) * p.RS + r];
        // @marginalrule GCV(:y_x): joint precision [[Λy + γ, −γ], [−γ, Λx + γ]]
        const double l11 = By[0][0] + gam, l22 = Bx[0][0] + gam, det = l11 * l22 - gam * gam;
        ok = ok && det > 0.0;
        const double v11 = l22 / det, v22 = l11 / det, v12 = gam / det;
        const double m1 = v11 * ay[0] + v12 * ax[0], m2 = v12 * ay[0] + v22 * ax[0];
        const double psi = (m1 - m2) * (m1 - m2) + v11 + v22 - 2.0 * v12, b = psi * A;
        p.stat[(long long)w[W_C1] * p.RS + r] = psi;   // (OP_GCV_ZMARG and OP_GCV_PREC read it; the node's joint term writes the same number again)
        double en = 0.0, em = 0.0, ev = 0.0;
        for (int i = 0; i < n; ++i) {
            const double ep = 1.4142135623730951 * gx[i], ec = gw[i] * exp(-0.5 * (kappa * ep + b * exp(-kappa * ep)) + 0.5 * ep * ep);
            em += ep * ec;
            en += ec;
        }
        em /= en;
        for (int i = 0; i < n; ++i) {
            const double ep = 1.4142135623730951 * gx[i], ec = gw[i] * exp(-0.5 * (kappa * ep + b * exp(-kappa * ep)) + 0.5 * ep * ep);
            ev += ec * (ep - em) * (ep - em);
        }
        ev /= en;
        ok = ok && ev > 0.0 && ev < 1.0e300;
        double mo[N], Vo[N][N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            mo[i] = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) Vo[i][j] = i == j ? 1.0 : 0.0;
        }
        mo[0] = em;
        Vo[0][0] = ev;
        store_msg<N, STRAND>(p, w[W_OUT], 1, r, mo, Vo, fl, reg);
    } break;
    case OP_GCV_ZMARG: {   // q(z) ∝ ELQ(z) · (the product of all OTHER messages into z, W_IN0): first two moments by cubature against that product
        double az[N], Bz[N][N];
        ok = load_msg<N, STRAND>(p, w[W_IN0], fl & F_IN0_WP, false, 1, r, az, Bz, reg);
        const double* c = p.cpool + w[W_C0];
        const double kappa = c[0], b = p.stat[(long long)w[W_C1] * p.RS + r] * exp(-c[1]);
        const int n = (int)c[2];
        const double *gx = c + 3, *gw = c + 3 + n;
        const double zm = az[0], sc = sqrt(2.0 * Bz[0][0]);
        double nrm = 0.0, mean = 0.0, var = 0.0;
        for (int i = 0; i < n; ++i) {
            const double pt = zm + sc * gx[i], cv = gw[i] * exp(-0.5 * (kappa * pt + b * exp(-kappa * pt)));
            mean += pt * cv;
            nrm += cv;
        }
        mean /= nrm;
        for (int i = 0; i < n; ++i) {
            const double pt = zm + sc * gx[i], cv = gw[i] * exp(-0.5 * (kappa * pt + b * exp(-kappa * pt)));
            var += cv * (pt - mean) * (pt - mean);
        }
        var /= nrm;
        ok = ok && var > 0.0 && var < 1.0e300;
        p.marg[(long long)w[W_OUT] * p.RS + r] = mean;
        p.marg[(long long)(w[W_OUT] + 1) * p.RS + r] = var;
        p.marg[(long long)(w[W_OUT] + 2) * p.RS + r] = log(var);
    } break;
    case OP_CAT_UPDATE: {   // q(z): W_OUT π[K] (precision-state array), W_VAL `out` (value, or F_VAL_MARG: its marginal), W_IN0 q(s) state α | E log s (−1: log p at W_C0),
                            // list: per component (marginal of m_k | −1 − constant value, state of p_k | −1 − constant noise block)
        const int K = w[W_N];
        const int* lst = p.aux + w[W_LIST];
        double y[N], Cy[N][N], u0;
        if (fl & F_VAL_MARG) load_marginal<N>(p, w[W_VAL], false, 0, 0, d, r, true, y, Cy, u0);
        else {
            load_value<N>(p, w[W_VAL], fl & F_VAL_SLOT, d, r, y);
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int j = 0; j < N; ++j) Cy[i][j] = 0.0;
        }
        double mx = -__builtin_huge_val();
        for (int k = 0; k < K; ++k) {
            double m[N], C[N][N], Wm[N][N], elw;
            const int mo = lst[2 * k], po = lst[2 * k + 1];
            if (mo >= 0) load_marginal<N>(p, mo, false, 0, 0, d, r, true, m, C, u0);
            else {
                load_value<N>(p, -1 - mo, false, d, r, m);
#pragma unroll
                for (int i = 0; i < N; ++i)
#pragma unroll
                    for (int j = 0; j < N; ++j) C[i][j] = 0.0;
            }
            if (po >= 0) {
                ld_full<N>(p.prec, po + 1 + d * (d + 1) / 2, d, p.RS, r, 0.0, Wm);
                elw = p.prec[(long long)(po + 1 + d * (d + 1) / 2 + 2 * d * d) * p.RS + r];
            } else {
                const double* c = p.cpool + (-1 - po);
                ld_cmat<N>(c + d * d, d, d, 0.0, Wm);
                elw = c[2 * d * d];
            }
            double q = 0.0;
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int j = 0; j < N; ++j) q += (i < d && j < d) ? Wm[i][j] * ((y[i] - m[i]) * (y[j] - m[j]) + C[i][j] + Cy[i][j]) : 0.0;
            const double els = w[W_IN0] >= 0 ? p.prec[(long long)(w[W_IN0] + K + k) * p.RS + r] : p.cpool[w[W_C0] + k];
            const double lg = els - 0.5 * (d * T_LOG2PI - elw + q);
            p.prec[(long long)(w[W_OUT] + k) * p.RS + r] = lg;
            mx = lg > mx ? lg : mx;
        }
        double Z = 0.0;
        for (int k = 0; k < K; ++k) {
            const double e = exp(p.prec[(long long)(w[W_OUT] + k) * p.RS + r] - mx);
            p.prec[(long long)(w[W_OUT] + k) * p.RS + r] = e;
            Z += e;
        }
        for (int k = 0; k < K; ++k) p.prec[(long long)(w[W_OUT] + k) * p.RS + r] /= Z;
        ok = Z > 0.0 && Z < 1.0e300;
    } break;
    case OP_LEAF: {
        double v[N], Sg[N][N], Wm[N][N], el;
        if (fl & F_VAL_MARG) ld_vec<N>(p.marg, w[W_VAL], d, p.RS, r, v);   // (the marginal of the PREVIOUS iteration: every marginal op of the sweep comes after every leaf)
        else load_value<N>(p, w[W_VAL], fl & F_VAL_SLOT, d, r, v);
        const bool wp = fl & F_OUT_WP;
        load_noise<N>(p, w, d, r, !wp, wp, Sg, Wm, el);
        if (wp) {
            double xi[N];
            matvec<N>(Wm, v, xi);
            if (fl & F_WEIGHT) {   // a mixture component: (π E[W] v, π E[W])
                const double wt = p.prec[(long long)w[W_LIST] * p.RS + r];
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    xi[i] *= wt;
#pragma unroll
                    for (int j = 0; j < N; ++j) Wm[i][j] *= wt;
                }
            }
            if (fl & F_MAY_MISS) {   // a `missing` observation sends nothing: the zero of the precision form (the compiler keeps such leaves in it)
                bool miss = false;
#pragma unroll
                for (int i = 0; i < N; ++i) miss = miss || (i < d && v[i] != v[i]);
                if (miss) {
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        xi[i] = 0.0;
#pragma unroll
                        for (int j = 0; j < N; ++j) Wm[i][j] = 0.0;
                    }
                }
            }
            store_msg<N, STRAND>(p, w[W_OUT], d, r, xi, Wm, fl, reg);
        } else
            store_msg<N, STRAND>(p, w[W_OUT], d, r, v, Sg, fl, reg);
    } break;
    case OP_NOISE: {
        double a[N], B[N][N], Sg[N][N], Wm[N][N], el;
        const bool wp = fl & F_IN0_WP;
        ok = load_msg<N, STRAND>(p, w[W_IN0], wp, wp, d, r, a, B, reg);
        load_noise<N>(p, w, d, r, !wp, wp, Sg, Wm, el);
        if (!wp) {
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int j = 0; j < N; ++j) B[i][j] += Sg[i][j];
            if (fl & F_OUT_WP) {   // every reader multiplies or marginalises in precision form: converted once here instead of once per reader
                double Bi[N][N], t[N], ld;
                ok = spd_inv<N>(B, Bi, ld) && ok;
                matvec<N>(Bi, a, t);
                store_msg<N, STRAND>(p, w[W_OUT], d, r, t, Bi, fl, reg);
            } else
                store_msg<N, STRAND>(p, w[W_OUT], d, r, a, B, fl, reg);
        } else {   // Λ' = Λ (Λ + W)⁻¹ W, ξ' = W (Λ + W)⁻¹ ξ: defined for a rank-deficient Λ, equal to (Λ⁻¹ + Σ)⁻¹ otherwise
            double G[N][N], Gi[N][N], t[N], xo[N], T1[N][N], Lo[N][N], ld;
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int j = 0; j < N; ++j) G[i][j] = B[i][j] + Wm[i][j];
            ok = spd_inv<N>(G, Gi, ld) && ok;
            matvec<N>(Gi, a, t);
            matvec<N>(Wm, t, xo);
            matmul<N>(Gi, Wm, T1);
            matmul<N>(B, T1, Lo);
            store_msg<N, STRAND>(p, w[W_OUT], d, r, xo, Lo, fl, reg);
        }
    } break;
    case OP_MUL_OUT: {   // in: dimension d1 (moment form), out: dimension d
        double a[N], V[N][N], A[N][N], m[N], T1[N][N], Vo[N][N];
        ok = load_msg<N, STRAND>(p, w[W_IN0], fl & F_IN0_WP, false, w[W_D1], r, a, V, reg);
        ld_cmat<N>(p.cpool + w[W_C0], d, w[W_D1], 0.0, A);
        matvec<N>(A, a, m);
        matmul<N>(A, V, T1);
        matmulT<N>(T1, A, Vo);
        store_msg<N, STRAND>(p, w[W_OUT], d, r, m, Vo, fl, reg);
    } break;
    case OP_MUL_IN: {    // in: the message toward `out`, dimension d (precision form); out: dimension d1
        double xi[N], L[N][N], A[N][N], xo[N], T1[N][N], Lo[N][N];
        ok = load_msg<N, STRAND>(p, w[W_IN0], fl & F_IN0_WP, true, d, r, xi, L, reg);
        ld_cmat<N>(p.cpool + w[W_C0], d, w[W_D1], 0.0, A);
#pragma unroll
        for (int i = 0; i < N; ++i)   // the padding of Λ beyond d meets zero rows of A
            if (i >= d) L[i][i] = 0.0;
        matTvec<N>(A, xi, xo);
        matTmul<N>(A, L, T1);
        matmul<N>(T1, A, Lo);
        store_msg<N, STRAND>(p, w[W_OUT], w[W_D1], r, xo, Lo, fl, reg);
    } break;
    case OP_ADD_OUT:
    case OP_ADD_IN: {
        if (op == OP_ADD_IN && (fl & F_IN0_WP)) {
            // the message from `out` in precision form (ξo, Λo), the other input as (ξ2, W2): Λ' = Λo (Λo + W2)⁻¹ W2, ξ' = W2 (Λo + W2)⁻¹ (ξo + ξ2) − ξ2 —
            // N(m_out − m2, V_out + V2) wherever V_out exists, and defined for a rank-deficient Λo (an observation map with fewer rows than columns
            // behind the `+`), where the moment form — and the reference's rule — is not
            double xo[N], Lo[N][N], x2[N], W2[N][N];
            ok = load_msg<N, STRAND>(p, w[W_IN0], true, true, d, r, xo, Lo, reg);
            ok = load_msg<N, STRAND>(p, w[W_IN1], fl & F_IN1_WP, true, d, r, x2, W2, reg) && ok;
            double G[N][N], Gi[N][N], t[N], s[N], xn[N], T1[N][N], Ln[N][N], ld;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                s[i] = xo[i] + x2[i];
#pragma unroll
                for (int j = 0; j < N; ++j) G[i][j] = Lo[i][j] + W2[i][j] - ((i >= d && i == j) ? 1.0 : 0.0);   // (both pads are 1 on the diagonal: keep one)
            }
            ok = spd_inv<N>(G, Gi, ld) && ok;
            matvec<N>(Gi, s, t);
            matvec<N>(W2, t, xn);
#pragma unroll
            for (int i = 0; i < N; ++i) xn[i] -= x2[i];
            matmul<N>(Gi, W2, T1);
            matmul<N>(Lo, T1, Ln);
            store_msg<N, STRAND>(p, w[W_OUT], d, r, xn, Ln, fl, reg);
            break;
        }
        double a0[N], V0[N][N], a1[N], V1[N][N];
        ok = load_msg<N, STRAND>(p, w[W_IN0], fl & F_IN0_WP, false, d, r, a0, V0, reg);
        ok = load_msg<N, STRAND>(p, w[W_IN1], fl & F_IN1_WP, false, d, r, a1, V1, reg) && ok;
        const double sg = op == OP_ADD_OUT ? 1.0 : -1.0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            a0[i] += sg * a1[i];
#pragma unroll
            for (int j = 0; j < N; ++j) V0[i][j] += V1[i][j];
        }
        store_msg<N, STRAND>(p, w[W_OUT], d, r, a0, V0, fl, reg);
    } break;
    case OP_SHIFT: {
        double a[N], B[N][N], c[N];
        const bool wp = fl & F_IN0_WP;
        ok = load_msg<N, STRAND>(p, w[W_IN0], wp, wp, d, r, a, B, reg);
        load_value<N>(p, w[W_VAL], fl & F_VAL_SLOT, d, r, c);
        const double sg = (fl & F_NEG) ? -1.0 : 1.0;
        if (wp) {
            double t[N];
#pragma unroll
            for (int i = 0; i < N; ++i)
                if (i >= d) B[i][i] = 0.0;
            matvec<N>(B, c, t);
#pragma unroll
            for (int i = 0; i < N; ++i) a[i] += sg * t[i];
        } else {
#pragma unroll
            for (int i = 0; i < N; ++i) a[i] += sg * c[i];
        }
        store_msg<N, STRAND>(p, w[W_OUT], d, r, a, B, fl, reg);
    } break;
    case OP_PRODUCT:
    case OP_MARGINAL: {
        double xi[N], L[N][N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            xi[i] = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) L[i][j] = 0.0;
        }
        const int n = w[W_N];
        const int* lst = p.aux + w[W_LIST];
        // a marginal of ONE message in moment form IS the message: taken through the precision and back, the two inversions would square the condition
        // number in the error (the unobserved end of a `*` / `+` chain); the inversion below then only supplies log|V|
        bool single = op == OP_MARGINAL && n == 1 && lst[1] == 0;
        int at = 0;
        if (op == OP_MARGINAL && (fl & F_MAY_MISS) && !single) {   // … and so is the marginal of one moment-form message and `missing` observations (zeros of the precision form)
            int n_mv = 0;
            bool info = false;
            for (int q = 0; q < n; ++q) {
                if (lst[2 * q + 1] == 0) {
                    ++n_mv;
                    at = q;
                    continue;
                }
                double a[N], B[N][N];
                load_msg<N, STRAND>(p, lst[2 * q], true, true, d, r, a, B, reg);
#pragma unroll
                for (int i = 0; i < N; ++i) info = info || (i < d && B[i][i] != 0.0);
            }
            single = n_mv == 1 && !info;
        }
        for (int q = 0; q < n; ++q) {   // left to right, in factor order (MessagesProductFromLeftToRight)
            if (single && q != at) continue;
            double a[N], B[N][N];
            ok = load_msg<N, STRAND>(p, lst[2 * q], lst[2 * q + 1] != 0, !single, d, r, a, B, reg) && ok;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                xi[i] += a[i];
#pragma unroll
                for (int j = 0; j < N; ++j) L[i][j] += (i < d && j < d) ? B[i][j] : 0.0;
            }
        }
        if (op == OP_PRODUCT) {
            store_msg<N, STRAND>(p, w[W_OUT], d, r, xi, L, fl, reg);
        } else {
#pragma unroll
            for (int i = 0; i < N; ++i)
                if (i >= d) L[i][i] = 1.0;
            double V[N][N], m[N], ld;
            ok = spd_inv<N>(L, V, ld) && ok;
            matvec<N>(V, xi, m);
            if (single) {
                load_msg<N, STRAND>(p, lst[2 * at], false, false, d, r, m, V, reg);
                ld = -ld;
            }
            st_vec<N>(p.marg, w[W_OUT], d, p.RS, r, m);
            st_sym<N>(p.marg, w[W_OUT] + d, d, p.RS, r, V);
            p.marg[(w[W_OUT] + d + d * (d + 1) / 2) * p.RS + r] = -ld;
        }
    } break;
    default: break;
    }
    if (!ok) atomicOr(p.status, 1);
}
// LIGHT: the instance for graphs without a `+` of two random inputs and without precision variables — the log-gamma / digamma code of OP_PREC_UPDATE and the
// two-inverse algebra of OP_FE_ADD2 would otherwise set the register budget of every Bethe term (320 VGPRs against the light instance's; tree_engine.hip picks)
template <int N, bool LIGHT = false>
__device__ __forceinline__ void eval_fe(const TreeParams& p, const int* __restrict__ w, long long r) {
    const int op = w[W_OP], d = w[W_D0], fl = w[W_FLAGS];
    bool ok = true;
    switch (op) {
#ifdef RXHIP_HOST_EMUL   // (the executor emits OP_FE_NOISE2M for these kernels; the message-only form stays as the host differential test's reference of the LDS body)
    case OP_FE_NOISE2: {
        // joint precision [[Lo + W, −W], [−W, Lm + W]]; with P = Lo + W, S = (Lm + W) − W P⁻¹ W:  log|J| = log|P| + log|S|,
        // V_μμ = S⁻¹, V_oμ = P⁻¹ W S⁻¹, V_oo = P⁻¹ + P⁻¹ W S⁻¹ W P⁻¹;  r = out − μ
        double xo[N], Lo[N][N], xm[N], Lm[N][N], Sg[N][N], Wm[N][N], el;
        if (w[W_IN0] >= 0) ok = load_msg<N>(p, w[W_IN0], fl & F_IN0_WP, true, d, r, xo, Lo);
        if (w[W_IN1] >= 0) ok = load_msg<N>(p, w[W_IN1], fl & F_IN1_WP, true, d, r, xm, Lm) && ok;
        load_noise<N>(p, w, d, r, false, true, Sg, Wm, el);
        double P[N][N], Pi[N][N], S[N][N], Si[N][N], ldP, ldS;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (w[W_IN0] < 0) xo[i] = 0.0;
            if (w[W_IN1] < 0) xm[i] = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const bool in = i < d && j < d;
                const double lo = (w[W_IN0] >= 0 && in) ? Lo[i][j] : 0.0, lm = (w[W_IN1] >= 0 && in) ? Lm[i][j] : 0.0;
                P[i][j] = lo + Wm[i][j];
                S[i][j] = lm + Wm[i][j];
            }
        }
        ok = spd_inv<N>(P, Pi, ldP) && ok;
        double PW[N][N], T1[N][N];
        matmul<N>(Pi, Wm, PW);        // P⁻¹ W
        matTmul<N>(Wm, PW, T1);       // W P⁻¹ W (W symmetric)
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) S[i][j] -= (i < d && j < d) ? T1[i][j] : 0.0;
        ok = spd_inv<N>(S, Si, ldS) && ok;
        // means: m_μ = S⁻¹ (ξ_μ + W P⁻¹ ξ_o), m_o = P⁻¹ (ξ_o + W m_μ)
        double t0[N], t1[N], mm[N], mo[N];
        matvec<N>(Pi, xo, t0);
        matvec<N>(Wm, t0, t1);
#pragma unroll
        for (int i = 0; i < N; ++i) t1[i] += xm[i];
        matvec<N>(Si, t1, mm);
        matvec<N>(Wm, mm, t0);
#pragma unroll
        for (int i = 0; i < N; ++i) t0[i] += xo[i];
        matvec<N>(Pi, t0, mo);
        // Cov(r) = V_oo − V_oμ − V_μo + V_μμ = P⁻¹ + (P⁻¹W − I) S⁻¹ (P⁻¹W − I)ᵀ
        double Dm[N][N], T2[N][N], E[N][N];
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) Dm[i][j] = PW[i][j] - (i == j ? 1.0 : 0.0);
        matmul<N>(Dm, Si, T2);
        matmulT<N>(T2, Dm, E);
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) E[i][j] += Pi[i][j] + (mo[i] - mm[i]) * (mo[j] - mm[j]);
        const double H = 0.5 * (2.0 * d * (T_LOG2PI + 1.0) - (ldP + ldS));
        double term = -H;
        if (fl & F_STAT) st_full<N>(p.stat, w[W_C1], d, p.RS, r, E);
        else term += 0.5 * (d * T_LOG2PI - el + trace_prod<N>(Wm, E, d));
        p.term[(long long)w[W_TERM] * p.RS + r] = term;
    } break;
#endif
    case OP_MARG_PUSH: {
        // q(out) of a deterministic node out = A·in IS the image of q(in) (exact on a tree): no product of the two messages on out's edges, so those
        // messages need not exist for the marginal's sake (the compiler drops the ones nobody else reads).  Only the Bethe terms and a caller who asks for
        // this (anonymous) variable read it: second phase.  A V Aᵀ singular (more rows than columns): log-determinant −∞, as the message route's entropy.
        double m[N], V[N][N], A[N][N], mo[N], T1[N][N], Vo[N][N], Vi[N][N], ld;
        const int din = w[W_D1];
        ld_vec<N>(p.marg, w[W_IN0], din, p.RS, r, m);
        ld_sym<N>(p.marg, w[W_IN0] + din, din, p.RS, r, 0.0, V);
        ld_cmat<N>(p.cpool + w[W_C0], d, din, 0.0, A);
        matvec<N>(A, m, mo);
        matmul<N>(A, V, T1);
        matmulT<N>(T1, A, Vo);
        st_vec<N>(p.marg, w[W_OUT], d, p.RS, r, mo);
        st_sym<N>(p.marg, w[W_OUT] + d, d, p.RS, r, Vo);
        if (w[W_IN1] >= 0) {   // a square map: log|A V Aᵀ| = log|V| + 2 log|det A|
            p.marg[(long long)(w[W_OUT] + d + d * (d + 1) / 2) * p.RS + r] = p.marg[(long long)(w[W_IN0] + din + din * (din + 1) / 2) * p.RS + r] + p.cpool[w[W_IN1]];
            break;
        }
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (i >= d) Vo[i][i] = 1.0;
        const bool pd = spd_inv<N>(Vo, Vi, ld);
        p.marg[(long long)(w[W_OUT] + d + d * (d + 1) / 2) * p.RS + r] = pd ? ld : -__builtin_huge_val();
    } break;
    case OP_FE_NOISE2M: {
        // The same joint q(a, b) of the node's two Gaussian interfaces, from what the sweep has already computed: with P = L_a + W (L_a: the message the
        // variable on side a sends to the node) the Schur complement S = L_b + W − W P⁻¹ W is the MARGINAL precision of b, so S⁻¹ = V_b and log|S| = −log|V_b| are
        // the stored marginal, and the means are the marginal means:  log|J| = log|P| − log|V_b|,  Cov(a − b) = P⁻¹ + (P⁻¹W − I) V_b (P⁻¹W − I)ᵀ.
        // One inverse and three products where the message-only form takes up to three inverses and five products.  Side a is the interface whose message is
        // stored in precision form (compiler's choice: no conversion); W_VAL / W_VAL2: the marginal slots of a and b.
        double xa[N], La[N][N], Sg[N][N], Wm[N][N], el;
        if (w[W_IN0] >= 0) ok = load_msg<N>(p, w[W_IN0], fl & F_IN0_WP, true, d, r, xa, La);
        load_noise<N>(p, w, d, r, false, true, Sg, Wm, el);
        double P[N][N], Pi[N][N], ldP;
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) P[i][j] = ((w[W_IN0] >= 0 && i < d && j < d) ? La[i][j] : 0.0) + Wm[i][j];
        ok = spd_inv<N>(P, Pi, ldP) && ok;
        double ma[N], mb[N], Vb[N][N], ldVb, unused;
        load_marginal<N>(p, w[W_VAL], fl & F_PUSH_A, w[W_IN1], w[W_LIST], d, r, false, ma, Vb, unused);
        load_marginal<N>(p, w[W_VAL2], fl & F_PUSH_B, w[W_IN2], w[W_N], d, r, true, mb, Vb, ldVb, (fl & F_PUSH_B) ? w[W_D1] : -1);
        if (fl & F_JOINT_B) {
            double xb[N], Lb[N][N], T1[N][N], T3[N][N], S[N][N], Si[N][N], t[N], u[N], ldS;
            ok = load_msg<N>(p, w[W_IN1], fl & F_IN1_WP, true, d, r, xb, Lb) && ok;
            matmul<N>(Pi, Wm, T1);
            matmul<N>(Wm, T1, T3);
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int j = 0; j < N; ++j) S[i][j] = (i < d && j < d) ? Lb[i][j] + Wm[i][j] - T3[i][j] : (i == j ? 1.0 : 0.0);
            ok = spd_inv<N>(S, Si, ldS) && ok;
            matvec<N>(Pi, xa, t);
            matvec<N>(Wm, t, u);
#pragma unroll
            for (int i = 0; i < N; ++i) u[i] = (i < d) ? u[i] + xb[i] : 0.0;
            matvec<N>(Si, u, mb);
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int j = 0; j < N; ++j) Vb[i][j] = Si[i][j];
            ldVb = -ldS;
        }
        if (fl & F_JOINT_MEAN) {
            double t[N], u[N];
            matvec<N>(Wm, mb, t);
#pragma unroll
            for (int i = 0; i < N; ++i) t[i] = (i < d) ? t[i] + ((w[W_IN0] >= 0) ? xa[i] : 0.0) : 0.0;
            matvec<N>(Pi, t, u);
#pragma unroll
            for (int i = 0; i < N; ++i) ma[i] = u[i];
        }
        double Dm[N][N], T2[N][N], E[N][N];
        matmul<N>(Pi, Wm, Dm);
#pragma unroll
        for (int i = 0; i < N; ++i) Dm[i][i] -= 1.0;
        matmul<N>(Dm, Vb, T2);
        matmulT<N>(T2, Dm, E);
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) E[i][j] += Pi[i][j] + (ma[i] - mb[i]) * (ma[j] - mb[j]);
        const double H = 0.5 * (2.0 * d * (T_LOG2PI + 1.0) - (ldP - ldVb));
        double term = -H;
        if (fl & F_FOLD_ENT) term += (double)w[W_OUT] * 0.5 * (d * (T_LOG2PI + 1.0) + ldVb);   // the side-b variable's own Bethe entropy term, folded in (its log|V| is at hand)
        if (fl & F_STAT) st_full<N>(p.stat, w[W_C1], d, p.RS, r, E);
        else term += 0.5 * (d * T_LOG2PI - el + trace_prod<N>(Wm, E, d));
        p.term[(long long)w[W_TERM] * p.RS + r] = term;
    } break;
    case OP_FE_NOISE_MF: {
        // @average_energy of the Gaussian node with both marginals (docs/src/manuals/inference/create-node.md:200-232 for the macro surface): U = ½[d log 2π −
        // (E) log|W| + tr(W E[(out − μ)(out − μ)ᵀ])], E[rrᵀ] = V_out + V_μ + (m_out − m_μ)(m_out − m_μ)ᵀ; the clusters' entropies −H[q(out)] − H[q(μ)] are booked
        // with the variables' own terms (OP_FE_ENT coefficients).  Random precision: E[rrᵀ] goes to the statistics of q(W).
        double Sg[N][N], Wm[N][N], el, ma[N], mb[N], Va[N][N], Vb[N][N], E[N][N], u0, u1;
        load_noise<N>(p, w, d, r, false, true, Sg, Wm, el);
        load_marginal<N>(p, w[W_VAL], false, 0, 0, d, r, true, ma, Va, u0);
        load_marginal<N>(p, w[W_VAL2], false, 0, 0, d, r, true, mb, Vb, u1);
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) E[i][j] = Va[i][j] + Vb[i][j] + (ma[i] - mb[i]) * (ma[j] - mb[j]);
        double term = 0.0;
        const double wt = (!LIGHT && (fl & F_WEIGHT)) ? p.prec[(long long)w[W_LIST] * p.RS + r] : 1.0;   // (a mixture component)
        if (!LIGHT && (fl & F_WEIGHT) && (fl & F_STAT)) {
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int j = 0; j < N; ++j) E[i][j] *= wt;
        }
        if (fl & F_STAT) st_full<N>(p.stat, w[W_C1], d, p.RS, r, E);
        else term = wt * 0.5 * (d * T_LOG2PI - el + trace_prod<N>(Wm, E, d));
        p.term[(long long)w[W_TERM] * p.RS + r] = term;
    } break;
    case OP_FE_NOISE1:
    case OP_FE_NOISE0: {
        double E[N][N], Sg[N][N], Wm[N][N], el, rv[N], H = 0.0;
        load_noise<N>(p, w, d, r, false, true, Sg, Wm, el);
        if (op == OP_FE_NOISE1) {
            double m[N], V[N][N], c[N], ldV;
            load_marginal<N>(p, w[W_IN0], fl & F_PUSH_A, w[W_IN1], w[W_D1], d, r, true, m, V, ldV, (fl & F_PUSH_A) ? w[W_IN2] : -1);
            H = 0.5 * (d * (T_LOG2PI + 1.0) + ldV);
            if (fl & F_FOLD_ENT) H *= (double)(1 - w[W_OUT]);   // (−H of the node + coef·H of the variable, folded: −(1 − coef)·H below)
            load_value<N>(p, w[W_VAL], fl & F_VAL_SLOT, d, r, c);
#pragma unroll
            for (int i = 0; i < N; ++i) rv[i] = m[i] - c[i];
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int j = 0; j < N; ++j) E[i][j] = V[i][j] + rv[i] * rv[j];
        } else {
            double a[N], b[N];
            load_value<N>(p, w[W_VAL], fl & F_VAL_SLOT, d, r, a);
            load_value<N>(p, w[W_VAL2], fl & F_VAL2_SLOT, d, r, b);
#pragma unroll
            for (int i = 0; i < N; ++i) rv[i] = a[i] - b[i];
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int j = 0; j < N; ++j) E[i][j] = rv[i] * rv[j];
        }
        double term = -H;
        bool miss = false;   // a `missing` observation: the node's energy and the entropy of the predicted value cancel — what is left is −H of the random interface
        if (fl & F_MAY_MISS) {
#pragma unroll
            for (int i = 0; i < N; ++i) miss = miss || (i < d && rv[i] != rv[i]);
        }
        const double wt = (!LIGHT && (fl & F_WEIGHT)) ? p.prec[(long long)w[W_LIST] * p.RS + r] : 1.0;   // (a mixture component: energy and moments × π_k, the entropy as it is)
        if (!LIGHT && (fl & F_WEIGHT) && (fl & F_STAT)) {
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int j = 0; j < N; ++j) E[i][j] *= wt;
        }
        if (fl & F_STAT) st_full<N>(p.stat, w[W_C1], d, p.RS, r, E);
        else if (!miss) term += wt * 0.5 * (d * T_LOG2PI - el + trace_prod<N>(Wm, E, d));
        p.term[(long long)w[W_TERM] * p.RS + r] = term;
    } break;
    case OP_FE_ENT: {
        double ldV;
        if (fl & F_PUSH_A) {
            double m[N], V[N][N];
            load_marginal<N>(p, w[W_IN0], true, w[W_C0], w[W_D1], d, r, true, m, V, ldV, w[W_IN1]);
        } else
            ldV = p.marg[(w[W_IN0] + d + d * (d + 1) / 2) * p.RS + r];
        p.term[(long long)w[W_TERM] * p.RS + r] = (double)w[W_N] * 0.5 * (d * (T_LOG2PI + 1.0) + ldV);
    } break;
    case OP_FE_ADD2: if (!LIGHT) {   // Lj = [[L1 + Lo, Lo], [Lo, L2 + Lo]]: log|Lj| = log|L1 + Lo| + log|L2 + Lo − Lo (L1 + Lo)⁻¹ Lo|
        double x[N], L1[N][N], L2[N][N], Lo[N][N];
        if (w[W_IN0] >= 0) ok = load_msg<N>(p, w[W_IN0], fl & F_IN0_WP, true, d, r, x, L1);
        if (w[W_IN1] >= 0) ok = load_msg<N>(p, w[W_IN1], fl & F_IN1_WP, true, d, r, x, L2) && ok;
        if (w[W_IN2] >= 0) ok = load_msg<N>(p, w[W_IN2], fl & F_IN2_WP, true, d, r, x, Lo) && ok;
        double P[N][N], Pi[N][N], S[N][N], Si[N][N], T1[N][N], T2[N][N], ldP, ldS;
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const bool in = i < d && j < d;
                const double l1 = (w[W_IN0] >= 0 && in) ? L1[i][j] : 0.0, l2 = (w[W_IN1] >= 0 && in) ? L2[i][j] : 0.0, lo = (w[W_IN2] >= 0 && in) ? Lo[i][j] : 0.0;
                Lo[i][j] = lo;
                P[i][j] = l1 + lo + ((!in && i == j) ? 1.0 : 0.0);
                S[i][j] = l2 + lo + ((!in && i == j) ? 1.0 : 0.0);
            }
        ok = spd_inv<N>(P, Pi, ldP) && ok;
        matmul<N>(Pi, Lo, T1);
        matmul<N>(Lo, T1, T2);
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) S[i][j] -= T2[i][j];
        ok = spd_inv<N>(S, Si, ldS) && ok;
        p.term[(long long)w[W_TERM] * p.RS + r] = -0.5 * (2.0 * d * (T_LOG2PI + 1.0) - (ldP + ldS));
    } break;
    case OP_GCV_PREC: if (!LIGHT) {   // W_IN0: the marginal slot of z; W_C1: ψ (the residual moment the node's joint term left); W_C0: κ | ω
        const double* c = p.cpool + w[W_C0];
        const double m = p.marg[(long long)w[W_IN0] * p.RS + r], v = p.marg[(long long)(w[W_IN0] + 1) * p.RS + r], psi = p.stat[(long long)w[W_C1] * p.RS + r];
        const double eg = exp(-c[1] - c[0] * m + 0.5 * c[0] * c[0] * v), elg = -(c[0] * m + c[1]);
        const int ps = w[W_PREC];
        p.prec[(long long)(ps + 2) * p.RS + r] = eg;
        p.prec[(long long)(ps + 3) * p.RS + r] = 1.0 / eg;
        p.prec[(long long)(ps + 4) * p.RS + r] = elg;
        p.term[(long long)w[W_TERM] * p.RS + r] = 0.5 * (T_LOG2PI - elg + eg * psi);
    } break;
    case OP_DIR_UPDATE: if (!LIGHT) {   // W_N K, list: the π slots of W_VAL2 switches; W_PREC the state α | E log s with the prior's concentrations at W_C0 (−1: constant log p at W_C0)
        const int K = w[W_N], nz = w[W_VAL2], ps = w[W_PREC];
        const int* lst = p.aux + w[W_LIST];
        double F = 0.0;
        for (int i = 0; i < nz; ++i)   // −H[q(z_i)]
            for (int k = 0; k < K; ++k) {
                const double pk = p.prec[(long long)(lst[i] + k) * p.RS + r];
                F += pk > 0.0 ? pk * log(pk) : 0.0;
            }
        const double* c = p.cpool + w[W_C0];
        if (ps < 0) {
            for (int k = 0; k < K; ++k) {
                double sk = 0.0;
                for (int i = 0; i < nz; ++i) sk += p.prec[(long long)(lst[i] + k) * p.RS + r];
                F -= sk * c[k];
            }
        } else {
            double asum = 0.0, a0sum = 0.0;
            for (int k = 0; k < K; ++k) {
                double sk = 0.0;
                for (int i = 0; i < nz; ++i) sk += p.prec[(long long)(lst[i] + k) * p.RS + r];
                const double al = c[k] + sk;
                p.prec[(long long)(ps + k) * p.RS + r] = al;
                asum += al;
                a0sum += c[k];
            }
            const double dsum = t_digamma(asum);
            double lB = -lgamma(asum), lB0 = -lgamma(a0sum), Us = 0.0, Hs = 0.0;
            for (int k = 0; k < K; ++k) {
                const double al = p.prec[(long long)(ps + k) * p.RS + r], dg = t_digamma(al), els = dg - dsum;
                p.prec[(long long)(ps + K + k) * p.RS + r] = els;
                F -= (al - c[k]) * els;          // −Σ_i π_ik E log s_k
                lB += lgamma(al);
                lB0 += lgamma(c[k]);
                Us += (c[k] - 1.0) * els;
                Hs += (al - 1.0) * dg;
            }
            F += (lB0 - Us) - (lB + (asum - K) * dsum - Hs);
        }
        if (p.want_fe) p.term[(long long)w[W_TERM] * p.RS + r] = F;
    } break;
    case OP_SUM_TERMS: {
        const int n = w[W_N];
        const int* lst = p.aux + w[W_LIST];
        double s = 0.0;
        for (int q = 0; q < n; ++q) s += p.term[(long long)lst[q] * p.RS + r];
        p.term[(long long)w[W_TERM] * p.RS + r] = s;
    } break;
    case OP_PREC_UPDATE: if (!LIGHT) {
        // prior block at c0: ν0 | S0⁻¹ (d²) | log|S0|;  stats: list of d² slots;  state at W_PREC
        const double* c = p.cpool + w[W_C0];
        const double nu0 = c[0], ldS0 = c[1 + d * d];
        double S0i[N][N], S[N][N];
        ld_cmat<N>(c + 1, d, d, 1.0, S0i);
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) S[i][j] = 0.0;
        const int n = w[W_N];
        const int* lst = p.aux + w[W_LIST];
        const int stride = (fl & F_WEIGHT) ? 2 : 1;
        double cnt = 0.0;   // the nodes this precision hangs on: one each, a mixture component π_k (its moments arrive weighted)
        for (int q = 0; q < n; ++q) {
            double E[N][N];
            ld_full<N>(p.stat, lst[stride * q], d, p.RS, r, 0.0, E);
            cnt += (stride == 2 && lst[2 * q + 1] >= 0) ? p.prec[(long long)lst[2 * q + 1] * p.RS + r] : 1.0;
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int j = 0; j < N; ++j) S[i][j] += E[i][j];
        }
        double Vi[N][N], V[N][N], Ss[N][N], ldVi;
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) {
                Ss[i][j] = (i < d && j < d) ? 0.5 * (S[i][j] + S[j][i]) : 0.0;
                Vi[i][j] = S0i[i][j] + Ss[i][j];
            }
        ok = spd_inv<N>(Vi, V, ldVi);
        const double nu = nu0 + cnt, ldV = -ldVi;
        const int ps = w[W_PREC], tri = d * (d + 1) / 2;
        double What[N][N], Whi[N][N];
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) {
                What[i][j] = nu * V[i][j];
                Whi[i][j] = Vi[i][j] / nu;
            }
        const double elw = t_mvdigamma(0.5 * nu, d) + d * T_LOG2 + ldV;
        p.prec[(long long)ps * p.RS + r] = nu;
        st_sym<N>(p.prec, ps + 1, d, p.RS, r, V);
        st_full<N>(p.prec, ps + 1 + tri, d, p.RS, r, What);
        st_full<N>(p.prec, ps + 1 + tri + d * d, d, p.RS, r, Whi);
        p.prec[(long long)(ps + 1 + tri + 2 * d * d) * p.RS + r] = elw;
        if (p.want_fe) {
            // the n likelihood nodes: ½[n d log 2π − n E log|W| + tr(Ŵ Σ E[rrᵀ])]; the prior node U − H[q(W)] (the terms of noise_kernels.hpp, per precision variable)
            double F = 0.5 * (cnt * (d * T_LOG2PI - elw) + trace_prod<N>(What, Ss, d));
            F += -(0.5 * (nu0 - d - 1.0) * elw - 0.5 * trace_prod<N>(S0i, What, d) - 0.5 * nu0 * d * T_LOG2 - 0.5 * nu0 * ldS0 - t_mvlgamma(0.5 * nu0, d));
            F -= 0.5 * (d + 1.0) * ldV + 0.5 * d * (d + 1.0) * T_LOG2 + t_mvlgamma(0.5 * nu, d) - 0.5 * (nu - d - 1.0) * t_mvdigamma(0.5 * nu, d) + 0.5 * nu * d;
            p.term[(long long)w[W_TERM] * p.RS + r] = F;
        }
    } break;
    default: break;
    }
    if (!ok) atomicOr(p.status, 1);
}

// one launch per level: items (op, replica) over the grid
template <int N, int PHASE>   // PHASE 0: the sweep; 1: the Bethe terms / q(W) updates; 2: the same without OP_FE_ADD2 / OP_PREC_UPDATE (the light instance)
__device__ __forceinline__ void eval_op(const TreeParams& p, const int* __restrict__ w, long long r) {
    if (PHASE == 0) eval_bp<N>(p, w, r);
    else if (PHASE == 1) eval_fe<N, false>(p, w, r);
    else eval_fe<N, true>(p, w, r);
}
#ifndef RXHIP_FE_WAVES
#define RXHIP_FE_WAVES 2      // wavefronts per SIMD the light Bethe-phase instance is compiled for.  Measured at 65 536 replicas (plain / two-branch chain, Bethe phase):
                              // 2 → 1.12 / 1.44 ms, 3 → 1.31 / 1.71 (96 bytes of scratch), 4 → 1.74 / 2.09 (220 bytes); the strand kernel at 3 spills too
                              // (2.35 → 2.78 ms): profiles/r06/tree_occupancy.txt.  A/B: make variants/… EXTRA=-DRXHIP_FE_WAVES=…
#endif
#ifndef RXHIP_STRAND_WAVES
#define RXHIP_STRAND_WAVES 2  // … the strand kernel
#endif
template <int N, int PHASE>
__global__ void __launch_bounds__(256, (PHASE == 2 && N == 4) ? RXHIP_FE_WAVES : 1) k_tree_ops(TreeParams p, int op0, int op1) {
    const long long total = (long long)(op1 - op0) * p.R;
    for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long long)gridDim.x * blockDim.x) {
        const long long o = it / p.R, r = it - o * p.R;
        eval_op<N, PHASE>(p, p.ops + (size_t)(op0 + o) * OP_WORDS, r);
    }
}
// (sweep phase at N ≤ 4: up to 512 threads — eight wavefronts share a CU's level barriers; the Bethe phase and the 8×8 instance need the registers of 256)
// the whole schedule in one launch: a workgroup owns `rb` replicas (a multiple of 16: whole 128-byte lines of every slot) and walks the levels with a
// workgroup barrier between them — for deep, narrow graphs (a chain is three levels per time step) where a launch per level would cost more than the level
template <int N, int PHASE>
__global__ void __launch_bounds__((PHASE == 0 && N <= 4) ? 512 : 256) k_tree_levels(TreeParams p, const int* __restrict__ lvl_ptr, int l0, int l1, int rb) {
    const long long r0 = (long long)blockIdx.x * rb;
    const int nr = (int)((p.R - r0) < rb ? (p.R - r0) : rb);
    for (int l = l0; l < l1; ++l) {
        const int o0 = lvl_ptr[l], o1 = lvl_ptr[l + 1];
        const int total = (o1 - o0) * nr;
        for (int it = threadIdx.x; it < total; it += blockDim.x) {
            const int o = it / nr, r = it - o * nr;
            eval_op<N, PHASE>(p, p.ops + (size_t)(o0 + o) * OP_WORDS, r0 + r);
        }
        __syncthreads();   // (waits for the level's stores: every reader of the next level is in this workgroup)
    }
}
// strand schedule: the host cuts the sweep's op graph into STRANDS — paths of dependent ops along which every message has its next reader right behind it —
// and levels the strands (a strand starts when every message it reads from another strand is complete).  An item is (strand, replica block): a wavefront
// walks the strand's ops for 64 replicas and hands each message to the next op IN REGISTERS; a message goes to HBM only if somebody outside the strand
// (a marginal, a product elsewhere, the Bethe phase) reads it.  A chain: 2T + 1 leaf strands, then the forward and the backward recursion side by side,
// then the marginals — three launches, the wide levels at full occupancy, the two recursions without a barrier or a dependent load between their ops.
template <int N>
__global__ void __launch_bounds__(64, N == 4 ? RXHIP_STRAND_WAVES : 1) k_tree_strands(TreeParams p, const int* __restrict__ sops, const int* __restrict__ strands, int s0, int s1) {
    const long long nrb = (p.R + 63) / 64, total = (long long)(s1 - s0) * nrb;
    for (long long b = blockIdx.x; b < total; b += gridDim.x) {
        const long long s = b / nrb, r = (b - s * nrb) * 64 + threadIdx.x;
        if (r >= p.R) continue;
        const int o0 = strands[2 * (s0 + s)], n = strands[2 * (s0 + s) + 1];
        RegMsg<N> reg;
        for (int o = o0; o < o0 + n; ++o) eval_bp<N, true>(p, sops + (size_t)o * OP_WORDS, r, &reg);
    }
}
// a lane owns a replica and walks the ops of the range in order: no barriers, no cross-lane dependencies, every wavefront evaluates the same op for 64
// replicas with unit-stride loads — the schedule of LARGE batches (from a few waves per SIMD on it is bound by the messages' HBM traffic, not by latency)
template <int N, int PHASE>
__global__ void __launch_bounds__(64) k_tree_walk(TreeParams p, int op0, int op1) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= p.R) return;
    for (int o = op0; o < op1; ++o) eval_op<N, PHASE>(p, p.ops + (size_t)o * OP_WORDS, r);
}
}  // namespace tree
}  // namespace rxhip
