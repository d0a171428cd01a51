// model_envelope.hpp — how hard a state-space model is for the information-form schedules of the MFMA chain path (d > 4), decided on the host before an engine
// is built.  Those schedules carry filtered PRECISIONS and invert them with panel sweeps (dense_kernels.hpp blk_inverse: 16 pivots per panel above one tile); a
// precision that is nearly singular — a vague prior or a slowly forgetting transition in directions the observations do not see — costs them digits a
// covariance-form Kalman recursion (the oracle's, the reference's message order) does not lose.  Measured on 1 800 random models with noise scales over four
// decades against the oracle's Kalman / RTS restatement (scripts/calib_dense_envelope.py, profiles/r06/dense_envelope.txt):
//     κ = max( cond(V₀ₚ⁻¹ + BᵀQ⁻¹B),  cond((1 − ρ²) P⁻¹ + BᵀQ⁻¹B) ),   V₀ₚ the prior of the first observed state, ρ the spectral radius of A, cond of the matrix scaled to unit diagonal
// (the first filtered precision; the filtered precision a long chain drifts to where nothing is observed: P / (1 − ρ²) is what the transition lets the
// variance grow to) separates the models on which posteriors stay within 1e-6 sd / free energies within 1e-8 from those on which they do not:
// smoothing above one 16×16 tile (d > 16): every failure has κ > 10⁴ (up to 19 sd wrong at κ = 3·10⁵); at d ≤ 16: κ > 3·10⁵; filtering (rxhip_run_filter) is the
// more delicate of the two, most of all for a barely observed state (dy = 1 … 3): its failures start at κ = 1.75·10³ (d > 16; 4·10⁻⁵ sd) and 2.4·10³ (d ≤ 16; 7·10⁻⁷ sd,
// 0.08 sd at 7·10³ — the error grows by 30× per step inside the first segment and is gone behind its end).  The limits below sit under the lowest failure of the calibration sets.  KNOWN RESIDUAL: engine fuzz seed 6024880 (d = 32, dy = 5, ρ = 0.986, κ = 9.4·10²) filters 0.06 sd
// wrong INSIDE the limit — one in ≈ 9 000 dense models of the fuzz population; its smoothing is exact.  A limit of 8·10² would catch it and refuses four d = 64 models of the test
// suite that are fine (κ = 8 … 9·10²); a second indicator tried for it (the prior in units of the process noise) refused the suite's identity-like transitions with small
// noise, which are fine as well.  Neither was kept: κ is a fence, not a proof — the fix is a covariance-form update in kd_forward;
// the benchmark models sit at κ = 1.6 (C3) and the random models of the test suite at κ ≤ 700.
// rxhip_lgssm_create refuses beyond ENVELOPE_* with RXHIP_ERR_UNSUPPORTED and no handle — which rxhip_create (the graph entry point) answers by handing the same
// graph to the node-array executor, whose symmetric one-pivot sweeps hold 1e-10 on these models — unless rxhip_set_conditioning_guard(0) was called.
#pragma once
#include <cmath>
#include <vector>

namespace rxhip {
namespace envelope {

constexpr double ENVELOPE_ONE_TILE = 2.0e3;   // d ≤ 16
constexpr double ENVELOPE_TILES = 1.0e3;      // d > 16

// lower Cholesky factor in place (row-major n×n; the upper triangle is left alone); false: not positive definite
inline bool cholesky(std::vector<double>& a, int n) {
    for (int j = 0; j < n; ++j) {
        double s = a[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) s -= a[(size_t)j * n + k] * a[(size_t)j * n + k];
        if (!(s > 0.0) || !std::isfinite(s)) return false;
        const double l = std::sqrt(s);
        a[(size_t)j * n + j] = l;
        for (int i = j + 1; i < n; ++i) {
            double t = a[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) t -= a[(size_t)i * n + k] * a[(size_t)j * n + k];
            a[(size_t)i * n + j] = t / l;
        }
    }
    return true;
}
// x ← (L Lᵀ)⁻¹ x
inline void chol_solve(const std::vector<double>& L, int n, double* x) {
    for (int i = 0; i < n; ++i) {
        double t = x[i];
        for (int k = 0; k < i; ++k) t -= L[(size_t)i * n + k] * x[k];
        x[i] = t / L[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double t = x[i];
        for (int k = i + 1; k < n; ++k) t -= L[(size_t)k * n + i] * x[k];
        x[i] = t / L[(size_t)i * n + i];
    }
}
// the inverse of a symmetric positive definite matrix; false: not positive definite
inline bool spd_inverse(const double* m, int n, std::vector<double>& inv) {
    std::vector<double> L(m, m + (size_t)n * n);
    if (!cholesky(L, n)) return false;
    inv.assign((size_t)n * n, 0.0);
    std::vector<double> col(n);
    for (int j = 0; j < n; ++j) {
        for (int i = 0; i < n; ++i) col[i] = i == j ? 1.0 : 0.0;
        chol_solve(L, n, col.data());
        for (int i = 0; i < n; ++i) inv[(size_t)i * n + j] = col[i];
    }
    return true;
}
// largest eigenvalue of a symmetric positive semi-definite matrix (power iteration from a fixed start; 60 steps: the ratio of the two largest
// eigenvalues enters the guard only through a factor the thresholds have room for)
inline double lambda_max(const std::vector<double>& M, int n) {
    std::vector<double> x(n), y(n);
    for (int i = 0; i < n; ++i) x[i] = 1.0 + 0.37 * ((i * 7919) % 13);
    double lam = 0.0;
    for (int it = 0; it < 60; ++it) {
        double nrm = 0.0;
        for (int i = 0; i < n; ++i) {
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += M[(size_t)i * n + k] * x[k];
            y[i] = s;
            nrm += s * s;
        }
        nrm = std::sqrt(nrm);
        if (!(nrm > 0.0)) return 0.0;
        lam = nrm;
        double xn = 0.0;
        for (int i = 0; i < n; ++i) xn += x[i] * x[i];
        lam = nrm / std::sqrt(xn);
        for (int i = 0; i < n; ++i) x[i] = y[i] / nrm;
    }
    return lam;
}
// condition number of a symmetric positive definite matrix: λmax by power iteration, λmin by inverse iteration on its Cholesky factor; +∞ if it has none
inline double cond_spd(const std::vector<double>& M0, int n) {
    // of D^-1/2 M D^-1/2, D = diag M: state components in different units (a position in metres next to a rate in radians per second) are not what costs
    // digits — every sweep of the chain path scales by the diagonal first — and must not count (tests/test_badly_scaled_models_gpu.py: 10^±3 per component)
    std::vector<double> M(M0);
    for (int i = 0; i < n; ++i)
        if (!(M0[(size_t)i * n + i] > 0.0)) return INFINITY;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) M[(size_t)i * n + j] = M0[(size_t)i * n + j] / std::sqrt(M0[(size_t)i * n + i] * M0[(size_t)j * n + j]);
    std::vector<double> L(M);
    if (!cholesky(L, n)) return INFINITY;
    std::vector<double> x(n);
    for (int i = 0; i < n; ++i) x[i] = 1.0 + 0.37 * ((i * 7919) % 13);
    double mu = 0.0;   // largest eigenvalue of M⁻¹
    for (int it = 0; it < 60; ++it) {
        double xn = 0.0;
        for (int i = 0; i < n; ++i) xn += x[i] * x[i];
        xn = std::sqrt(xn);
        for (int i = 0; i < n; ++i) x[i] /= xn;
        chol_solve(L, n, x.data());
        double yn = 0.0;
        for (int i = 0; i < n; ++i) yn += x[i] * x[i];
        mu = std::sqrt(yn);
    }
    return lambda_max(M, n) * mu;
}
// spectral radius of a square matrix: ‖A^1024‖^(1/1024) by ten squarings, rescaled after each (Gelfand's formula; unlike ‖A‖₂ it does not see a change of the
// state's units, x → S x, A → S A S⁻¹: six decades between components move it by 1.4 %)
inline double spectral_radius(const double* A, int n) {
    std::vector<double> M(A, A + (size_t)n * n), T((size_t)n * n);
    double logs = 0.0;
    for (int it = 0; it < 10; ++it) {
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double s = 0.0;
                for (int k = 0; k < n; ++k) s += M[(size_t)i * n + k] * M[(size_t)k * n + j];
                T[(size_t)i * n + j] = s;
            }
        double f = 0.0;
        for (double x : T) f += x * x;
        f = std::sqrt(f);
        if (!(f > 0.0) || !std::isfinite(f)) return f > 0.0 ? INFINITY : 0.0;
        for (size_t i = 0; i < T.size(); ++i) M[i] = T[i] / f;
        logs = 2.0 * logs + std::log(f);
    }
    return std::exp(logs / 1024.0);
}
// κ of one model (A, B, P, Q, V0 row-major); +∞ if a covariance is not positive definite (the engine's own creation reports that by name)
inline double kappa(int d, int dy, const double* A, const double* B, const double* P, const double* Q, const double* V0, bool prior_through_transition) {
    std::vector<double> Qi, Pi, V0i, obs((size_t)d * d, 0.0), t((size_t)dy * d);
    if (!spd_inverse(Q, dy, Qi) || !spd_inverse(P, d, Pi)) return INFINITY;
    for (int a = 0; a < dy; ++a)       // t = Q⁻¹ B
        for (int j = 0; j < d; ++j) {
            double s = 0.0;
            for (int b = 0; b < dy; ++b) s += Qi[(size_t)a * dy + b] * B[(size_t)b * d + j];
            t[(size_t)a * d + j] = s;
        }
    for (int i = 0; i < d; ++i)        // obs = Bᵀ Q⁻¹ B
        for (int j = 0; j < d; ++j) {
            double s = 0.0;
            for (int a = 0; a < dy; ++a) s += B[(size_t)a * d + i] * t[(size_t)a * d + j];
            obs[(size_t)i * d + j] = s;
        }
    std::vector<double> V0p(V0, V0 + (size_t)d * d);
    if (prior_through_transition) {    // the first observed state is A x₀ + noise: A V0 Aᵀ + P
        std::vector<double> av((size_t)d * d);
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) {
                double s = 0.0;
                for (int k = 0; k < d; ++k) s += A[(size_t)i * d + k] * V0[(size_t)k * d + j];
                av[(size_t)i * d + j] = s;
            }
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) {
                double s = P[(size_t)i * d + j];
                for (int k = 0; k < d; ++k) s += av[(size_t)i * d + k] * A[(size_t)j * d + k];
                V0p[(size_t)i * d + j] = s;
            }
    }
    if (!spd_inverse(V0p.data(), d, V0i)) return INFINITY;
    std::vector<double> first(obs), drift(obs);
    const double rho = spectral_radius(A, d);
    const double forget = std::fmax(1.0 - rho * rho, 1.0e-6);
    for (size_t i = 0; i < (size_t)d * d; ++i) {
        first[i] += V0i[i];
        drift[i] += forget * Pi[i];
    }
    return std::fmax(cond_spd(first, d), cond_spd(drift, d));
}

}  // namespace envelope
}  // namespace rxhip
