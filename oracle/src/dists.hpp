// ORACLE — TEST INFRASTRUCTURE ONLY (see rng.hpp).
//
// dists.hpp — CPU restatement of the distributions on the SMC/MH path:
//   modppl/src/modeling/dists/distribution.rs:10-18  trait Distribution { logpdf, random }
//   modppl/src/modeling/dists/normal.rs:13-27        Normal
//   modppl/src/modeling/dists/uniform.rs:21-33       UniformContinuous
//   modppl/src/modeling/dists/bernoulli.rs:11-19     Bernoulli
//   modppl/src/modeling/dists/categorical.rs:12-32   Categorical
//   modppl/src/modeling/dists/mvnormal.rs:14-38      MvNormal
//   modppl/tests/pointed_model/types_2d.rs:14-32     Uniform2D (test-side distribution)
//
// Two arithmetic modes:
//   literal   : libm exp/log — the reference's own operations (Rust f64::ln/exp call libm).
//   canonical : mp_exp/mp_log from modppl_amd/csrc/mp_math.h — the single definition the GPU
//               evaluates too, so that samples and weights can be compared bit for bit.
#pragma once
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../modppl_amd/csrc/mp_math.h"
#include "rng.hpp"

namespace oracle {

struct Panic : std::runtime_error {  // every reference `panic!`/`assert!`/`unwrap` lands here
    using std::runtime_error::runtime_error;
};

inline bool& canonical_mode() {
    static bool m = false;
    return m;
}
inline double o_ln(double x) { return canonical_mode() ? mp_log(x) : std::log(x); }
inline double o_exp(double x) { return canonical_mode() ? mp_exp(x) : std::exp(x); }
inline double o_sqrt(double x) { return std::sqrt(x); }

// ---- normal.rs:13-27 ------------------------------------------------------------------
struct NormalParams { double mu, std; };
struct Normal {
    double logpdf(const double& x, NormalParams p) const {
        const double z = (x - p.mu) / p.std;
        // z.abs().powf(2.) == z*z (pow(|z|, 2) is exact-squared in libm and LLVM folds it)
        const double az = std::fabs(z);
        return -(az * az + o_ln(2. * M_PI)) / 2. - o_ln(p.std);
    }
    double random(Rng& rng, NormalParams p) const {
        for (;;) {  // the reference recurses on rejection (normal.rs:22)
            const double u = rng.u01() * 2. - 1.;
            const double v = rng.u01() * 2. - 1.;
            const double r = u * u + v * v;
            if (r == 0. || r > 1.) continue;
            const double c = o_sqrt(-2. * o_ln(r) / r);
            return u * c * p.std + p.mu;
        }
    }
};
static const Normal normal{};

// ---- uniform.rs:8-33 ------------------------------------------------------------------
struct UniformParams { double a, b; };
inline void check_bounds(double a, double b) {
    if (a >= b) throw Panic("a >= b in [a, b]; b > a is required.");
}
struct UniformContinuous {
    double logpdf(const double& x, UniformParams p) const {
        check_bounds(p.a, p.b);
        return (p.a <= x && x <= p.b) ? -o_ln(p.b - p.a) : -INFINITY;
    }
    double random(Rng& rng, UniformParams p) const {
        check_bounds(p.a, p.b);
        return rng.u01() * (p.b - p.a) + p.a;
    }
};
static const UniformContinuous uniform{};

// ---- bernoulli.rs:11-19 ---------------------------------------------------------------
struct Bernoulli {
    double logpdf(const bool& a, double p) const { return o_ln(a ? p : 1. - p); }
    bool random(Rng& rng, double p) const { return p > rng.u01(); }
};
static const Bernoulli bernoulli{};

// ---- categorical.rs:12-32 -------------------------------------------------------------
struct Categorical {
    static void check_sum(const std::vector<double>& probs) {
        double s = 0.;
        for (double p : probs) s += p;
        if (!(std::fabs(s - 1.0) <= 1e-8)) throw Panic("categorical: probs do not sum to 1 (eps 1e-8)");
    }
    double logpdf(const int64_t& x, const std::vector<double>& probs) const {
        check_sum(probs);
        return (x < (int64_t)probs.size()) ? o_ln(probs[(size_t)x]) : -INFINITY;
    }
    int64_t random(Rng& rng, const std::vector<double>& probs) const {
        check_sum(probs);
        const double u = rng.u01();
        return scan(u, probs);
    }
    // categorical.rs:24-31 verbatim semantics, including the u == 0 -> -1 quirk and the
    // out-of-bounds read when the running sum never reaches u (a Rust index panic).
    static int64_t scan(double u, const std::vector<double>& probs) {
        double t = 0.;
        int64_t x = 0;
        while (t < u) {
            if ((size_t)x >= probs.size()) throw Panic("categorical: index out of bounds");
            t += probs[(size_t)x];
            x += 1;
        }
        return x - 1;
    }
};
static const Categorical categorical{};

// ---- mvnormal.rs:14-38 ----------------------------------------------------------------
// Row-major dense k x k helper standing in for nalgebra::DMatrix (nalgebra 0.32.2, an
// un-vendored dependency: determinant / try_inverse / cholesky are restated from their
// published definitions; the reference's mvnormal KATs pin them to 1.2e-7).
struct Mat {
    int n = 0;
    std::vector<double> a;  // row-major
    Mat() {}
    Mat(int n_, std::vector<double> v) : n(n_), a(std::move(v)) {}
    double& operator()(int i, int j) { return a[(size_t)i * n + j]; }
    double operator()(int i, int j) const { return a[(size_t)i * n + j]; }
};
inline double determinant(Mat m) {  // LU with partial pivoting
    const int n = m.n;
    double det = 1.;
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r)
            if (std::fabs(m(r, c)) > std::fabs(m(piv, c))) piv = r;
        if (m(piv, c) == 0.) return 0.;
        if (piv != c) {
            for (int j = 0; j < n; ++j) std::swap(m(piv, j), m(c, j));
            det = -det;
        }
        det *= m(c, c);
        for (int r = c + 1; r < n; ++r) {
            const double f = m(r, c) / m(c, c);
            for (int j = c; j < n; ++j) m(r, j) -= f * m(c, j);
        }
    }
    return det;
}
inline bool try_inverse(Mat m, Mat& inv) {  // Gauss–Jordan with partial pivoting
    const int n = m.n;
    inv = Mat(n, std::vector<double>((size_t)n * n, 0.));
    for (int i = 0; i < n; ++i) inv(i, i) = 1.;
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r)
            if (std::fabs(m(r, c)) > std::fabs(m(piv, c))) piv = r;
        if (m(piv, c) == 0.) return false;
        if (piv != c)
            for (int j = 0; j < n; ++j) { std::swap(m(piv, j), m(c, j)); std::swap(inv(piv, j), inv(c, j)); }
        const double d = m(c, c);
        for (int j = 0; j < n; ++j) { m(c, j) /= d; inv(c, j) /= d; }
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            const double f = m(r, c);
            if (f == 0.) continue;
            for (int j = 0; j < n; ++j) { m(r, j) -= f * m(c, j); inv(r, j) -= f * inv(c, j); }
        }
    }
    return true;
}
inline bool cholesky_l(const Mat& m, Mat& L) {
    const int n = m.n;
    L = Mat(n, std::vector<double>((size_t)n * n, 0.));
    for (int j = 0; j < n; ++j) {
        double d = m(j, j);
        for (int k = 0; k < j; ++k) d -= L(j, k) * L(j, k);
        if (!(d > 0.)) return false;
        L(j, j) = o_sqrt(d);
        for (int i = j + 1; i < n; ++i) {
            double s = m(i, j);
            for (int k = 0; k < j; ++k) s -= L(i, k) * L(j, k);
            L(i, j) = s / L(j, j);
        }
    }
    return true;
}
// mvnormal.rs:30-33 — `transform` of a covariance without a Cholesky factor: eigenvectors * diag(sqrt(eigenvalues)).
// nalgebra's symmetric_eigen (un-vendored) fixes neither the order nor the sign of the eigenvectors, which
// `transform * z` depends on; the build's definition (cyclic Jacobi, upper-triangle sweep order) is restated here.
inline Mat symmetric_eigen_transform(Mat a) {
    const int n = a.n;
    Mat v(n, std::vector<double>((size_t)n * n, 0.));
    for (int i = 0; i < n; ++i) v(i, i) = 1.;
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0.;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) off += a(p, q) * a(p, q);
        if (off == 0.) break;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = a(p, q);
                if (apq == 0.) continue;
                const double theta = (a(q, q) - a(p, p)) / (2. * apq);
                const double t = (theta >= 0. ? 1. : -1.) / (std::fabs(theta) + std::sqrt(theta * theta + 1.));
                const double c = 1. / std::sqrt(t * t + 1.), s = t * c;
                for (int k = 0; k < n; ++k) { const double x = a(k, p), y = a(k, q); a(k, p) = c * x - s * y; a(k, q) = s * x + c * y; }
                for (int k = 0; k < n; ++k) { const double x = a(p, k), y = a(q, k); a(p, k) = c * x - s * y; a(q, k) = s * x + c * y; }
                for (int k = 0; k < n; ++k) { const double x = v(k, p), y = v(k, q); v(k, p) = c * x - s * y; v(k, q) = s * x + c * y; }
            }
    }
    double lmax = 0.;
    for (int j = 0; j < n; ++j) lmax = std::fmax(lmax, std::fabs(a(j, j)));
    Mat T(n, std::vector<double>((size_t)n * n, 0.));
    for (int j = 0; j < n; ++j) {
        double lam = a(j, j);
        if (lam < 0. && lam >= -64. * 2.220446049250313e-16 * lmax) lam = 0.;   // round-off of a zero eigenvalue
        const double sq = std::sqrt(lam);
        for (int i = 0; i < n; ++i) T(i, j) = v(i, j) * sq;
    }
    return T;
}
// mvnormal.rs:26-34
inline Mat mvnormal_transform(const Mat& cov) {
    Mat L;
    if (cholesky_l(cov, L)) return L;
    return symmetric_eigen_transform(cov);
}
// sum_k a_k b_k: the reference's multiply-then-add, or (canonical mode) the k-ascending fma chain the matrix cores evaluate
inline double dense_dot(const double* a, size_t sa, const double* b, size_t sb, int k) {
    double acc = 0.;
    if (canonical_mode()) { for (int i = 0; i < k; ++i) acc = std::fma(a[i * sa], b[i * sb], acc); }
    else { for (int i = 0; i < k; ++i) acc += a[i * sa] * b[i * sb]; }
    return acc;
}

struct MvNormalParams { std::vector<double> mu; Mat cov; };
struct MvNormal {
    double logpdf(const std::vector<double>& x, const MvNormalParams& p) const {
        const int k = (int)p.mu.size();
        const double cov_det = determinant(p.cov);       // per call, as mvnormal.rs:17
        Mat cov_inv;
        if (!try_inverse(p.cov, cov_inv)) throw Panic("mvnormal: covariance not invertible");
        std::vector<double> c((size_t)k);
        for (int i = 0; i < k; ++i) c[(size_t)i] = x[(size_t)i] - p.mu[(size_t)i];
        double maha = 0.;  // (c^T * inv) * c
        for (int j = 0; j < k; ++j) {
            double r = 0.;
            for (int i = 0; i < k; ++i) r += c[(size_t)i] * cov_inv(i, j);
            maha += r * c[(size_t)j];
        }
        return -((double)k * o_ln(2. * M_PI) + o_ln(cov_det) + maha) / 2.;
    }
    // the same two functions for models that define their dense products through dense_dot (mp_lgssm_dense): full k x k
    // transform (Cholesky or eigen form), per-call determinant / inverse as in the reference
    double logpdf_dense(const std::vector<double>& x, const MvNormalParams& p) const {
        const int k = (int)p.mu.size();
        const double cov_det = determinant(p.cov);
        Mat cov_inv;
        if (!try_inverse(p.cov, cov_inv)) throw Panic("mvnormal: covariance not invertible");
        std::vector<double> c((size_t)k), r((size_t)k);
        for (int i = 0; i < k; ++i) c[(size_t)i] = x[(size_t)i] - p.mu[(size_t)i];
        for (int j = 0; j < k; ++j) r[(size_t)j] = dense_dot(c.data(), 1, &cov_inv.a[(size_t)j], (size_t)k, k);
        const double maha = dense_dot(r.data(), 1, c.data(), 1, k);
        return -((double)k * o_ln(2. * M_PI) + o_ln(cov_det) + maha) / 2.;
    }
    std::vector<double> random_dense(Rng& rng, const MvNormalParams& p) const {
        const int k = (int)p.mu.size();
        const Mat T = mvnormal_transform(p.cov);
        std::vector<double> z((size_t)k), out((size_t)k);
        for (int j = 0; j < k; ++j) z[(size_t)j] = normal.random(rng, {0., 1.});  // index order, one stream
        for (int i = 0; i < k; ++i) out[(size_t)i] = dense_dot(&T.a[(size_t)i * k], 1, z.data(), 1, k) + p.mu[(size_t)i];
        return out;
    }
    std::vector<double> random(Rng& rng, const MvNormalParams& p) const {
        const int k = (int)p.mu.size();
        Mat L;
        if (!cholesky_l(p.cov, L)) L = symmetric_eigen_transform(p.cov);   // mvnormal.rs:30-33
        std::vector<double> z((size_t)k), out((size_t)k);
        for (int j = 0; j < k; ++j) z[(size_t)j] = normal.random(rng, {0., 1.});  // index order
        for (int i = 0; i < k; ++i) {
            double s = 0.;
            for (int j = 0; j <= i; ++j) s += L(i, j) * z[(size_t)j];
            out[(size_t)i] = s + p.mu[(size_t)i];
        }
        return out;
    }
};
static const MvNormal mvnormal{};

// ---- tests/pointed_model/types_2d.rs:8-32 ---------------------------------------------
struct Bounds { double xmin, xmax, ymin, ymax; };
struct Uniform2D {
    double logpdf(const std::vector<double>& p, Bounds b) const {
        if (b.xmin <= p[0] && p[0] <= b.xmax && b.ymin <= p[1] && p[1] <= b.ymax)
            return -o_ln((b.xmax - b.xmin) * (b.ymax - b.ymin));
        return -INFINITY;
    }
    std::vector<double> random(Rng& rng, Bounds b) const {
        if (!(b.xmax > b.xmin) || !(b.ymax > b.ymin)) throw Panic("uniform_2d: bad bounds");
        const double x = rng.u01() * (b.xmax - b.xmin) + b.xmin;
        const double y = rng.u01() * (b.ymax - b.ymin) + b.ymin;
        return {x, y};
    }
};
static const Uniform2D uniform_2d{};

// ---- lib.rs:34-45 ---------------------------------------------------------------------
inline double logsumexp(const std::vector<double>& xs) {
    double max = -INFINITY;
    for (double x : xs) max = std::fmax(max, x);  // f64::max
    if (max == -INFINITY) return -INFINITY;
    double sum_exp = 0.;
    for (double x : xs) sum_exp += o_exp(x - max);
    return max + o_ln(sum_exp);
}

}  // namespace oracle
