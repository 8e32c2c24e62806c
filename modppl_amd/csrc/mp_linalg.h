// mp_linalg.h — host-side dense helpers for model constants (any k; model sites use k <= 16): determinant and inverse of a
// covariance, evaluated ONCE per model instead of once per logpdf call as modppl does
// (modppl/src/modeling/dists/mvnormal.rs:17-18 calls nalgebra's determinant() and try_inverse() inside
// every logpdf).  LU / Gauss-Jordan with partial pivoting, row-major.
#pragma once
#include <cmath>
#include <utility>
#include <vector>

inline double mp_host_det(std::vector<double> m, int n) {
    double det = 1.;
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r)
            if (std::fabs(m[r * n + c]) > std::fabs(m[piv * n + c])) piv = r;
        if (m[piv * n + c] == 0.) return 0.;
        if (piv != c) {
            for (int j = 0; j < n; ++j) std::swap(m[piv * n + j], m[c * n + j]);
            det = -det;
        }
        det *= m[c * n + c];
        for (int r = c + 1; r < n; ++r) {
            const double f = m[r * n + c] / m[c * n + c];
            for (int j = c; j < n; ++j) m[r * n + j] -= f * m[c * n + j];
        }
    }
    return det;
}
inline bool mp_host_inverse(std::vector<double> m, int n, std::vector<double>& inv) {
    inv.assign((size_t)n * n, 0.);
    for (int i = 0; i < n; ++i) inv[i * n + i] = 1.;
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r)
            if (std::fabs(m[r * n + c]) > std::fabs(m[piv * n + c])) piv = r;
        if (m[piv * n + c] == 0.) return false;
        if (piv != c)
            for (int j = 0; j < n; ++j) {
                std::swap(m[piv * n + j], m[c * n + j]);
                std::swap(inv[piv * n + j], inv[c * n + j]);
            }
        const double d = m[c * n + c];
        for (int j = 0; j < n; ++j) {
            m[c * n + j] /= d;
            inv[c * n + j] /= d;
        }
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            const double f = m[r * n + c];
            if (f == 0.) continue;
            for (int j = 0; j < n; ++j) {
                m[r * n + j] -= f * m[c * n + j];
                inv[r * n + j] -= f * inv[c * n + j];
            }
        }
    }
    return true;
}

// lower Cholesky factor (mvnormal.rs:27-29 `cov.cholesky().l()`): false if the matrix is not positive definite (the
// reference then falls back to an eigendecomposition, which is not restated)
inline bool mp_host_cholesky(const std::vector<double>& m, int n, std::vector<double>& L) {
    L.assign((size_t)n * n, 0.);
    for (int j = 0; j < n; ++j) {
        double d = m[j * n + j];
        for (int k = 0; k < j; ++k) d -= L[j * n + k] * L[j * n + k];
        if (!(d > 0.)) return false;
        L[j * n + j] = std::sqrt(d);
        for (int i = j + 1; i < n; ++i) {
            double s = m[i * n + j];
            for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = s / L[j * n + j];
        }
    }
    return true;
}

// mvnormal.rs:30-33: the fallback `transform` of a covariance without a Cholesky factor (positive semi-definite, singular):
// eigenvectors * diag(sqrt(eigenvalues)).  nalgebra's symmetric_eigen is an un-vendored dependency; its result is defined only
// up to the order and sign of the eigenvectors, which `transform * z` depends on, so the build fixes one: cyclic Jacobi
// rotations (upper-triangle sweep order, at most 64 sweeps), eigenpairs in the order the diagonal ends up in.  A negative
// eigenvalue gives NaN columns, as `.sqrt()` does in the reference — except round-off: an eigenvalue in [-64 eps lambda_max, 0)
// (what a zero eigenvalue of a singular covariance comes out as) is taken as zero, where the reference's result depends on
// the last bit of nalgebra's iteration.
inline void mp_host_sym_eigen_transform(std::vector<double> a, int n, std::vector<double>& T) {
    std::vector<double> v((size_t)n * n, 0.);
    for (int i = 0; i < n; ++i) v[i * n + i] = 1.;
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0.;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) off += a[p * n + q] * a[p * n + q];
        if (off == 0.) break;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = a[p * n + q];
                if (apq == 0.) continue;
                const double theta = (a[q * n + q] - a[p * n + p]) / (2. * apq);
                const double t = (theta >= 0. ? 1. : -1.) / (std::fabs(theta) + std::sqrt(theta * theta + 1.));
                const double c = 1. / std::sqrt(t * t + 1.), s_ = t * c;
                for (int k = 0; k < n; ++k) {   // A <- J^T A J
                    const double akp = a[k * n + p], akq = a[k * n + q];
                    a[k * n + p] = c * akp - s_ * akq;
                    a[k * n + q] = s_ * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = a[p * n + k], aqk = a[q * n + k];
                    a[p * n + k] = c * apk - s_ * aqk;
                    a[q * n + k] = s_ * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = v[k * n + p], vkq = v[k * n + q];
                    v[k * n + p] = c * vkp - s_ * vkq;
                    v[k * n + q] = s_ * vkp + c * vkq;
                }
            }
    }
    double lmax = 0.;
    for (int j = 0; j < n; ++j) lmax = std::fmax(lmax, std::fabs(a[j * n + j]));
    T.assign((size_t)n * n, 0.);
    for (int j = 0; j < n; ++j) {
        double lam = a[j * n + j];
        if (lam < 0. && lam >= -64. * 2.220446049250313e-16 * lmax) lam = 0.;
        const double sq = std::sqrt(lam);
        for (int i = 0; i < n; ++i) T[i * n + j] = v[i * n + j] * sq;
    }
}
// mvnormal.rs:26-34: Cholesky factor if there is one, the eigen form otherwise
inline void mp_host_mvnormal_transform(const std::vector<double>& cov, int n, std::vector<double>& T) {
    if (!mp_host_cholesky(cov, n, T)) mp_host_sym_eigen_transform(cov, n, T);
}
