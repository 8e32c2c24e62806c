// mp_linalg.h — host-side dense helpers for model constants (k <= 16): determinant and inverse of a
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
