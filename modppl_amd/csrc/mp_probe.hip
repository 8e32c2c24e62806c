// mp_probe.hip — device self-test probes (include/modppl_hip_probe.h).
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/modppl_hip.h"
#include "../../include/modppl_hip_probe.h"
#include <vector>

#include "mp_dists.h"
#include "mp_linalg.h"

__global__ void k_probe_math(int op, const double* a, const double* b, const double* c, long long n, double* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double r = 0.;
    switch (op) {
    case MP_PROBE_EXP: r = mp_exp(a[i]); break;
    case MP_PROBE_LOG: r = mp_log(a[i]); break;
    case MP_PROBE_SQRT: r = mp_sqrt(a[i]); break;
    case MP_PROBE_DIV: r = a[i] / b[i]; break;
    case MP_PROBE_NORMAL_LOGPDF: r = mp_normal_logpdf(a[i], b[i], c[i]); break;
    case MP_PROBE_NORMAL_LOGPDF_H: r = mp_normal_logpdf_h(a[i], b[i], c[i], mp_log(c[i]), mp_rcp_hoist(c[i])); break;
    case MP_PROBE_DIV_HOISTED: r = mp_rcp_hoistable(b[i]) ? mp_div_hoisted(a[i], b[i], 1.0 / b[i]) : a[i] / b[i]; break;
    }
    out[i] = r;
}
__global__ void k_probe_normal(uint32_t k0, uint32_t k1, uint32_t slot0, uint32_t step, uint32_t domain, uint32_t site, double mu, double sd,
                               long long n, double* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    mp_stream s;
    s.k0 = k0; s.k1 = k1; s.slot = slot0 + (uint32_t)i; s.step = step;
    mp_site st(s, domain, site);
    out[i] = mp_normal_sample(st, mu, sd);
}
__global__ void k_probe_u01(uint32_t k0, uint32_t k1, uint32_t slot0, uint32_t step, uint32_t domain, uint32_t site, uint32_t attempt,
                            long long n, double* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const mp_u64x2 b = mp_philox4x32_10(slot0 + (uint32_t)i, step, (domain << 16) | site, attempt, k0, k1);
    out[2 * i] = mp_u01(b.a);
    out[2 * i + 1] = mp_u01(b.b);
}

// mvnormal (mvnormal.rs:14-38) of dimension k <= MP_PROBE_MAX_K with the covariance constants hoisted by the host (mp_linalg.h;
// the reference takes any k; model sites are compiled for k <= 16, this general-k form serves the distribution on its own):
#define MP_PROBE_MAX_K 64
// out[i] = logpdf(x_i; mu, cov) in the reference's operation order (chain = 0) or as the matrix cores' fma chain (chain = 1)
__global__ void k_probe_mvnormal_logpdf(int k, int chain, const double* x, const double* mu, const double* cov_inv, double ln_det, long long n, double* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double c[MP_PROBE_MAX_K];
    for (int a = 0; a < k; ++a) c[a] = x[i * k + a] - mu[a];
    double maha = 0.;
    for (int j = 0; j < k; ++j) {
        double r = 0.;
        for (int a = 0; a < k; ++a) r = chain ? fma(c[a], cov_inv[a * k + j], r) : r + c[a] * cov_inv[a * k + j];
        maha = chain ? fma(r, c[j], maha) : maha + r * c[j];
    }
    out[i] = -((double)k * MP_LN_2PI_CANON + ln_det + maha) / 2.;
}
// out[i][0..k) = transform * z + mu, z_j ~ normal(0, 1) in index order from the stream of (slot0 + i, step, domain, site)
__global__ void k_probe_mvnormal_sample(int k, int chain, uint32_t k0, uint32_t k1, uint32_t slot0, uint32_t step, uint32_t domain, uint32_t site,
                                        const double* mu, const double* transform, long long n, double* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    mp_stream s;
    s.k0 = k0; s.k1 = k1; s.slot = slot0 + (uint32_t)i; s.step = step;
    mp_site st(s, domain, site);
    double z[MP_PROBE_MAX_K];
    for (int j = 0; j < k; ++j) z[j] = mp_normal_sample(st, 0., 1.);
    for (int a = 0; a < k; ++a) {
        double acc = 0.;
        for (int j = 0; j < k; ++j) acc = chain ? fma(transform[a * k + j], z[j], acc) : acc + transform[a * k + j] * z[j];
        out[i * k + a] = acc + mu[a];
    }
}

// One v_mfma_f64_16x16x4_f64: D[16][16] = A[16][4] * B[4][16] + C[16][16] (row-major host arrays), by one wave.
// Operand lanes (cdna_hip_programming.md §3): A lane l = A[l & 15][l >> 4]; B lane l = B[l >> 4][l & 15];
// C/D lane l, register r = [(l >> 4) + 4 r][l & 15].
typedef double mp_f64x4 __attribute__((ext_vector_type(4)));
__global__ void k_probe_mfma_f64(const double* A, const double* B, const double* C, double* D) {
    const int l = threadIdx.x;
    const double a = A[(l & 15) * 4 + (l >> 4)];
    const double b = B[(l >> 4) * 16 + (l & 15)];
    mp_f64x4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[((l >> 4) + 4 * r) * 16 + (l & 15)];
    const mp_f64x4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = d[r];
}

#define PCK(call)                                                              \
    do {                                                                       \
        hipError_t e_ = (call);                                                \
        if (e_ != hipSuccess) { rc = MP_ERR_HIP; goto done; }                  \
    } while (0)

extern "C" {

int32_t mp_probe_math(int32_t op, const double* a, const double* b, const double* c, int64_t n, double* out, int32_t device) {
    int32_t rc = MP_OK;
    double *da = nullptr, *db = nullptr, *dc = nullptr, *dout = nullptr;
    const size_t bytes = sizeof(double) * (size_t)n;
    PCK(hipSetDevice(device));
    PCK(hipMalloc(&da, bytes)); PCK(hipMalloc(&db, bytes)); PCK(hipMalloc(&dc, bytes)); PCK(hipMalloc(&dout, bytes));
    PCK(hipMemcpy(da, a, bytes, hipMemcpyHostToDevice));
    if (b) PCK(hipMemcpy(db, b, bytes, hipMemcpyHostToDevice)); else PCK(hipMemset(db, 0, bytes));
    if (c) PCK(hipMemcpy(dc, c, bytes, hipMemcpyHostToDevice)); else PCK(hipMemset(dc, 0, bytes));
    hipLaunchKernelGGL(k_probe_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, op, da, db, dc, (long long)n, dout);
    PCK(hipGetLastError());
    PCK(hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost));
done:
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dc); (void)hipFree(dout);
    return rc;
}

int32_t mp_probe_normal_sample(uint64_t seed, uint32_t slot0, uint32_t step, uint32_t domain, uint32_t site, double mu, double sd,
                               int64_t n, double* out, int32_t device) {
    int32_t rc = MP_OK;
    double* dout = nullptr;
    const size_t bytes = sizeof(double) * (size_t)n;
    PCK(hipSetDevice(device));
    PCK(hipMalloc(&dout, bytes));
    hipLaunchKernelGGL(k_probe_normal, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (uint32_t)seed, (uint32_t)(seed >> 32), slot0, step,
                       domain, site, mu, sd, (long long)n, dout);
    PCK(hipGetLastError());
    PCK(hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost));
done:
    (void)hipFree(dout);
    return rc;
}

int32_t mp_probe_u01(uint64_t seed, uint32_t slot0, uint32_t step, uint32_t domain, uint32_t site, uint32_t attempt, int64_t n,
                     double* out, int32_t device) {
    int32_t rc = MP_OK;
    double* dout = nullptr;
    const size_t bytes = sizeof(double) * 2 * (size_t)n;
    PCK(hipSetDevice(device));
    PCK(hipMalloc(&dout, bytes));
    hipLaunchKernelGGL(k_probe_u01, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (uint32_t)seed, (uint32_t)(seed >> 32), slot0, step,
                       domain, site, attempt, (long long)n, dout);
    PCK(hipGetLastError());
    PCK(hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost));
done:
    (void)hipFree(dout);
    return rc;
}

int32_t mp_probe_mvnormal(int32_t k, int32_t chain, const double* x, const double* mu, const double* cov, int64_t n, double* logpdf_out,
                          uint64_t seed, uint32_t slot0, uint32_t step, uint32_t domain, uint32_t site, double* sample_out, int32_t device) {
    if (k < 1 || k > MP_PROBE_MAX_K || !mu || !cov || n < 1) return MP_ERR_INVALID_ARG;
    int32_t rc = MP_OK;
    const std::vector<double> c(cov, cov + (size_t)k * k);
    std::vector<double> inv, T;
    const double det = mp_host_det(c, k);
    double *dx = nullptr, *dmu = nullptr, *dm = nullptr, *dout = nullptr;
    PCK(hipSetDevice(device));
    PCK(hipMalloc(&dmu, sizeof(double) * k)); PCK(hipMalloc(&dm, sizeof(double) * k * k)); PCK(hipMalloc(&dout, sizeof(double) * (size_t)n * k));
    PCK(hipMemcpy(dmu, mu, sizeof(double) * k, hipMemcpyHostToDevice));
    if (logpdf_out) {
        if (!x || !mp_host_inverse(c, k, inv)) { rc = MP_ERR_INVALID_ARG; goto done; }   // try_inverse().unwrap() panics (mvnormal.rs:18)
        PCK(hipMalloc(&dx, sizeof(double) * (size_t)n * k));
        PCK(hipMemcpy(dx, x, sizeof(double) * (size_t)n * k, hipMemcpyHostToDevice));
        PCK(hipMemcpy(dm, inv.data(), sizeof(double) * k * k, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_probe_mvnormal_logpdf, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, k, chain, dx, dmu, dm, mp_log(det), (long long)n, dout);
        PCK(hipGetLastError());
        PCK(hipMemcpy(logpdf_out, dout, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
    }
    if (sample_out) {
        mp_host_mvnormal_transform(c, k, T);   // Cholesky factor, or the eigen form when there is none (mvnormal.rs:26-34)
        PCK(hipMemcpy(dm, T.data(), sizeof(double) * k * k, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_probe_mvnormal_sample, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, k, chain, (uint32_t)seed, (uint32_t)(seed >> 32), slot0, step,
                           domain, site, dmu, dm, (long long)n, dout);
        PCK(hipGetLastError());
        PCK(hipMemcpy(sample_out, dout, sizeof(double) * (size_t)n * k, hipMemcpyDeviceToHost));
    }
done:
    (void)hipFree(dx); (void)hipFree(dmu); (void)hipFree(dm); (void)hipFree(dout);
    return rc;
}

int32_t mp_probe_mfma_f64(const double* A16x4, const double* B4x16, const double* C16x16, double* D16x16, int32_t device) {
    int32_t rc = MP_OK;
    double *da = nullptr, *db = nullptr, *dc = nullptr, *dd = nullptr;
    PCK(hipSetDevice(device));
    PCK(hipMalloc(&da, sizeof(double) * 64)); PCK(hipMalloc(&db, sizeof(double) * 64)); PCK(hipMalloc(&dc, sizeof(double) * 256)); PCK(hipMalloc(&dd, sizeof(double) * 256));
    PCK(hipMemcpy(da, A16x4, sizeof(double) * 64, hipMemcpyHostToDevice));
    PCK(hipMemcpy(db, B4x16, sizeof(double) * 64, hipMemcpyHostToDevice));
    PCK(hipMemcpy(dc, C16x16, sizeof(double) * 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_probe_mfma_f64, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
    PCK(hipGetLastError());
    PCK(hipMemcpy(D16x16, dd, sizeof(double) * 256, hipMemcpyDeviceToHost));
done:
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dc); (void)hipFree(dd);
    return rc;
}

}  // extern "C"
