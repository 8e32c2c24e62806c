// mp_probe.hip — device self-test probes (include/modppl_hip_probe.h).
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/modppl_hip.h"
#include "../../include/modppl_hip_probe.h"
#include "mp_dists.h"

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

}  // extern "C"
