/* modppl_hip_probe.h — device self-test probes of libmodppl_hip.so.
 *
 * Not part of the drop-in boundary: these evaluate the path's scalar building blocks on the GPU
 * over caller-provided arrays so that tests can compare them bit for bit with the CPU checker
 * (deterministic exp/log of mp_math.h, IEEE sqrt/div, the Philox stream, the polar normal
 * sampler of modppl/src/modeling/dists/normal.rs:19-27).
 */
#ifndef MODPPL_HIP_PROBE_H
#define MODPPL_HIP_PROBE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum mp_probe_op { MP_PROBE_EXP = 0, MP_PROBE_LOG = 1, MP_PROBE_SQRT = 2, MP_PROBE_DIV = 3, MP_PROBE_NORMAL_LOGPDF = 4,
                   MP_PROBE_DIV_HOISTED = 5 /* mp_div_hoisted(a, b, RN(1 / b)): the division by a hoisted constant, mp_math.h */,
                   MP_PROBE_NORMAL_LOGPDF_H = 6 /* mp_normal_logpdf_h(a, b, c, mp_log(c), mp_rcp_hoist(c)): no division inside */ };
/* out[i] = op(a[i], b[i], c[i]) on the device; b, c may be NULL for unary ops (host pointers). */
int32_t mp_probe_math(int32_t op, const double* a, const double* b, const double* c, int64_t n, double* out, int32_t device);
/* out[i] = normal.random with Philox (seed, slot = slot0 + i, step, domain, site) and params (mu, sd). */
int32_t mp_probe_normal_sample(uint64_t seed, uint32_t slot0, uint32_t step, uint32_t domain, uint32_t site, double mu, double sd,
                               int64_t n, double* out, int32_t device);
/* out[2*i], out[2*i+1] = the two u01 of Philox block (slot0 + i, step, domain, site, attempt). */
int32_t mp_probe_u01(uint64_t seed, uint32_t slot0, uint32_t step, uint32_t domain, uint32_t site, uint32_t attempt, int64_t n,
                     double* out, int32_t device);

/* mvnormal of dimension k <= 64 (modppl/src/modeling/dists/mvnormal.rs:14-38; model sites are compiled for k <= 16), determinant / inverse / transform hoisted to the
 * host once: logpdf_out[i] = logpdf(x[i][0..k); mu, cov) when logpdf_out != NULL; sample_out[i][0..k) = random with the Philox
 * stream (seed, slot0 + i, step, domain, site) when sample_out != NULL (covariances without a Cholesky factor take the
 * eigen transform, :30-33).  chain = 0: the reference's multiply-then-add order; 1: the matrix cores' fma chain. */
int32_t mp_probe_mvnormal(int32_t k, int32_t chain, const double* x, const double* mu, const double* cov, int64_t n, double* logpdf_out,
                          uint64_t seed, uint32_t slot0, uint32_t step, uint32_t domain, uint32_t site, double* sample_out, int32_t device);
/* One v_mfma_f64_16x16x4_f64 on one wave: D = A * B + C (row-major A[16][4], B[4][16], C, D[16][16]): pins the matrix
 * core's accumulation order, which the dense-transition kernels and the CPU checker's canonical matvec restate. */
int32_t mp_probe_mfma_f64(const double* A16x4, const double* B4x16, const double* C16x16, double* D16x16, int32_t device);

#ifdef __cplusplus
}
#endif
#endif
