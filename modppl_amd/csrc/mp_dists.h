// mp_dists.h — the distributions of the hot path as inlinable host/device functions, same
// operations in the same order as modppl's CPU code:
//   normal      modppl/src/modeling/dists/normal.rs:13-27   (logpdf; Marsaglia polar sampler)
//   uniform     modppl/src/modeling/dists/uniform.rs:21-33
//   bernoulli   modppl/src/modeling/dists/bernoulli.rs:11-19
//   categorical modppl/src/modeling/dists/categorical.rs:22-32 (small fixed-size tables)
// Transcendentals are mp_exp/mp_log (mp_math.h); compile with -ffp-contract=off.
#pragma once
#include "mp_math.h"
#include "mp_philox.h"

// Sequential uniform stream of ONE site: the n-th uniform is half (n & 1) of Philox block n >> 1
// (counter layout: mp_philox.h).  `blk` counts whole blocks for samplers that use both halves.
struct mp_site {
    mp_stream s;
    uint32_t domain, site;
    uint32_t blk;
    MP_PHD mp_site(const mp_stream& s_, uint32_t domain_, uint32_t site_) : s(s_), domain(domain_), site(site_), blk(0) {}
    MP_PHD mp_u64x2 next_block() { return s.draw(domain, site, blk++); }
};

// mp_log(2*pi) evaluated once (tests/test_math.py checks the bits against mp_log itself).
#define MP_LN_2PI_CANON (mp_u2f(0x3FFD67F1C864BEB4ull))

// normal.rs:13-17; `ln_sd` = mp_log(sd), hoisted by callers whose sd is a model constant.
MP_HD double mp_normal_logpdf_ln(double x, double mu, double sd, double ln_sd) {
    const double z = (x - mu) / sd;
    const double az = fabs(z);
    return -(az * az + MP_LN_2PI_CANON) / 2. - ln_sd;
}
MP_HD double mp_normal_logpdf(double x, double mu, double sd) { return mp_normal_logpdf_ln(x, mu, sd, mp_log(sd)); }
// ... with the reciprocal of sd hoisted as well (mp_rcp_hoist; 0 = not hoisted): same bits, no division (mp_div_hoisted)
MP_HD double mp_normal_logpdf_h(double x, double mu, double sd, double ln_sd, double rcp_sd) {
    // (the branch-free core: outside its exact range — |x - mu| < 2^-960, or a quotient that overflows — z * z is 0 or infinite
    // whichever of the two quotients it is computed from, and an infinite x - mu stays infinite: the same log-density bits for every
    // input, without the division's code in every site of a model.  mp_probe op MP_PROBE_NORMAL_LOGPDF_H holds it to that.)
    const double z = rcp_sd != 0. ? mp_div_hoisted_core(x - mu, sd, rcp_sd) : (x - mu) / sd;
    const double az = fabs(z);
    return -(az * az + MP_LN_2PI_CANON) / 2. - ln_sd;
}

// The accepted pair (u, r = u*u + v*v) of the polar method does not depend on (mu, sd), so the
// rejection loop can run ahead of the model (mp_pf.hip, k_propagate) and the model consumes it here.
// normal.rs:25-26: c = sqrt(-2 ln r / r); u*c*std + mu.
// The standard deviate z = u*c is the parameter-free part: `u * c * std + mu` evaluates as ((u*c)*std) + mu, so a kernel may
// produce z anywhere (another lane, another launch) and the model finishes with z*sd + mu — same operations, same order.
MP_HD double mp_std_normal_from_pair(double u, double r) {
    const double c = mp_sqrt(-2. * mp_log(r) / r);
    return u * c;
}
MP_HD double mp_normal_from_pair(double u, double r, double mu, double sd) { return mp_std_normal_from_pair(u, r) * sd + mu; }
// one attempt of the polar method on a Philox block: (u, r) and whether normal.rs:22 accepts it
MP_HD bool mp_polar_attempt(const mp_u64x2& b, double* u_out, double* r_out) {
    const double u = mp_u01(b.a) * 2. - 1.;
    const double v = mp_u01(b.b) * 2. - 1.;
    const double r = u * u + v * v;
    *u_out = u;
    *r_out = r;
    return !(r == 0. || r > 1.);
}

// normal.rs:19-27: u,v = 2*u01-1; r = u*u+v*v; reject r == 0 or r > 1 (the reference recurses);
// c = sqrt(-2 ln r / r); return u*c*std + mu.  Only u*c is used, as in the reference.
MP_HD double mp_normal_sample(mp_site& st, double mu, double sd) {
    for (;;) {
        const mp_u64x2 b = st.next_block();
        const double u = mp_u01(b.a) * 2. - 1.;
        const double v = mp_u01(b.b) * 2. - 1.;
        const double r = u * u + v * v;
        if (r == 0. || r > 1.) continue;
        return mp_normal_from_pair(u, r, mu, sd);
    }
}

// uniform.rs:21-33 (bounds are model constants here; a >= b is rejected on the host)
MP_HD double mp_uniform_logpdf(double x, double a, double b) { return (a <= x && x <= b) ? -mp_log(b - a) : MP_NEG_INF; }
MP_HD double mp_uniform_sample(mp_site& st, double a, double b) {
    const mp_u64x2 blk = st.next_block();
    return mp_u01(blk.a) * (b - a) + a;
}

// bernoulli.rs:11-19
MP_HD double mp_bernoulli_logpdf(bool v, double p) { return mp_log(v ? p : 1. - p); }
MP_HD bool mp_bernoulli_sample(mp_site& st, double p) {
    const mp_u64x2 blk = st.next_block();
    return p > mp_u01(blk.a);
}

// categorical.rs:22-32 over a small table: t=0; x=0; while t<u {t+=p[x]; x+=1}; return x-1.
// u == 0 returns -1 in the reference (and then panics as a usize index); here it is clamped to 0.
// Running past the table (sum < u) is an index panic there; here it is clamped to n-1.
MP_HD int mp_categorical_scan(double u, const double* probs, int n) {
    double t = 0.;
    int x = 0;
    while (t < u && x < n) {
        t += probs[x];
        x += 1;
    }
    return x > 0 ? x - 1 : 0;
}

// mvnormal.rs:14-22 with the per-call determinant/inverse hoisted: cov_inv (row-major KxK) and
// ln_det = mp_log(det cov) are model constants.  Quadratic form in the reference's order:
// (c^T * cov_inv) first, then dotted with c.
template <int K>
MP_HD double mp_mvnormal_logpdf_pre(const double* x, const double* mu, const double* cov_inv, double ln_det) {
    double c[K];
#pragma unroll
    for (int i = 0; i < K; ++i) c[i] = x[i] - mu[i];
    double maha = 0.;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        double r = 0.;
#pragma unroll
        for (int i = 0; i < K; ++i) r += c[i] * cov_inv[i * K + j];
        maha += r * c[j];
    }
    return -((double)K * MP_LN_2PI_CANON + ln_det + maha) / 2.;
}

// ---- dense K x K forms in the MATRIX CORE's accumulation order --------------------------------------------------------
// v_mfma_f64_16x16x4_f64 computes D = fma(a_3, b_3, fma(a_2, b_2, fma(a_1, b_1, fma(a_0, b_0, C)))): a k-ascending fma chain
// from C (pinned on the device by tests/test_gpu_math.py::test_mfma_f64_accumulation_order).  Models whose transition is a
// dense matvec (mp_lgssm_dense) define their products as that chain, so the scalar form below, the MFMA kernel
// (k_propagate_dense16) and the CPU checker's canonical arithmetic give the same bits; the literal checker keeps
// nalgebra's multiply-then-add.
template <int K>
MP_HD double mp_dot_chain(const double* a, int stride_a, const double* b, int stride_b, double c0) {
    double acc = c0;
#pragma unroll
    for (int k = 0; k < K; ++k) acc = fma(a[k * stride_a], b[k * stride_b], acc);
    return acc;
}
// mvnormal.rs:14-22, covariance constants hoisted (cov_inv row-major, ln_det): r_j = sum_i c_i inv[i][j]; maha = sum_j r_j c_j
template <int K>
MP_HD double mp_mvnormal_logpdf_chain(const double* x, const double* mu, const double* cov_inv, double ln_det) {
    double c[K], r[K];
#pragma unroll
    for (int i = 0; i < K; ++i) c[i] = x[i] - mu[i];
#pragma unroll
    for (int j = 0; j < K; ++j) r[j] = mp_dot_chain<K>(c, 1, cov_inv + j, K, 0.);
    const double maha = mp_dot_chain<K>(r, 1, c, 1, 0.);
    return -((double)K * MP_LN_2PI_CANON + ln_det + maha) / 2.;
}
// mvnormal.rs:24-37: transform * z + mu with z_j ~ normal(0, 1) in index order from ONE site's stream (`transform` = lower
// Cholesky factor, or the eigen form V sqrt(diag) when the covariance has no Cholesky factor: any K x K matrix here)
template <int K>
MP_HD void mp_mvnormal_sample_chain(mp_site& st, const double* mu, const double* transform, double* out) {
    double z[K];
#pragma unroll 1
    for (int j = 0; j < K; ++j) z[j] = mp_normal_sample(st, 0., 1.);
#pragma unroll
    for (int i = 0; i < K; ++i) out[i] = mp_dot_chain<K>(transform + i * K, 1, z, 1, 0.) + mu[i];
}

// categorical.rs:12-32 over a small probability table
MP_HD double mp_categorical_logpdf(int x, const double* probs, int n) { return (x >= 0 && x < n) ? mp_log(probs[x]) : MP_NEG_INF; }
MP_HD int mp_categorical_sample(mp_site& st, const double* probs, int n) {
    const mp_u64x2 blk = st.next_block();
    return mp_categorical_scan(mp_u01(blk.a), probs, n);
}
