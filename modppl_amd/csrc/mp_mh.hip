// mp_mh.hip — batched Metropolis-Hastings / regenerative MH over independent chains (gfx950).
//
//   K7a k_mh_drift   metropolis_hastings (modppl/src/inference/mh.rs:9-40) with hierarchical_drift_proposal
//   K7b k_regen_mh   regenerative_metropolis_hastings (mh.rs:54-67) -> DynGenFn::regenerate
// One lane = one chain; the chain state (is_linear, a, b, c) and the cached per-observation
// log-densities live in registers for all n_iters iterations of a launch, so HBM traffic is
// 2 x 32 B per chain per launch: these kernels are fp64-VALU bound, not HBM bound.
//
// Weight rules restated from the handler (one model source, static dispatch):
//   Regenerate (dyngenfn.rs:213-273, 393-446): masked site -> redraw from the prior, diff = Unknown, no weight;
//     unmasked site visited while diff == NoChange -> early return; unmasked site visited after the flip ->
//     weight += logp_new - logp_prev (in visiting order); the `coeffs` sub-call contributes its own weight
//     (0.0 here: a, b, c have constant priors) and flips the outer diff, so every y_i is rescored.
//   Update (dyngenfn.rs:143-211, 321-391) under the drift proposal: constrained site with a previous value ->
//     weight -= prev.weight; weight += logp_new; then the y_i as above.  alpha = weight - fwd + bwd (mh.rs:34).
// Philox: chain = slot, MH iteration (1-based) = step; model redraws DOM_MODEL, proposal draws DOM_PROPOSAL,
// accept uniform DOM_ACCEPT site 0.
#include <hip/hip_runtime.h>

#include <memory>
#include <string>
#include <vector>

#include "../../include/modppl_hip.h"
#include "mp_dists.h"
#include "mp_linalg.h"

typedef unsigned long long u64;
#define MH_MAX_DATA 16
#define MH_THREADS 256

static thread_local std::string g_mh_err;
extern "C" const char* mp_last_error(void);
int32_t mp_set_error(int32_t code, const std::string& msg);  // mp_pf.hip
// mp_pf.hip: L, log-ML, lnw[n] and M categorical draws from a device array of log-weights (importance.rs:21-27, 44-47)
int32_t mp_is_finish_device(const double* d_logw, uint64_t n, uint64_t num_ret, uint64_t seed, int32_t device, void* stream, double* log_ml_estimate,
                            double* log_normalized_weights, uint64_t* resampled_indices);
#define MHCK(call)                                                                                   \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) return mp_set_error(MP_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct mh_data {
    int n;
    double xs[MH_MAX_DATA], ys[MH_MAX_DATA];
};

// mean of y_i: hierarchical.rs:38 / :43   `coeffs.0 + coeffs.1 * x` / `... + coeffs.2 * x * x`
__device__ __forceinline__ double mh_mean(bool is_lin, double a, double b, double c, double x) {
    return is_lin ? a + b * x : a + b * x + c * x * x;
}
#define MH_NOISE 0.1
#define MH_RCP_NOISE (1.0 / MH_NOISE)   // RN(1 / 0.1) = 10.0, folded by the compiler: the observations are scored without a division (mp_div_hoisted)
// mp_log(0.1), hoisted: computed on the host with the same mp_log and passed in

__global__ __launch_bounds__(MH_THREADS) void k_mh_init(u64 n, uint32_t k0, uint32_t k1, int constrain,
                                                        int* __restrict__ is_lin_out, double* __restrict__ a_out, double* __restrict__ b_out,
                                                        double* __restrict__ c_out) {
    const u64 i = (u64)blockIdx.x * MH_THREADS + threadIdx.x;
    if (i >= n) return;
    mp_stream s;
    s.k0 = k0; s.k1 = k1; s.slot = (uint32_t)i; s.step = 0;
    bool is_lin;
    if (constrain >= 0) {
        is_lin = constrain != 0;  // constrained: scored, not sampled (dyngenfn.rs:122-131)
    } else {
        mp_site st(s, MP_DOM_MODEL, MP_SITE_IS_LINEAR);
        is_lin = mp_bernoulli_sample(st, 0.7);
    }
    mp_site sa(s, MP_DOM_MODEL, MP_SITE_A), sb(s, MP_DOM_MODEL, MP_SITE_B), sc(s, MP_DOM_MODEL, MP_SITE_C);
    const double a = mp_normal_sample(sa, 0., 1.);
    const double b = mp_normal_sample(sb, 0., 1.);
    const double c = is_lin ? 0. : mp_normal_sample(sc, 0., 1.);
    is_lin_out[i] = is_lin ? 1 : 0;
    a_out[i] = a; b_out[i] = b; c_out[i] = c;
}

// trace.logjp: sum of all choice log-densities in site order
// The "(y, i)" choices of a chain's trace: the handle's data until a regenerate with an empty mask re-simulated them
// (ys_chain[i][k], dyngenfn.rs:571), per chain afterwards.
__device__ __forceinline__ void mh_load_ys(const mh_data& data, const double* ys_chain, u64 i, double (&y)[MH_MAX_DATA]) {
#pragma unroll
    for (int k = 0; k < MH_MAX_DATA; ++k) y[k] = (k < data.n) ? (ys_chain ? ys_chain[i * (u64)data.n + k] : data.ys[k]) : 0.;
}
__global__ __launch_bounds__(MH_THREADS) void k_mh_logjp(u64 n, mh_data data_, double ln_noise, const int* __restrict__ is_lin_in,
                                                         const double* __restrict__ a_in, const double* __restrict__ b_in,
                                                         const double* __restrict__ c_in, double* __restrict__ out, const double* __restrict__ ys_chain) {
    const u64 i = (u64)blockIdx.x * MH_THREADS + threadIdx.x;
    if (i >= n) return;
    mh_data data = data_;
    mh_load_ys(data_, ys_chain, i, data.ys);
    const bool is_lin = is_lin_in[i] != 0;
    const double a = a_in[i], b = b_in[i], c = c_in[i];
    double lj = mp_bernoulli_logpdf(is_lin, 0.7);
    lj += mp_normal_logpdf_ln(a, 0., 1., 0.);
    lj += mp_normal_logpdf_ln(b, 0., 1., 0.);
    if (!is_lin) lj += mp_normal_logpdf_ln(c, 0., 1., 0.);
    for (int k = 0; k < data.n; ++k) lj += mp_normal_logpdf_h(data.ys[k], mh_mean(is_lin, a, b, c, data.xs[k]), MH_NOISE, ln_noise, MH_RCP_NOISE);
    out[i] = lj;
}

struct mh_mask {
    int n;
    int cycle;
    int sites[4];
};

// KIND: 0 regenerative MH, 1 MH with hierarchical_drift_proposal, 2 MH with add_or_remove_param_proposal
// regenerative_metropolis_hastings with an EMPTY mask (mh.rs:54-67 -> dyngenfn.rs:563-583): `mask.is_leaf()` makes the mask
// the trace's whole schema, so every site is visited masked — is_linear, the coeffs sub-call's a, b (, c), and the
// "(y, i)" sites that were observations — removed and redrawn from its distribution; nothing unmasked is revisited, so
// the weight is 0 and `ln u < 0` accepts every move.  A branch switch is fine here: the sub-call's mask is the OLD
// sub-trace's schema; a vanished c is collected by gc, a new c finds no previous value and is simulated (:253-257).
// The chain's observations become part of its state from then on (ys_chain).  Sites: is_linear 0, a 1, b 2, c 3, (y, k) 4 + k.
__global__ __launch_bounds__(MH_THREADS) void k_mh_regenerate_all(u64 n, uint32_t k0, uint32_t k1, uint32_t iter0, int n_iters, mh_data data,
                                                                  int* __restrict__ is_lin_io, double* __restrict__ a_io, double* __restrict__ b_io,
                                                                  double* __restrict__ c_io, double* __restrict__ ys_chain,
                                                                  u64* __restrict__ accepted_total) {
    const u64 i = (u64)blockIdx.x * MH_THREADS + threadIdx.x;
    u64 acc = 0;
    if (i < n && n_iters > 0) {
        mp_stream s;
        s.k0 = k0; s.k1 = k1; s.slot = (uint32_t)i;
        bool is_lin = false;
        double a = 0., b = 0., c = 0.;
        for (int it = 0; it < n_iters; ++it) {
            s.step = iter0 + (uint32_t)it;
            const mp_u64x2 ub = s.draw(MP_DOM_ACCEPT, 0u, 0u);
            if (!(mp_log(mp_u01(ub.a)) < 0.)) continue;   // mh.rs:62 with weight 0: never taken (u01 < 1); the trace of the LAST accepted move stays
            ++acc;
            if (it != n_iters - 1) continue;              // every site is redrawn from scratch: only the last move's draws survive
            mp_site sl(s, MP_DOM_MODEL, MP_SITE_IS_LINEAR), sa(s, MP_DOM_MODEL, MP_SITE_A), sb(s, MP_DOM_MODEL, MP_SITE_B), sc(s, MP_DOM_MODEL, MP_SITE_C);
            is_lin = mp_bernoulli_sample(sl, 0.7);
            a = mp_normal_sample(sa, 0., 1.);
            b = mp_normal_sample(sb, 0., 1.);
            c = is_lin ? 0. : mp_normal_sample(sc, 0., 1.);
            for (int k = 0; k < data.n; ++k) {
                mp_site sy(s, MP_DOM_MODEL, (uint32_t)(MP_SITE_Y0 + k));
                ys_chain[i * (u64)data.n + k] = mp_normal_sample(sy, mh_mean(is_lin, a, b, c, data.xs[k]), MH_NOISE);
            }
            is_lin_io[i] = is_lin ? 1 : 0;
            a_io[i] = a; b_io[i] = b; c_io[i] = c;
        }
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(accepted_total, acc);
}
__global__ __launch_bounds__(MH_THREADS) void k_mh_broadcast_ys(u64 n, mh_data data, double* __restrict__ ys_chain) {
    const u64 i = (u64)blockIdx.x * MH_THREADS + threadIdx.x;
    if (i >= n) return;
    for (int k = 0; k < data.n; ++k) ys_chain[i * (u64)data.n + k] = data.ys[k];
}

template <int KIND>
__global__ __launch_bounds__(MH_THREADS) void k_mh_iterate(u64 n, uint32_t k0, uint32_t k1, uint32_t iter0, int n_iters, mh_data data,
                                                           double ln_noise, mh_mask mask, double drift_std, double ln_drift_std, double rcp_drift_std,
                                                           int* __restrict__ is_lin_io, double* __restrict__ a_io, double* __restrict__ b_io,
                                                           double* __restrict__ c_io, u64* __restrict__ accepted_total,
                                                           const double* __restrict__ ys_chain) {
    const u64 i = (u64)blockIdx.x * MH_THREADS + threadIdx.x;
    u64 acc = 0;
    if (i < n) {
        mh_load_ys(data, ys_chain, i, data.ys);
        bool is_lin = is_lin_io[i] != 0;
        double a = a_io[i], b = b_io[i], c = c_io[i];
        double ly[MH_MAX_DATA];
#pragma unroll
        for (int k = 0; k < MH_MAX_DATA; ++k)
            ly[k] = (k < data.n) ? mp_normal_logpdf_h(data.ys[k], mh_mean(is_lin, a, b, c, data.xs[k]), MH_NOISE, ln_noise, MH_RCP_NOISE) : 0.;
        mp_stream s;
        s.k0 = k0; s.k1 = k1; s.slot = (uint32_t)i;
        for (int it = 0; it < n_iters; ++it) {
            s.step = iter0 + (uint32_t)it;
            double na = a, nb = b, nc = c;
            bool nl = is_lin;   // proposed structure (only add_or_remove changes it)
            double w = 0.;
            double fwd = 0., bwd = 0.;
            if (KIND == 1) {
                // ---- proposal.propose: simulate hierarchical_drift_proposal (hierarchical.rs:62-70) ----
                mp_site pa(s, MP_DOM_PROPOSAL, MP_SITE_A), pb(s, MP_DOM_PROPOSAL, MP_SITE_B), pc(s, MP_DOM_PROPOSAL, MP_SITE_C);
                na = mp_normal_sample(pa, a, drift_std);
                fwd += mp_normal_logpdf_h(na, a, drift_std, ln_drift_std, rcp_drift_std);
                nb = mp_normal_sample(pb, b, drift_std);
                fwd += mp_normal_logpdf_h(nb, b, drift_std, ln_drift_std, rcp_drift_std);
                if (!is_lin) {
                    nc = mp_normal_sample(pc, c, drift_std);
                    fwd += mp_normal_logpdf_h(nc, c, drift_std, ln_drift_std, rcp_drift_std);
                }
                // ---- model.update(trace, args, NoChange, fwd_choices): inner update of `coeffs` ----
                double dw = 0.;
                dw -= mp_normal_logpdf_ln(a, 0., 1., 0.);
                dw += mp_normal_logpdf_ln(na, 0., 1., 0.);
                dw -= mp_normal_logpdf_ln(b, 0., 1., 0.);
                dw += mp_normal_logpdf_ln(nb, 0., 1., 0.);
                if (!is_lin) {
                    dw -= mp_normal_logpdf_ln(c, 0., 1., 0.);
                    dw += mp_normal_logpdf_ln(nc, 0., 1., 0.);
                }
                dw = dw - 0.;  // gc: weight - complement_weight (nothing unvisited)
                w += dw;
            } else if (KIND == 2) {
                // ---- proposal.propose: simulate add_or_remove_param_proposal (hierarchical.rs:48-61), std 0.025 passed as
                // drift_std; visiting order coeffs/a, coeffs/b, is_linear, coeffs/c ----
                mp_site pa(s, MP_DOM_PROPOSAL, MP_SITE_A), pb(s, MP_DOM_PROPOSAL, MP_SITE_B), pl(s, MP_DOM_PROPOSAL, MP_SITE_IS_LINEAR),
                    pc(s, MP_DOM_PROPOSAL, MP_SITE_C);
                na = mp_normal_sample(pa, a, drift_std);
                fwd += mp_normal_logpdf_h(na, a, drift_std, ln_drift_std, rcp_drift_std);
                nb = mp_normal_sample(pb, b, drift_std);
                fwd += mp_normal_logpdf_h(nb, b, drift_std, ln_drift_std, rcp_drift_std);
                nl = mp_bernoulli_sample(pl, 0.5);
                fwd += mp_bernoulli_logpdf(nl, 0.5);
                if (!nl) {
                    const double prev_c = is_lin ? 0. : c;   // tr.data.search("coeffs/c") (:54-58)
                    nc = mp_normal_sample(pc, prev_c, drift_std);
                    fwd += mp_normal_logpdf_h(nc, prev_c, drift_std, ln_drift_std, rcp_drift_std);
                }
                // ---- model.update(trace, args, NoChange, fwd_choices) (dyngenfn.rs:143-211, 321-391): is_linear is a
                // constrained site with a previous value; `coeffs` is updated by linear() or quadratic() as the NEW
                // is_linear says, on the previous sub-trace: c constrained with or without a previous value, or left
                // unvisited and collected by the inner gc (its weight leaves, the value goes to the discard) ----
                w -= mp_bernoulli_logpdf(is_lin, 0.7);
                w += mp_bernoulli_logpdf(nl, 0.7);
                double dw = 0.;
                dw -= mp_normal_logpdf_ln(a, 0., 1., 0.);
                dw += mp_normal_logpdf_ln(na, 0., 1., 0.);
                dw -= mp_normal_logpdf_ln(b, 0., 1., 0.);
                dw += mp_normal_logpdf_ln(nb, 0., 1., 0.);
                if (!nl) {
                    if (!is_lin) dw -= mp_normal_logpdf_ln(c, 0., 1., 0.);
                    dw += mp_normal_logpdf_ln(nc, 0., 1., 0.);
                    dw = dw - 0.;
                } else {
                    dw = dw - (is_lin ? 0. : mp_normal_logpdf_ln(c, 0., 1., 0.));
                }
                w += dw;
            } else {
                // ---- model.regenerate(trace, args, NoChange, mask): inner regenerate of `coeffs` ----
                bool flipped = false;
                double dw = 0.;
                const int nm = mask.cycle ? 1 : mask.n;
                int m0 = 0;
                if (mask.cycle) m0 = (int)((iter0 - 1u + (uint32_t)it) % (uint32_t)mask.n);
                bool ma = false, mb = false, mc = false;
                for (int q = 0; q < nm; ++q) {
                    const int site = mask.sites[mask.cycle ? m0 : q];
                    ma |= site == MP_SITE_A; mb |= site == MP_SITE_B; mc |= site == MP_SITE_C;
                }
                if (ma) { mp_site st(s, MP_DOM_MODEL, MP_SITE_A); na = mp_normal_sample(st, 0., 1.); flipped = true; }
                else if (flipped) dw += 0.;  // logp_new - logp_prev of an unchanged value under a constant prior
                if (mb) { mp_site st(s, MP_DOM_MODEL, MP_SITE_B); nb = mp_normal_sample(st, 0., 1.); flipped = true; }
                else if (flipped) dw += 0.;
                if (!is_lin) {
                    if (mc) { mp_site st(s, MP_DOM_MODEL, MP_SITE_C); nc = mp_normal_sample(st, 0., 1.); flipped = true; }
                    else if (flipped) dw += 0.;
                }
                w += dw;
            }
            // ---- the observed sites: diff is Unknown, every y_i is rescored in visiting order ----
            double lnew[MH_MAX_DATA];
#pragma unroll
            for (int k = 0; k < MH_MAX_DATA; ++k) {
                if (k < data.n) {
                    lnew[k] = mp_normal_logpdf_h(data.ys[k], mh_mean(nl, na, nb, nc, data.xs[k]), MH_NOISE, ln_noise, MH_RCP_NOISE);
                    w += lnew[k] - ly[k];
                } else {
                    lnew[k] = 0.;
                }
            }
            double alpha = w;
            if (KIND == 2) {
                w = w - 0.;  // gc of the outer update
                // ---- proposal.assess((new trace, ()), discard): the backward proposal scores the old values; its c site
                // exists iff the OLD is_linear was false, centred on the new c if the new trace has one ----
                bwd += mp_normal_logpdf_h(a, na, drift_std, ln_drift_std, rcp_drift_std);
                bwd += mp_normal_logpdf_h(b, nb, drift_std, ln_drift_std, rcp_drift_std);
                bwd += mp_bernoulli_logpdf(is_lin, 0.5);
                if (!is_lin) bwd += mp_normal_logpdf_h(c, nl ? 0. : nc, drift_std, ln_drift_std, rcp_drift_std);
                alpha = w - fwd + bwd;  // mh.rs:34
            }
            if (KIND == 1) {
                w = w - 0.;  // gc of the outer update
                // ---- proposal.assess((new trace, args), discard): generate with the old values constrained ----
                bwd += mp_normal_logpdf_h(a, na, drift_std, ln_drift_std, rcp_drift_std);
                bwd += mp_normal_logpdf_h(b, nb, drift_std, ln_drift_std, rcp_drift_std);
                if (!is_lin) bwd += mp_normal_logpdf_h(c, nc, drift_std, ln_drift_std, rcp_drift_std);
                alpha = w - fwd + bwd;  // mh.rs:34
            }
            const mp_u64x2 ub = s.draw(MP_DOM_ACCEPT, 0u, 0u);
            if (mp_log(mp_u01(ub.a)) < alpha) {  // mh.rs:35 / :62
                a = na; b = nb; c = nl ? 0. : nc;   // a linear trace has no coeffs/c (read_coeffs, hierarchical.rs:5-16)
                is_lin = nl;
#pragma unroll
                for (int k = 0; k < MH_MAX_DATA; ++k) ly[k] = lnew[k];
                ++acc;
            }
        }
        a_io[i] = a; b_io[i] = b; c_io[i] = c;
        if (KIND == 2) is_lin_io[i] = is_lin ? 1 : 0;
    }
    // one atomic per wave
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(accepted_total, acc);
}

// ---------------------------------------------------------------------------------------------
// pointed 2-D model (modppl/tests/dyngenfns/simple.rs:27-41): latent ~ uniform_2d(bounds); obs ~ mvnormal(latent, cov),
// observed; mh with pointed_2d_drift_proposal: latent' ~ mvnormal(latent, noise)  (tests/mh.rs:50-68).
// Sites: latent = 1, obs = 2.  mvnormal.random = L z + mu with z_j ~ normal(0, 1) in index order, L = lower Cholesky
// (mvnormal.rs:24-37); determinant, inverse and Cholesky factor are per-model constants here, per call in modppl.
// ---------------------------------------------------------------------------------------------
struct pointed_params {
    double xmin, xmax, ymin, ymax;
    double neg_ln_area;       // -mp_log((xmax - xmin) * (ymax - ymin)): types_2d.rs uniform_2d.logpdf inside the bounds
    double cov_inv[4], ln_det_cov;
    double obs[2];
};
struct pointed_noise {
    double l00, l10, l11;     // lower Cholesky factor of the proposal covariance
    double inv[4], ln_det;
};
__device__ __forceinline__ double pointed_prior(const pointed_params& P, const double* p) {
    return (P.xmin <= p[0] && p[0] <= P.xmax && P.ymin <= p[1] && p[1] <= P.ymax) ? P.neg_ln_area : MP_NEG_INF;
}
__global__ __launch_bounds__(MH_THREADS) void k_pointed_init(u64 n, uint32_t k0, uint32_t k1, pointed_params P, double* __restrict__ lat) {
    const u64 i = (u64)blockIdx.x * MH_THREADS + threadIdx.x;
    if (i >= n) return;
    mp_stream s;
    s.k0 = k0; s.k1 = k1; s.slot = (uint32_t)i; s.step = 0;
    const mp_u64x2 b = s.draw(MP_DOM_MODEL, 1u, 0u);   // the two uniforms of uniform_2d.random: halves of one block
    lat[2 * i] = mp_u01(b.a) * (P.xmax - P.xmin) + P.xmin;
    lat[2 * i + 1] = mp_u01(b.b) * (P.ymax - P.ymin) + P.ymin;
}
__global__ __launch_bounds__(MH_THREADS) void k_pointed_iterate(u64 n, uint32_t k0, uint32_t k1, uint32_t iter0, int n_iters, pointed_params P,
                                                                pointed_noise N, double* __restrict__ lat, u64* __restrict__ accepted_total) {
    const u64 i = (u64)blockIdx.x * MH_THREADS + threadIdx.x;
    u64 acc = 0;
    if (i < n) {
        double x[2] = {lat[2 * i], lat[2 * i + 1]};
        double lp_lat = pointed_prior(P, x);
        double lp_obs = mp_mvnormal_logpdf_pre<2>(P.obs, x, P.cov_inv, P.ln_det_cov);
        mp_stream s;
        s.k0 = k0; s.k1 = k1; s.slot = (uint32_t)i;
        for (int it = 0; it < n_iters; ++it) {
            s.step = iter0 + (uint32_t)it;
            // proposal.propose: latent' = L z + latent
            mp_site ps(s, MP_DOM_PROPOSAL, 1u);
            const double z0 = mp_normal_sample(ps, 0., 1.);
            const double z1 = mp_normal_sample(ps, 0., 1.);
            double nx[2];
            nx[0] = (0. + N.l00 * z0) + x[0];
            nx[1] = ((0. + N.l10 * z0) + N.l11 * z1) + x[1];
            const double fwd = mp_mvnormal_logpdf_pre<2>(nx, x, N.inv, N.ln_det);
            // model.update: latent constrained with a previous value, obs rescored (diff is Unknown), gc of nothing
            double w = 0.;
            w -= lp_lat;
            const double nlp_lat = pointed_prior(P, nx);
            w += nlp_lat;
            const double nlp_obs = mp_mvnormal_logpdf_pre<2>(P.obs, nx, P.cov_inv, P.ln_det_cov);
            w += nlp_obs - lp_obs;
            w = w - 0.;
            // proposal.assess on the new trace: the old latent under mvnormal(latent', noise)
            const double bwd = mp_mvnormal_logpdf_pre<2>(x, nx, N.inv, N.ln_det);
            const double alpha = w - fwd + bwd;  // mh.rs:34
            const mp_u64x2 ub = s.draw(MP_DOM_ACCEPT, 0u, 0u);
            if (mp_log(mp_u01(ub.a)) < alpha) {
                x[0] = nx[0]; x[1] = nx[1];
                lp_lat = nlp_lat; lp_obs = nlp_obs;
                ++acc;
            }
        }
        lat[2 * i] = x[0]; lat[2 * i + 1] = x[1];
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(accepted_total, acc);
}
__global__ __launch_bounds__(MH_THREADS) void k_pointed_logjp(u64 n, pointed_params P, const double* __restrict__ lat, double* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * MH_THREADS + threadIdx.x;
    if (i >= n) return;
    const double x[2] = {lat[2 * i], lat[2 * i + 1]};
    out[i] = pointed_prior(P, x) + mp_mvnormal_logpdf_pre<2>(P.obs, x, P.cov_inv, P.ln_det_cov);
}

struct mh_fn_ops;   // mp_mh_fn.h: chains of a registered generative function
struct mp_mh {
    u64 n = 0, seed = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    mh_data data{};
    double ln_noise = 0.;
    int* is_lin = nullptr;
    double *a = nullptr, *b = nullptr, *c = nullptr, *tmp = nullptr;
    u64* d_acc = nullptr;
    u64 iters = 0;
    int kind = MP_MH_MODEL_HIERARCHICAL;
    pointed_params pointed{};
    double* lat = nullptr;   // pointed model: [n][2]
    double* ys_chain = nullptr;   // [n][n_data]: per-chain "(y, i)" choices once an empty-mask regenerate re-simulated them
    std::shared_ptr<mh_fn_ops> fn;   // registered function: the trace table below replaces the fields above
    double* fvals = nullptr;         // [2 n_sites][n]: values, then the sub-tries' running weights
    uint32_t* fpresent = nullptr;    // [n]
    // scratch of the standalone GFI calls (mp_fn_*), allocated on first use
    double* gfi_vals = nullptr;      // [n_sites][n]: the discard of an update / the choices of a proposal
    uint32_t* gfi_present = nullptr; // [n]
    double* gfi_cons = nullptr;      // [n_sites][n]: per-chain constraint values
    uint32_t* gfi_cpresent = nullptr; // [n]: per-chain constraint presence
    double* d_data = nullptr;        // models with declared data sites (mp_genfn.h): [2][n_obs] = the covariates (params), the observed values
};
#include "mp_mh_fn.h"

// acceptance count (and, for registered functions, the count of chains that reached a state the reference panics on)
static int32_t mh_finish(mp_mh* h, uint64_t* accepted) {
    if (!accepted && !h->fn) return MP_OK;
    u64 tot[2] = {0, 0};
    MHCK(hipMemcpyAsync(tot, h->d_acc, sizeof(u64) * (h->fn ? 2 : 1), hipMemcpyDeviceToHost, h->stream));
    MHCK(hipStreamSynchronize(h->stream));
    if (accepted) *accepted = tot[0];
    if (tot[1])
        return mp_set_error(MP_ERR_STATE, std::to_string(tot[1]) + " chain-moves reached a case the reference panics on: constraints nobody consumed "
                            "(generate / update / assess, or generate(args, sub) of an unmasked sub-call whose old choices the new branch does not visit: "
                            "dyngenfn.rs:526-529)");
    return MP_OK;
}

extern "C" {

int32_t mp_mh_create(int32_t model_kind, const double* xs, const double* ys, int32_t n_data, int32_t constrain_is_linear,
                     uint64_t n_chains, uint64_t seed, int32_t device, void* stream, mp_mh** out) {
    if (!out) return mp_set_error(MP_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    if (model_kind != MP_MH_MODEL_HIERARCHICAL) return mp_set_error(MP_ERR_UNSUPPORTED, "only MP_MH_MODEL_HIERARCHICAL is compiled in");
    if (!xs || !ys || n_data < 1 || n_data > MH_MAX_DATA) return mp_set_error(MP_ERR_INVALID_ARG, "1 <= n_data <= 16 and xs, ys non-null");
    if (n_chains == 0 || n_chains > 0xFFFFFFFFull) return mp_set_error(MP_ERR_INVALID_ARG, "n_chains must be in [1, 2^32)");
    if (constrain_is_linear < -1 || constrain_is_linear > 1) return mp_set_error(MP_ERR_INVALID_ARG, "constrain_is_linear in {-1, 0, 1}");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        return mp_set_error(MP_ERR_HIP, "no HIP device visible: the gfx950 path has no CPU fallback");
    }
    struct Cleanup { void operator()(mp_mh* p) const { (void)mp_mh_destroy(p); } };   // an early return frees what was allocated so far
    std::unique_ptr<mp_mh, Cleanup> h(new mp_mh());
    h->n = n_chains; h->seed = seed; h->device = device;
    h->data.n = n_data;
    for (int k = 0; k < MH_MAX_DATA; ++k) { h->data.xs[k] = k < n_data ? xs[k] : 0.; h->data.ys[k] = k < n_data ? ys[k] : 0.; }
    h->ln_noise = mp_log(MH_NOISE);
    MHCK(hipSetDevice(device));
    if (stream) h->stream = (hipStream_t)stream;
    else { MHCK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)); h->own_stream = true; }
    MHCK(hipMalloc(&h->is_lin, sizeof(int) * n_chains));
    MHCK(hipMalloc(&h->a, sizeof(double) * n_chains));
    MHCK(hipMalloc(&h->b, sizeof(double) * n_chains));
    MHCK(hipMalloc(&h->c, sizeof(double) * n_chains));
    MHCK(hipMalloc(&h->tmp, sizeof(double) * n_chains * 4));
    MHCK(hipMalloc(&h->d_acc, sizeof(u64)));
    hipLaunchKernelGGL(k_mh_init, dim3((unsigned)((n_chains + MH_THREADS - 1) / MH_THREADS)), dim3(MH_THREADS), 0, h->stream, h->n,
                       (uint32_t)seed, (uint32_t)(seed >> 32), constrain_is_linear, h->is_lin, h->a, h->b, h->c);
    MHCK(hipGetLastError());
    MHCK(hipStreamSynchronize(h->stream));
    *out = h.release();
    return MP_OK;
}

static int32_t mh_run(mp_mh* h, int kind, const mh_mask& mask, double drift_std, int32_t n_iters, uint64_t* accepted) {
    if (n_iters < 0) return mp_set_error(MP_ERR_INVALID_ARG, "n_iters < 0");
    MHCK(hipSetDevice(h->device));
    MHCK(hipMemsetAsync(h->d_acc, 0, sizeof(u64), h->stream));
    const unsigned grid = (unsigned)((h->n + MH_THREADS - 1) / MH_THREADS);
    const uint32_t iter0 = (uint32_t)(h->iters + 1);
    const double ln_ds = kind ? mp_log(drift_std) : 0.;
    if (kind == 2)
        hipLaunchKernelGGL(k_mh_iterate<2>, dim3(grid), dim3(MH_THREADS), 0, h->stream, h->n, (uint32_t)h->seed, (uint32_t)(h->seed >> 32), iter0,
                           n_iters, h->data, h->ln_noise, mask, drift_std, ln_ds, mp_rcp_hoist(drift_std), h->is_lin, h->a, h->b, h->c, h->d_acc, (const double*)h->ys_chain);
    else if (kind == 1)
        hipLaunchKernelGGL(k_mh_iterate<1>, dim3(grid), dim3(MH_THREADS), 0, h->stream, h->n, (uint32_t)h->seed, (uint32_t)(h->seed >> 32), iter0,
                           n_iters, h->data, h->ln_noise, mask, drift_std, ln_ds, mp_rcp_hoist(drift_std), h->is_lin, h->a, h->b, h->c, h->d_acc, (const double*)h->ys_chain);
    else
        hipLaunchKernelGGL(k_mh_iterate<0>, dim3(grid), dim3(MH_THREADS), 0, h->stream, h->n, (uint32_t)h->seed, (uint32_t)(h->seed >> 32), iter0,
                           n_iters, h->data, h->ln_noise, mask, drift_std, ln_ds, mp_rcp_hoist(drift_std), h->is_lin, h->a, h->b, h->c, h->d_acc, (const double*)h->ys_chain);
    MHCK(hipGetLastError());
    h->iters += (u64)n_iters;
    if (accepted) {
        MHCK(hipMemcpyAsync(accepted, h->d_acc, sizeof(u64), hipMemcpyDeviceToHost, h->stream));
        MHCK(hipStreamSynchronize(h->stream));
    }
    return MP_OK;
}

int32_t mp_mh_create_pointed(const double* bounds, const double* obs_cov, const double* obs, uint64_t n_chains, uint64_t seed, int32_t device,
                             void* stream, mp_mh** out) {
    if (!out) return mp_set_error(MP_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    if (!bounds || !obs_cov || !obs) return mp_set_error(MP_ERR_INVALID_ARG, "null argument");
    if (!(bounds[1] > bounds[0]) || !(bounds[3] > bounds[2])) return mp_set_error(MP_ERR_INVALID_ARG, "bounds = {xmin, xmax, ymin, ymax} with xmax > xmin, ymax > ymin");
    if (n_chains == 0 || n_chains > 0xFFFFFFFFull) return mp_set_error(MP_ERR_INVALID_ARG, "n_chains must be in [1, 2^32)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        return mp_set_error(MP_ERR_HIP, "no HIP device visible: the gfx950 path has no CPU fallback");
    }
    const std::vector<double> cov(obs_cov, obs_cov + 4);
    std::vector<double> inv;
    const double det = mp_host_det(cov, 2);
    if (!(det > 0.) || !mp_host_inverse(cov, 2, inv)) return mp_set_error(MP_ERR_INVALID_ARG, "obs covariance must be invertible with a positive determinant");
    struct Cleanup { void operator()(mp_mh* p) const { (void)mp_mh_destroy(p); } };   // an early return frees what was allocated so far
    std::unique_ptr<mp_mh, Cleanup> h(new mp_mh());
    h->kind = MP_MH_MODEL_POINTED_2D;
    h->n = n_chains; h->seed = seed; h->device = device;
    pointed_params& P = h->pointed;
    P.xmin = bounds[0]; P.xmax = bounds[1]; P.ymin = bounds[2]; P.ymax = bounds[3];
    P.neg_ln_area = -mp_log((P.xmax - P.xmin) * (P.ymax - P.ymin));
    for (int q = 0; q < 4; ++q) P.cov_inv[q] = inv[q];
    P.ln_det_cov = mp_log(det);
    P.obs[0] = obs[0]; P.obs[1] = obs[1];
    MHCK(hipSetDevice(device));
    if (stream) h->stream = (hipStream_t)stream;
    else { MHCK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)); h->own_stream = true; }
    MHCK(hipMalloc(&h->lat, sizeof(double) * 2 * n_chains));
    MHCK(hipMalloc(&h->tmp, sizeof(double) * n_chains));
    MHCK(hipMalloc(&h->d_acc, sizeof(u64)));
    hipLaunchKernelGGL(k_pointed_init, dim3((unsigned)((n_chains + MH_THREADS - 1) / MH_THREADS)), dim3(MH_THREADS), 0, h->stream, h->n, (uint32_t)seed,
                       (uint32_t)(seed >> 32), P, h->lat);
    MHCK(hipGetLastError());
    *out = h.release();
    return MP_OK;
}

// chains of a registered generative function with EMPTY traces (the callers below fill them by generate or simulate)
struct mh_cleanup { void operator()(mp_mh* p) const { (void)mp_mh_destroy(p); } };
static int32_t fn_alloc(int32_t model_kind, const double* params, int32_t n_params, uint64_t n_chains, uint64_t seed, int32_t device, void* stream,
                        std::unique_ptr<mp_mh, mh_cleanup>& out) {
    auto it = mh_fn_models().find(model_kind);
    if (it == mh_fn_models().end()) return mp_set_error(MP_ERR_UNSUPPORTED, "no generative function of this kind is registered (MP_REGISTER_MH_MODEL, mp_mh_models.h)");
    if (n_params < 0 || (n_params > 0 && !params)) return mp_set_error(MP_ERR_INVALID_ARG, "bad params");
    if (n_chains == 0 || n_chains > 0xFFFFFFFFull) return mp_set_error(MP_ERR_INVALID_ARG, "n_chains must be in [1, 2^32)");
    std::string err;
    std::shared_ptr<mh_fn_ops> ops = it->second(params, n_params, err);
    if (!ops) return mp_set_error(MP_ERR_INVALID_ARG, err);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        return mp_set_error(MP_ERR_HIP, "no HIP device visible: the gfx950 path has no CPU fallback");
    }
    std::unique_ptr<mp_mh, mh_cleanup> h(new mp_mh());
    h->kind = model_kind;
    h->fn = ops;
    h->n = n_chains; h->seed = seed; h->device = device;
    MHCK(hipSetDevice(device));
    if (stream) h->stream = (hipStream_t)stream;
    else { MHCK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)); h->own_stream = true; }
    MHCK(hipMalloc(&h->fvals, sizeof(double) * n_chains * 2 * (size_t)ops->ns()));   // [n_sites] values + [n_sites] sub-trie running weights (rows of sub-call ids)
    MHCK(hipMalloc(&h->fpresent, sizeof(uint32_t) * n_chains * (size_t)fn_words(ops->ns())));
    MHCK(hipMalloc(&h->tmp, sizeof(double) * n_chains));
    MHCK(hipMalloc(&h->d_acc, sizeof(u64) * 2));
    MHCK(hipMemsetAsync(h->d_acc, 0, sizeof(u64) * 2, h->stream));
    MHCK(hipMemsetAsync(h->fpresent, 0, sizeof(uint32_t) * n_chains * (size_t)fn_words(ops->ns()), h->stream));
    out = std::move(h);
    return MP_OK;
}
static int32_t fn_shared_constraints(int ns, const int32_t* sites, const double* values, int32_t n_cons, mp_fn_consspec& cs) {
    if (n_cons < 0 || (n_cons > 0 && (!sites || !values))) return mp_set_error(MP_ERR_INVALID_ARG, "bad constraints");
    for (int q = 0; q < n_cons; ++q) {
        const int s = sites[q];
        if (s < 0 || s >= ns) return mp_set_error(MP_ERR_INVALID_ARG, "constraint site out of range");
        if (cs.bits & (1ull << s)) return mp_set_error(MP_ERR_INVALID_ARG, "a site is constrained twice");
        cs.bits |= 1ull << s;
        cs.val[s] = values[q];
    }
    return MP_OK;
}
// N x model.generate(args, constraints) at Philox step 0 (tests/mh.rs:91, importance.rs:18-20): the weights stay in h->tmp
static int32_t fn_create_generate(int32_t model_kind, const double* params, int32_t n_params, const int32_t* constraint_sites, const double* constraint_values,
                                  int32_t n_constraints, uint64_t n_chains, uint64_t seed, int32_t device, void* stream, std::unique_ptr<mp_mh, mh_cleanup>& h) {
    int32_t rc = fn_alloc(model_kind, params, n_params, n_chains, seed, device, stream, h);
    if (rc != MP_OK) return rc;
    if (n_constraints < 0 || (n_constraints > 0 && (!constraint_sites || !constraint_values))) return mp_set_error(MP_ERR_INVALID_ARG, "bad constraints");
    // declared data sites (mp_genfn.h): observation j is site id ns + j here and nowhere else — its value goes into a shared array
    const int ns = h->fn->ns(), nd = h->fn->n_data();
    std::vector<int32_t> rsites;
    std::vector<double> rvals;
    if (nd > 0) {
        if (n_params != nd) return mp_set_error(MP_ERR_INVALID_ARG, "a model with declared data sites takes one covariate per observation as its params");
        std::vector<double> buf((size_t)2 * nd);
        std::vector<char> seen((size_t)nd, 0);
        for (int j = 0; j < nd; ++j) buf[(size_t)j] = params[j];
        for (int q = 0; q < n_constraints; ++q) {
            const int s = constraint_sites[q];
            if (s >= ns && s < ns + nd) {
                if (seen[(size_t)(s - ns)]) return mp_set_error(MP_ERR_INVALID_ARG, "a site is constrained twice");
                seen[(size_t)(s - ns)] = 1;
                buf[(size_t)nd + (size_t)(s - ns)] = constraint_values[q];
            } else { rsites.push_back(s); rvals.push_back(constraint_values[q]); }
        }
        for (int j = 0; j < nd; ++j)
            if (!seen[(size_t)j])
                return mp_set_error(MP_ERR_UNSUPPORTED, "every declared data site must be constrained (an observation drawn from its prior would be per-chain state: "
                                                        "write the model with ordinary sites for that)");
        MHCK(hipMalloc(&h->d_data, sizeof(double) * buf.size()));
        MHCK(hipMemcpyAsync(h->d_data, buf.data(), sizeof(double) * buf.size(), hipMemcpyHostToDevice, h->stream));
        MHCK(hipStreamSynchronize(h->stream));   // (`buf` is a local)
        h->fn->bind_data(h->d_data, h->d_data + nd);
        constraint_sites = rsites.data(); constraint_values = rvals.data(); n_constraints = (int32_t)rsites.size();
    }
    mp_fn_consspec cs{};
    rc = fn_shared_constraints(ns, constraint_sites, constraint_values, n_constraints, cs);
    if (rc != MP_OK) return rc;
    rc = h->fn->generate(h.get(), cs, nullptr, nullptr, 0u);
    if (rc != MP_OK) return rc;
    return mh_finish(h.get(), nullptr);   // a constraint on a site the model never visits is the reference's panic
}
int32_t mp_mh_create_fn(int32_t model_kind, const double* params, int32_t n_params, const int32_t* constraint_sites, const double* constraint_values,
                        int32_t n_constraints, uint64_t n_chains, uint64_t seed, int32_t device, void* stream, mp_mh** out) {
    return mp_fn_generate_create(model_kind, params, n_params, constraint_sites, constraint_values, n_constraints, n_chains, seed, device, stream, nullptr, out);
}
int32_t mp_fn_generate_create(int32_t model_kind, const double* params, int32_t n_params, const int32_t* constraint_sites, const double* constraint_values,
                              int32_t n_constraints, uint64_t n_chains, uint64_t seed, int32_t device, void* stream, double* weights_out, mp_mh** out) {
    if (!out) return mp_set_error(MP_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    std::unique_ptr<mp_mh, mh_cleanup> h;
    int32_t rc = fn_create_generate(model_kind, params, n_params, constraint_sites, constraint_values, n_constraints, n_chains, seed, device, stream, h);
    if (rc != MP_OK) return rc;
    if (weights_out) {
        MHCK(hipMemcpyAsync(weights_out, h->tmp, sizeof(double) * h->n, hipMemcpyDeviceToHost, h->stream));
        MHCK(hipStreamSynchronize(h->stream));
    }
    *out = h.release();
    return MP_OK;
}
int32_t mp_fn_simulate_create(int32_t model_kind, const double* params, int32_t n_params, uint64_t n_chains, uint64_t seed, int32_t device, void* stream,
                              double* logjp_out, mp_mh** out) {
    if (!out) return mp_set_error(MP_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    std::unique_ptr<mp_mh, mh_cleanup> h;
    int32_t rc = fn_alloc(model_kind, params, n_params, n_chains, seed, device, stream, h);
    if (rc != MP_OK) return rc;
    if (h->fn->n_data() > 0) return mp_set_error(MP_ERR_UNSUPPORTED, "simulate: the model declares data sites (observed by definition, mp_genfn.h)");
    rc = h->fn->simulate(h.get(), 0u);
    if (rc != MP_OK) return rc;
    if (logjp_out) MHCK(hipMemcpyAsync(logjp_out, h->tmp, sizeof(double) * h->n, hipMemcpyDeviceToHost, h->stream));
    MHCK(hipStreamSynchronize(h->stream));
    *out = h.release();
    return MP_OK;
}
// importance_sampling / importance_resampling over a registered generative function (importance.rs:12-50): N x generate, then the
// canonical normalisation and the M categorical draws of the filters' importance path (mp_is_finish_device, mp_pf.hip)
static int32_t fn_importance(int32_t model_kind, const double* params, int32_t n_params, const int32_t* constraint_sites, const double* constraint_values,
                             int32_t n_constraints, uint64_t num_samples, uint64_t num_ret_samples, uint64_t seed, int32_t device, double* log_ml_estimate,
                             double* log_normalized_weights, uint64_t* resampled_indices, mp_mh** traces_out) {
    if (traces_out) *traces_out = nullptr;
    std::unique_ptr<mp_mh, mh_cleanup> h;
    int32_t rc = fn_create_generate(model_kind, params, n_params, constraint_sites, constraint_values, n_constraints, num_samples, seed, device, nullptr, h);
    if (rc != MP_OK) return rc;
    rc = mp_is_finish_device(h->tmp, h->n, num_ret_samples, seed, device, h->stream, log_ml_estimate, log_normalized_weights, resampled_indices);
    if (rc != MP_OK) return rc;
    if (traces_out) *traces_out = h.release();
    return MP_OK;
}
int32_t mp_fn_importance_sampling(int32_t model_kind, const double* params, int32_t n_params, const int32_t* constraint_sites, const double* constraint_values,
                                  int32_t n_constraints, uint64_t num_samples, uint64_t seed, int32_t device, double* log_ml_estimate,
                                  double* log_normalized_weights, mp_mh** traces_out) {
    return fn_importance(model_kind, params, n_params, constraint_sites, constraint_values, n_constraints, num_samples, 0, seed, device, log_ml_estimate,
                         log_normalized_weights, nullptr, traces_out);
}
int32_t mp_fn_importance_resampling(int32_t model_kind, const double* params, int32_t n_params, const int32_t* constraint_sites, const double* constraint_values,
                                    int32_t n_constraints, uint64_t num_samples, uint64_t num_ret_samples, uint64_t seed, int32_t device,
                                    double* log_ml_estimate, double* log_normalized_weights, uint64_t* resampled_indices, mp_mh** traces_out) {
    if (num_ret_samples > 0 && !resampled_indices) return mp_set_error(MP_ERR_INVALID_ARG, "resampled_indices is null");
    return fn_importance(model_kind, params, n_params, constraint_sites, constraint_values, n_constraints, num_samples, num_ret_samples, seed, device,
                         log_ml_estimate, log_normalized_weights, resampled_indices, traces_out);
}

// Presence words across the C ABI: [chain][W] 32-bit words on the host side, W = (n_sites + 31) / 32 (one word up to 32 sites);
// [word][chain] on the device (mp_mh_fn.h fn_bits_load).  For W = 1 the two are the same array.
static int fn_words_of(const mp_mh* h) { return fn_words(h->fn->ns()); }
static int32_t present_to_host(mp_mh* h, const uint32_t* d_present, uint32_t* host_out) {
    const int W = fn_words_of(h);
    if (W == 1) {
        MHCK(hipMemcpyAsync(host_out, d_present, sizeof(uint32_t) * h->n, hipMemcpyDeviceToHost, h->stream));
        MHCK(hipStreamSynchronize(h->stream));
        return MP_OK;
    }
    std::vector<uint32_t> t((size_t)W * h->n);
    MHCK(hipMemcpyAsync(t.data(), d_present, sizeof(uint32_t) * t.size(), hipMemcpyDeviceToHost, h->stream));
    MHCK(hipStreamSynchronize(h->stream));
    for (u64 i = 0; i < h->n; ++i)
        for (int w = 0; w < W; ++w) host_out[i * (u64)W + w] = t[(size_t)w * h->n + i];
    return MP_OK;
}
static int32_t present_to_device(mp_mh* h, const uint32_t* host_in, uint32_t* d_present) {
    const int W = fn_words_of(h);
    if (W == 1) {
        MHCK(hipMemcpyAsync(d_present, host_in, sizeof(uint32_t) * h->n, hipMemcpyHostToDevice, h->stream));
        MHCK(hipStreamSynchronize(h->stream));
        return MP_OK;
    }
    std::vector<uint32_t> t((size_t)W * h->n);
    for (u64 i = 0; i < h->n; ++i)
        for (int w = 0; w < W; ++w) t[(size_t)w * h->n + i] = host_in[i * (u64)W + w];
    MHCK(hipMemcpyAsync(d_present, t.data(), sizeof(uint32_t) * t.size(), hipMemcpyHostToDevice, h->stream));
    MHCK(hipStreamSynchronize(h->stream));   // `t` is a local
    return MP_OK;
}

int32_t mp_mh_n_sites(mp_mh* h, int32_t* out) {
    if (!h || !out) return mp_set_error(MP_ERR_INVALID_ARG, "null argument");
    if (!h->fn) return mp_set_error(MP_ERR_UNSUPPORTED, "chains of a registered generative function only (mp_mh_create_fn)");
    *out = h->fn->ns();
    return MP_OK;
}

int32_t mp_mh_read_trace(mp_mh* h, double* values, uint32_t* present) {
    if (!h || !values || !present) return mp_set_error(MP_ERR_INVALID_ARG, "null argument");
    if (!h->fn) return mp_set_error(MP_ERR_UNSUPPORTED, "chains of a registered generative function only (mp_mh_create_fn)");
    MHCK(hipSetDevice(h->device));
    const int ns = h->fn->ns();
    std::vector<double> v((size_t)ns * h->n);
    MHCK(hipMemcpyAsync(v.data(), h->fvals, sizeof(double) * v.size(), hipMemcpyDeviceToHost, h->stream));
    { const int32_t rcp = present_to_host(h, h->fpresent, present); if (rcp != MP_OK) return rcp; }   // (waits for the stream)
    for (u64 i = 0; i < h->n; ++i)
        for (int k = 0; k < ns; ++k) values[i * (u64)ns + k] = v[(size_t)k * h->n + i];
    return MP_OK;
}

static int32_t mh_fn_step(mp_mh* h, int32_t proposal_kind, const double* args, int32_t n_args, int32_t n_iters, uint64_t* accepted) {
    if (n_iters < 0) return mp_set_error(MP_ERR_INVALID_ARG, "n_iters < 0");
    if (n_args < 0 || (n_args > 0 && !args)) return mp_set_error(MP_ERR_INVALID_ARG, "bad proposal_args");
    MHCK(hipSetDevice(h->device));
    MHCK(hipMemsetAsync(h->d_acc, 0, sizeof(u64) * 2, h->stream));
    const int32_t rc = h->fn->mh(h, proposal_kind, args, n_args, n_iters);
    if (rc != MP_OK) return rc;
    h->iters += (u64)n_iters;
    return mh_finish(h, accepted);
}
static int32_t mh_fn_regen(mp_mh* h, const int32_t* mask_sites, int32_t n_mask, int32_t cycle, int32_t n_iters, uint64_t* accepted) {
    if (n_iters < 0) return mp_set_error(MP_ERR_INVALID_ARG, "n_iters < 0");
    if (n_mask < 0 || n_mask > MP_FN_MAX_SITES || (n_mask > 0 && !mask_sites)) return mp_set_error(MP_ERR_INVALID_ARG, "bad mask");
    mp_fn_maskspec m{};
    for (int q = 0; q < n_mask; ++q) {
        if (mask_sites[q] < 0 || mask_sites[q] >= h->fn->ns()) return mp_set_error(MP_ERR_INVALID_ARG, "mask site out of range");
        m.bits |= 1ull << mask_sites[q];
        m.cycle[q] = (unsigned char)mask_sites[q];
    }
    m.n_cycle = (cycle && n_mask > 0) ? n_mask : 0;
    if (n_mask == 0 && h->fn->n_data() > 0)
        return mp_set_error(MP_ERR_UNSUPPORTED, "regen_mh with the empty mask re-simulates the whole schema, observed sites included (dyngenfn.rs:571): not for a model with declared data sites");
    MHCK(hipSetDevice(h->device));
    MHCK(hipMemsetAsync(h->d_acc, 0, sizeof(u64) * 2, h->stream));
    const int32_t rc = h->fn->regen(h, m, n_iters);
    if (rc != MP_OK) return rc;
    h->iters += (u64)n_iters;
    return mh_finish(h, accepted);
}

int32_t mp_mh_step(mp_mh* h, int32_t proposal_kind, const double* proposal_args, int32_t n_proposal_args, int32_t n_iters, uint64_t* accepted) {
    if (!h) return mp_set_error(MP_ERR_INVALID_ARG, "null handle");
    if (h->fn) return mh_fn_step(h, proposal_kind, proposal_args, n_proposal_args, n_iters, accepted);
    mh_mask none{};
    if (proposal_kind == MP_MH_PROPOSAL_POINTED_DRIFT) {
        if (h->kind != MP_MH_MODEL_POINTED_2D) return mp_set_error(MP_ERR_INVALID_ARG, "pointed_2d_drift_proposal needs chains of the pointed 2-D model");
        if (!proposal_args || n_proposal_args != 4) return mp_set_error(MP_ERR_INVALID_ARG, "pointed_2d_drift_proposal takes the 2x2 noise covariance (row-major)");
        if (n_iters < 0) return mp_set_error(MP_ERR_INVALID_ARG, "n_iters < 0");
        const std::vector<double> cov(proposal_args, proposal_args + 4);
        std::vector<double> L, inv;
        // A covariance without a Cholesky factor takes mvnormal.random's eigen fallback in the reference (mvnormal.rs:30-33; built for
        // filter sites, mp_linalg.h) — but mh also ASSESSES the proposal (mh.rs:27-34 -> mvnormal.logpdf, :17-18): a singular
        // covariance panics there in try_inverse().unwrap(), an indefinite one gives NaN through sqrt of a negative eigenvalue.
        // No such proposal has a defined acceptance ratio, so the status replaces that panic.
        if (!mp_host_cholesky(cov, 2, L)) return mp_set_error(MP_ERR_INVALID_ARG, "noise covariance is not positive definite: the reference's mh panics on it (mvnormal.logpdf: try_inverse / sqrt of a negative eigenvalue, mvnormal.rs:17-18,30-33)");
        if (!mp_host_inverse(cov, 2, inv)) return mp_set_error(MP_ERR_INVALID_ARG, "noise covariance is not invertible");
        pointed_noise N;
        N.l00 = L[0]; N.l10 = L[2]; N.l11 = L[3];
        for (int q = 0; q < 4; ++q) N.inv[q] = inv[q];
        N.ln_det = mp_log(mp_host_det(cov, 2));
        MHCK(hipSetDevice(h->device));
        MHCK(hipMemsetAsync(h->d_acc, 0, sizeof(u64), h->stream));
        hipLaunchKernelGGL(k_pointed_iterate, dim3((unsigned)((h->n + MH_THREADS - 1) / MH_THREADS)), dim3(MH_THREADS), 0, h->stream, h->n,
                           (uint32_t)h->seed, (uint32_t)(h->seed >> 32), (uint32_t)(h->iters + 1), n_iters, h->pointed, N, h->lat, h->d_acc);
        MHCK(hipGetLastError());
        h->iters += (u64)n_iters;
        if (accepted) {
            MHCK(hipMemcpyAsync(accepted, h->d_acc, sizeof(u64), hipMemcpyDeviceToHost, h->stream));
            MHCK(hipStreamSynchronize(h->stream));
        }
        return MP_OK;
    }
    if (h->kind != MP_MH_MODEL_HIERARCHICAL) return mp_set_error(MP_ERR_INVALID_ARG, "this proposal belongs to the hierarchical model");
    if (proposal_kind == MP_MH_PROPOSAL_HIERARCHICAL_ADD_OR_REMOVE) {
        if (n_proposal_args != 0) return mp_set_error(MP_ERR_INVALID_ARG, "add_or_remove_param_proposal takes no arguments");
        return mh_run(h, 2, none, 0.025, n_iters, accepted);   // hierarchical.rs:51-59: every normal of the proposal has std 0.025
    }
    if (proposal_kind != MP_MH_PROPOSAL_HIERARCHICAL_DRIFT) return mp_set_error(MP_ERR_UNSUPPORTED, "unknown proposal kind");
    if (!proposal_args || n_proposal_args != 1 || !(proposal_args[0] > 0.)) return mp_set_error(MP_ERR_INVALID_ARG, "drift proposal takes {drift_std > 0}");
    return mh_run(h, 1, none, proposal_args[0], n_iters, accepted);
}

int32_t mp_regen_mh_step(mp_mh* h, const int32_t* mask_sites, int32_t n_mask, int32_t cycle, int32_t n_iters, uint64_t* accepted) {
    if (!h) return mp_set_error(MP_ERR_INVALID_ARG, "null handle");
    if (h->fn) return mh_fn_regen(h, mask_sites, n_mask, cycle, n_iters, accepted);
    if (h->kind != MP_MH_MODEL_HIERARCHICAL) return mp_set_error(MP_ERR_UNSUPPORTED, "regen_mh: the hierarchical model's hand-written kernels, or chains of a registered function (mp_mh_create_fn)");
    if (n_mask < 0 || (n_mask > 0 && !mask_sites)) return mp_set_error(MP_ERR_INVALID_ARG, "bad mask");
    if (n_mask == 0) {
        // empty mask = the trace's whole schema (dyngenfn.rs:571): every site, the observed ones included, is redrawn
        if (n_iters < 0) return mp_set_error(MP_ERR_INVALID_ARG, "n_iters < 0");
        MHCK(hipSetDevice(h->device));
        const unsigned grid = (unsigned)((h->n + MH_THREADS - 1) / MH_THREADS);
        if (!h->ys_chain) {
            MHCK(hipMalloc(&h->ys_chain, sizeof(double) * h->n * (size_t)h->data.n));
            hipLaunchKernelGGL(k_mh_broadcast_ys, dim3(grid), dim3(MH_THREADS), 0, h->stream, h->n, h->data, h->ys_chain);
        }
        MHCK(hipMemsetAsync(h->d_acc, 0, sizeof(u64), h->stream));
        hipLaunchKernelGGL(k_mh_regenerate_all, dim3(grid), dim3(MH_THREADS), 0, h->stream, h->n, (uint32_t)h->seed, (uint32_t)(h->seed >> 32),
                           (uint32_t)(h->iters + 1), n_iters, h->data, h->is_lin, h->a, h->b, h->c, h->ys_chain, h->d_acc);
        MHCK(hipGetLastError());
        h->iters += (u64)n_iters;
        if (accepted) {
            MHCK(hipMemcpyAsync(accepted, h->d_acc, sizeof(u64), hipMemcpyDeviceToHost, h->stream));
            MHCK(hipStreamSynchronize(h->stream));
        }
        return MP_OK;
    }
    if (n_mask > 3) return mp_set_error(MP_ERR_INVALID_ARG, "at most 3 mask sites");
    mh_mask m{};
    m.n = n_mask; m.cycle = cycle ? 1 : 0;
    for (int q = 0; q < n_mask; ++q) {
        if (mask_sites[q] == MP_SITE_IS_LINEAR)
            return mp_set_error(MP_ERR_UNSUPPORTED, "masking is_linear: quadratic->linear leaves residual constraints and panics in the reference (dyngenfn.rs:425,526-529)");
        if (mask_sites[q] < MP_SITE_A || mask_sites[q] > MP_SITE_C) return mp_set_error(MP_ERR_INVALID_ARG, "unknown mask site");
        m.sites[q] = mask_sites[q];
    }
    return mh_run(h, 0, m, 1., n_iters, accepted);
}

int32_t mp_mh_read_state(mp_mh* h, double* out) {
    if (!h || !out) return mp_set_error(MP_ERR_INVALID_ARG, "null argument");
    if (h->fn) return mp_set_error(MP_ERR_UNSUPPORTED, "chains of a registered generative function: mp_mh_read_trace");
    MHCK(hipSetDevice(h->device));
    if (h->kind == MP_MH_MODEL_POINTED_2D) {
        MHCK(hipMemcpyAsync(out, h->lat, sizeof(double) * 2 * h->n, hipMemcpyDeviceToHost, h->stream));
        MHCK(hipStreamSynchronize(h->stream));
        return MP_OK;
    }
    std::vector<int> il(h->n);
    std::vector<double> a(h->n), b(h->n), c(h->n);
    MHCK(hipMemcpyAsync(il.data(), h->is_lin, sizeof(int) * h->n, hipMemcpyDeviceToHost, h->stream));
    MHCK(hipMemcpyAsync(a.data(), h->a, sizeof(double) * h->n, hipMemcpyDeviceToHost, h->stream));
    MHCK(hipMemcpyAsync(b.data(), h->b, sizeof(double) * h->n, hipMemcpyDeviceToHost, h->stream));
    MHCK(hipMemcpyAsync(c.data(), h->c, sizeof(double) * h->n, hipMemcpyDeviceToHost, h->stream));
    MHCK(hipStreamSynchronize(h->stream));
    for (u64 i = 0; i < h->n; ++i) { out[4 * i] = il[i]; out[4 * i + 1] = a[i]; out[4 * i + 2] = b[i]; out[4 * i + 3] = c[i]; }
    return MP_OK;
}

int32_t mp_mh_read_logjp(mp_mh* h, double* out) {
    if (!h || !out) return mp_set_error(MP_ERR_INVALID_ARG, "null argument");
    MHCK(hipSetDevice(h->device));
    if (h->fn) {
        const int32_t rc = h->fn->logjp(h);
        if (rc != MP_OK) return rc;
    } else if (h->kind == MP_MH_MODEL_POINTED_2D)
        hipLaunchKernelGGL(k_pointed_logjp, dim3((unsigned)((h->n + MH_THREADS - 1) / MH_THREADS)), dim3(MH_THREADS), 0, h->stream, h->n, h->pointed, h->lat,
                           h->tmp);
    else
    hipLaunchKernelGGL(k_mh_logjp, dim3((unsigned)((h->n + MH_THREADS - 1) / MH_THREADS)), dim3(MH_THREADS), 0, h->stream, h->n, h->data, h->ln_noise,
                       h->is_lin, h->a, h->b, h->c, h->tmp, (const double*)h->ys_chain);
    MHCK(hipGetLastError());
    MHCK(hipMemcpyAsync(out, h->tmp, sizeof(double) * h->n, hipMemcpyDeviceToHost, h->stream));
    MHCK(hipStreamSynchronize(h->stream));
    return MP_OK;
}

int32_t mp_mh_read_observations(mp_mh* h, double* out) {
    if (!h || !out) return mp_set_error(MP_ERR_INVALID_ARG, "null argument");
    if (h->fn || h->kind != MP_MH_MODEL_HIERARCHICAL) return mp_set_error(MP_ERR_UNSUPPORTED, "the hierarchical model's (y, i) choices (registered functions: mp_mh_read_trace)");
    MHCK(hipSetDevice(h->device));
    if (h->ys_chain) {
        MHCK(hipMemcpyAsync(out, h->ys_chain, sizeof(double) * h->n * (size_t)h->data.n, hipMemcpyDeviceToHost, h->stream));
        MHCK(hipStreamSynchronize(h->stream));
    } else {
        for (u64 i = 0; i < h->n; ++i)
            for (int k = 0; k < h->data.n; ++k) out[i * (u64)h->data.n + k] = h->data.ys[k];
    }
    return MP_OK;
}

// ---- the GFI operations one at a time, for chains of a registered generative function (gfi.rs:57-90) -------------------------
// constraints of one call: shared {sites, values} or a per-chain table {chain_values[n][n_sites], chain_present[n]} -> device
struct gfi_cons {
    mp_fn_consspec cs{};
    const double* d_vals = nullptr;
    const uint32_t* d_present = nullptr;
};
static int32_t gfi_scratch(mp_mh* h) {
    const size_t ns = (size_t)h->fn->ns();
    if (!h->gfi_vals) {
        MHCK(hipMalloc(&h->gfi_vals, sizeof(double) * ns * h->n));
        MHCK(hipMalloc(&h->gfi_present, sizeof(uint32_t) * h->n * (size_t)fn_words_of(h)));
        MHCK(hipMalloc(&h->gfi_cons, sizeof(double) * ns * h->n));
        MHCK(hipMalloc(&h->gfi_cpresent, sizeof(uint32_t) * h->n * (size_t)fn_words_of(h)));
    }
    return MP_OK;
}
static int32_t gfi_constraints(mp_mh* h, const int32_t* sites, const double* values, int32_t n_cons, const double* chain_values, const uint32_t* chain_present,
                               gfi_cons& out) {
    const int ns = h->fn->ns();
    if (chain_values || chain_present) {
        if (!chain_values || !chain_present) return mp_set_error(MP_ERR_INVALID_ARG, "per-chain constraints need both chain_values[n][n_sites] and chain_present[n]");
        if (sites || values || n_cons) return mp_set_error(MP_ERR_INVALID_ARG, "constraints are either shared (sites, values) or per chain, not both");
        std::vector<double> t((size_t)ns * h->n);   // [chain][site] -> [site][chain]
        for (u64 i = 0; i < h->n; ++i) {
            const int W = fn_words(ns);
            const uint32_t top = chain_present[i * (u64)W + (W - 1)];   // the last word: bits beyond the model's sites
            if ((ns & 31) && (top >> (ns & 31))) return mp_set_error(MP_ERR_INVALID_ARG, "chain_present names a site the model does not have");
            for (int k = 0; k < ns; ++k) t[(size_t)k * h->n + i] = chain_values[i * (u64)ns + k];
        }
        MHCK(hipMemcpyAsync(h->gfi_cons, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice, h->stream));
        { const int32_t rcp = present_to_device(h, chain_present, h->gfi_cpresent); if (rcp != MP_OK) return rcp; }   // (waits for the stream: `t` is a local)
        out.d_vals = h->gfi_cons;
        out.d_present = h->gfi_cpresent;
        return MP_OK;
    }
    if (n_cons < 0 || (n_cons > 0 && (!sites || !values))) return mp_set_error(MP_ERR_INVALID_ARG, "bad constraints");
    for (int q = 0; q < n_cons; ++q) {
        const int sidx = sites[q];
        if (sidx < 0 || sidx >= ns) return mp_set_error(MP_ERR_INVALID_ARG, "constraint site out of range");
        if (out.cs.bits & (1ull << sidx)) return mp_set_error(MP_ERR_INVALID_ARG, "a site is constrained twice");
        out.cs.bits |= 1ull << sidx;
        out.cs.val[sidx] = values[q];
    }
    return MP_OK;
}
// A standalone GFI call in two halves: gfi_check validates the handle and prepares scratch WITHOUT touching the chains'
// state; gfi_take_step, called once every argument has been accepted and the kernel is about to be launched, clears the counters and
// takes the Philox step (the caller's, or the next MH iteration's, which the call then consumes) — a rejected call leaves `iters` alone.
static int32_t gfi_check(mp_mh* h) {
    if (!h) return mp_set_error(MP_ERR_INVALID_ARG, "null handle");
    if (!h->fn) return mp_set_error(MP_ERR_UNSUPPORTED, "chains of a registered generative function only (mp_mh_create_fn)");
    MHCK(hipSetDevice(h->device));
    return gfi_scratch(h);
}
static int32_t gfi_take_step(mp_mh* h, uint32_t rng_step, uint32_t* step) {
    MHCK(hipMemsetAsync(h->d_acc, 0, sizeof(u64) * 2, h->stream));
    *step = rng_step ? rng_step : (uint32_t)(++h->iters);
    return MP_OK;
}
static int32_t gfi_weights(mp_mh* h, double* weights_out) {
    if (weights_out) MHCK(hipMemcpyAsync(weights_out, h->tmp, sizeof(double) * h->n, hipMemcpyDeviceToHost, h->stream));
    return mh_finish(h, nullptr);   // (waits for the stream; chains that reached a case the reference panics on are an error)
}
static int32_t gfi_table(mp_mh* h, double* values_out, uint32_t* present_out) {   // the discard / the choices: [site][chain] -> [chain][site]
    if (!values_out && !present_out) return MP_OK;
    const int ns = h->fn->ns();
    std::vector<double> v((size_t)ns * h->n);
    MHCK(hipMemcpyAsync(v.data(), h->gfi_vals, sizeof(double) * v.size(), hipMemcpyDeviceToHost, h->stream));
    if (present_out) { const int32_t rcp = present_to_host(h, h->gfi_present, present_out); if (rcp != MP_OK) return rcp; }
    MHCK(hipStreamSynchronize(h->stream));
    if (values_out)
        for (u64 i = 0; i < h->n; ++i)
            for (int k = 0; k < ns; ++k) values_out[i * (u64)ns + k] = v[(size_t)k * h->n + i];
    return MP_OK;
}

int32_t mp_fn_update(mp_mh* h, int32_t argdiff, uint32_t rng_step, const int32_t* sites, const double* values, int32_t n_constraints,
                     const double* chain_values, const uint32_t* chain_present, double* weights_out, double* discard_values_out,
                     uint32_t* discard_present_out) {
    int32_t rc = gfi_check(h);
    if (rc != MP_OK) return rc;
    if (argdiff != MP_ARGDIFF_NOCHANGE && argdiff != MP_ARGDIFF_UNKNOWN) return mp_set_error(MP_ERR_INVALID_ARG, "argdiff: MP_ARGDIFF_NOCHANGE or MP_ARGDIFF_UNKNOWN");
    gfi_cons c;
    rc = gfi_constraints(h, sites, values, n_constraints, chain_values, chain_present, c);
    if (rc != MP_OK) return rc;
    const bool want = discard_values_out || discard_present_out;
    uint32_t step = 0;
    rc = gfi_take_step(h, rng_step, &step);
    if (rc != MP_OK) return rc;
    rc = h->fn->update(h, c.cs, c.d_vals, c.d_present, argdiff == MP_ARGDIFF_UNKNOWN, step, want);
    if (rc != MP_OK) return rc;
    rc = gfi_weights(h, weights_out);
    if (rc != MP_OK) return rc;
    return want ? gfi_table(h, discard_values_out, discard_present_out) : MP_OK;
}

int32_t mp_fn_regenerate(mp_mh* h, int32_t argdiff, uint32_t rng_step, const int32_t* mask_sites, int32_t n_mask, double* weights_out) {
    int32_t rc = gfi_check(h);
    if (rc != MP_OK) return rc;
    if (argdiff != MP_ARGDIFF_NOCHANGE && argdiff != MP_ARGDIFF_UNKNOWN) return mp_set_error(MP_ERR_INVALID_ARG, "argdiff: MP_ARGDIFF_NOCHANGE or MP_ARGDIFF_UNKNOWN");
    if (n_mask < 0 || (n_mask > 0 && !mask_sites)) return mp_set_error(MP_ERR_INVALID_ARG, "bad mask");
    uint64_t bits = 0;
    for (int q = 0; q < n_mask; ++q) {
        if (mask_sites[q] < 0 || mask_sites[q] >= h->fn->ns()) return mp_set_error(MP_ERR_INVALID_ARG, "mask site out of range");
        bits |= 1ull << mask_sites[q];
    }
    if (n_mask == 0 && h->fn->n_data() > 0)
        return mp_set_error(MP_ERR_UNSUPPORTED, "regenerate with the empty mask re-simulates the whole schema, observed sites included (dyngenfn.rs:571): not for a model with declared data sites");
    uint32_t step = 0;
    rc = gfi_take_step(h, rng_step, &step);
    if (rc != MP_OK) return rc;
    rc = h->fn->regenerate(h, bits, argdiff == MP_ARGDIFF_UNKNOWN, step);
    if (rc != MP_OK) return rc;
    return gfi_weights(h, weights_out);
}

int32_t mp_fn_assess(mp_mh* h, int32_t proposal_kind, const double* proposal_args, int32_t n_proposal_args, uint32_t rng_step, const int32_t* sites,
                     const double* values, int32_t n_constraints, const double* chain_values, const uint32_t* chain_present, double* weights_out) {
    int32_t rc = gfi_check(h);
    if (rc != MP_OK) return rc;
    gfi_cons c;
    rc = gfi_constraints(h, sites, values, n_constraints, chain_values, chain_present, c);
    if (rc != MP_OK) return rc;
    if (proposal_kind >= 0 && (n_proposal_args < 0 || (n_proposal_args > 0 && !proposal_args))) return mp_set_error(MP_ERR_INVALID_ARG, "bad proposal_args");
    uint32_t step = 0;
    rc = gfi_take_step(h, rng_step, &step);
    if (rc != MP_OK) return rc;
    if (proposal_kind < 0) rc = h->fn->assess(h, c.cs, c.d_vals, c.d_present, step);
    else rc = h->fn->assess_proposal(h, proposal_kind, proposal_args, n_proposal_args, c.cs, c.d_vals, c.d_present, step);
    if (rc != MP_OK) return rc;
    return gfi_weights(h, weights_out);
}

int32_t mp_fn_propose(mp_mh* h, int32_t proposal_kind, const double* proposal_args, int32_t n_proposal_args, uint32_t rng_step,
                      double* choice_values_out, uint32_t* choice_present_out, double* weights_out) {
    int32_t rc = gfi_check(h);
    if (rc != MP_OK) return rc;
    if (n_proposal_args < 0 || (n_proposal_args > 0 && !proposal_args)) return mp_set_error(MP_ERR_INVALID_ARG, "bad proposal_args");
    uint32_t step = 0;
    rc = gfi_take_step(h, rng_step, &step);
    if (rc != MP_OK) return rc;
    rc = h->fn->propose(h, proposal_kind, proposal_args, n_proposal_args, step);
    if (rc != MP_OK) return rc;
    rc = gfi_weights(h, weights_out);
    if (rc != MP_OK) return rc;
    return gfi_table(h, choice_values_out, choice_present_out);
}

int32_t mp_fn_generate(mp_mh* h, uint32_t rng_step, const int32_t* sites, const double* values, int32_t n_constraints, const double* chain_values,
                       const uint32_t* chain_present, double* weights_out) {
    int32_t rc = gfi_check(h);
    if (rc != MP_OK) return rc;
    gfi_cons c;
    rc = gfi_constraints(h, sites, values, n_constraints, chain_values, chain_present, c);
    if (rc != MP_OK) return rc;
    uint32_t step = 0;
    rc = gfi_take_step(h, rng_step, &step);
    if (rc != MP_OK) return rc;
    rc = h->fn->generate(h, c.cs, c.d_vals, c.d_present, step);
    if (rc != MP_OK) return rc;
    return gfi_weights(h, weights_out);
}

int32_t mp_fn_simulate(mp_mh* h, uint32_t rng_step, double* logjp_out) {
    int32_t rc = gfi_check(h);
    if (rc != MP_OK) return rc;
    if (h->fn->n_data() > 0) return mp_set_error(MP_ERR_UNSUPPORTED, "simulate: the model declares data sites (observed by definition, mp_genfn.h)");
    uint32_t step = 0;
    rc = gfi_take_step(h, rng_step, &step);
    if (rc != MP_OK) return rc;
    rc = h->fn->simulate(h, step);
    if (rc != MP_OK) return rc;
    return gfi_weights(h, logjp_out);
}

int32_t mp_mh_iterations(mp_mh* h, uint64_t* out) {
    if (!h || !out) return mp_set_error(MP_ERR_INVALID_ARG, "null argument");
    *out = h->iters;
    return MP_OK;
}

int32_t mp_mh_destroy(mp_mh* h) {
    if (!h) return MP_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(h->is_lin); (void)hipFree(h->a); (void)hipFree(h->b); (void)hipFree(h->c); (void)hipFree(h->tmp); (void)hipFree(h->d_acc);
    (void)hipFree(h->lat); (void)hipFree(h->ys_chain); (void)hipFree(h->fvals); (void)hipFree(h->fpresent);
    (void)hipFree(h->gfi_vals); (void)hipFree(h->gfi_present); (void)hipFree(h->gfi_cons); (void)hipFree(h->gfi_cpresent); (void)hipFree(h->d_data);
    if (h->own_stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return MP_OK;
}

}  // extern "C"
