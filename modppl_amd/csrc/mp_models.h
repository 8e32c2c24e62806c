// mp_models.h — the static, handler-polymorphic stand-in for `dyngen!` Unfold kernels.
//
// In modppl a model is ONE function over a dynamic effect handler, and the four GFI methods
// are four interpretations of it (modppl/src/modeling/dyngenfn.rs:39-93; macro rewrite
// `dist(args) %= addr` -> `g.sample_at(&dist, args, addr)`: modppl-macros/src/lib.rs:20-113).
// A closure over `Arc<dyn Any>` tries cannot run on a GPU, so here a model is ONE functor
// templated on the handler type; addresses become compile-time site ids, choices become dense per-particle
// rows, and the handler policies (mp_generate_handler below, the MH policies in mp_mh.hip) restate the
// weight rules of sample_at.  A kernel is
//
//     template <class H> void operator()(H& g, int64_t t, const double* prev, double* next) const
//
// (`prev`/`next` = DIM_STATE doubles), matching DynUnfold's kernel signature
// `fn(&mut DynGenFnHandler, (i64, State)) -> State` (modppl/src/modeling/dynunfold.rs:7-10).
// Which sites are constrained is part of the handler (`OBS_OF(site)`), exactly as it is part
// of the constraint trie in the reference.
#pragma once
#include "mp_dists.h"

#define MP_MAX_OBS 16
#define MP_MAX_STATE 16

struct mp_obs {
    double v[MP_MAX_OBS];
};
struct mp_state0 {
    double v[MP_MAX_STATE];
};

// ---------------------------------------------------------------------------------------
// Generate-mode handler of DynUnfold::generate / update(Extend) (dynunfold.rs:48-61, 80-95):
//   constrained site : x = constraint; logp = logpdf(x); weight += logp     (dyngenfn.rs:122-131)
//   free site        : x ~ dist.random(prng); (logp only feeds the trie)    (dyngenfn.rs:132-136)
// Model::obs_of(site) >= 0 names the constraint slot of a site (compile time).
// ---------------------------------------------------------------------------------------
//
// Free normal sites may be fed from pre-drawn STANDARD deviates z = u*c of the polar method (pz, indexed by
// Model::normal_index): z does not depend on the site's parameters, so the rejection loops run ahead of the model
// (workgroup-cooperatively in k_propagate, or one launch earlier inside the resample's lookup kernel) and the model
// finishes with z*sd + mu (mp_dists.h).  With pz == nullptr the loop runs in place.
template <class Model>
struct mp_generate_handler {
    mp_stream rng;
    const double* obs;
    const double* pz;
    double weight;
    MP_HD mp_generate_handler(const mp_stream& r, const double* o, const double* pz_ = nullptr)
        : rng(r), obs(o), pz(pz_), weight(0.) {}

    // transcendental functions of a model body go through the handler, so that the CPU checker's interpretation of the same
    // functor can apply its own arithmetic mode (oracle/src/functor_adapter.hpp)
    MP_HD double exp_(double x) const { return mp_exp(x); }
    MP_HD double log_(double x) const { return mp_log(x); }
    MP_HD double sin_(double x) const { return mp_sin(x); }
    MP_HD double cos_(double x) const { return mp_cos(x); }
    MP_HD double atan2_(double y, double x) const { return mp_atan2(y, x); }
    // ln_sd: mp_log(sd) when the caller has it hoisted (a model constant), else NaN -> computed here
    template <int SITE>
    MP_HD double normal(double mu, double sd, double ln_sd) {
        constexpr int k = Model::obs_of(SITE);
        if constexpr (k >= 0) {
            const double x = obs[k];
            weight += mp_normal_logpdf_ln(x, mu, sd, ln_sd);
            return x;
        } else {
            if (pz) {
                constexpr int ni = Model::normal_index(SITE);
                return pz[ni] * sd + mu;
            }
            mp_site st(rng, MP_DOM_MODEL, (uint32_t)SITE);
            return mp_normal_sample(st, mu, sd);
        }
    }
    template <int SITE>
    MP_HD double normal(double mu, double sd) {
        return normal<SITE>(mu, sd, mp_log(sd));
    }
    // constrained mvnormal site with hoisted covariance constants (dim K = Model::DIM_OBS slots k..k+K-1)
    template <int SITE, int K>
    MP_HD void mvnormal_observed(const double* mu, const double* cov_inv, double ln_det, const double* /*chol: Simulate only*/,
                                 const double* /*cov: for interpreters that work from the covariance itself (the CPU checker)*/ = nullptr) {
        constexpr int k = Model::obs_of(SITE);
        static_assert(k >= 0, "mvnormal_observed: the site must be constrained on this path");
        weight += mp_mvnormal_logpdf_pre<K>(obs + k, mu, cov_inv, ln_det);
    }
    // mvnormal site of dimension K in the matrix core's accumulation order (mp_dists.h): constrained -> scored, free -> drawn
    // (transform z + mu, the K normals one after the other from this site's stream).  x: the value (in: constraint slot / out: draw).
    template <int SITE, int K>
    MP_HD void mvnormal_chain(const double* mu, const double* transform, const double* cov_inv, double ln_det, double* x) {
        constexpr int k = Model::obs_of(SITE);
        if constexpr (k >= 0) {
#pragma unroll
            for (int i = 0; i < K; ++i) x[i] = obs[k + i];
            weight += mp_mvnormal_logpdf_chain<K>(x, mu, cov_inv, ln_det);
        } else {
            mp_site st(rng, MP_DOM_MODEL, (uint32_t)SITE);
            mp_mvnormal_sample_chain<K>(st, mu, transform, x);
        }
    }
    // categorical over a small table; values travel as doubles in the constraint / state arrays
    template <int SITE>
    MP_HD int categorical(const double* probs, int n) {
        constexpr int k = Model::obs_of(SITE);
        if constexpr (k >= 0) {
            const int x = (int)obs[k];
            weight += mp_categorical_logpdf(x, probs, n);
            return x;
        } else {
            mp_site st(rng, MP_DOM_MODEL, (uint32_t)SITE);
            return mp_categorical_sample(st, probs, n);
        }
    }
    // uniform_2d (modppl/tests/pointed_model/types_2d.rs:14-32): both coordinates from ONE site, i.e. from consecutive
    // uniforms of its stream = the two halves of one Philox block.  Free sites only on this path.
    template <int SITE>
    MP_HD void uniform_2d(double xmin, double xmax, double ymin, double ymax, double* out) {
        const mp_u64x2 b = rng.draw(MP_DOM_MODEL, (uint32_t)SITE, 0u);
        out[0] = mp_u01(b.a) * (xmax - xmin) + xmin;
        out[1] = mp_u01(b.b) * (ymax - ymin) + ymin;
    }
    template <int SITE>
    MP_HD double uniform(double a, double b) {
        constexpr int k = Model::obs_of(SITE);
        if constexpr (k >= 0) {
            const double x = obs[k];
            weight += mp_uniform_logpdf(x, a, b);
            return x;
        } else {
            mp_site st(rng, MP_DOM_MODEL, (uint32_t)SITE);
            return mp_uniform_sample(st, a, b);
        }
    }
};

// ---------------------------------------------------------------------------------------
// Simulate-mode handler of DynUnfold::simulate (dynunfold.rs:22-39; sample_at Simulate arm, dyngenfn.rs:104-113): every
// site is drawn from its distribution, the ones that are observations on the filtering path too (their values go to
// obs_out in constraint-slot order).  No weight: a simulated trace only has choices.
// ---------------------------------------------------------------------------------------
template <class Model>
struct mp_simulate_handler {
    mp_stream rng;
    double* obs_out;
    MP_HD mp_simulate_handler(const mp_stream& r, double* o) : rng(r), obs_out(o) {}
    MP_HD double exp_(double x) const { return mp_exp(x); }
    MP_HD double log_(double x) const { return mp_log(x); }
    MP_HD double sin_(double x) const { return mp_sin(x); }
    MP_HD double cos_(double x) const { return mp_cos(x); }
    MP_HD double atan2_(double y, double x) const { return mp_atan2(y, x); }

    template <int SITE>
    MP_HD double normal(double mu, double sd, double /*ln_sd*/) {
        mp_site st(rng, MP_DOM_MODEL, (uint32_t)SITE);
        const double x = mp_normal_sample(st, mu, sd);
        constexpr int k = Model::obs_of(SITE);
        if constexpr (k >= 0) obs_out[k] = x;
        return x;
    }
    template <int SITE>
    MP_HD double normal(double mu, double sd) { return normal<SITE>(mu, sd, 0.); }
    // mvnormal.random (mvnormal.rs:24-37): L z + mu, z_j ~ normal(0, 1) in index order from the site's stream
    template <int SITE, int K>
    MP_HD void mvnormal_observed(const double* mu, const double*, double, const double* chol, const double* = nullptr) {
        constexpr int k = Model::obs_of(SITE);
        mp_site st(rng, MP_DOM_MODEL, (uint32_t)SITE);
        double z[K];
        for (int j = 0; j < K; ++j) z[j] = mp_normal_sample(st, 0., 1.);
        for (int i = 0; i < K; ++i) {
            double acc = 0.;
            for (int j = 0; j <= i; ++j) acc += chol[i * K + j] * z[j];
            obs_out[k + i] = acc + mu[i];
        }
    }
    template <int SITE, int K>
    MP_HD void mvnormal_chain(const double* mu, const double* transform, const double*, double, double* x) {
        mp_site st(rng, MP_DOM_MODEL, (uint32_t)SITE);
        mp_mvnormal_sample_chain<K>(st, mu, transform, x);
        constexpr int k = Model::obs_of(SITE);
        if constexpr (k >= 0) {
#pragma unroll
            for (int i = 0; i < K; ++i) obs_out[k + i] = x[i];
        }
    }
    template <int SITE>
    MP_HD int categorical(const double* probs, int n) {
        mp_site st(rng, MP_DOM_MODEL, (uint32_t)SITE);
        const int x = mp_categorical_sample(st, probs, n);
        constexpr int k = Model::obs_of(SITE);
        if constexpr (k >= 0) obs_out[k] = (double)x;
        return x;
    }
    template <int SITE>
    MP_HD void uniform_2d(double xmin, double xmax, double ymin, double ymax, double* out) {
        const mp_u64x2 b = rng.draw(MP_DOM_MODEL, (uint32_t)SITE, 0u);
        out[0] = mp_u01(b.a) * (xmax - xmin) + xmin;
        out[1] = mp_u01(b.b) * (ymax - ymin) + ymin;
    }
    template <int SITE>
    MP_HD double uniform(double a, double b) {
        mp_site st(rng, MP_DOM_MODEL, (uint32_t)SITE);
        return mp_uniform_sample(st, a, b);
    }
};

// ---------------------------------------------------------------------------------------
// LGSSM d=1 (BASELINE.json configs 1-2):
//   t==0: x ~ normal(mu0, sig0) %= "x";  t>0: x ~ normal(a*x_prev, sig_x) %= "x";  normal(x, sig_y) %= "y"
// ---------------------------------------------------------------------------------------
struct mp_lgssm1 {
    static constexpr int DIM_STATE = 1, DIM_OBS = 1;
    enum { X = 0, Y = 1 };
    static constexpr int obs_of(int site) { return site == Y ? 0 : -1; }
    // free normal sites, in pre-draw order
    static constexpr int MAX_NORMALS = 1;
    static constexpr int normal_index(int site) { return site == X ? 0 : -1; }
    MP_HD int n_normals(int64_t) const { return 1; }
    MP_HD uint32_t normal_site(int) const { return X; }
    double mu0, sig0, a, sig_x, sig_y;
    double ln_sig_y;  // mp_log(sig_y), hoisted (same bits on host and device)

    template <class H>
    MP_HD void operator()(H& g, int64_t t, const double* prev, double* next) const {
        double x;
        if (t == 0) x = g.template normal<X>(mu0, sig0);
        else x = g.template normal<X>(a * prev[0], sig_x);
        g.template normal<Y>(x, sig_y, ln_sig_y);
        next[0] = x;
    }
};

// ---------------------------------------------------------------------------------------
// spiral_kernel — modppl/tests/dyngenfns/unfold.rs:14-32 (the reference's own DynUnfold model):
//   t==0: r ~ uniform(0,1) %= "r"; theta ~ uniform(0,2pi) %= "theta"
//   t>0 : dr ~ normal(0,0.1) %= "dr"; dtheta ~ normal(0.4,0.2) %= "dtheta"; pol = prev + (dr, dtheta)
//   mvnormal(polar_to_cartesian(pol), 0.001*I) %= "obs"
// ---------------------------------------------------------------------------------------
struct mp_spiral {
    static constexpr int DIM_STATE = 2, DIM_OBS = 2;
    enum { R = 0, THETA = 1, OBS = 2 };
    static constexpr int obs_of(int site) { return site == OBS ? 0 : -1; }
    static constexpr int obs_dim(int site) { return site == OBS ? 2 : 1; }   // a vector-valued site owns obs slots obs_of .. obs_of + obs_dim - 1
    static constexpr int MAX_NORMALS = 2;
    static constexpr int normal_index(int site) { return site; }  // dr -> 0, dtheta -> 1
    MP_HD int n_normals(int64_t t) const { return t == 0 ? 0 : 2; }
    MP_HD uint32_t normal_site(int idx) const { return (uint32_t)idx; }
    double cov_inv[4];
    double ln_det;
    double chol[4];   // lower Cholesky factor of the covariance, row-major (mvnormal.random in Simulate mode)
    double cov[4];    // the covariance itself, row-major: what the reference's mvnormal is called with (mvnormal.rs:12-37 derives the rest per call)

    template <class H>
    MP_HD void operator()(H& g, int64_t t, const double* prev, double* next) const {
        double pol0, pol1;
        if (t == 0) {
            pol0 = g.template uniform<R>(0., 1.);
            pol1 = g.template uniform<THETA>(0., 2. * MP_PI);
        } else {
            const double dr = g.template normal<R>(0., 0.1);
            const double dtheta = g.template normal<THETA>(0.4, 0.2);
            pol0 = prev[0] + dr;
            pol1 = prev[1] + dtheta;
        }
        const double pos[2] = {pol0 * g.cos_(pol1), pol0 * g.sin_(pol1)};
        g.template mvnormal_observed<OBS, 2>(pos, cov_inv, ln_det, chol, cov);
        next[0] = pol0;
        next[1] = pol1;
    }
};

// ---------------------------------------------------------------------------------------
// HMM — modppl/tests/hmm/model.rs:33-80: new_state ~ categorical(prior | transition column);
// weight = categorical.logpdf(observation, emission column of the new state).
// Column-stochastic tables, stored emission[o][s], transition[s2][s1] as the reference builds them.
// ---------------------------------------------------------------------------------------
#define MP_HMM_MAX 8
struct mp_hmm {
    static constexpr int DIM_STATE = 1, DIM_OBS = 1;
    enum { STATE = 0, OBSV = 1 };
    static constexpr int obs_of(int site) { return site == OBSV ? 0 : -1; }
    static constexpr int MAX_NORMALS = 1;  // none used; arrays need a non-zero extent
    static constexpr int normal_index(int) { return 0; }
    MP_HD int n_normals(int64_t) const { return 0; }
    MP_HD uint32_t normal_site(int) const { return 0; }
    int n_states, n_obs;
    double prior[MP_HMM_MAX];
    double emission_col[MP_HMM_MAX][MP_HMM_MAX];    // [state][obs]  : column `state` of the emission matrix
    double transition_col[MP_HMM_MAX][MP_HMM_MAX];  // [prev][next]  : column `prev` of the transition matrix

    template <class H>
    MP_HD void operator()(H& g, int64_t t, const double* prev, double* next) const {
        int s;
        if (t == 0) {
            s = g.template categorical<STATE>(prior, n_states);
        } else {
            int ps = (int)prev[0];
            ps = ps < 0 ? 0 : (ps >= n_states ? n_states - 1 : ps);
            s = g.template categorical<STATE>(transition_col[ps], n_states);
        }
        g.template categorical<OBSV>(emission_col[s], n_obs);
        next[0] = (double)s;
    }
};

// ---------------------------------------------------------------------------------------
// Bearings-only tracker, d=4 (BASELINE.json config 3), dt = 1:
//   t==0: px ~ normal(p0x, sig_p0); py ~ normal(p0y, sig_p0); vx ~ normal(0, sig_v0); vy ~ normal(0, sig_v0)
//   t>0 : ax ~ normal(0, sig_a); ay ~ normal(0, sig_a); v' = v + a; p' = (p + v) + 0.5*a
//   theta ~ normal(atan2(py, px), sig_theta) observed (no wrap)
// ---------------------------------------------------------------------------------------
struct mp_bearings {
    static constexpr int DIM_STATE = 4, DIM_OBS = 1;
    enum { S0 = 0, S1 = 1, S2 = 2, S3 = 3, THETA = 4 };
    static constexpr int obs_of(int site) { return site == THETA ? 0 : -1; }
    static constexpr int MAX_NORMALS = 4;
    static constexpr int normal_index(int site) { return site; }
    MP_HD int n_normals(int64_t t) const { return t == 0 ? 4 : 2; }
    MP_HD uint32_t normal_site(int idx) const { return (uint32_t)idx; }
    double p0x, p0y, sig_p0, sig_v0, sig_a, sig_theta;
    double ln_sig_theta;

    template <class H>
    MP_HD void operator()(H& g, int64_t t, const double* prev, double* next) const {
        double px, py, vx, vy;
        if (t == 0) {
            px = g.template normal<S0>(p0x, sig_p0);
            py = g.template normal<S1>(p0y, sig_p0);
            vx = g.template normal<S2>(0., sig_v0);
            vy = g.template normal<S3>(0., sig_v0);
        } else {
            const double ax = g.template normal<S0>(0., sig_a);
            const double ay = g.template normal<S1>(0., sig_a);
            px = (prev[0] + prev[2]) + 0.5 * ax;
            py = (prev[1] + prev[3]) + 0.5 * ay;
            vx = prev[2] + ax;
            vy = prev[3] + ay;
        }
        g.template normal<THETA>(g.atan2_(py, px), sig_theta, ln_sig_theta);
        next[0] = px; next[1] = py; next[2] = vx; next[3] = vy;
    }
};

// ---------------------------------------------------------------------------------------
// Banded LGSSM, d = D (BASELINE.json config 5): A = a*(I + band*B), B = ones on the two off-diagonals
//   t==0: x_j ~ normal(0, sig0) %= "x/j";  t>0: x_j ~ normal(a*(x_j + band*(x_{j-1}+x_{j+1})), sig_x) %= "x/j"
//   normal(x_j, sig_y) %= "y/j" observed, j = 0..D-1.    sites: x/j -> j, y/j -> D + j
// ---------------------------------------------------------------------------------------
template <int D>
struct mp_lgssm_band {
    static constexpr int DIM_STATE = D, DIM_OBS = D;
    static constexpr int obs_of(int site) { return site >= D ? site - D : -1; }
    static constexpr int MAX_NORMALS = D;
    static constexpr int normal_index(int site) { return site; }
    MP_HD int n_normals(int64_t) const { return D; }
    MP_HD uint32_t normal_site(int idx) const { return (uint32_t)idx; }
    double a, band, sig0, sig_x, sig_y;
    double ln_sig_y;

    template <class H, int J>
    MP_HD void site(H& g, int64_t t, const double* prev, double* next) const {
        double x;
        if (t == 0) {
            x = g.template normal<J>(0., sig0);
        } else {
            const double nb = (J > 0 ? prev[J > 0 ? J - 1 : 0] : 0.) + (J < D - 1 ? prev[J < D - 1 ? J + 1 : 0] : 0.);
            x = g.template normal<J>(a * (prev[J] + band * nb), sig_x);
        }
        g.template normal<D + J>(x, sig_y, ln_sig_y);
        next[J] = x;
        if constexpr (J + 1 < D) site<H, J + 1>(g, t, prev, next);
    }
    template <class H>
    MP_HD void operator()(H& g, int64_t t, const double* prev, double* next) const {
        site<H, 0>(g, t, prev, next);
    }
};

// ---------------------------------------------------------------------------------------
// Dense LGSSM, d = D: the state transition really is a dense matvec (BASELINE.json north star: "MFMA only where a state
// transition really is a dense matvec").  Two mvnormal sites (mvnormal.rs:14-38), covariance constants hoisted to the host:
//   t==0: x ~ mvnormal(0, sig0^2 I) %= "x";   t>0: x ~ mvnormal(A x_prev, Q) %= "x";   mvnormal(x, R) %= "y" observed
// Every product is a k-ascending fma chain (mp_dists.h): the scalar interpretation below (any handler) and the MFMA kernel
// k_propagate_dense16 (Generate mode, mp_pf_kernels.h) agree bit for bit.  The matrices live in device memory (4 x D x D
// doubles do not fit kernel arguments): mats = [A | TQ | T0 | Rinv | TR], row-major; TQ / T0 / TR = mvnormal.random's
// `transform` of Q, sig0^2 I and R (TR is only used by Simulate, which draws the observation too).
// ---------------------------------------------------------------------------------------
template <int D>
struct mp_lgssm_dense {
    static constexpr int DIM_STATE = D, DIM_OBS = D;
    enum { X = 0, Y = 1 };
    static constexpr int obs_of(int site) { return site == Y ? 0 : -1; }
    static constexpr int MAX_NORMALS = 1;  // no independent normal sites: the D normals of "x" are ONE site's sequential stream
    static constexpr int normal_index(int) { return 0; }
    MP_HD int n_normals(int64_t) const { return 0; }
    MP_HD uint32_t normal_site(int) const { return 0; }
    const double* mats;   // [5][D][D]
    double ln_det_R;
    MP_HD const double* A() const { return mats; }
    MP_HD const double* TQ() const { return mats + D * D; }
    MP_HD const double* T0() const { return mats + 2 * D * D; }
    MP_HD const double* Rinv() const { return mats + 3 * D * D; }
    MP_HD const double* TR() const { return mats + 4 * D * D; }

    template <class H>
    MP_HD void operator()(H& g, int64_t t, const double* prev, double* next) const {
        double mean[D], y[D];
#pragma unroll
        for (int i = 0; i < D; ++i) mean[i] = (t == 0) ? 0. : mp_dot_chain<D>(A() + i * D, 1, prev, 1, 0.);
        g.template mvnormal_chain<X, D>(mean, t == 0 ? T0() : TQ(), nullptr, 0., next);
        g.template mvnormal_chain<Y, D>(next, TR(), Rinv(), ln_det_R, y);
    }
};

// ---------------------------------------------------------------------------------------
// Static (T = 1) models of the reference's importance tests (modppl/tests/importance.rs), run through the Unfold entry
// points with a single time step; later steps leave the state alone and score nothing.
//   pointed_2d_model — tests/dyngenfns/simple.rs:27-34: latent ~ uniform_2d(bounds); mvnormal(latent, cov) %= "obs"
//   line_model       — tests/dyngenfns/simple.rs:9-24: slope ~ normal(0,1); intercept ~ normal(0,2);
//                      ys/i ~ normal(slope * x_i + intercept, 0.1), observed
// ---------------------------------------------------------------------------------------
struct mp_pointed2d {
    static constexpr int DIM_STATE = 2, DIM_OBS = 2;
    enum { LATENT = 0, OBS = 1 };
    static constexpr int obs_of(int site) { return site == OBS ? 0 : -1; }
    static constexpr int MAX_NORMALS = 1;  // none used
    static constexpr int normal_index(int) { return 0; }
    MP_HD int n_normals(int64_t) const { return 0; }
    MP_HD uint32_t normal_site(int) const { return 0; }
    double xmin, xmax, ymin, ymax;
    double cov_inv[4];
    double ln_det;
    double chol[4];   // lower Cholesky factor of the covariance, row-major (mvnormal.random in Simulate mode)

    template <class H>
    MP_HD void operator()(H& g, int64_t t, const double* prev, double* next) const {
        if (t != 0) { next[0] = prev[0]; next[1] = prev[1]; return; }
        double latent[2];
        g.template uniform_2d<LATENT>(xmin, xmax, ymin, ymax, latent);
        g.template mvnormal_observed<OBS, 2>(latent, cov_inv, ln_det, chol);
        next[0] = latent[0];
        next[1] = latent[1];
    }
};

template <int N>
struct mp_line {
    static constexpr int DIM_STATE = 2, DIM_OBS = N;
    enum { SLOPE = 0, INTERCEPT = 1, YS = 2 };   // ys/i = site YS + i
    static constexpr int obs_of(int site) { return site >= YS ? site - YS : -1; }
    static constexpr int MAX_NORMALS = 2;
    static constexpr int normal_index(int site) { return site; }
    MP_HD int n_normals(int64_t t) const { return t == 0 ? 2 : 0; }
    MP_HD uint32_t normal_site(int idx) const { return (uint32_t)idx; }
    double xs[N];
    double ln_noise;   // mp_log(0.1)

    template <class H, int J>
    MP_HD void ys(H& g, double slope, double intercept) const {
        g.template normal<YS + J>(slope * xs[J] + intercept, 0.1, ln_noise);
        if constexpr (J + 1 < N) ys<H, J + 1>(g, slope, intercept);
    }
    template <class H>
    MP_HD void operator()(H& g, int64_t t, const double* prev, double* next) const {
        if (t != 0) { next[0] = prev[0]; next[1] = prev[1]; return; }
        const double slope = g.template normal<SLOPE>(0., 1.);
        const double intercept = g.template normal<INTERCEPT>(0., 2.);
        ys<H, 0>(g, slope, intercept);
        next[0] = slope;
        next[1] = intercept;
    }
};

// ---------------------------------------------------------------------------------------
// Registration layer (f3: one model source).  A translation unit that wants the models registered defines
// MP_MODEL_REGISTRAR to the name of a function template `template <class M> int R(int kind, bool (*parse)(const
// mp_model_desc&, M&, std::string&))` before including this header: mp_pf.hip registers a device factory (ModelOpsT<M>),
// the CPU checker its two interpretations of the same functor.  Everywhere else the macro expands to nothing.
// ---------------------------------------------------------------------------------------
// (active in the device pass too: the registrar's instantiation is what makes hipcc emit the model's kernels there)
#include <string>
#if defined(MP_MODEL_REGISTRAR)
#define MP_REGISTER_UNFOLD_MODEL(KIND, TYPE, PARSE) static const int mp_registered_##TYPE = MP_MODEL_REGISTRAR<TYPE>(KIND, PARSE);
#else
#define MP_REGISTER_UNFOLD_MODEL(KIND, TYPE, PARSE)
#endif
#include "mp_models_extra.h"
