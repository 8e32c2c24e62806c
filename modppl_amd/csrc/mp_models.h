// mp_models.h — the static, handler-polymorphic stand-in for `dyngen!` Unfold kernels.
//
// In modppl a model is ONE function over a dynamic effect handler, and the four GFI methods
// are four interpretations of it (modppl/src/modeling/dyngenfn.rs:39-93; macro rewrite
// `dist(args) %= addr` -> `g.sample_at(&dist, args, addr)`: modppl-macros/src/lib.rs:20-113).
// A closure over `Arc<dyn Any>` tries cannot run on a GPU, so here a model is ONE functor
// templated on the handler type; addresses become compile-time site ids, choices become SoA
// columns, and the handler policies (mp_handlers below, MH policies in mp_mh.h) restate the
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
// Free normal sites may be fed from pre-drawn polar pairs (pu/pr, indexed by Model::normal_index):
// the pair of a site does not depend on its parameters, so k_propagate runs all rejection loops
// of a lane first (a lane-local work queue over particles x sites) and the wave does not idle on
// its slowest lane once per site.  With pu == nullptr the loop runs in place.
template <class Model>
struct mp_generate_handler {
    mp_stream rng;
    const double* obs;
    const double* pu;
    const double* pr;
    double weight;
    MP_HD mp_generate_handler(const mp_stream& r, const double* o, const double* pu_ = nullptr, const double* pr_ = nullptr)
        : rng(r), obs(o), pu(pu_), pr(pr_), weight(0.) {}

    // ln_sd: mp_log(sd) when the caller has it hoisted (a model constant), else NaN -> computed here
    template <int SITE>
    MP_HD double normal(double mu, double sd, double ln_sd) {
        constexpr int k = Model::obs_of(SITE);
        if constexpr (k >= 0) {
            const double x = obs[k];
            weight += mp_normal_logpdf_ln(x, mu, sd, ln_sd);
            return x;
        } else {
            if (pu) {
                constexpr int ni = Model::normal_index(SITE);
                return mp_normal_from_pair(pu[ni], pr[ni], mu, sd);
            }
            mp_site st(rng, MP_DOM_MODEL, (uint32_t)SITE);
            return mp_normal_sample(st, mu, sd);
        }
    }
    template <int SITE>
    MP_HD double normal(double mu, double sd) {
        return normal<SITE>(mu, sd, mp_log(sd));
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
