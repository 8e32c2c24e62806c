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
template <class Model>
struct mp_generate_handler {
    mp_stream rng;
    const double* obs;
    double weight;
    MP_HD mp_generate_handler(const mp_stream& r, const double* o) : rng(r), obs(o), weight(0.) {}

    template <int SITE>
    MP_HD double normal(double mu, double sd) {
        constexpr int k = Model::obs_of(SITE);
        if constexpr (k >= 0) {
            const double x = obs[k];
            weight += mp_normal_logpdf(x, mu, sd);
            return x;
        } else {
            mp_site st(rng, MP_DOM_MODEL, (uint32_t)SITE);
            return mp_normal_sample(st, mu, sd);
        }
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
    double mu0, sig0, a, sig_x, sig_y;

    template <class H>
    MP_HD void operator()(H& g, int64_t t, const double* prev, double* next) const {
        double x;
        if (t == 0) x = g.template normal<X>(mu0, sig0);
        else x = g.template normal<X>(a * prev[0], sig_x);
        g.template normal<Y>(x, sig_y);
        next[0] = x;
    }
};
