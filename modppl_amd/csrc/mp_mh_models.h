// mp_mh_models.h — models and proposals for mh / regen_mh written ONCE against the handler interface of mp_genfn.h
// (the static stand-in for `dyngen!` bodies) and registered by one line each; nothing else to touch.
//
// A model is a functor `template <class H> void operator()(H& g) const` with `static constexpr int NS` sites (plus `sub_of(site)` and
// `is_bool(site)`, which only the checker's dynamic interpretation reads); a proposal is
// `template <class H, class T> void operator()(H& g, const T& tr) const` over the SAME site ids (a proposal's addresses are the
// model's: mh.rs:17-23 feeds its choices to model.update as constraints).  Both are trivially copyable (they travel to the
// kernels by value) and are filled from the C ABI's double arrays by a parse function.
//   MP_REGISTER_MH_MODEL(kind, Type, parse)                          kinds >= 100 are free
//   MP_REGISTER_MH_PROPOSAL(proposal_kind, ModelType, Type, parse)
// mp_mh.hip instantiates k_fn_generate / k_fn_simulate / k_fn_regen / k_fn_mh / k_fn_logjp for every registration; chains are created with
// mp_mh_create_fn and driven by the same mp_mh_step / mp_regen_mh_step as the hand-written kernels.
#pragma once
#include <string>
#include <vector>

#include "mp_genfn.h"
#include "mp_linalg.h"

// ---------------------------------------------------------------------------------------
// hierarchical_model (modppl/tests/dyngenfns/hierarchical.rs:17-47), kind 101 — the functor form of the model the
// hand-written k_mh_iterate kernels restate; tests/test_gpu_mh.py requires the two to agree bit for bit.
//   params = xs[0 .. n_data), n_data <= 16; the observations are constraints on the sites Y0 + k of mp_mh_create_fn.
// ---------------------------------------------------------------------------------------
struct mp_hier_fn {
    static constexpr int MAX_DATA = 16;
    static constexpr int NS = 4 + MAX_DATA;
    enum { IS_LINEAR = 0, A = 1, B = 2, C = 3, Y0 = 4 };   // = enum mp_mh_site
    static constexpr uint32_t COEFFS = (1u << A) | (1u << B) | (1u << C);   // the sub-trace at "coeffs"
    // what a DYNAMIC interpretation of the functor needs to know about its sites (the CPU checker runs the same source through
    // tries: oracle/src/mh_functor_adapter.hpp): the sub-call a site lives in (0: top level), and whether its value is a bool
    static constexpr uint32_t sub_of(int site) { return (site >= A && site <= C) ? COEFFS : 0u; }
    static constexpr bool is_bool(int site) { return site == IS_LINEAR; }
    int n;
    double xs[MAX_DATA];
    double ln_noise;   // mp_log(0.1)

    template <class H, int J>
    MP_HD void ys(H& g, bool lin, double a, double b, double c) const {
        if (J < n) {
            const double x = xs[J];
            g.template normal<Y0 + J>(lin ? a + b * x : a + b * x + c * x * x, 0.1, ln_noise);   // hierarchical.rs:38 / :43
        }
        if constexpr (J + 1 < MAX_DATA) ys<H, J + 1>(g, lin, a, b, c);
    }
    template <class H>
    MP_HD void operator()(H& g) const {
        const bool lin = g.template bernoulli<IS_LINEAR>(0.7);
        // let coeffs = linear() / quadratic() /= "coeffs" (:17-30, :37, :42): the coefficients are the sub-call's return value
        const mp_fn_ret co = g.template call<COEFFS>([&](H& q) {
            mp_fn_ret r{};
            r.v[0] = q.template normal<A>(0., 1., 0.);
            r.v[1] = q.template normal<B>(0., 1., 0.);
            if (!lin) r.v[2] = q.template normal<C>(0., 1., 0.);
            return r;
        });
        ys<H, 0>(g, lin, co.v[0], co.v[1], co.v[2]);
    }
};
inline bool mp_parse_hier_fn(const double* params, int n_params, mp_hier_fn& m, std::string& err) {
    if (!params || n_params < 1 || n_params > mp_hier_fn::MAX_DATA) { err = "hierarchical model: params = xs[0 .. n_data), 1 <= n_data <= 16"; return false; }
    m.n = n_params;
    for (int k = 0; k < mp_hier_fn::MAX_DATA; ++k) m.xs[k] = k < n_params ? params[k] : 0.;
    m.ln_noise = mp_log(0.1);
    return true;
}
MP_REGISTER_MH_MODEL(101, mp_hier_fn, mp_parse_hier_fn)

// hierarchical_drift_proposal(tr, drift_std) (hierarchical.rs:62-70); args = {drift_std}
struct mp_hier_drift_fn {
    double sd, ln_sd;
    template <class H, class T>
    MP_HD void operator()(H& g, const T& tr) const {
        g.template normal<mp_hier_fn::A>(tr.val[mp_hier_fn::A], sd, ln_sd);
        g.template normal<mp_hier_fn::B>(tr.val[mp_hier_fn::B], sd, ln_sd);
        if (tr.val[mp_hier_fn::IS_LINEAR] == 0.) g.template normal<mp_hier_fn::C>(tr.val[mp_hier_fn::C], sd, ln_sd);
    }
};
inline bool mp_parse_hier_drift_fn(const double* args, int n_args, mp_hier_drift_fn& p, std::string& err) {
    if (!args || n_args != 1 || !(args[0] > 0.)) { err = "drift proposal takes {drift_std > 0}"; return false; }
    p.sd = args[0];
    p.ln_sd = mp_log(args[0]);
    return true;
}
MP_REGISTER_MH_PROPOSAL(1, mp_hier_fn, mp_hier_drift_fn, mp_parse_hier_drift_fn)

// add_or_remove_param_proposal(tr) (hierarchical.rs:48-61): visiting order coeffs/a, coeffs/b, is_linear, coeffs/c; no args
struct mp_hier_add_or_remove_fn {
    double sd, ln_sd;   // 0.025
    template <class H, class T>
    MP_HD void operator()(H& g, const T& tr) const {
        g.template normal<mp_hier_fn::A>(tr.val[mp_hier_fn::A], sd, ln_sd);
        g.template normal<mp_hier_fn::B>(tr.val[mp_hier_fn::B], sd, ln_sd);
        if (!g.template bernoulli<mp_hier_fn::IS_LINEAR>(0.5)) {
            const double prev_c = tr.get(mp_hier_fn::C, 0.);   // tr.data.search("coeffs/c") (:54-58)
            g.template normal<mp_hier_fn::C>(prev_c, sd, ln_sd);
        }
    }
};
inline bool mp_parse_hier_add_or_remove_fn(const double*, int n_args, mp_hier_add_or_remove_fn& p, std::string& err) {
    if (n_args != 0) { err = "add_or_remove_param_proposal takes no arguments"; return false; }
    p.sd = 0.025;
    p.ln_sd = mp_log(0.025);
    return true;
}
MP_REGISTER_MH_PROPOSAL(2, mp_hier_fn, mp_hier_add_or_remove_fn, mp_parse_hier_add_or_remove_fn)

// ---------------------------------------------------------------------------------------
// hierarchical_model with DECLARED data sites, kind 105 (round 5; mp_genfn.h "DECLARED DATA SITES"): the same model — same sites 0..3,
// same sub-call, same expression for an observation's mean — whose "(y, j)" sites are not register-resident sites 4 + j but data:
// any number of observations (`for (i, x) in xs.iter().enumerate()`, hierarchical.rs:36-45, has no bound), four sites of trace.
//   params = xs[0 .. n_obs); the observations are constraints on the site ids NS + j = 4 + j of the creating call, all of them.
// The same proposals (kinds 1, 2: they address sites 0..3 only).
// ---------------------------------------------------------------------------------------
struct mp_hier_data_fn {
    static constexpr int NS = 4;
    static constexpr bool HAS_DATA = true;
    static constexpr int MAX_OBS = 1 << 20;
    enum { IS_LINEAR = 0, A = 1, B = 2, C = 3 };
    static constexpr uint32_t COEFFS = (1u << A) | (1u << B) | (1u << C);
    static constexpr uint32_t sub_of(int site) { return (site >= A && site <= C) ? COEFFS : 0u; }
    static constexpr bool is_bool(int site) { return site == IS_LINEAR; }
    int n_obs;
    const double* xs;   // [n_obs] the design (params), ...
    const double* ys;   // [n_obs] ... the observed values: shared arrays in the handle's memory space (bind)
    double ln_noise;    // mp_log(0.1)
    double rcp_noise;   // mp_rcp_hoist(0.1): an observation's (y - mu) / 0.1 without a division, same bits
    struct latents { bool lin; double a, b, c; };
    template <class V>
    MP_HD latents latents_of(const V& v) const { return latents{v.val[IS_LINEAR] != 0., v.val[A], v.val[B], v.val[C]}; }
    MP_HD mp_fn_normal datum(int j, const latents& l) const {
        const double x = xs[j];
        return mp_fn_normal{l.lin ? l.a + l.b * x : l.a + l.b * x + l.c * x * x, 0.1, ln_noise, rcp_noise};   // hierarchical.rs:38 / :43
    }
    MP_HD double obs(int j) const { return ys[j]; }
    void bind(const double* cov, const double* obs_) { xs = cov; ys = obs_; }
    template <class H>
    MP_HD void operator()(H& g) const {
        const bool lin = g.template bernoulli<IS_LINEAR>(0.7);
        const mp_fn_ret co = g.template call<COEFFS>([&](H& q) {
            mp_fn_ret r{};
            r.v[0] = q.template normal<A>(0., 1., 0.);
            r.v[1] = q.template normal<B>(0., 1., 0.);
            if (!lin) r.v[2] = q.template normal<C>(0., 1., 0.);
            return r;
        });
        g.data(*this, latents{lin, co.v[0], co.v[1], co.v[2]});
    }
};
inline bool mp_parse_hier_data_fn(const double* params, int n_params, mp_hier_data_fn& m, std::string& err) {
    if (!params || n_params < 1 || n_params > mp_hier_data_fn::MAX_OBS) { err = "hierarchical model (declared data): params = xs[0 .. n_obs), n_obs >= 1"; return false; }
    m.n_obs = n_params;
    m.xs = nullptr; m.ys = nullptr;
    m.ln_noise = mp_log(0.1);
    m.rcp_noise = mp_rcp_hoist(0.1);
    return true;
}
MP_REGISTER_MH_MODEL(105, mp_hier_data_fn, mp_parse_hier_data_fn)
struct mp_hier_data_drift_fn : mp_hier_drift_fn {};
inline bool mp_parse_hier_data_drift_fn(const double* args, int n_args, mp_hier_data_drift_fn& p, std::string& err) { return mp_parse_hier_drift_fn(args, n_args, p, err); }
MP_REGISTER_MH_PROPOSAL(1, mp_hier_data_fn, mp_hier_data_drift_fn, mp_parse_hier_data_drift_fn)
struct mp_hier_data_add_or_remove_fn : mp_hier_add_or_remove_fn {};
inline bool mp_parse_hier_data_add_or_remove_fn(const double* args, int n_args, mp_hier_data_add_or_remove_fn& p, std::string& err) {
    return mp_parse_hier_add_or_remove_fn(args, n_args, p, err);
}
MP_REGISTER_MH_PROPOSAL(2, mp_hier_data_fn, mp_hier_data_add_or_remove_fn, mp_parse_hier_data_add_or_remove_fn)

// ---------------------------------------------------------------------------------------
// Robust regression with outlier indicators, kind 102 — a model that exists ONLY here (no hand-written kernel, no hand-written
// restatement in the checker): the test of the generic layer proper.  The shape of Gen's MCMC tutorial model:
//   line() /= "line":  slope ~ normal(0, 2) %= "slope";  intercept ~ normal(0, 2) %= "intercept"
//   for k:  is_outlier_k ~ bernoulli(0.1) %= ("outlier", k);  y_k ~ normal(slope x_k + intercept, is_outlier_k ? 5 : 0.5) %= ("y", k)
//   params = xs[0 .. n_data), n_data <= 12; the observations are constraints on the sites Y0 + k.
// Moves: proposal 1 = drift of the line {std}; proposal 2 = flip of one indicator {k} (proposes the other value with
// probability 0.95); regen_mh over any sites (block resimulation of indicators, of the line, or both).
// ---------------------------------------------------------------------------------------
struct mp_robust_line_fn {
    static constexpr int MAX_DATA = 12;
    static constexpr int NS = 2 + 2 * MAX_DATA;
    enum { SLOPE = 0, INTERCEPT = 1, OUT0 = 2, Y0 = 2 + MAX_DATA };
    static constexpr uint32_t LINE = (1u << SLOPE) | (1u << INTERCEPT);
    static constexpr uint32_t sub_of(int site) { return site <= INTERCEPT ? LINE : 0u; }
    static constexpr bool is_bool(int site) { return site >= OUT0 && site < Y0; }
    int n;
    double xs[MAX_DATA];
    double ln_prior_sd, ln_sd_in, ln_sd_out;   // mp_log(2), mp_log(0.5), mp_log(5)

    template <class H, int J>
    MP_HD void points(H& g, double slope, double intercept) const {
        if (J < n) {
            const bool out = g.template bernoulli<OUT0 + J>(0.1);
            g.template normal<Y0 + J>(slope * xs[J] + intercept, out ? 5. : 0.5, out ? ln_sd_out : ln_sd_in);
        }
        if constexpr (J + 1 < MAX_DATA) points<H, J + 1>(g, slope, intercept);
    }
    template <class H>
    MP_HD void operator()(H& g) const {
        const mp_fn_ret line = g.template call<LINE>([&](H& q) {
            mp_fn_ret r{};
            r.v[0] = q.template normal<SLOPE>(0., 2., ln_prior_sd);
            r.v[1] = q.template normal<INTERCEPT>(0., 2., ln_prior_sd);
            return r;
        });
        points<H, 0>(g, line.v[0], line.v[1]);
    }
};
inline bool mp_parse_robust_line_fn(const double* params, int n_params, mp_robust_line_fn& m, std::string& err) {
    if (!params || n_params < 1 || n_params > mp_robust_line_fn::MAX_DATA) { err = "robust line: params = xs[0 .. n_data), 1 <= n_data <= 12"; return false; }
    m.n = n_params;
    for (int k = 0; k < mp_robust_line_fn::MAX_DATA; ++k) m.xs[k] = k < n_params ? params[k] : 0.;
    m.ln_prior_sd = mp_log(2.); m.ln_sd_in = mp_log(0.5); m.ln_sd_out = mp_log(5.);
    return true;
}
MP_REGISTER_MH_MODEL(102, mp_robust_line_fn, mp_parse_robust_line_fn)

struct mp_robust_line_drift_fn {
    double sd, ln_sd;
    template <class H, class T>
    MP_HD void operator()(H& g, const T& tr) const {
        g.template normal<mp_robust_line_fn::SLOPE>(tr.val[mp_robust_line_fn::SLOPE], sd, ln_sd);
        g.template normal<mp_robust_line_fn::INTERCEPT>(tr.val[mp_robust_line_fn::INTERCEPT], sd, ln_sd);
    }
};
inline bool mp_parse_robust_line_drift_fn(const double* args, int n_args, mp_robust_line_drift_fn& p, std::string& err) {
    if (!args || n_args != 1 || !(args[0] > 0.)) { err = "line drift proposal takes {std > 0}"; return false; }
    p.sd = args[0];
    p.ln_sd = mp_log(args[0]);
    return true;
}
MP_REGISTER_MH_PROPOSAL(1, mp_robust_line_fn, mp_robust_line_drift_fn, mp_parse_robust_line_drift_fn)

struct mp_robust_line_flip_fn {
    int k;
    template <class H, class T, int J>
    MP_HD void flip(H& g, const T& tr) const {
        if (J == k) g.template bernoulli<mp_robust_line_fn::OUT0 + J>(tr.val[mp_robust_line_fn::OUT0 + J] != 0. ? 0.05 : 0.95);
        if constexpr (J + 1 < mp_robust_line_fn::MAX_DATA) flip<H, T, J + 1>(g, tr);
    }
    template <class H, class T>
    MP_HD void operator()(H& g, const T& tr) const { flip<H, T, 0>(g, tr); }
};
inline bool mp_parse_robust_line_flip_fn(const double* args, int n_args, mp_robust_line_flip_fn& p, std::string& err) {
    if (!args || n_args != 1 || !(args[0] >= 0.) || !(args[0] < mp_robust_line_fn::MAX_DATA) || args[0] != (double)(int)args[0]) {
        err = "flip proposal takes {k}: the index of the data point whose indicator is re-proposed";
        return false;
    }
    p.k = (int)args[0];
    return true;
}
MP_REGISTER_MH_PROPOSAL(2, mp_robust_line_fn, mp_robust_line_flip_fn, mp_parse_robust_line_flip_fn)

// ---------------------------------------------------------------------------------------
// A line whose noise regime is chosen BEFORE the line's sub-call, kind 103 (functor only):
//   big ~ bernoulli(0.3) %= "big";  (slope, intercept) = line() /= "line";  y_k ~ normal(slope x_k + intercept, big ? 2 : 0.5)
//   params = xs[0 .. n_data), n_data <= 10; the observations are constraints on the sites Y0 + k.
// What it is for: a change upstream of an UNTOUCHED sub-call — regen_mh with mask {big} takes trace_at's
// `generate(args, sub)` arm (weight += new_weight - sub.weight(), dyngenfn.rs:424-428), mh with the toggle proposal the
// `update(sub, args, Unknown, {})` arm (:371-381): both work on the sub-trie's running weight.
// Moves: proposal 1 = toggle of `big` (proposes the other value with probability 0.9); proposal 2 = drift of the line {std}.
// ---------------------------------------------------------------------------------------
struct mp_scaled_line_fn {
    static constexpr int MAX_DATA = 10;
    static constexpr int NS = 3 + MAX_DATA;
    enum { BIG = 0, SLOPE = 1, INTERCEPT = 2, Y0 = 3 };
    static constexpr uint32_t LINE = (1u << SLOPE) | (1u << INTERCEPT);
    static constexpr uint32_t sub_of(int site) { return (site == SLOPE || site == INTERCEPT) ? LINE : 0u; }
    static constexpr bool is_bool(int site) { return site == BIG; }
    int n;
    double xs[MAX_DATA];
    double ln_prior_sd, ln_sd_small, ln_sd_big;   // mp_log(2), mp_log(0.5), mp_log(2)

    template <class H, int J>
    MP_HD void points(H& g, bool big, double slope, double intercept) const {
        if (J < n) g.template normal<Y0 + J>(slope * xs[J] + intercept, big ? 2. : 0.5, big ? ln_sd_big : ln_sd_small);
        if constexpr (J + 1 < MAX_DATA) points<H, J + 1>(g, big, slope, intercept);
    }
    template <class H>
    MP_HD void operator()(H& g) const {
        const bool big = g.template bernoulli<BIG>(0.3);
        const mp_fn_ret line = g.template call<LINE>([&](H& q) {
            mp_fn_ret r{};
            r.v[0] = q.template normal<SLOPE>(0., 2., ln_prior_sd);
            r.v[1] = q.template normal<INTERCEPT>(0., 2., ln_prior_sd);
            return r;
        });
        points<H, 0>(g, big, line.v[0], line.v[1]);
    }
};
inline bool mp_parse_scaled_line_fn(const double* params, int n_params, mp_scaled_line_fn& m, std::string& err) {
    if (!params || n_params < 1 || n_params > mp_scaled_line_fn::MAX_DATA) { err = "scaled line: params = xs[0 .. n_data), 1 <= n_data <= 10"; return false; }
    m.n = n_params;
    for (int k = 0; k < mp_scaled_line_fn::MAX_DATA; ++k) m.xs[k] = k < n_params ? params[k] : 0.;
    m.ln_prior_sd = mp_log(2.); m.ln_sd_small = mp_log(0.5); m.ln_sd_big = mp_log(2.);
    return true;
}
MP_REGISTER_MH_MODEL(103, mp_scaled_line_fn, mp_parse_scaled_line_fn)

struct mp_scaled_line_toggle_fn {
    int unused;
    template <class H, class T>
    MP_HD void operator()(H& g, const T& tr) const {
        g.template bernoulli<mp_scaled_line_fn::BIG>(tr.val[mp_scaled_line_fn::BIG] != 0. ? 0.1 : 0.9);
    }
};
inline bool mp_parse_scaled_line_toggle_fn(const double*, int n_args, mp_scaled_line_toggle_fn& p, std::string& err) {
    if (n_args != 0) { err = "toggle proposal takes no arguments"; return false; }
    p.unused = 0;
    return true;
}
MP_REGISTER_MH_PROPOSAL(1, mp_scaled_line_fn, mp_scaled_line_toggle_fn, mp_parse_scaled_line_toggle_fn)

struct mp_scaled_line_drift_fn {
    double sd, ln_sd;
    template <class H, class T>
    MP_HD void operator()(H& g, const T& tr) const {
        g.template normal<mp_scaled_line_fn::SLOPE>(tr.val[mp_scaled_line_fn::SLOPE], sd, ln_sd);
        g.template normal<mp_scaled_line_fn::INTERCEPT>(tr.val[mp_scaled_line_fn::INTERCEPT], sd, ln_sd);
    }
};
inline bool mp_parse_scaled_line_drift_fn(const double* args, int n_args, mp_scaled_line_drift_fn& p, std::string& err) {
    if (!args || n_args != 1 || !(args[0] > 0.)) { err = "line drift proposal takes {std > 0}"; return false; }
    p.sd = args[0];
    p.ln_sd = mp_log(args[0]);
    return true;
}
MP_REGISTER_MH_PROPOSAL(2, mp_scaled_line_fn, mp_scaled_line_drift_fn, mp_parse_scaled_line_drift_fn)

// ---------------------------------------------------------------------------------------
// Two LEVELS of sub-calls (kind 113): trace_at inside trace_at (dyngenfn.rs:283-449 is recursive through GenFn::update /
// regenerate / generate of the callee, which is a DynGenFn with a handler of its own).  No reference test nests calls; the model is
// built so that every arm is reached at depth two — constraints that land in the inner call only, a structure change in the middle
// one (gc there), a change upstream of both (update(sub, Unknown, {}) twice over; generate(args, sub) twice over under regenerate).
//   a ~ normal(0, 2) %= "a"
//   e = mid(a) /= "mid":     b ~ normal(a, 1) %= "b"
//                            (c, f) = inner(b) /= "inner":   c ~ normal(b, 0.5) %= "c";  f ~ bernoulli(0.3) %= "f"
//                            if f { d ~ normal(c, 1) %= "d" }
//                            e ~ normal(c + b + (f ? d : 0), 0.7) %= "e"
//   y_j ~ normal(e x_j, 0.3) %= "y_j",  j < n_data   (constraints);  params = xs[0 .. n_data), n_data <= 4
// Moves: proposal 1 = drift of b and c {std}; proposal 2 = flip of f (proposes the other value with probability 0.8, and d ~ normal(c, 1) with a true f);
// proposal 3 = drift of a {std}.
// ---------------------------------------------------------------------------------------
struct mp_nested_fn {
    static constexpr int MAX_DATA = 4;
    static constexpr int NS = 6 + MAX_DATA;
    enum { A = 0, B = 1, C = 2, F = 3, D = 4, E = 5, Y0 = 6 };
    static constexpr uint32_t INNER = (1u << C) | (1u << F);
    static constexpr uint32_t MID = (1u << B) | INNER | (1u << D) | (1u << E);   // (an outer call's sites include the inner call's)
    static constexpr uint32_t sub_of(int site) { return (site == C || site == F) ? INNER : ((site == B || site == D || site == E) ? MID : 0u); }
    static constexpr uint32_t outer_of(int site) { return (site == C || site == F) ? MID : 0u; }
    static constexpr bool is_bool(int site) { return site == F; }
    int n;
    double xs[MAX_DATA];
    double ln2, ln_half, ln_07, ln_03;

    template <class H, int J>
    MP_HD void points(H& g, double e) const {
        if (J < n) g.template normal<Y0 + J>(e * xs[J], 0.3, ln_03);
        if constexpr (J + 1 < MAX_DATA) points<H, J + 1>(g, e);
    }
    template <class H>
    MP_HD void operator()(H& g) const {
        const double a = g.template normal<A>(0., 2., ln2);
        const mp_fn_ret mid = g.template call<MID>([&](H& q) {
            const double b = q.template normal<B>(a, 1., 0.);
            const mp_fn_ret in = q.template call<INNER>([&](H& q2) {
                mp_fn_ret r{};
                r.v[0] = q2.template normal<C>(b, 0.5, ln_half);
                r.v[1] = q2.template bernoulli<F>(0.3) ? 1. : 0.;
                return r;
            });
            double d = 0.;
            if (in.v[1] != 0.) d = q.template normal<D>(in.v[0], 1., 0.);
            mp_fn_ret r{};
            r.v[0] = q.template normal<E>(in.v[0] + b + d, 0.7, ln_07);
            return r;
        });
        points<H, 0>(g, mid.v[0]);
    }
};
inline bool mp_parse_nested_fn(const double* params, int n_params, mp_nested_fn& m, std::string& err) {
    if (!params || n_params < 1 || n_params > mp_nested_fn::MAX_DATA) { err = "nested calls: params = xs[0 .. n_data), 1 <= n_data <= 4"; return false; }
    m.n = n_params;
    for (int k = 0; k < mp_nested_fn::MAX_DATA; ++k) m.xs[k] = k < n_params ? params[k] : 0.;
    m.ln2 = mp_log(2.); m.ln_half = mp_log(0.5); m.ln_07 = mp_log(0.7); m.ln_03 = mp_log(0.3);
    return true;
}
MP_REGISTER_MH_MODEL(113, mp_nested_fn, mp_parse_nested_fn)

struct mp_nested_drift_bc_fn {
    double sd, ln_sd;
    template <class H, class T>
    MP_HD void operator()(H& g, const T& tr) const {
        g.template normal<mp_nested_fn::B>(tr.val[mp_nested_fn::B], sd, ln_sd);
        g.template normal<mp_nested_fn::C>(tr.val[mp_nested_fn::C], sd, ln_sd);
    }
};
struct mp_nested_flip_fn {
    int unused;
    template <class H, class T>
    MP_HD void operator()(H& g, const T& tr) const {
        // (a move that switches `d` on proposes it as well — and its reverse then finds the dropped `d` in the discard it assesses: a
        // proposal that left it out would be the reference's "not all constraints were consumed" panic, as for hierarchical.rs:60-70)
        if (g.template bernoulli<mp_nested_fn::F>(tr.val[mp_nested_fn::F] != 0. ? 0.2 : 0.8))
            g.template normal<mp_nested_fn::D>(tr.val[mp_nested_fn::C], 1., 0.);
    }
};
struct mp_nested_drift_a_fn {
    double sd, ln_sd;
    template <class H, class T>
    MP_HD void operator()(H& g, const T& tr) const {
        g.template normal<mp_nested_fn::A>(tr.val[mp_nested_fn::A], sd, ln_sd);
    }
};
template <class P>
inline bool mp_parse_nested_drift(const double* args, int n_args, P& p, std::string& err) {
    if (!args || n_args != 1 || !(args[0] > 0.)) { err = "drift proposal takes {std > 0}"; return false; }
    p.sd = args[0];
    p.ln_sd = mp_log(args[0]);
    return true;
}
inline bool mp_parse_nested_flip(const double*, int n_args, mp_nested_flip_fn& p, std::string& err) {
    if (n_args != 0) { err = "flip proposal takes no arguments"; return false; }
    p.unused = 0;
    return true;
}
MP_REGISTER_MH_PROPOSAL(1, mp_nested_fn, mp_nested_drift_bc_fn, mp_parse_nested_drift<mp_nested_drift_bc_fn>)
MP_REGISTER_MH_PROPOSAL(2, mp_nested_fn, mp_nested_flip_fn, mp_parse_nested_flip)
MP_REGISTER_MH_PROPOSAL(3, mp_nested_fn, mp_nested_drift_a_fn, mp_parse_nested_drift<mp_nested_drift_a_fn>)

// ---------------------------------------------------------------------------------------
// More than 32 sites (kind 114): the reference's traces are tries and have no size limit; here a trace is a dense row with one
// presence bit per site, 32 bits wide up to 32 sites and 64 beyond (mp_genfn.h mp_fn_bits_t; two 32-bit words per chain through
// the C ABI).  A line with 30 observations, then — at site ids ABOVE 32 — a sub-call whose second choice comes and goes and six more
// observations that depend on it: constraints, masks, proposals, the discard, gc and a sub-trie's running weight all live in the high word.
//   slope ~ normal(0, 2) %= "slope";  intercept ~ normal(0, 2) %= "intercept";  big ~ bernoulli(0.3) %= "big"
//   y_j ~ normal(slope x_j + intercept, big ? 2 : 0.5) %= ("y", j),  j < 30
//   off = offsets() /= "off":   o1 ~ normal(0, 1) %= "o1";  if big { o2 ~ normal(o1, 1) %= "o2" };  return o1 + o2
//   z_k ~ normal(off + 0.1 k, 0.5) %= ("z", k),  k < 6
//   params = xs[0 .. 30); the observations are constraints on Y0 + j and Z0 + k.
// Moves: proposal 1 = drift of slope, intercept and o1 {std}; proposal 2 = toggle of `big` (proposes the other value with
// probability 0.8, and o2 ~ normal(o1, 1) with a true `big`: its reverse finds the dropped o2 in the discard it assesses).
// ---------------------------------------------------------------------------------------
struct mp_wide_fn {
    static constexpr int N_Y = 30, N_Z = 6;
    enum { SLOPE = 0, INTERCEPT = 1, BIG = 2, Y0 = 3, O1 = Y0 + N_Y, O2 = O1 + 1, Z0 = O2 + 1 };
    static constexpr int NS = Z0 + N_Z;   // 41
    static constexpr uint64_t OFF = (uint64_t(1) << O1) | (uint64_t(1) << O2);
    static constexpr uint64_t sub_of(int site) { return (site == O1 || site == O2) ? OFF : uint64_t(0); }
    static constexpr bool is_bool(int site) { return site == BIG; }
    double xs[N_Y];
    double ln2, ln_half;

    template <class H, int J>
    MP_HD void ys(H& g, bool big, double slope, double intercept) const {
        g.template normal<Y0 + J>(slope * xs[J] + intercept, big ? 2. : 0.5, big ? ln2 : ln_half);
        if constexpr (J + 1 < N_Y) ys<H, J + 1>(g, big, slope, intercept);
    }
    template <class H, int K>
    MP_HD void zs(H& g, double off) const {
        g.template normal<Z0 + K>(off + 0.1 * (double)K, 0.5, ln_half);
        if constexpr (K + 1 < N_Z) zs<H, K + 1>(g, off);
    }
    template <class H>
    MP_HD void operator()(H& g) const {
        const double slope = g.template normal<SLOPE>(0., 2., ln2);
        const double intercept = g.template normal<INTERCEPT>(0., 2., ln2);
        const bool big = g.template bernoulli<BIG>(0.3);
        ys<H, 0>(g, big, slope, intercept);
        const mp_fn_ret off = g.template call<OFF>([&](H& q) {
            mp_fn_ret r{};
            const double o1 = q.template normal<O1>(0., 1., 0.);
            r.v[0] = o1;
            if (big) r.v[0] = o1 + q.template normal<O2>(o1, 1., 0.);
            return r;
        });
        zs<H, 0>(g, off.v[0]);
    }
};
inline bool mp_parse_wide_fn(const double* params, int n_params, mp_wide_fn& m, std::string& err) {
    if (!params || n_params != mp_wide_fn::N_Y) { err = "wide model: params = xs[0 .. 30)"; return false; }
    for (int k = 0; k < mp_wide_fn::N_Y; ++k) m.xs[k] = params[k];
    m.ln2 = mp_log(2.); m.ln_half = mp_log(0.5);
    return true;
}
MP_REGISTER_MH_MODEL(114, mp_wide_fn, mp_parse_wide_fn)

struct mp_wide_drift_fn {
    double sd, ln_sd;
    template <class H, class T>
    MP_HD void operator()(H& g, const T& tr) const {
        g.template normal<mp_wide_fn::SLOPE>(tr.val[mp_wide_fn::SLOPE], sd, ln_sd);
        g.template normal<mp_wide_fn::INTERCEPT>(tr.val[mp_wide_fn::INTERCEPT], sd, ln_sd);
        g.template normal<mp_wide_fn::O1>(tr.val[mp_wide_fn::O1], sd, ln_sd);
    }
};
struct mp_wide_toggle_fn {
    int unused;
    template <class H, class T>
    MP_HD void operator()(H& g, const T& tr) const {
        if (g.template bernoulli<mp_wide_fn::BIG>(tr.val[mp_wide_fn::BIG] != 0. ? 0.2 : 0.8))
            g.template normal<mp_wide_fn::O2>(tr.val[mp_wide_fn::O1], 1., 0.);
    }
};
inline bool mp_parse_wide_drift_fn(const double* args, int n_args, mp_wide_drift_fn& p, std::string& err) {
    if (!args || n_args != 1 || !(args[0] > 0.)) { err = "drift proposal takes {std > 0}"; return false; }
    p.sd = args[0];
    p.ln_sd = mp_log(args[0]);
    return true;
}
inline bool mp_parse_wide_toggle_fn(const double*, int n_args, mp_wide_toggle_fn& p, std::string& err) {
    if (n_args != 0) { err = "toggle proposal takes no arguments"; return false; }
    p.unused = 0;
    return true;
}
MP_REGISTER_MH_PROPOSAL(1, mp_wide_fn, mp_wide_drift_fn, mp_parse_wide_drift_fn)
MP_REGISTER_MH_PROPOSAL(2, mp_wide_fn, mp_wide_toggle_fn, mp_parse_wide_toggle_fn)

// ---------------------------------------------------------------------------------------
// The reference's own Update regression functions (modppl/tests/dyngenfn.rs:30-53), kinds 110 - 112: what its known-answer
// tests for `update` run (:55-114: -0.5, -2.517551, 0.4, -1.098612 twice) — here so that the same calls can be made on the
// device through mp_fn_update (tests/test_gpu_gfi.py).  No params, no proposals.
//   110  b ~ bernoulli(0.25) %= "b"; if b { normal(0, 1) %= "x" }
//   111  m ~ uniform(0, 1) %= "m"; normal(m, 1) %= "x"; normal(m, 1) %= "y"
//   112  b ~ bernoulli(0.25) %= "b"; if b { prototype(1.0) /= "sub" }   with prototype = normal(1, noise) at three addresses (the
//        reference's has 2999: the sub-call is new and unconstrained in the test, so its size does not enter the weight)
// ---------------------------------------------------------------------------------------
struct mp_kat_bx_fn {
    static constexpr int NS = 2;
    enum { B = 0, X = 1 };
    static constexpr uint32_t sub_of(int) { return 0u; }
    static constexpr bool is_bool(int site) { return site == B; }
    int unused;
    template <class H>
    MP_HD void operator()(H& g) const {
        if (g.template bernoulli<B>(0.25)) g.template normal<X>(0., 1., 0.);
    }
};
struct mp_kat_mxy_fn {
    static constexpr int NS = 3;
    enum { M = 0, X = 1, Y = 2 };
    static constexpr uint32_t sub_of(int) { return 0u; }
    static constexpr bool is_bool(int) { return false; }
    int unused;
    template <class H>
    MP_HD void operator()(H& g) const {
        const double m = g.template uniform<M>(0., 1.);
        g.template normal<X>(m, 1., 0.);
        g.template normal<Y>(m, 1., 0.);
    }
};
struct mp_kat_bsub_fn {
    static constexpr int NS = 4;
    enum { B = 0, S1 = 1, S2 = 2, S3 = 3 };
    static constexpr uint32_t SUB = (1u << S1) | (1u << S2) | (1u << S3);
    static constexpr uint32_t sub_of(int site) { return site >= S1 ? SUB : 0u; }
    static constexpr bool is_bool(int site) { return site == B; }
    int unused;
    template <class H>
    MP_HD void operator()(H& g) const {
        if (g.template bernoulli<B>(0.25)) {
            (void)g.template call<SUB>([&](H& q) {
                mp_fn_ret r{};
                r.v[0] = q.template normal<S1>(1., 1., 0.) + q.template normal<S2>(1., 1., 0.) + q.template normal<S3>(1., 1., 0.);
                return r;
            });
        }
    }
};
template <class M>
inline bool mp_parse_kat_fn(const double*, int n_params, M& m, std::string& err) {
    if (n_params != 0) { err = "the reference's regression functions take no parameters"; return false; }
    m.unused = 0;
    return true;
}
MP_REGISTER_MH_MODEL(110, mp_kat_bx_fn, mp_parse_kat_fn<mp_kat_bx_fn>)
MP_REGISTER_MH_MODEL(111, mp_kat_mxy_fn, mp_parse_kat_fn<mp_kat_mxy_fn>)
MP_REGISTER_MH_MODEL(112, mp_kat_bsub_fn, mp_parse_kat_fn<mp_kat_bsub_fn>)

// ---------------------------------------------------------------------------------------
// pointed_2d_model + pointed_2d_drift_proposal (modppl/tests/dyngenfns/simple.rs:27-41; driven by tests/mh.rs:50-68), kind 120 —
// vector-valued sites through the generic layer: the functor form of the model the hand-written k_pointed_iterate restates
// (tests/test_gpu_mh.py holds the two, and the checker's own restatement, to the same bits).
//   latent ~ uniform_2d(bounds) %= "latent";  obs ~ mvnormal(latent, cov) %= "obs"
//   params = {xmin, xmax, ymin, ymax, cov row-major[4]}; the observation is a constraint on OBS (two values: slots OBS, OBS + 1).
//   proposal 1 (args = noise covariance row-major[4]):  mvnormal(tr["latent"], noise) %= "latent"
// Site ids: LATENT = 1 as in the hand-written kernel and the checker's restatement (slots 1, 2), OBS = 3 (slots 3, 4).
// ---------------------------------------------------------------------------------------
struct mp_pointed_fn {
    static constexpr int NS = 5;
    enum { LATENT = 1, OBS = 3 };
    static constexpr uint32_t sub_of(int) { return 0u; }
    static constexpr bool is_bool(int) { return false; }
    static constexpr int dim_of(int site) { return (site == LATENT || site == OBS) ? 2 : ((site == LATENT + 1 || site == OBS + 1) ? 0 : 1); }   // 0: a vector's further slot
    double xmin, xmax, ymin, ymax, neg_ln_area;
    double cov[4], cov_inv[4], chol[4], ln_det;
    template <class H>
    MP_HD void operator()(H& g) const {
        double latent[2], obs[2];
        g.template uniform_2d<LATENT>(xmin, xmax, ymin, ymax, neg_ln_area, latent);
        g.template mvnormal2<OBS>(latent, cov, cov_inv, ln_det, chol, obs);
    }
};
// the hoisted constants of a 2 x 2 covariance: inverse, ln det, lower Cholesky factor (zeros where there is none)
inline bool mp_pointed_cov(const double* c4, double* cov, double* inv, double* chol, double* ln_det, std::string& err) {
    const std::vector<double> c(c4, c4 + 4);
    std::vector<double> iv, L;
    const double det = mp_host_det(c, 2);
    if (!(det > 0.) || !mp_host_inverse(c, 2, iv)) { err = "covariance must be invertible with a positive determinant"; return false; }
    if (!mp_host_cholesky(c, 2, L)) { err = "covariance without a Cholesky factor"; return false; }
    for (int q = 0; q < 4; ++q) { cov[q] = c[q]; inv[q] = iv[q]; chol[q] = L[q]; }
    *ln_det = mp_log(det);
    return true;
}
inline bool mp_parse_pointed_fn(const double* params, int n_params, mp_pointed_fn& m, std::string& err) {
    if (!params || n_params != 8) { err = "pointed model: params = {xmin, xmax, ymin, ymax, cov row-major[4]}"; return false; }
    if (!(params[1] > params[0]) || !(params[3] > params[2])) { err = "pointed model: xmax > xmin and ymax > ymin"; return false; }
    m.xmin = params[0]; m.xmax = params[1]; m.ymin = params[2]; m.ymax = params[3];
    m.neg_ln_area = -mp_log((m.xmax - m.xmin) * (m.ymax - m.ymin));
    return mp_pointed_cov(params + 4, m.cov, m.cov_inv, m.chol, &m.ln_det, err);
}
MP_REGISTER_MH_MODEL(120, mp_pointed_fn, mp_parse_pointed_fn)

struct mp_pointed_drift_fn {
    double cov[4], cov_inv[4], chol[4], ln_det;
    template <class H, class T>
    MP_HD void operator()(H& g, const T& tr) const {
        const double prev[2] = {tr.val[mp_pointed_fn::LATENT], tr.val[mp_pointed_fn::LATENT + 1]};
        double out[2];
        g.template mvnormal2<mp_pointed_fn::LATENT>(prev, cov, cov_inv, ln_det, chol, out);
    }
};
inline bool mp_parse_pointed_drift_fn(const double* args, int n_args, mp_pointed_drift_fn& p, std::string& err) {
    if (!args || n_args != 4) { err = "pointed drift proposal takes the noise covariance, row-major[4]"; return false; }
    return mp_pointed_cov(args, p.cov, p.cov_inv, p.chol, &p.ln_det, err);
}
MP_REGISTER_MH_PROPOSAL(1, mp_pointed_fn, mp_pointed_drift_fn, mp_parse_pointed_drift_fn)
