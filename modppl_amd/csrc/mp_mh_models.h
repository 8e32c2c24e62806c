// mp_mh_models.h — models and proposals for mh / regen_mh written ONCE against the handler interface of mp_genfn.h
// (the static stand-in for `dyngen!` bodies) and registered by one line each; nothing else to touch.
//
// A model is a functor `template <class H> void operator()(H& g) const` with `static constexpr int NS` sites; a proposal is
// `template <class H, class T> void operator()(H& g, const T& tr) const` over the SAME site ids (a proposal's addresses are the
// model's: mh.rs:17-23 feeds its choices to model.update as constraints).  Both are trivially copyable (they travel to the
// kernels by value) and are filled from the C ABI's double arrays by a parse function.
//   MP_REGISTER_MH_MODEL(kind, Type, parse)                          kinds >= 100 are free
//   MP_REGISTER_MH_PROPOSAL(proposal_kind, ModelType, Type, parse)
// mp_mh.hip instantiates k_fn_init / k_fn_regen / k_fn_mh / k_fn_logjp for every registration; chains are created with
// mp_mh_create_fn and driven by the same mp_mh_step / mp_regen_mh_step as the hand-written kernels.
#pragma once
#include <string>

#include "mp_genfn.h"

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
        double a = 0., b = 0., c = 0.;
        // linear() / quadratic() /= "coeffs" (:17-30, :37, :42)
        g.template call<COEFFS>([&](H& q) {
            a = q.template normal<A>(0., 1., 0.);
            b = q.template normal<B>(0., 1., 0.);
            if (!lin) c = q.template normal<C>(0., 1., 0.);
            return 0;
        });
        ys<H, 0>(g, lin, a, b, c);
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
