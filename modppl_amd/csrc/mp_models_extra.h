// mp_models_extra.h — models added through the registration layer: ONE file per model, nothing else to touch.
//
// A model here is (1) a functor written once against the handler interface (the static stand-in for a `dyngen!` body,
// modppl-macros/src/lib.rs:20-113: `dist(args) %= addr` becomes `g.template dist<SITE>(args)`), (2) a host function that
// fills it from a C-ABI descriptor, and (3) one MP_REGISTER_UNFOLD_MODEL line.  The same source is then
//   * compiled for the device by mp_pf.hip (k_propagate / k_simulate / the resample kernels are instantiated for it:
//     Generate and Simulate interpretations), and
//   * interpreted by the CPU checker (oracle/src/functor_adapter.hpp), which runs the SAME functor against its own,
//     independent handlers — the flat-array engine and the dynamic trie handler `DynGenFnHandler::sample_at` — and its
//     own distributions, so the parity tests cover the new model with no hand-written restatement.
// Restrictions of the adapter path: scalar sites (normal, uniform, categorical); transcendental functions inside the
// functor go through the handler (`g.exp_(x)`: mp_exp on the device, the checker's libm / canonical switch there).
// Kinds >= 100 are free for such models; Python reaches them with modppl_amd.UnfoldModel(kind, dim_state, dim_obs, params, name).
#pragma once

// ---------------------------------------------------------------------------------------
// Stochastic volatility (the standard nonlinear SMC benchmark), kind 100, dim_state = dim_obs = 1:
//   t==0: h ~ normal(mu, sig0) %= "h";   t>0: h ~ normal(mu + phi (h_prev - mu), sigma) %= "h";   normal(0, exp(h / 2)) %= "y" observed
// params = {mu, phi, sigma, sig0}
// ---------------------------------------------------------------------------------------
struct mp_stochvol {
    static constexpr int DIM_STATE = 1, DIM_OBS = 1;
    enum { H = 0, Y = 1 };
    static constexpr int obs_of(int site) { return site == Y ? 0 : -1; }
    static constexpr int MAX_NORMALS = 1;
    static constexpr int normal_index(int site) { return site == H ? 0 : -1; }
    MP_HD int n_normals(int64_t) const { return 1; }
    MP_HD uint32_t normal_site(int) const { return H; }
    double mu, phi, sigma, sig0;

    template <class G>
    MP_HD void operator()(G& g, int64_t t, const double* prev, double* next) const {
        double h;
        if (t == 0) h = g.template normal<H>(mu, sig0);
        else h = g.template normal<H>(mu + phi * (prev[0] - mu), sigma);
        g.template normal<Y>(0., g.exp_(h * 0.5));
        next[0] = h;
    }
};
inline bool mp_parse_stochvol(const mp_model_desc& m, mp_stochvol& k, std::string& err) {
    if (m.n_params != 4 || !m.params || m.dim_state != 1 || m.dim_obs != 1) { err = "stochastic volatility: params = {mu, phi, sigma, sig0}, dim_state = dim_obs = 1"; return false; }
    k.mu = m.params[0]; k.phi = m.params[1]; k.sigma = m.params[2]; k.sig0 = m.params[3];
    if (!(k.sigma > 0.) || !(k.sig0 > 0.)) { err = "stochastic volatility: standard deviations must be > 0"; return false; }
    return true;
}
MP_REGISTER_UNFOLD_MODEL(100, mp_stochvol, mp_parse_stochvol)
