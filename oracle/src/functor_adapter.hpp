// ORACLE — TEST INFRASTRUCTURE ONLY (see rng.hpp).
//
// functor_adapter.hpp — runs a model functor of modppl_amd/csrc/mp_models.h (the ONE source of a model added through
// MP_REGISTER_UNFOLD_MODEL) against the checker's OWN interpreters, so that a new model needs no hand-written restatement
// here while the handler semantics and the distributions under test stay independent of the product:
//   * SoaFunctorModel<M>   the flat-array engine (soa.hpp): constrained site -> this file scores it with dists.hpp's
//                          logpdf and adds to the weight; free site -> dists.hpp's sampler on the checker's Rng
//                          (dyngenfn.rs:115-141, Generate arm), exactly as the hand-inlined kernels of soa.hpp do;
//   * make_functor_unfold  a DynUnfold whose kernel body calls DynGenFnHandler::sample_at (dyngenfn.hpp, the restated
//                          dyngenfn.rs:100-275 with its trie bookkeeping) once per site of the functor: `g.normal<SITE>(mu, sd)`
//                          becomes `sample_at(normal, {mu, sd}, "s<SITE>")` — what the dyngen! macro would have expanded to
//                          (modppl-macros/src/lib.rs:20-113), in all four handler modes.
// What is shared with the product is the model BODY (which sites, in which order, with which parameters); what is not:
// Philox (rng.hpp), the distributions (dists.hpp), the weight rules (dyngenfn.hpp), the arithmetic mode (o_exp / o_ln).
// The hand-written models of models.hpp / soa.hpp stay as they are: they restate the REFERENCE's test models and are the
// cross-check of this adapter (tests/test_oracle_functor.py).
#pragma once
#include <functional>
#include <map>

#include "soa.hpp"

namespace oracle {

// ---- flat-array interpretation -------------------------------------------------------------------------------------
template <class M>
struct SoaFunctorHandler {
    Rng& r;
    const double* obs;
    double weight = 0.;
    SoaFunctorHandler(Rng& r_, const double* o) : r(r_), obs(o) {}
    double exp_(double x) const { return o_exp(x); }
    double log_(double x) const { return o_ln(x); }
    double sin_(double x) const { return o_sin(x); }
    double cos_(double x) const { return o_cos(x); }
    double atan2_(double y, double x) const { return o_atan2(y, x); }
    template <int SITE>
    double normal(double mu, double sd) {
        constexpr int k = M::obs_of(SITE);
        if constexpr (k >= 0) { weight += oracle::normal.logpdf(obs[k], {mu, sd}); return obs[k]; }
        else { r.at(DOM_MODEL, (uint32_t)SITE); return oracle::normal.random(r, {mu, sd}); }
    }
    template <int SITE>
    double normal(double mu, double sd, double /*ln_sd hoisted by the device form*/) { return normal<SITE>(mu, sd); }
    template <int SITE>
    double uniform(double a, double b) {
        constexpr int k = M::obs_of(SITE);
        if constexpr (k >= 0) { weight += oracle::uniform.logpdf(obs[k], {a, b}); return obs[k]; }
        else { r.at(DOM_MODEL, (uint32_t)SITE); return oracle::uniform.random(r, {a, b}); }
    }
    template <int SITE>
    int categorical(const double* probs, int n) {
        constexpr int k = M::obs_of(SITE);
        const Vec p(probs, probs + n);
        if constexpr (k >= 0) { weight += oracle::categorical.logpdf((int64_t)obs[k], p); return (int)obs[k]; }
        else { r.at(DOM_MODEL, (uint32_t)SITE); return (int)oracle::categorical.random(r, p); }
    }
    // mvnormal site of dimension K (mvnormal.rs:12-37): the checker works from the covariance itself — determinant, inverse and
    // Cholesky / eigen transform per call, as the reference does — and ignores the constants the device form hoists
    template <int SITE, int K>
    void mvnormal_observed(const double* mu, const double*, double, const double*, const double* cov) {
        constexpr int k = M::obs_of(SITE);
        static_assert(k >= 0, "mvnormal_observed: constrained on the Generate path");
        if (!cov) throw Panic("functor adapter: the functor must pass the covariance of its mvnormal site");
        weight += oracle::mvnormal.logpdf(Vec(obs + k, obs + k + K), MvNormalParams{Vec(mu, mu + K), Mat(K, Vec(cov, cov + K * K))});
    }
};
template <class M>
struct SoaFunctorModel : SoaModel {
    M m;
    explicit SoaFunctorModel(const M& m_) : m(m_) { dim_state = M::DIM_STATE; dim_obs = M::DIM_OBS; }
    double kernel(Rng& r, int64_t t, const double* prev, double* next, const double* obs) const override {
        SoaFunctorHandler<M> g(r, obs);
        m(g, t, prev, next);
        return g.weight;
    }
};

// ---- dynamic (trie) interpretation ----------------------------------------------------------------------------------
inline std::string functor_addr(int site) { return "s" + std::to_string(site); }
inline uint32_t functor_site_of(const std::string& a) { return (uint32_t)std::stoul(a.substr(1)); }
template <class M>
struct DynFunctorHandler {
    using A = std::pair<int64_t, Vec>;
    DynGenFnHandler<A, Vec>& g;
    explicit DynFunctorHandler(DynGenFnHandler<A, Vec>& g_) : g(g_) {}
    double exp_(double x) const { return o_exp(x); }
    double log_(double x) const { return o_ln(x); }
    double sin_(double x) const { return o_sin(x); }
    double cos_(double x) const { return o_cos(x); }
    double atan2_(double y, double x) const { return o_atan2(y, x); }
    template <int SITE>
    double normal(double mu, double sd) { return g.template sample_at<double>(oracle::normal, NormalParams{mu, sd}, functor_addr(SITE)); }
    template <int SITE>
    double normal(double mu, double sd, double) { return normal<SITE>(mu, sd); }
    template <int SITE>
    double uniform(double a, double b) { return g.template sample_at<double>(oracle::uniform, UniformParams{a, b}, functor_addr(SITE)); }
    template <int SITE>
    int categorical(const double* probs, int n) { return (int)g.template sample_at<int64_t>(oracle::categorical, Vec(probs, probs + n), functor_addr(SITE)); }
    template <int SITE, int K>
    void mvnormal_observed(const double* mu, const double*, double, const double*, const double* cov) {
        if (!cov) throw Panic("functor adapter: the functor must pass the covariance of its mvnormal site");
        (void)g.template sample_at<Vec>(oracle::mvnormal, MvNormalParams{Vec(mu, mu + K), Mat(K, Vec(cov, cov + K * K))}, functor_addr(SITE));
    }
};
template <class M>
DynUnfold<Vec> make_functor_unfold(const M& m) {
    using A = std::pair<int64_t, Vec>;
    using H = DynGenFnHandler<A, Vec>;
    DynGenFn<A, Vec> k(
        [m](H& g, A ta) -> Vec {
            DynFunctorHandler<M> h(g);
            Vec prev = ta.second, next((size_t)M::DIM_STATE, 0.);
            prev.resize((size_t)M::DIM_STATE, 0.);
            m(h, ta.first, prev.data(), next.data());
            return next;
        },
        functor_site_of);
    return DynUnfold<Vec>(std::move(k));
}
// constraints of one time step for the functor's observed sites: obs slot k belongs to the site with obs_of(site) == k
template <class M, class = void>
struct functor_obs_dim { static constexpr int of(int) { return 1; } };
template <class M>
struct functor_obs_dim<M, std::void_t<decltype(M::obs_dim(0))>> { static constexpr int of(int site) { return M::obs_dim(site); } };
template <class M>
DynTrie functor_constraints(const double* y) {
    DynTrie c;
    for (int site = 0; site < 64; ++site) {
        const int k = M::obs_of(site);
        if (k < 0 || k >= M::DIM_OBS) continue;
        const int dim = functor_obs_dim<M>::of(site);   // a vector-valued site (mvnormal) is constrained with a Vec
        if (dim == 1) c.observe(functor_addr(site), arc(y[k]));
        else c.observe(functor_addr(site), arc(Vec(y + k, y + k + dim)));
    }
    return c;
}

// ---- registry: kind -> the two interpretations ------------------------------------------------------------------------
struct FunctorEntry {
    int dim_state, dim_obs;
    std::function<std::unique_ptr<SoaModel>(const mp_model_desc&)> soa;
    std::function<DynUnfold<Vec>(const mp_model_desc&)> dyn;
    std::function<DynTrie(const double*)> constraints;
};
inline std::map<int, FunctorEntry>& functor_registry() {
    static std::map<int, FunctorEntry> r;
    return r;
}
template <class M>
int register_functor_model(int kind, bool (*parse)(const mp_model_desc&, M&, std::string&)) {
    auto build = [parse](const mp_model_desc& d) {
        M k{};
        std::string err;
        if (!parse(d, k, err)) throw Panic(err);
        return k;
    };
    FunctorEntry e;
    e.dim_state = M::DIM_STATE; e.dim_obs = M::DIM_OBS;
    e.soa = [build](const mp_model_desc& d) -> std::unique_ptr<SoaModel> { return std::make_unique<SoaFunctorModel<M>>(build(d)); };
    e.dyn = [build](const mp_model_desc& d) { return make_functor_unfold<M>(build(d)); };
    e.constraints = [](const double* y) { return functor_constraints<M>(y); };
    functor_registry()[kind] = std::move(e);
    return kind;
}

}  // namespace oracle

// the product's model sources, interpreted by the registrar above (mp_models.h, MP_REGISTER_UNFOLD_MODEL)
#define MP_MODEL_REGISTRAR oracle::register_functor_model
#include "../../modppl_amd/csrc/mp_models.h"

// ---- cross-check of the adapter: the product's OWN functors for models that models.hpp / soa.hpp restate by hand, under test-only
// kinds 1001.. (tests/test_oracle_functor.py runs kind k + 1000 against the hand-written kind k, both engines, both arithmetic modes)
namespace oracle {
inline bool parse_lgssm1_functor(const mp_model_desc& m, mp_lgssm1& k, std::string& err) {
    if (m.n_params != 5) { err = "lgssm1: 5 params"; return false; }
    k = mp_lgssm1{m.params[0], m.params[1], m.params[2], m.params[3], m.params[4], 0.};
    return true;
}
inline bool parse_bearings_functor(const mp_model_desc& m, mp_bearings& k, std::string& err) {
    if (m.n_params != 6) { err = "bearings: 6 params"; return false; }
    k = mp_bearings{m.params[0], m.params[1], m.params[2], m.params[3], m.params[4], m.params[5], 0.};
    return true;
}
inline bool parse_band4_functor(const mp_model_desc& m, mp_lgssm_band<4>& k, std::string& err) {
    if (m.n_params != 6 || (int)m.params[0] != 4) { err = "lgssm_band<4>: {4, a, band, sig0, sig_x, sig_y}"; return false; }
    k = mp_lgssm_band<4>{m.params[1], m.params[2], m.params[3], m.params[4], m.params[5], 0.};
    return true;
}
inline bool parse_spiral_functor(const mp_model_desc&, mp_spiral& k, std::string&) {
    // only the covariance matters here (unfold.rs:29): the checker's mvnormal derives everything else per call
    k = mp_spiral{};
    k.cov[0] = 0.001; k.cov[1] = 0.; k.cov[2] = 0.; k.cov[3] = 0.001;
    return true;
}
static const int functor_check_2 = register_functor_model<mp_spiral>(1000 + MP_MODEL_SPIRAL, parse_spiral_functor);
static const int functor_check_1 = register_functor_model<mp_lgssm1>(1000 + MP_MODEL_LGSSM1, parse_lgssm1_functor);
static const int functor_check_4 = register_functor_model<mp_bearings>(1000 + MP_MODEL_BEARINGS, parse_bearings_functor);
static const int functor_check_5 = register_functor_model<mp_lgssm_band<4>>(1000 + MP_MODEL_LGSSM_BAND, parse_band4_functor);
}  // namespace oracle
