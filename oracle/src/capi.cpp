// ORACLE — TEST INFRASTRUCTURE ONLY (see rng.hpp).
//
// capi.cpp — C entry points so that tests/ (ctypes) can drive the restated reference.  The
// particle-filter entry points mirror include/modppl_hip.h one for one with an `oracle_`
// prefix, so a parity test issues the same call sequence to both libraries.
//
// variant bits of oracle_pf_create:
//   1 = canonical (mp_math transcendental functions + fixed-point resampling CDF)
//   2 = SoA "algorithm-faithful" engine instead of the dynamic-handler "structure-faithful" one
//   4 = (literal, dynamic engine only) binary-search the sequential running sum instead of the
//       O(N) scan per draw — index-identical, makes N >= 10^4 feasible
#include <cstdio>
#include <cstring>
#include <memory>

#include "../../include/modppl_hip.h"
#include "soa.hpp"
#include "functor_adapter.hpp"
#include "mh_functor_adapter.hpp"
#include "../../modppl_amd/csrc/mp_binomial.h"

using namespace oracle;

static thread_local std::string g_err;
static int32_t fail(int32_t code, const std::string& msg) { g_err = msg; return code; }
static int32_t classify(const Panic& p) {
    const std::string m = p.what();
    if (m.find("constraints") != std::string::npos) return MP_ERR_CONSTRAINTS;
    if (m.find("-inf") != std::string::npos || m.find("sum to 1") != std::string::npos) return MP_ERR_DEGENERATE;
    if (m.find("before init_step") != std::string::npos || m.find("twice") != std::string::npos) return MP_ERR_STATE;
    return MP_ERR_INVALID_ARG;
}
#define GUARD(...)                                               \
    try { __VA_ARGS__; return MP_OK; }                           \
    catch (const Panic& p) { return fail(classify(p), p.what()); } \
    catch (const std::exception& e) { return fail(MP_ERR_INVALID_ARG, e.what()); }

struct IPf {
    virtual ~IPf() {}
    virtual void init_step(const double* args0, const double* obs, int n_steps) = 0;
    virtual void step(const double* obs, int n_steps) = 0;
    virtual double ess(int mode) = 0;
    virtual double resample() = 0;
    virtual void set_scheme(int s_) { if (s_ != MP_RESAMPLE_MULTINOMIAL) throw Panic("oracle: only multinomial here (the reference has no other); systematic lives in the canonical SoA engine"); }
    virtual double log_ml() = 0;
    virtual void read_state(double* out) = 0;
    virtual void read_logw(double* out) = 0;
    virtual void read_parents(uint32_t* out) = 0;
    virtual int64_t time() = 0;
    virtual size_t trajectory(uint64_t i, double* out) { (void)i; (void)out; throw Panic("trajectory: not recorded by this engine"); }
};

// ---- structure-faithful engine: ParticleSystem over DynUnfold ---------------------------------
template <class State>
struct DynPf : IPf {
    using PS = ParticleSystem<State, std::vector<DynTrie>, std::vector<State>>;
    DynUnfold<State> model;
    std::unique_ptr<PS> ps;
    int dim_state, dim_obs;
    std::function<DynTrie(const double*)> mk_constraints;
    std::function<State(const double*)> mk_state;
    std::function<void(const State&, double*)> put_state;
    bool initialised = false;

    std::vector<DynTrie> constraints(const double* obs, int n_steps) const {
        std::vector<DynTrie> v;
        for (int k = 0; k < n_steps; ++k) v.push_back(mk_constraints(obs + (size_t)k * dim_obs));
        return v;
    }
    void init_step(const double* args0, const double* obs, int n_steps) override {
        if (initialised) throw Panic("init_step called twice");
        std::vector<double> z((size_t)dim_state, 0.);
        // the reference always passes (1, args) to generate (particle_filter.rs:66); more than one
        // constraint is the build's multi-step extension: generate at t=0, then Extend.
        ps->init_step(mk_state(args0 ? args0 : z.data()), constraints(obs, 1));
        initialised = true;
        if (n_steps > 1) step(obs + dim_obs, n_steps - 1);
    }
    void step(const double* obs, int n_steps) override {
        if (!initialised) throw Panic("step before init_step");
        for (int k = 0; k < n_steps; ++k) ps->step(constraints(obs + (size_t)k * dim_obs, 1));
    }
    double ess(int mode) override {
        if (mode == MP_ESS_REFERENCE) return ps->effective_sample_size();
        if (ps->canonical_resampling) return canonical_normalize(ps->log_weights, ps->num_particles).c.ess;
        const double L = logsumexp(ps->log_weights);
        std::vector<double> two;
        for (double w : ps->log_weights) two.push_back(2.0 * (w - L));
        return o_exp(-logsumexp(two));
    }
    double resample() override { return ps->resample(); }
    double log_ml() override { return ps->log_marginal_likelihood_estimate(); }
    void read_state(double* out) override {
        for (size_t i = 0; i < ps->traces.size(); ++i) put_state(ps->traces[i].retv->back(), out + i * (size_t)dim_state);
    }
    void read_logw(double* out) override { std::memcpy(out, ps->log_weights.data(), ps->num_particles * sizeof(double)); }
    void read_parents(uint32_t* out) override { for (size_t i = 0; i < ps->num_particles; ++i) out[i] = (uint32_t)ps->parents[i]; }
    int64_t time() override { return ps->traces.empty() ? 0 : ps->traces[0].args.first; }
    size_t trajectory(uint64_t i, double* out) override {  // traces[i].retv: Vec<State>
        const auto& rv = *ps->traces.at(i).retv;
        for (size_t t = 0; t < rv.size(); ++t) put_state(rv[t], out + t * (size_t)dim_state);
        return rv.size();
    }
};

struct HmmPf : IPf {
    using PS = ParticleSystem<int, HmmData, std::vector<size_t>>;
    HMM model;
    std::unique_ptr<PS> ps;
    bool initialised = false;
    explicit HmmPf(HmmParams p) : model(std::move(p)) {}
    static HmmData mk(const double* obs) { return HmmData{{std::nullopt}, {(size_t)obs[0]}}; }
    void init_step(const double*, const double* obs, int n_steps) override {
        if (initialised) throw Panic("init_step called twice");
        ps->init_step(0, mk(obs));
        initialised = true;
        if (n_steps > 1) step(obs + 1, n_steps - 1);
    }
    void step(const double* obs, int n_steps) override {
        if (!initialised) throw Panic("step before init_step");
        for (int k = 0; k < n_steps; ++k) ps->step(mk(obs + k));
    }
    double ess(int mode) override {
        if (mode == MP_ESS_REFERENCE) return ps->effective_sample_size();
        if (ps->canonical_resampling) return canonical_normalize(ps->log_weights, ps->num_particles).c.ess;
        const double L = logsumexp(ps->log_weights);
        std::vector<double> two;
        for (double w : ps->log_weights) two.push_back(2.0 * (w - L));
        return o_exp(-logsumexp(two));
    }
    double resample() override { return ps->resample(); }
    double log_ml() override { return ps->log_marginal_likelihood_estimate(); }
    void read_state(double* out) override { for (size_t i = 0; i < ps->traces.size(); ++i) out[i] = (double)*ps->traces[i].data.first.back(); }
    void read_logw(double* out) override { std::memcpy(out, ps->log_weights.data(), ps->num_particles * sizeof(double)); }
    void read_parents(uint32_t* out) override { for (size_t i = 0; i < ps->num_particles; ++i) out[i] = (uint32_t)ps->parents[i]; }
    int64_t time() override { return ps->traces.empty() ? 0 : (int64_t)ps->traces[0].data.first.size(); }
};

// ---- algorithm-faithful engine -----------------------------------------------------------------
struct SoaEngine : IPf {
    std::unique_ptr<SoaModel> model;
    std::unique_ptr<SoaPf> pf;
    void init_step(const double* a, const double* obs, int n) override { pf->init_step(a, obs, n); }
    void step(const double* obs, int n) override { pf->step(obs, n); }
    double ess(int mode) override { return pf->ess(mode == MP_ESS_FRESH); }
    double resample() override { return pf->resample(); }
    void set_scheme(int s_) override { pf->scheme = s_; }
    double log_ml() override { return pf->log_ml_estimate(); }
    void read_state(double* out) override { std::memcpy(out, pf->x.data(), pf->x.size() * sizeof(double)); }
    void read_logw(double* out) override { std::memcpy(out, pf->logw.data(), pf->n * sizeof(double)); }
    void read_parents(uint32_t* out) override { std::memcpy(out, pf->parents.data(), pf->n * sizeof(uint32_t)); }
    int64_t time() override { return pf->t; }
};

struct oracle_pf {
    std::unique_ptr<IPf> impl;
    bool canonical;
    // each call runs under the handle's math mode
    struct Scope { bool prev; Scope(bool c) : prev(canonical_mode()) { canonical_mode() = c; } ~Scope() { canonical_mode() = prev; } };
};

static HmmParams hmm_from_params(const double* p, int n) {
    HmmParams h;
    if (n < 2) throw Panic("hmm: params too short");
    h.n_states = (int)p[0]; h.n_obs = (int)p[1];
    const int S = h.n_states, O = h.n_obs;
    if (n != 2 + S + O * S + S * S) throw Panic("hmm: params length mismatch");
    h.prior.assign(p + 2, p + 2 + S);
    h.emission.assign(p + 2 + S, p + 2 + S + O * S);
    h.transition.assign(p + 2 + S + O * S, p + 2 + S + O * S + S * S);
    return h;
}

extern "C" {

const char* oracle_last_error(void) { return g_err.c_str(); }

int32_t oracle_pf_create(const mp_model_desc* m, uint64_t n, uint64_t seed, const mp_shard* shard, uint32_t flags,
                         int32_t variant, oracle_pf** out) {
    (void)flags;
    GUARD({
        if (!m || !out || n == 0) throw Panic("null/zero argument");
        const bool canon = variant & 1, soa = variant & 2, fast = variant & 4;
        auto h = std::make_unique<oracle_pf>();
        h->canonical = canon;
        if (soa) {
            auto e = std::make_unique<SoaEngine>();
            if (m->kind == MP_MODEL_LGSSM1) {
                if (m->n_params != 5) throw Panic("lgssm1: 5 params");
                e->model = std::make_unique<SoaLgssm1>(LgssmParams{m->params[0], m->params[1], m->params[2], m->params[3], m->params[4]});
            } else if (m->kind == MP_MODEL_SPIRAL) {
                e->model = std::make_unique<SoaSpiral>();
            } else if (m->kind == MP_MODEL_HMM) {
                e->model = std::make_unique<SoaHmm>(hmm_from_params(m->params, m->n_params));
            } else if (m->kind == MP_MODEL_BEARINGS) {
                if (m->n_params != 6) throw Panic("bearings: 6 params");
                e->model = std::make_unique<SoaBearings>(BearingsParams{m->params[0], m->params[1], m->params[2], m->params[3], m->params[4], m->params[5]});
            } else if (m->kind == MP_MODEL_LGSSM_BAND) {
                if (m->n_params != 6) throw Panic("lgssm_band: 6 params");
                e->model = std::make_unique<SoaLgssmBand>(BandParams{(int)m->params[0], m->params[1], m->params[2], m->params[3], m->params[4], m->params[5]});
            } else if (m->kind == MP_MODEL_POINTED_2D) {
                if (m->n_params != 8) throw Panic("pointed_2d: 8 params");
                e->model = std::make_unique<SoaPointed>(Bounds{m->params[0], m->params[1], m->params[2], m->params[3]},
                                                        Mat(2, std::vector<double>(m->params + 4, m->params + 8)));
            } else if (m->kind == MP_MODEL_LINE) {
                e->model = std::make_unique<SoaLine>(Vec(m->params, m->params + m->n_params));
            } else if (m->kind == MP_MODEL_LGSSM_DENSE) {
                const int D = (int)m->params[0];
                if (m->n_params != 2 + 3 * D * D) throw Panic("lgssm_dense: params = {D, sig0, A, Q, R}");
                const double* q = m->params + 2;
                e->model = std::make_unique<SoaLgssmDense>(D, m->params[1], Mat(D, std::vector<double>(q, q + D * D)),
                                                           Mat(D, std::vector<double>(q + D * D, q + 2 * D * D)),
                                                           Mat(D, std::vector<double>(q + 2 * D * D, q + 3 * D * D)));
            } else if (functor_registry().count(m->kind)) {
                e->model = functor_registry()[m->kind].soa(*m);   // the product's model functor, the checker's handlers and distributions
            } else throw Panic("unsupported model kind for the SoA engine");
            e->pf = std::make_unique<SoaPf>(e->model.get(), (size_t)n, seed, canon, shard ? shard->n_global : 0, shard ? shard->slot_offset : 0);
            h->impl = std::move(e);
        } else {
            if (shard && (shard->n_global != n || shard->slot_offset != 0)) throw Panic("dynamic engine is unsharded");
            if (m->kind == MP_MODEL_LGSSM1) {
                if (m->n_params != 5) throw Panic("lgssm1: 5 params");
                auto e = std::make_unique<DynPf<double>>();
                e->model = make_lgssm_model(LgssmParams{m->params[0], m->params[1], m->params[2], m->params[3], m->params[4]});
                e->dim_state = 1; e->dim_obs = 1;
                e->mk_constraints = [](const double* y) { DynTrie c; c.observe("y", arc(y[0])); return c; };
                e->mk_state = [](const double* a) { return a[0]; };
                e->put_state = [](const double& s, double* o) { o[0] = s; };
                e->ps = std::make_unique<DynPf<double>::PS>(e->model, (size_t)n, seed);
                e->ps->canonical_resampling = canon; e->ps->fast_search = fast;
                h->impl = std::move(e);
            } else if (m->kind == MP_MODEL_SPIRAL) {
                auto e = std::make_unique<DynPf<Vec>>();
                e->model = make_spiral_model();
                e->dim_state = 2; e->dim_obs = 2;
                e->mk_constraints = [](const double* y) { DynTrie c; c.observe("obs", arc(Vec{y[0], y[1]})); return c; };
                e->mk_state = [](const double* a) { return Vec{a[0], a[1]}; };
                e->put_state = [](const Vec& s, double* o) { o[0] = s[0]; o[1] = s[1]; };
                e->ps = std::make_unique<DynPf<Vec>::PS>(e->model, (size_t)n, seed);
                e->ps->canonical_resampling = canon; e->ps->fast_search = fast;
                h->impl = std::move(e);
            } else if (m->kind == MP_MODEL_BEARINGS || m->kind == MP_MODEL_LGSSM_BAND) {
                auto e = std::make_unique<DynPf<Vec>>();
                int d, dobs;
                if (m->kind == MP_MODEL_BEARINGS) {
                    if (m->n_params != 6) throw Panic("bearings: 6 params");
                    e->model = make_bearings_model(BearingsParams{m->params[0], m->params[1], m->params[2], m->params[3], m->params[4], m->params[5]});
                    d = 4; dobs = 1;
                    e->mk_constraints = [](const double* y) { DynTrie c; c.observe("theta", arc(y[0])); return c; };
                } else {
                    if (m->n_params != 6) throw Panic("lgssm_band: 6 params");
                    d = dobs = (int)m->params[0];
                    e->model = make_lgssm_band_model(BandParams{d, m->params[1], m->params[2], m->params[3], m->params[4], m->params[5]});
                    e->mk_constraints = [d](const double* y) { DynTrie c; for (int j = 0; j < d; ++j) c.observe("y/" + std::to_string(j), arc(y[j])); return c; };
                }
                e->dim_state = d; e->dim_obs = dobs;
                e->mk_state = [d](const double* a) { return Vec(a, a + d); };
                e->put_state = [d](const Vec& s, double* o) { for (int j = 0; j < d; ++j) o[j] = s[(size_t)j]; };
                e->ps = std::make_unique<DynPf<Vec>::PS>(e->model, (size_t)n, seed);
                e->ps->canonical_resampling = canon; e->ps->fast_search = fast;
                h->impl = std::move(e);
            } else if (m->kind == MP_MODEL_POINTED_2D || m->kind == MP_MODEL_LINE) {
                auto e = std::make_unique<DynPf<Vec>>();
                int dobs;
                if (m->kind == MP_MODEL_POINTED_2D) {
                    if (m->n_params != 8) throw Panic("pointed_2d: 8 params");
                    e->model = make_pointed_unfold(Bounds{m->params[0], m->params[1], m->params[2], m->params[3]},
                                                   Mat(2, std::vector<double>(m->params + 4, m->params + 8)));
                    dobs = 2;
                    e->mk_constraints = [](const double* y) { DynTrie c; c.observe("obs", arc(Vec{y[0], y[1]})); return c; };
                } else {
                    dobs = m->n_params;
                    e->model = make_line_unfold(Vec(m->params, m->params + m->n_params));
                    e->mk_constraints = [dobs](const double* y) { DynTrie c; for (int j = 0; j < dobs; ++j) c.observe("ys/" + std::to_string(j), arc(y[j])); return c; };
                }
                e->dim_state = 2; e->dim_obs = dobs;
                e->mk_state = [](const double* a) { return Vec{a ? a[0] : 0., a ? a[1] : 0.}; };
                e->put_state = [](const Vec& s_, double* o) { o[0] = s_[0]; o[1] = s_[1]; };
                e->ps = std::make_unique<DynPf<Vec>::PS>(e->model, (size_t)n, seed);
                e->ps->canonical_resampling = canon; e->ps->fast_search = fast;
                h->impl = std::move(e);
            } else if (m->kind == MP_MODEL_HMM) {
                auto e = std::make_unique<HmmPf>(hmm_from_params(m->params, m->n_params));
                e->ps = std::make_unique<HmmPf::PS>(e->model, (size_t)n, seed);
                e->ps->canonical_resampling = canon; e->ps->fast_search = fast;
                h->impl = std::move(e);
            } else if (functor_registry().count(m->kind)) {
                // the product's model functor interpreted by the dynamic handler: one sample_at per site, tries and all
                const FunctorEntry& fe = functor_registry()[m->kind];
                auto e = std::make_unique<DynPf<Vec>>();
                const int d = fe.dim_state, dobs = fe.dim_obs;
                e->model = fe.dyn(*m);
                e->dim_state = d; e->dim_obs = dobs;
                e->mk_constraints = fe.constraints;
                e->mk_state = [d](const double* a) { Vec v((size_t)d, 0.); if (a) for (int j = 0; j < d; ++j) v[(size_t)j] = a[j]; return v; };
                e->put_state = [d](const Vec& s_, double* o) { for (int j = 0; j < d; ++j) o[j] = s_[(size_t)j]; };
                e->ps = std::make_unique<DynPf<Vec>::PS>(e->model, (size_t)n, seed);
                e->ps->canonical_resampling = canon; e->ps->fast_search = fast;
                h->impl = std::move(e);
            } else throw Panic("unsupported model kind");
        }
        *out = h.release();
    })
}
int32_t oracle_pf_set_threads(oracle_pf* h, int32_t threads) {
    GUARD({ auto* e = dynamic_cast<SoaEngine*>(h->impl.get()); if (!e) throw Panic("threads: SoA engine only"); e->pf->threads = threads; })
}
int32_t oracle_pf_init_step(oracle_pf* h, const double* args0, const double* obs, int32_t n_steps) {
    GUARD({ if (n_steps < 1) throw Panic("constraints: need at least one step"); oracle_pf::Scope s(h->canonical); h->impl->init_step(args0, obs, n_steps); })
}
int32_t oracle_pf_step(oracle_pf* h, const double* obs, int32_t n_steps) {
    GUARD({ if (n_steps < 1) throw Panic("constraints: need at least one step"); oracle_pf::Scope s(h->canonical); h->impl->step(obs, n_steps); })
}
int32_t oracle_pf_effective_sample_size(oracle_pf* h, int32_t mode, double* out) {
    GUARD({ oracle_pf::Scope s(h->canonical); *out = h->impl->ess(mode); })
}
int32_t oracle_pf_resample(oracle_pf* h, int32_t scheme, double* ltw) {
    GUARD({ h->impl->set_scheme(scheme);
            oracle_pf::Scope s(h->canonical); const double L = h->impl->resample(); if (ltw) *ltw = L; })
}
int32_t oracle_pf_log_marginal_likelihood_estimate(oracle_pf* h, double* out) {
    GUARD({ oracle_pf::Scope s(h->canonical); *out = h->impl->log_ml(); })
}
int32_t oracle_pf_read_state(oracle_pf* h, double* out) { GUARD({ h->impl->read_state(out); }) }
int32_t oracle_pf_read_log_weights(oracle_pf* h, double* out) { GUARD({ h->impl->read_logw(out); }) }
int32_t oracle_pf_read_parents(oracle_pf* h, uint32_t* out) { GUARD({ h->impl->read_parents(out); }) }
int32_t oracle_pf_read_trajectory(oracle_pf* h, uint64_t i, double* out, int32_t* t_steps) {
    GUARD({ *t_steps = (int32_t)h->impl->trajectory(i, out); })
}
int32_t oracle_pf_time(oracle_pf* h, int64_t* out) { GUARD({ *out = h->impl->time(); }) }
int32_t oracle_pf_destroy(oracle_pf* h) { delete h; return MP_OK; }

// ---- sharded phases (SoA engine, canonical) ------------------------------------------------------
static SoaPf* soa_of(oracle_pf* h) {
    auto* e = dynamic_cast<SoaEngine*>(h->impl.get());
    if (!e || !h->canonical) throw Panic("shard phases: SoA engine in canonical mode only");
    return e->pf.get();
}
int32_t oracle_pf_shard_tiles(oracle_pf* h, double* tm, uint64_t* tW, uint64_t* tW2) {
    GUARD({ oracle_pf::Scope s(true); soa_of(h)->shard_tiles(tm, tW, tW2); })
}
int32_t oracle_pf_shard_route(oracle_pf* h, int32_t scheme, const double* tm_all, const uint64_t* tW_all, const uint64_t* tW2_all, uint64_t nt_all,
                              int32_t world, int32_t rank, uint64_t* req_out, int64_t* send_counts) {
    GUARD({ oracle_pf::Scope s(true); soa_of(h)->scheme = scheme; soa_of(h)->shard_route(tm_all, tW_all, tW2_all, (size_t)nt_all, world, rank, req_out, send_counts); })
}
int32_t oracle_pf_shard_resolve(oracle_pf* h, const uint64_t* req_in, uint64_t n_req, double* rows) {
    GUARD({ soa_of(h)->shard_resolve(req_in, n_req, rows); })
}
int32_t oracle_pf_shard_scatter(oracle_pf* h, const double* rows, double* L) {
    GUARD({ const double l = soa_of(h)->shard_scatter(rows); if (L) *L = l; })
}
int32_t oracle_pf_shard_query(oracle_pf* h, const double* tm_all, const uint64_t* tW_all, const uint64_t* tW2_all, uint64_t nt_all, double* lml, double* ess) {
    GUARD({ oracle_pf::Scope s(true); soa_of(h)->shard_query(tm_all, tW_all, tW2_all, (size_t)nt_all, lml, ess); })
}
// "owner keeps" form (mirrors mp_pf_shard_owned_count / _expand / _commit of include/modppl_hip.h)
int32_t oracle_pf_shard_owned_count(oracle_pf* h, int32_t scheme, const double* tm_all, const uint64_t* tW_all, const uint64_t* tW2_all, uint64_t nt_all,
                                    int32_t world, int32_t rank, uint64_t* counts) {
    GUARD({ oracle_pf::Scope s(true); soa_of(h)->scheme = scheme; soa_of(h)->shard_owned_count(tm_all, tW_all, tW2_all, (size_t)nt_all, world, rank, counts); })
}
int32_t oracle_pf_shard_owned_expand(oracle_pf* h, int32_t rank, double* send, uint64_t* n_sent) {
    GUARD({ const uint64_t s = soa_of(h)->shard_owned_expand(rank, send); if (n_sent) *n_sent = s; })
}
int32_t oracle_pf_shard_owned_adopt(oracle_pf* h, int32_t rank, const double* recv, uint64_t n_recv, double* L) {
    GUARD({ const double l = soa_of(h)->shard_owned_adopt(rank, recv, n_recv); if (L) *L = l; })
}

// ---- importance.rs:12-50 ------------------------------------------------------------------------
// variant bit 1 = canonical, bit 2 = SoA engine (else the generic importance_resampling over the dynamic DynUnfold).
int32_t oracle_importance_resampling(const mp_model_desc* m, const double* args0, const double* obs, int32_t n_steps, uint64_t num_samples,
                                     uint64_t num_ret_samples, uint64_t seed, int32_t variant, double* log_ml_estimate,
                                     double* log_normalized_weights, uint64_t* resampled_indices, double* final_states) {
    GUARD({
        const bool canon = variant & 1, soa = variant & 2;
        oracle_pf::Scope scope(canon);
        if (soa) {
            oracle_pf* h = nullptr;
            if (oracle_pf_create(m, num_samples, seed, nullptr, 0, variant, &h) != MP_OK) throw Panic(g_err);
            std::unique_ptr<oracle_pf> guard(h);
            auto* e = dynamic_cast<SoaEngine*>(h->impl.get());
            e->pf->init_step(args0, obs, n_steps);
            const std::vector<double>& w = e->pf->logw;
            double L;
            CanonFull c;
            if (canon) { c = canonical_normalize(w, num_samples); if (c.c.degenerate()) throw Panic("all log-weights are -inf"); L = c.c.L; }
            else L = logsumexp(w);
            if (log_ml_estimate) *log_ml_estimate = L - o_ln((double)num_samples);
            std::vector<double> probs;
            for (size_t i = 0; i < w.size(); ++i) {
                const double lnw = w[i] - L;
                if (log_normalized_weights) log_normalized_weights[i] = lnw;
                if (!canon) probs.push_back(o_exp(lnw));
            }
            if (final_states) std::memcpy(final_states, e->pf->x.data(), e->pf->x.size() * sizeof(double));
            if (resampled_indices) {
                std::vector<double> cdf;
                if (!canon) { Categorical::check_sum(probs); double t = 0.; for (double p : probs) { t += p; cdf.push_back(t); } }
                for (uint64_t j = 0; j < num_ret_samples; ++j) {
                    Rng r = resample_rng(seed, DOM_IS, 0, j);
                    if (canon) resampled_indices[j] = canonical_parent(c, canonical_target(r.u52(), c.c.Q));
                    else {
                        const double u = r.u01();
                        if (!(0. < u)) throw Panic("categorical returned -1 (u == 0)");
                        const size_t p = (size_t)(std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin());
                        if (p >= cdf.size()) throw Panic("categorical: index out of bounds");
                        resampled_indices[j] = p;
                    }
                }
            }
        } else {
            auto run = [&](auto& model, auto mk_state, auto mk_constraints, auto put_state, int d, int dobs) {
                std::vector<DynTrie> cons;
                for (int k = 0; k < n_steps; ++k) cons.push_back(mk_constraints(obs + (size_t)k * dobs));
                std::vector<double> z((size_t)d, 0.);
                auto res = importance_resampling(seed, model, std::make_pair((int64_t)n_steps, mk_state(args0 ? args0 : z.data())), cons,
                                                 (uint32_t)num_samples, (uint32_t)num_ret_samples, canon);
                if (log_ml_estimate) *log_ml_estimate = res.log_ml_estimate;
                for (size_t i = 0; i < res.traces.size(); ++i) {
                    if (log_normalized_weights) log_normalized_weights[i] = res.log_normalized_weights[i];
                    if (final_states) put_state(res.traces[i].retv->back(), final_states + i * (size_t)d);
                }
                if (resampled_indices) for (size_t j = 0; j < res.resampled_indices.size(); ++j) resampled_indices[j] = res.resampled_indices[j];
            };
            if (m->kind == MP_MODEL_LGSSM1) {
                auto model = make_lgssm_model(LgssmParams{m->params[0], m->params[1], m->params[2], m->params[3], m->params[4]});
                run(model, [](const double* a) { return a[0]; }, [](const double* y) { DynTrie c; c.observe("y", arc(y[0])); return c; },
                    [](const double& s_, double* o) { o[0] = s_; }, 1, 1);
            } else if (m->kind == MP_MODEL_SPIRAL) {
                auto model = make_spiral_model();
                run(model, [](const double* a) { return Vec{a[0], a[1]}; }, [](const double* y) { DynTrie c; c.observe("obs", arc(Vec{y[0], y[1]})); return c; },
                    [](const Vec& s_, double* o) { o[0] = s_[0]; o[1] = s_[1]; }, 2, 2);
            } else if (m->kind == MP_MODEL_POINTED_2D) {   // tests/importance.rs:17-50 (the model of test_importance_handcoded, DynGenFn form)
                auto model = make_pointed_unfold(Bounds{m->params[0], m->params[1], m->params[2], m->params[3]}, Mat(2, std::vector<double>(m->params + 4, m->params + 8)));
                run(model, [](const double* a) { return Vec{a[0], a[1]}; }, [](const double* y) { DynTrie c; c.observe("obs", arc(Vec{y[0], y[1]})); return c; },
                    [](const Vec& s_, double* o) { o[0] = s_[0]; o[1] = s_[1]; }, 2, 2);
            } else if (m->kind == MP_MODEL_LINE) {         // tests/importance.rs:54-76 (test_importance_dyngenfn)
                const int dobs = m->n_params;
                auto model = make_line_unfold(Vec(m->params, m->params + dobs));
                run(model, [](const double* a) { return Vec{a[0], a[1]}; },
                    [dobs](const double* y) { DynTrie c; for (int j = 0; j < dobs; ++j) c.observe("ys/" + std::to_string(j), arc(y[j])); return c; },
                    [](const Vec& s_, double* o) { o[0] = s_[0]; o[1] = s_[1]; }, 2, dobs);
            } else throw Panic("dynamic importance_resampling: lgssm1 / spiral / pointed_2d / line only");
        }
    })
}

// ---- DynUnfold::simulate (dynunfold.rs:22-39) over n traces: states[n][T][d], obs[n][T][dobs] ------------------------
int32_t oracle_unfold_simulate(const mp_model_desc* m, const double* args0, int32_t n_steps, uint64_t n, uint64_t seed, int32_t canon,
                               double* states, double* obs) {
    GUARD({
        oracle_pf::Scope scope(canon != 0);
        auto run = [&](auto& model, auto mk_state, auto put_state, auto read_obs, int d, int dobs) {
            std::vector<double> z((size_t)d, 0.);
            for (uint64_t i = 0; i < n; ++i) {
                Rng r; r.seed = seed; r.slot = (uint32_t)i; r.step = 0;
                auto tr = model.simulate(r, std::make_pair((int64_t)n_steps, mk_state(args0 ? args0 : z.data())));
                for (int t = 0; t < n_steps; ++t) {
                    put_state((*tr.retv)[(size_t)t], states + (i * (uint64_t)n_steps + (uint64_t)t) * (uint64_t)d);
                    read_obs(tr.data[(size_t)t], obs + (i * (uint64_t)n_steps + (uint64_t)t) * (uint64_t)dobs);
                }
            }
        };
        auto vec_state = [](int d) { return [d](const double* a) { return Vec(a, a + d); }; };
        auto vec_put = [](int d) { return [d](const Vec& s_, double* o) { for (int j = 0; j < d; ++j) o[j] = s_[(size_t)j]; }; };
        if (m->kind == MP_MODEL_LGSSM1) {
            auto model = make_lgssm_model(LgssmParams{m->params[0], m->params[1], m->params[2], m->params[3], m->params[4]});
            run(model, [](const double* a) { return a[0]; }, [](const double& s_, double* o) { o[0] = s_; },
                [](const DynTrie& c, double* o) { o[0] = c.read<double>("y"); }, 1, 1);
        } else if (m->kind == MP_MODEL_SPIRAL) {
            auto model = make_spiral_model();
            run(model, vec_state(2), vec_put(2), [](const DynTrie& c, double* o) { const Vec v = c.read<Vec>("obs"); o[0] = v[0]; o[1] = v[1]; }, 2, 2);
        } else if (m->kind == MP_MODEL_BEARINGS) {
            auto model = make_bearings_model(BearingsParams{m->params[0], m->params[1], m->params[2], m->params[3], m->params[4], m->params[5]});
            run(model, vec_state(4), vec_put(4), [](const DynTrie& c, double* o) { o[0] = c.read<double>("theta"); }, 4, 1);
        } else if (m->kind == MP_MODEL_LGSSM_BAND) {
            const int d = (int)m->params[0];
            auto model = make_lgssm_band_model(BandParams{d, m->params[1], m->params[2], m->params[3], m->params[4], m->params[5]});
            run(model, vec_state(d), vec_put(d), [d](const DynTrie& c, double* o) { for (int j = 0; j < d; ++j) o[j] = c.read<double>("y/" + std::to_string(j)); }, d, d);
        } else if (m->kind == MP_MODEL_POINTED_2D) {
            auto model = make_pointed_unfold(Bounds{m->params[0], m->params[1], m->params[2], m->params[3]}, Mat(2, std::vector<double>(m->params + 4, m->params + 8)));
            run(model, vec_state(2), vec_put(2), [](const DynTrie& c, double* o) { if (c.search("obs")) { const Vec v = c.read<Vec>("obs"); o[0] = v[0]; o[1] = v[1]; } else { o[0] = o[1] = 0.; } }, 2, 2);
        } else if (m->kind == MP_MODEL_LINE) {
            const int dobs = m->n_params;
            auto model = make_line_unfold(Vec(m->params, m->params + dobs));
            run(model, vec_state(2), vec_put(2),
                [dobs](const DynTrie& c, double* o) { for (int j = 0; j < dobs; ++j) o[j] = c.search("ys") ? c.read<double>("ys/" + std::to_string(j)) : 0.; }, 2, dobs);
        } else if (functor_registry().count(m->kind)) {
            // a registered functor model, Simulate mode of the dynamic handler; observed sites read back by their obs slot
            const FunctorEntry& fe = functor_registry()[m->kind];
            auto model = fe.dyn(*m);
            const int d = fe.dim_state, dobs = fe.dim_obs;
            const DynTrie probe = fe.constraints(std::vector<double>((size_t)dobs, 0.).data());   // the observed sites' addresses
            std::vector<std::string> addr((size_t)dobs);
            for (int site = 0; site < 64; ++site)
                if (probe.search(functor_addr(site))) {
                    // which slot: the site whose constraint carries y[k] — functor_constraints fills slots in obs_of order
                    std::vector<double> mark((size_t)dobs);
                    for (int k = 0; k < dobs; ++k) mark[(size_t)k] = (double)(k + 1);
                    addr[(size_t)(fe.constraints(mark.data()).read<double>(functor_addr(site)) - 1.)] = functor_addr(site);
                }
            run(model, vec_state(d), vec_put(d),
                [dobs, addr](const DynTrie& c, double* o) { for (int k = 0; k < dobs; ++k) o[k] = c.read<double>(addr[(size_t)k]); }, d, dobs);
        } else throw Panic("simulate: unsupported model kind");
    })
}

// ---- mh.rs over N independent chains of a REGISTERED functor model (mh_functor_adapter.hpp) ----------------------------
struct oracle_mhfn {
    std::shared_ptr<MhFnModel> m;
    std::vector<MhFnTrace> traces;
    uint64_t seed = 0, iters = 0;
    bool canonical = false;
};
int32_t oracle_mhfn_create(int32_t kind, const double* params, int32_t n_params, const int32_t* cons_sites, const double* cons_vals, int32_t n_cons,
                           uint64_t n_chains, uint64_t seed, int32_t canon, oracle_mhfn** out) {
    GUARD({
        auto it = mhfn_models().find(kind);
        if (it == mhfn_models().end()) throw Panic("no MH functor model of this kind");
        auto h = std::make_unique<oracle_mhfn>();
        h->seed = seed; h->canonical = canon != 0;
        oracle_pf::Scope scope(h->canonical);
        h->m = it->second(params, n_params);
        for (uint64_t i = 0; i < n_chains; ++i) {
            Rng r; r.seed = seed; r.slot = (uint32_t)i; r.step = 0;
            h->traces.push_back(h->m->model().generate(r, 0, h->m->constraints(cons_sites, cons_vals, n_cons)).first);
        }
        *out = h.release();
    })
}
int32_t oracle_mhfn_n_sites(oracle_mhfn* h, int32_t* out) { GUARD({ *out = h->m->ns(); }) }
int32_t oracle_mhfn_step(oracle_mhfn* h, int32_t proposal_kind, const double* args, int32_t n_args, int32_t n_iters, uint64_t* accepted) {
    GUARD({
        oracle_pf::Scope scope(h->canonical);
        const MhFnProposal proposal = h->m->proposal(proposal_kind, args, n_args);
        uint64_t acc = 0;
        for (size_t i = 0; i < h->traces.size(); ++i) {
            for (int k = 0; k < n_iters; ++k) {
                Rng r; r.seed = h->seed; r.slot = (uint32_t)i; r.step = (uint32_t)(h->iters + 1 + (uint64_t)k);
                auto [tr, ok] = metropolis_hastings<int, mp_fn_ret, int>(r, h->m->model(), std::move(h->traces[i]), proposal, 0);
                h->traces[i] = std::move(tr);
                acc += ok;
            }
        }
        h->iters += (uint64_t)n_iters;
        if (accepted) *accepted = acc;
    })
}
int32_t oracle_mhfn_regen(oracle_mhfn* h, const int32_t* mask_sites, int32_t n_mask, int32_t cycle, int32_t n_iters, uint64_t* accepted) {
    GUARD({
        oracle_pf::Scope scope(h->canonical);
        uint64_t acc = 0;
        for (size_t i = 0; i < h->traces.size(); ++i) {
            for (int k = 0; k < n_iters; ++k) {
                AddrMap mask;   // empty: the trace's whole schema (dyngenfn.rs:571)
                if (cycle && n_mask > 0) mask.visit(h->m->flat_addr(mask_sites[(h->iters + (uint64_t)k) % (uint64_t)n_mask]));
                else for (int q = 0; q < n_mask; ++q) mask.visit(h->m->flat_addr(mask_sites[q]));
                Rng r; r.seed = h->seed; r.slot = (uint32_t)i; r.step = (uint32_t)(h->iters + 1 + (uint64_t)k);
                auto [tr, ok] = regenerative_metropolis_hastings<int, mp_fn_ret>(r, h->m->model(), std::move(h->traces[i]), mask);
                h->traces[i] = std::move(tr);
                acc += ok;
            }
        }
        h->iters += (uint64_t)n_iters;
        if (accepted) *accepted = acc;
    })
}
// ---- the GFI operations one at a time (gfi.rs:57-90), chain by chain through the dynamic machinery: the checker of mp_fn_* ----
static DynTrie mhfn_chain_constraints(oracle_mhfn* h, const int32_t* sites, const double* vals, int32_t n_cons, const double* chain_values,
                                      const uint64_t* chain_present, size_t i) {
    if (!chain_values) return h->m->constraints(sites, vals, n_cons);
    const int ns = h->m->ns();
    std::vector<int32_t> s2;
    std::vector<double> v2;
    for (int k = 0; k < ns; ++k)
        if ((chain_present[i] >> k) & 1u) { s2.push_back(k); v2.push_back(chain_values[i * (size_t)ns + k]); }
    return h->m->constraints(s2.data(), v2.data(), (int)s2.size());
}
static uint32_t mhfn_step(oracle_mhfn* h, uint32_t rng_step) { return rng_step ? rng_step : (uint32_t)(++h->iters); }
int32_t oracle_mhfn_update(oracle_mhfn* h, int32_t argdiff, uint32_t rng_step, const int32_t* sites, const double* vals, int32_t n_cons,
                           const double* chain_values, const uint64_t* chain_present, double* weights, double* discard_values, uint64_t* discard_present) {
    GUARD({
        oracle_pf::Scope scope(h->canonical);
        const uint32_t step = mhfn_step(h, rng_step);
        const int ns = h->m->ns();
        for (size_t i = 0; i < h->traces.size(); ++i) {
            Rng r; r.seed = h->seed; r.slot = (uint32_t)i; r.step = step;
            auto [tr, discard, w] = h->m->model().update(r, std::move(h->traces[i]), 0, argdiff ? ArgDiff::Unknown : ArgDiff::NoChange,
                                                        mhfn_chain_constraints(h, sites, vals, n_cons, chain_values, chain_present, i));
            h->traces[i] = std::move(tr);
            if (weights) weights[i] = w;
            if (discard_values && discard_present) h->m->view(discard, discard_values + i * (size_t)ns, discard_present + i);
        }
    })
}
int32_t oracle_mhfn_regenerate(oracle_mhfn* h, int32_t argdiff, uint32_t rng_step, const int32_t* mask_sites, int32_t n_mask, double* weights) {
    GUARD({
        oracle_pf::Scope scope(h->canonical);
        const uint32_t step = mhfn_step(h, rng_step);
        for (size_t i = 0; i < h->traces.size(); ++i) {
            AddrMap mask;
            for (int q = 0; q < n_mask; ++q) mask.visit(h->m->flat_addr(mask_sites[q]));
            Rng r; r.seed = h->seed; r.slot = (uint32_t)i; r.step = step;
            auto [tr, w] = h->m->model().regenerate(r, std::move(h->traces[i]), 0, argdiff ? ArgDiff::Unknown : ArgDiff::NoChange, mask);
            h->traces[i] = std::move(tr);
            if (weights) weights[i] = w;
        }
    })
}
int32_t oracle_mhfn_assess(oracle_mhfn* h, int32_t proposal_kind, const double* args, int32_t n_args, uint32_t rng_step, const int32_t* sites,
                           const double* vals, int32_t n_cons, const double* chain_values, const uint64_t* chain_present, double* weights) {
    GUARD({
        oracle_pf::Scope scope(h->canonical);
        const uint32_t step = mhfn_step(h, rng_step);
        for (size_t i = 0; i < h->traces.size(); ++i) {
            Rng r; r.seed = h->seed; r.slot = (uint32_t)i; r.step = step;
            DynTrie c = mhfn_chain_constraints(h, sites, vals, n_cons, chain_values, chain_present, i);
            if (proposal_kind < 0) weights[i] = h->m->model().assess(r, 0, std::move(c));
            else {
                const MhFnProposal proposal = h->m->proposal(proposal_kind, args, n_args);
                weights[i] = proposal.assess(r, {&h->traces[i], 0}, std::move(c));
            }
        }
    })
}
int32_t oracle_mhfn_propose(oracle_mhfn* h, int32_t proposal_kind, const double* args, int32_t n_args, uint32_t rng_step, double* choice_values,
                            uint64_t* choice_present, double* weights) {
    GUARD({
        oracle_pf::Scope scope(h->canonical);
        const uint32_t step = mhfn_step(h, rng_step);
        const int ns = h->m->ns();
        const MhFnProposal proposal = h->m->proposal(proposal_kind, args, n_args);
        for (size_t i = 0; i < h->traces.size(); ++i) {
            Rng r; r.seed = h->seed; r.slot = (uint32_t)i; r.step = step;
            auto [choices, w] = proposal.propose(r, {&h->traces[i], 0});
            if (weights) weights[i] = w;
            h->m->view(choices, choice_values + i * (size_t)ns, choice_present + i);
        }
    })
}
// (trace, weight) = model.generate(args, constraints) / trace = model.simulate(args) on every chain (gfi.rs:51-55): the checker of
// mp_fn_generate / mp_fn_simulate; the traces are replaced
int32_t oracle_mhfn_generate(oracle_mhfn* h, uint32_t rng_step, const int32_t* sites, const double* vals, int32_t n_cons, const double* chain_values,
                             const uint64_t* chain_present, double* weights) {
    GUARD({
        oracle_pf::Scope scope(h->canonical);
        const uint32_t step = mhfn_step(h, rng_step);
        for (size_t i = 0; i < h->traces.size(); ++i) {
            Rng r; r.seed = h->seed; r.slot = (uint32_t)i; r.step = step;
            auto [tr, w] = h->m->model().generate(r, 0, mhfn_chain_constraints(h, sites, vals, n_cons, chain_values, chain_present, i));
            h->traces[i] = std::move(tr);
            if (weights) weights[i] = w;
        }
    })
}
int32_t oracle_mhfn_simulate(oracle_mhfn* h, uint32_t rng_step, double* logjp) {
    GUARD({
        oracle_pf::Scope scope(h->canonical);
        const uint32_t step = mhfn_step(h, rng_step);
        for (size_t i = 0; i < h->traces.size(); ++i) {
            Rng r; r.seed = h->seed; r.slot = (uint32_t)i; r.step = step;
            h->traces[i] = h->m->model().simulate(r, 0);
            if (logjp) logjp[i] = h->traces[i].logjp;
        }
    })
}
// importance_sampling / importance_resampling (importance.rs:12-50) over a registered functor model through the GENERIC functions of
// inference.hpp (the ones the Unfold checker uses): the N traces come back as a chain handle
int32_t oracle_mhfn_importance(int32_t kind, const double* params, int32_t n_params, const int32_t* cons_sites, const double* cons_vals, int32_t n_cons,
                               uint64_t num_samples, uint64_t num_ret, uint64_t seed, int32_t canon, double* log_ml, double* lnw, uint64_t* idx,
                               oracle_mhfn** traces_out) {
    GUARD({
        auto it = mhfn_models().find(kind);
        if (it == mhfn_models().end()) throw Panic("no MH functor model of this kind");
        auto h = std::make_unique<oracle_mhfn>();
        h->seed = seed; h->canonical = canon != 0;
        oracle_pf::Scope scope(h->canonical);
        h->m = it->second(params, n_params);
        auto res = importance_resampling<int, DynTrie, mp_fn_ret>(seed, h->m->model(), 0, h->m->constraints(cons_sites, cons_vals, n_cons), (uint32_t)num_samples,
                                                                  (uint32_t)num_ret, h->canonical);
        if (log_ml) *log_ml = res.log_ml_estimate;
        if (lnw) for (size_t i = 0; i < res.log_normalized_weights.size(); ++i) lnw[i] = res.log_normalized_weights[i];
        if (idx) for (size_t j = 0; j < res.resampled_indices.size(); ++j) idx[j] = (uint64_t)res.resampled_indices[j];
        h->traces = std::move(res.traces);
        if (traces_out) *traces_out = h.release();
    })
}
int32_t oracle_mhfn_read_trace(oracle_mhfn* h, double* values, uint64_t* present) {
    GUARD({
        const int ns = h->m->ns();
        for (size_t i = 0; i < h->traces.size(); ++i) h->m->view(h->traces[i].data, values + i * (size_t)ns, present + i);
    })
}
int32_t oracle_mhfn_read_logjp(oracle_mhfn* h, double* out) { GUARD({ for (size_t i = 0; i < h->traces.size(); ++i) out[i] = h->traces[i].logjp; }) }
int32_t oracle_mhfn_destroy(oracle_mhfn* h) { delete h; return MP_OK; }

// ---- the product's static MH handlers (mp_genfn.h) on the host, chain by chain: a checkee for the suite without a GPU ----
struct oracle_mhfn_static { std::shared_ptr<MhFnStatic> r; uint64_t n = 0; };
int32_t oracle_mhfn_static_create(int32_t kind, const double* params, int32_t n_params, const int32_t* cons_sites, const double* cons_vals, int32_t n_cons,
                                  uint64_t n_chains, uint64_t seed, oracle_mhfn_static** out) {
    GUARD({
        auto it = mhfn_static_models().find(kind);
        if (it == mhfn_static_models().end()) throw Panic("no MH functor model of this kind");
        auto h = std::make_unique<oracle_mhfn_static>();
        h->r = it->second(params, n_params);
        h->n = n_chains;
        h->r->create(n_chains, seed, cons_sites, cons_vals, n_cons);
        *out = h.release();
    })
}
int32_t oracle_mhfn_static_step(oracle_mhfn_static* h, int32_t proposal_kind, const double* args, int32_t n_args, int32_t n_iters, uint64_t* accepted) {
    GUARD({ const uint64_t a = h->r->mh(proposal_kind, args, n_args, n_iters); if (accepted) *accepted = a; })
}
int32_t oracle_mhfn_static_regen(oracle_mhfn_static* h, const int32_t* mask_sites, int32_t n_mask, int32_t cycle, int32_t n_iters, uint64_t* accepted) {
    GUARD({ const uint64_t a = h->r->regen(mask_sites, n_mask, cycle, n_iters); if (accepted) *accepted = a; })
}
int32_t oracle_mhfn_static_update(oracle_mhfn_static* h, const int32_t* sites, const double* vals, int32_t n_cons, int32_t unknown, uint32_t step,
                                  double* weights, uint64_t* disc_present) {
    GUARD({ h->r->update(sites, vals, n_cons, unknown, step, weights, disc_present); })
}
int32_t oracle_mhfn_static_read(oracle_mhfn_static* h, double* values, uint64_t* present, uint64_t* panics) {
    GUARD({ h->r->read(values, present); if (panics) *panics = h->r->panics(); })
}
int32_t oracle_mhfn_static_destroy(oracle_mhfn_static* h) { delete h; return MP_OK; }

// ---- mh.rs over N independent chains of hierarchical_model ------------------------------------
struct oracle_mh {
    Hierarchical hm;
    std::vector<Hierarchical::TraceT> traces;
    Vec xs;
    uint64_t seed = 0, iters = 0;
    bool canonical = false;
};
int32_t oracle_mh_create(const double* xs, const double* ys, int32_t n_data, int32_t constrain_is_linear, uint64_t n_chains, uint64_t seed,
                         int32_t canon, oracle_mh** out) {
    GUARD({
        auto h = std::make_unique<oracle_mh>();
        h->seed = seed; h->canonical = canon != 0;
        h->xs.assign(xs, xs + n_data);
        oracle_pf::Scope scope(h->canonical);
        for (uint64_t i = 0; i < n_chains; ++i) {
            DynTrie observations;  // tests/mh.rs:89
            for (int k = 0; k < n_data; ++k) observations.observe("(y, " + std::to_string(k) + ")", arc(ys[k]));
            if (constrain_is_linear >= 0) observations.observe("is_linear", arc(constrain_is_linear != 0));
            Rng r; r.seed = seed; r.slot = (uint32_t)i; r.step = 0;
            h->traces.push_back(h->hm.model.generate(r, h->xs, observations).first);  // tests/mh.rs:91
        }
        *out = h.release();
    })
}
static int32_t oracle_mh_run(oracle_mh* h, int kind, double drift_std, int32_t n_iters, uint64_t* accepted) {
    GUARD({
        oracle_pf::Scope scope(h->canonical);
        auto proposal = kind == 2 ? h->hm.add_or_remove_proposal() : h->hm.drift_proposal();
        uint64_t acc = 0;
        for (size_t i = 0; i < h->traces.size(); ++i) {
            for (int k = 0; k < n_iters; ++k) {
                Rng r; r.seed = h->seed; r.slot = (uint32_t)i; r.step = (uint32_t)(h->iters + 1 + (uint64_t)k);
                auto [tr, ok] = metropolis_hastings<Vec, Vec, double>(r, h->hm.model, std::move(h->traces[i]), proposal, drift_std);
                h->traces[i] = std::move(tr);
                acc += ok;
            }
        }
        h->iters += (uint64_t)n_iters;
        if (accepted) *accepted = acc;
    })
}
int32_t oracle_mh_step(oracle_mh* h, double drift_std, int32_t n_iters, uint64_t* accepted) { return oracle_mh_run(h, 1, drift_std, n_iters, accepted); }
// mh(&hierarchical_model, trace, &add_or_remove_param_proposal, ())  — tests/mh.rs:94
int32_t oracle_mh_step_add_or_remove(oracle_mh* h, int32_t n_iters, uint64_t* accepted) { return oracle_mh_run(h, 2, 0., n_iters, accepted); }
int32_t oracle_regen_mh_step(oracle_mh* h, const int32_t* mask_sites, int32_t n_mask, int32_t cycle, int32_t n_iters, uint64_t* accepted) {
    GUARD({
        oracle_pf::Scope scope(h->canonical);
        static const char* names[4] = {"is_linear", "coeffs/a", "coeffs/b", "coeffs/c"};
        uint64_t acc = 0;
        for (size_t i = 0; i < h->traces.size(); ++i) {
            for (int k = 0; k < n_iters; ++k) {
                AddrMap mask;
                if (cycle) mask.visit(names[mask_sites[(h->iters + (uint64_t)k) % (uint64_t)n_mask]]);
                else for (int q = 0; q < n_mask; ++q) mask.visit(names[mask_sites[q]]);
                Rng r; r.seed = h->seed; r.slot = (uint32_t)i; r.step = (uint32_t)(h->iters + 1 + (uint64_t)k);
                auto [tr, ok] = regenerative_metropolis_hastings<Vec, Vec>(r, h->hm.model, std::move(h->traces[i]), mask);
                h->traces[i] = std::move(tr);
                acc += ok;
            }
        }
        h->iters += (uint64_t)n_iters;
        if (accepted) *accepted = acc;
    })
}
int32_t oracle_mh_read_state(oracle_mh* h, double* out) {
    GUARD({
        for (size_t i = 0; i < h->traces.size(); ++i) {
            const auto& d = h->traces[i].data;
            const bool il = d.read<bool>("is_linear");
            out[4 * i] = il ? 1. : 0.;
            out[4 * i + 1] = d.read<double>("coeffs/a");
            out[4 * i + 2] = d.read<double>("coeffs/b");
            out[4 * i + 3] = il ? 0. : d.read<double>("coeffs/c");
        }
    })
}
int32_t oracle_mh_read_observations(oracle_mh* h, int32_t n_data, double* out) {
    GUARD({
        for (size_t i = 0; i < h->traces.size(); ++i)
            for (int k = 0; k < n_data; ++k) out[i * (size_t)n_data + k] = h->traces[i].data.read<double>("(y, " + std::to_string(k) + ")");
    })
}
int32_t oracle_mh_read_logjp(oracle_mh* h, double* out) { GUARD({ for (size_t i = 0; i < h->traces.size(); ++i) out[i] = h->traces[i].logjp; }) }
int32_t oracle_mh_destroy(oracle_mh* h) { delete h; return MP_OK; }

// ---- pointed 2-D model under mh (tests/mh.rs:50-68) ---------------------------------------------
struct oracle_mh_pointed {
    Pointed2D pm;
    std::vector<Pointed2D::TraceT> traces;
    uint64_t seed = 0, iters = 0;
    bool canonical = false;
};
static Mat mat2(const double* m) { return Mat(2, std::vector<double>{m[0], m[1], m[2], m[3]}); }
int32_t oracle_mh_pointed_create(const double* bounds, const double* cov, const double* obs, uint64_t n_chains, uint64_t seed, int32_t canon,
                                 oracle_mh_pointed** out) {
    GUARD({
        auto h = std::make_unique<oracle_mh_pointed>();
        h->seed = seed; h->canonical = canon != 0;
        oracle_pf::Scope scope(h->canonical);
        const Bounds b{bounds[0], bounds[1], bounds[2], bounds[3]};
        for (uint64_t i = 0; i < n_chains; ++i) {
            DynTrie observations;  // tests/mh.rs:58-59
            observations.observe("obs", arc(Vec{obs[0], obs[1]}));
            Rng r; r.seed = seed; r.slot = (uint32_t)i; r.step = 0;
            h->traces.push_back(h->pm.model.generate(r, {b, mat2(cov)}, observations).first);  // :61
        }
        *out = h.release();
    })
}
int32_t oracle_mh_pointed_step(oracle_mh_pointed* h, const double* noise, int32_t n_iters, uint64_t* accepted) {
    GUARD({
        oracle_pf::Scope scope(h->canonical);
        auto proposal = h->pm.drift_proposal();
        uint64_t acc = 0;
        for (size_t i = 0; i < h->traces.size(); ++i) {
            for (int k = 0; k < n_iters; ++k) {
                Rng r; r.seed = h->seed; r.slot = (uint32_t)i; r.step = (uint32_t)(h->iters + 1 + (uint64_t)k);
                auto [tr, ok] = metropolis_hastings<Pointed2D::Args, Vec, Mat>(r, h->pm.model, std::move(h->traces[i]), proposal, mat2(noise));  // :64
                h->traces[i] = std::move(tr);
                acc += ok;
            }
        }
        h->iters += (uint64_t)n_iters;
        if (accepted) *accepted = acc;
    })
}
int32_t oracle_mh_pointed_read_state(oracle_mh_pointed* h, double* out) {
    GUARD({
        for (size_t i = 0; i < h->traces.size(); ++i) {
            const Vec v = h->traces[i].data.read<Vec>("latent");
            out[2 * i] = v[0]; out[2 * i + 1] = v[1];
        }
    })
}
int32_t oracle_mh_pointed_read_logjp(oracle_mh_pointed* h, double* out) { GUARD({ for (size_t i = 0; i < h->traces.size(); ++i) out[i] = h->traces[i].logjp; }) }
int32_t oracle_mh_pointed_destroy(oracle_mh_pointed* h) { delete h; return MP_OK; }

// ---- math / rng / distribution probes ---------------------------------------------------------
void oracle_mp_exp(const double* x, int64_t n, double* out) { for (int64_t i = 0; i < n; ++i) out[i] = mp_exp(x[i]); }
// mp_div_hoisted (mp_math.h) on the host: the product's division by a hoisted constant, for the bit-for-bit check against `/`
void oracle_mp_div_hoisted(const double* x, const double* d, int64_t n, double* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = mp_rcp_hoistable(d[i]) ? mp_div_hoisted(x[i], d[i], 1.0 / d[i]) : x[i] / d[i];
}
// out[0][i] = mp_normal_logpdf_h (hoisted reciprocal, no division), out[1][i] = mp_normal_logpdf_ln (the division): must be equal bits
void oracle_mp_normal_logpdf_both(const double* x, const double* mu, const double* sd, int64_t n, double* out_h, double* out_div) {
    for (int64_t i = 0; i < n; ++i) {
        const double ln = mp_log(sd[i]);
        out_h[i] = mp_normal_logpdf_h(x[i], mu[i], sd[i], ln, mp_rcp_hoist(sd[i]));
        out_div[i] = mp_normal_logpdf_ln(x[i], mu[i], sd[i], ln);
    }
}
// Binomial(n, a / b) of the split multinomial's rank counts: the restatement (inference.hpp) and the product's own header
// (mp_binomial.h) compiled for the host, case by case — must be the same integers
void oracle_binomial_both(const uint64_t* n, const uint64_t* a, const uint64_t* b, const uint32_t* node, int64_t cases, uint64_t seed, uint32_t rc,
                          uint64_t* out_restated, uint64_t* out_product) {
    for (int64_t i = 0; i < cases; ++i) {
        out_restated[i] = oracle::canonical_binomial(n[i], a[i], b[i], seed, rc, node[i]);
        out_product[i] = mp_binomial_ratio(n[i], a[i], b[i], node[i], rc, (uint32_t)seed, (uint32_t)(seed >> 32));
    }
}
// the product header's lane-wise form of the same variate (what the device's eight lanes per tree node compute), host-compiled
void oracle_binomial_lanes(const uint64_t* n, const uint64_t* a, const uint64_t* b, const uint32_t* node, int64_t cases, uint64_t seed, uint32_t rc,
                           uint64_t* out) {
    for (int64_t i = 0; i < cases; ++i) out[i] = mp_binomial_ratio_lanes(n[i], a[i], b[i], node[i], rc, (uint32_t)seed, (uint32_t)(seed >> 32));
}
void oracle_split_counts(const uint64_t* mass, int32_t world, uint64_t n_global, uint64_t seed, uint32_t rc, uint64_t* out) {
    const std::vector<uint64_t> c = oracle::canonical_split_counts(std::vector<uint64_t>(mass, mass + world), n_global, seed, rc);
    for (int32_t r = 0; r < world; ++r) out[r] = c[(size_t)r];
}
double oracle_stirling_tail(double k) { return oracle::canonical_stirling_tail(k); }
void oracle_mp_log(const double* x, int64_t n, double* out) { for (int64_t i = 0; i < n; ++i) out[i] = mp_log(x[i]); }
void oracle_mp_sin(const double* x, int64_t n, double* out) { for (int64_t i = 0; i < n; ++i) out[i] = mp_sin(x[i]); }
void oracle_mp_cos(const double* x, int64_t n, double* out) { for (int64_t i = 0; i < n; ++i) out[i] = mp_cos(x[i]); }
void oracle_mp_atan2(const double* y, const double* x, int64_t n, double* out) { for (int64_t i = 0; i < n; ++i) out[i] = mp_atan2(y[i], x[i]); }
void oracle_philox(const uint32_t* ctr, const uint32_t* key, uint32_t* out) { philox4x32_10(ctr, key, out); }
void oracle_u01_stream(uint64_t seed, uint32_t slot, uint32_t step, uint32_t domain, uint32_t site, int64_t n, double* out) {
    Rng r; r.seed = seed; r.slot = slot; r.step = step; r.at(domain, site);
    for (int64_t i = 0; i < n; ++i) out[i] = r.u01();
}
double oracle_logsumexp(const double* x, int64_t n, int32_t canon) {
    oracle_pf::Scope s(canon); return logsumexp(std::vector<double>(x, x + n));
}
double oracle_normal_logpdf(double x, double mu, double std_, int32_t canon) { oracle_pf::Scope s(canon); return normal.logpdf(x, {mu, std_}); }
double oracle_normal_random(uint64_t seed, uint32_t slot, uint32_t step, uint32_t domain, uint32_t site, double mu, double std_, int32_t canon) {
    oracle_pf::Scope s(canon); Rng r; r.seed = seed; r.slot = slot; r.step = step; r.at(domain, site); return normal.random(r, {mu, std_});
}
double oracle_uniform_logpdf(double x, double a, double b) { try { return uniform.logpdf(x, {a, b}); } catch (const Panic&) { return NAN; } }
double oracle_bernoulli_logpdf(int32_t v, double p) { return bernoulli.logpdf(v != 0, p); }
double oracle_uniform2d_logpdf(double x, double y, double xmin, double xmax, double ymin, double ymax) {
    return uniform_2d.logpdf(Vec{x, y}, Bounds{xmin, xmax, ymin, ymax});
}
double oracle_mvnormal_logpdf(int32_t k, const double* x, const double* mu, const double* cov) {
    return mvnormal.logpdf(Vec(x, x + k), MvNormalParams{Vec(mu, mu + k), Mat(k, Vec(cov, cov + k * k))});
}
int64_t oracle_categorical_scan(double u, const double* probs, int64_t n) {
    try { return Categorical::scan(u, std::vector<double>(probs, probs + n)); } catch (const Panic&) { return -2; }
}
void oracle_mvnormal_random(uint64_t seed, uint32_t slot, int32_t k, const double* mu, const double* cov, double* out) {
    Rng r; r.seed = seed; r.slot = slot; r.at(DOM_MODEL, 0);
    Vec v = mvnormal.random(r, MvNormalParams{Vec(mu, mu + k), Mat(k, Vec(cov, cov + k * k))});
    for (int i = 0; i < k; ++i) out[i] = v[(size_t)i];
}
// the dense forms (dense_dot: multiply-then-add when canon == 0, the matrix cores' fma chain when 1), any slot / step / site
double oracle_mvnormal_logpdf_dense(int32_t k, const double* x, const double* mu, const double* cov, int32_t canon) {
    oracle_pf::Scope s(canon);
    return mvnormal.logpdf_dense(Vec(x, x + k), MvNormalParams{Vec(mu, mu + k), Mat(k, Vec(cov, cov + k * k))});
}
void oracle_mvnormal_random_dense(uint64_t seed, uint32_t slot, uint32_t step, uint32_t domain, uint32_t site, int32_t k, const double* mu,
                                  const double* cov, int32_t canon, double* out) {
    oracle_pf::Scope s(canon);
    Rng r; r.seed = seed; r.slot = slot; r.step = step; r.at(domain, site);
    Vec v = mvnormal.random_dense(r, MvNormalParams{Vec(mu, mu + k), Mat(k, Vec(cov, cov + k * k))});
    for (int i = 0; i < k; ++i) out[i] = v[(size_t)i];
}
// canonical resampling spec, exposed piecewise
int32_t oracle_canonical_normalize(const double* logw, int64_t n, uint64_t n_global, double* L, double* ess, uint64_t* Q, uint64_t* cum) {
    CanonFull c = canonical_normalize(std::vector<double>(logw, logw + n), n_global);
    *L = c.c.L; *ess = c.c.ess; *Q = c.c.Q;
    if (cum) std::memcpy(cum, c.t.cum.data(), (size_t)n * sizeof(uint64_t));  // tile-local inclusive prefixes
    return c.c.degenerate() ? MP_ERR_DEGENERATE : MP_OK;
}
uint64_t oracle_canonical_target(uint64_t k52, uint64_t Q) { return canonical_target(k52, Q); }
double oracle_kalman_log_ml(const double* params, const double* ys, int32_t T) {
    return kalman_log_ml(LgssmParams{params[0], params[1], params[2], params[3], params[4]}, Vec(ys, ys + T));
}
void oracle_lgssm_simulate_observations(const double* params, uint64_t seed, int32_t T, int32_t canon, double* out) {
    oracle_pf::Scope s(canon);
    Vec ys = lgssm_simulate_observations(LgssmParams{params[0], params[1], params[2], params[3], params[4]}, seed, T);
    std::memcpy(out, ys.data(), (size_t)T * sizeof(double));
}
double oracle_hmm_forward(const double* params, int32_t n_params, const double* obs, int32_t T) {
    HmmParams p = hmm_from_params(params, n_params);
    std::vector<size_t> o; for (int t = 0; t < T; ++t) o.push_back((size_t)obs[t]);
    return hmm_forward_alg(p, o);
}

}  // extern "C"

#include "kats.hpp"
