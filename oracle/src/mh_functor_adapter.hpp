// ORACLE — TEST INFRASTRUCTURE ONLY (see rng.hpp).
//
// mh_functor_adapter.hpp — runs the model and proposal functors of modppl_amd/csrc/mp_mh_models.h (the ONE source of an MH
// model added through MP_REGISTER_MH_MODEL / MP_REGISTER_MH_PROPOSAL) against the checker's OWN dynamic machinery: tries,
// DynGenFnHandler::sample_at / trace_at / gc (dyngenfn.hpp: the restated dyngenfn.rs:100-483 with its weight and discard
// bookkeeping), metropolis_hastings / regenerative_metropolis_hastings (inference.hpp: mh.rs:9-67), dists.hpp's
// distributions and rng.hpp's Philox.  What the `dyngen!` macro would have expanded a body to (modppl-macros/src/lib.rs:20-113):
//   g.template normal<SITE>(mu, sd, ln_sd)   ->  g.sample_at(normal, {mu, sd}, "s<SITE>")
//   g.template bernoulli<SITE>(p)            ->  g.sample_at(bernoulli, p, "s<SITE>")
//   g.template call<SITES>(body)             ->  g.trace_at(DynGenFn{body}, (), "c<SITES>")   a real sub-trace with its own trie
// A proposal addresses the model's sites by their full paths ("c<SITES>/s<SITE>" for a site inside a sub-call:
// M::sub_of(site) tells) and reads the trace it is given through a dense view (val[], has(), get()).
// Shared with the product: the functor BODIES (which sites, in which order, with which parameters).  Not shared: everything
// that interprets them — in particular none of mp_genfn.h's handler rules, which are what the GPU tests check against this.
#pragma once
#include <functional>
#include <map>
#include <memory>

#include "inference.hpp"
#include "../../modppl_amd/csrc/mp_genfn.h"   // mp_fn_ret (the functor contract); none of its handlers is used here

namespace oracle {

inline std::string mhfn_local_addr(int site) { return "s" + std::to_string(site); }
inline std::string mhfn_sub_addr(uint64_t sites) { return "c" + std::to_string(sites); }
// models with two levels of sub-calls also say which call encloses a site's innermost one (M::outer_of: its SITES, or 0)
template <class M, class = void>
struct mhfn_has_outer : std::false_type {};
template <class M>
struct mhfn_has_outer<M, std::void_t<decltype(M::outer_of(0))>> : std::true_type {};
template <class M>
uint64_t mhfn_outer(int s) {
    if constexpr (mhfn_has_outer<M>::value) return M::outer_of(s);
    else return 0u;
}
template <class M>
std::string mhfn_flat_addr(int site) {
    const uint64_t sub = M::sub_of(site), outer = mhfn_outer<M>(site);
    std::string a = outer ? mhfn_sub_addr(outer) + "/" : std::string();
    if (sub) a += mhfn_sub_addr(sub) + "/";
    return a + mhfn_local_addr(site);
}
inline uint32_t mhfn_site_of(const std::string& a) {   // the Philox site of an address: the id after the last 's'
    return (uint32_t)std::stoul(a.substr(a.rfind('s') + 1));
}

// how many slots a site's value takes: models with vector-valued sites say so (M::dim_of: 2 for the head of a 2-vector, 0 for
// its second slot), every other model's sites are scalars
template <class M, class = void>
struct mhfn_has_dim : std::false_type {};
template <class M>
struct mhfn_has_dim<M, std::void_t<decltype(M::dim_of(0))>> : std::true_type {};
template <class M>
int mhfn_dim(int s) {
    if constexpr (mhfn_has_dim<M>::value) return M::dim_of(s);
    else return 1;
}

using MhFnGen = DynGenFn<int, mp_fn_ret>;      // args = () as an int; retv = the body's mp_fn_ret (mp_genfn.h)
using MhFnH = DynGenFnHandler<int, mp_fn_ret>;
using MhFnTrace = Trace<int, DynTrie, mp_fn_ret>;

// the handler a functor body sees: top level (a proposal, or the model's own frame) or inside one sub-call
template <class M>
struct DynMhHandler {
    MhFnH& g;
    bool inside;      // in a sub-call the addresses are local to its trie
    bool flat;        // (unused for models: their top-level sites are local to the model's own trie)
    double exp_(double x) const { return o_exp(x); }
    double log_(double x) const { return o_ln(x); }
    template <int SITE>
    std::string addr() const { return (flat && !inside) ? mhfn_flat_addr<M>(SITE) : mhfn_local_addr(SITE); }
    template <int SITE>
    double normal(double mu, double sd, double /*ln_sd hoisted by the device form*/) {
        return g.template sample_at<double>(oracle::normal, NormalParams{mu, sd}, addr<SITE>());
    }
    template <int SITE>
    double normal(double mu, double sd) { return normal<SITE>(mu, sd, 0.); }
    template <int SITE>
    bool bernoulli(double p) { return g.template sample_at<bool>(oracle::bernoulli, p, addr<SITE>()); }
    template <int SITE>
    double uniform(double a, double b) { return g.template sample_at<double>(oracle::uniform, UniformParams{a, b}, addr<SITE>()); }
    // vector-valued sites: one address, a Vec value (the checker's own uniform_2d / mvnormal, the latter from the covariance itself)
    template <int SITE>
    void uniform_2d(double xmin, double xmax, double ymin, double ymax, double /*neg_ln_area: hoisted by the device form*/, double* out) {
        const Vec v = g.template sample_at<Vec>(oracle::uniform_2d, Bounds{xmin, xmax, ymin, ymax}, addr<SITE>());
        out[0] = v[0]; out[1] = v[1];
    }
    template <int SITE>
    void mvnormal2(const double* mu, const double* cov, const double* /*cov_inv*/, double /*ln_det*/, const double* /*chol*/, double* out) {
        const Vec v = g.template sample_at<Vec>(oracle::mvnormal, MvNormalParams{Vec{mu[0], mu[1]}, Mat(2, {cov[0], cov[1], cov[2], cov[3]})}, addr<SITE>());
        out[0] = v[0]; out[1] = v[1];
    }

    // declared data sites (mp_genfn.h): here they are what they are in the reference — `normal(mu_j, sd_j) %= ("y", j)` in a loop, ordinary
    // sample_at calls on the trie at address s<NS + j>, with the trie's own stored weights for the previous log-densities
    template <class Model, class L>
    void data(const Model& m, const L& lat) {
        for (int j = 0; j < m.n_obs; ++j) {
            const mp_fn_normal d = m.datum(j, lat);
            (void)g.template sample_at<double>(oracle::normal, NormalParams{d.mu, d.sd}, mhfn_local_addr(M::NS + j));
        }
    }
    // trace_at: under Update / Regenerate with nothing touched and diff NoChange the body is NOT run and the stored retv comes
    // back (dyngenfn.rs:362-366, 415-419) — which is why a functor may take a sub-call's results from its return value only
    template <uint64_t SITES, class Body>
    mp_fn_ret call(Body&& body) {
        const uint32_t dom = g.domain;
        MhFnGen sub([&body](MhFnH& g2, int) -> mp_fn_ret {
            DynMhHandler<M> h2{g2, true, false};
            return body(h2);
        }, mhfn_site_of, dom);
        return g.template trace_at<int, mp_fn_ret>(sub, 0, mhfn_sub_addr(SITES));
    }
};

// dense view of a trie trace for proposal bodies (tr.val[SITE], tr.has(SITE), tr.get(SITE, dflt)) and for the test's reads
template <class M>
struct MhFnView {
    double val[M::NS];
    uint64_t present = 0;   // one bit per site (up to 64 sites)
    bool has(int site) const { return (present >> site) & 1u; }
    double get(int site, double dflt) const { return has(site) ? val[site] : dflt; }
    explicit MhFnView(const DynTrie& data) {
        for (int s = 0; s < M::NS; ++s) val[s] = 0.;
        for (int s = 0; s < M::NS; ++s) {
            const uint64_t sub = M::sub_of(s), outer = mhfn_outer<M>(s);
            const Trie* node = nullptr;
            const Trie* at = &data;
            if (outer) at = at->search(mhfn_sub_addr(outer));   // (term searches: a missing sub-trie is "absent", not a panic)
            if (at && sub) at = at->search(mhfn_sub_addr(sub));
            node = at ? at->search(mhfn_local_addr(s)) : nullptr;
            if (!node || !node->value) continue;
            if (mhfn_dim<M>(s) == 0) continue;   // (a vector's further slot: filled with its head)
            present |= 1ull << s;
            if (const double* d = std::any_cast<double>(node->value->get())) val[s] = *d;
            else if (const bool* b = std::any_cast<bool>(node->value->get())) val[s] = *b ? 1. : 0.;
            else if (const Vec* v = std::any_cast<Vec>(node->value->get())) {
                if ((int)v->size() != mhfn_dim<M>(s)) throw Panic("mh functor adapter: a vector choice of the wrong length at site " + std::to_string(s));
                for (size_t j = 0; j < v->size(); ++j) { val[s + (int)j] = (*v)[j]; present |= 1ull << (s + (int)j); }
            } else throw Panic("mh functor adapter: a choice that is neither f64, bool nor Vec at site " + std::to_string(s));
        }
    }
};

using MhFnPArgs = std::pair<const MhFnTrace*, int>;
using MhFnProposal = DynGenFn<MhFnPArgs, int>;
// a proposal's frame: full paths, no sub-calls (a proposal addresses the model's sites directly, hierarchical.rs:48-70)
template <class M>
struct MhFnFlatHandler {
    DynGenFnHandler<MhFnPArgs, int>& g;
    template <int SITE>
    double normal(double mu, double sd, double) { return g.template sample_at<double>(oracle::normal, NormalParams{mu, sd}, mhfn_flat_addr<M>(SITE)); }
    template <int SITE>
    double normal(double mu, double sd) { return normal<SITE>(mu, sd, 0.); }
    template <int SITE>
    bool bernoulli(double q) { return g.template sample_at<bool>(oracle::bernoulli, q, mhfn_flat_addr<M>(SITE)); }
    template <int SITE>
    double uniform(double a, double b) { return g.template sample_at<double>(oracle::uniform, UniformParams{a, b}, mhfn_flat_addr<M>(SITE)); }
    // vector-valued sites: one address, a Vec value (the checker's own uniform_2d / mvnormal, the latter from the covariance itself)
    template <int SITE>
    void uniform_2d(double xmin, double xmax, double ymin, double ymax, double /*neg_ln_area: hoisted by the device form*/, double* out) {
        const Vec v = g.template sample_at<Vec>(oracle::uniform_2d, Bounds{xmin, xmax, ymin, ymax}, mhfn_flat_addr<M>(SITE));
        out[0] = v[0]; out[1] = v[1];
    }
    template <int SITE>
    void mvnormal2(const double* mu, const double* cov, const double* /*cov_inv*/, double /*ln_det*/, const double* /*chol*/, double* out) {
        const Vec v = g.template sample_at<Vec>(oracle::mvnormal, MvNormalParams{Vec{mu[0], mu[1]}, Mat(2, {cov[0], cov[1], cov[2], cov[3]})}, mhfn_flat_addr<M>(SITE));
        out[0] = v[0]; out[1] = v[1];
    }

    double exp_(double x) const { return o_exp(x); }
    double log_(double x) const { return o_ln(x); }
};

struct MhFnModel {
    virtual ~MhFnModel() {}
    virtual int ns() const = 0;
    virtual const MhFnGen& model() const = 0;
    virtual DynTrie constraints(const int32_t* sites, const double* vals, int n) const = 0;
    virtual std::string flat_addr(int site) const = 0;
    virtual MhFnProposal proposal(int kind, const double* args, int n_args) const = 0;
    virtual void view(const DynTrie& data, double* vals, uint64_t* present) const = 0;
};
template <class M>
using MhFnProposalFactory = std::function<MhFnProposal(const double*, int)>;
template <class M>
std::map<int, MhFnProposalFactory<M>>& mhfn_proposals() {
    static std::map<int, MhFnProposalFactory<M>> r;
    return r;
}
template <class M>
struct MhFnModelT : MhFnModel {
    M m;
    MhFnGen gen;
    std::vector<double> cov;   // (declared data sites) the covariates the functor's datum() reads; the observed values live in the tries
    explicit MhFnModelT(const M& m_, const double* params = nullptr, int n_params = 0) : m(m_) {
        if constexpr (mp_fn_has_data<M>::value) {
            cov.assign(params, params + n_params);
            m.bind(cov.data(), nullptr);
        }
        const M mm = m;
        gen = MhFnGen([mm](MhFnH& g, int) -> mp_fn_ret {
            DynMhHandler<M> h{g, false, false};
            mm(h);
            return mp_fn_ret{};
        }, mhfn_site_of, DOM_MODEL);
    }
    int ns() const override { return M::NS; }
    const MhFnGen& model() const override { return gen; }
    std::string flat_addr(int site) const override { return mhfn_flat_addr<M>(site); }
    DynTrie constraints(const int32_t* sites, const double* vals, int n) const override {
        DynTrie c;
        for (int q = 0; q < n; ++q) {
            const int s = sites[q];
            if constexpr (mp_fn_has_data<M>::value) {
                if (s >= M::NS && s < M::NS + m.n_obs) { c.observe(mhfn_local_addr(s), arc(vals[q])); continue; }   // observation s - NS
            }
            if (s < 0 || s >= M::NS) throw Panic("constraint site out of range");
            const int d = mhfn_dim<M>(s);
            if (d == 0) continue;   // a vector's further slot: taken with its head below
            if (d > 1) {            // a vector-valued site: its d slots make ONE choice at the head's address
                Vec v((size_t)d);
                for (int j = 0; j < d; ++j) {
                    int at = -1;
                    for (int q2 = 0; q2 < n; ++q2) if (sites[q2] == s + j) at = q2;
                    if (at < 0) throw Panic("constraint on a vector-valued site must name all of its slots");
                    v[(size_t)j] = vals[at];
                }
                c.observe(mhfn_flat_addr<M>(s), arc(v));
            } else if (M::is_bool(s)) c.observe(mhfn_flat_addr<M>(s), arc(vals[q] != 0.));
            else c.observe(mhfn_flat_addr<M>(s), arc(vals[q]));
        }
        return c;
    }
    MhFnProposal proposal(int kind, const double* args, int n_args) const override {
        auto it = mhfn_proposals<M>().find(kind);
        if (it == mhfn_proposals<M>().end()) throw Panic("no proposal of this kind is registered for the model");
        return it->second(args, n_args);
    }
    void view(const DynTrie& data, double* vals, uint64_t* present) const override {
        const MhFnView<M> v(data);
        for (int s = 0; s < M::NS; ++s) vals[s] = v.val[s];
        *present = v.present;
    }
};

// ---- the PRODUCT's static handlers (mp_genfn.h: mp_fn_handler<NS, MODE>) compiled for the host and run chain by chain — not a
// checker but a checkee: the same code the k_fn_* kernels run per lane, so that the suite without a GPU can hold the handler rules
// against the dynamic machinery above as well (tests/test_oracle_mh_functor.py).  Canonical arithmetic by construction (mp_math.h).
struct MhFnStatic {
    virtual ~MhFnStatic() {}
    virtual int ns() const = 0;
    virtual void create(uint64_t n_chains, uint64_t seed, const int32_t* sites, const double* vals, int n_cons) = 0;
    virtual uint64_t mh(int kind, const double* args, int n_args, int n_iters) = 0;
    virtual uint64_t regen(const int32_t* mask_sites, int n_mask, int cycle, int n_iters) = 0;
    virtual void read(double* vals, uint64_t* present) const = 0;
    virtual uint64_t panics() const = 0;
    // mp_fn_update's per-lane work (k_fn_update) with constraints shared by all chains: weights and discard presence out
    virtual void update(const int32_t* sites, const double* vals, int n_cons, int unknown, uint32_t step, double* weights, uint64_t* disc_present) = 0;
};
template <class M>
struct MhFnStaticT;
template <class M>
using MhFnStaticProposal = std::function<uint64_t(MhFnStaticT<M>&, const double*, int, int)>;
template <class M>
std::map<int, MhFnStaticProposal<M>>& mhfn_static_proposals() {
    static std::map<int, MhFnStaticProposal<M>> r;
    return r;
}
template <class M>
struct MhFnStaticT : MhFnStatic {
    M model;
    std::vector<mp_fn_trace<M::NS>> tr;
    uint64_t seed = 0, iters = 0, n_panic = 0;
    std::vector<double> cov, obsv;   // (declared data sites) the shared arrays the product's handler reads
    explicit MhFnStaticT(const M& m, const double* params = nullptr, int n_params = 0) : model(m) {
        if constexpr (mp_fn_has_data<M>::value) cov.assign(params, params + n_params);
    }
    int ns() const override { return M::NS; }
    uint64_t panics() const override { return n_panic; }
    mp_stream stream(size_t i, uint32_t step) const {
        mp_stream s;
        s.k0 = (uint32_t)seed; s.k1 = (uint32_t)(seed >> 32); s.slot = (uint32_t)i; s.step = step;
        return s;
    }
    void create(uint64_t n_chains, uint64_t seed_, const int32_t* sites, const double* vals, int n_cons) override {
        seed = seed_; iters = 0;
        mp_fn_trace<M::NS> c;
        mp_fn_clear(c);
        if constexpr (mp_fn_has_data<M>::value) {
            obsv.assign((size_t)model.n_obs, 0.);
            model.bind(cov.data(), obsv.data());
        }
        for (int q = 0; q < n_cons; ++q) {
            if constexpr (mp_fn_has_data<M>::value) {
                if (sites[q] >= M::NS) { obsv[(size_t)(sites[q] - M::NS)] = vals[q]; continue; }
            }
            c.present |= mp_fn_bits_t<M::NS>(1) << sites[q]; c.val[sites[q]] = vals[q];
        }
        tr.resize(n_chains);
        for (size_t i = 0; i < tr.size(); ++i) {   // k_fn_init
            const mp_stream s = stream(i, 0);
            mp_fn_handler<M::NS, MP_FN_GENERATE, M> g(s, MP_DOM_MODEL, nullptr, &c);
            model(g);
            g.finish();
            n_panic += g.panic;
            tr[i] = g.tr;
        }
    }
    uint64_t regen(const int32_t* mask_sites, int n_mask, int cycle, int n_iters) override {   // k_fn_regen
        uint64_t acc = 0;
        using bits_t = mp_fn_bits_t<M::NS>;
        bits_t bits = 0;
        for (int q = 0; q < n_mask; ++q) bits |= bits_t(1) << mask_sites[q];
        for (size_t i = 0; i < tr.size(); ++i) {
            mp_fn_trace<M::NS> cur = tr[i];
            for (int it = 0; it < n_iters; ++it) {
                const mp_stream s = stream(i, (uint32_t)(iters + 1 + (uint64_t)it));
                bits_t m = (cycle && n_mask > 0) ? bits_t(1) << mask_sites[(iters + (uint64_t)it) % (uint64_t)n_mask] : bits;
                if (m == 0u) m = cur.present;
                mp_fn_handler<M::NS, MP_FN_REGENERATE, M> g(s, MP_DOM_MODEL, &cur, nullptr, m);
                model(g);
                g.finish();
                n_panic += g.panic;
                const mp_u64x2 ub = s.draw(MP_DOM_ACCEPT, 0u, 0u);
                if (mp_log(mp_u01(ub.a)) < g.weight) { cur = g.tr; ++acc; }
            }
            tr[i] = cur;
        }
        iters += (uint64_t)n_iters;
        return acc;
    }
    template <class P>
    uint64_t mh_with(const P& proposal, int n_iters) {   // k_fn_mh
        uint64_t acc = 0;
        for (size_t i = 0; i < tr.size(); ++i) {
            mp_fn_trace<M::NS> cur = tr[i];
            for (int it = 0; it < n_iters; ++it) {
                const mp_stream s = stream(i, (uint32_t)(iters + 1 + (uint64_t)it));
                mp_fn_handler<M::NS, MP_FN_SIMULATE, M> p(s, MP_DOM_PROPOSAL, nullptr, nullptr);
                proposal(p, cur);
                const double fwd = p.weight;
                mp_fn_handler<M::NS, MP_FN_UPDATE, M> g(s, MP_DOM_MODEL, &cur, &p.tr);
                model(g);
                g.finish();
                mp_fn_trace<M::NS> disc = cur;
                disc.present = g.discarded;
                mp_fn_handler<M::NS, MP_FN_GENERATE, M> q(s, MP_DOM_PROPOSAL, nullptr, &disc);
                proposal(q, g.tr);
                q.finish();
                n_panic += (g.panic || q.panic);
                const double alpha = g.weight - fwd + q.weight;
                const mp_u64x2 ub = s.draw(MP_DOM_ACCEPT, 0u, 0u);
                if (mp_log(mp_u01(ub.a)) < alpha) { cur = g.tr; ++acc; }
            }
            tr[i] = cur;
        }
        iters += (uint64_t)n_iters;
        return acc;
    }
    uint64_t mh(int kind, const double* args, int n_args, int n_iters) override {
        auto it = mhfn_static_proposals<M>().find(kind);
        if (it == mhfn_static_proposals<M>().end()) throw Panic("no proposal of this kind is registered for the model");
        return it->second(*this, args, n_args, n_iters);
    }
    void update(const int32_t* sites, const double* vals, int n_cons, int unknown, uint32_t step, double* weights, uint64_t* disc_present) override {
        mp_fn_trace<M::NS> c;
        mp_fn_clear(c);
        for (int q = 0; q < n_cons; ++q) { c.present |= mp_fn_bits_t<M::NS>(1) << sites[q]; c.val[sites[q]] = vals[q]; }
        for (size_t i = 0; i < tr.size(); ++i) {
            const mp_stream s = stream(i, step);
            mp_fn_handler<M::NS, MP_FN_UPDATE, M> g(s, MP_DOM_MODEL, &tr[i], &c);
            g.changed = unknown != 0;
            model(g);
            g.finish();
            n_panic += g.panic;
            weights[i] = g.weight;
            disc_present[i] = g.discarded;
            tr[i] = g.tr;
        }
    }
    void read(double* vals, uint64_t* present) const override {
        for (size_t i = 0; i < tr.size(); ++i) {
            present[i] = tr[i].present;
            for (int k = 0; k < M::NS; ++k) vals[i * (size_t)M::NS + k] = tr[i].has(k) ? tr[i].val[k] : 0.;
        }
    }
};
using MhFnStaticFactory = std::function<std::shared_ptr<MhFnStatic>(const double*, int)>;
inline std::map<int, MhFnStaticFactory>& mhfn_static_models() {
    static std::map<int, MhFnStaticFactory> r;
    return r;
}

using MhFnFactory = std::function<std::shared_ptr<MhFnModel>(const double*, int)>;
inline std::map<int, MhFnFactory>& mhfn_models() {
    static std::map<int, MhFnFactory> r;
    return r;
}
template <class M>
int mhfn_register_model(int kind, bool (*parse)(const double*, int, M&, std::string&)) {
    mhfn_models()[kind] = [parse](const double* params, int n) -> std::shared_ptr<MhFnModel> {
        M m{};
        std::string err;
        if (!parse(params, n, m, err)) throw Panic(err);
        return std::make_shared<MhFnModelT<M>>(m, params, n);
    };
    mhfn_static_models()[kind] = [parse](const double* params, int n) -> std::shared_ptr<MhFnStatic> {
        M m{};
        std::string err;
        if (!parse(params, n, m, err)) throw Panic(err);
        return std::make_shared<MhFnStaticT<M>>(m, params, n);
    };
    return kind;
}
template <class M, class P>
int mhfn_register_proposal(int kind, bool (*parse)(const double*, int, P&, std::string&)) {
    mhfn_proposals<M>()[kind] = [parse](const double* args, int n_args) -> MhFnProposal {
        P p{};
        std::string err;
        if (!parse(args, n_args, p, err)) throw Panic(err);
        return MhFnProposal([p](DynGenFnHandler<MhFnPArgs, int>& g, MhFnPArgs pa) -> int {
            MhFnFlatHandler<M> h{g};
            const MhFnView<M> tr(pa.first->data);
            p(h, tr);
            return 0;
        }, mhfn_site_of, DOM_PROPOSAL);
    };
    mhfn_static_proposals<M>()[kind] = [parse](MhFnStaticT<M>& r, const double* args, int n_args, int n_iters) -> uint64_t {
        P p{};
        std::string err;
        if (!parse(args, n_args, p, err)) throw Panic(err);
        return r.mh_with(p, n_iters);
    };
    return kind;
}

}  // namespace oracle

// the product's MH model sources, interpreted by the registrars above
#define MP_REGISTER_MH_MODEL(KIND, TYPE, PARSE) static const int oracle_mh_registered_##TYPE = oracle::mhfn_register_model<TYPE>(KIND, PARSE);
#define MP_REGISTER_MH_PROPOSAL(KIND, MODEL, TYPE, PARSE) static const int oracle_mh_registered_##TYPE = oracle::mhfn_register_proposal<MODEL, TYPE>(KIND, PARSE);
#include "../../modppl_amd/csrc/mp_mh_models.h"
