// mp_genfn.h — static generative functions for the MH kernels: ONE functor over a handler type, four interpretations.
//
// In modppl a model or proposal is a closure over `DynGenFnHandler` and `simulate / generate / update / regenerate` are
// four interpretations of it (modppl/src/modeling/dyngenfn.rs:39-93).  Here it is a functor
//
//     template <class H> MP_HD void operator()(H& g) const                      (a model)
//     template <class H, class T> MP_HD void operator()(H& g, const T& tr) const (a proposal: `tr` is the trace it reads)
//
// whose addresses are compile-time site ids 0 .. NS-1 (NS <= 64: the presence / mask words are 32 bits wide up to 32 sites, 64 beyond)
// and whose choices are doubles (a bool is 0 / 1):
//     g.template normal<SITE>(mu, sd, ln_sd)     `normal(mu, sd) %= addr`    -> sample_at (dyngenfn.rs:100-273)
//     g.template bernoulli<SITE>(p)              `bernoulli(p) %= addr`
//     g.template uniform<SITE>(a, b)             `uniform(a, b) %= addr`
//     g.template uniform_2d<SITE>(.., out), g.template mvnormal2<SITE>(mu, cov, .., out)   vector-valued sites: K = 2 consecutive slots
//                                                SITE, SITE + 1 hold the value, SITE is the address (at_k)
//     g.template call<SITES>(body)               `gen_fn(args) /= addr`      -> trace_at  (dyngenfn.rs:283-449);
//                                                SITES = bit set of the sites of the sub-trace, body = [&](H& q) { ...; return mp_fn_ret{...}; }
//                                                What the caller needs from the sub-call travels in that RETURN VALUE, never in
//                                                captured variables: an untouched sub-call under diff NoChange is not executed by
//                                                a dynamic interpreter (trace_at returns the stored retv, dyngenfn.rs:362-366), so a
//                                                body's side effects would be lost there (here the body is replayed: same result).
// A trace is the dense row mp_fn_trace<NS> (value, log-density and a presence bit per site) kept in registers: every site id
// is a template argument, so after inlining each slot is a scalar and the slots a model never touches do not exist.
//
// The handler restates the weight rules of sample_at / trace_at / gc, in the reference's order of operations:
//   SIMULATE   x ~ dist; the trace's score (`propose`'s weight, trace.logjp) += logp                       (:106-114)
//   GENERATE   constrained: x = constraint, weight += logp; free: x ~ dist                                 (:116-141)
//   UPDATE     constrained: weight -= prev.logp (the old choice goes to the discard), weight += logp, diff = Unknown;
//              free with a previous value: diff NoChange -> kept as it is; diff Unknown -> weight += logp - prev.logp;
//              free without one: x ~ dist, diff = Unknown.  gc: unvisited previous sites leave, weight - their logp  (:143-211, 453-470)
//   REGENERATE masked: x ~ dist, diff = Unknown, no weight; unmasked: as a free site of UPDATE; gc without weight   (:213-273)
//   call       the sub-call's weight is accumulated from 0, closed by its own gc, and added to the caller's as ONE term;
//              a constrained / masked / new sub-call sets the caller's diff to Unknown; an untouched one under NoChange is
//              replayed (its body runs, every site returns its previous value: the static form of `return retv`)     (:321-446)
//              `Regenerate` through an UNMASKED sub-call after an upstream change is generate(args, sub) with the old sub-trace as
//              constraints, weight += new_weight - sub.weight() (:424-428): sub.weight() is the sub-trie's RUNNING weight — the
//              history of its inserts and removes, not a fresh sum — so a trace carries it per sub-call (`subw`, indexed by the
//              sub-call's lowest site) and the handler applies the trie's own += / -= to it, in the trie's order (trie.rs:118-184).
// What is NOT restated: leftover constraints (the reference's panic): `panic`, which the kernels report as an error.
//
// DECLARED DATA SITES (round 5).  A model's observations are ordinary sites in the reference (`normal(mu, 0.1) %= ("y", i)` in a loop
// of any length, hierarchical.rs:33-47) whose addresses are in the constraints when the trace is generated and are never masked or
// proposed afterwards.  As register-resident sites they cost a model what its latents cost — a value, a log-density, presence /
// visited / consumed bits, a slot in every accept select — and cap it at MP_FN_MAX_SITES.  A model may instead DECLARE them:
//     static constexpr bool HAS_DATA = true;   int n_obs;            (any number: nothing about them lives in registers)
//     struct latents { ... };                                         what the observations' distributions depend on
//     latents latents_of(const V& view) const;                        ... read off a trace (view.val[site], view.has(site))
//     mp_fn_normal datum(int j, const latents& l) const;              the distribution of observation j
//     double obs(int j) const;                                        its observed value (a shared array: bind(cov, obs))
// and visit them with ONE call `g.data(*this, latents{...})` at the point of the body where the loop over ("y", j) stands.  The handler
// then applies sample_at's rules to each j in order with nothing stored: GENERATE weight += logp; UPDATE / REGENERATE under diff
// Unknown weight += logp_new - logp_prev, where logp_prev — the log-density the previous trace holds for the site — is RECOMPUTED from
// the previous trace's latents (the same expression on the same inputs: the same bits); under NoChange the site is kept as it is.
// Across the C ABI observation j is site id NS + j (constraints of the creating generate only).  What a declared data site cannot do
// is become a chain's own state: simulate, a generate that leaves one unconstrained, and the empty mask of regenerate (the whole
// schema, dyngenfn.rs:571, re-simulates the observed sites too) are MP_ERR_UNSUPPORTED for such models.
#pragma once
#include "mp_dists.h"

#include <type_traits>

#define MP_FN_MAX_SITES 64
// one bit per site: a 32-bit word for models of up to 32 sites (everything the reference's tests need), 64 bits beyond
template <int NS>
using mp_fn_bits_t = typename std::conditional<(NS > 32), uint64_t, uint32_t>::type;

// the return value of a sub-call body (and of a model): up to four doubles
struct mp_fn_ret {
    double v[4];
};

template <int NS>
struct mp_fn_trace {
    double val[NS];
    double lp[NS];
    double subw[NS];   // [lowest site of a sub-call] the running weight of that sub-trie
    mp_fn_bits_t<NS> present;
    MP_HD bool has(int site) const { return ((present >> site) & 1u) != 0u; }
    // tr.data.read(addr) of a proposal body; `dflt` when the address is absent (hierarchical.rs:54-58 `search`)
    MP_HD double get(int site, double dflt) const { return has(site) ? val[site] : dflt; }
};
template <int NS>
MP_HD void mp_fn_clear(mp_fn_trace<NS>& t) {
    t.present = 0;
#pragma unroll
    for (int k = 0; k < NS; ++k) { t.val[k] = 0.; t.lp[k] = 0.; t.subw[k] = 0.; }
}
// trace.logjp: the choices' log-densities in site order
template <int NS>
MP_HD double mp_fn_logjp(const mp_fn_trace<NS>& t) {
    double s = 0.;
    bool first = true;
#pragma unroll
    for (int k = 0; k < NS; ++k)
        if (t.has(k)) { s = first ? t.lp[k] : s + t.lp[k]; first = false; }
    return s;
}

struct mp_fn_normal {
    double mu, sd, ln_sd;
    double rcp_sd;   // mp_rcp_hoist(sd) for a model constant, or 0 (what `mp_fn_normal{mu, sd, ln_sd}` leaves it at): the division (x - mu) / sd then
                     // costs no division — the same bits either way (mp_math.h mp_div_hoisted, held to `/` on host and device)
    MP_HD double sample(mp_site& st) const { return mp_normal_sample(st, mu, sd); }
    MP_HD double logpdf(double x) const { return mp_normal_logpdf_h(x, mu, sd, ln_sd, rcp_sd); }
};
struct mp_fn_bernoulli {
    double p;
    MP_HD double sample(mp_site& st) const { return mp_bernoulli_sample(st, p) ? 1. : 0.; }
    MP_HD double logpdf(double x) const { return mp_bernoulli_logpdf(x != 0., p); }
};

struct mp_fn_uniform {
    double a, b;
    MP_HD double sample(mp_site& st) const { return mp_uniform_sample(st, a, b); }
    MP_HD double logpdf(double x) const { return mp_uniform_logpdf(x, a, b); }
};

// a scalar distribution as a one-value site
template <class D>
struct mp_fn_scalar {
    D d;
    MP_HD void sample(mp_site& st, double* x) const { x[0] = d.sample(st); }
    MP_HD double logpdf(const double* x) const { return d.logpdf(x[0]); }
};
// uniform_2d (modppl/tests/pointed_model/types_2d.rs:14-32): both coordinates from ONE site = the two halves of one Philox block
struct mp_fn_uniform_2d {
    double xmin, xmax, ymin, ymax, neg_ln_area;   // neg_ln_area = -mp_log((xmax - xmin) * (ymax - ymin)), hoisted
    MP_HD void sample(mp_site& st, double* x) const {
        const mp_u64x2 b = st.next_block();
        x[0] = mp_u01(b.a) * (xmax - xmin) + xmin;
        x[1] = mp_u01(b.b) * (ymax - ymin) + ymin;
    }
    MP_HD double logpdf(const double* p) const { return (xmin <= p[0] && p[0] <= xmax && ymin <= p[1] && p[1] <= ymax) ? neg_ln_area : MP_NEG_INF; }
};
// mvnormal of dimension 2 (mvnormal.rs:12-37) with the per-call nalgebra work hoisted: cov_inv (row-major), ln_det, and the lower
// Cholesky factor {l00, l10, l11} for `random` = L z + mu with z_0, z_1 ~ normal(0, 1) one after the other from the site's stream
struct mp_fn_mvnormal2 {
    double mu[2];
    double cov_inv[4];
    double ln_det;
    double l00, l10, l11;
    MP_HD void sample(mp_site& st, double* x) const {
        const double z0 = mp_normal_sample(st, 0., 1.);
        const double z1 = mp_normal_sample(st, 0., 1.);
        x[0] = (0. + l00 * z0) + mu[0];
        x[1] = ((0. + l10 * z0) + l11 * z1) + mu[1];
    }
    MP_HD double logpdf(const double* x) const { return mp_mvnormal_logpdf_pre<2>(x, mu, cov_inv, ln_det); }
};

enum mp_fn_mode { MP_FN_SIMULATE = 0, MP_FN_GENERATE = 1, MP_FN_UPDATE = 2, MP_FN_REGENERATE = 3 };

template <class M, class = void>
struct mp_fn_has_data : std::false_type {};
template <class M>
struct mp_fn_has_data<M, std::void_t<decltype(M::HAS_DATA)>> : std::integral_constant<bool, M::HAS_DATA> {};

// Which sub-call of a frame an unvisited site belongs to, for gc: the LARGEST of the model's sub-call site sets around site k that lies
// strictly inside the frame (the whole schema, or the SITES of the call whose gc this is) — 0 when k is one of the frame's own leaves.
// SUBS = the model type (sub_of(site): the innermost call's sites; outer_of(site), models with two levels: the call around that one).
template <class SUBS, class = void>
struct mp_fn_subs_outer { static constexpr uint64_t of(int) { return 0u; } };
template <class SUBS>
struct mp_fn_subs_outer<SUBS, std::void_t<decltype(SUBS::outer_of(0))>> { static constexpr uint64_t of(int k) { return SUBS::outer_of(k); } };
template <class SUBS>
constexpr uint64_t mp_fn_child_call(int k, uint64_t frame) {
    const uint64_t outer = mp_fn_subs_outer<SUBS>::of(k), inner = SUBS::sub_of(k);
    if (outer && (outer & ~frame) == 0u && outer != frame) return outer;
    if (inner && (inner & ~frame) == 0u && inner != frame) return inner;
    return 0u;
}

template <int NS, int MODE, class SUBS = void>
struct mp_fn_handler {
    static_assert(NS <= MP_FN_MAX_SITES, "site ids are bits of a 64-bit word at most");
    using bits_t = mp_fn_bits_t<NS>;
    const mp_stream& rng;
    uint32_t dom;                     // Philox domain of this function's draws (MP_DOM_MODEL / MP_DOM_PROPOSAL); site id = site
    const mp_fn_trace<NS>* prev;      // UPDATE / REGENERATE: the previous trace
    const mp_fn_trace<NS>* cons;      // GENERATE / UPDATE: the constraints (presence bits + values)
    bits_t mask;                      // REGENERATE: the masked sites
    mp_fn_trace<NS> tr;               // the trace being built
    double weight;                    // SIMULATE: the trace's score
    bool changed;                     // diff == ArgDiff::Unknown
    bits_t visited, consumed, discarded;
    bool panic;
    double sw;            // inside a sub-call: the running weight of its trie
    bool in_sub;          // the sites visited belong to a sub-trie whose weight is being kept
    bool from_prev;       // REGENERATE through an unmasked sub-call after an upstream change: generate with the old choices as constraints
    double dlp;           // GENERATE: the declared data sites' log-densities, summed in order (trace.logjp's share of them)

    MP_HD mp_fn_handler(const mp_stream& r, uint32_t dom_, const mp_fn_trace<NS>* prev_, const mp_fn_trace<NS>* cons_, bits_t mask_ = 0)
        : rng(r), dom(dom_), prev(prev_), cons(cons_), mask(mask_), weight(0.), changed(false), visited(0), consumed(0), discarded(0),
          panic(false), sw(0.), in_sub(false), from_prev(false), dlp(0.) {
        mp_fn_clear(tr);
    }
    // the model's declared data sites (top of this file), all of them, in order; `lat` = what their distributions depend on, as the
    // body has just computed it.  Top level only (not inside a sub-call).
    template <class Model, class L>
    MP_HD void data(const Model& m, const L& lat) {
        if constexpr (MODE == MP_FN_SIMULATE) {
            panic = true;   // an observation drawn from its prior would be per-chain state: not what a declared data site is
        } else if constexpr (MODE == MP_FN_GENERATE) {
            for (int j = 0; j < m.n_obs; ++j) {
                const double lp = m.datum(j, lat).logpdf(m.obs(j));
                weight += lp;                                    // constrained: weight += logp (dyngenfn.rs:116-131)
                dlp = j ? dlp + lp : lp;
            }
        } else {
            if (changed) {                                       // diff Unknown: weight += logp - prev.logp (:180-190, :236-243)
                const L old = m.latents_of(*prev);
                for (int j = 0; j < m.n_obs; ++j) {
                    const double y = m.obs(j);
                    weight += m.datum(j, lat).logpdf(y) - m.datum(j, old).logpdf(y);
                }
            }                                                    // NoChange: every one is kept as it is
        }
    }
    MP_HD double exp_(double x) const { return mp_exp(x); }
    MP_HD double log_(double x) const { return mp_log(x); }

    // One site of K values (K = 1: a scalar choice; K > 1: a vector-valued one, e.g. uniform_2d or a 2-d mvnormal — the value
    // occupies the slots SITE .. SITE + K - 1 of the trace, its log-density sits in lp[SITE], its address is SITE: constraints, masks
    // and the Philox site id name the FIRST slot).  Dist: sample(mp_site&, double* x), logpdf(const double* x).
    template <int SITE, int K, class Dist>
    MP_HD void at_k(const Dist& d, double* x) {
        static_assert(SITE >= 0 && SITE + K <= NS && K >= 1, "site id out of range");
        constexpr bits_t bit = bits_t(1) << SITE;
        constexpr bits_t vbits = (bits_t)((K >= 32 ? uint64_t(0) : (uint64_t(1) << K)) - 1u) << SITE;
        visited |= vbits;
        double lp;
#define MP_FN_PUT_() do { _Pragma("unroll") for (int j_ = 0; j_ < K; ++j_) { tr.val[SITE + j_] = x[j_]; tr.lp[SITE + j_] = 0.; } tr.lp[SITE] = lp; tr.present |= vbits; } while (0)
#define MP_FN_PREV_() do { _Pragma("unroll") for (int j_ = 0; j_ < K; ++j_) x[j_] = prev->val[SITE + j_]; } while (0)
        // (`sw`: what the trie does to the weight of the sub-trie this site lives in — remove: -= the old choice's, w_observe / insert:
        // += the new one's, trie.rs:118-184 — in the order sample_at does them)
        if constexpr (MODE == MP_FN_SIMULATE) {
            mp_site st(rng, dom, (uint32_t)SITE);
            d.sample(st, x);
            lp = d.logpdf(x);
            weight += lp;
            if (in_sub) sw += lp;
        } else if constexpr (MODE == MP_FN_GENERATE) {
            if (cons->present & bit) {
                consumed |= vbits;
#pragma unroll
                for (int j_ = 0; j_ < K; ++j_) x[j_] = cons->val[SITE + j_];
                lp = d.logpdf(x);
                weight += lp;
            } else {
                mp_site st(rng, dom, (uint32_t)SITE);
                d.sample(st, x);
                lp = d.logpdf(x);
            }
            if (in_sub) sw += lp;
        } else {
            const bool had = (prev->present & bit) != 0;
            if (from_prev) {   // generate(args, sub): the old choice is the constraint (:116-131); a site the old sub-trace lacks is drawn
                if (had) {
                    MP_FN_PREV_();
                    lp = d.logpdf(x);
                    weight += lp;
                } else {
                    mp_site st(rng, dom, (uint32_t)SITE);
                    d.sample(st, x);
                    lp = d.logpdf(x);
                }
                sw += lp;
                MP_FN_PUT_();
                return;
            }
            bool fresh = false;   // drawn from the distribution
            if constexpr (MODE == MP_FN_UPDATE) {
                if (cons->present & bit) {
                    consumed |= vbits;
                    if (had) {
                        weight -= prev->lp[SITE]; discarded |= vbits;
                        if (in_sub) sw -= prev->lp[SITE];
                    }
#pragma unroll
                    for (int j_ = 0; j_ < K; ++j_) x[j_] = cons->val[SITE + j_];
                    lp = d.logpdf(x);
                    changed = true;
                    weight += lp;
                    if (in_sub) sw += lp;
                    MP_FN_PUT_();
                    return;
                }
            } else {
                fresh = (mask & bit) != 0;
            }
            if (in_sub && had) sw -= prev->lp[SITE];   // trace.data.remove(addr) comes first in every arm
            if (!fresh && had) {
                MP_FN_PREV_();
                if (!changed) {
                    lp = prev->lp[SITE];   // NoChange: the call goes back into the trace as it was
                } else {
                    lp = d.logpdf(x);
                    weight += lp - prev->lp[SITE];
                }
            } else {
                mp_site st(rng, dom, (uint32_t)SITE);
                d.sample(st, x);
                lp = d.logpdf(x);
                changed = true;
            }
            if (in_sub) sw += lp;
        }
        MP_FN_PUT_();
#undef MP_FN_PUT_
#undef MP_FN_PREV_
    }
    template <int SITE, class Dist>
    MP_HD double at(const Dist& d) {
        double x[1];
        at_k<SITE, 1>(mp_fn_scalar<Dist>{d}, x);
        return x[0];
    }
    template <int SITE>
    MP_HD double normal(double mu, double sd, double ln_sd) { return at<SITE>(mp_fn_normal{mu, sd, ln_sd}); }
    template <int SITE>
    MP_HD double normal(double mu, double sd) { return at<SITE>(mp_fn_normal{mu, sd, mp_log(sd)}); }
    template <int SITE>
    MP_HD bool bernoulli(double p) { return at<SITE>(mp_fn_bernoulli{p}) != 0.; }
    template <int SITE>
    MP_HD double uniform(double a, double b) { return at<SITE>(mp_fn_uniform{a, b}); }
    // vector-valued sites (two slots each): `uniform_2d(bounds) %= addr`, `mvnormal(mu, cov) %= addr` with the covariance constants
    // hoisted by the functor (cov itself, row-major, is what a dynamic interpretation hands its own mvnormal)
    template <int SITE>
    MP_HD void uniform_2d(double xmin, double xmax, double ymin, double ymax, double neg_ln_area, double* out) {
        at_k<SITE, 2>(mp_fn_uniform_2d{xmin, xmax, ymin, ymax, neg_ln_area}, out);
    }
    template <int SITE>
    MP_HD void mvnormal2(const double* mu, const double* /*cov*/, const double* cov_inv, double ln_det, const double* chol, double* out) {
        at_k<SITE, 2>(mp_fn_mvnormal2{{mu[0], mu[1]}, {cov_inv[0], cov_inv[1], cov_inv[2], cov_inv[3]}, ln_det, chol[0], chol[2], chol[3]}, out);
    }

    // previous choices of the frame FRAME (a call's SITES, or everything) that this visit did not reach: they leave the trace.  What
    // leaves is what `Trie::collect` removes (trie.rs:222-246): the frame's own unvisited leaves, each with its log-density, and — round 5 —
    // every sub-call of the frame that the body did not enter at all, as ONE term: that sub-trie's RUNNING weight (`self.weight -=
    // sub.weight`, trie.rs:161-184; the history of its inserts and removes, kept in subw[its lowest site]), not a fresh sum of its
    // leaves' log-densities — the same real number, other last bits once the sub-trie has a history.  (A sub-call that was entered closed
    // with a gc of its own, which marked what it took: an unvisited site inside a sub-call means the whole call was skipped.)  Whole
    // sub-calls first, then leaves, each in site order.
    template <uint64_t FRAME>
    MP_HD double collect() {
        constexpr bits_t sites = (bits_t)FRAME;
        const bits_t un = prev->present & sites & ~visited;
        double c = 0.;
        bits_t leaves = un;
        if constexpr (!std::is_void<SUBS>::value) {
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const bits_t X = (bits_t)mp_fn_child_call<SUBS>(k, FRAME);
                if (X != 0u && (X & ((bits_t(1) << k) - 1u)) == 0u && ((X >> k) & 1u)) {   // k is the call's lowest site: where its running weight is kept
                    if (un & X) {
                        c += prev->subw[k];
                        if (in_sub) sw -= prev->subw[k];   // trace.data.remove(addr) of the whole sub-trie
                    }
                    leaves &= ~X;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NS; ++k)
            if ((leaves >> k) & 1u) {
                c += prev->lp[k];
                if (in_sub) sw -= prev->lp[k];   // collect removes them from the (sub-)trie
            }
        discarded |= un;
        visited |= un;   // they have left: the caller's own gc (finish, or an enclosing collect) must not take them a second time — a sub-call's
                         // gc removes them from the SUB-trie, and the outer gc walks the outer trie only (dyngenfn.rs:453-483).  (Until round 4
                         // update() subtracted the log-density of a choice dropped inside a sub-call twice: found by the standalone
                         // mp_fn_update test against the trie engine; the fused mh tests only ever rejected such moves.)
        return c;
    }

    // Sub-calls nest: a body may itself `call` (SITES of the outer call then includes the inner call's sites, and the outer call owns
    // at least one site below them: its lowest site names where the trace keeps its running weight).  Every call is a frame — its own
    // weight from 0, its own trie weight `sw` — and on the way out it does to the ENCLOSING frame's trie what trace_at does to
    // `trace.data`: remove(addr) took the old sub-trie's weight off when the call began (:207, :221, :252-), insert(addr, sub) puts the
    // new one on (:186, :200, :244).
    template <uint64_t SITES64, class Body>
    MP_HD auto call(Body&& body) {
        static_assert(SITES64 != 0u, "a sub-call names the set of its sites");
        static_assert(NS > 32 || (SITES64 >> 32) == 0u, "a sub-call's sites are sites of the model");
        constexpr bits_t SITES = (bits_t)SITES64;
        constexpr int ID = __builtin_ctzll(SITES64);   // where the trace keeps this sub-trie's running weight
        const bool o_in = in_sub;                  // the enclosing frame: is it a sub-trie whose weight is being kept, and that weight so far
        const double o_sw = sw;
        if constexpr (MODE == MP_FN_SIMULATE) {
            // the sub-trace's choices are the caller's; `propose`'s weight is the whole trie's
            in_sub = true; sw = 0.;
            auto r = body(*this);
            const double w_new = sw;
            tr.subw[ID] = w_new;
            in_sub = o_in; sw = o_in ? o_sw + w_new : o_sw;
            return r;
        } else if constexpr (MODE == MP_FN_GENERATE) {
            // generate(args, choices): weight += d_weight as one term (:316-319); simulate when nothing is constrained
            const double w_out = weight;
            const bool any = (cons->present & SITES) != 0;
            weight = 0.;
            in_sub = true; sw = 0.;
            auto r = body(*this);
            const double w_new = sw;
            tr.subw[ID] = w_new;
            in_sub = o_in; sw = o_in ? o_sw + w_new : o_sw;
            weight = any ? w_out + weight : w_out;
            return r;
        } else {
            const bits_t had = prev->present & SITES;
            const double w_out = weight;
            if (from_prev) {
                // inside an enclosing generate(args, old sub-trace): this call is that generate's own trace_at — the old choices of
                // its sites are its constraints, the enclosing trie is a fresh one (nothing to remove from it)
                weight = 0.;
                sw = 0.;
                auto r = body(*this);
                const double w_new = sw;
                tr.subw[ID] = w_new;
                sw = o_sw + w_new;
                weight = had ? w_out + weight : w_out;
                return r;
            }
            bits_t touched;
            if constexpr (MODE == MP_FN_UPDATE) touched = cons->present & SITES;
            else touched = mask & SITES;
            const double e_sw = (o_in && had) ? o_sw - prev->subw[ID] : o_sw;   // the enclosing trie after trace.data.remove(addr)
            if (!touched && had && !changed) {
                in_sub = false;         // replay: every site returns its previous value and log-density; no trie is touched but the
                auto r = body(*this);   // enclosing one, which gets the sub-trie back as it was
                weight = w_out;
                changed = false;
                tr.subw[ID] = prev->subw[ID];
                in_sub = o_in; sw = o_in ? e_sw + prev->subw[ID] : o_sw;
                return r;
            }
            if (MODE == MP_FN_REGENERATE && !touched && had) {
                // generate(args, sub): new_weight from 0 over the old choices, a fresh sub-trie; weight += new_weight - sub.weight()
                weight = 0.;
                from_prev = true; sw = 0.;
                auto r = body(*this);
                from_prev = false;
                if (had & ~visited) panic = true;   // "not all constraints were consumed" (:526-529)
                weight = w_out + (weight - prev->subw[ID]);
                const double w_new = sw;
                tr.subw[ID] = w_new;
                in_sub = o_in; sw = o_in ? e_sw + w_new : o_sw;
                changed = true;
                return r;
            }
            weight = 0.;
            in_sub = true; sw = had ? prev->subw[ID] : 0.;
            auto r = body(*this);
            if (had) {
                const double c = collect<SITES64>();
                if constexpr (MODE == MP_FN_UPDATE) weight = weight - c;
            }
            const double w_new = sw;
            tr.subw[ID] = w_new;
            in_sub = o_in; sw = o_in ? e_sw + w_new : o_sw;
            weight = (touched || had) ? w_out + weight : w_out;
            changed = true;
            return r;
        }
    }

    // the outer gc of update / regenerate (dyngenfn.rs:453-483); constraints nobody consumed are the reference's panic
    MP_HD void finish() {
        if constexpr (MODE == MP_FN_UPDATE) {
            const double c = collect<~uint64_t(0)>();
            weight = weight - c;
            if (cons->present & ~consumed) panic = true;
        } else if constexpr (MODE == MP_FN_REGENERATE) {
            (void)collect<~uint64_t(0)>();
        } else if constexpr (MODE == MP_FN_GENERATE) {
            if (cons->present & ~consumed) panic = true;
        }
    }
};
