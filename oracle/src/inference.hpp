// ORACLE — TEST INFRASTRUCTURE ONLY (see rng.hpp).
//
// inference.hpp — CPU restatement of the Unfold combinator and of the inference library:
//   modppl/src/modeling/dynunfold.rs:7-100          DynUnfold (simulate / generate / update Extend)
//   modppl/src/inference/particle_filter.rs:8-121   ParticleSystem
//   modppl/src/inference/importance.rs:12-50        importance_sampling / importance_resampling
//   modppl/src/inference/mh.rs:9-75                 metropolis_hastings / regenerative_metropolis_hastings
//
// Seeded-stream convention (the reference has none; mp_philox.h defines it):
//   * particle i / chain i owns Philox slot i; an Unfold kernel call at time t uses step t;
//   * the i-th categorical draw of a resample is uniform number i of the resample's stream (rng.hpp resample_rng:
//     Philox block i >> 1 of (step = resample count, DOM_RESAMPLE), half i & 1);
//   * MH iteration `it` (1-based; step 0 is the initial generate) uses step `it`; the accept
//     uniform is (DOM_ACCEPT, site 0).
//
// Resampling arithmetic, two modes:
//   literal   : particle_filter.rs:27-41 + categorical.rs:22-32 as written (fp64 running sum).
//   canonical : the order-free fixed-point CDF the GPU uses, spelled out
//               in `canonical_normalize` / `canonical_parent` below and in DESIGN.md §4.
#pragma once
#include <cstring>

#include "dyngenfn.hpp"

namespace oracle {

// ---- dynunfold.rs ---------------------------------------------------------------------
template <class State>
struct DynUnfold : GenFn<std::pair<int64_t, State>, std::vector<DynTrie>, std::vector<State>> {
    using A = std::pair<int64_t, State>;
    using TraceT = Trace<A, std::vector<DynTrie>, std::vector<State>>;
    using KH = DynGenFnHandler<A, State>;
    DynGenFn<A, State> kernel;

    DynUnfold() {}
    explicit DynUnfold(DynGenFn<A, State> k) : kernel(std::move(k)) {}

    TraceT simulate(Rng& rng, A fa) const override {
        auto [final_t, state] = fa;
        if (!(final_t >= 1)) throw Panic("assert final_t >= 1");
        TraceT vt{{final_t, state}, {}, std::vector<State>{}, 0.};
        for (int64_t t = 0; t < final_t; ++t) {
            rng.step = (uint32_t)t;
            KH g = kernel.handler(KH::Simulate, rng);
            g.trace = Trace<A, DynTrie, State>{{t, state}, DynTrie(), std::nullopt, 0.};
            state = kernel.func(g, {t, state});
            vt.retv->push_back(state);
            vt.data.push_back(std::move(g.trace.data));
            vt.logjp += g.trace.logjp;  // stays 0: per-step logjp is never set
        }
        return vt;
    }
    std::pair<TraceT, double> generate(Rng& rng, A fa, std::vector<DynTrie> vc) const override {
        auto [final_t, state] = fa;
        if (!(final_t >= 1)) throw Panic("assert final_t >= 1");
        TraceT vt{{final_t, state}, {}, std::vector<State>{}, 0.};
        double gen_weight = 0.;
        int64_t t = 0;
        for (auto& constraints : vc) {
            rng.step = (uint32_t)t;
            KH g = kernel.handler(KH::Generate, rng);
            g.trace = Trace<A, DynTrie, State>{{t, state}, DynTrie(), std::nullopt, 0.};
            g.constraints = std::move(constraints);
            state = kernel.func(g, {t, state});
            if (!g.constraints.is_empty()) throw Panic("assert constraints.is_empty()");
            vt.retv->push_back(state);
            vt.data.push_back(std::move(g.trace.data));
            vt.logjp += g.trace.logjp;
            gen_weight += g.weight;
            ++t;
        }
        return {std::move(vt), gen_weight};
    }
    std::tuple<TraceT, std::vector<DynTrie>, double> update(Rng& rng, TraceT vt, A fa, ArgDiff diff, std::vector<DynTrie> vc) const override {
        const int64_t final_t = fa.first;
        if (!(final_t >= 1)) throw Panic("assert final_t >= 1");
        const int64_t prev_t = vt.args.first;
        if (final_t - prev_t != (int64_t)vc.size()) throw Panic("assert final_t - prev_t == vec_constraints.len()");
        State state = vt.retv->back();
        double update_weight = 0.;
        if (diff != ArgDiff::Extend) throw Panic("Can't handle GF change type");
        int64_t k = 0;
        for (auto& constraints : vc) {
            const int64_t t = prev_t + k;
            rng.step = (uint32_t)t;
            KH g = kernel.handler(KH::Generate, rng);
            g.trace = Trace<A, DynTrie, State>{{t, state}, DynTrie(), std::nullopt, 0.};
            g.constraints = std::move(constraints);
            state = kernel.func(g, {t, state});
            if (!g.constraints.is_empty()) throw Panic("assert constraints.is_empty()");
            vt.args.first += 1;
            vt.retv->push_back(state);
            vt.data.push_back(std::move(g.trace.data));
            vt.logjp += g.trace.logjp;
            update_weight += g.weight;
            ++k;
        }
        return {std::move(vt), std::vector<DynTrie>((size_t)(final_t - prev_t)), update_weight};
    }
};

// ---- canonical (order-free) normalisation: the spec both oracle and GPU implement ---------
inline int ceil_log2_u64(uint64_t n) {
    int b = 0;
    while (((uint64_t)1 << b) < n) ++b;
    return b;
}
// The spec is hierarchical so that the per-particle work needs no global quantity (the GPU does it inside the
// propagate kernel) and so that shards never need each other's maxima:
//   level 0, tile b = 2048 consecutive GLOBAL slots:  m_b = max lw;  a_i = mp_exp(lw_i - m_b);
//            q_i = rint(a_i * 2^51);  cum = tile-local inclusive prefix;  W_b = sum q_i;  W2_b = sum rint(a_i^2 * 2^51)
//   level 1, over tiles:  m = max_b m_b;  S = 62 - ceil(log2 N_global);
//            T_b  = rint((double)W_b  * mp_exp(m_b - m)      * 2^(S-51));   Q  = sum T_b
//            T2_b = rint((double)W2_b * mp_exp(2*(m_b - m))  * 2^(S-51));   Q2 = sum T2_b
//            L = m + mp_log(Q * 2^-S);  ESS = (Q*2^-S)^2 / (Q2*2^-S)
//   draw:    target in [1, Q] (canonical_target / _systematic);  tile b = first with inclT_b >= target;
//            r = target - exclT_b;  lt = clamp((u64)ceil((double)r * ((double)W_b / (double)T_b)), 1, W_b);
//            parent = first row of tile b with cum >= lt.
// All sums are integer (any order / sharding gives the same bits); the only fp operations are single IEEE ops and
// mp_exp / mp_log, evaluated identically on host and device.
constexpr uint64_t CANON_TILE = 2048;
struct CanonTiles {  // level 0 of a contiguous run of slots starting at a tile boundary
    std::vector<double> m;       // per tile
    std::vector<uint64_t> W, W2; // per tile
    std::vector<uint64_t> cum;   // per element: tile-local inclusive prefix
};
inline CanonTiles canonical_tiles(const std::vector<double>& logw) {
    CanonTiles t;
    const size_t n = logw.size(), nt = (n + CANON_TILE - 1) / CANON_TILE;
    t.m.assign(nt, -INFINITY); t.W.assign(nt, 0); t.W2.assign(nt, 0); t.cum.assign(n, 0);
    const double scale = std::ldexp(1.0, 51);
    for (size_t b = 0; b < nt; ++b) {
        const size_t lo = b * CANON_TILE, hi = std::min(n, lo + CANON_TILE);
        double mb = -INFINITY;
        for (size_t i = lo; i < hi; ++i) mb = std::fmax(mb, logw[i]);
        t.m[b] = mb;
        const bool ok = (mb > -INFINITY) && (mb < INFINITY);
        uint64_t run = 0, run2 = 0;
        for (size_t i = lo; i < hi; ++i) {
            const double a = ok ? mp_exp(logw[i] - mb) : 0.;
            const double r = std::rint(a * scale), r2 = std::rint((a * a) * scale);
            run += (r >= 0.) ? (uint64_t)r : 0;
            run2 += (r2 >= 0.) ? (uint64_t)r2 : 0;
            t.cum[i] = run;
        }
        t.W[b] = run; t.W2[b] = run2;
    }
    return t;
}
struct CanonNorm {
    int S = 0;
    double m = -INFINITY; // global max log-weight
    uint64_t Q = 0, Q2 = 0;
    double L = -INFINITY, ess = 0.;
    std::vector<uint64_t> inclT;  // inclusive prefix of T_b over ALL tiles of the job
    std::vector<uint64_t> W;      // W_b of all tiles
    bool degenerate() const { return !(m > -INFINITY) || !(m < INFINITY) || Q == 0; }
};
// level 1 from the (m_b, W_b, W2_b) of all tiles of the job
inline CanonNorm canonical_combine(const std::vector<double>& tm, const std::vector<uint64_t>& tW, const std::vector<uint64_t>& tW2, uint64_t n_global) {
    CanonNorm c;
    c.S = 62 - ceil_log2_u64(n_global);
    for (double x : tm) c.m = std::fmax(c.m, x);
    c.W = tW;
    c.inclT.assign(tm.size(), 0);
    if (!(c.m > -INFINITY) || !(c.m < INFINITY)) return c;
    const double sc = std::ldexp(1.0, c.S - 51), inv = std::ldexp(1.0, -c.S);
    uint64_t run = 0;
    for (size_t b = 0; b < tm.size(); ++b) {
        const double f = mp_exp(tm[b] - c.m);
        const double t = std::rint((double)tW[b] * f * sc);
        const double t2 = std::rint((double)tW2[b] * mp_exp(2. * (tm[b] - c.m)) * sc);
        run += (t >= 0.) ? (uint64_t)t : 0;
        c.Q2 += (t2 >= 0.) ? (uint64_t)t2 : 0;
        c.inclT[b] = run;
    }
    c.Q = run;
    const double Qs = (double)c.Q * inv, Q2s = (double)c.Q2 * inv;
    c.L = c.m + mp_log(Qs);
    c.ess = (Qs * Qs) / Q2s;
    return c;
}
// tile and tile-local target of a global target
inline void canonical_locate(const CanonNorm& c, uint64_t target, size_t* tile, uint64_t* lt) {
    size_t lo = 0, hi = c.inclT.size();
    while (lo < hi) { const size_t mid = (lo + hi) / 2; if (c.inclT[mid] >= target) hi = mid; else lo = mid + 1; }
    if (lo >= c.inclT.size()) lo = c.inclT.size() - 1;
    const uint64_t excl = lo ? c.inclT[lo - 1] : 0;
    const uint64_t T = c.inclT[lo] - excl, r = target - excl;
    const double ratio = (double)c.W[lo] / (double)T;
    double v = std::ceil((double)r * ratio);
    uint64_t x = (v >= 1.) ? (uint64_t)v : 1;
    if (x > c.W[lo]) x = c.W[lo];
    if (x < 1) x = 1;
    *tile = lo; *lt = x;
}
// first row of tile `tile` (rows [tile*2048 - first_slot, ...) of `cum`) with cum >= lt; index into `cum`
inline size_t canonical_row(const std::vector<uint64_t>& cum, size_t local_tile, uint64_t lt) {
    const size_t lo0 = local_tile * CANON_TILE, hi0 = std::min(cum.size(), lo0 + CANON_TILE);
    size_t lo = lo0, hi = hi0;
    while (lo < hi) { const size_t mid = (lo + hi) / 2; if (cum[mid] >= lt) hi = mid; else lo = mid + 1; }
    return lo < hi0 ? lo : hi0 - 1;
}
// convenience for an unsharded weight vector: everything at once
struct CanonFull { CanonTiles t; CanonNorm c; };
inline CanonFull canonical_normalize(const std::vector<double>& logw, uint64_t n_global) {
    CanonFull f;
    f.t = canonical_tiles(logw);
    f.c = canonical_combine(f.t.m, f.t.W, f.t.W2, n_global);
    return f;
}
inline size_t canonical_parent(const CanonFull& f, uint64_t target) {
    size_t tile; uint64_t lt;
    canonical_locate(f.c, target, &tile, &lt);
    return canonical_row(f.t.cum, tile, lt);
}
// target = max(1, ceil(k * Q / 2^52))
inline uint64_t canonical_target(uint64_t k52, uint64_t Q) {
    const unsigned __int128 p = (unsigned __int128)k52 * Q + (((unsigned __int128)1 << 52) - 1);
    const uint64_t t = (uint64_t)(p >> 52);
    return t < 1 ? 1 : t;
}
// systematic (extension): p = g*2^32 + k32; target = ((p*Q) >> 32) / N + 1
inline uint64_t canonical_target_systematic(uint64_t g, uint32_t k32, uint64_t Q, uint64_t n_global) {
    const unsigned __int128 p = ((unsigned __int128)g << 32) | k32;
    const unsigned __int128 a = (p * Q) >> 32;
    return (uint64_t)(a / n_global) + 1;
}
inline uint32_t canonical_systematic_k32(uint64_t seed, uint32_t rc) {
    Rng r; r.seed = seed; r.slot = 0; r.step = rc; r.at(DOM_RESAMPLE, 1);
    return (uint32_t)(r.bits64() >> 32);
}
// stratified (extension): one k32 per global output slot g (site 2)
inline uint32_t canonical_stratified_k32(uint64_t seed, uint32_t rc, uint64_t g) {
    return resample_k32(seed, DOM_RESAMPLE, 2, rc, g);
}
// scheme 1 (systematic) or 2 (stratified)
inline uint64_t canonical_target_lattice(int scheme, uint64_t seed, uint32_t rc, uint64_t g, uint64_t Q, uint64_t n_global) {
    return canonical_target_systematic(g, scheme == 2 ? canonical_stratified_k32(seed, rc, g) : canonical_systematic_k32(seed, rc), Q, n_global);
}

// ---- split multinomial resample (sharded filters, opt-in) -----------------------------------------------
// particle_filter.rs:37-41 draws N i.i.d. parents from the normalised weights.  Over G ranks the number landing on each rank is
// Multinomial(N; M_r / Q) and, given the counts, a rank's parents are i.i.d. from its own weights.  The counts are drawn by
// binary splitting of the ranks (padded with empty ranks to a power of two): node k of the heap (root 1) with n_k draws and
// mass S_k gives its left child Binomial(n_k, S_left / S_k) of them.  One binomial variate (Hoermann 1993): BTRS for
// n p >= 10, search from 0 on the probability recurrence below; the smaller of the two masses is the one sampled (p <= 1/2).
// Uniforms: stream (slot = node, step = resample count, domain RESAMPLE, site 3); attempt a of the sampler takes uniforms
// 2a (U) and 2a + 1 (V).  Restated independently of modppl_amd/csrc/mp_binomial.h; tests compare the two bit for bit.
inline double canonical_stirling_tail(double k) {
    static const double tab[10] = {0.0810614667953272,  0.0413406959554092, 0.0276779256849983, 0.02079067210376509, 0.0166446911898211,
                                   0.0138761288230707,  0.0118967099458917, 0.0104112652619720, 0.00925546218271273, 0.00833056343336287};
    if (k <= 9.) return tab[(int)k];
    const double kp1 = k + 1., kp1sq = kp1 * kp1;
    return (1.0 / 12. - (1.0 / 360. - 1.0 / 1260. / kp1sq) / kp1sq) / kp1;
}
inline uint64_t canonical_binomial_small_p(uint64_t n_u, double p, uint64_t seed, uint32_t rc, uint32_t node) {
    Rng rng; rng.seed = seed; rng.slot = node; rng.step = rc; rng.at(DOM_RESAMPLE, 3);
    const double n = (double)n_u, q = 1. - p;
    if (n * p < 10.) {
        const double s = p / q, f0 = mp_exp(n * mp_log(q));
        for (uint32_t att = 0; att < (1u << 16); ++att) {
            rng.n = 2 * att;
            double u = rng.u01(), f = f0, k = 0.;
            bool ok = true;
            while (u >= f) {
                u -= f;
                k += 1.;
                if (k > n || k > 512.) { ok = false; break; }
                f *= (n - k + 1.) / k * s;
            }
            if (ok) return (uint64_t)k;
        }
        return 0;
    }
    const double spq = std::sqrt(n * p * q);
    const double b = 1.15 + 2.53 * spq;
    const double a = -0.0873 + 0.0248 * b + 0.01 * p;
    const double c = n * p + 0.5;
    const double v_r = 0.92 - 4.2 / b;
    const double r = p / q;
    const double alpha = (2.83 + 5.1 / b) * spq;
    const double m = std::floor((n + 1.) * p);
    for (uint32_t att = 0; att < (1u << 16); ++att) {
        rng.n = 2 * att;
        const double u = rng.u01() - 0.5;
        double v = rng.u01();
        const double us = 0.5 - std::fabs(u);
        const double k = std::floor((2. * a / us + b) * u + c);
        if (!(k >= 0. && k <= n)) continue;
        if (us >= 0.07 && v <= v_r) return (uint64_t)k;
        v = mp_log(v * alpha / (a / (us * us) + b));
        const double ub = (m + 0.5) * mp_log((m + 1.) / (r * (n - m + 1.))) + (n + 1.) * mp_log((n - m + 1.) / (n - k + 1.)) +
                          (k + 0.5) * mp_log(r * (n - k + 1.) / (k + 1.)) + canonical_stirling_tail(m) + canonical_stirling_tail(n - m) -
                          canonical_stirling_tail(k) - canonical_stirling_tail(n - k);
        if (v <= ub) return (uint64_t)k;
    }
    return (uint64_t)m;
}
inline uint64_t canonical_binomial(uint64_t n, uint64_t a, uint64_t b, uint64_t seed, uint32_t rc, uint32_t node) {
    if (n == 0 || a == 0) return 0;
    if (a >= b) return n;
    const uint64_t other = b - a;
    if (a <= other) return canonical_binomial_small_p(n, (double)a / (double)b, seed, rc, node);
    return n - canonical_binomial_small_p(n, (double)other / (double)b, seed, rc, node);
}
// offspring per rank: mass[r] = rank r's share of Q
inline void canonical_split_node(const std::vector<uint64_t>& mass, size_t a, size_t b, uint64_t n, uint32_t node, uint64_t seed, uint32_t rc,
                                 std::vector<uint64_t>& out) {
    if (b - a == 1) { if (a < out.size()) out[a] = n; return; }
    const size_t mid = a + (b - a) / 2;
    uint64_t S = 0, Sl = 0;
    for (size_t r = a; r < b && r < mass.size(); ++r) { S += mass[r]; if (r < mid) Sl += mass[r]; }
    const uint64_t left = canonical_binomial(n, Sl, S, seed, rc, node);
    canonical_split_node(mass, a, mid, left, 2 * node, seed, rc, out);
    canonical_split_node(mass, mid, b, n - left, 2 * node + 1, seed, rc, out);
}
inline std::vector<uint64_t> canonical_split_counts(const std::vector<uint64_t>& mass, uint64_t n_global, uint64_t seed, uint32_t rc) {
    size_t P = 1;
    while (P < mass.size()) P *= 2;
    std::vector<uint64_t> out(mass.size(), 0);
    canonical_split_node(mass, 0, P, n_global, 1, seed, rc, out);
    return out;
}
// draw j of rank `rank` (its own stream: uniform j of (slot = j >> 1 block, step = resample count, domain RESAMPLE, site 0), block
// word 3 = the rank — rank 0's stream IS the single filter's, so a world of one draws the single filter's parents)
inline uint64_t canonical_split_u52(uint64_t seed, uint32_t rc, int rank, uint64_t j) {
    const uint32_t ctr[4] = {(uint32_t)(j >> 1), rc, (uint32_t)DOM_RESAMPLE << 16, (uint32_t)rank};
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t o[4];
    philox4x32_10(ctr, key, o);
    const uint64_t bits = (j & 1) ? (((uint64_t)o[3] << 32) | o[2]) : (((uint64_t)o[1] << 32) | o[0]);
    return bits >> 12;
}

// ---- particle_filter.rs ---------------------------------------------------------------
template <class Args, class Data, class Ret>
struct ParticleSystem {
    using F = GenFn<std::pair<int64_t, Args>, Data, Ret>;
    using TraceT = Trace<std::pair<int64_t, Args>, Data, Ret>;
    size_t num_particles;
    const F* model;
    std::vector<TraceT> traces;
    std::vector<double> log_weights, log_normalized_weights, two_times_log_normalized_weights, normalized_weights;
    std::vector<size_t> parents;
    uint64_t seed;              // stands in for `rng: ThreadRng`
    uint32_t resample_count = 0;
    double log_ml_estimate = 0.;
    bool canonical_resampling = false;
    bool fast_search = false;   // binary search over the same sequential running sum (index-identical)
    double canon_ess_stale;     // canonical-mode ESS as of the last resample (1/N before any)

    ParticleSystem(const F& m, size_t n, uint64_t seed_)  // particle_filter.rs:44-57
        : num_particles(n), model(&m), log_weights(n, 0.), log_normalized_weights(n, 0.),
          two_times_log_normalized_weights(n, 0.), normalized_weights(n, 0.), parents(n, 0), seed(seed_),
          canon_ess_stale(1.0 / (double)n) {}

    Rng rng_for(size_t i) const { Rng r; r.seed = seed; r.slot = (uint32_t)i; return r; }

    double normalize_weights() {  // :27-35
        const double log_total_weight = logsumexp(log_weights);
        for (size_t i = 0; i < num_particles; ++i) {
            log_normalized_weights[i] = log_weights[i] - log_total_weight;
            two_times_log_normalized_weights[i] = 2.0 * log_normalized_weights[i];
            normalized_weights[i] = o_exp(log_normalized_weights[i]);
        }
        return log_total_weight;
    }
    void multinomial_resampling() {  // :37-41
        std::vector<double> cdf;
        if (fast_search) {
            Categorical::check_sum(normalized_weights);
            cdf.resize(num_particles);
            double t = 0.;
            for (size_t i = 0; i < num_particles; ++i) { t += normalized_weights[i]; cdf[i] = t; }
        }
        for (size_t i = 0; i < num_particles; ++i) {
            Rng r = resample_rng(seed, DOM_RESAMPLE, resample_count, i);
            int64_t p;
            if (!fast_search) {
                p = categorical.random(r, normalized_weights);  // clones + re-sums per draw in the reference
            } else {
                const double u = r.u01();
                if (!(0. < u)) p = -1;
                else {  // first x with cdf[x] >= u  ==  the while(t<u) scan's exit index
                    size_t lo = 0, hi = num_particles;
                    while (lo < hi) { size_t mid = (lo + hi) / 2; if (cdf[mid] >= u) hi = mid; else lo = mid + 1; }
                    if (lo >= num_particles) throw Panic("categorical: index out of bounds");
                    p = (int64_t)lo;
                }
            }
            if (p < 0) throw Panic("categorical returned -1 (u == 0): usize index panic in the reference");
            parents[i] = (size_t)p;
        }
    }
    void init_step(Args args, Data constraints) {  // :60-70
        for (size_t i = 0; i < num_particles; ++i) {
            Rng r = rng_for(i);
            auto [trace, log_weight] = model->generate(r, {1, args}, constraints);
            traces.push_back(std::move(trace));
            log_weights[i] = log_weight;
        }
    }
    void step(Data constraints) {  // :73-96 (consumes self in the reference)
        std::vector<TraceT> tmp_traces;
        std::vector<double> tmp_log_weights;
        size_t i = 0;
        for (auto& trace : traces) {
            auto args = trace.args;
            std::pair<int64_t, Args> new_args{args.first + 1, args.second};
            Rng r = rng_for(i);
            auto [new_trace, discard, log_weight] = model->update(r, std::move(trace), new_args, ArgDiff::Extend, constraints);
            (void)discard;
            tmp_traces.push_back(std::move(new_trace));
            tmp_log_weights.push_back(log_weights[i] + log_weight);
            ++i;
        }
        traces = std::move(tmp_traces);
        log_weights = std::move(tmp_log_weights);
    }
    double effective_sample_size() const {  // :98-100 (stale buffers: SURVEY §7 quirk)
        if (canonical_resampling) return canon_ess_stale;
        return o_exp(-logsumexp(two_times_log_normalized_weights));
    }
    double resample() {  // :103-116
        if (log_weights.size() != num_particles) throw Panic("resample before init_step (index out of bounds)");
        double log_total_weight;
        if (!canonical_resampling) {
            log_total_weight = normalize_weights();
            log_ml_estimate += log_total_weight - o_ln((double)num_particles);
            multinomial_resampling();
        } else {
            CanonFull c = canonical_normalize(log_weights, num_particles);
            if (c.c.degenerate()) throw Panic("all log-weights are -inf: normalized weights are NaN");
            log_total_weight = c.c.L;
            canon_ess_stale = c.c.ess;
            log_ml_estimate += log_total_weight - o_ln((double)num_particles);
            for (size_t i = 0; i < num_particles; ++i) {
                Rng r = resample_rng(seed, DOM_RESAMPLE, resample_count, i);
                parents[i] = canonical_parent(c, canonical_target(r.u52(), c.c.Q));
            }
        }
        ++resample_count;
        std::vector<TraceT> tmp_traces;
        tmp_traces.reserve(num_particles);
        for (size_t i = 0; i < num_particles; ++i) tmp_traces.push_back(traces[parents[i]]);
        traces = std::move(tmp_traces);
        std::fill(log_weights.begin(), log_weights.end(), 0.);
        return log_total_weight;
    }
    double log_marginal_likelihood_estimate() const {  // :119-121
        if (canonical_resampling) {
            CanonFull c = canonical_normalize(log_weights, num_particles);
            return log_ml_estimate + c.c.L - o_ln((double)num_particles);
        }
        return log_ml_estimate + logsumexp(log_weights) - o_ln((double)num_particles);
    }
};

// ---- importance.rs --------------------------------------------------------------------
template <class Args, class Data, class Ret>
struct ImportanceResult {
    std::vector<Trace<Args, Data, Ret>> traces;
    std::vector<double> log_normalized_weights;
    double log_ml_estimate;
    std::vector<size_t> resampled_indices;
};
template <class Args, class Data, class Ret>
ImportanceResult<Args, Data, Ret> importance_sampling(uint64_t seed, const GenFn<Args, Data, Ret>& model, Args model_args,
                                                      Data constraints, uint32_t num_samples, bool canonical = false,
                                                      CanonFull* canon_out = nullptr) {
    ImportanceResult<Args, Data, Ret> out;
    std::vector<double> w;
    for (uint32_t i = 0; i < num_samples; ++i) {  // importance.rs:18-20
        Rng r; r.seed = seed; r.slot = i;
        auto [tr, wi] = model.generate(r, model_args, constraints);
        out.traces.push_back(std::move(tr));
        w.push_back(wi);
    }
    double log_total_weight;
    if (!canonical) {
        log_total_weight = logsumexp(w);  // :21
    } else {
        CanonFull c = canonical_normalize(w, num_samples);
        if (c.c.degenerate()) throw Panic("all log-weights are -inf");
        log_total_weight = c.c.L;
        if (canon_out) *canon_out = std::move(c);
    }
    out.log_ml_estimate = log_total_weight - o_ln((double)num_samples);  // :22
    for (double wi : w) out.log_normalized_weights.push_back(wi - log_total_weight);  // :23-25
    return out;
}
template <class Args, class Data, class Ret>
ImportanceResult<Args, Data, Ret> importance_resampling(uint64_t seed, const GenFn<Args, Data, Ret>& model, Args model_args,
                                                        Data constraints, uint32_t num_samples, uint32_t num_ret_samples,
                                                        bool canonical = false) {
    CanonFull c;
    auto out = importance_sampling(seed, model, model_args, constraints, num_samples, canonical, &c);
    if (!canonical) {
        std::vector<double> probs;  // :44
        for (double w : out.log_normalized_weights) probs.push_back(o_exp(w));
        for (uint32_t j = 0; j < num_ret_samples; ++j) {  // :45-47
            Rng r = resample_rng(seed, DOM_IS, 0, j);
            const int64_t p = categorical.random(r, probs);
            if (p < 0) throw Panic("categorical returned -1");
            out.resampled_indices.push_back((size_t)p);
        }
    } else {
        for (uint32_t j = 0; j < num_ret_samples; ++j) {
            Rng r = resample_rng(seed, DOM_IS, 0, j);
            out.resampled_indices.push_back(canonical_parent(c, canonical_target(r.u52(), c.c.Q)));
        }
    }
    return out;
}

// ---- mh.rs ----------------------------------------------------------------------------
// The proposal receives (Weak<Trace>, ProposalArgs); here a const pointer to the trace.
template <class Args, class Ret>
using DynTraceT = Trace<Args, DynTrie, Ret>;

template <class Args, class Ret, class PArgs>
std::pair<DynTraceT<Args, Ret>, bool> metropolis_hastings(
    Rng& rng, const GenFn<Args, DynTrie, Ret>& model, DynTraceT<Args, Ret> trace,
    const GenFn<std::pair<const DynTraceT<Args, Ret>*, PArgs>, DynTrie, int>& proposal, PArgs proposal_args,
    double* alpha_out = nullptr) {
    DynTraceT<Args, Ret> prev_trace = trace;  // mh.rs:15
    auto [fwd_choices, fwd_weight] = proposal.propose(rng, {&trace, proposal_args});  // :17-19
    Args args = trace.args;
    auto [new_trace, discard, weight] = model.update(rng, std::move(trace), args, ArgDiff::NoChange, std::move(fwd_choices));  // :23
    const double bwd_weight = proposal.assess(rng, {&new_trace, proposal_args}, std::move(discard));  // :25-27
    const double alpha = weight - fwd_weight + bwd_weight;  // :34
    if (alpha_out) *alpha_out = alpha;
    rng.at(DOM_ACCEPT, 0);
    if (o_ln(rng.u01()) < alpha) return {std::move(new_trace), true};  // :35-36
    return {std::move(prev_trace), false};
}

template <class Args, class Ret>
std::pair<DynTraceT<Args, Ret>, bool> regenerative_metropolis_hastings(
    Rng& rng, const GenFn<Args, DynTrie, Ret>& model, DynTraceT<Args, Ret> trace, const AddrMap& mask,
    double* weight_out = nullptr) {
    DynTraceT<Args, Ret> prev_trace = trace;  // mh.rs:59
    Args args = trace.args;
    auto [new_trace, weight] = model.regenerate(rng, std::move(trace), args, ArgDiff::NoChange, mask);  // :61
    if (weight_out) *weight_out = weight;
    rng.at(DOM_ACCEPT, 0);
    if (o_ln(rng.u01()) < weight) return {std::move(new_trace), true};  // :62-63
    return {std::move(prev_trace), false};
}

}  // namespace oracle
