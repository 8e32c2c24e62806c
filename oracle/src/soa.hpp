// ORACLE — TEST INFRASTRUCTURE ONLY (see rng.hpp).
//
// soa.hpp — the "algorithm-faithful" CPU variant of BASELINE.md §3: the same operations in the
// same order as the restated reference (dists.hpp / inference.hpp), but particle state kept
// in flat arrays and the handler's per-choice map bookkeeping elided, so that N = 2^20 is
// feasible on a CPU.  Each `kernel()` below is the hand-inlined Generate interpretation of the
// corresponding dyngen model in models.hpp (constrained site: score and add to the weight;
// free site: sample from the prior, no weight — dyngenfn.rs:115-141).  tests/test_oracle_pf.py
// checks it against the dynamic-handler version index for index.
//
// Resampling: `literal` = sequential fp64 running sum + binary search (index-identical to
// categorical.rs:24-31 for every u > 0); `canonical` = the fixed-point CDF of inference.hpp.
#pragma once
#include <algorithm>
#include <thread>

#include "models.hpp"

namespace oracle {

struct SoaModel {
    int dim_state = 1, dim_obs = 1;
    virtual ~SoaModel() {}
    // One Unfold kernel call in Generate mode with the observation sites constrained.
    // prev/next: dim_state doubles; returns the weight (sum of constrained logpdfs).
    virtual double kernel(Rng& r, int64_t t, const double* prev, double* next, const double* obs) const = 0;
};

struct SoaLgssm1 : SoaModel {
    LgssmParams p;
    explicit SoaLgssm1(LgssmParams p_) : p(p_) { dim_state = 1; dim_obs = 1; }
    double kernel(Rng& r, int64_t t, const double* prev, double* next, const double* obs) const override {
        r.at(DOM_MODEL, 0);
        const double x = (t == 0) ? normal.random(r, {p.mu0, p.sig0}) : normal.random(r, {p.a * prev[0], p.sig_x});
        next[0] = x;
        return normal.logpdf(obs[0], {x, p.sig_y});
    }
};

struct SoaSpiral : SoaModel {  // tests/dyngenfns/unfold.rs:14-32
    SoaSpiral() { dim_state = 2; dim_obs = 2; }
    double kernel(Rng& r, int64_t t, const double* prev, double* next, const double* obs) const override {
        Vec pol;
        if (t == 0) {
            r.at(DOM_MODEL, 0); const double rr = uniform.random(r, {0., 1.});
            r.at(DOM_MODEL, 1); const double th = uniform.random(r, {0., 2. * M_PI});
            pol = {rr, th};
        } else {
            r.at(DOM_MODEL, 0); const double dr = normal.random(r, {0., 0.1});
            r.at(DOM_MODEL, 1); const double dth = normal.random(r, {0.4, 0.2});
            pol = {prev[0] + dr, prev[1] + dth};
        }
        const Vec pos = polar_to_cartesian(pol);
        next[0] = pol[0]; next[1] = pol[1];
        return mvnormal.logpdf(Vec{obs[0], obs[1]}, MvNormalParams{pos, Mat(2, {0.001, 0., 0., 0.001})});
    }
};

struct SoaBearings : SoaModel {
    BearingsParams p;
    explicit SoaBearings(BearingsParams p_) : p(p_) { dim_state = 4; dim_obs = 1; }
    double kernel(Rng& r, int64_t t, const double* prev, double* next, const double* obs) const override {
        double px, py, vx, vy;
        if (t == 0) {
            r.at(DOM_MODEL, 0); px = normal.random(r, {p.p0x, p.sig_p0});
            r.at(DOM_MODEL, 1); py = normal.random(r, {p.p0y, p.sig_p0});
            r.at(DOM_MODEL, 2); vx = normal.random(r, {0., p.sig_v0});
            r.at(DOM_MODEL, 3); vy = normal.random(r, {0., p.sig_v0});
        } else {
            r.at(DOM_MODEL, 0); const double ax = normal.random(r, {0., p.sig_a});
            r.at(DOM_MODEL, 1); const double ay = normal.random(r, {0., p.sig_a});
            px = (prev[0] + prev[2]) + 0.5 * ax;
            py = (prev[1] + prev[3]) + 0.5 * ay;
            vx = prev[2] + ax;
            vy = prev[3] + ay;
        }
        next[0] = px; next[1] = py; next[2] = vx; next[3] = vy;
        return normal.logpdf(obs[0], {o_atan2(py, px), p.sig_theta});
    }
};

struct SoaLgssmBand : SoaModel {
    BandParams p;
    explicit SoaLgssmBand(BandParams p_) : p(p_) { dim_state = p.D; dim_obs = p.D; }
    double kernel(Rng& r, int64_t t, const double* prev, double* next, const double* obs) const override {
        double w = 0.;
        for (int j = 0; j < p.D; ++j) {
            r.at(DOM_MODEL, (uint32_t)j);
            double x;
            if (t == 0) x = normal.random(r, {0., p.sig0});
            else {
                const double nb = (j > 0 ? prev[j - 1] : 0.) + (j < p.D - 1 ? prev[j + 1] : 0.);
                x = normal.random(r, {p.a * (prev[j] + p.band * nb), p.sig_x});
            }
            next[j] = x;
            w += normal.logpdf(obs[j], {x, p.sig_y});
        }
        return w;
    }
};

struct SoaHmm : SoaModel {  // tests/hmm/model.rs:33-80
    HmmParams p;
    explicit SoaHmm(HmmParams p_) : p(std::move(p_)) { dim_state = 1; dim_obs = 1; }
    double kernel(Rng& r, int64_t t, const double* prev, double* next, const double* obs) const override {
        r.at(DOM_MODEL, 0);
        const Vec probs = (t == 0) ? p.prior : p.transition_col((int)prev[0]);
        const int64_t s = categorical.random(r, probs);
        next[0] = (double)s;
        return categorical.logpdf((int64_t)obs[0], p.emission_col((int)s));
    }
};

struct SoaPf {
    const SoaModel* model;
    size_t n;            // local particles
    uint64_t n_global, slot_offset;
    uint64_t seed;
    bool canonical;      // canonical resampling arithmetic (math mode is the global canonical_mode())
    int threads = 1;
    int64_t t = 0;       // Unfold steps taken (trace.args.0)
    uint32_t resample_count = 0;
    double log_ml = 0.;
    double ess_stale;
    std::vector<double> x, x_tmp, logw, lnw;  // x: [n][dim_state]
    std::vector<uint32_t> parents;
    bool initialised = false;

    SoaPf(const SoaModel* m, size_t n_, uint64_t seed_, bool canon, uint64_t n_global_ = 0, uint64_t off = 0)
        : model(m), n(n_), n_global(n_global_ ? n_global_ : n_), slot_offset(off), seed(seed_), canonical(canon),
          ess_stale(1.0 / (double)(n_global_ ? n_global_ : n_)), x(n_ * (size_t)m->dim_state, 0.), x_tmp(x.size()),
          logw(n_, 0.), lnw(n_, 0.), parents(n_, 0) {}

    template <class F>
    void parallel_for(F f) const {
        if (threads <= 1) { f(0, n); return; }
        std::vector<std::thread> th;
        for (int k = 0; k < threads; ++k) {
            const size_t b = n * (size_t)k / (size_t)threads, e = n * (size_t)(k + 1) / (size_t)threads;
            th.emplace_back([=] { f(b, e); });
        }
        for (auto& t_ : th) t_.join();
    }
    void propagate(const double* args0, const double* obs, int n_steps, bool init) {
        const int d = model->dim_state;
        for (int k = 0; k < n_steps; ++k) {
            const int64_t tk = t;
            const double* yk = obs + (size_t)k * model->dim_obs;
            parallel_for([&](size_t b, size_t e) {
                std::vector<double> prev((size_t)d), next((size_t)d);
                for (size_t i = b; i < e; ++i) {
                    Rng r; r.seed = seed; r.slot = (uint32_t)(slot_offset + i); r.step = (uint32_t)tk;
                    for (int j = 0; j < d; ++j) prev[(size_t)j] = (tk == 0) ? (args0 ? args0[j] : 0.) : x[i * d + j];
                    const double w = model->kernel(r, tk, prev.data(), next.data(), yk);
                    for (int j = 0; j < d; ++j) x[i * d + j] = next[(size_t)j];
                    logw[i] = (init && k == 0) ? w : logw[i] + w;
                }
            });
            ++t;
        }
    }
    void init_step(const double* args0, const double* obs, int n_steps) {
        if (initialised) throw Panic("init_step called twice");
        propagate(args0, obs, n_steps, true);
        initialised = true;
    }
    void step(const double* obs, int n_steps) {
        if (!initialised) throw Panic("step before init_step");
        propagate(nullptr, obs, n_steps, false);
    }
    double ess(bool fresh) const {
        if (!fresh) return ess_stale;
        if (canonical) return canonical_normalize(logw, n_global).ess;
        const double L = logsumexp(logw);
        std::vector<double> two(n);
        for (size_t i = 0; i < n; ++i) two[i] = 2.0 * (logw[i] - L);
        return o_exp(-logsumexp(two));
    }
    int scheme = 0;  // 0 multinomial, 1 systematic (canonical mode only)
    double resample() {
        if (!initialised) throw Panic("resample before init_step");
        if (scheme == 1 && !canonical) throw Panic("systematic resampling: canonical mode only (no reference counterpart)");
        const int d = model->dim_state;
        double L;
        if (!canonical) {
            L = logsumexp(logw);
            std::vector<double> cdf(n), two(n);
            double run = 0., sum = 0.;
            for (size_t i = 0; i < n; ++i) {
                lnw[i] = logw[i] - L;
                two[i] = 2.0 * lnw[i];
                const double w = o_exp(lnw[i]);
                sum += w;
                run += w; cdf[i] = run;
            }
            if (!(std::fabs(sum - 1.0) <= 1e-8)) throw Panic("categorical: probs do not sum to 1 (eps 1e-8)");
            ess_stale = o_exp(-logsumexp(two));
            log_ml += L - o_ln((double)n);
            parallel_for([&](size_t b, size_t e) {
                for (size_t i = b; i < e; ++i) {
                    Rng r; r.seed = seed; r.slot = (uint32_t)(slot_offset + i); r.step = resample_count; r.at(DOM_RESAMPLE, 0);
                    const double u = r.u01();
                    if (!(0. < u)) throw Panic("categorical returned -1 (u == 0)");
                    const size_t p = (size_t)(std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin());
                    if (p >= n) throw Panic("categorical: index out of bounds");
                    parents[i] = (uint32_t)p;
                }
            });
        } else {
            CanonNorm c = canonical_normalize(logw, n_global);
            if (c.m == -INFINITY) throw Panic("all log-weights are -inf");
            L = c.L;
            ess_stale = c.ess;
            log_ml += L - o_ln((double)n_global);
            parallel_for([&](size_t b, size_t e) {
                for (size_t i = b; i < e; ++i) {
                    Rng r; r.seed = seed; r.slot = (uint32_t)(slot_offset + i); r.step = resample_count; r.at(DOM_RESAMPLE, 0);
                    if (scheme == 1) parents[i] = (uint32_t)canonical_parent(c.cum, canonical_target_systematic(slot_offset + i, canonical_systematic_k32(seed, resample_count), c.Q, n_global));
                    else parents[i] = (uint32_t)canonical_parent(c.cum, canonical_target(r.u52(), c.Q));
                }
            });
        }
        ++resample_count;
        for (size_t i = 0; i < n; ++i)
            for (int j = 0; j < d; ++j) x_tmp[i * d + j] = x[(size_t)parents[i] * d + j];
        x.swap(x_tmp);
        std::fill(logw.begin(), logw.end(), 0.);
        return L;
    }
    // ---- sharded resample, phase by phase (mirrors mp_pf_shard_* of include/modppl_hip.h) ----------
    CanonNorm sh_c;
    std::vector<uint32_t> sh_req_slot;
    double sh_L = 0.;
    double shard_local_max() const { double m = -INFINITY; for (double w : logw) m = std::fmax(m, w); return m; }
    void shard_normalize(double gmax, uint64_t* totals) {
        if (!initialised) throw Panic("resample before init_step");
        sh_c = canonical_normalize(logw, n_global, &gmax);
        totals[0] = sh_c.Q; totals[1] = sh_c.Q2;
    }
    static void shard_scalars(const uint64_t* totals_all, int world, int S, double m, double* L, double* ess, uint64_t* Q) {
        uint64_t q = 0, q2 = 0;
        for (int r = 0; r < world; ++r) { q += totals_all[2 * r]; q2 += totals_all[2 * r + 1]; }
        const double inv = ldexp_pow2(-S);
        const double Qs = (double)q * inv, Q2s = (double)q2 * inv;
        *L = m + mp_log(Qs);
        *ess = (Qs * Qs) / Q2s;
        *Q = q;
    }
    void shard_route(const uint64_t* totals_all, int world, int rank, uint64_t* req_out, int64_t* send_counts) {
        (void)rank;
        if (sh_c.m == -INFINITY) throw Panic("all log-weights are -inf");
        double L, ess; uint64_t Q;
        shard_scalars(totals_all, world, sh_c.S, sh_c.m, &L, &ess, &Q);
        std::vector<uint64_t> incl((size_t)world);
        uint64_t run = 0;
        for (int r = 0; r < world; ++r) { run += totals_all[2 * r]; incl[(size_t)r] = run; }
        std::vector<int> dest(n);
        std::vector<uint64_t> lt(n);
        for (int r = 0; r < world; ++r) send_counts[r] = 0;
        for (size_t i = 0; i < n; ++i) {
            Rng r; r.seed = seed; r.slot = (uint32_t)(slot_offset + i); r.step = resample_count; r.at(DOM_RESAMPLE, 0);
            const uint64_t target = scheme == 1 ? canonical_target_systematic(slot_offset + i, canonical_systematic_k32(seed, resample_count), Q, n_global)
                                                : canonical_target(r.u52(), Q);
            int s_ = 0;
            while (s_ < world - 1 && incl[(size_t)s_] < target) ++s_;
            dest[i] = s_;
            lt[i] = target - (s_ ? incl[(size_t)s_ - 1] : 0);
            send_counts[s_] += 1;
        }
        std::vector<uint64_t> start((size_t)world, 0);
        for (int r = 1; r < world; ++r) start[(size_t)r] = start[(size_t)r - 1] + (uint64_t)send_counts[r - 1];
        sh_req_slot.assign(n, 0);
        for (size_t i = 0; i < n; ++i) {  // stable
            const uint64_t pos = start[(size_t)dest[i]]++;
            req_out[pos] = lt[i];
            sh_req_slot[pos] = (uint32_t)i;
        }
        sh_L = L;
        ess_stale = ess;
        log_ml += L - o_ln((double)n_global);
    }
    void shard_resolve(const uint64_t* req_in, uint64_t n_req, double* rows) const {
        const int d = model->dim_state;
        for (uint64_t q = 0; q < n_req; ++q) {
            const size_t p = canonical_parent(sh_c.cum, req_in[q]);
            for (int j = 0; j < d; ++j) rows[q * (uint64_t)(d + 1) + j] = x[p * d + j];
            rows[q * (uint64_t)(d + 1) + d] = (double)(slot_offset + p);
        }
    }
    double shard_scatter(const double* rows) {
        const int d = model->dim_state;
        for (size_t pos = 0; pos < n; ++pos) {
            const size_t i = sh_req_slot[pos];
            for (int j = 0; j < d; ++j) x_tmp[i * d + j] = rows[pos * (size_t)(d + 1) + j];
            parents[i] = (uint32_t)rows[pos * (size_t)(d + 1) + d];
        }
        x.swap(x_tmp);
        std::fill(logw.begin(), logw.end(), 0.);
        ++resample_count;
        return sh_L;
    }
    void shard_query(const uint64_t* totals_all, int world, double* lml, double* ess) const {
        double L, e; uint64_t Q;
        shard_scalars(totals_all, world, sh_c.S, sh_c.m, &L, &e, &Q);
        if (lml) *lml = log_ml + L - o_ln((double)n_global);
        if (ess) *ess = e;
    }

    double log_ml_estimate() const {
        if (canonical) return log_ml + canonical_normalize(logw, n_global).L - o_ln((double)n_global);
        return log_ml + logsumexp(logw) - o_ln((double)n);
    }
};

}  // namespace oracle
