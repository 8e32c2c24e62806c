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

// static models of tests/importance.rs (one generate; later steps leave the state alone)
struct SoaPointed : SoaModel {   // tests/dyngenfns/simple.rs:27-34
    Bounds b; Mat cov;
    SoaPointed(Bounds b_, Mat c) : b(b_), cov(std::move(c)) { dim_state = 2; dim_obs = 2; }
    double kernel(Rng& r, int64_t t, const double* prev, double* next, const double* obs) const override {
        if (t != 0) { next[0] = prev[0]; next[1] = prev[1]; return 0.; }
        r.at(DOM_MODEL, 0);
        const Vec latent = uniform_2d.random(r, b);
        next[0] = latent[0]; next[1] = latent[1];
        return mvnormal.logpdf(Vec{obs[0], obs[1]}, MvNormalParams{latent, cov});
    }
};
struct SoaLine : SoaModel {      // tests/dyngenfns/simple.rs:9-24
    Vec xs;
    explicit SoaLine(Vec xs_) : xs(std::move(xs_)) { dim_state = 2; dim_obs = (int)xs.size(); }
    double kernel(Rng& r, int64_t t, const double* prev, double* next, const double* obs) const override {
        if (t != 0) { next[0] = prev[0]; next[1] = prev[1]; return 0.; }
        r.at(DOM_MODEL, 0); const double slope = normal.random(r, {0., 1.});
        r.at(DOM_MODEL, 1); const double intercept = normal.random(r, {0., 2.});
        double w = 0.;
        for (size_t i = 0; i < xs.size(); ++i) w += normal.logpdf(obs[i], {slope * xs[i] + intercept, 0.1});
        next[0] = slope; next[1] = intercept;
        return w;
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

// dense LGSSM (MP_MODEL_LGSSM_DENSE): two mvnormal sites per step, per-call determinant / inverse / transform as the
// reference's mvnormal does (mvnormal.rs:14-38)
struct SoaLgssmDense : SoaModel {
    int D; double sig0; Mat A, Q, R;
    SoaLgssmDense(int D_, double sig0_, Mat A_, Mat Q_, Mat R_) : D(D_), sig0(sig0_), A(std::move(A_)), Q(std::move(Q_)), R(std::move(R_)) { dim_state = D; dim_obs = D; }
    double kernel(Rng& r, int64_t t, const double* prev, double* next, const double* obs) const override {
        Vec mean((size_t)D, 0.);
        Mat cov = Q;
        if (t == 0) {
            cov = Mat(D, std::vector<double>((size_t)D * D, 0.));
            for (int i = 0; i < D; ++i) cov(i, i) = sig0 * sig0;
        } else {
            for (int i = 0; i < D; ++i) mean[(size_t)i] = dense_dot(&A.a[(size_t)i * D], 1, prev, 1, D);
        }
        r.at(DOM_MODEL, 0);
        const Vec x = mvnormal.random_dense(r, MvNormalParams{mean, cov});
        for (int i = 0; i < D; ++i) next[i] = x[(size_t)i];
        return mvnormal.logpdf_dense(Vec(obs, obs + D), MvNormalParams{x, R});
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
    // lib.rs:34-45 with the exps evaluated in parallel and summed in index order (bit-identical to logsumexp())
    double logsumexp_mt(const std::vector<double>& xs) {
        if (threads <= 1) return logsumexp(xs);
        double max = -INFINITY;
        for (double v : xs) max = std::fmax(max, v);
        if (max == -INFINITY) return -INFINITY;
        std::vector<double> ev(xs.size());
        parallel_for([&](size_t b, size_t e) {
            for (size_t i = b; i < e; ++i) ev[i] = o_exp(xs[i] - max);
        });
        double sum_exp = 0.;
        for (double v : ev) sum_exp += v;
        return max + o_ln(sum_exp);
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
        if (canonical) return canonical_normalize(logw, n_global).c.ess;
        const double L = logsumexp(logw);
        std::vector<double> two(n);
        for (size_t i = 0; i < n; ++i) two[i] = 2.0 * (logw[i] - L);
        return o_exp(-logsumexp(two));
    }
    int scheme = 0;  // 0 multinomial, 1 systematic, 2 stratified (1, 2: canonical mode only)
    double resample() {
        if (!initialised) throw Panic("resample before init_step");
        if (scheme != 0 && !canonical) throw Panic("systematic / stratified resampling: canonical mode only (no reference counterpart)");
        const int d = model->dim_state;
        double L;
        if (!canonical) {
            // Element-wise work (exp) may run on several threads; every SUM keeps the reference's sequential order
            // (lib.rs:41 fold, categorical.rs:26-31 running sum), so the results do not depend on the thread count.
            L = logsumexp_mt(logw);
            std::vector<double> cdf(n), two(n), wv(n);
            parallel_for([&](size_t b, size_t e) {
                for (size_t i = b; i < e; ++i) {
                    lnw[i] = logw[i] - L;
                    two[i] = 2.0 * lnw[i];
                    wv[i] = o_exp(lnw[i]);
                }
            });
            double run = 0., sum = 0.;
            for (size_t i = 0; i < n; ++i) {
                sum += wv[i];
                run += wv[i]; cdf[i] = run;
            }
            if (!(std::fabs(sum - 1.0) <= 1e-8)) throw Panic("categorical: probs do not sum to 1 (eps 1e-8)");
            ess_stale = o_exp(-logsumexp_mt(two));
            log_ml += L - o_ln((double)n);
            parallel_for([&](size_t b, size_t e) {
                for (size_t i = b; i < e; ++i) {
                    Rng r = resample_rng(seed, DOM_RESAMPLE, resample_count, slot_offset + i);
                    const double u = r.u01();
                    if (!(0. < u)) throw Panic("categorical returned -1 (u == 0)");
                    const size_t p = (size_t)(std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin());
                    if (p >= n) throw Panic("categorical: index out of bounds");
                    parents[i] = (uint32_t)p;
                }
            });
        } else {
            CanonFull c = canonical_normalize(logw, n_global);
            if (c.c.degenerate()) throw Panic("all log-weights are -inf");
            L = c.c.L;
            ess_stale = c.c.ess;
            log_ml += L - o_ln((double)n_global);
            parallel_for([&](size_t b, size_t e) {
                for (size_t i = b; i < e; ++i) {
                    Rng r = resample_rng(seed, DOM_RESAMPLE, resample_count, slot_offset + i);
                    if (scheme != 0) parents[i] = (uint32_t)canonical_parent(c, canonical_target_lattice(scheme, seed, resample_count, slot_offset + i, c.c.Q, n_global));
                    else parents[i] = (uint32_t)canonical_parent(c, canonical_target(r.u52(), c.c.Q));
                }
            });
        }
        ++resample_count;
        parallel_for([&](size_t b, size_t e) {
            for (size_t i = b; i < e; ++i)
                for (int j = 0; j < d; ++j) x_tmp[i * d + j] = x[(size_t)parents[i] * d + j];
        });
        x.swap(x_tmp);
        std::fill(logw.begin(), logw.end(), 0.);
        return L;
    }
    // ---- sharded resample, phase by phase (mirrors mp_pf_shard_* of include/modppl_hip.h) ----------
    // Shards are tile-aligned (slot_offset and n multiples of 2048), so a shard's level-0 tiles are tiles of the job.
    CanonTiles sh_t;
    CanonNorm sh_c;
    std::vector<uint32_t> sh_req_slot;
    void shard_tiles(double* tm, uint64_t* tW, uint64_t* tW2) {
        if (!initialised) throw Panic("resample before init_step");
        if (slot_offset % CANON_TILE) throw Panic("shards must start at a tile boundary");
        sh_t = canonical_tiles(logw);
        for (size_t b = 0; b < sh_t.m.size(); ++b) { tm[b] = sh_t.m[b]; tW[b] = sh_t.W[b]; tW2[b] = sh_t.W2[b]; }
    }
    size_t n_tiles() const { return (n + CANON_TILE - 1) / CANON_TILE; }
    void shard_combine(const double* tm_all, const uint64_t* tW_all, const uint64_t* tW2_all, size_t nt_all) {
        sh_c = canonical_combine(std::vector<double>(tm_all, tm_all + nt_all), std::vector<uint64_t>(tW_all, tW_all + nt_all),
                                 std::vector<uint64_t>(tW2_all, tW2_all + nt_all), n_global);
    }
    // requests are pairs (tile index inside the owner's shard, tile-local target)
    void shard_route(const double* tm_all, const uint64_t* tW_all, const uint64_t* tW2_all, size_t nt_all, int world, int rank,
                     uint64_t* req_out, int64_t* send_counts) {
        (void)rank;
        shard_combine(tm_all, tW_all, tW2_all, nt_all);
        if (sh_c.degenerate()) throw Panic("all log-weights are -inf");
        const size_t nt_local = nt_all / (size_t)world;
        std::vector<int> dest(n);
        std::vector<uint64_t> rt(n), rl(n);
        for (int r = 0; r < world; ++r) send_counts[r] = 0;
        for (size_t i = 0; i < n; ++i) {
            Rng r = resample_rng(seed, DOM_RESAMPLE, resample_count, slot_offset + i);
            const uint64_t target = scheme != 0 ? canonical_target_lattice(scheme, seed, resample_count, slot_offset + i, sh_c.Q, n_global)
                                                : canonical_target(r.u52(), sh_c.Q);
            size_t tile; uint64_t lt;
            canonical_locate(sh_c, target, &tile, &lt);
            dest[i] = (int)(tile / nt_local);
            rt[i] = tile % nt_local;
            rl[i] = lt;
            send_counts[dest[i]] += 1;
        }
        std::vector<uint64_t> start((size_t)world, 0);
        for (int r = 1; r < world; ++r) start[(size_t)r] = start[(size_t)r - 1] + (uint64_t)send_counts[r - 1];
        sh_req_slot.assign(n, 0);
        for (size_t i = 0; i < n; ++i) {  // stable
            const uint64_t pos = start[(size_t)dest[i]]++;
            req_out[2 * pos] = rt[i];
            req_out[2 * pos + 1] = rl[i];
            sh_req_slot[pos] = (uint32_t)i;
        }
        ess_stale = sh_c.ess;
        log_ml += sh_c.L - o_ln((double)n_global);
    }
    void shard_resolve(const uint64_t* req_in, uint64_t n_req, double* rows) const {
        const int d = model->dim_state;
        for (uint64_t q = 0; q < n_req; ++q) {
            const size_t p = canonical_row(sh_t.cum, (size_t)req_in[2 * q], req_in[2 * q + 1]);
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
        return sh_c.L;
    }
    void shard_query(const double* tm_all, const uint64_t* tW_all, const uint64_t* tW2_all, size_t nt_all, double* lml, double* ess) {
        shard_combine(tm_all, tW_all, tW2_all, nt_all);
        if (lml) *lml = log_ml + sh_c.L - o_ln((double)n_global);
        if (ess) *ess = sh_c.ess;
    }

    // ---- "owner keeps" form of the sharded resample (mirrors mp_pf_shard_owned_* of include/modppl_hip.h) ----
    // The N draws are those of the single filter (same Philox counters, same targets, hence the same parent for every
    // draw g as particle_filter.rs:37-41 over the whole population); what changes is where an offspring is put: it stays
    // on the rank that owns its parent.  A rank's offspring, in the order of their draws g, fill its slots 0, 1, ...; what
    // exceeds n slots is its surplus.  Unit u of the job's surplus (donors in rank order, each donor's offspring n, n+1,
    // ... in order) fills unit u of the job's deficit (receivers in rank order, each receiver's slots c_s, c_s+1, ...).
    std::vector<uint32_t> sh_own;        // local parent row of every own offspring, in draw order
    std::vector<uint64_t> sh_call;       // offspring per rank
    void shard_owned_count(const double* tm_all, const uint64_t* tW_all, const uint64_t* tW2_all, size_t nt_all, int world, int rank,
                           uint64_t* c_all) {
        shard_combine(tm_all, tW_all, tW2_all, nt_all);
        if (sh_c.degenerate()) throw Panic("all log-weights are -inf");
        const size_t nt_local = nt_all / (size_t)world;
        sh_own.clear();
        sh_call.assign((size_t)world, 0);
        if (scheme == 3) {
            // split multinomial (inference.hpp: canonical_split_counts): the counts from the rank masses, then this rank's own
            // c_r draws inside its own share (lo, lo + M_r] of the job's fixed-point mass
            std::vector<uint64_t> mass((size_t)world);
            for (int r = 0; r < world; ++r)
                mass[(size_t)r] = sh_c.inclT[(size_t)(r + 1) * nt_local - 1] - (r ? sh_c.inclT[(size_t)r * nt_local - 1] : 0);
            sh_call = canonical_split_counts(mass, n_global, seed, resample_count);
            const uint64_t lo = rank ? sh_c.inclT[(size_t)rank * nt_local - 1] : 0;
            for (uint64_t j = 0; j < sh_call[(size_t)rank]; ++j) {
                const uint64_t target = lo + canonical_target(canonical_split_u52(seed, resample_count, rank, j), mass[(size_t)rank]);
                size_t tile; uint64_t lt;
                canonical_locate(sh_c, target, &tile, &lt);
                if ((int)(tile / nt_local) != rank) throw Panic("split multinomial: a rank's own draw left its tiles");
                sh_own.push_back((uint32_t)canonical_row(sh_t.cum, tile % nt_local, lt));
            }
            for (int r = 0; r < world; ++r) c_all[r] = sh_call[(size_t)r];
            ess_stale = sh_c.ess;
            log_ml += sh_c.L - o_ln((double)n_global);
            return;
        }
        for (uint64_t g = 0; g < n_global; ++g) {
            Rng r = resample_rng(seed, DOM_RESAMPLE, resample_count, g);
            const uint64_t target = scheme != 0 ? canonical_target_lattice(scheme, seed, resample_count, g, sh_c.Q, n_global)
                                                : canonical_target(r.u52(), sh_c.Q);
            size_t tile; uint64_t lt;
            canonical_locate(sh_c, target, &tile, &lt);
            const int own = (int)(tile / nt_local);
            sh_call[(size_t)own] += 1;
            if (own == rank) sh_own.push_back((uint32_t)canonical_row(sh_t.cum, tile % nt_local, lt));
        }
        for (int r = 0; r < world; ++r) c_all[r] = sh_call[(size_t)r];
        ess_stale = sh_c.ess;
        log_ml += sh_c.L - o_ln((double)n_global);
    }
    // slots [0, min(c, n)) <- own offspring in draw order; the surplus rows {x[0..d), global parent id} in unit order to `send`
    // (its length is max(c - n, 0)); returns the number of rows written to `send`
    uint64_t shard_owned_expand(int rank, double* send) {
        const int d = model->dim_state;
        uint64_t sent = 0;
        (void)rank;
        for (size_t p = 0; p < sh_own.size(); ++p) {
            const size_t i = sh_own[p];
            if (p < n) {
                for (int j = 0; j < d; ++j) x_tmp[p * d + j] = x[i * d + j];
                parents[p] = (uint32_t)(slot_offset + i);
            } else {
                for (int j = 0; j < d; ++j) send[sent * (uint64_t)(d + 1) + j] = x[i * d + j];
                send[sent * (uint64_t)(d + 1) + d] = (double)(slot_offset + i);
                ++sent;
            }
        }
        return sent;
    }
    // slots [c, n) <- the received rows in unit order (n_recv = max(n - c, 0) of them)
    double shard_owned_adopt(int rank, const double* recv, uint64_t n_recv) {
        const int d = model->dim_state;
        const uint64_t c = sh_call[(size_t)rank];
        if ((c < n ? n - c : 0) != n_recv) throw Panic("owned_adopt: row count does not match the deficit of this rank");
        for (uint64_t k = 0; k < n_recv; ++k) {
            const size_t p = (size_t)(c + k);
            for (int j = 0; j < d; ++j) x_tmp[p * d + j] = recv[k * (uint64_t)(d + 1) + j];
            parents[p] = (uint32_t)recv[k * (uint64_t)(d + 1) + d];
        }
        x.swap(x_tmp);
        std::fill(logw.begin(), logw.end(), 0.);
        ++resample_count;
        return sh_c.L;
    }

    double log_ml_estimate() const {
        if (canonical) return log_ml + canonical_normalize(logw, n_global).c.L - o_ln((double)n_global);
        return log_ml + logsumexp(logw) - o_ln((double)n);
    }
};

}  // namespace oracle
