// ORACLE — TEST INFRASTRUCTURE ONLY (see rng.hpp).
//
// models.hpp — the models of the path, written against the restated dynamic handler exactly as
// a modppl user would write them with `dyngen!` (`dist(args) %= addr` == g.sample_at(dist,args,addr),
// `f(args) /= addr` == g.trace_at(f,args,addr); modppl-macros/src/lib.rs:20-113):
//   * spiral_model        — modppl/tests/dyngenfns/unfold.rs:14-32 (the reference's own Unfold model)
//   * hierarchical_model  — modppl/tests/dyngenfns/hierarchical.rs:18-70 (+ drift proposal)
//   * HMM                 — modppl/tests/hmm/model.rs:33-80, forward algorithm tests/hmm/forward.rs:3-22
//   * LGSSM / bearings    — BASELINE.json configs; not in the reference, written in
//                           the same DSL so that the same handler semantics apply.
// Site ids (the Philox "site" of each address) are the static slot ids of the device kernels.
#pragma once
#include "inference.hpp"

namespace oracle {

using Vec = std::vector<double>;

// ---------------------------------------------------------------------------------------
// LGSSM d=1:  t==0: x ~ normal(mu0, sig0);  t>0: x ~ normal(a*x_prev, sig_x);  y ~ normal(x, sig_y)
// ---------------------------------------------------------------------------------------
struct LgssmParams { double mu0 = 0., sig0 = 1., a = 0.9, sig_x = 0.5, sig_y = 1.0; };

inline DynUnfold<double> make_lgssm_model(LgssmParams p) {
    using A = std::pair<int64_t, double>;
    using H = DynGenFnHandler<A, double>;
    DynGenFn<A, double> k(
        [p](H& g, A ta) -> double {
            const int64_t t = ta.first;
            const double prev = ta.second;
            double x;
            if (t == 0) x = g.template sample_at<double>(normal, NormalParams{p.mu0, p.sig0}, "x");
            else x = g.template sample_at<double>(normal, NormalParams{p.a * prev, p.sig_x}, "x");
            g.template sample_at<double>(normal, NormalParams{x, p.sig_y}, "y");
            return x;
        },
        [](const std::string& a) -> uint32_t { return a == "x" ? 0u : 1u; });
    return DynUnfold<double>(std::move(k));
}

// scalar Kalman filter log marginal likelihood: the closed-form ground truth for the LGSSM.
inline double kalman_log_ml(LgssmParams p, const Vec& ys) {
    double mean = p.mu0, var = p.sig0 * p.sig0, ll = 0.;
    for (size_t t = 0; t < ys.size(); ++t) {
        if (t > 0) { mean = p.a * mean; var = p.a * p.a * var + p.sig_x * p.sig_x; }
        const double s = var + p.sig_y * p.sig_y;
        const double r = ys[t] - mean;
        ll += -0.5 * (std::log(2. * M_PI) + std::log(s) + r * r / s);
        const double kgain = var / s;
        mean += kgain * r;
        var = (1. - kgain) * var;
    }
    return ll;
}

// Observations simulated once from the model itself (DOM_DATA stream, slot 0).
inline Vec lgssm_simulate_observations(LgssmParams p, uint64_t seed, int T) {
    Vec ys;
    double x = 0.;
    for (int t = 0; t < T; ++t) {
        Rng r; r.seed = seed; r.slot = 0; r.step = (uint32_t)t;
        r.at(DOM_DATA, 0);
        x = (t == 0) ? normal.random(r, {p.mu0, p.sig0}) : normal.random(r, {p.a * x, p.sig_x});
        r.at(DOM_DATA, 1);
        ys.push_back(normal.random(r, {x, p.sig_y}));
    }
    return ys;
}

// ---------------------------------------------------------------------------------------
// spiral model — tests/dyngenfns/unfold.rs:10-32
// ---------------------------------------------------------------------------------------
inline double o_cos(double x) { return canonical_mode() ? mp_cos(x) : std::cos(x); }
inline double o_sin(double x) { return canonical_mode() ? mp_sin(x) : std::sin(x); }
inline double o_atan2(double y, double x) { return canonical_mode() ? mp_atan2(y, x) : std::atan2(y, x); }
inline Vec polar_to_cartesian(const Vec& pol) { return {pol[0] * o_cos(pol[1]), pol[0] * o_sin(pol[1])}; }

inline DynUnfold<Vec> make_spiral_model() {
    using A = std::pair<int64_t, Vec>;
    using H = DynGenFnHandler<A, Vec>;
    DynGenFn<A, Vec> k(
        [](H& g, A ta) -> Vec {
            const int64_t t = ta.first;
            const Vec& prev_pol = ta.second;
            Vec pol, pos;
            if (t == 0) {
                const double r = g.template sample_at<double>(uniform, UniformParams{0., 1.}, "r");
                const double theta = g.template sample_at<double>(uniform, UniformParams{0., 2. * M_PI}, "theta");
                pol = {r, theta};
                pos = polar_to_cartesian(pol);
            } else {
                const double dr = g.template sample_at<double>(normal, NormalParams{0., 0.1}, "dr");
                const double dtheta = g.template sample_at<double>(normal, NormalParams{0.4, 0.2}, "dtheta");
                pol = {prev_pol[0] + dr, prev_pol[1] + dtheta};
                pos = polar_to_cartesian(pol);
            }
            g.template sample_at<Vec>(mvnormal, MvNormalParams{pos, Mat(2, {0.001, 0., 0., 0.001})}, "obs");
            return pol;
        },
        [](const std::string& a) -> uint32_t {
            if (a == "r" || a == "dr") return 0u;
            if (a == "theta" || a == "dtheta") return 1u;
            return 2u;  // obs
        });
    return DynUnfold<Vec>(std::move(k));
}

// ---------------------------------------------------------------------------------------
// Bearings-only tracker d=4 (BASELINE.json config 3), dt = 1.  State (px,py,vx,vy).
// ---------------------------------------------------------------------------------------
struct BearingsParams { double p0x, p0y, sig_p0, sig_v0, sig_a, sig_theta; };
inline DynUnfold<Vec> make_bearings_model(BearingsParams p) {
    using A = std::pair<int64_t, Vec>;
    using H = DynGenFnHandler<A, Vec>;
    DynGenFn<A, Vec> k(
        [p](H& g, A ta) -> Vec {
            const int64_t t = ta.first;
            const Vec& prev = ta.second;
            double px, py, vx, vy;
            if (t == 0) {
                px = g.template sample_at<double>(normal, NormalParams{p.p0x, p.sig_p0}, "px");
                py = g.template sample_at<double>(normal, NormalParams{p.p0y, p.sig_p0}, "py");
                vx = g.template sample_at<double>(normal, NormalParams{0., p.sig_v0}, "vx");
                vy = g.template sample_at<double>(normal, NormalParams{0., p.sig_v0}, "vy");
            } else {
                const double ax = g.template sample_at<double>(normal, NormalParams{0., p.sig_a}, "ax");
                const double ay = g.template sample_at<double>(normal, NormalParams{0., p.sig_a}, "ay");
                px = (prev[0] + prev[2]) + 0.5 * ax;
                py = (prev[1] + prev[3]) + 0.5 * ay;
                vx = prev[2] + ax;
                vy = prev[3] + ay;
            }
            g.template sample_at<double>(normal, NormalParams{o_atan2(py, px), p.sig_theta}, "theta");
            return Vec{px, py, vx, vy};
        },
        [](const std::string& a) -> uint32_t {
            if (a == "px" || a == "ax") return 0u;
            if (a == "py" || a == "ay") return 1u;
            if (a == "vx") return 2u;
            if (a == "vy") return 3u;
            return 4u;  // theta
        });
    return DynUnfold<Vec>(std::move(k));
}

// ---------------------------------------------------------------------------------------
// Banded LGSSM d=D (BASELINE.json config 5): sites "x/j" -> j, "y/j" -> D + j.
// ---------------------------------------------------------------------------------------
struct BandParams { int D; double a, band, sig0, sig_x, sig_y; };
inline DynUnfold<Vec> make_lgssm_band_model(BandParams p) {
    using A = std::pair<int64_t, Vec>;
    using H = DynGenFnHandler<A, Vec>;
    const int D = p.D;
    DynGenFn<A, Vec> k(
        [p](H& g, A ta) -> Vec {
            const int64_t t = ta.first;
            const Vec& prev = ta.second;
            const int D_ = p.D;
            Vec x((size_t)D_);
            for (int j = 0; j < D_; ++j) {
                const std::string sj = std::to_string(j);
                if (t == 0) {
                    x[(size_t)j] = g.template sample_at<double>(normal, NormalParams{0., p.sig0}, "x/" + sj);
                } else {
                    const double nb = (j > 0 ? prev[(size_t)j - 1] : 0.) + (j < D_ - 1 ? prev[(size_t)j + 1] : 0.);
                    x[(size_t)j] = g.template sample_at<double>(normal, NormalParams{p.a * (prev[(size_t)j] + p.band * nb), p.sig_x}, "x/" + sj);
                }
                g.template sample_at<double>(normal, NormalParams{x[(size_t)j], p.sig_y}, "y/" + sj);
            }
            return x;
        },
        [D](const std::string& a) -> uint32_t {
            const uint32_t j = (uint32_t)std::stoi(a.substr(2));
            return a[0] == 'x' ? j : (uint32_t)D + j;
        });
    return DynUnfold<Vec>(std::move(k));
}

// ---------------------------------------------------------------------------------------
// static models of tests/importance.rs, wrapped as single-step Unfold kernels (t > 0: no sites, state unchanged)
//   pointed_2d_model — tests/dyngenfns/simple.rs:27-34;  line_model + obs_model — simple.rs:9-24
// ---------------------------------------------------------------------------------------
inline DynUnfold<Vec> make_pointed_unfold(Bounds b, Mat cov) {
    using A = std::pair<int64_t, Vec>;
    using H = DynGenFnHandler<A, Vec>;
    DynGenFn<A, Vec> k(
        [b, cov](H& g, A ta) -> Vec {
            if (ta.first != 0) return ta.second;
            const Vec latent = g.template sample_at<Vec>(uniform_2d, b, "latent");
            g.template sample_at<Vec>(mvnormal, MvNormalParams{latent, cov}, "obs");
            return latent;
        },
        [](const std::string& a) -> uint32_t { return a == "latent" ? 0u : 1u; });
    return DynUnfold<Vec>(std::move(k));
}
inline DynUnfold<Vec> make_line_unfold(Vec xs) {
    using A = std::pair<int64_t, Vec>;
    using H = DynGenFnHandler<A, Vec>;
    // obs_model(slope, intercept, xs) /= "ys": a sub-generative-function call, as in the reference
    auto obs_model = std::make_shared<DynGenFn<std::tuple<double, double, Vec>, Vec>>(
        [](DynGenFnHandler<std::tuple<double, double, Vec>, Vec>& g, std::tuple<double, double, Vec> a) -> Vec {
            const auto& [slope, intercept, xs_] = a;
            Vec out;
            for (size_t i = 0; i < xs_.size(); ++i)
                out.push_back(g.template sample_at<double>(normal, NormalParams{slope * xs_[i] + intercept, 0.1}, std::to_string(i)));
            return out;
        },
        [](const std::string& a) -> uint32_t { return 2u + (uint32_t)std::stoul(a); });
    DynGenFn<A, Vec> k(
        [xs, obs_model](H& g, A ta) -> Vec {
            if (ta.first != 0) return ta.second;
            const double slope = g.template sample_at<double>(normal, NormalParams{0., 1.}, "slope");
            const double intercept = g.template sample_at<double>(normal, NormalParams{0., 2.}, "intercept");
            g.template trace_at<std::tuple<double, double, Vec>, Vec>(*obs_model, {slope, intercept, xs}, "ys");
            return Vec{slope, intercept};
        },
        [](const std::string& a) -> uint32_t { return a == "slope" ? 0u : (a == "intercept" ? 1u : 2u); });
    return DynUnfold<Vec>(std::move(k));
}

// ---------------------------------------------------------------------------------------
// hierarchical model + proposals — tests/dyngenfns/hierarchical.rs:18-70
// ---------------------------------------------------------------------------------------
struct Hierarchical {
    using CoefL = std::pair<double, double>;
    using CoefQ = std::tuple<double, double, double>;
    static uint32_t site_of(const std::string& a) {
        if (a == "is_linear") return 0u;
        if (a == "a" || a == "coeffs/a") return 1u;
        if (a == "b" || a == "coeffs/b") return 2u;
        if (a == "c" || a == "coeffs/c") return 3u;
        // "(y, i)" -> 4 + i: constrained everywhere except under a regenerate with an empty mask (= the whole schema,
        // dyngenfn.rs:571), which redraws the observed sites too: each needs a stream of its own
        const size_t c = a.find(", ");
        return 4u + (c == std::string::npos ? 0u : (uint32_t)std::stoul(a.substr(c + 2)));
    }
    DynGenFn<int, CoefL> linear;        // Args = unit -> int 0
    DynGenFn<int, CoefQ> quadratic;
    DynGenFn<Vec, Vec> model;
    Hierarchical() {
        linear = DynGenFn<int, CoefL>(
            [](DynGenFnHandler<int, CoefL>& g, int) -> CoefL {
                const double a = g.template sample_at<double>(normal, NormalParams{0., 1.}, "a");
                const double b = g.template sample_at<double>(normal, NormalParams{0., 1.}, "b");
                return {a, b};
            }, site_of);
        quadratic = DynGenFn<int, CoefQ>(
            [](DynGenFnHandler<int, CoefQ>& g, int) -> CoefQ {
                const double a = g.template sample_at<double>(normal, NormalParams{0., 1.}, "a");
                const double b = g.template sample_at<double>(normal, NormalParams{0., 1.}, "b");
                const double c = g.template sample_at<double>(normal, NormalParams{0., 1.}, "c");
                return {a, b, c};
            }, site_of);
        model = DynGenFn<Vec, Vec>(
            [this](DynGenFnHandler<Vec, Vec>& g, Vec xs) -> Vec {
                const double noise = 0.1;
                Vec out;
                if (g.template sample_at<bool>(bernoulli, 0.7, "is_linear")) {
                    const CoefL co = g.template trace_at<int, CoefL>(linear, 0, "coeffs");
                    for (size_t i = 0; i < xs.size(); ++i)
                        out.push_back(g.template sample_at<double>(normal, NormalParams{co.first + co.second * xs[i], noise},
                                                                   "(y, " + std::to_string(i) + ")"));
                } else {
                    const CoefQ co = g.template trace_at<int, CoefQ>(quadratic, 0, "coeffs");
                    for (size_t i = 0; i < xs.size(); ++i)
                        out.push_back(g.template sample_at<double>(
                            normal, NormalParams{std::get<0>(co) + std::get<1>(co) * xs[i] + std::get<2>(co) * xs[i] * xs[i], noise},
                            "(y, " + std::to_string(i) + ")"));
                }
                return out;
            }, site_of);
    }
    using TraceT = Trace<Vec, DynTrie, Vec>;
    using PArgs = std::pair<const TraceT*, double>;
    // hierarchical_drift_proposal — hierarchical.rs:62-70
    DynGenFn<PArgs, int> drift_proposal() const {
        return DynGenFn<PArgs, int>(
            [](DynGenFnHandler<PArgs, int>& g, PArgs pa) -> int {
                const TraceT* tr = pa.first;
                const double drift_std = pa.second;
                g.template sample_at<double>(normal, NormalParams{tr->data.read<double>("coeffs/a"), drift_std}, "coeffs/a");
                g.template sample_at<double>(normal, NormalParams{tr->data.read<double>("coeffs/b"), drift_std}, "coeffs/b");
                if (!tr->data.read<bool>("is_linear"))
                    g.template sample_at<double>(normal, NormalParams{tr->data.read<double>("coeffs/c"), drift_std}, "coeffs/c");
                return 0;
            }, site_of, DOM_PROPOSAL);
    }
    // add_or_remove_param_proposal — hierarchical.rs:48-61 (the structure-changing move of tests/mh.rs:94); the
    // proposal argument `()` travels as an unused double
    DynGenFn<PArgs, int> add_or_remove_proposal() const {
        return DynGenFn<PArgs, int>(
            [](DynGenFnHandler<PArgs, int>& g, PArgs pa) -> int {
                const TraceT* tr = pa.first;
                g.template sample_at<double>(normal, NormalParams{tr->data.read<double>("coeffs/a"), 0.025}, "coeffs/a");
                g.template sample_at<double>(normal, NormalParams{tr->data.read<double>("coeffs/b"), 0.025}, "coeffs/b");
                if (!g.template sample_at<bool>(bernoulli, 0.5, "is_linear")) {
                    const double prev_c = tr->data.search("coeffs/c") ? tr->data.read<double>("coeffs/c") : 0.;
                    g.template sample_at<double>(normal, NormalParams{prev_c, 0.025}, "coeffs/c");
                }
                return 0;
            }, site_of, DOM_PROPOSAL);
    }
};

// ---------------------------------------------------------------------------------------
// pointed 2-D model + drift proposal — tests/dyngenfns/simple.rs:27-41 (driven by tests/mh.rs:50-68)
// ---------------------------------------------------------------------------------------
struct Pointed2D {
    using Args = std::pair<Bounds, Mat>;                 // (bounds, cov)
    using TraceT = Trace<Args, DynTrie, Vec>;
    using PArgs = std::pair<const TraceT*, Mat>;         // (Weak<trace>, noise)
    static uint32_t site_of(const std::string& a) { return a == "latent" ? 1u : 2u; }
    DynGenFn<Args, Vec> model;
    Pointed2D() {
        model = DynGenFn<Args, Vec>(
            [](DynGenFnHandler<Args, Vec>& g, Args a) -> Vec {
                const Vec latent = g.template sample_at<Vec>(uniform_2d, a.first, "latent");
                return g.template sample_at<Vec>(mvnormal, MvNormalParams{latent, a.second}, "obs");
            }, site_of);
    }
    DynGenFn<PArgs, int> drift_proposal() const {
        return DynGenFn<PArgs, int>(
            [](DynGenFnHandler<PArgs, int>& g, PArgs pa) -> int {
                const Vec prev_latent = pa.first->data.read<Vec>("latent");
                g.template sample_at<Vec>(mvnormal, MvNormalParams{prev_latent, pa.second}, "latent");
                return 0;
            }, site_of, DOM_PROPOSAL);
    }
};

// ---------------------------------------------------------------------------------------
// HMM — tests/hmm/model.rs:8-80, tests/hmm/forward.rs:3-22, tests/hmm/trace.rs
// Matrices are stored column-stochastic as the reference builds them (dmatrix![..].transpose()):
// column j = distribution given state j.
// ---------------------------------------------------------------------------------------
struct HmmParams {
    int n_states = 0, n_obs = 0;
    Vec prior;        // [n_states]
    Vec emission;     // emission(o, s) = emission[o * n_states + s]   (rows = observation)
    Vec transition;   // transition(s2, s1) = transition[s2 * n_states + s1]
    Vec emission_col(int s) const { Vec v; for (int o = 0; o < n_obs; ++o) v.push_back(emission[(size_t)o * n_states + s]); return v; }
    Vec transition_col(int s) const { Vec v; for (int s2 = 0; s2 < n_states; ++s2) v.push_back(transition[(size_t)s2 * n_states + s]); return v; }
};
using HmmData = std::pair<std::vector<std::optional<size_t>>, std::vector<std::optional<size_t>>>;
struct HMM : GenFn<std::pair<int64_t, int>, HmmData, std::vector<size_t>> {
    using A = std::pair<int64_t, int>;
    using TraceT = Trace<A, HmmData, std::vector<size_t>>;
    HmmParams params;
    explicit HMM(HmmParams p) : params(std::move(p)) {}
    // model.rs:33-41 (kernel); the time index of the new state is the Philox step.
    double kernel(Rng& rng, TraceT& trace, const Vec& state_probs, size_t new_observation) const {
        rng.step = (uint32_t)trace.data.first.size();
        rng.at(DOM_MODEL, 0);
        const size_t new_state = (size_t)categorical.random(rng, state_probs);
        const Vec obs_probs = params.emission_col((int)new_state);
        trace.data.first.push_back(new_state);          // extend(): tests/hmm/trace.rs:9-13
        trace.data.second.push_back(new_observation);
        trace.args.first += 1;
        const double weight = categorical.logpdf((int64_t)new_observation, obs_probs);
        trace.logjp += weight;
        return weight;
    }
    TraceT simulate(Rng&, A) const override { throw Panic("not implemented"); }
    std::pair<TraceT, double> generate(Rng& rng, A args, HmmData constraints) const override {
        if (args.first != 1) throw Panic("only expect generate to be called to initialize the state (T = 1)");
        const size_t new_observation = *constraints.second[0];
        TraceT trace{args, HmmData{}, std::vector<size_t>{new_observation}, 0.};
        const double w = kernel(rng, trace, params.prior, new_observation);
        return {std::move(trace), w};
    }
    std::tuple<TraceT, HmmData, double> update(Rng& rng, TraceT trace, A, ArgDiff diff, HmmData constraints) const override {
        if (diff != ArgDiff::Extend) throw Panic("Can't handle GF change type");
        const size_t new_observation = *constraints.second.back();
        const size_t prev_state = *trace.data.first.back();
        const double w = kernel(rng, trace, params.transition_col((int)prev_state), new_observation);
        return {std::move(trace), HmmData{}, w};
    }
};
inline double hmm_forward_alg(const HmmParams& p, const std::vector<size_t>& observations) {  // forward.rs:3-22
    double marginal_likelihood = 1.0;
    Vec alpha = p.prior;
    for (size_t obs : observations) {
        Vec post((size_t)p.n_states);
        double evidence = 0.;
        for (int s = 0; s < p.n_states; ++s) { post[(size_t)s] = alpha[(size_t)s] * p.emission[obs * p.n_states + s]; evidence += post[(size_t)s]; }
        for (int s = 0; s < p.n_states; ++s) post[(size_t)s] /= evidence;
        for (int s2 = 0; s2 < p.n_states; ++s2) {
            double a = 0.;
            for (int s = 0; s < p.n_states; ++s) a += p.transition[(size_t)s2 * p.n_states + s] * post[(size_t)s];
            alpha[(size_t)s2] = a;
        }
        marginal_likelihood *= evidence;
    }
    return marginal_likelihood;
}

}  // namespace oracle
