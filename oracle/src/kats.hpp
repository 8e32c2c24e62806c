// ORACLE — TEST INFRASTRUCTURE ONLY (see rng.hpp).
//
// kats.hpp — the reference's own handler-semantics tests, re-staged against the restated
// handler so that its Update / Regenerate / gc weight rules are pinned by the reference's
// known answers (modppl/tests/dyngenfn.rs).  Each function returns the quantities the
// reference test asserts on; tests/test_oracle_kats.py holds the expected constants
// (-0.5, -2.517551, 0.4, -1.098612 ...) quoted from the reference test file.
#pragma once
#include "models.hpp"

namespace oracle {

// tests/dyngenfn.rs:7-15  DynGenFn_prototype
inline DynGenFn<double, double> kat_prototype() {
    return DynGenFn<double, double>([](DynGenFnHandler<double, double>& g, double noise) -> double {
        double sum = 0.;
        for (int i = 1; i < 3000; ++i) sum += g.template sample_at<double>(normal, NormalParams{1., noise}, std::to_string(i));
        return sum;
    }, [](const std::string& a) -> uint32_t { return (uint32_t)std::stoi(a); });
}
// :32-38
inline DynGenFn<int, int> kat_sample_at_regression() {
    return DynGenFn<int, int>([](DynGenFnHandler<int, int>& g, int) -> int {
        const bool b = g.template sample_at<bool>(bernoulli, 0.25, "b");
        if (b) g.template sample_at<double>(normal, NormalParams{0., 1.}, "x");
        return 0;
    }, [](const std::string& a) -> uint32_t { return a == "b" ? 0u : 1u; });
}
// :40-46
inline DynGenFn<int, int> kat_trace_at_regression(const DynGenFn<double, double>* proto) {
    return DynGenFn<int, int>([proto](DynGenFnHandler<int, int>& g, int) -> int {
        const bool b = g.template sample_at<bool>(bernoulli, 0.25, "b");
        if (b) g.template trace_at<double, double>(*proto, 1.0, "sub");
        return 0;
    }, [](const std::string&) -> uint32_t { return 0u; });
}
// :48-53
inline DynGenFn<int, int> kat_sample_at_regression2() {
    return DynGenFn<int, int>([](DynGenFnHandler<int, int>& g, int) -> int {
        const double m = g.template sample_at<double>(uniform, UniformParams{0., 1.}, "m");
        g.template sample_at<double>(normal, NormalParams{m, 1.}, "x");
        g.template sample_at<double>(normal, NormalParams{m, 1.}, "y");
        return 0;
    }, [](const std::string& a) -> uint32_t { return a == "m" ? 0u : (a == "x" ? 1u : 2u); });
}

struct PoissonKat {  // logpdf only (poisson.rs is off the path; used by tests/dyngenfn.rs:280-300 as a constrained site)
    double logpdf(const int64_t& k, double lambda) const { return (double)k * std::log(lambda) - lambda - std::lgamma((double)k + 1.); }
    int64_t random(Rng&, double) const { throw Panic("PoissonKat: sampler not restated (off path)"); }
};

}  // namespace oracle

extern "C" {

// tests/dyngenfn.rs:55-114: out[0..4] = the five update weights
int32_t oracle_kat_update_weights(uint64_t seed, double* out) {
    GUARD({
        Rng r; r.seed = seed;
        auto f1 = kat_sample_at_regression();
        {   // test_sample_at_update_prev_and_constrained  -> -0.5
            DynTrie c; c.observe("b", arc(true)); c.observe("x", arc(0.0));
            auto tr = f1.generate(r, 0, c).first;
            DynTrie c2; c2.observe("x", arc(1.0));
            out[0] = std::get<2>(f1.update(r, tr, 0, ArgDiff::Unknown, c2));
        }
        {   // test_sample_at_update_no_prev_and_constrained -> -2.517551
            DynTrie c; c.observe("b", arc(false));
            auto tr = f1.generate(r, 0, c).first;
            DynTrie c2; c2.observe("b", arc(true)); c2.observe("x", arc(1.0));
            out[1] = std::get<2>(f1.update(r, tr, 0, ArgDiff::Unknown, c2));
        }
        {   // test_update_sample_at_prev_and_unconstrained -> 0.4
            auto f2 = kat_sample_at_regression2();
            DynTrie c; c.observe("m", arc(1.0)); c.observe("x", arc(1.0)); c.observe("y", arc(-0.3));
            auto tr = f2.generate(r, 0, c).first;
            DynTrie c2; c2.observe("m", arc(0.5));
            out[2] = std::get<2>(f2.update(r, tr, 0, ArgDiff::Unknown, c2));
        }
        {   // test_update_no_prev_and_unconstrained (sample_at) -> -1.098612
            DynTrie c; c.observe("b", arc(false));
            auto tr = f1.generate(r, 0, c).first;
            DynTrie c2; c2.observe("b", arc(true));
            out[3] = std::get<2>(f1.update(r, tr, 0, ArgDiff::Unknown, c2));
        }
        {   // same through trace_at -> -1.098612
            auto proto = kat_prototype();
            auto f3 = kat_trace_at_regression(&proto);
            DynTrie c; c.observe("b", arc(false));
            auto tr = f3.generate(r, 0, c).first;
            DynTrie c2; c2.observe("b", arc(true));
            out[4] = std::get<2>(f3.update(r, tr, 0, ArgDiff::Unknown, c2));
        }
    })
}

// tests/dyngenfn.rs:116-131: residual constraints must panic.  returns 1 if both panic.
int32_t oracle_kat_residual_panics(uint64_t seed) {
    Rng r; r.seed = seed;
    auto proto = kat_prototype();
    int hits = 0;
    try { DynTrie c; c.observe("abc", arc(0.)); proto.generate(r, 0.1, c); } catch (const Panic&) { ++hits; }
    try { DynTrie c; c.observe("abc", arc(0.)); auto tr = proto.simulate(r, 0.1); proto.update(r, tr, 0.1, ArgDiff::NoChange, c); } catch (const Panic&) { ++hits; }
    return hits == 2 ? 1 : 0;
}

// tests/dyngenfn.rs:180-301 test_update.
// out[0]=discard has branch==true, [1]=discard x == old x, [2]=discard u/a == old a, [3]=#leaves in discard,
// [4]=#non-leaves in discard, [5]=new branch==false, [6]=new y, [7]=new v/b, [8]=#leaves new, [9]=#non-leaves new,
// [10]=|expected_new_logjp - logjp|, [11]=|expected_weight - weight|,
// loopy: [12]=discard a, [13]=|logjp err|, [14]=|weight err|,
// hierarchical_update: [15]=discard has value/1, [16]=discard has value/2, [17]=weight, [18]=expected weight
int32_t oracle_kat_update(uint64_t seed, double* out) {
    GUARD({
        Rng r; r.seed = seed;
        auto site = [](const std::string& a) -> uint32_t {
            if (a == "branch") return 0u; if (a == "x" || a == "y") return 1u; return 2u; };
        DynGenFn<int, double> bar([](DynGenFnHandler<int, double>& g, int) { return g.template sample_at<double>(normal, NormalParams{0., 1.}, "a"); }, site);
        DynGenFn<int, double> baz([](DynGenFnHandler<int, double>& g, int) { return g.template sample_at<double>(normal, NormalParams{0., 1.}, "b"); }, site);
        DynGenFn<int, double> foo([&](DynGenFnHandler<int, double>& g, int) -> double {
            if (g.template sample_at<bool>(bernoulli, 0.4, "branch")) {
                g.template sample_at<double>(normal, NormalParams{0., 1.}, "x");
                return g.template trace_at<int, double>(bar, 0, "u");
            } else {
                g.template sample_at<double>(normal, NormalParams{0., 1.}, "y");
                return g.template trace_at<int, double>(baz, 0, "v");
            }
        }, site);
        DynTrie c; c.observe("branch", arc(true));
        auto trace = foo.generate(r, 0, c).first;
        const double x = trace.data.read<double>("x"), a = trace.data.read<double>("u/a");
        const double y = 1.123, b = -2.1;
        DynTrie c2; c2.observe("branch", arc(false)); c2.observe("y", arc(y)); c2.observe("v/b", arc(b));
        auto [nt, discard, weight] = foo.update(r, trace, 0, ArgDiff::NoChange, c2);
        auto count = [](const DynTrie& t, bool leaf) { int n = 0; for (auto& kv : t.mapping) n += (kv.second.is_leaf() == leaf); return n; };
        out[0] = discard.read<bool>("branch") ? 1. : 0.;
        out[1] = discard.read<double>("x") == x ? 1. : 0.;
        out[2] = discard.read<double>("u/a") == a ? 1. : 0.;
        out[3] = count(discard, true); out[4] = count(discard, false);
        out[5] = nt.data.read<bool>("branch") ? 0. : 1.;
        out[6] = nt.data.read<double>("y"); out[7] = nt.data.read<double>("v/b");
        out[8] = count(nt.data, true); out[9] = count(nt.data, false);
        const double prev_logjp = bernoulli.logpdf(true, 0.4) + normal.logpdf(x, {0., 1.}) + normal.logpdf(a, {0., 1.});
        const double exp_new = bernoulli.logpdf(false, 0.4) + normal.logpdf(y, {0., 1.}) + normal.logpdf(b, {0., 1.});
        out[10] = std::fabs(exp_new - nt.logjp);
        out[11] = std::fabs((exp_new - prev_logjp) - weight);

        DynGenFn<int, int> loopy([](DynGenFnHandler<int, int>& g, int) -> int {
            const double a_ = g.template sample_at<double>(normal, NormalParams{0., 1.}, "a");
            for (int i = 0; i < 5; ++i) g.template sample_at<double>(normal, NormalParams{a_, 1.}, "data/" + std::to_string(i));
            return 0;
        }, [](const std::string&) -> uint32_t { return 0u; });
        DynTrie lc; lc.observe("a", arc(0.));
        for (int i = 0; i < 5; ++i) lc.observe("data/" + std::to_string(i), arc(0.));
        auto ltrace = loopy.generate(r, 0, lc).first;
        DynTrie lc2; lc2.observe("a", arc(1.));
        auto [lnt, ldiscard, lweight] = loopy.update(r, ltrace, 0, ArgDiff::NoChange, lc2);
        out[12] = ldiscard.read<double>("a");
        const double lprev = 6. * normal.logpdf(0., {0., 1.});
        const double lnew = normal.logpdf(1., {0., 1.}) + 5. * normal.logpdf(0., {1., 1.});
        out[13] = std::fabs(lnew - lnt.logjp);
        out[14] = std::fabs((lnew - lprev) - lweight);

        PoissonKat poisson;
        DynGenFn<int, int> hier([&](DynGenFnHandler<int, int>& g, int) -> int {
            const int64_t k = g.template sample_at<int64_t>(poisson, 5., "k");
            for (int64_t i = 0; i < k; ++i) g.template sample_at<double>(uniform, UniformParams{0., 1.}, "value/" + std::to_string(i));
            return 0;
        }, [](const std::string& a_) -> uint32_t { return a_ == "k" ? 0u : 1u + (uint32_t)std::stoi(a_.substr(6)); });
        DynTrie hc; hc.observe("k", arc((int64_t)3));
        auto htrace = hier.generate(r, 0, hc).first;
        DynTrie hc2; hc2.observe("k", arc((int64_t)1));
        auto [hnt, hdiscard, hweight] = hier.update(r, htrace, 0, ArgDiff::Unknown, hc2);
        (void)hnt;
        out[15] = hdiscard.search("value/1") ? 1. : 0.;
        out[16] = hdiscard.search("value/2") ? 1. : 0.;
        out[17] = hweight;
        out[18] = poisson.logpdf(1, 5.) - poisson.logpdf(3, 5.) - uniform.logpdf(0.5, {0., 1.}) - uniform.logpdf(0.5, {0., 1.});
    })
}

// tests/dyngenfn.rs:303-388 test_regenerate: 10 rounds; out[3*i+0]=|logjp err|, [3*i+1]=|weight err|, [3*i+2]=structure ok
int32_t oracle_kat_regenerate(uint64_t seed, double* out) {
    GUARD({
        auto site = [](const std::string& a) -> uint32_t {
            if (a == "branch") return 0u; if (a == "x" || a == "y") return 1u; return 2u; };
        DynGenFn<double, double> bar([](DynGenFnHandler<double, double>& g, double mu) { return g.template sample_at<double>(normal, NormalParams{mu, 1.}, "a"); }, site);
        DynGenFn<double, double> baz([](DynGenFnHandler<double, double>& g, double mu) { return g.template sample_at<double>(normal, NormalParams{mu, 1.}, "b"); }, site);
        DynGenFn<double, double> foo([&](DynGenFnHandler<double, double>& g, double mu) -> double {
            if (g.template sample_at<bool>(bernoulli, 0.4, "branch")) {
                g.template sample_at<double>(normal, NormalParams{mu, 1.}, "x");
                return g.template trace_at<double, double>(bar, mu, "u");
            } else {
                g.template sample_at<double>(normal, NormalParams{mu, 1.}, "y");
                return g.template trace_at<double, double>(baz, mu, "v");
            }
        }, site);
        Rng r; r.seed = seed;
        double mu = 0.123;
        DynTrie c; c.observe("branch", arc(true));
        auto trace = foo.generate(r, mu, c).first;
        AddrMap mask; mask.visit("branch");
        for (int i = 0; i < 10; ++i) {
            const bool prev_branch = trace.data.read<bool>("branch");
            const double prev_mu = mu;
            Rng ru; ru.seed = seed; ru.slot = 7; ru.step = (uint32_t)i; ru.at(DOM_DATA, 0);
            mu = ru.u01();
            r.step = (uint32_t)(i + 1);
            auto [nt, weight] = foo.regenerate(r, trace, mu, ArgDiff::Unknown, mask);
            trace = nt;
            const bool br = trace.data.read<bool>("branch");
            const double v1 = br ? trace.data.read<double>("x") : trace.data.read<double>("y");
            const double v2 = br ? trace.data.read<double>("u/a") : trace.data.read<double>("v/b");
            const double exp_logjp = normal.logpdf(v1, {mu, 1.}) + normal.logpdf(v2, {mu, 1.}) + bernoulli.logpdf(br, 0.4);
            out[3 * i + 0] = std::fabs(exp_logjp - trace.logjp);
            double exp_w = 0.;
            if (br == prev_branch)
                exp_w = normal.logpdf(v1, {mu, 1.}) + normal.logpdf(v2, {mu, 1.}) - normal.logpdf(v1, {prev_mu, 1.}) - normal.logpdf(v2, {prev_mu, 1.});
            out[3 * i + 1] = std::fabs(exp_w - weight);
            int leaves = 0, inner = 0;
            for (auto& kv : trace.data.mapping) { leaves += kv.second.is_leaf(); inner += !kv.second.is_leaf(); }
            const bool ok = leaves == 2 && inner == 1 && (br ? (trace.data.search("x") && !trace.data.search("u")->is_leaf())
                                                             : (trace.data.search("y") && !trace.data.search("v")->is_leaf()));
            out[3 * i + 2] = ok ? 1. : 0.;
        }
    })
}

// tests/dyngenfn.rs:163-178 test_simulate: logjp == ln p or ln(1-p); returns |err|
double oracle_kat_simulate(uint64_t seed) {
    Rng r; r.seed = seed;
    DynGenFn<double, bool> foo([](DynGenFnHandler<double, bool>& g, double p) { return g.template sample_at<bool>(bernoulli, p, "x"); },
                               [](const std::string&) -> uint32_t { return 0u; });
    const double p = 0.4;
    auto tr = foo.simulate(r, p);
    const bool x = tr.data.read<bool>("x");
    if (x != *tr.retv) return 1e9;
    return std::fabs(tr.logjp - (x ? std::log(p) : std::log(1. - p)));
}

}  // extern "C"
