// ORACLE — TEST INFRASTRUCTURE ONLY (see rng.hpp).
//
// dyngenfn.hpp — CPU restatement of the Generative Function Interface and of the dynamic
// effect handler every model on the path is interpreted by:
//   modppl/src/gfi.rs:5-111                  Trace, GenFn (simulate/generate/update/regenerate/
//                                            call/propose/assess), ArgDiff
//   modppl/src/modeling/dyngenfn.rs:39-93    DynGenFnHandler {Simulate, Generate, Update, Regenerate}
//   modppl/src/modeling/dyngenfn.rs:100-275  sample_at (four arms)
//   modppl/src/modeling/dyngenfn.rs:283-449  trace_at  (four arms)
//   modppl/src/modeling/dyngenfn.rs:454-486  gc
//   modppl/src/modeling/dyngenfn.rs:503-583  impl GenFn for DynGenFn
//
// The one addition: the reference constructs `ThreadRng::default()` inside each call; here a
// seeded `Rng` (slot, step) is threaded through instead and `site_of(addr)` names the Philox
// site of each address (the static slot id the device kernels use).
#pragma once
#include <functional>
#include <tuple>

#include "trie.hpp"

namespace oracle {

// ---- gfi.rs ---------------------------------------------------------------------------
template <class Args, class Data, class Ret>
struct Trace {
    Args args;
    Data data;
    std::optional<Ret> retv;
    double logjp = 0.;
};

enum class ArgDiff { NoChange, Unknown, Extend };

template <class Args, class Data, class Ret>
struct GenFn {
    using TraceT = Trace<Args, Data, Ret>;
    virtual ~GenFn() {}
    virtual TraceT simulate(Rng& rng, Args args) const = 0;
    virtual std::pair<TraceT, double> generate(Rng& rng, Args args, Data constraints) const = 0;
    virtual std::tuple<TraceT, Data, double> update(Rng& rng, TraceT trace, Args args, ArgDiff diff, Data constraints) const = 0;
    virtual std::pair<TraceT, double> regenerate(Rng&, TraceT, Args, ArgDiff, const AddrMap&) const {
        throw Panic("regenerate: impl not found");  // gfi.rs:66-73
    }
    Ret call(Rng& rng, Args args) const { return *simulate(rng, std::move(args)).retv; }
    std::pair<Data, double> propose(Rng& rng, Args args) const {  // gfi.rs:81-84
        TraceT t = simulate(rng, std::move(args));
        return {std::move(t.data), t.logjp};
    }
    double assess(Rng& rng, Args args, Data constraints) const {  // gfi.rs:87-90
        return generate(rng, std::move(args), std::move(constraints)).second;
    }
};

// ---- dyngenfn.rs ----------------------------------------------------------------------
using SiteFn = std::function<uint32_t(const std::string&)>;

template <class A, class T>
struct DynGenFnHandler {
    enum Mode { Simulate, Generate, Update, Regenerate } mode;
    Rng* prng;
    Trace<A, DynTrie, T> trace;
    // Generate / Update
    double weight = 0.;
    DynTrie constraints;
    // Update / Regenerate
    ArgDiff diff = ArgDiff::NoChange;
    DynTrie discard;
    AddrMap visitor;
    const AddrMap* mask = nullptr;
    AddrMap mask_storage;  // when an empty mask is replaced by the trace schema (dyngenfn.rs:571)
    uint32_t domain = DOM_MODEL;
    SiteFn site_of;

    template <class V>
    static DynValue downcast_check(const DynValue& v, const std::string& addr) {
        if (!std::any_cast<V>(v.get())) throw Panic("error: downcast failed at " + addr);
        return v;
    }
    template <class V, class Dist, class W>
    V draw(const Dist& dist, const W& args, const std::string& addr) {
        prng->at(domain, site_of ? site_of(addr) : 0u);
        return dist.random(*prng, args);
    }

    // dyngenfn.rs:100-275
    template <class V, class Dist, class W>
    V sample_at(const Dist& dist, const W& args, const std::string& addr) {
        switch (mode) {
        case Simulate: {
            V x = draw<V>(dist, args, addr);
            const double logp = dist.logpdf(x, args);
            trace.data.w_observe(addr, arc(x), logp);
            return x;
        }
        case Generate: {
            DynValue x; double logp;
            if (auto choice = constraints.remove(addr)) {
                x = downcast_check<V>(choice->expect_inner("error: no value found in " + addr), addr);
                logp = dist.logpdf(*std::any_cast<V>(x.get()), args);
                weight += logp;
            } else {
                V xv = draw<V>(dist, args, addr);
                logp = dist.logpdf(xv, args);
                x = arc(xv);
            }
            trace.data.w_observe(addr, x, logp);
            return *std::any_cast<V>(x.get());
        }
        case Update: {
            visitor.visit(addr);
            DynValue x; double logp;
            if (auto choice = constraints.remove(addr)) {
                if (auto call = trace.data.remove(addr)) {
                    weight -= call->weight();
                    discard.insert(addr, std::move(*call));
                }
                x = downcast_check<V>(choice->expect_inner("error: no value found in " + addr), addr);
                logp = dist.logpdf(*std::any_cast<V>(x.get()), args);
                diff = ArgDiff::Unknown;
                weight += logp;
            } else if (auto call = trace.data.remove(addr)) {
                if (diff == ArgDiff::NoChange) {
                    DynValue xv = downcast_check<V>(call->expect_inner("error: no value found in " + addr), addr);
                    trace.data.insert(addr, std::move(*call));
                    return *std::any_cast<V>(xv.get());
                } else if (diff == ArgDiff::Unknown) {
                    const double prev_logp = call->weight();
                    x = downcast_check<V>(call->expect_inner("error: no value found in " + addr), addr);
                    logp = dist.logpdf(*std::any_cast<V>(x.get()), args);
                    weight += logp - prev_logp;
                } else {
                    throw Panic("update: ArgDiff::Extend not supported");
                }
            } else {
                V xv = draw<V>(dist, args, addr);
                x = arc(xv);
                logp = dist.logpdf(xv, args);
                diff = ArgDiff::Unknown;
            }
            trace.data.w_observe(addr, x, logp);
            return *std::any_cast<V>(x.get());
        }
        case Regenerate: {
            visitor.visit(addr);
            DynValue x; double logp;
            if (mask->search(addr)) {
                trace.data.remove(addr);  // remove (if has previous)
                V xv = draw<V>(dist, args, addr);
                x = arc(xv);
                logp = dist.logpdf(xv, args);
                diff = ArgDiff::Unknown;
            } else if (auto call = trace.data.remove(addr)) {
                if (diff == ArgDiff::NoChange) {
                    DynValue xv = downcast_check<V>(call->expect_inner("error: no value found in " + addr), addr);
                    trace.data.insert(addr, std::move(*call));
                    return *std::any_cast<V>(xv.get());
                } else if (diff == ArgDiff::Unknown) {
                    const double prev_logp = call->weight();
                    x = downcast_check<V>(call->expect_inner("error: no value found in " + addr), addr);
                    logp = dist.logpdf(*std::any_cast<V>(x.get()), args);
                    weight += logp - prev_logp;
                } else {
                    throw Panic("regenerate: ArgDiff::Extend not supported");
                }
            } else {
                V xv = draw<V>(dist, args, addr);
                x = arc(xv);
                logp = dist.logpdf(xv, args);
                diff = ArgDiff::Unknown;
            }
            trace.data.w_observe(addr, x, logp);
            return *std::any_cast<V>(x.get());
        }
        }
        throw Panic("unreachable");
    }

    // dyngenfn.rs:283-449
    template <class X, class Y>
    Y trace_at(const GenFn<X, DynTrie, Y>& gen_fn, const X& args, const std::string& addr) {
        switch (mode) {
        case Simulate: {
            auto sub = gen_fn.simulate(*prng, args);
            sub.data.replace_inner(arc(*sub.retv));
            trace.data.insert(addr, std::move(sub.data));
            return *sub.retv;
        }
        case Generate: {
            DynTrie sub; std::optional<Y> retv;
            if (auto choices = constraints.remove(addr)) {
                auto [st, dw] = gen_fn.generate(*prng, args, std::move(*choices));
                weight += dw;
                sub = std::move(st.data); retv = st.retv;
            } else {
                auto st = gen_fn.simulate(*prng, args);
                sub = std::move(st.data); retv = st.retv;
            }
            sub.replace_inner(arc(*retv));
            trace.data.insert(addr, std::move(sub));
            return *retv;
        }
        case Update: {
            visitor.visit(addr);
            DynTrie sub; std::optional<Y> retv;
            if (auto choices = constraints.remove(addr)) {
                if (auto prev = trace.data.remove(addr)) {
                    const double logjp = prev->weight();
                    Trace<X, DynTrie, Y> st{args, std::move(*prev), std::nullopt, logjp};
                    auto [nt, subdiscard, dw] = gen_fn.update(*prng, std::move(st), args, diff, std::move(*choices));
                    if (!subdiscard.is_empty()) discard.insert(addr, std::move(subdiscard));
                    diff = ArgDiff::Unknown;
                    weight += dw;
                    sub = std::move(nt.data); retv = nt.retv;
                } else {
                    auto [nt, dw] = gen_fn.generate(*prng, args, std::move(*choices));
                    diff = ArgDiff::Unknown;
                    weight += dw;
                    sub = std::move(nt.data); retv = nt.retv;
                }
            } else if (auto prev = trace.data.remove(addr)) {
                if (diff == ArgDiff::NoChange) {
                    const Y* r = std::any_cast<Y>(prev->expect_inner("no retv at " + addr).get());
                    if (!r) throw Panic("downcast failed at " + addr);
                    Y rv = *r;
                    trace.data.insert(addr, std::move(*prev));
                    return rv;
                } else if (diff == ArgDiff::Unknown) {
                    const double logjp = prev->weight();
                    Trace<X, DynTrie, Y> st{args, std::move(*prev), std::nullopt, logjp};
                    auto [nt, subdiscard, dw] = gen_fn.update(*prng, std::move(st), args, ArgDiff::Unknown, DynTrie());
                    if (!subdiscard.is_empty()) discard.insert(addr, std::move(subdiscard));
                    weight += dw;
                    sub = std::move(nt.data); retv = nt.retv;
                } else {
                    throw Panic("update: ArgDiff::Extend not supported");
                }
            } else {
                auto st = gen_fn.simulate(*prng, args);
                diff = ArgDiff::Unknown;
                sub = std::move(st.data); retv = st.retv;
            }
            sub.replace_inner(arc(*retv));
            trace.data.insert(addr, std::move(sub));
            return *retv;
        }
        case Regenerate: {
            visitor.visit(addr);
            const AddrMap* submask = mask->search(addr);
            DynTrie sub; std::optional<Y> retv;
            if (auto prev = trace.data.remove(addr)) {
                const double logjp = prev->weight();
                if (submask) {
                    Trace<X, DynTrie, Y> st{args, std::move(*prev), std::nullopt, logjp};
                    auto [nt, dw] = gen_fn.regenerate(*prng, std::move(st), args, diff, *submask);
                    diff = ArgDiff::Unknown;
                    weight += dw;
                    sub = std::move(nt.data); retv = nt.retv;
                } else if (diff == ArgDiff::NoChange) {
                    const Y* r = std::any_cast<Y>(prev->expect_inner("no retv at " + addr).get());
                    if (!r) throw Panic("downcast failed at " + addr);
                    Y rv = *r;
                    trace.data.insert(addr, std::move(*prev));
                    return rv;
                } else if (diff == ArgDiff::Unknown) {
                    const double prev_weight = prev->weight();
                    auto [nt, new_weight] = gen_fn.generate(*prng, args, std::move(*prev));
                    weight += new_weight - prev_weight;
                    sub = std::move(nt.data); retv = nt.retv;
                } else {
                    throw Panic("regenerate: ArgDiff::Extend not supported");
                }
            } else {
                auto st = gen_fn.simulate(*prng, args);
                diff = ArgDiff::Unknown;
                sub = std::move(st.data); retv = st.retv;
            }
            sub.replace_inner(arc(*retv));
            trace.data.insert(addr, std::move(sub));
            return *retv;
        }
        }
        throw Panic("unreachable");
    }

    // dyngenfn.rs:454-486
    void gc() {
        if (mode == Update) {
            const AddrMap schema = trace.data.schema();
            DynTrie data, complement; double cw;
            Trie::collect(std::move(trace.data), schema.complement(visitor), data, complement, cw);
            discard.merge(std::move(complement));
            trace.data = std::move(data);
            trace.logjp = 0.;
            weight = weight - cw;
        } else if (mode == Regenerate) {
            const AddrMap schema = trace.data.schema();
            DynTrie data, complement; double cw;
            Trie::collect(std::move(trace.data), schema.complement(visitor), data, complement, cw);
            trace.data = std::move(data);
            trace.logjp = 0.;
        } else {
            throw Panic("garbage-collect (gc): called outside of update or regenerate context");
        }
    }
};

// dyngenfn.rs:489-583
template <class Args, class Ret>
struct DynGenFn : GenFn<Args, DynTrie, Ret> {
    using H = DynGenFnHandler<Args, Ret>;
    using TraceT = Trace<Args, DynTrie, Ret>;
    std::function<Ret(H&, Args)> func;
    SiteFn site_of;
    uint32_t domain = DOM_MODEL;
    DynGenFn() {}
    DynGenFn(std::function<Ret(H&, Args)> f, SiteFn s = nullptr, uint32_t dom = DOM_MODEL)
        : func(std::move(f)), site_of(std::move(s)), domain(dom) {}

    H handler(typename H::Mode m, Rng& rng) const {
        H g; g.mode = m; g.prng = &rng; g.site_of = site_of; g.domain = domain;
        return g;
    }
    TraceT simulate(Rng& rng, Args args) const override {
        H g = handler(H::Simulate, rng);
        g.trace = TraceT{args, DynTrie(), std::nullopt, 0.};
        Ret retv = func(g, args);
        g.trace.retv = retv;
        g.trace.logjp = g.trace.data.weight();
        return std::move(g.trace);
    }
    std::pair<TraceT, double> generate(Rng& rng, Args args, DynTrie constraints) const override {
        constraints.take_inner();  // in case constraints came from a proposal
        H g = handler(H::Generate, rng);
        g.trace = TraceT{args, DynTrie(), std::nullopt, 0.};
        g.constraints = std::move(constraints);
        Ret retv = func(g, args);
        if (!g.constraints.is_empty()) throw Panic("generate error: not all constraints were consumed!");
        g.trace.logjp = g.trace.data.weight();
        g.trace.retv = retv;
        return {std::move(g.trace), g.weight};
    }
    std::tuple<TraceT, DynTrie, double> update(Rng& rng, TraceT trace, Args args, ArgDiff diff, DynTrie constraints) const override {
        constraints.take_inner();
        H g = handler(H::Update, rng);
        g.trace = std::move(trace);
        g.diff = diff;
        g.constraints = std::move(constraints);
        Ret retv = func(g, args);
        g.gc();
        if (!g.constraints.is_empty()) throw Panic("update error: not all constraints were consumed!");
        g.trace.logjp = g.trace.data.weight();
        g.trace.retv = retv;
        return {std::move(g.trace), std::move(g.discard), g.weight};
    }
    std::pair<TraceT, double> regenerate(Rng& rng, TraceT trace, Args args, ArgDiff diff, const AddrMap& mask) const override {
        H g = handler(H::Regenerate, rng);
        if (mask.is_leaf()) { g.mask_storage = trace.data.schema(); g.mask = &g.mask_storage; }
        else g.mask = &mask;
        g.trace = std::move(trace);
        g.diff = diff;
        Ret retv = func(g, args);
        g.gc();
        g.trace.logjp = g.trace.data.weight();
        g.trace.retv = retv;
        return {std::move(g.trace), g.weight};
    }
};

}  // namespace oracle
