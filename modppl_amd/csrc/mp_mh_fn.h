// mp_mh_fn.h — mh / regen_mh for REGISTERED generative functions (mp_mh_models.h): the kernels below are the four GFI
// interpretations of mp_genfn.h applied to whatever functor a registration names, so a new model or proposal needs no kernel.
// Included by mp_mh.hip (after struct mp_mh); not a public header.
//
//   K7g k_fn_mh<Model, Proposal>   metropolis_hastings (modppl/src/inference/mh.rs:9-40):
//         (fwd_choices, fwd) = proposal.propose(trace)          -> SIMULATE, Philox domain MP_DOM_PROPOSAL
//         (trace', discard, w) = model.update(trace, NoChange, fwd_choices) -> UPDATE (+ gc)
//         bwd = proposal.assess(trace', discard)                -> GENERATE with the discard as constraints
//         accept iff ln u < w - fwd + bwd                       -> MP_DOM_ACCEPT site 0
//   K7h k_fn_regen<Model>          regenerative_metropolis_hastings (mh.rs:54-67): REGENERATE, accept iff ln u < w; an empty
//         mask is the trace's whole schema (dyngenfn.rs:571)
// One lane = one chain; the trace lives in registers for all n_iters iterations of a launch and in a site-major table
// vals[site][chain] + present[chain] between launches (log-densities are recomputed on load: GENERATE with every choice
// constrained).  Philox: chain = slot, iteration (1-based) = step, site id = site — the addressing of the hand-written kernels.
#pragma once
#include <functional>
#include <map>

#include "mp_genfn.h"

struct mp_fn_maskspec {
    uint64_t bits;      // cycle == 0: the masked sites (0 = empty mask = whole schema)
    int n_cycle;        // > 0: iteration k masks only cycle[k % n_cycle]
    unsigned char cycle[MP_FN_MAX_SITES];
};
struct mp_fn_consspec {
    uint64_t bits;
    double val[MP_FN_MAX_SITES];
};

// site k is where a sub-call of the model keeps its sub-trie's running weight (the lowest site of the sub-call): rows NS + k of
// the chain table hold it between launches — unlike the log-densities it cannot be recomputed from the values (it is the
// history of the trie's inserts and removes, mp_genfn.h)
template <class M>
__host__ __device__ constexpr bool fn_is_sub_id(int k) {
    return M::sub_of(k) != 0u && ((uint64_t)M::sub_of(k) & ((uint64_t(1) << k) - 1u)) == 0u;
}
// Presence words between launches and across the C ABI: 32-bit words, one per chain up to 32 sites, two beyond — word w of chain i at
// present[w * n + i] on the device ([chain][word] on the host side of the ABI: the same thing for one word)
__host__ __device__ constexpr int fn_words(int ns) { return (ns + 31) / 32; }
template <int NS>
__device__ __forceinline__ mp_fn_bits_t<NS> fn_bits_load(const uint32_t* __restrict__ present, u64 i, u64 n) {
    mp_fn_bits_t<NS> b = present[i];
    if constexpr (NS > 32) b |= (mp_fn_bits_t<NS>)present[n + i] << 32;
    return b;
}
template <int NS>
__device__ __forceinline__ void fn_bits_store(uint32_t* __restrict__ present, u64 i, u64 n, mp_fn_bits_t<NS> b) {
    present[i] = (uint32_t)b;
    if constexpr (NS > 32) present[n + i] = (uint32_t)((uint64_t)b >> 32);
}
template <class M>
__device__ __forceinline__ void fn_load(const M& model, const mp_stream& s, u64 i, u64 n, const double* __restrict__ vals,
                                        const uint32_t* __restrict__ present, mp_fn_trace<M::NS>& out, double* data_lp = nullptr) {
    mp_fn_trace<M::NS> c;
    c.present = fn_bits_load<M::NS>(present, i, n);
#pragma unroll
    for (int k = 0; k < M::NS; ++k) {
        c.val[k] = ((c.present >> k) & 1u) ? vals[(u64)k * n + i] : 0.;
        c.lp[k] = 0.;
        c.subw[k] = 0.;
    }
    mp_fn_handler<M::NS, MP_FN_GENERATE, M> g(s, MP_DOM_MODEL, nullptr, &c);
    model(g);
    out = g.tr;
    if (data_lp) *data_lp = g.dlp;   // (declared data sites: their share of trace.logjp)
#pragma unroll
    for (int k = 0; k < M::NS; ++k)
        if (fn_is_sub_id<M>(k)) out.subw[k] = vals[(u64)(M::NS + k) * n + i];
}
template <class M>
__device__ __forceinline__ void fn_store(const mp_fn_trace<M::NS>& t, u64 i, u64 n, double* __restrict__ vals, uint32_t* __restrict__ present) {
    fn_bits_store<M::NS>(present, i, n, t.present);
#pragma unroll
    for (int k = 0; k < M::NS; ++k) {
        vals[(u64)k * n + i] = t.has(k) ? t.val[k] : 0.;
        if (fn_is_sub_id<M>(k)) vals[(u64)(M::NS + k) * n + i] = t.subw[k];
    }
}
__device__ __forceinline__ void fn_count(u64 acc, bool panic, u64* __restrict__ totals) {
    u64 p = panic ? 1ull : 0ull;
    for (int o = 32; o > 0; o >>= 1) { acc += __shfl_xor(acc, o, 64); p += __shfl_xor(p, o, 64); }
    if ((threadIdx.x & 63) == 0) {
        if (acc) atomicAdd(totals, acc);
        if (p) atomicAdd(totals + 1, p);
    }
}

template <class M>
__global__ __launch_bounds__(MH_THREADS) void k_fn_logjp(u64 n, M model, const double* __restrict__ vals, const uint32_t* __restrict__ present,
                                                         double* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * MH_THREADS + threadIdx.x;
    if (i >= n) return;
    mp_stream s;
    s.k0 = 0; s.k1 = 0; s.slot = 0; s.step = 0;   // nothing is drawn: every choice is constrained
    mp_fn_trace<M::NS> cur;
    double dlp = 0.;
    fn_load(model, s, i, n, vals, present, cur, &dlp);
    out[i] = mp_fn_has_data<M>::value ? mp_fn_logjp(cur) + dlp : mp_fn_logjp(cur);
}

template <class M>
__global__ __launch_bounds__(MH_THREADS) void k_fn_regen(u64 n, uint32_t k0, uint32_t k1, uint32_t iter0, int n_iters, M model, mp_fn_maskspec mask,
                                                         double* __restrict__ vals, uint32_t* __restrict__ present, u64* __restrict__ totals) {
    const u64 i = (u64)blockIdx.x * MH_THREADS + threadIdx.x;
    u64 acc = 0;
    bool panic = false;
    if (i < n) {
        mp_stream s;
        s.k0 = k0; s.k1 = k1; s.slot = (uint32_t)i; s.step = 0;
        mp_fn_trace<M::NS> cur;
        fn_load(model, s, i, n, vals, present, cur);
        for (int it = 0; it < n_iters; ++it) {
            s.step = iter0 + (uint32_t)it;
            mp_fn_bits_t<M::NS> m = (mp_fn_bits_t<M::NS>)mask.bits;
            if (mask.n_cycle > 0) m = mp_fn_bits_t<M::NS>(1) << mask.cycle[(iter0 - 1u + (uint32_t)it) % (uint32_t)mask.n_cycle];
            if (m == 0u) m = cur.present;   // mask.is_leaf(): the whole schema (dyngenfn.rs:571)
            mp_fn_handler<M::NS, MP_FN_REGENERATE, M> g(s, MP_DOM_MODEL, &cur, nullptr, m);
            model(g);
            g.finish();
            panic |= g.panic;
            const mp_u64x2 ub = s.draw(MP_DOM_ACCEPT, 0u, 0u);
            if (mp_log(mp_u01(ub.a)) < g.weight) {   // mh.rs:62
                cur = g.tr;
                ++acc;
            }
        }
        fn_store<M>(cur, i, n, vals, present);
    }
    fn_count(acc, panic, totals);
}

template <class M, class P>
__global__ __launch_bounds__(MH_THREADS) void k_fn_mh(u64 n, uint32_t k0, uint32_t k1, uint32_t iter0, int n_iters, M model, P proposal,
                                                      double* __restrict__ vals, uint32_t* __restrict__ present, u64* __restrict__ totals) {
    const u64 i = (u64)blockIdx.x * MH_THREADS + threadIdx.x;
    u64 acc = 0;
    bool panic = false;
    if (i < n) {
        mp_stream s;
        s.k0 = k0; s.k1 = k1; s.slot = (uint32_t)i; s.step = 0;
        mp_fn_trace<M::NS> cur;
        fn_load(model, s, i, n, vals, present, cur);
        for (int it = 0; it < n_iters; ++it) {
            s.step = iter0 + (uint32_t)it;
            mp_fn_handler<M::NS, MP_FN_SIMULATE, M> p(s, MP_DOM_PROPOSAL, nullptr, nullptr);
            proposal(p, cur);
            const double fwd = p.weight;
            mp_fn_handler<M::NS, MP_FN_UPDATE, M> g(s, MP_DOM_MODEL, &cur, &p.tr);
            model(g);
            g.finish();
            // the discard: the previous values of what update replaced or collected
            mp_fn_trace<M::NS> disc = cur;
            disc.present = g.discarded;
            mp_fn_handler<M::NS, MP_FN_GENERATE, M> q(s, MP_DOM_PROPOSAL, nullptr, &disc);
            proposal(q, g.tr);
            q.finish();
            panic |= g.panic | q.panic;
            const double alpha = g.weight - fwd + q.weight;   // mh.rs:34
            const mp_u64x2 ub = s.draw(MP_DOM_ACCEPT, 0u, 0u);
            if (mp_log(mp_u01(ub.a)) < alpha) {
                cur = g.tr;
                ++acc;
            }
        }
        fn_store<M>(cur, i, n, vals, present);
    }
    fn_count(acc, panic, totals);
}

// ---------------------------------------------------------------------------------------
// The GFI operations ONE AT A TIME (gfi.rs:57-90: update / regenerate / assess / propose), for callers that compose their own
// inference moves: what k_fn_mh / k_fn_regen fuse, exposed per call (include/modppl_hip.h mp_fn_*).  Every chain, one launch.
// Constraints: shared by every chain (cs.bits, cs.val) or per chain (a table cvals[site][chain] + cpresent[chain]).  `step` is the
// Philox step of whatever the call draws.
// ---------------------------------------------------------------------------------------
template <int NS>
__device__ __forceinline__ void fn_cons(const mp_fn_consspec& cs, const double* __restrict__ cvals, const uint32_t* __restrict__ cpresent, u64 i, u64 n,
                                        mp_fn_trace<NS>& c) {
    // shared: the sites cs.bits with the values cs.val; per chain: the table cvals[site][chain] + cpresent[chain] (what fn_emit writes:
    // the choices of a propose, the discard of an update — a proposal's choices differ in their SITES from chain to chain)
    c.present = cpresent ? fn_bits_load<NS>(cpresent, i, n) : (mp_fn_bits_t<NS>)cs.bits;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const bool on = (c.present >> k) & 1u;
        c.val[k] = on ? (cvals ? cvals[(u64)k * n + i] : cs.val[k]) : 0.;
        c.lp[k] = 0.;
        c.subw[k] = 0.;
    }
}
template <int NS>
__device__ __forceinline__ void fn_emit(const mp_fn_trace<NS>& t, mp_fn_bits_t<NS> bits, u64 i, u64 n, double* __restrict__ vals, uint32_t* __restrict__ present) {
    fn_bits_store<NS>(present, i, n, bits);
#pragma unroll
    for (int k = 0; k < NS; ++k) vals[(u64)k * n + i] = ((bits >> k) & 1u) ? t.val[k] : 0.;
}
// (new_trace, discard, weight) = model.update(trace, args, diff, constraints)   gfi.rs:57-64, dyngenfn.rs:536-560; the trace is replaced
template <class M>
__global__ __launch_bounds__(MH_THREADS) void k_fn_update(u64 n, uint32_t k0, uint32_t k1, uint32_t step, M model, mp_fn_consspec cs,
                                                          const double* __restrict__ cvals, const uint32_t* __restrict__ cpresent, int unknown, double* __restrict__ vals,
                                                          uint32_t* __restrict__ present, double* __restrict__ w_out, double* __restrict__ dvals,
                                                          uint32_t* __restrict__ dpresent, u64* __restrict__ totals) {
    const u64 i = (u64)blockIdx.x * MH_THREADS + threadIdx.x;
    bool panic = false;
    if (i < n) {
        mp_stream s;
        s.k0 = k0; s.k1 = k1; s.slot = (uint32_t)i; s.step = step;
        mp_fn_trace<M::NS> cur, c;
        fn_load(model, s, i, n, vals, present, cur);
        fn_cons<M::NS>(cs, cvals, cpresent, i, n, c);
        mp_fn_handler<M::NS, MP_FN_UPDATE, M> g(s, MP_DOM_MODEL, &cur, &c);
        g.changed = unknown != 0;   // ArgDiff::Unknown: every revisited choice is re-scored (dyngenfn.rs:180-190)
        model(g);
        g.finish();
        panic = g.panic;
        if (!panic) fn_store<M>(g.tr, i, n, vals, present);   // (a chain that reached the reference's panic keeps the trace it had: ADVICE round 4)
        w_out[i] = g.weight;
        if (dvals) fn_emit<M::NS>(cur, g.discarded, i, n, dvals, dpresent);   // the discard: the previous values of what was replaced or collected
    }
    fn_count(0, panic, totals);
}
// (new_trace, weight) = model.regenerate(trace, args, diff, mask)   gfi.rs:66-73, dyngenfn.rs:562-583; the trace is replaced
template <class M>
__global__ __launch_bounds__(MH_THREADS) void k_fn_regenerate(u64 n, uint32_t k0, uint32_t k1, uint32_t step, M model, uint64_t mask, int unknown,
                                                              double* __restrict__ vals, uint32_t* __restrict__ present, double* __restrict__ w_out,
                                                              u64* __restrict__ totals) {
    const u64 i = (u64)blockIdx.x * MH_THREADS + threadIdx.x;
    bool panic = false;
    if (i < n) {
        mp_stream s;
        s.k0 = k0; s.k1 = k1; s.slot = (uint32_t)i; s.step = step;
        mp_fn_trace<M::NS> cur;
        fn_load(model, s, i, n, vals, present, cur);
        const mp_fn_bits_t<M::NS> m = mask ? (mp_fn_bits_t<M::NS>)mask : cur.present;   // mask.is_leaf(): the whole schema (dyngenfn.rs:571)
        mp_fn_handler<M::NS, MP_FN_REGENERATE, M> g(s, MP_DOM_MODEL, &cur, nullptr, m);
        g.changed = unknown != 0;
        model(g);
        g.finish();
        panic = g.panic;
        if (!panic) fn_store<M>(g.tr, i, n, vals, present);
        w_out[i] = g.weight;
    }
    fn_count(0, panic, totals);
}
// weight = model.assess(args, constraints) = generate(args, constraints).1   gfi.rs:85-90; the chains' traces are not touched
template <class M>
__global__ __launch_bounds__(MH_THREADS) void k_fn_assess(u64 n, uint32_t k0, uint32_t k1, uint32_t step, M model, mp_fn_consspec cs,
                                                          const double* __restrict__ cvals, const uint32_t* __restrict__ cpresent, double* __restrict__ w_out,
                                                          u64* __restrict__ totals) {
    const u64 i = (u64)blockIdx.x * MH_THREADS + threadIdx.x;
    bool panic = false;
    if (i < n) {
        mp_stream s;
        s.k0 = k0; s.k1 = k1; s.slot = (uint32_t)i; s.step = step;
        mp_fn_trace<M::NS> c;
        fn_cons<M::NS>(cs, cvals, cpresent, i, n, c);
        mp_fn_handler<M::NS, MP_FN_GENERATE, M> g(s, MP_DOM_MODEL, nullptr, &c);
        model(g);
        g.finish();
        panic = g.panic;
        w_out[i] = g.weight;
    }
    fn_count(0, panic, totals);
}
// (trace, weight) = model.generate(args, constraints)   gfi.rs:53-55, dyngenfn.rs:513-521: every chain's trace is REPLACED by the new one
// (constrained sites take their constraint and score into the weight, the others are drawn from their priors: the internal proposal of
// importance_sampling, importance.rs:18-20)
template <class M>
__global__ __launch_bounds__(MH_THREADS) void k_fn_generate(u64 n, uint32_t k0, uint32_t k1, uint32_t step, M model, mp_fn_consspec cs,
                                                            const double* __restrict__ cvals, const uint32_t* __restrict__ cpresent, double* __restrict__ vals,
                                                            uint32_t* __restrict__ present, double* __restrict__ w_out, u64* __restrict__ totals) {
    const u64 i = (u64)blockIdx.x * MH_THREADS + threadIdx.x;
    bool panic = false;
    if (i < n) {
        mp_stream s;
        s.k0 = k0; s.k1 = k1; s.slot = (uint32_t)i; s.step = step;
        mp_fn_trace<M::NS> c;
        fn_cons<M::NS>(cs, cvals, cpresent, i, n, c);
        mp_fn_handler<M::NS, MP_FN_GENERATE, M> g(s, MP_DOM_MODEL, nullptr, &c);
        model(g);
        g.finish();
        panic = g.panic;
        if (!panic) fn_store<M>(g.tr, i, n, vals, present);   // (a chain that reached the reference's panic keeps the trace it had)
        w_out[i] = g.weight;
    }
    fn_count(0, panic, totals);
}
// trace = model.simulate(args)   gfi.rs:51, dyngenfn.rs:503-511: every site drawn, the observed ones too; out: trace.logjp
template <class M>
__global__ __launch_bounds__(MH_THREADS) void k_fn_simulate(u64 n, uint32_t k0, uint32_t k1, uint32_t step, M model, double* __restrict__ vals,
                                                            uint32_t* __restrict__ present, double* __restrict__ w_out) {
    const u64 i = (u64)blockIdx.x * MH_THREADS + threadIdx.x;
    if (i >= n) return;
    mp_stream s;
    s.k0 = k0; s.k1 = k1; s.slot = (uint32_t)i; s.step = step;
    mp_fn_handler<M::NS, MP_FN_SIMULATE, M> g(s, MP_DOM_MODEL, nullptr, nullptr);
    model(g);
    fn_store<M>(g.tr, i, n, vals, present);
    w_out[i] = g.weight;
}
// (choices, weight) = proposal.propose((trace, args)) = simulate -> (data, logjp)   gfi.rs:78-83, mh.rs:17-19
template <class M, class P>
__global__ __launch_bounds__(MH_THREADS) void k_fn_propose(u64 n, uint32_t k0, uint32_t k1, uint32_t step, M model, P proposal,
                                                           const double* __restrict__ vals, const uint32_t* __restrict__ present,
                                                           double* __restrict__ w_out, double* __restrict__ cvals_out, uint32_t* __restrict__ cpresent_out) {
    const u64 i = (u64)blockIdx.x * MH_THREADS + threadIdx.x;
    if (i >= n) return;
    mp_stream s;
    s.k0 = k0; s.k1 = k1; s.slot = (uint32_t)i; s.step = step;
    mp_fn_trace<M::NS> cur;
    fn_load(model, s, i, n, vals, present, cur);
    mp_fn_handler<M::NS, MP_FN_SIMULATE, M> p(s, MP_DOM_PROPOSAL, nullptr, nullptr);
    proposal(p, cur);
    w_out[i] = p.weight;
    fn_emit<M::NS>(p.tr, p.tr.present, i, n, cvals_out, cpresent_out);
}
// weight = proposal.assess((trace, args), constraints)   mh.rs:25-27 (the backward weight: constraints = the discard of an update)
template <class M, class P>
__global__ __launch_bounds__(MH_THREADS) void k_fn_assess_proposal(u64 n, uint32_t k0, uint32_t k1, uint32_t step, M model, P proposal, mp_fn_consspec cs,
                                                                   const double* __restrict__ cvals, const uint32_t* __restrict__ cpresent,
                                                                   const double* __restrict__ vals, const uint32_t* __restrict__ present,
                                                                   double* __restrict__ w_out, u64* __restrict__ totals) {
    const u64 i = (u64)blockIdx.x * MH_THREADS + threadIdx.x;
    bool panic = false;
    if (i < n) {
        mp_stream s;
        s.k0 = k0; s.k1 = k1; s.slot = (uint32_t)i; s.step = step;
        mp_fn_trace<M::NS> cur, c;
        fn_load(model, s, i, n, vals, present, cur);
        fn_cons<M::NS>(cs, cvals, cpresent, i, n, c);
        mp_fn_handler<M::NS, MP_FN_GENERATE, M> q(s, MP_DOM_PROPOSAL, nullptr, &c);
        proposal(q, cur);
        q.finish();
        panic = q.panic;
        w_out[i] = q.weight;
    }
    fn_count(0, panic, totals);
}

// ---------------------------------------------------------------------------------------
// registry: model kind -> factory; (model type, proposal kind) -> launcher
// ---------------------------------------------------------------------------------------
struct mh_fn_ops {
    virtual ~mh_fn_ops() {}
    virtual int ns() const = 0;
    virtual int32_t regen(mp_mh* h, const mp_fn_maskspec& m, int n_iters) = 0;
    virtual int32_t mh(mp_mh* h, int proposal_kind, const double* args, int n_args, int n_iters) = 0;
    virtual int32_t logjp(mp_mh* h) = 0;
    // declared data sites (mp_genfn.h): how many, and where the model finds the shared arrays {covariates, observed values}
    virtual int n_data() const { return 0; }
    virtual void bind_data(const double* /*d_cov*/, const double* /*d_obs*/) {}
    // the GFI operations one at a time (mp_fn_*): weights into h->tmp, discard / choices into h->gfi_vals / h->gfi_present
    virtual int32_t update(mp_mh* h, const mp_fn_consspec& cs, const double* d_cvals, const uint32_t* d_cpresent, int unknown, uint32_t step, bool want_discard) = 0;
    virtual int32_t regenerate(mp_mh* h, uint64_t mask, int unknown, uint32_t step) = 0;
    virtual int32_t assess(mp_mh* h, const mp_fn_consspec& cs, const double* d_cvals, const uint32_t* d_cpresent, uint32_t step) = 0;
    virtual int32_t generate(mp_mh* h, const mp_fn_consspec& cs, const double* d_cvals, const uint32_t* d_cpresent, uint32_t step) = 0;
    virtual int32_t simulate(mp_mh* h, uint32_t step) = 0;
    virtual int32_t propose(mp_mh* h, int proposal_kind, const double* args, int n_args, uint32_t step) = 0;
    virtual int32_t assess_proposal(mp_mh* h, int proposal_kind, const double* args, int n_args, const mp_fn_consspec& cs, const double* d_cvals,
                                    const uint32_t* d_cpresent, uint32_t step) = 0;
};
static inline unsigned fn_grid(const mp_mh* h) { return (unsigned)((h->n + MH_THREADS - 1) / MH_THREADS); }

template <class M>
struct mh_fn_launcher {   // what a registered proposal can be launched as
    std::function<int32_t(mp_mh*, const M&, const double*, int, int)> mh;   // (args, n_args, n_iters)
    std::function<int32_t(mp_mh*, const M&, const double*, int, uint32_t)> propose;   // (args, n_args, step)
    std::function<int32_t(mp_mh*, const M&, const double*, int, const mp_fn_consspec&, const double*, const uint32_t*, uint32_t)> assess;
};
template <class M>
static std::map<int, mh_fn_launcher<M>>& mh_fn_proposals() {
    static std::map<int, mh_fn_launcher<M>> r;
    return r;
}
template <class M>
struct mh_fn_ops_t : mh_fn_ops {
    M model;
    int ns() const override { return M::NS; }
    int n_data() const override {
        if constexpr (mp_fn_has_data<M>::value) return model.n_obs;
        else return 0;
    }
    void bind_data(const double* d_cov, const double* d_obs) override {
        if constexpr (mp_fn_has_data<M>::value) model.bind(d_cov, d_obs);
    }
    int32_t regen(mp_mh* h, const mp_fn_maskspec& m, int n_iters) override {
        hipLaunchKernelGGL(k_fn_regen<M>, dim3(fn_grid(h)), dim3(MH_THREADS), 0, h->stream, h->n, (uint32_t)h->seed, (uint32_t)(h->seed >> 32),
                           (uint32_t)(h->iters + 1), n_iters, model, m, h->fvals, h->fpresent, h->d_acc);
        MHCK(hipGetLastError());
        return MP_OK;
    }
    int32_t mh(mp_mh* h, int proposal_kind, const double* args, int n_args, int n_iters) override {
        auto& reg = mh_fn_proposals<M>();
        auto it = reg.find(proposal_kind);
        if (it == reg.end()) return mp_set_error(MP_ERR_UNSUPPORTED, "no proposal of this kind is registered for the model (MP_REGISTER_MH_PROPOSAL)");
        return it->second.mh(h, model, args, n_args, n_iters);
    }
    int32_t update(mp_mh* h, const mp_fn_consspec& cs, const double* d_cvals, const uint32_t* d_cpresent, int unknown, uint32_t step, bool want_discard) override {
        hipLaunchKernelGGL(k_fn_update<M>, dim3(fn_grid(h)), dim3(MH_THREADS), 0, h->stream, h->n, (uint32_t)h->seed, (uint32_t)(h->seed >> 32), step, model, cs,
                           d_cvals, d_cpresent, unknown, h->fvals, h->fpresent, h->tmp, want_discard ? h->gfi_vals : (double*)nullptr,
                           want_discard ? h->gfi_present : (uint32_t*)nullptr, h->d_acc);
        MHCK(hipGetLastError());
        return MP_OK;
    }
    int32_t regenerate(mp_mh* h, uint64_t mask, int unknown, uint32_t step) override {
        hipLaunchKernelGGL(k_fn_regenerate<M>, dim3(fn_grid(h)), dim3(MH_THREADS), 0, h->stream, h->n, (uint32_t)h->seed, (uint32_t)(h->seed >> 32), step, model,
                           mask, unknown, h->fvals, h->fpresent, h->tmp, h->d_acc);
        MHCK(hipGetLastError());
        return MP_OK;
    }
    int32_t assess(mp_mh* h, const mp_fn_consspec& cs, const double* d_cvals, const uint32_t* d_cpresent, uint32_t step) override {
        hipLaunchKernelGGL(k_fn_assess<M>, dim3(fn_grid(h)), dim3(MH_THREADS), 0, h->stream, h->n, (uint32_t)h->seed, (uint32_t)(h->seed >> 32), step, model, cs,
                           d_cvals, d_cpresent, h->tmp, h->d_acc);
        MHCK(hipGetLastError());
        return MP_OK;
    }
    int32_t generate(mp_mh* h, const mp_fn_consspec& cs, const double* d_cvals, const uint32_t* d_cpresent, uint32_t step) override {
        hipLaunchKernelGGL(k_fn_generate<M>, dim3(fn_grid(h)), dim3(MH_THREADS), 0, h->stream, h->n, (uint32_t)h->seed, (uint32_t)(h->seed >> 32), step, model, cs,
                           d_cvals, d_cpresent, h->fvals, h->fpresent, h->tmp, h->d_acc);
        MHCK(hipGetLastError());
        return MP_OK;
    }
    int32_t simulate(mp_mh* h, uint32_t step) override {
        hipLaunchKernelGGL(k_fn_simulate<M>, dim3(fn_grid(h)), dim3(MH_THREADS), 0, h->stream, h->n, (uint32_t)h->seed, (uint32_t)(h->seed >> 32), step, model,
                           h->fvals, h->fpresent, h->tmp);
        MHCK(hipGetLastError());
        return MP_OK;
    }
    int32_t propose(mp_mh* h, int proposal_kind, const double* args, int n_args, uint32_t step) override {
        auto& reg = mh_fn_proposals<M>();
        auto it = reg.find(proposal_kind);
        if (it == reg.end()) return mp_set_error(MP_ERR_UNSUPPORTED, "no proposal of this kind is registered for the model (MP_REGISTER_MH_PROPOSAL)");
        return it->second.propose(h, model, args, n_args, step);
    }
    int32_t assess_proposal(mp_mh* h, int proposal_kind, const double* args, int n_args, const mp_fn_consspec& cs, const double* d_cvals,
                            const uint32_t* d_cpresent, uint32_t step) override {
        auto& reg = mh_fn_proposals<M>();
        auto it = reg.find(proposal_kind);
        if (it == reg.end()) return mp_set_error(MP_ERR_UNSUPPORTED, "no proposal of this kind is registered for the model (MP_REGISTER_MH_PROPOSAL)");
        return it->second.assess(h, model, args, n_args, cs, d_cvals, d_cpresent, step);
    }
    int32_t logjp(mp_mh* h) override {
        hipLaunchKernelGGL(k_fn_logjp<M>, dim3(fn_grid(h)), dim3(MH_THREADS), 0, h->stream, h->n, model, (const double*)h->fvals,
                           (const uint32_t*)h->fpresent, h->tmp);
        MHCK(hipGetLastError());
        return MP_OK;
    }
};

typedef std::function<std::shared_ptr<mh_fn_ops>(const double*, int, std::string&)> mh_fn_factory;
static std::map<int, mh_fn_factory>& mh_fn_models() {
    static std::map<int, mh_fn_factory> r;
    return r;
}
template <class M>
static int mp_mh_register_model(int kind, bool (*parse)(const double*, int, M&, std::string&)) {
    static_assert(std::is_trivially_copyable<M>::value, "a model functor travels to the kernels by value");
    mh_fn_models()[kind] = [parse](const double* params, int n, std::string& err) -> std::shared_ptr<mh_fn_ops> {
        auto ops = std::make_shared<mh_fn_ops_t<M>>();
        if (!parse(params, n, ops->model, err)) return nullptr;
        return ops;
    };
    return kind;
}
template <class M, class P>
static int mp_mh_register_proposal(int kind, bool (*parse)(const double*, int, P&, std::string&)) {
    static_assert(std::is_trivially_copyable<P>::value, "a proposal functor travels to the kernels by value");
    mh_fn_launcher<M> L;
    L.mh = [parse](mp_mh* h, const M& model, const double* args, int n_args, int n_iters) -> int32_t {
        P prop;
        std::string err;
        if (!parse(args, n_args, prop, err)) return mp_set_error(MP_ERR_INVALID_ARG, err);
        hipLaunchKernelGGL((k_fn_mh<M, P>), dim3(fn_grid(h)), dim3(MH_THREADS), 0, h->stream, h->n, (uint32_t)h->seed, (uint32_t)(h->seed >> 32),
                           (uint32_t)(h->iters + 1), n_iters, model, prop, h->fvals, h->fpresent, h->d_acc);
        MHCK(hipGetLastError());
        return MP_OK;
    };
    L.propose = [parse](mp_mh* h, const M& model, const double* args, int n_args, uint32_t step) -> int32_t {
        P prop;
        std::string err;
        if (!parse(args, n_args, prop, err)) return mp_set_error(MP_ERR_INVALID_ARG, err);
        hipLaunchKernelGGL((k_fn_propose<M, P>), dim3(fn_grid(h)), dim3(MH_THREADS), 0, h->stream, h->n, (uint32_t)h->seed, (uint32_t)(h->seed >> 32), step,
                           model, prop, (const double*)h->fvals, (const uint32_t*)h->fpresent, h->tmp, h->gfi_vals, h->gfi_present);
        MHCK(hipGetLastError());
        return MP_OK;
    };
    L.assess = [parse](mp_mh* h, const M& model, const double* args, int n_args, const mp_fn_consspec& cs, const double* d_cvals,
                       const uint32_t* d_cpresent, uint32_t step) -> int32_t {
        P prop;
        std::string err;
        if (!parse(args, n_args, prop, err)) return mp_set_error(MP_ERR_INVALID_ARG, err);
        hipLaunchKernelGGL((k_fn_assess_proposal<M, P>), dim3(fn_grid(h)), dim3(MH_THREADS), 0, h->stream, h->n, (uint32_t)h->seed, (uint32_t)(h->seed >> 32),
                           step, model, prop, cs, d_cvals, d_cpresent, (const double*)h->fvals, (const uint32_t*)h->fpresent, h->tmp, h->d_acc);
        MHCK(hipGetLastError());
        return MP_OK;
    };
    mh_fn_proposals<M>()[kind] = L;
    return kind;
}
#define MP_REGISTER_MH_MODEL(KIND, TYPE, PARSE) static const int mp_mh_registered_##TYPE = mp_mh_register_model<TYPE>(KIND, PARSE);
#define MP_REGISTER_MH_PROPOSAL(KIND, MODEL, TYPE, PARSE) static const int mp_mh_registered_##TYPE = mp_mh_register_proposal<MODEL, TYPE>(KIND, PARSE);
#include "mp_mh_models.h"
