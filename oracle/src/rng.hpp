// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product; only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
//
// rng.hpp — the oracle's stand-in for `rand::rngs::ThreadRng` + `u01`
// (modppl/src/modeling/dists/distribution.rs:5-7).
//
// The reference RNG cannot be seeded, so "same uniforms" is defined by the build
// (modppl_amd/csrc/mp_philox.h documents the counter layout).  This file is an independent
// second implementation of that definition (Philox4x32-10 written from the published
// algorithm, not included from the product) so that the device stream is checked against
// something other than itself.  Pinned by the Random123 known-answer vectors in
// tests/test_oracle_rng.py.
#pragma once
#include <cstdint>

namespace oracle {

inline void philox4x32_10(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4]) {
    uint32_t c[4] = {ctr_in[0], ctr_in[1], ctr_in[2], ctr_in[3]};
    uint32_t k[2] = {key_in[0], key_in[1]};
    for (int round = 0; round < 10; ++round) {
        if (round > 0) {
            k[0] += 0x9E3779B9u;  // golden ratio
            k[1] += 0xBB67AE85u;  // sqrt(3)-1
        }
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n[4] = {(uint32_t)(p1 >> 32) ^ c[1] ^ k[0], (uint32_t)p1,
                               (uint32_t)(p0 >> 32) ^ c[3] ^ k[1], (uint32_t)p0};
        c[0] = n[0]; c[1] = n[1]; c[2] = n[2]; c[3] = n[3];
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

enum Domain : uint32_t { DOM_MODEL = 0, DOM_RESAMPLE = 1, DOM_ACCEPT = 2, DOM_PROPOSAL = 3, DOM_DATA = 4, DOM_IS = 5 };

// A sequential uniform stream for ONE (slot, step, domain, site): the n-th u01() is half
// (n & 1) of Philox block n >> 1.  `Rng` plays the role of `&mut ThreadRng` in every
// `Distribution::random(rng, params)` call.
struct Rng {
    uint64_t seed = 0;
    uint32_t slot = 0, step = 0, domain = DOM_MODEL, site = 0;
    uint32_t n = 0;  // uniforms consumed at this site so far

    void at(uint32_t domain_, uint32_t site_) { domain = domain_; site = site_; n = 0; }

    uint64_t bits64() {
        const uint32_t ctr[4] = {slot, step, (domain << 16) | site, n >> 1};
        const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
        uint32_t o[4];
        philox4x32_10(ctr, key, o);
        const uint64_t r = (n & 1) ? (((uint64_t)o[3] << 32) | o[2]) : (((uint64_t)o[1] << 32) | o[0]);
        ++n;
        return r;
    }
    // 52-bit integer k with u = k * 2^-52 (rand 0.8.5 UniformFloat<f64>::sample: bits >> 12 into
    // the mantissa of [1,2), minus 1; scale = 1, low = 0).
    uint64_t u52() { return bits64() >> 12; }
    double u01() { return (double)u52() * 0x1p-52; }
};

// The categorical draws of ONE resample (or of importance_resampling's M draws) are one sequential uniform stream, as in the
// reference, whose `for i in 0..N { categorical.random(&mut rng, ..) }` consumes one rng (particle_filter.rs:38-40,
// importance.rs:45-47): draw g is uniform number g of that stream = half (g & 1) of Philox block g >> 1, the block index
// carried in the counter's slot field (site 0 of `domain`).
inline Rng resample_rng(uint64_t seed, uint32_t domain, uint32_t step, uint64_t g) {
    Rng r; r.seed = seed; r.slot = (uint32_t)(g >> 1); r.step = step; r.at(domain, 0);
    r.n = (uint32_t)(g & 1);
    return r;
}
inline uint64_t resample_u52(uint64_t seed, uint32_t domain, uint32_t step, uint64_t g) { return resample_rng(seed, domain, step, g).u52(); }
inline double resample_u01(uint64_t seed, uint32_t domain, uint32_t step, uint64_t g) {
    return (double)resample_u52(seed, domain, step, g) * 0x1p-52;
}
// 32-bit uniforms of a lattice scheme, one per output slot (stratified): word (g & 3) of Philox block g >> 2 at `site`
inline uint32_t resample_k32(uint64_t seed, uint32_t domain, uint32_t site, uint32_t step, uint64_t g) {
    const uint32_t ctr[4] = {(uint32_t)(g >> 2), step, (domain << 16) | site, 0u};
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t o[4];
    philox4x32_10(ctr, key, o);
    return o[g & 3];
}

}  // namespace oracle
