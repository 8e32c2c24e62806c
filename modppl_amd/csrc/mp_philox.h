// mp_philox.h — the seeded random stream of the MI355X path.
//
// modppl draws every uniform from `rand::rngs::ThreadRng`, which cannot be seeded
// (modppl/src/modeling/dists/distribution.rs:5-7; ThreadRng::default() constructed per call
// at modppl/src/modeling/dynunfold.rs:29,50,82 and modppl/src/inference/mh.rs:35,60).  The
// path therefore has no reference stream to reproduce; this header DEFINES the stream:
// Philox4x32-10 (Salmon et al., SC'11), keyed by the user seed, with the counter naming
// exactly which uniform is being asked for:
//
//     counter = ( slot , step , (domain << 16) | site , attempt )
//
//   slot    global particle index / MH chain index (independent of how slots are sharded)
//   step    Unfold time index t (0-based kernel time), resample count, or MH iteration
//   domain  which consumer: model choice, resample draw, accept test, data simulation ...
//   site    which random choice inside the model (the static stand-in for a trie address)
//   attempt retry counter of a rejection sampler / k-th draw at the same site
//
// One call yields 128 bits = two 52-bit uniforms (u, v): exactly what one attempt of the
// reference's polar normal sampler consumes (normal.rs:19-20).
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MP_PHD __host__ __device__ __forceinline__
#else
#define MP_PHD inline
#endif

enum mp_rng_domain : uint32_t {
    MP_DOM_MODEL = 0,     // choices made by the model kernel
    MP_DOM_RESAMPLE = 1,  // categorical draws of the resampler
    MP_DOM_ACCEPT = 2,    // MH accept/reject uniform
    MP_DOM_PROPOSAL = 3,  // choices made by an MH proposal
    MP_DOM_DATA = 4,      // synthetic observation generation (bench / fixtures)
    MP_DOM_IS = 5,        // importance_resampling's M categorical draws
};

struct mp_u64x2 {
    uint64_t a, b;
};

MP_PHD void mp_mulhilo32(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
    // one 32x32->64 multiply (v_mad_u64_u32 on gfx950) instead of a mul_lo/mul_hi pair
    const uint64_t p = (uint64_t)a * (uint64_t)b;
    hi = (uint32_t)(p >> 32);
    lo = (uint32_t)p;
}

MP_PHD mp_u64x2 mp_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                 uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        mp_mulhilo32(M0, c0, hi0, lo0);
        mp_mulhilo32(M1, c2, hi1, lo1);
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0;
        c1 = lo1;
        c2 = n2;
        c3 = lo0;
        k0 += W0;
        k1 += W1;
    }
    mp_u64x2 out;
    out.a = ((uint64_t)c1 << 32) | c0;
    out.b = ((uint64_t)c3 << 32) | c2;
    return out;
}

// 64 random bits -> the 52-bit grid in [0,1) that rand-0.8 `Uniform::new(0., 1.)` produces
// (bits >> 12 placed in the mantissa of [1,2), minus 1): k * 2^-52, k in [0, 2^52).
MP_PHD uint64_t mp_u52(uint64_t bits) { return bits >> 12; }
// k * 2^-52 without an integer-to-double conversion: 1.mantissa in [1, 2) has spacing 2^-52, so (1 + k 2^-52) - 1 is exact.
MP_PHD double mp_u01(uint64_t bits) { return __builtin_bit_cast(double, (bits >> 12) | 0x3FF0000000000000ull) - 1.0; }

struct mp_stream {
    uint32_t k0, k1;  // seed
    uint32_t slot, step;
    MP_PHD mp_u64x2 draw(uint32_t domain, uint32_t site, uint32_t attempt) const {
        return mp_philox4x32_10(slot, step, (domain << 16) | site, attempt, k0, k1);
    }
};
