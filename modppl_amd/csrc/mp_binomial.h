// mp_binomial.h — Binomial(n, a / b) variates for the rank counts of the split multinomial resample (mp_pf_shard_kernels.h).
//
// particle_filter.rs:37-41 draws N i.i.d. parents from the job's normalised weights.  The number of those that land on the G
// ranks is Multinomial(N; M_0 / Q .. M_{G-1} / Q) (M_r = rank r's share of the fixed-point mass Q), and given the counts the
// parents of a rank are i.i.d. from that rank's own weights: drawing the counts first and then c_r parents per rank IS that
// law, at O(n) work per rank instead of the O(N) enumeration the single filter's uniform stream needs (exchange = "owned").
// The counts come from a binary splitting of the ranks (mp_split_counts below); each split is ONE binomial variate:
//     n * p >= 10   BTRS, the transformed rejection sampler with squeeze of W. Hoermann, "The generation of binomial random
//                   variates", J. Stat. Comput. Simul. 46 (1993), Algorithm BTRS — restated here from the published steps
//     n * p <  10   sequential search from 0 on the probability recurrence (the same paper's inversion fallback, BINV)
// with p <= 1/2 always (the smaller of the two masses is the one sampled).  Every uniform is named by a Philox counter
// ( node , resample count , RESAMPLE << 16 | 3 , attempt ): attempt a of the sampler at tree node `node` takes block a as its
// pair (U, V), so the variate is a pure function of (seed, resample count, node, n, a, b) — every rank computes the same counts
// without talking to the others.  The arithmetic is single IEEE operations, mp_log and mp_exp (mp_math.h): host and device
// agree bit for bit (oracle/src/inference.hpp restates it; tests/test_split_binomial.py checks both the bits and the law).
#pragma once
#include <stdint.h>

#include "mp_math.h"
#include "mp_philox.h"

constexpr uint32_t MP_SITE_SPLIT_COUNTS = 3u;   // site word of the counts' uniforms in domain RESAMPLE (0: the draws, 1 / 2: lattices)
constexpr uint32_t MP_BINOMIAL_MAX_ATTEMPTS = 1u << 16;

// log(k!) - [ log(sqrt(2 pi)) + (k + 1/2) log(k + 1) - (k + 1) ]: tabulated below 10, the Stirling series beyond (Hoermann §3)
MP_HD double mp_stirling_tail(double k) {
    if (k <= 9.) {
        const int i = (int)k;
        // (a chain of selects, not a table: no constant-address-space array in a device function)
        return i == 0 ? 0.0810614667953272 : i == 1 ? 0.0413406959554092 : i == 2 ? 0.0276779256849983 : i == 3 ? 0.02079067210376509
             : i == 4 ? 0.0166446911898211 : i == 5 ? 0.0138761288230707 : i == 6 ? 0.0118967099458917 : i == 7 ? 0.0104112652619720
             : i == 8 ? 0.00925546218271273 : 0.00833056343336287;
    }
    const double kp1 = k + 1.;
    const double kp1sq = kp1 * kp1;
    return (1.0 / 12. - (1.0 / 360. - 1.0 / 1260. / kp1sq) / kp1sq) / kp1;
}

// BTRS in three pieces, so that a device caller can evaluate several attempts of one variate in different lanes (mp_pf_shard_kernels.h,
// mp_split_counts) with the very operations the sequential sampler below performs: the constants of (n, p); the cheap part of an
// attempt (candidate k, range check, squeeze); the expensive acceptance test of an attempt the squeeze did not decide.
struct mp_btrs {
    double n, p, spq, b, a, c, v_r, r, alpha, m;
};
MP_HD void mp_btrs_setup(mp_btrs& T, double n, double p) {
    const double q = 1. - p;
    T.n = n; T.p = p;
    T.spq = mp_sqrt(n * p * q);
    T.b = 1.15 + 2.53 * T.spq;
    T.a = -0.0873 + 0.0248 * T.b + 0.01 * p;
    T.c = n * p + 0.5;
    T.v_r = 0.92 - 4.2 / T.b;
    T.r = p / q;
    T.alpha = (2.83 + 5.1 / T.b) * T.spq;
    T.m = floor((n + 1.) * p);
}
// 0: rejected (k out of range); 1: accepted inside the squeeze (~ 86 % of the accepted pairs); 2: undecided — mp_btrs_slow says
MP_HD int mp_btrs_fast(const mp_btrs& T, double U01, double V01, double* k_out) {
    const double u = U01 - 0.5;
    const double us = 0.5 - fabs(u);
    const double k = floor((2. * T.a / us + T.b) * u + T.c);
    *k_out = k;
    if (!(k >= 0. && k <= T.n)) return 0;             // (also what an infinite or NaN k from us == 0 falls into)
    if (us >= 0.07 && V01 <= T.v_r) return 1;
    return 2;
}
// the acceptance test of an attempt (U, V, k) the squeeze did not decide is v <= ub, eight pieces of two shapes — piece 0 is v, pieces
// 1 .. 3 the terms c log(x / y) of ub, pieces 4 .. 7 its Stirling tails — so that a device caller can have them evaluated by eight lanes
// (mp_split_counts: lane j piece j) and add them up in the order mp_btrs_accept states
MP_HD double mp_btrs_logterm(double c, double x, double y) { return c * mp_log(x / y); }
MP_HD double mp_btrs_piece(const mp_btrs& T, int j, double U01, double V01, double k) {
    const double us = 0.5 - fabs(U01 - 0.5);
    const double n = T.n, m = T.m, r = T.r;
    const int q = j & 3;
    if (j < 4) {
        const double c = q == 0 ? 1. : (q == 1 ? m + 0.5 : (q == 2 ? n + 1. : k + 0.5));
        const double x = q == 0 ? V01 * T.alpha : (q == 1 ? m + 1. : (q == 2 ? n - m + 1. : r * (n - k + 1.)));
        const double y = q == 0 ? T.a / (us * us) + T.b : (q == 1 ? r * (n - m + 1.) : (q == 2 ? n - k + 1. : k + 1.));
        return mp_btrs_logterm(c, x, y);
    }
    return mp_stirling_tail(q == 0 ? m : (q == 1 ? n - m : (q == 2 ? k : n - k)));
}
MP_HD bool mp_btrs_accept(double v, double t1, double t2, double t3, double s_m, double s_nm, double s_k, double s_nk) {
    return v <= t1 + t2 + t3 + s_m + s_nm - s_k - s_nk;
}
MP_HD bool mp_btrs_slow(const mp_btrs& T, double U01, double V01, double k) {
    return mp_btrs_accept(mp_btrs_piece(T, 0, U01, V01, k), mp_btrs_piece(T, 1, U01, V01, k), mp_btrs_piece(T, 2, U01, V01, k),
                          mp_btrs_piece(T, 3, U01, V01, k), mp_btrs_piece(T, 4, U01, V01, k), mp_btrs_piece(T, 5, U01, V01, k),
                          mp_btrs_piece(T, 6, U01, V01, k), mp_btrs_piece(T, 7, U01, V01, k));
}

// X ~ Binomial(n, p), 0 < p <= 1/2, n >= 1; uniforms: blocks 0, 1, ... of (node, rc, RESAMPLE << 16 | 3, .)
MP_HD uint64_t mp_binomial_small_p(uint64_t n_u, double p, uint32_t node, uint32_t rc, uint32_t k0, uint32_t k1) {
    const double n = (double)n_u;
    const double q = 1. - p;
    const uint32_t c2 = ((uint32_t)MP_DOM_RESAMPLE << 16) | MP_SITE_SPLIT_COUNTS;
    if (n * p < 10.) {
        // P(X = 0) = q^n; P(X = k) = P(X = k - 1) (n - k + 1) / k * p / q; walk up from 0 until the uniform is used up
        const double s = p / q;
        const double f0 = mp_exp(n * mp_log(q));
        for (uint32_t att = 0; att < MP_BINOMIAL_MAX_ATTEMPTS; ++att) {
            double u = mp_u01(mp_philox4x32_10(node, rc, c2, att, k0, k1).a);
            double f = f0;
            double k = 0.;
            bool ok = true;
            while (u >= f) {
                u -= f;
                k += 1.;
                if (k > n || k > 512.) { ok = false; break; }   // rounding left a sliver of u beyond the last term: next attempt
                f *= (n - k + 1.) / k * s;
            }
            if (ok) return (uint64_t)k;
        }
        return 0ull;
    }
    mp_btrs T;
    mp_btrs_setup(T, n, p);
    for (uint32_t att = 0; att < MP_BINOMIAL_MAX_ATTEMPTS; ++att) {
        const mp_u64x2 blk = mp_philox4x32_10(node, rc, c2, att, k0, k1);
        const double U = mp_u01(blk.a), V = mp_u01(blk.b);
        double k;
        const int st = mp_btrs_fast(T, U, V, &k);
        if (st == 1 || (st == 2 && mp_btrs_slow(T, U, V, k))) return (uint64_t)k;
    }
    return (uint64_t)T.m;
}

// X ~ Binomial(n, a / b) for integer masses 0 <= a <= b, b > 0 (the left child's share of a node's mass)
MP_HD uint64_t mp_binomial_ratio(uint64_t n, uint64_t a, uint64_t b, uint32_t node, uint32_t rc, uint32_t k0, uint32_t k1) {
    if (n == 0ull || a == 0ull) return 0ull;
    if (a >= b) return n;
    const uint64_t other = b - a;
    if (a <= other) return mp_binomial_small_p(n, (double)a / (double)b, node, rc, k0, k1);
    return n - mp_binomial_small_p(n, (double)other / (double)b, node, rc, k0, k1);
}

// The same variate the way mp_split_counts' lanes find it (mp_pf_shard_kernels.h), restated sequentially for the host: A attempts
// "side by side", the squeeze's verdicts first, the expensive test only for undecided attempts in front of the first accepted one, the
// first accepted attempt wins; the sequential sampler where its inversion branch applies or nothing among the A attempts is accepted.
// tests/test_split_binomial.py holds it to mp_binomial_ratio bit for bit.
MP_HD uint64_t mp_binomial_ratio_lanes(uint64_t n, uint64_t a, uint64_t b, uint32_t node, uint32_t rc, uint32_t k0, uint32_t k1) {
    constexpr int A = 8;
    if (n == 0ull || a == 0ull) return 0ull;
    if (a >= b) return n;
    const uint64_t other = b - a;
    const bool flipped = a > other;
    const double p = (double)(flipped ? other : a) / (double)b;
    const double nd = (double)n;
    if (nd * p < 10.) return mp_binomial_ratio(n, a, b, node, rc, k0, k1);
    mp_btrs T;
    mp_btrs_setup(T, nd, p);
    double U[A], V[A], k[A];
    int st[A];
    uint32_t acc = 0u, und = 0u;
    for (int att = 0; att < A; ++att) {
        const mp_u64x2 blk = mp_philox4x32_10(node, rc, ((uint32_t)MP_DOM_RESAMPLE << 16) | MP_SITE_SPLIT_COUNTS, (uint32_t)att, k0, k1);
        U[att] = mp_u01(blk.a);
        V[att] = mp_u01(blk.b);
        st[att] = mp_btrs_fast(T, U[att], V[att], &k[att]);
        if (st[att] == 1) acc |= 1u << att;
        if (st[att] == 2) und |= 1u << att;
    }
    und &= acc ? ((acc & (0u - acc)) - 1u) : 0xFFu;
    while (und) {
        int cand = 0;
        while (!((und >> cand) & 1u)) ++cand;
        double pc[8];
        for (int j = 0; j < 8; ++j) pc[j] = mp_btrs_piece(T, j, U[cand], V[cand], k[cand]);
        if (mp_btrs_accept(pc[0], pc[1], pc[2], pc[3], pc[4], pc[5], pc[6], pc[7])) {
            acc |= 1u << cand;
            break;
        }
        und &= und - 1u;
    }
    if (acc == 0u) return mp_binomial_ratio(n, a, b, node, rc, k0, k1);
    int first = 0;
    while (!((acc >> first) & 1u)) ++first;
    const uint64_t kk = (uint64_t)k[first];
    return flipped ? n - kk : kk;
}

// Levels of the splitting tree over `world` ranks: its leaves are the ranks padded with empty ones to a power of two
MP_HD int mp_split_levels(int world) {
    int L = 0;
    while ((1 << L) < world) ++L;
    return L;
}
