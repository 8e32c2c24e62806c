// mp_math.h — deterministic fp64 exp/log shared by the gfx950 kernels and by the
// host-side checker.
//
// Why this exists: modppl's CPU path calls libm (`f64::exp`, `f64::ln`:
// modppl/src/lib.rs:41-43, modeling/dists/normal.rs:16,25).  glibc's exp/log and ROCm's
// ocml exp/log differ in the last ulp on a small fraction of inputs, and a one-ulp change in
// one weight can move a resample index.  Bit-exact resample indices between the device and a
// CPU checker therefore need ONE definition of exp and log that is evaluated with the same
// IEEE-754 operations, in the same order, on both sides.  Everything here uses only
// + - * / fma rint and integer bit manipulation; compile with -ffp-contract=off on both
// sides so that no other contraction happens.
//
// Accuracy (measured in tests/test_math.py against 80-bit long double): < 1 ulp.
//   mp_exp: Cody–Waite reduction by ln2 (hi/lo) + degree-13 Taylor polynomial, fma Horner.
//   mp_log: the classical fdlibm e_log.c scheme (argument reduced to [sqrt(2)/2, sqrt(2)),
//           s = f/(2+f), even/odd split minimax polynomial Lg1..Lg7).  fdlibm is
//           "Copyright (C) 1993 by Sun Microsystems, Inc. Permission to use, copy, modify,
//           and distribute this software is freely granted, provided that this notice is
//           preserved."  Only the published constants and the scheme are used.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MP_HD __host__ __device__ __forceinline__
#else
#define MP_HD inline
#endif

MP_HD uint64_t mp_f2u(double x) { return __builtin_bit_cast(uint64_t, x); }
MP_HD double mp_u2f(uint64_t u) { return __builtin_bit_cast(double, u); }

#define MP_INF (mp_u2f(0x7FF0000000000000ull))
#define MP_NEG_INF (mp_u2f(0xFFF0000000000000ull))
#define MP_LN_2PI 1.8378770664093453  /* ln(2*pi), nearest double: 0x3FFD67F1C864BEB5 */
#define MP_2PI 6.283185307179586      /* 2*pi as the reference computes it: 2.*PI */
#define MP_PI 3.141592653589793

// ---------------------------------------------------------------------------------------
// exp
// ---------------------------------------------------------------------------------------
MP_HD double mp_exp(double x) {
    if (x != x) return x;
    if (x > 709.782712893384) return MP_INF;
    if (x < -745.1332191019412) return 0.0;
    const double INV_LN2 = 1.4426950408889634;
    const double LN2_HI = 6.93147180369123816490e-01;  // 0x3FE62E42FEE00000
    const double LN2_LO = 1.90821492927058770002e-10;  // 0x3DEA39EF35793C76
    const double kf = rint(x * INV_LN2);
    double r = fma(-kf, LN2_HI, x);
    r = fma(-kf, LN2_LO, r);
    // P(r) = sum_{n=2..13} r^(n-2)/n!
    double p = 1.6059043836821613e-10;   // 1/13!
    p = fma(p, r, 2.08767569878681e-09); // 1/12!
    p = fma(p, r, 2.505210838544172e-08);   // 1/11!
    p = fma(p, r, 2.755731922398589e-07);   // 1/10!
    p = fma(p, r, 2.7557319223985893e-06);  // 1/9!
    p = fma(p, r, 2.48015873015873e-05);    // 1/8!
    p = fma(p, r, 1.984126984126984e-04);   // 1/7!
    p = fma(p, r, 1.388888888888889e-03);   // 1/6!
    p = fma(p, r, 8.333333333333333e-03);   // 1/5!
    p = fma(p, r, 4.1666666666666664e-02);  // 1/4!
    p = fma(p, r, 1.6666666666666666e-01);  // 1/3!
    p = fma(p, r, 0.5);
    const double y = 1.0 + fma(r * r, p, r);  // in [0.70, 1.42]
    const int k = (int)kf;
    if (k > 1023) {
        return y * mp_u2f((uint64_t)(k - 1 + 1023) << 52) * 2.0;
    }
    if (k < -1022) {
        // k >= -1075: y * 2^(k+54) is normal and exact, the last multiply rounds once.
        return y * mp_u2f((uint64_t)(k + 54 + 1023) << 52) * 5.551115123125783e-17;  // 2^-54
    }
    return y * mp_u2f((uint64_t)(k + 1023) << 52);
}

// ---------------------------------------------------------------------------------------
// log
// ---------------------------------------------------------------------------------------
MP_HD double mp_log(double x) {
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t u = mp_f2u(x);
    uint32_t hx = (uint32_t)(u >> 32);
    int k = 0;
    if (hx < 0x00100000u || (hx >> 31)) {
        if ((u << 1) == 0) return MP_NEG_INF;        // log(+-0) = -inf
        if (hx >> 31) return mp_u2f(0x7FF8000000000000ull);  // log(x<0) = NaN
        k -= 54;                                     // subnormal: scale up
        x *= 18014398509481984.0;                    // 2^54
        u = mp_f2u(x);
        hx = (uint32_t)(u >> 32);
    } else if (hx >= 0x7FF00000u) {
        return x;                                    // inf or NaN
    } else if (u == 0x3FF0000000000000ull) {
        return 0.0;
    }
    // x = 2^k * m, m in [sqrt(2)/2, sqrt(2))
    hx += 0x3FF00000u - 0x3FE6A09Eu;
    k += (int)(hx >> 20) - 0x3FF;
    hx = (hx & 0x000FFFFFu) + 0x3FE6A09Eu;
    u = ((uint64_t)hx << 32) | (u & 0xFFFFFFFFull);
    const double m = mp_u2f(u);
    const double f = m - 1.0;
    const double hfsq = 0.5 * f * f;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    const double dk = (double)k;
    return s * (hfsq + R) + dk * LN2_LO - hfsq + f + dk * LN2_HI;
}

// sqrt and division are IEEE-754 correctly rounded for fp64 both in SSE2 and in the gfx950
// expansion hipcc emits without fast-math (checked bit-for-bit in tests/test_gpu_math.py).
MP_HD double mp_sqrt(double x) { return sqrt(x); }
