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
#define MP_2PI 6.283185307179586      /* 2*pi as the reference computes it: 2.*PI */
#define MP_PI 3.141592653589793

// ---------------------------------------------------------------------------------------
// x / d with the reciprocal r = RN(1 / d) hoisted (d is a model constant: an observation noise, a drift width): the bits of the
// IEEE division without the division.  q0 = RN(x r) is within two ulps of x / d; one residual correction (rem = x - q0 d, exact
// in an fma) makes it faithful, a second one correctly rounded (Markstein's theorem: correct rounding from a faithful quotient
// and the correctly rounded reciprocal; the one exception, a divisor whose significand is all ones, is refused by the hoisting
// side: mp_rcp_hoistable).  mp_div_hoisted falls back to the division itself outside 2^-900 < |x| < 2^900 (zero, subnormal
// neighbourhoods, infinities, NaN); a log-density needs no fallback (mp_normal_logpdf_h): where the branch-free core is not the
// division's bits the quotient's SQUARE underflows to zero or overflows to infinity either way.
// 5 full-rate instructions against ~13 with a quarter-rate v_rcp_f64 (tools/func_cost.hip: 88 -> 25 cycles per wave).
// Checked bit for bit against `/` on 2^24 random numerators per divisor on host and device (tests/test_math.py, test_gpu_math.py).
// ---------------------------------------------------------------------------------------
// The corrected quotient, branch-free: exact (= RN(x / d)) for 2^-960 < |x| < 2^960; an infinite or NaN x r is passed through
// (x = +-inf gives +-inf, as the division does); below that range the last bits may differ from the division's and +-0 loses its sign.
MP_HD double mp_div_hoisted_core(double x, double d, double r) {
    const double q0 = x * r;
    const double q1 = fma(fma(-q0, d, x), r, q0);
    const double q2 = fma(fma(-q1, d, x), r, q1);
    return (fabs(q0) <= 1.7976931348623157e308) ? q2 : q0;
}
// ... and with the division itself outside the exact range: RN(x / d) for every x
MP_HD double mp_div_hoisted(double x, double d, double r) {
    const double ax = fabs(x);
    if (!(ax > 0x1p-900 && ax < 0x1p900)) return x / d;
    return mp_div_hoisted_core(x, d, r);
}
// whether a divisor may be hoisted: finite, well inside the exponent range, significand not all ones
MP_HD bool mp_rcp_hoistable(double d) {
    const uint64_t u = mp_f2u(fabs(d));
    return fabs(d) > 0x1p-100 && fabs(d) < 0x1p100 && (u & 0x000FFFFFFFFFFFFFull) != 0x000FFFFFFFFFFFFFull;
}
// the hoisted reciprocal, or 0 = "not hoisted" (mp_normal_logpdf_h then divides)
MP_HD double mp_rcp_hoist(double d) { return mp_rcp_hoistable(d) ? 1.0 / d : 0.0; }

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

// ---------------------------------------------------------------------------------------
// sin / cos / atan2 — same contract as mp_exp/mp_log: one definition, IEEE ops only, evaluated
// identically on host and device.  Classical fdlibm schemes (k_sin.c, k_cos.c, e_rem_pio2.c medium
// case, s_atan.c, e_atan2.c; Sun Microsystems notice above applies), constants as published.
// Domain of the argument reduction: |x| < 2^20 * pi/2 (~1.6e6); beyond that NaN is returned (the
// models on the path keep angles within a few turns).
// ---------------------------------------------------------------------------------------
MP_HD double mp_ksin(double x, double y) {  // |x| <= pi/4, y = tail
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = x * x;
    const double w = z * z;
    const double r = S2 + z * (S3 + z * S4) + z * w * (S5 + z * S6);
    const double v = z * x;
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
MP_HD double mp_kcos(double x, double y) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = x * x;
    const double w = z * z;
    const double r = z * (C1 + z * (C2 + z * C3)) + w * w * (C4 + z * (C5 + z * C6));
    const double hz = 0.5 * z;
    const double ww = 1.0 - hz;
    return ww + (((1.0 - ww) - hz) + (z * r - x * y));
}
// x = n*(pi/2) + (y0 + y1), |y0| <= pi/4 (+tiny); returns n mod 4 (non-negative), or -1 out of domain
MP_HD int mp_rem_pio2(double x, double* y0, double* y1) {
    const double INVPIO2 = 6.36619772367581382433e-01;
    const double P1 = 1.57079632673412561417e+00;   // first 33 bits of pi/2
    const double P1T = 6.07710050650619224932e-11;  // pi/2 - P1
    const double P2 = 6.07710050630396597660e-11;   // second 33 bits
    const double P2T = 2.02226624879595063154e-21;  // pi/2 - (P1 + P2)
    if (!(fabs(x) < 1647099.0)) return -1;           // also catches NaN / inf
    const double fn = rint(x * INVPIO2);
    double r = x - fn * P1;                          // exact: fn < 2^21, P1 has 33 bits
    double w = fn * P1T;
    double y = r - w;
    // second iteration when cancellation ate too many bits (difference of exponents > 16)
    const int ex = (int)((mp_f2u(x) >> 52) & 0x7FF);
    const int ey = (int)((mp_f2u(y) >> 52) & 0x7FF);
    if (ex - ey > 16) {
        const double t = r;
        w = fn * P2;
        r = t - w;
        w = fn * P2T - ((t - r) - w);
        y = r - w;
    }
    *y0 = y;
    *y1 = (r - y) - w;
    const long long n = (long long)fn;
    return (int)(n & 3);
}
MP_HD double mp_sin(double x) {
    if (fabs(x) <= 0.78539816339744828) return mp_ksin(x, 0.);
    double y0, y1;
    const int n = mp_rem_pio2(x, &y0, &y1);
    if (n < 0) return mp_u2f(0x7FF8000000000000ull);
    switch (n) {
    case 0: return mp_ksin(y0, y1);
    case 1: return mp_kcos(y0, y1);
    case 2: return -mp_ksin(y0, y1);
    default: return -mp_kcos(y0, y1);
    }
}
MP_HD double mp_cos(double x) {
    if (fabs(x) <= 0.78539816339744828) return mp_kcos(x, 0.);
    double y0, y1;
    const int n = mp_rem_pio2(x, &y0, &y1);
    if (n < 0) return mp_u2f(0x7FF8000000000000ull);
    switch (n) {
    case 0: return mp_kcos(y0, y1);
    case 1: return -mp_ksin(y0, y1);
    case 2: return -mp_kcos(y0, y1);
    default: return mp_ksin(y0, y1);
    }
}

MP_HD double mp_atan(double x) {
    const double atanhi[4] = {4.63647609000806093515e-01, 7.85398163397448278999e-01, 9.82793723247329054082e-01, 1.57079632679489655800e+00};
    const double atanlo[4] = {2.26987774529616870924e-17, 3.06161699786838301793e-17, 1.39033110312309984516e-17, 6.12323399573676603587e-17};
    const double aT[11] = {3.33333333333329318027e-01,  -1.99999999998764832476e-01, 1.42857142725034663711e-01,  -1.11111104054623557880e-01,
                           9.09088713343650656196e-02,  -7.69187620504482999495e-02, 6.66107313738753120669e-02,  -5.83357013379057348645e-02,
                           4.97687799461593236017e-02,  -3.65315727442169155270e-02, 1.62858201153657823623e-02};
    if (x != x) return x;
    const bool neg = (mp_f2u(x) >> 63) != 0;
    double ax = fabs(x);
    int id;
    if (ax >= 7.3786976294838206e19) {  // 2^66
        const double r = atanhi[3] + 7.52316384526264005100e-37;
        return neg ? -r : r;
    }
    if (ax < 0.4375) {
        if (ax < 3.7252902984619141e-09) return x;  // 2^-28
        id = -1;
    } else if (ax < 1.1875) {
        if (ax < 0.6875) { id = 0; ax = (2.0 * ax - 1.0) / (2.0 + ax); }
        else { id = 1; ax = (ax - 1.0) / (ax + 1.0); }
    } else {
        if (ax < 2.4375) { id = 2; ax = (ax - 1.5) / (1.0 + 1.5 * ax); }
        else { id = 3; ax = -1.0 / ax; }
    }
    const double z = ax * ax;
    const double w = z * z;
    const double s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    const double s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) {
        const double r = ax - ax * (s1 + s2);
        return neg ? -r : r;
    }
    const double r = atanhi[id] - ((ax * (s1 + s2) - atanlo[id]) - ax);
    return neg ? -r : r;
}
// atan2(y, x) with the usual IEEE special cases (f64::atan2 semantics)
MP_HD double mp_atan2(double y, double x) {
    const double PI = 3.1415926535897931160E+00, PI_LO = 1.2246467991473531772E-16;
    if (x != x || y != y) return x + y;
    const uint64_t ux = mp_f2u(x), uy = mp_f2u(y);
    const bool xneg = (ux >> 63) != 0, yneg = (uy >> 63) != 0;
    const double ax = fabs(x), ay = fabs(y);
    if (ux == 0x3FF0000000000000ull) return mp_atan(y);  // x == 1
    if (ay == 0.) return xneg ? (yneg ? -PI : PI) : y;   // y = +-0
    if (ax == 0.) return yneg ? -PI / 2 : PI / 2;
    if (ax == MP_INF) {
        if (ay == MP_INF) { const double r = xneg ? 3.0 * (PI / 4) : PI / 4; return yneg ? -r : r; }
        const double r = xneg ? PI : 0.0;
        return yneg ? -r : r;
    }
    if (ay == MP_INF) return yneg ? -PI / 2 : PI / 2;
    const int ex = (int)((ux >> 52) & 0x7FF), ey = (int)((uy >> 52) & 0x7FF);
    double z;
    if (ey - ex > 60) z = PI / 2 + 0.5 * PI_LO;           // |y/x| > 2^60
    else if (xneg && (ex - ey) > 60) z = 0.0;             // |y/x| < 2^-60, x < 0
    else z = mp_atan(fabs(y / x));
    if (!xneg) return yneg ? -z : z;
    const double r = PI - (z - PI_LO);
    return yneg ? -r : r;
}
