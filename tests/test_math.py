"""CPU: accuracy and edge cases of the deterministic exp/log (mp_math.h) against 80-bit long double,
and the Philox stream against the Random123 known-answer vectors."""
import ctypes as C

import numpy as np

from tests import oracle_lib as O


def _ulp_err(got, ref_ld):
    got_ld = got.astype(np.longdouble)
    ulp = np.spacing(np.abs(ref_ld.astype(np.float64))).astype(np.longdouble)
    return np.abs(got_ld - ref_ld) / ulp


def test_mp_exp_accuracy(oracle):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-700, 700, 400000), rng.uniform(-40, 1, 400000), rng.normal(0, 1e-3, 100000)])
    out = np.empty_like(x)
    oracle.oracle_mp_exp(O.dptr(x), x.size, O.dptr(out))
    err = _ulp_err(out, np.exp(x.astype(np.longdouble)))
    assert err.max() < 1.0, err.max()
    assert np.mean(out == np.exp(x)) > 0.85  # agrees with glibc on most inputs, differs by 1 ulp on some


def test_mp_log_accuracy(oracle):
    rng = np.random.default_rng(1)
    x = np.concatenate([np.exp(rng.uniform(-700, 700, 400000)), rng.uniform(0, 1, 400000), 1 + rng.normal(0, 1e-4, 100000)])
    out = np.empty_like(x)
    oracle.oracle_mp_log(O.dptr(x), x.size, O.dptr(out))
    err = _ulp_err(out, np.log(x.astype(np.longdouble)))
    assert err.max() < 1.0, err.max()


def test_mp_edge_cases(oracle):
    x = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 709.782712893384, 709.79, -745.13, -745.14, -1000.0, 1e-320])
    out = np.empty_like(x)
    oracle.oracle_mp_exp(O.dptr(x), x.size, O.dptr(out))
    assert out[0] == 1.0 and out[1] == 1.0 and out[2] == np.inf and out[3] == 0.0 and np.isnan(out[4])
    assert np.isfinite(out[5]) and out[6] == np.inf and out[7] > 0 and out[8] == 0.0 and out[9] == 0.0 and out[10] == 1.0
    y = np.array([0.0, -0.0, 1.0, -1.0, np.inf, np.nan, 5e-324, 2.2250738585072014e-308, 2 * np.pi])
    o2 = np.empty_like(y)
    oracle.oracle_mp_log(O.dptr(y), y.size, O.dptr(o2))
    assert o2[0] == -np.inf and o2[1] == -np.inf and o2[2] == 0.0 and np.isnan(o2[3]) and o2[4] == np.inf and np.isnan(o2[5])
    assert abs(o2[6] - -744.4400719213812) < 1e-12 and abs(o2[7] - -708.3964185322641) < 1e-12
    # the hoisted constant of mp_dists.h: MP_LN_2PI_CANON must be mp_log(2*pi) bit for bit
    assert o2[8].view(np.uint64) == 0x3FFD67F1C864BEB4


def test_philox_known_answers(oracle):
    """Random123 kat_vectors, philox4x32-10."""
    def ph(ctr, key):
        c = (C.c_uint32 * 4)(*ctr)
        k = (C.c_uint32 * 2)(*key)
        o = (C.c_uint32 * 4)()
        oracle.oracle_philox(c, k, o)
        return list(o)
    assert ph([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert ph([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert ph([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_u01_grid_and_stream_layout(oracle):
    n = 4096
    u = np.empty(n)
    oracle.oracle_u01_stream(99, 3, 4, 1, 2, n, O.dptr(u))
    assert u.min() >= 0.0 and u.max() < 1.0
    assert np.all(u * 2.0 ** 52 == np.floor(u * 2.0 ** 52))  # the 52-bit grid of rand 0.8 Uniform<f64>
    assert abs(u.mean() - 0.5) < 0.02
    # n-th uniform = half (n & 1) of block n >> 1: the stream restarts identically
    v = np.empty(10)
    oracle.oracle_u01_stream(99, 3, 4, 1, 2, 10, O.dptr(v))
    assert np.array_equal(u[:10], v)
    w = np.empty(10)
    oracle.oracle_u01_stream(99, 3, 4, 1, 3, 10, O.dptr(w))  # another site: another stream
    assert not np.array_equal(v, w)


def test_normal_sampler_moments(oracle):
    # modppl/tests/dists.rs:107-118
    s = np.array([oracle.oracle_normal_random(1, i, 0, 0, 0, 1.64, 0.025, 0) for i in range(50000)])
    assert abs(s.mean() - 1.64) < 0.001 and abs(s.std(ddof=1) - 0.025) < 0.001


def _div_cases(rng, n):
    """numerators x divisors for mp_div_hoisted: model constants (noise levels, drift widths), awkward significands, and the
    numerator ranges a log-density sees — plus the ranges that must take the real division (tiny, huge, zero, inf, NaN)"""
    ds = [0.1, 0.5, 5.0, 2.0, 1.0, 0.02, 0.05, 0.025, 0.3, 1.0 / 3.0, 7.0, 1e-3, 123.456, np.pi, 1.0 + 2.0 ** -52, 2.0 - 2.0 ** -52,
          0.9999999999999999, 1.5, 0.15, 0.2, 0.25, 0.4, 20.0, 1e10, 1e-10, -0.1, -3.7]
    xs, dd = [], []
    for d in ds:
        x = np.concatenate([rng.normal(0, 1, n), rng.normal(0, 1, n) * np.exp(rng.uniform(-60, 60, n)), rng.uniform(-1, 1, n // 4) * 2.0 ** rng.integers(-1000, 1000, n // 4),
                            [0.0, -0.0, np.inf, -np.inf, np.nan, 1e-310, -1e-310, 2.0 ** -900, 2.0 ** 900, 1.7976931348623157e308, 5e-324]])
        xs.append(x)
        dd.append(np.full(x.size, d))
    return np.concatenate(xs), np.concatenate(dd)


def test_division_by_a_hoisted_constant_is_the_ieee_division(oracle):
    """mp_div_hoisted(x, d, RN(1 / d)) == x / d, bit for bit (mp_math.h: Markstein's correction twice), on the host build"""
    x, d = _div_cases(np.random.default_rng(7), 1 << 18)
    out = np.empty_like(x)
    oracle.oracle_mp_div_hoisted(O.dptr(x), O.dptr(d), x.size, O.dptr(out))
    with np.errstate(all="ignore"):
        want = x / d
    assert np.array_equal(out.view(np.uint64)[~np.isnan(want)], want.view(np.uint64)[~np.isnan(want)])
    assert np.all(np.isnan(out[np.isnan(want)]))


def test_logpdf_with_a_hoisted_reciprocal_has_the_bits_of_the_dividing_one(oracle):
    """mp_normal_logpdf_h (branch-free corrected quotient) == mp_normal_logpdf_ln (IEEE division) for EVERY input: the ranges where
    the quotient itself may differ are exactly those where its square underflows or overflows"""
    x, d = _div_cases(np.random.default_rng(9), 1 << 17)
    d = np.abs(d)
    mu = np.where(np.arange(x.size) % 3 == 0, 0.0, np.random.default_rng(10).normal(0, 1, x.size))
    a, b = np.empty_like(x), np.empty_like(x)
    oracle.oracle_mp_normal_logpdf_both(O.dptr(x), O.dptr(mu), O.dptr(d), x.size, O.dptr(a), O.dptr(b))
    nan = np.isnan(b)
    assert np.array_equal(a.view(np.uint64)[~nan], b.view(np.uint64)[~nan])
    assert np.all(np.isnan(a[nan]))
