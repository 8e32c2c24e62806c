"""GPU: the scalar building blocks evaluated on the device are bit-identical to the host:
mp_exp / mp_log (same source), IEEE sqrt and division, the Philox stream and the polar sampler."""
import ctypes as C

import numpy as np
import pytest

from tests import oracle_lib as O

pytestmark = pytest.mark.gpu
DP = C.POINTER(C.c_double)


def _probe(hiplib, op, a, b=None, c=None):
    from modppl_amd import capi

    a = np.ascontiguousarray(a, dtype=np.float64)
    out = np.empty_like(a)
    p = lambda v: None if v is None else np.ascontiguousarray(v, dtype=np.float64).ctypes.data_as(DP)
    capi.check(hiplib.mp_probe_math(op, p(a), p(b), p(c), a.size, out.ctypes.data_as(DP), 0))
    return out


def _bits(x):
    return np.asarray(x, dtype=np.float64).view(np.uint64)


def test_exp_log_bitwise(hiplib, oracle):
    rng = np.random.default_rng(0)
    n = 1 << 21
    xe = np.concatenate([rng.uniform(-745, 709, n), rng.uniform(-40, 0, n), rng.normal(0, 1, n),
                         [0.0, -0.0, 1.0, -1.0, 709.78, 709.79, -745.13, -745.14, -800, 800, np.inf, -np.inf, np.nan, 1e-300, -1e-300]])
    ref = np.empty_like(xe)
    oracle.oracle_mp_exp(O.dptr(xe), xe.size, O.dptr(ref))
    assert np.array_equal(_bits(_probe(hiplib, 0, xe)), _bits(ref))
    xl = np.concatenate([np.exp(rng.uniform(-700, 700, n)), rng.uniform(0, 2, n), rng.uniform(0, 1, n) ** 8,
                         [0.0, -0.0, 1.0, -1.0, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, np.inf, np.nan, 0.5, 2.0]])
    ref = np.empty_like(xl)
    oracle.oracle_mp_log(O.dptr(xl), xl.size, O.dptr(ref))
    assert np.array_equal(_bits(_probe(hiplib, 1, xl)), _bits(ref))


def test_sqrt_div_ieee(hiplib):
    rng = np.random.default_rng(1)
    n = 1 << 21
    a = np.concatenate([np.exp(rng.uniform(-700, 700, n)), rng.uniform(0, 4, n), [0.0, 4.0, 2.0, 1e-310, np.inf]])
    assert np.array_equal(_bits(_probe(hiplib, 2, a)), _bits(np.sqrt(a)))
    x = np.concatenate([rng.normal(0, 1, n) * np.exp(rng.uniform(-300, 300, n)), rng.normal(0, 3, n)])
    y = np.concatenate([np.exp(rng.uniform(-300, 300, n)) * rng.choice([-1, 1], n), rng.uniform(0.01, 3, n)])
    with np.errstate(all="ignore"):
        assert np.array_equal(_bits(_probe(hiplib, 3, x, y)), _bits(x / y))


def test_division_by_a_hoisted_constant_is_the_ieee_division_on_device(hiplib):
    """mp_div_hoisted (mp_math.h) == x / d, bit for bit, on the device: what the observation log-densities of lgssm1 / bearings /
    the banded models and of the MH kernels are divided with"""
    from tests.test_math import _div_cases

    x, d = _div_cases(np.random.default_rng(8), 1 << 18)
    got = _probe(hiplib, 5, x, d)
    with np.errstate(all="ignore"):
        want = x / d
    ok = ~np.isnan(want)
    assert np.array_equal(_bits(got[ok]), _bits(want[ok]))
    assert np.all(np.isnan(got[~ok]))


def test_logpdf_with_a_hoisted_reciprocal_on_device(hiplib):
    """mp_normal_logpdf_h == mp_normal_logpdf (the dividing form) on the device for every input, extremes included"""
    from tests.test_math import _div_cases

    x, d = _div_cases(np.random.default_rng(11), 1 << 17)
    d = np.abs(d)
    mu = np.where(np.arange(x.size) % 3 == 0, 0.0, np.random.default_rng(12).normal(0, 1, x.size))
    a, b = _probe(hiplib, 6, x, mu, d), _probe(hiplib, 4, x, mu, d)
    nan = np.isnan(b)
    assert np.array_equal(_bits(a[~nan]), _bits(b[~nan]))
    assert np.all(np.isnan(a[nan]))


def test_normal_logpdf_kats_on_device(hiplib):
    # modppl/tests/dists.rs:120-136 (epsilon there: f32::EPSILON)
    x = np.array([1.4, 2.8, -3.14])
    mu = np.array([0.9, 1.8, 8.0])
    sd = np.array([0.5, 1.0, 20.0])
    got = _probe(hiplib, 4, x, mu, sd)
    assert np.allclose(got, [-0.7257913526447272, -1.4189385332046727, -4.069795306758664], rtol=0, atol=1e-15)


def test_philox_stream_and_sampler(hiplib, oracle):
    from modppl_amd import capi

    n = 1 << 16
    out = np.empty(2 * n)
    capi.check(hiplib.mp_probe_u01(77, 5, 3, 1, 2, 0, n, out.ctypes.data_as(DP), 0))
    ref = np.empty(2)
    for i in [0, 1, 17, n - 1]:
        oracle.oracle_u01_stream(77, 5 + i, 3, 1, 2, 2, O.dptr(ref))
        assert np.array_equal(out[2 * i:2 * i + 2], ref)
    assert 0.0 <= out.min() and out.max() < 1.0
    s = np.empty(n)
    capi.check(hiplib.mp_probe_normal_sample(123, 0, 9, 0, 0, 1.64, 0.025, n, s.ctypes.data_as(DP), 0))
    for i in range(0, n, 997):
        assert s[i] == oracle.oracle_normal_random(123, i, 9, 0, 0, 1.64, 0.025, 1)
    # modppl/tests/dists.rs:113-118 moment check
    assert abs(s.mean() - 1.64) < 0.001 and abs(s.std(ddof=1) - 0.025) < 0.001
