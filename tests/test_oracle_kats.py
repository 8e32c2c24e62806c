"""CPU: the oracle (restated reference) pinned by the reference's own known-answer tests.

Every expected constant below is quoted from a reference test file (cited per test); none is
produced by the build.  These run without a GPU."""
import ctypes as C
import math

import numpy as np
import pytest

from tests import oracle_lib as O

LOGPDF_EPSILON = float(np.finfo(np.float32).eps)  # modppl/tests/dists.rs:8


def test_normal_logpdf_kats(oracle):
    # modppl/tests/dists.rs:120-136
    for canon in (0, 1):
        assert abs(oracle.oracle_normal_logpdf(1.4, 0.9, 0.5, canon) - -0.7257913526447272) < LOGPDF_EPSILON
        assert abs(oracle.oracle_normal_logpdf(2.8, 1.8, 1.0, canon) - -1.4189385332046727) < LOGPDF_EPSILON
        assert abs(oracle.oracle_normal_logpdf(-3.14, 8.0, 20.0, canon) - -4.069795306758664) < LOGPDF_EPSILON
    # tighter than the reference asks: literal mode reproduces the quoted digits to 1 ulp
    assert abs(oracle.oracle_normal_logpdf(1.4, 0.9, 0.5, 0) - -0.7257913526447272) < 1e-15


def _mv(oracle, x, mu, cov):
    a = lambda v: np.ascontiguousarray(v, dtype=np.float64)
    x, mu, cov = a(x), a(mu), a(cov)
    return oracle.oracle_mvnormal_logpdf(len(x), O.dptr(x), O.dptr(mu), O.dptr(cov))


def test_mvnormal_logpdf_kats(oracle):
    # modppl/tests/dists.rs:164-183
    assert abs(_mv(oracle, [1.1, 5.8], [1.3, 5.6], [1.0, -0.81, -0.81, 2.5]) - -2.1642100746383357) < LOGPDF_EPSILON
    assert abs(_mv(oracle, [30.1, -46.8], [0.0, 6.0], [496.0, 0.13, 0.13, 500.0]) - -11.750458919763666) < LOGPDF_EPSILON
    assert abs(_mv(oracle, [1.2, 5.1, -7.8], [1.4, 5.0, -7.4], [1.0, 0.1, 0.9, 0.1, 1.3, 0.4, 0.9, 0.4, 1.75]) - -2.873267436425841) < LOGPDF_EPSILON


def test_uniform_bernoulli_identities(oracle):
    # modppl/tests/dists.rs:28-29, 43-48
    assert oracle.oracle_bernoulli_logpdf(1, 0.11) == math.log(0.11)
    assert oracle.oracle_bernoulli_logpdf(0, 0.11) == math.log(1.0 - 0.11)
    true_p = 1.0 / (3.14 - 0.5)
    assert oracle.oracle_uniform_logpdf(0.9, 0.5, 3.14) == pytest.approx(math.log(true_p), abs=1e-15)
    assert oracle.oracle_uniform_logpdf(2.1, 0.5, 3.14) == pytest.approx(math.log(true_p), abs=1e-15)
    assert oracle.oracle_uniform_logpdf(0.4, 0.5, 3.14) == -math.inf
    assert math.isnan(oracle.oracle_uniform_logpdf(0.4, 3.0, 1.0))  # a >= b panics in the reference (uniform.rs:8-12)


def test_uniform2d_kat(oracle):
    # modppl/tests/test_pointed.rs:13,20,23
    assert abs(oracle.oracle_uniform2d_logpdf(1.0, -0.5, 0.0, 2.5, -1.0, 0.25) - -1.1394342831883648) <= np.finfo(np.float64).eps
    assert oracle.oracle_uniform2d_logpdf(-1.0, 0.0, 0.0, 2.5, -1.0, 0.25) == -math.inf


def test_update_weight_kats(oracle):
    # modppl/tests/dyngenfn.rs:55-114
    out = np.zeros(5)
    assert oracle.oracle_kat_update_weights(1, O.dptr(out)) == 0
    assert out[0] == -0.5                      # :65 assert_eq!
    assert abs(out[1] - -2.517551) < 1e-6      # :78
    assert abs(out[2] - 0.4) < 1e-6            # :92
    assert abs(out[3] - -1.098612) < 1e-6      # :104
    assert abs(out[4] - -1.098612) < 1e-6      # :113


def test_residual_constraints_panic(oracle):
    # modppl/tests/dyngenfn.rs:116-131 (#[should_panic] x2)
    assert oracle.oracle_kat_residual_panics(3) == 1


def test_update_discard_and_weights(oracle):
    # modppl/tests/dyngenfn.rs:180-301
    out = np.zeros(19)
    assert oracle.oracle_kat_update(11, O.dptr(out)) == 0, oracle.oracle_last_error()
    assert out[0] == 1 and out[1] == 1 and out[2] == 1      # discard: branch==true, x, u/a (:214-216)
    assert out[3] == 2 and out[4] == 1                      # :217-218
    assert out[5] == 1 and out[6] == 1.123 and out[7] == -2.1   # :222-224
    assert out[8] == 2 and out[9] == 1                      # :225-226
    assert out[10] < 1e-3 and out[11] < 1e-3                # :238-239
    assert out[12] == 0.0 and out[13] < 1e-3 and out[14] < 1e-3  # loopy :268-275
    assert out[15] == 1 and out[16] == 1                    # :291-292
    assert out[17] == out[18]                               # :293-300 assert_eq! on the exact expression


def test_regenerate_formula(oracle):
    # modppl/tests/dyngenfn.rs:303-388, epsilon 1e-3 there
    for seed in (1, 2, 3, 4, 5):
        out = np.zeros(30)
        assert oracle.oracle_kat_regenerate(seed, O.dptr(out)) == 0, oracle.oracle_last_error()
        assert np.all(out[0::3] < 1e-3)   # logjp
        assert np.all(out[1::3] < 1e-3)   # weight
        assert np.all(out[2::3] == 1.0)   # structure


def test_simulate_logjp(oracle):
    # modppl/tests/dyngenfn.rs:163-178
    for seed in range(8):
        assert oracle.oracle_kat_simulate(seed) == 0.0


def test_categorical_scan_quirks(oracle):
    # categorical.rs:24-31: u == 0 returns -1; running off the end is an index panic (-2 here)
    p = np.array([0.1, 0.3, 0.2, 0.1, 0.05, 0.25])
    assert oracle.oracle_categorical_scan(0.0, O.dptr(p), 6) == -1
    assert oracle.oracle_categorical_scan(0.05, O.dptr(p), 6) == 0
    assert oracle.oracle_categorical_scan(0.1, O.dptr(p), 6) == 0    # t < u is strict: t=0.1 >= u stops
    assert oracle.oracle_categorical_scan(0.100001, O.dptr(p), 6) == 1
    assert oracle.oracle_categorical_scan(0.999, O.dptr(p), 6) == 5
    q = np.array([0.5, 0.4])
    assert oracle.oracle_categorical_scan(0.95, O.dptr(q), 2) == -2


def test_categorical_frequencies(oracle):
    # modppl/tests/dists.rs:84-104 with the seeded stream
    p = np.array([0.1, 0.3, 0.2, 0.1, 0.05, 0.25])
    n = 50000
    u = np.empty(n)
    oracle.oracle_u01_stream(5, 0, 0, 0, 0, n, O.dptr(u))
    idx = np.searchsorted(np.cumsum(p), u, side="left")
    for i in range(6):
        assert abs((idx == i).mean() - p[i]) < 0.01


def test_hmm_forward_algorithm(oracle):
    # modppl/tests/particle_filter.rs:10-33: 2-state HMM equals brute-force enumeration to 1e-16
    prior = [0.4, 0.6]
    emis = np.array([[0.1, 0.9], [0.7, 0.3]]).T      # dmatrix![0.1,0.9;0.7,0.3].transpose()
    trans = np.array([[0.5, 0.5], [0.2, 0.8]]).T
    obs = [1, 0]
    true = 0.0
    for z0 in (0, 1):
        for z1 in (0, 1):
            true += prior[z0] * emis[obs[0], z0] * trans[z1, z0] * emis[obs[1], z1]
    params = np.array([2, 2] + prior + emis.reshape(-1).tolist() + trans.reshape(-1).tolist(), dtype=np.float64)
    o = np.array(obs, dtype=np.float64)
    got = oracle.oracle_hmm_forward(O.dptr(params), len(params), O.dptr(o), 2)
    assert abs(got - true) <= 1e-16


HMM3 = dict(
    prior=[0.2, 0.3, 0.5],
    emis=np.array([[0.1, 0.2, 0.7], [0.2, 0.7, 0.1], [0.7, 0.2, 0.1]]).T,
    trans=np.array([[0.4, 0.4, 0.2], [0.2, 0.3, 0.5], [0.9, 0.05, 0.05]]).T,
)


def hmm3_params():
    return np.array([3, 3] + HMM3["prior"] + HMM3["emis"].reshape(-1).tolist() + HMM3["trans"].reshape(-1).tolist(), dtype=np.float64)


def test_hmm_particle_filter_reference_e2e(oracle):
    # modppl/tests/particle_filter.rs:35-79: 10 000 particles, obs [0,0,1,2], |lml - ln forward| <= 0.03
    params = hmm3_params()
    data = np.array([0.0, 0.0, 1.0, 2.0])
    expected = math.log(oracle.oracle_hmm_forward(O.dptr(params), len(params), O.dptr(data), 4))
    for variant in (O.VARIANT_FAST_SEARCH, O.VARIANT_SOA, O.VARIANT_SOA | O.VARIANT_CANONICAL):
        pf = O.OraclePF(3, 1, 1, params, 10000, 42, variant)
        pf.init_step(data[:1])
        for t in range(1, 4):
            pf.step(data[t:t + 1])
            pf.effective_sample_size()
            pf.resample()
        assert abs(pf.log_marginal_likelihood_estimate() - expected) <= 0.03
