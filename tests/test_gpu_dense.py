"""mvnormal beyond k = 2 on the device (modppl/src/modeling/dists/mvnormal.rs:14-38) and the dense-transition LGSSM whose
products run on the matrix cores: the reference's three logpdf constants evaluated ON THE GPU, general-k logpdf / random
against the CPU checker's per-call LU path, the eigen `transform` of singular covariances (:30-33), the accumulation order
of v_mfma_f64_16x16x4_f64, and the d = 16 filter in lockstep with the canonical checker (MFMA kernel == scalar kernel ==
checker, bit for bit)."""
import ctypes as C
from fractions import Fraction

import numpy as np
import pytest

from tests import oracle_lib as O

pytestmark = pytest.mark.gpu
DP = C.POINTER(C.c_double)
MP_MODEL_LGSSM_DENSE = 8


def dptr(a):
    return a.ctypes.data_as(DP)


def dev_logpdf(hiplib, x, mu, cov, chain=0):
    x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
    mu, cov = np.ascontiguousarray(mu, dtype=np.float64), np.ascontiguousarray(cov, dtype=np.float64)
    out = np.zeros(x.shape[0])
    rc = hiplib.mp_probe_mvnormal(len(mu), chain, dptr(x), dptr(mu), dptr(cov), x.shape[0], dptr(out), 0, 0, 0, 0, 0, None, 0)
    assert rc == 0
    return out


def dev_sample(hiplib, mu, cov, n, seed, slot0=0, step=0, site=0, chain=0):
    mu, cov = np.ascontiguousarray(mu, dtype=np.float64), np.ascontiguousarray(cov, dtype=np.float64)
    out = np.zeros((n, len(mu)))
    rc = hiplib.mp_probe_mvnormal(len(mu), chain, None, dptr(mu), dptr(cov), n, None, seed, slot0, step, 0, site, dptr(out), 0)
    assert rc == 0
    return out


def test_reference_mvnormal_logpdf_kats_on_device(hiplib):
    """modppl/tests/dists.rs:164-183 — the three constants, the 3 x 3 one included, to LOGPDF_EPSILON = f32::EPSILON."""
    eps = 1.1920929e-07
    cases = [([1.1, 5.8], [1.3, 5.6], [[1., -0.81], [-0.81, 2.5]], -2.1642100746383357),
             ([30.1, -46.8], [0., 6.], [[496., 0.13], [0.13, 500.]], -11.750458919763666),
             ([1.2, 5.1, -7.8], [1.4, 5.0, -7.4], [[1., 0.1, 0.9], [0.1, 1.3, 0.4], [0.9, 0.4, 1.75]], -2.873267436425841)]
    for x, mu, cov, want in cases:
        for chain in (0, 1):
            got = dev_logpdf(hiplib, x, mu, cov, chain)[0]
            assert abs(got - want) <= eps, (got, want, chain)


def spd(k, rng, scale=1.0):
    m = rng.normal(size=(k, k))
    return scale * (m @ m.T / k + 0.5 * np.eye(k))


def test_general_k_logpdf_and_random_vs_checker(hiplib, oracle):
    """k = 3 .. 64 (the reference's mvnormal takes any k; model sites are compiled for k <= 16, the general-k form is the probe's):
    the device (constants hoisted once) against the checker's per-call determinant / inverse / Cholesky — bit for bit in both
    operation orders (the hoisted routines restate the same eliminations)."""
    rng = np.random.default_rng(7)
    for k in (3, 5, 8, 16, 24, 40, 64):
        mu, cov = rng.normal(size=k), spd(k, rng)
        xs = rng.normal(size=(64, k)) * 1.5
        for chain in (0, 1):
            got = dev_logpdf(hiplib, xs, mu, cov, chain)
            want = np.array([oracle.oracle_mvnormal_logpdf_dense(k, dptr(np.ascontiguousarray(x)), dptr(mu), dptr(np.ascontiguousarray(cov)), chain) for x in xs])
            if chain == 1:
                assert np.array_equal(got, want), (k, np.abs(got - want).max())
            else:   # the literal checker calls libm's log; the device its own mp_log: last-ulp differences allowed
                assert np.allclose(got, want, rtol=1e-14, atol=0), (k, np.abs(got - want).max())
            s = dev_sample(hiplib, mu, cov, 32, seed=99, slot0=5, step=3, site=2, chain=chain)
            ref = np.zeros(k)
            for i in range(32):
                oracle.oracle_mvnormal_random_dense(99, 5 + i, 3, 0, 2, k, dptr(mu), dptr(np.ascontiguousarray(cov)), 1, dptr(ref))
                if chain == 1:
                    assert np.array_equal(s[i], ref), (k, i)


def test_eigen_transform_for_singular_covariance(hiplib, oracle):
    """mvnormal.rs:30-33: a positive semi-definite covariance of rank 2 has no Cholesky factor; samples then come from the
    eigen transform: same bits as the checker's restatement, and the right second moments."""
    rng = np.random.default_rng(3)
    b = rng.normal(size=(4, 2))
    cov = b @ b.T          # rank 2, 4 x 4
    mu = np.array([1.0, -2.0, 0.5, 3.0])
    s = dev_sample(hiplib, mu, cov, 200000, seed=11, chain=1)
    assert np.isfinite(s).all()
    assert np.allclose(np.cov(s.T), cov, rtol=0.03, atol=0.03)
    assert np.allclose(s.mean(0), mu, atol=0.02)
    ref = np.zeros(4)
    for i in range(16):
        oracle.oracle_mvnormal_random_dense(11, i, 0, 0, 0, 4, dptr(mu), dptr(np.ascontiguousarray(cov)), 1, dptr(ref))
        assert np.array_equal(s[i], ref)


def test_mfma_f64_accumulation_order(hiplib):
    """v_mfma_f64_16x16x4_f64 is a k-ascending fma chain from C — the definition the dense models' products restate."""
    rng = np.random.default_rng(5)
    for _ in range(4):
        A = rng.normal(size=(16, 4)) * np.exp(rng.normal(size=(16, 4)) * 3)
        B = rng.normal(size=(4, 16)) * np.exp(rng.normal(size=(4, 16)) * 3)
        Cm = rng.normal(size=(16, 16)) * np.exp(rng.normal(size=(16, 16)) * 3)
        D = np.zeros((16, 16))
        assert hiplib.mp_probe_mfma_f64(dptr(A), dptr(B), dptr(Cm), dptr(D), 0) == 0
        want = np.zeros((16, 16))
        for i in range(16):
            for j in range(16):
                acc = Cm[i, j]
                for k in range(4):
                    acc = float(Fraction(A[i, k]) * Fraction(B[k, j]) + Fraction(acc))   # one rounding per step: fma
                want[i, j] = acc
        assert np.array_equal(D, want)


def dense_problem(seed, D=16, singular_q=False):
    rng = np.random.default_rng(seed)
    A = 0.9 * np.eye(D) + 0.08 * rng.normal(size=(D, D)) / np.sqrt(D)
    if singular_q:
        b = rng.normal(size=(D, D - 3)) * 0.3
        Q = b @ b.T
    else:
        Q = spd(D, rng, 0.25)
    R = spd(D, rng, 1.0)
    return A, Q, R


def dense_params(A, Q, R, sig0):
    D = A.shape[0]
    return np.concatenate([[D, sig0], A.reshape(-1), Q.reshape(-1), R.reshape(-1)])


def lockstep_dense(n, T, seed, singular_q=False, resample_every=1):
    import modppl_amd

    A, Q, R = dense_problem(1 + int(singular_q), singular_q=singular_q)
    obs = np.random.default_rng(2).normal(0, 1.5, size=(T, 16))
    pf = modppl_amd.ParticleSystem(modppl_amd.lgssm_dense_model(A, Q, R, 1.0), n, seed)
    ref = O.OraclePF(MP_MODEL_LGSSM_DENSE, 16, 16, dense_params(A, Q, R, 1.0), n, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA, threads=4)
    pf.init_step(None, obs[:1])
    ref.init_step(obs[:1])
    for t in range(1, T):
        assert np.array_equal(pf.log_weights, ref.log_weights()), f"log-weights differ at t={t}"
        assert np.array_equal(pf.states(), ref.state()), f"states differ at t={t}"
        if t % resample_every == 0:
            assert pf.resample() == ref.resample()
            assert np.array_equal(pf.parents, ref.parents())
        pf.step(obs[t:t + 1])
        ref.step(obs[t:t + 1])
    assert pf.log_marginal_likelihood_estimate() == ref.log_marginal_likelihood_estimate()
    return pf


@pytest.mark.parametrize("mfma", ["1", "0"])
def test_dense_lgssm_lockstep_bit_exact(monkeypatch, mfma, diag):
    """d = 16 dense transition: the matrix-core kernel (MP_DENSE_MFMA=1, default) and the scalar handler form (=0) both equal
    the canonical checker bit for bit — states, log-weights, parents, log total weight, log-ML.  3000 particles: a ragged
    last tile; 2048 + 64: a tile with one 64-particle round."""
    monkeypatch.setenv("MP_DENSE_MFMA", mfma)
    lockstep_dense(3000, 5, seed=21)
    lockstep_dense(2048 + 64, 4, seed=22, resample_every=2)


def test_dense_lgssm_singular_process_noise():
    """Q of rank 13: mvnormal.random takes the eigen transform (mvnormal.rs:30-33) inside the filter too."""
    lockstep_dense(1500, 4, seed=23, singular_q=True)


def matrix_kalman_log_ml(A, Q, R, sig0, obs):
    D = A.shape[0]
    m, P, ll = np.zeros(D), sig0 * sig0 * np.eye(D), 0.0
    for t, y in enumerate(obs):
        if t > 0:
            m, P = A @ m, A @ P @ A.T + Q
        S = P + R
        r = y - m
        ll += -0.5 * (D * np.log(2 * np.pi) + np.linalg.slogdet(S)[1] + r @ np.linalg.solve(S, r))
        K = P @ np.linalg.inv(S)
        m, P = m + K @ r, (np.eye(D) - K) @ P
    return ll


def test_dense_lgssm_against_matrix_kalman_filter():
    """size-independent check at 2^19 particles: the SMC log-ML estimate against the exact matrix Kalman filter."""
    import modppl_amd

    A, Q, R = dense_problem(4)
    R = R * 4.0   # a weakly informative observation keeps the bootstrap filter healthy at d = 16
    rng = np.random.default_rng(8)
    x = rng.normal(size=16)
    obs = []
    for t in range(6):
        if t > 0:
            x = A @ x + np.linalg.cholesky(Q) @ rng.normal(size=16)
        obs.append(x + np.linalg.cholesky(R) @ rng.normal(size=16))
    obs = np.array(obs)
    pf = modppl_amd.ParticleSystem(modppl_amd.lgssm_dense_model(A, Q, R, 1.0), 1 << 19, 5)
    pf.init_step(None, obs[:1])
    for t in range(1, len(obs)):
        pf.resample(sync=False)
        pf.step(obs[t:t + 1])
    got, want = pf.log_marginal_likelihood_estimate(), matrix_kalman_log_ml(A, Q, R, 1.0, obs)
    assert abs(got - want) < 0.5, (got, want)
