"""GPU parity for the other models on the path, each in lockstep with the canonical CPU checker
(bit-exact parents / states / log-weights / log total weight) and cross-checked against the
structure-faithful dynamic-handler engine at a small size."""
import math

import numpy as np
import pytest

from tests import oracle_lib as O
from tests.test_oracle_kats import HMM3, hmm3_params

pytestmark = pytest.mark.gpu


def lockstep(model, kind, params, n, seed, obs, args0=None, resample_every=1):
    import modppl_amd

    ds, do = model.dim_state, model.dim_obs
    pf = modppl_amd.ParticleSystem(model, n, seed)
    ref = O.OraclePF(kind, ds, do, params, n, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA, threads=4)
    pf.init_step(args0, obs[:1])
    ref.init_step(obs[:1], args0)
    for t in range(1, len(obs)):
        assert np.array_equal(pf.log_weights, ref.log_weights()), f"log-weights differ at t={t}"
        assert np.array_equal(pf.states(), ref.state()), f"states differ at t={t}"
        if t % resample_every == 0:
            assert pf.resample() == ref.resample()
            assert np.array_equal(pf.parents, ref.parents())
            assert np.array_equal(pf.states(), ref.state())
        pf.step(obs[t:t + 1])
        ref.step(obs[t:t + 1])
    a, b = pf.log_marginal_likelihood_estimate(), ref.log_marginal_likelihood_estimate()
    assert a == b
    return pf, ref, a


def spiral_obs(T):
    ang = 0.7
    return np.array([[0.4 * math.cos(2 * math.pi * t / T + ang), 0.4 * math.sin(2 * math.pi * t / T + ang)] for t in range(T)])


def test_spiral_model_smc_rs_shape():
    """modppl/tests/smc.rs:48-91: spiral Unfold, 500 particles x 20 steps, resample every step."""
    import modppl_amd

    obs = spiral_obs(20)
    pf, ref, lml = lockstep(modppl_amd.spiral_model(), 2, np.zeros(0), 500, 3, obs, args0=[0.0, 0.0])
    assert np.isfinite(lml)
    # and against the structure-faithful engine (dynamic handler, per-call LU determinant/inverse) in canonical arithmetic
    dyn = O.OraclePF(2, 2, 2, np.zeros(0), 500, 3, O.VARIANT_CANONICAL)
    dyn.init_step(obs[:1], [0.0, 0.0])
    for t in range(1, 20):
        dyn.resample()
        dyn.step(obs[t:t + 1])
    assert dyn.log_marginal_likelihood_estimate() == lml
    assert np.array_equal(dyn.state(), pf.states())


def test_spiral_larger_and_sparse_resampling():
    import modppl_amd

    lockstep(modppl_amd.spiral_model(), 2, np.zeros(0), 20000, 11, spiral_obs(12), args0=[0.0, 0.0], resample_every=3)


def test_hmm_reference_e2e_on_gpu():
    """modppl/tests/particle_filter.rs:35-79 on the GPU: 10 000 particles, obs [0,0,1,2], |lml - ln forward| <= 0.03."""
    import modppl_amd

    params = hmm3_params()
    data = np.array([0.0, 0.0, 1.0, 2.0])
    L = O.load()
    expected = math.log(L.oracle_hmm_forward(O.dptr(params), len(params), O.dptr(data), 4))
    model = modppl_amd.hmm_model(HMM3["prior"], HMM3["emis"], HMM3["trans"])
    pf = modppl_amd.ParticleSystem(model, 10000, 42)
    ref = O.OraclePF(3, 1, 1, params, 10000, 42, O.VARIANT_CANONICAL | O.VARIANT_SOA)
    pf.init_step(None, data[:1])
    ref.init_step(data[:1])
    for t in range(1, 4):  # the reference test's loop: step, ESS, resample
        pf.step(data[t:t + 1])
        ref.step(data[t:t + 1])
        assert pf.effective_sample_size() == ref.effective_sample_size()
        assert pf.resample() == ref.resample()
        assert np.array_equal(pf.parents, ref.parents())
        assert np.array_equal(pf.states(), ref.state())
    lml = pf.log_marginal_likelihood_estimate()
    assert lml == ref.log_marginal_likelihood_estimate()
    assert abs(lml - expected) <= 0.03


def test_hmm_rejects_unnormalised_tables():
    import modppl_amd
    from modppl_amd import capi

    bad = modppl_amd.hmm_model([0.5, 0.6], np.eye(2), np.eye(2))
    with pytest.raises(modppl_amd.ModpplError) as e:
        modppl_amd.ParticleSystem(bad, 10, 1)
    assert e.value.code == capi.MP_ERR_INVALID_ARG


BEAR = np.array([1.0, 1.0, 1.0, 0.1, 0.05, 0.02])


def bearings_obs(T, seed=5):
    rng = np.random.default_rng(seed)
    p = np.array([1.2, 0.8]); v = np.array([0.05, 0.08])
    out = []
    for _ in range(T):
        out.append([math.atan2(p[1], p[0]) + 0.02 * rng.normal()])
        a = 0.05 * rng.normal(size=2)
        p = p + v + 0.5 * a
        v = v + a
    return np.array(out)


def test_bearings_parity_4096():
    """BASELINE config 3 model; SURVEY §8d: parity vs the canonical oracle at N = 4096."""
    import modppl_amd

    obs = bearings_obs(25)
    pf, ref, lml = lockstep(modppl_amd.bearings_model(*BEAR), 4, BEAR, 4096, 17, obs)
    assert np.isfinite(lml)
    dyn = O.OraclePF(4, 4, 1, BEAR, 4096, 17, O.VARIANT_CANONICAL)
    dyn.init_step(obs[:1])
    for t in range(1, 25):
        dyn.resample()
        dyn.step(obs[t:t + 1])
    assert dyn.log_marginal_likelihood_estimate() == lml


def test_bearings_full_size_properties():
    """BASELINE config 3 size: 4M particles, T = 50, through mp_pf_run; properties only."""
    import modppl_amd

    obs = bearings_obs(50)
    n = 1 << 22
    pf = modppl_amd.ParticleSystem(modppl_amd.bearings_model(*BEAR), n, 23)
    pf.run(None, obs)
    lml = pf.log_marginal_likelihood_estimate()
    assert np.isfinite(lml)
    x = pf.states()
    assert x.shape == (n, 4) and np.all(np.isfinite(x))
    est = math.atan2(x[:, 1].mean(), x[:, 0].mean())
    assert abs(est - obs[-1, 0]) < 0.1  # the posterior bearing tracks the last observation
    par = pf.parents
    assert par.max() < n and np.all(pf.log_weights == 0.0)
    pf2 = modppl_amd.ParticleSystem(modppl_amd.bearings_model(*BEAR), n, 23)
    pf2.run(None, obs)
    assert np.array_equal(pf2.parents, par) and pf2.log_marginal_likelihood_estimate() == lml


@pytest.mark.parametrize("D", [2, 4, 16])
def test_lgssm_band_parity(D):
    """BASELINE config 5 model (d=16) and smaller bands."""
    import modppl_amd

    params = np.array([D, 0.9, 0.05, 1.0, 0.5, 1.0])
    rng = np.random.default_rng(D)
    obs = rng.normal(0, 1.2, size=(10, D))
    n = 3000 if D == 16 else 5000
    lockstep(modppl_amd.lgssm_band_model(D), 5, params, n, 31 + D, obs)
    if D == 2:
        dyn = O.OraclePF(5, D, D, params, 400, 7, O.VARIANT_CANONICAL)
        soa = O.OraclePF(5, D, D, params, 400, 7, O.VARIANT_CANONICAL | O.VARIANT_SOA)
        for e in (dyn, soa):
            e.init_step(obs[:1])
            for t in range(1, 6):
                e.resample()
                e.step(obs[t:t + 1])
        assert dyn.log_marginal_likelihood_estimate() == soa.log_marginal_likelihood_estimate()
        assert np.array_equal(dyn.state(), soa.state())


def band_kalman_log_ml(obs, D, a, band, sig0, sig_x, sig_y):
    """Exact log marginal likelihood of the banded LGSSM (matrix Kalman filter, numpy): x0 ~ N(0, sig0^2 I),
    x' = A x + sig_x z with A = a (I + band (shift up + shift down)), y = x + sig_y e."""
    A = a * (np.eye(D) + band * (np.eye(D, k=1) + np.eye(D, k=-1)))
    Q, R = sig_x ** 2 * np.eye(D), sig_y ** 2 * np.eye(D)
    m, P = np.zeros(D), sig0 ** 2 * np.eye(D)
    ll = 0.0
    for t, y in enumerate(obs):
        if t > 0:
            m, P = A @ m, A @ P @ A.T + Q
        S = P + R
        r = y - m
        ll += -0.5 * (D * math.log(2 * math.pi) + np.linalg.slogdet(S)[1] + r @ np.linalg.solve(S, r))
        K = P @ np.linalg.inv(S)
        m, P = m + K @ r, (np.eye(D) - K) @ P
    return ll


def test_lgssm_band16_c5_shard_size_properties():
    """BASELINE config 5, one GPU's share (2^21 particles, d = 16) through mp_pf_run: finite, reproducible, parents in
    range, and the log-ML estimate against the exact matrix Kalman filter (observations simulated from the model, so the
    bootstrap filter keeps a usable ESS in 16 dimensions; tolerance 0.5 nats on a value of a few hundred)."""
    import modppl_amd

    D, T, n = 16, 12, 1 << 21
    a, band, sig0, sig_x, sig_y = 0.9, 0.05, 1.0, 0.5, 1.0
    rng = np.random.default_rng(16)
    A = a * (np.eye(D) + band * (np.eye(D, k=1) + np.eye(D, k=-1)))
    x = sig0 * rng.normal(size=D)
    obs = []
    for t in range(T):
        if t > 0:
            x = A @ x + sig_x * rng.normal(size=D)
        obs.append(x + sig_y * rng.normal(size=D))
    obs = np.array(obs)
    pf = modppl_amd.ParticleSystem(modppl_amd.lgssm_band_model(D, a, band, sig0, sig_x, sig_y), n, 5)
    pf.run(None, obs)
    lml = pf.log_marginal_likelihood_estimate()
    exact = band_kalman_log_ml(obs, D, a, band, sig0, sig_x, sig_y)
    assert np.isfinite(lml) and abs(lml - exact) < 0.5, (lml, exact)
    xs = pf.states()
    assert xs.shape == (n, D) and np.all(np.isfinite(xs))
    par = pf.parents
    assert par.max() < n and np.all(pf.log_weights == 0.0)
    # traces[i] = traces[parents[i]]: after the final resample every state row is one of the pre-resample rows
    pf2 = modppl_amd.ParticleSystem(modppl_amd.lgssm_band_model(D, a, band, sig0, sig_x, sig_y), n, 5)
    pf2.run(None, obs)
    assert np.array_equal(pf2.parents, par) and pf2.log_marginal_likelihood_estimate() == lml and np.array_equal(pf2.states(), xs)


def test_hmm_impossible_emissions_minus_inf_weights():
    """Erasures of this domain: particles whose log-weight is -inf (an emission the particle's state cannot produce).
    They must never be chosen as parents, the ESS counts only the live ones, and everything stays bit-exact."""
    import modppl_amd

    prior = [0.5, 0.5]
    emis = [[1.0, 0.2], [0.0, 0.8]]          # emission[o][s]: state 0 never emits symbol 1
    trans = [[0.9, 0.3], [0.1, 0.7]]
    params = np.concatenate([[2, 2], prior, np.array(emis).reshape(-1), np.array(trans).reshape(-1)])
    data = np.array([1.0, 0.0, 1.0, 1.0, 0.0])
    n = 6000
    pf = modppl_amd.ParticleSystem(modppl_amd.hmm_model(prior, emis, trans), n, 7)
    ref = O.OraclePF(3, 1, 1, params, n, 7, O.VARIANT_CANONICAL | O.VARIANT_SOA)
    pf.init_step(None, data[:1])
    ref.init_step(data[:1])
    for t in range(1, len(data)):
        w, x = pf.log_weights, pf.states()[:, 0]
        dead = np.isneginf(w)
        if data[t - 1] == 1.0:
            assert dead.any() and np.array_equal(dead, x == 0.0)      # exactly the particles in state 0
        assert pf.effective_sample_size(fresh=True) == ref.effective_sample_size(1)
        assert pf.effective_sample_size(fresh=True) <= (~dead).sum() + 1e-9
        assert pf.resample() == ref.resample()
        par = pf.parents
        assert np.array_equal(par, ref.parents())
        assert not dead[par].any()                                     # a dead particle has no offspring
        pf.step(data[t:t + 1])
        ref.step(data[t:t + 1])
    assert pf.log_marginal_likelihood_estimate() == ref.log_marginal_likelihood_estimate()


def test_maximum_job_size_and_beyond():
    """2^24 particles (the tile table's limit) run; one more tile is refused with a status, not a fault."""
    import modppl_amd
    from modppl_amd import ModpplError, capi

    n = 1 << 24
    ys = O.lgssm_observations(3)
    pf = modppl_amd.ParticleSystem(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), n, 5)
    pf.init_step(None, ys[:1])
    L0 = pf.resample()
    pf.step(ys[1:2])
    assert pf.effective_sample_size(fresh=True) > 0.3 * n
    pf.resample()
    par = pf.parents
    assert par.shape == (n,) and int(par.max()) < n
    assert np.isfinite(L0) and abs(pf.log_marginal_likelihood_estimate() - O.kalman_log_ml(ys[:2])) < 0.01
    del pf
    with pytest.raises(ModpplError) as e:
        modppl_amd.ParticleSystem(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), n + 1, 5)
    assert e.value.code == capi.MP_ERR_UNSUPPORTED
