"""GPU: a model added through the registration layer (one block of modppl_amd/csrc/mp_models_extra.h, kind 100: stochastic
volatility) gets Generate and Simulate parity against the CPU checker with no restatement written for it — the checker interprets
the same functor with its own handlers and distributions (oracle/src/functor_adapter.hpp, cross-checked in
tests/test_oracle_functor.py)."""
import numpy as np
import pytest

from tests import oracle_lib as O
from tests.test_oracle_functor import SV, sv_observations

pytestmark = pytest.mark.gpu


def test_registered_model_filter_lockstep_bit_exact():
    import modppl_amd

    n, seed, T = 5000, 12, 8
    obs = sv_observations(T)
    model = modppl_amd.stochastic_volatility_model(*SV)
    pf = modppl_amd.ParticleSystem(model, n, seed)
    ref = O.OraclePF(100, 1, 1, SV, n, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA)
    pf.init_step(None, obs[:1])
    ref.init_step(obs[:1])
    for t in range(1, T):
        assert np.array_equal(pf.log_weights, ref.log_weights())
        assert np.array_equal(pf.states(), ref.state())
        assert pf.resample() == ref.resample()
        assert np.array_equal(pf.parents, ref.parents())
        pf.step(obs[t:t + 1])
        ref.step(obs[t:t + 1])
    assert pf.log_marginal_likelihood_estimate() == ref.log_marginal_likelihood_estimate()
    # and the dynamic trie engine (one sample_at per site) says the same
    dyn = O.OraclePF(100, 1, 1, SV, n, seed, O.VARIANT_CANONICAL)
    dyn.init_step(obs[:1])
    for t in range(1, T):
        dyn.resample()
        dyn.step(obs[t:t + 1])
    assert np.array_equal(dyn.state(), pf.states())


def test_registered_model_simulate_bit_exact():
    import modppl_amd

    xs, ys = modppl_amd.simulate(modppl_amd.stochastic_volatility_model(*SV), None, 6, 3000, 17)
    rx, ry = O.unfold_simulate(100, 1, 1, SV, 6, 3000, 17)
    assert np.array_equal(xs, rx) and np.array_equal(ys, ry)


def test_registered_model_full_size_sanity():
    """2^20 particles: finite log-ML, healthy ESS, offspring counts with the multinomial law's mean"""
    import modppl_amd

    n, T = 1 << 20, 6
    obs = sv_observations(T)
    pf = modppl_amd.ParticleSystem(modppl_amd.stochastic_volatility_model(*SV), n, 3)
    pf.init_step(None, obs[:1])
    for t in range(1, T):
        pf.resample(sync=False)
        pf.step(obs[t:t + 1])
    assert np.isfinite(pf.log_marginal_likelihood_estimate())
    ess = pf.effective_sample_size(fresh=True)
    assert 0.05 * n < ess <= n


def test_stochastic_volatility_against_a_grid_filter():
    """An EXTERNAL check of the registered example: the checker interprets the product's own functor, so a wrong line inside
    mp_stochvol::operator() (the AR(1) mean, exp(h / 2) as the observation's standard deviation) would pass every test above.
    Here the model is written down independently — numpy, straight from its definition — as a deterministic grid filter
    (the latent h on 3001 points, trapezoid weights): its log marginal likelihood must agree with the particle estimate at
    2^20 particles within Monte Carlo error."""
    import modppl_amd

    mu, phi, sigma, sig0 = SV
    T = 8
    obs = sv_observations(T)
    # the grid filter
    sd_stat = max(sig0, sigma / np.sqrt(max(1e-12, 1.0 - phi * phi)))
    hs = np.linspace(mu - 9.0 * sd_stat, mu + 9.0 * sd_stat, 3001)
    dh = hs[1] - hs[0]
    wq = np.full(hs.size, dh); wq[0] = wq[-1] = dh / 2

    def npdf(x, m, s):
        return np.exp(-0.5 * ((x - m) / s) ** 2) / (s * np.sqrt(2.0 * np.pi))

    trans = npdf(hs[None, :], mu + phi * (hs[:, None] - mu), sigma)     # p(h' | h), rows = h
    pred = npdf(hs, mu, sig0)
    log_ml = 0.0
    for t in range(T):
        if t > 0:
            pred = (post * wq) @ trans
        like = npdf(obs[t, 0], 0.0, np.exp(hs / 2.0))
        z = float(np.sum(pred * like * wq))
        log_ml += np.log(z)
        post = pred * like / z
    # the particle filter
    n = 1 << 20
    pf = modppl_amd.ParticleSystem(modppl_amd.stochastic_volatility_model(*SV), n, 20241008)
    pf.init_step(None, obs[:1])
    for t in range(1, T):
        pf.resample(sync=False)
        pf.step(obs[t:t + 1])
    est = pf.log_marginal_likelihood_estimate()
    assert abs(est - log_ml) < 0.01, (est, log_ml)
