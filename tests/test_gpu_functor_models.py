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
