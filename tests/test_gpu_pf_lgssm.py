"""GPU parity: the HIP particle filter (through the C ABI) against the CPU checker on the same
seeded inputs.  Bar: bit-exact parents / states / log-weights vs the `canonical` checker;
log-ML within 1e-12 rel of the `literal` checker (libm, sequential fp64 sums) and within the
Monte-Carlo band of the closed-form Kalman value."""
import numpy as np
import pytest

from tests import oracle_lib as O

pytestmark = pytest.mark.gpu


def _mk(n, seed):
    import modppl_amd

    return modppl_amd.ParticleSystem(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), n, seed)


@pytest.mark.parametrize("n", [1, 63, 1000, 4096, 4097, 10000])
def test_lockstep_bit_exact_vs_canonical(n):
    ys = O.lgssm_observations(12)
    seed = 1234 + n
    pf = _mk(n, seed)
    ref = O.OraclePF(1, 1, 1, O.LGSSM_PARAMS, n, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA)
    assert pf.effective_sample_size() == pytest.approx(1.0 / n, rel=0, abs=0)  # stale-ESS quirk before any resample
    pf.init_step(None, ys[:1])
    ref.init_step(ys[:1])
    for t in range(1, len(ys)):
        assert np.array_equal(pf.log_weights, ref.log_weights())
        assert pf.effective_sample_size(fresh=True) == ref.effective_sample_size(1)
        Lg, Lr = pf.resample(), ref.resample()
        assert Lg == Lr
        assert np.array_equal(pf.parents, ref.parents())
        assert np.array_equal(pf.states(), ref.state())
        assert pf.effective_sample_size() == ref.effective_sample_size(0)
        assert np.all(pf.log_weights == 0.0)
        pf.step(ys[t:t + 1])
        ref.step(ys[t:t + 1])
    assert pf.log_marginal_likelihood_estimate() == ref.log_marginal_likelihood_estimate()
    assert pf.time == len(ys)


@pytest.mark.parametrize("scheme", [0, 1])
def test_single_kernel_resampler_mode_lockstep(monkeypatch, scheme, diag):
    """MP_DEFERRED_LOOKUPS=0 (a supported A/B mode: k_resample_gather draws, looks up and clones in one launch) with dim_state 1,
    whose states otherwise live in the row table only (the flag saying so survived that resample, and the next
    step read the PRE-resample rows).  step, resample, step, read_state against the checker, with and without reads in between."""
    monkeypatch.setenv("MP_DEFERRED_LOOKUPS", "0")
    ys = O.lgssm_observations(9)
    n, seed = 5000, 77
    for peek in (False, True):
        pf = _mk(n, seed)
        ref = O.OraclePF(1, 1, 1, O.LGSSM_PARAMS, n, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA)
        pf.init_step(None, ys[:1])
        ref.init_step(ys[:1])
        for t in range(1, len(ys)):
            if peek:
                assert pf.resample(scheme) == ref.resample(scheme)
                assert np.array_equal(pf.states(), ref.state())
                assert np.array_equal(pf.parents, ref.parents())
            else:
                pf.resample(scheme, sync=False)
                ref.resample(scheme)
            pf.step(ys[t:t + 1])
            ref.step(ys[t:t + 1])
            if peek or t % 3 == 0:
                assert np.array_equal(pf.states(), ref.state())
                assert np.array_equal(pf.log_weights, ref.log_weights())
        assert np.array_equal(pf.states(), ref.state())
        assert pf.log_marginal_likelihood_estimate() == ref.log_marginal_likelihood_estimate()


def test_weights_accumulate_without_resample():
    ys = O.lgssm_observations(6)
    n, seed = 2000, 5
    pf = _mk(n, seed)
    ref = O.OraclePF(1, 1, 1, O.LGSSM_PARAMS, n, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA)
    pf.init_step(None, ys[:2])  # two constraints in one call (multi-step extension)
    ref.init_step(ys[:2])
    pf.step(ys[2:5])
    ref.step(ys[2:5])
    assert np.array_equal(pf.log_weights, ref.log_weights())
    assert pf.log_marginal_likelihood_estimate() == ref.log_marginal_likelihood_estimate()
    assert pf.resample() == ref.resample()
    assert np.array_equal(pf.parents, ref.parents())


def test_c1_config_vs_literal_and_kalman():
    """BASELINE config 1 shape (N=1000, T=50) on the GPU vs the literal (libm, fp64-sum) checker."""
    ys = O.lgssm_observations(50)
    n, seed = 1000, 20241008
    pf = _mk(n, seed)
    lit = O.OraclePF(1, 1, 1, O.LGSSM_PARAMS, n, seed, 0)  # structure-faithful dynamic handler, literal arithmetic
    pf.init_step(None, ys[:1])
    lit.init_step(ys[:1])
    mism = 0
    for t in range(1, 50):
        pf.resample()
        lit.resample()
        mism += int((pf.parents != lit.parents()).sum())
        pf.step(ys[t:t + 1])
        lit.step(ys[t:t + 1])
    a, b = pf.log_marginal_likelihood_estimate(), lit.log_marginal_likelihood_estimate()
    assert abs(a - b) <= 1e-12 * abs(b)  # north-star tolerance is 1e-6 rel
    assert mism == 0  # expected ~N*sqrt(N)*eps per step; 0 at this size
    assert abs(a - O.kalman_log_ml(ys)) < 0.5  # N=1000 Monte-Carlo band


def test_full_size_run_and_properties():
    """BASELINE config 2 size (N=2^20, T=50) through mp_pf_run; size-independent properties + Kalman."""
    ys = O.lgssm_observations(50)
    n = 1 << 20
    pf = _mk(n, 7)
    pf.run(None, ys)
    pf.synchronize()
    lml = pf.log_marginal_likelihood_estimate()
    assert abs(lml - O.kalman_log_ml(ys)) < 0.02  # sd of the PF estimate at 2^20 is ~1e-3 per step
    par = pf.parents
    assert par.max() < n
    # multinomial offspring counts: mean 1, the empirical fraction of childless parents ~ E[(1-w_i)^N]
    counts = np.bincount(par, minlength=n)
    assert counts.sum() == n
    assert 0.2 < (counts == 0).mean() < 0.6
    assert np.all(pf.log_weights == 0.0)
    # determinism: a second filter with the same seed reproduces the parents bit for bit
    pf2 = _mk(n, 7)
    pf2.run(None, ys)
    assert np.array_equal(pf2.parents, par)
    assert pf2.log_marginal_likelihood_estimate() == lml


@pytest.mark.parametrize("n", [1 << 20, (1 << 22) + 4097])
def test_full_size_bit_exact_three_steps(n):
    """N=2^20 against the canonical SoA checker for a few steps (a full T=50 takes the CPU ~10 s); from 2^22 the
    resample takes its tile table from the once-per-resample table kernel instead of building it in every workgroup."""
    ys = O.lgssm_observations(4)
    seed = 99
    pf = _mk(n, seed)
    ref = O.OraclePF(1, 1, 1, O.LGSSM_PARAMS, n, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA, threads=8)
    pf.init_step(None, ys[:1])
    ref.init_step(ys[:1])
    for t in range(1, 4):
        assert pf.resample() == ref.resample()
        assert np.array_equal(pf.parents, ref.parents())
        pf.step(ys[t:t + 1])
        ref.step(ys[t:t + 1])
    assert np.array_equal(pf.log_weights, ref.log_weights())
    assert np.array_equal(pf.states(), ref.state())


def test_error_statuses():
    from modppl_amd import ModpplError, capi

    pf = _mk(100, 1)
    with pytest.raises(ModpplError) as e:
        pf.step([0.1])  # step before init_step
    assert e.value.code == capi.MP_ERR_STATE
    with pytest.raises(ModpplError) as e:
        pf.resample()
    assert e.value.code == capi.MP_ERR_STATE
    pf.init_step(None, [0.3])
    with pytest.raises(ModpplError) as e:
        pf.init_step(None, [0.3])
    assert e.value.code == capi.MP_ERR_STATE
    with pytest.raises(ModpplError) as e:
        pf.step([])
    assert e.value.code == capi.MP_ERR_CONSTRAINTS
    # degenerate weights: an observation at +inf gives logpdf = -inf for every particle
    pf.step([np.inf])
    with pytest.raises(ModpplError) as e:
        pf.resample()
    assert e.value.code == capi.MP_ERR_DEGENERATE


def test_states_into_a_caller_buffer():
    """states(out=): the copy lands in the caller's array (a pinned one takes it at the link's rate: bench.py's PCIe-inclusive
    figure); a wrong shape or type is refused before anything is copied"""
    import torch
    import modppl_amd
    from modppl_amd.capi import ModpplError

    n = 70001
    ys = O.lgssm_observations(3)
    pf = modppl_amd.ParticleSystem(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), n, 5)
    pf.init_step(None, ys[:1])
    pf.resample()
    pf.step(ys[1:2])
    want = pf.states()
    pinned = torch.empty((n, 1), dtype=torch.float64).pin_memory().numpy()
    got = pf.states(out=pinned)
    assert got is pinned and np.array_equal(got, want)
    plain = np.zeros((n, 1))
    assert np.array_equal(pf.states(out=plain), want)
    for bad in (np.zeros((n, 1), dtype=np.float32), np.zeros((n + 1, 1)), np.zeros((2 * n, 1))[::2], [0.0] * n):
        with pytest.raises(ModpplError):
            pf.states(out=bad)


@pytest.mark.parametrize("d", [1, 4])
def test_parents_survive_a_step_after_a_lazy_resample(d):
    """particle_filter.rs:73-96 keeps `parents` across `step`; here a resample that only drew leaves them as {target, start row} per slot and the
    next propagate consumes the states from there — the parents must still be the last resample's when asked for later."""
    import modppl_amd

    n, seed, T = 5000, 4, 5
    if d == 1:
        model, obs, kind, params = modppl_amd.lgssm_model(*O.LGSSM_PARAMS), O.lgssm_observations(T).reshape(T, 1), 1, O.LGSSM_PARAMS
    else:
        model, obs = modppl_amd.lgssm_band_model(d), np.random.default_rng(3).normal(0, 1.2, size=(T, d))
        kind, params = 5, np.array([d, 0.9, 0.05, 1.0, 0.5, 1.0])
    pf = modppl_amd.ParticleSystem(model, n, seed)
    ref = O.OraclePF(kind, d, d, params, n, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA)
    pf.init_step(None, obs[:1])
    ref.init_step(obs[:1])
    for t in range(1, T):
        pf.resample(sync=False)          # nothing materialised
        ref.resample()
        want = ref.parents().copy()
        pf.step(obs[t:t + 1])            # consumes the segment-ordered states
        ref.step(obs[t:t + 1])
        if t % 2:
            pf.step(obs[t:t + 1])        # and a second step without a resample in between
            ref.step(obs[t:t + 1])
        assert np.array_equal(pf.parents, want), f"parents went stale at t={t}"
        assert np.array_equal(pf.states(), ref.state())
