"""GPU: systematic and stratified resampling, ESS-triggered resampling (extensions; modppl has unconditional multinomial
only, particle_filter.rs:37-41).  Checked against the canonical checker bit for bit and through the properties that
define them."""
import numpy as np
import pytest

from tests import oracle_lib as O

pytestmark = pytest.mark.gpu
SYS, STRAT = 1, 2


@pytest.mark.parametrize("scheme", [SYS, STRAT])
@pytest.mark.parametrize("n", [1000, 4097, 100000, 1 << 20])
def test_systematic_bit_exact_and_properties(n, scheme):
    import modppl_amd

    SYS = scheme   # same lattice; stratified draws one uniform per output slot

    ys = O.lgssm_observations(6)
    seed = 3 + n
    pf = modppl_amd.ParticleSystem(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), n, seed)
    ref = O.OraclePF(1, 1, 1, O.LGSSM_PARAMS, n, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA, threads=8)
    pf.init_step(None, ys[:1])
    ref.init_step(ys[:1])
    for t in range(1, 6):
        w = pf.log_weights
        assert pf.resample(scheme=SYS) == ref.resample(SYS)
        par = pf.parents
        assert np.array_equal(par, ref.parents())
        assert np.array_equal(pf.states(), ref.state())
        assert np.all(np.diff(par.astype(np.int64)) >= 0)            # sorted parents: coalesced gather
        p = np.exp(w - np.logaddexp.reduce(w))
        counts = np.bincount(par, minlength=n)
        assert np.all(np.abs(counts - n * p) < (1.0 if scheme == 1 else 2.0) + 1e-6 * n)   # offspring within 1 (2: stratified) of N w_i
        pf.step(ys[t:t + 1])
        ref.step(ys[t:t + 1])
    assert pf.log_marginal_likelihood_estimate() == ref.log_marginal_likelihood_estimate()


@pytest.mark.parametrize("SYS", [1, 2])
def test_systematic_sharded_world1(SYS):
    import modppl_amd
    from modppl_amd.distributed import ShardedParticleSystem

    ys = O.lgssm_observations(5)
    n, seed = 30000, 8
    a = modppl_amd.ParticleSystem(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), n, seed)
    b = ShardedParticleSystem(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), n, seed)
    a.init_step(None, ys[:1])
    b.init_step(None, ys[:1])
    for t in range(1, 5):
        assert a.resample(scheme=SYS) == b.resample(scheme=SYS)
        assert np.array_equal(a.parents, b.parents) and np.array_equal(a.states(), b.states())
        a.step(ys[t:t + 1])
        b.step(ys[t:t + 1])


@pytest.mark.parametrize("frac", [0.0, 0.5, 1.0])
def test_ess_triggered_resampling(frac):
    """maybe_resample(): the same decisions and results as the checker driven by its own fresh ESS."""
    import modppl_amd

    ys = O.lgssm_observations(12)
    n, seed = 20000, 44
    pf = modppl_amd.ParticleSystem(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), n, seed)
    ref = O.OraclePF(1, 1, 1, O.LGSSM_PARAMS, n, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA, threads=4)
    pf.init_step(None, ys[:1])
    ref.init_step(ys[:1])
    n_res = 0
    for t in range(1, 12):
        did, ess, ltw = pf.maybe_resample(frac)
        ess_ref = ref.effective_sample_size(1)
        assert ess == ess_ref
        assert did == (ess_ref < frac * n)
        if did:
            n_res += 1
            assert ltw == ref.resample()
            assert np.array_equal(pf.parents, ref.parents())
        pf.step(ys[t:t + 1])
        ref.step(ys[t:t + 1])
    assert np.array_equal(pf.log_weights, ref.log_weights())
    assert pf.log_marginal_likelihood_estimate() == ref.log_marginal_likelihood_estimate()
    assert (n_res == 0) if frac == 0.0 else (n_res > 0)


@pytest.mark.parametrize("scheme", [SYS, STRAT])
def test_lattice_schemes_beyond_48k_lds(scheme):
    """2^23 particles = 4096 tiles: the tile table of the single-kernel resample needs more than the default 48 KB of
    dynamic LDS (every instantiation must have asked for it)."""
    import modppl_amd

    n = 1 << 23
    ys = O.lgssm_observations(3)
    pf = modppl_amd.ParticleSystem(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), n, 4)
    pf.init_step(None, ys[:1])
    pf.step(ys[1:2])
    w = pf.log_weights
    pf.resample(scheme=scheme)
    par = pf.parents.astype(np.int64)
    assert par.shape == (n,) and np.all(np.diff(par) >= 0) and par[-1] < n
    p = np.exp(w - np.logaddexp.reduce(w))
    counts = np.bincount(par, minlength=n)
    assert np.all(np.abs(counts - n * p) < (1.0 if scheme == SYS else 2.0) + 1e-6 * n)


@pytest.mark.parametrize("scheme", [0, SYS, STRAT])
@pytest.mark.parametrize("kind", ["band16", "dense16"])
def test_collapsed_weights_long_walks_wide_states(kind, scheme):
    """d = 16 with informative observations: a few hundred particles carry the population, so a guide cell in front of a heavy row holds
    hundreds of light rows and 1 / 1024 of a tile's draws start a walk there.  Under a lattice scheme the wide-state propagate kernels
    finish such walks by bisection (their WALKB instantiation, chosen by the host); the parents must be the checker's — which finds
    them by binary search — for every scheme, and the walks must really have been long."""
    import modppl_amd

    n, T, seed = 3 * 2048 + 77, 7, 6
    if kind == "band16":
        model, okind, params = modppl_amd.lgssm_band_model(16), 5, np.array([16, 0.9, 0.05, 1.0, 0.5, 1.0])
        ref = O.OraclePF(okind, 16, 16, params, n, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA, threads=4)
    else:
        from tests.test_gpu_dense import MP_MODEL_LGSSM_DENSE, dense_params, dense_problem
        A, Q, R = dense_problem(7)
        model = modppl_amd.lgssm_dense_model(A, Q, R, 1.0)
        ref = O.OraclePF(MP_MODEL_LGSSM_DENSE, 16, 16, dense_params(A, Q, R, 1.0), n, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA, threads=4)
    obs = np.random.default_rng(3).normal(0, 1.2, size=(T, 16))
    pf = modppl_amd.ParticleSystem(model, n, seed)
    pf.init_step(None, obs[:1])
    ref.init_step(obs[:1])
    longest = 0
    for t in range(1, T):
        lw = pf.log_weights
        assert np.array_equal(lw, ref.log_weights())
        assert pf.resample(scheme) == ref.resample(scheme)
        par = pf.parents.astype(np.int64)
        assert np.array_equal(par, ref.parents())
        # how far a walk from the start of a guide cell could be: the longest run of rows of one tile whose weights together are
        # below 1 / 1024 of the tile's (what one cell spans)
        w = np.exp(lw - lw.max())
        for b in range(0, n, 2048):
            wt = w[b:b + 2048]
            c = np.cumsum(wt) / wt.sum()
            cell = np.floor(c * 1024).astype(np.int64)
            longest = max(longest, int(np.bincount(cell).max()))
        pf.step(obs[t:t + 1])
        ref.step(obs[t:t + 1])
    assert np.array_equal(pf.states(), ref.state())
    assert longest > 100, longest      # some guide cell held a hundred rows and more: the long-walk case was there
