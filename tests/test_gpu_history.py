"""GPU: `traces[i].retv` (Vec<State>, read at modppl/tests/smc.rs:67) rebuilt from the recorded ancestry equals
the trajectories the structure-faithful engine keeps by deep-cloning traces at every resample."""
import numpy as np
import pytest

from tests import oracle_lib as O

pytestmark = pytest.mark.gpu


def test_trajectories_match_cloned_traces():
    import modppl_amd
    from modppl_amd import capi

    ys = O.lgssm_observations(12)
    n, seed = 800, 4
    pf = modppl_amd.ParticleSystem(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), n, seed, flags=capi.MP_PF_RECORD_HISTORY)
    dyn = O.OraclePF(1, 1, 1, O.LGSSM_PARAMS, n, seed, O.VARIANT_CANONICAL)  # ParticleSystem over DynUnfold, traces cloned on resample
    pf.init_step(None, ys[:1])
    dyn.init_step(ys[:1])
    for t in range(1, 12):
        if t % 4 != 0:   # leave some steps without a resample
            pf.resample()
            dyn.resample()
        pf.step(ys[t:t + 1])
        dyn.step(ys[t:t + 1])
    for i in (0, 1, 17, n - 1):
        a, b = pf.trajectory(i), dyn.trajectory(i)
        assert a.shape == (12, 1) and np.array_equal(a, b)
    assert np.array_equal(pf.trajectory(5)[-1], pf.states()[5])
    # every lineage at once (tests/smc.rs:67 walks all particles): one kernel over the pooled event log
    allp = pf.trajectories()
    assert allp.shape == (n, 12, 1)
    for i in range(0, n, 37):
        assert np.array_equal(allp[i], dyn.trajectory(i))
    assert np.array_equal(pf.trajectories(100, 50), allp[100:150])
    assert np.array_equal(allp[:, -1, :], pf.states())


def test_trajectories_wide_state_many_events():
    """d = 4, more events than one history slab holds (the slabs grow geometrically; nothing is allocated per step)."""
    import modppl_amd
    from modppl_amd import capi

    n, seed, T = 3000, 2, 40
    obs = np.random.default_rng(3).normal(0, 1.2, size=(T, 4))
    pf = modppl_amd.ParticleSystem(modppl_amd.lgssm_band_model(4), n, seed, flags=capi.MP_PF_RECORD_HISTORY)
    ref = O.OraclePF(5, 4, 4, np.array([4, 0.9, 0.05, 1.0, 0.5, 1.0]), n, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA)
    pf.init_step(None, obs[:1])
    ref.init_step(obs[:1])
    xs, ps = [ref.state().copy()], []
    for t in range(1, T):
        pf.resample(sync=False)
        ref.resample()
        ps.append(ref.parents().copy())
        pf.step(obs[t:t + 1])
        ref.step(obs[t:t + 1])
        xs.append(ref.state().copy())
    got = pf.trajectories()
    # lineage of particle i from the checker's per-step states and parents
    for i in (0, 1, 999, n - 1):
        a, path = i, []
        for t in range(T - 1, -1, -1):
            path.append(xs[t][a])
            if t > 0:
                a = ps[t - 1][a]
        assert np.array_equal(got[i], np.array(path[::-1]))


def test_trajectory_needs_flag():
    import modppl_amd
    from modppl_amd import capi

    pf = modppl_amd.ParticleSystem(modppl_amd.lgssm_model(), 64, 1)
    pf.init_step(None, [0.1])
    with pytest.raises(modppl_amd.ModpplError) as e:
        pf.trajectory(0)
    assert e.value.code == capi.MP_ERR_STATE
