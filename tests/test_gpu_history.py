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


def test_trajectory_needs_flag():
    import modppl_amd
    from modppl_amd import capi

    pf = modppl_amd.ParticleSystem(modppl_amd.lgssm_model(), 64, 1)
    pf.init_step(None, [0.1])
    with pytest.raises(modppl_amd.ModpplError) as e:
        pf.trajectory(0)
    assert e.value.code == capi.MP_ERR_STATE
