"""GPU parity of GenFn::simulate over the Unfold models (DynUnfold::simulate, modppl/src/modeling/dynunfold.rs:22-39): states
and sampled observations bit-exact against the trie-addressed restatement in canonical arithmetic."""
import numpy as np
import pytest

from tests import oracle_lib as O

pytestmark = pytest.mark.gpu


def cases():
    import modppl_amd

    return [
        ("lgssm", modppl_amd.lgssm_model(*O.LGSSM_PARAMS), 1, 1, 1, O.LGSSM_PARAMS, None, 6),
        ("spiral", modppl_amd.spiral_model(), 2, 2, 2, np.zeros(0), [0.0, 0.0], 5),
        ("bearings", modppl_amd.bearings_model(), 4, 4, 1, np.array([1.0, 1.0, 1.0, 0.1, 0.05, 0.02]), None, 5),
        ("band4", modppl_amd.lgssm_band_model(4), 5, 4, 4, np.array([4, 0.9, 0.05, 1.0, 0.5, 1.0]), None, 4),
        ("pointed", modppl_amd.pointed_2d_model(), 6, 2, 2, np.array([-5.0, 5.0, -5.0, 5.0, 1.0, -0.6, -0.6, 2.0]), [0.0, 0.0], 1),
        ("line", modppl_amd.line_model(), 7, 2, 11, np.arange(-5.0, 6.0), [0.0, 0.0], 1),
    ]


def test_simulate_bit_exact_all_models():
    import modppl_amd

    n, seed = 3000, 17
    for name, model, kind, ds, do, params, args0, T in cases():
        xs, ys = modppl_amd.simulate(model, args0, T, n, seed)
        rx, ry = O.unfold_simulate(kind, ds, do, params, T, n, seed, args0=args0)
        assert np.array_equal(xs, rx), name
        assert np.array_equal(ys, ry), name


def test_simulate_then_filter_roundtrip():
    """Data simulated on the device from the LGSSM, then filtered on the device: the log-ML estimate sits on the exact
    Kalman value of those observations (2^18 particles)."""
    import bench as B
    import modppl_amd

    xs, ys = modppl_amd.simulate(modppl_amd.lgssm_model(*B.LGSSM_PARAMS), None, 30, 4, 99)
    obs = ys[2, :, 0]
    pf = modppl_amd.ParticleSystem(modppl_amd.lgssm_model(*B.LGSSM_PARAMS), 1 << 18, 5)
    pf.run(None, obs)
    assert abs(pf.log_marginal_likelihood_estimate() - B.kalman_log_ml(obs)) < 0.05


def test_simulate_statuses():
    import modppl_amd
    from modppl_amd import ModpplError, capi

    with pytest.raises(ModpplError) as e:
        modppl_amd.simulate(modppl_amd.hmm_model([0.5, 0.5], [[0.9, 0.2], [0.1, 0.8]], [[0.7, 0.3], [0.3, 0.7]]), None, 3, 10, 1)
    assert e.value.code == capi.MP_ERR_UNSUPPORTED
    with pytest.raises(ModpplError) as e:
        modppl_amd.simulate(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), None, 0, 10, 1)
    assert e.value.code == capi.MP_ERR_INVALID_ARG
