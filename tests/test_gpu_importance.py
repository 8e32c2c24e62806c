"""GPU parity of importance_sampling / importance_resampling (modppl/src/inference/importance.rs:12-50)."""
import numpy as np
import pytest

from tests import oracle_lib as O

pytestmark = pytest.mark.gpu


def test_importance_resampling_lgssm_bit_exact():
    import modppl_amd

    ys = O.lgssm_observations(8)
    n, m, seed = 10000, 100, 5   # tests/importance.rs uses 10 000 samples and sqrt(N) resampled traces
    states, idx, lml = modppl_amd.importance_resampling(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), None, ys, n, m, seed)
    st2, lnw, lml2 = modppl_amd.importance_sampling(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), None, ys, n, seed)
    r_lml, r_lnw, r_idx, r_xs = O.importance_resampling(1, 1, 1, O.LGSSM_PARAMS, ys, n, m, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA)
    assert lml == r_lml and lml2 == r_lml
    assert np.array_equal(idx, r_idx)
    assert np.array_equal(states, r_xs) and np.array_equal(st2, r_xs)
    assert np.array_equal(lnw, r_lnw)
    assert abs(np.logaddexp.reduce(lnw)) < 1e-12  # normalised
    # structure-faithful generic importance_resampling over the dynamic DynUnfold, literal arithmetic (libm, fp64 CDF)
    d_lml, d_lnw, d_idx, d_xs = O.importance_resampling(1, 1, 1, O.LGSSM_PARAMS, ys, 2000, 50, seed, 0)
    s3, i3, l3 = modppl_amd.importance_resampling(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), None, ys, 2000, 50, seed)
    assert abs(l3 - d_lml) <= 1e-12 * abs(d_lml)
    assert np.array_equal(i3, d_idx)
    assert np.allclose(s3, d_xs, rtol=0, atol=1e-14)
    # importance sampling of a T=8 LGSSM: the estimate is unbiased for exp(log-ML); loose Kalman band
    assert abs(lml - O.kalman_log_ml(ys)) < 1.0


def test_importance_resampling_spiral():
    import modppl_amd
    from tests.test_gpu_pf_models import spiral_obs

    obs = spiral_obs(20)[:3]
    n, m, seed = 4096, 64, 9
    states, idx, lml = modppl_amd.importance_resampling(modppl_amd.spiral_model(), [0.0, 0.0], obs, n, m, seed)
    r_lml, r_lnw, r_idx, r_xs = O.importance_resampling(2, 2, 2, np.zeros(0), obs, n, m, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA, args0=[0.0, 0.0])
    assert lml == r_lml and np.array_equal(idx, r_idx) and np.array_equal(states, r_xs)


def test_importance_one_million():
    """tests/importance.rs:90-92: 'works with ~1,000,000 particles'."""
    import modppl_amd

    ys = O.lgssm_observations(5)
    n = 1 << 20
    states, idx, lml = modppl_amd.importance_resampling(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), None, ys, n, 1024, 3)
    assert abs(lml - O.kalman_log_ml(ys)) < 0.05
    assert idx.max() < n and states.shape == (n, 1)
