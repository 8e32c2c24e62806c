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


def test_importance_reference_models():
    """The two models of the reference's own importance tests, 10 000 samples and 1 000 resampled traces as there:
    pointed_2d_model (tests/importance.rs:17-50: uniform_2d prior, dense mvnormal likelihood) and line_model
    (:54-76: two normal priors, 11 observations through the obs_model sub-call).  Bit-exact against the SoA checker
    and, at a smaller size, against the generic importance_resampling over the trie-addressed restatement."""
    import modppl_amd

    n, m, seed = 10000, 1000, 12
    bounds, cov = [-5.0, 5.0, -5.0, 5.0], [1.0, -0.6, -0.6, 2.0]
    obs = np.array([[0.0, 0.0]])
    pparams = np.array(bounds + cov)
    states, idx, lml = modppl_amd.importance_resampling(modppl_amd.pointed_2d_model(bounds, cov), [0.0, 0.0], obs, n, m, seed)
    r_lml, r_lnw, r_idx, r_xs = O.importance_resampling(6, 2, 2, pparams, obs, n, m, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA, args0=[0.0, 0.0])
    assert lml == r_lml and np.array_equal(idx, r_idx) and np.array_equal(states, r_xs)
    assert np.all(np.abs(states) <= 5.0)
    # evidence of obs = 0 under a flat prior on the 10 x 10 box: integral of the mvnormal density / 100 ~ 1/100
    assert abs(lml - np.log(1.0 / 100.0)) < 0.05
    post = states[idx]
    assert np.allclose(np.cov(post.T), np.array(cov).reshape(2, 2), atol=0.35)
    d_lml, d_lnw, d_idx, d_xs = O.importance_resampling(6, 2, 2, pparams, obs, 1500, 40, seed, O.VARIANT_CANONICAL, args0=[0.0, 0.0])
    s3, i3, l3 = modppl_amd.importance_resampling(modppl_amd.pointed_2d_model(bounds, cov), [0.0, 0.0], obs, 1500, 40, seed)
    assert l3 == d_lml and np.array_equal(i3, d_idx) and np.array_equal(s3, d_xs)

    xs = np.arange(-5.0, 6.0)
    rng = np.random.default_rng(0)
    ys = (0.5 * xs - 1.0 + 0.1 * rng.normal(size=xs.size)).reshape(1, -1)   # tests/importance.rs:66-67
    states, idx, lml = modppl_amd.importance_resampling(modppl_amd.line_model(xs), [0.0, 0.0], ys, n, m, seed)
    r_lml, r_lnw, r_idx, r_xs = O.importance_resampling(7, 2, 11, xs, ys, n, m, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA, args0=[0.0, 0.0])
    assert lml == r_lml and np.array_equal(idx, r_idx) and np.array_equal(states, r_xs)
    best = states[idx[0]]
    assert abs(best[0] - 0.5) < 0.2 and abs(best[1] + 1.0) < 0.5   # the resampled traces sit near slope 0.5, intercept -1
    d_lml, d_lnw, d_idx, d_xs = O.importance_resampling(7, 2, 11, xs, ys, 1500, 40, seed, O.VARIANT_CANONICAL, args0=[0.0, 0.0])
    s3, i3, l3 = modppl_amd.importance_resampling(modppl_amd.line_model(xs), [0.0, 0.0], ys, 1500, 40, seed)
    assert l3 == d_lml and np.array_equal(i3, d_idx) and np.array_equal(s3, d_xs)


def test_importance_sampling_returns_every_trace():
    """importance_sampling returns ALL N traces (importance.rs:26-27): every sample's choices at every step, against the checker's
    per-particle trajectories (the structure-faithful engine keeps each trace's `retv`), for a scalar and a wider model."""
    import modppl_amd
    from tests.test_gpu_pf_models import spiral_obs

    ys = O.lgssm_observations(6)
    n, seed = 5000, 5
    traj, lnw, lml = modppl_amd.importance_sampling(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), None, ys, n, seed, full_traces=True)
    st, lnw2, lml2 = modppl_amd.importance_sampling(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), None, ys, n, seed)
    assert traj.shape == (n, 6, 1)
    assert lml == lml2 and np.array_equal(lnw, lnw2)
    assert np.array_equal(traj[:, -1, :], st)          # retv.last()
    ref = O.OraclePF(1, 1, 1, O.LGSSM_PARAMS, n, seed, O.VARIANT_CANONICAL)   # structure-faithful: traces with their retv
    ref.init_step(ys)
    for i in (0, 1, 777, n - 1):
        assert np.array_equal(traj[i], np.asarray(ref.trajectory(i)).reshape(6, 1))
    obs = spiral_obs(20)[:4]
    traj, lnw, lml = modppl_amd.importance_sampling(modppl_amd.spiral_model(), [0.0, 0.0], obs, 3000, 9, full_traces=True)
    ref = O.OraclePF(2, 2, 2, np.zeros(0), 3000, 9, O.VARIANT_CANONICAL)
    ref.init_step(obs, args0=[0.0, 0.0])
    for i in (0, 2999, 1234):
        assert np.array_equal(traj[i], np.asarray(ref.trajectory(i)).reshape(4, 2))
