"""One model source: a functor of modppl_amd/csrc/mp_models.h registered with MP_REGISTER_UNFOLD_MODEL is run by
the CPU checker through its OWN interpreters (oracle/src/functor_adapter.hpp: the flat-array engine and the dynamic trie handler,
dists.hpp's distributions, rng.hpp's Philox) — no hand-written restatement.  Here: the adapter is itself cross-checked, by running
the product's functors of models that the checker ALSO restates by hand (test-only kinds 1000 + k) against those restatements."""
import numpy as np
import pytest

from tests import oracle_lib as O

def _spiral_obs(T):
    t = np.arange(T)
    r, th = 0.5 + 0.02 * t, 0.3 + 0.4 * t
    return np.stack([r * np.cos(th), r * np.sin(th)], axis=1) + 0.01 * np.random.default_rng(2).normal(size=(T, 2))


CASES = [
    ("lgssm1", 1, 1, 1, O.LGSSM_PARAMS, lambda T: O.lgssm_observations(T).reshape(T, 1)),
    # uniform sites at t = 0, normal sites after, and a vector-valued mvnormal site (mvnormal.rs:12-37) constrained with a Vec
    ("spiral", 2, 2, 2, np.zeros(0), _spiral_obs),
    ("bearings", 4, 4, 1, np.array([1.0, 1.0, 1.0, 0.1, 0.05, 0.02]),
     lambda T: (np.arctan2(1.0 + 0.05 * np.arange(T), 1.0 + 0.1 * np.arange(T)) + 0.01 * np.sin(np.arange(T))).reshape(T, 1)),
    ("band4", 5, 4, 4, np.array([4, 0.9, 0.05, 1.0, 0.5, 1.0]), lambda T: np.random.default_rng(3).normal(0, 1.2, size=(T, 4))),
]


def run(kind, ds, do, params, n, seed, variant, obs):
    pf = O.OraclePF(kind, ds, do, params, n, seed, variant)
    pf.init_step(obs[:1])
    out = []
    for t in range(1, len(obs)):
        L = pf.resample()
        out.append((L, pf.parents().copy()))
        pf.step(obs[t:t + 1])
    return out, pf.state().copy(), pf.log_weights().copy(), pf.log_marginal_likelihood_estimate()


@pytest.mark.parametrize("name,kind,ds,do,params,mk_obs", CASES)
@pytest.mark.parametrize("variant", [O.VARIANT_SOA, O.VARIANT_SOA | O.VARIANT_CANONICAL, O.VARIANT_FAST_SEARCH, O.VARIANT_CANONICAL])
def test_functor_adapter_equals_hand_written_restatement(name, kind, ds, do, params, mk_obs, variant):
    """flat-array engine and dynamic trie engine, literal and canonical arithmetic: the functor run through the adapter gives the
    same states, log-weights, parents, log total weights and log-ML as the hand-written restatement of the same model"""
    T = 5
    n = 400 if not (variant & O.VARIANT_SOA) else 3000
    obs = mk_obs(T)
    a = run(kind, ds, do, params, n, 5, variant, obs)
    b = run(1000 + kind, ds, do, params, n, 5, variant, obs)
    for (La, pa), (Lb, pb) in zip(a[0], b[0]):
        assert La == Lb and np.array_equal(pa, pb), name
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3] == b[3], name


SV = np.array([-1.0, 0.95, 0.25, 0.8])


def sv_observations(T, seed=4):
    rng = np.random.default_rng(seed)
    h, ys = SV[0] + SV[3] * rng.normal(), []
    for t in range(T):
        if t:
            h = SV[0] + SV[1] * (h - SV[0]) + SV[2] * rng.normal()
        ys.append(np.exp(h / 2) * rng.normal())
    return np.array(ys).reshape(T, 1)


def test_registered_model_runs_in_both_engines():
    """the stochastic-volatility model exists in ONE place (mp_models_extra.h); the checker's two engines agree on it"""
    obs = sv_observations(6)
    for canon in (0, O.VARIANT_CANONICAL):
        a = run(100, 1, 1, SV, 500, 9, O.VARIANT_SOA | canon, obs)
        b = run(100, 1, 1, SV, 500, 9, canon | O.VARIANT_FAST_SEARCH, obs)
        for (La, pa), (Lb, pb) in zip(a[0], b[0]):
            assert La == Lb and np.array_equal(pa, pb)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        assert np.isfinite(a[3])
    xs, ys = O.unfold_simulate(100, 1, 1, SV, 5, 2000, 3)
    assert np.isfinite(xs).all() and np.isfinite(ys).all()
    assert abs(xs[:, 0, 0].mean() - SV[0]) < 0.1 and abs(xs[:, 0, 0].std() - SV[3]) < 0.05
