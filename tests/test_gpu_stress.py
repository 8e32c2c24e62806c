"""GPU: randomised sweep of the owner-keeps exchange and of the unsharded filter against the CPU checker — sizes around
tile / chunk / super-chunk edges, odd world sizes, all three resampling schemes, tails that collapse the weights.  Twelve
seeded cases by default (MP_STRESS_CASES=N for more)."""
import os

import numpy as np
import pytest

from tests import oracle_lib as O
from tests.owned_ref import OwnedReference
from tests.test_gpu_owned import _ByHand, _model

pytestmark = pytest.mark.gpu
CASES = int(os.environ.get("MP_STRESS_CASES", "12"))


def _cases():
    rng = np.random.default_rng(20241008)
    out = []
    for k in range(CASES):
        world = int(rng.choice([1, 2, 3, 4, 5, 8]))
        tiles = int(rng.integers(1, 5))
        n = 2048 * tiles                                   # shards are tile-aligned
        scheme = int(rng.integers(0, 3))
        cap = int(rng.choice([0, 8, 64, 4096]))
        d = int(rng.choice([1, 1, 4, 16]))
        tail = float(rng.choice([6.0, 14.0, 30.0]))
        out.append((k, d, world, n, cap, scheme, tail))
    return out


@pytest.mark.parametrize("k,d,world,n,cap,scheme,tail", _cases())
def test_owner_keeps_random_configurations(k, d, world, n, cap, scheme, tail):
    model, obs = _model(d, 5)
    if d == 1:
        obs = obs.copy()
        obs[2] = tail
    N, seed = n * world, 1000 + k
    hip = _ByHand(model, n, world, seed)
    ref = OwnedReference(model, N, seed, world)
    for e in hip.eng:
        e.init_step(None, obs[:1])
    ref.init_step(None, obs[:1])
    for t in range(1, len(obs)):
        assert hip.resample(cap, scheme) == ref.resample(scheme)
        assert list(hip.counts) == list(ref.counts)
        if (k + t) % 2:
            assert np.array_equal(hip.cat(lambda e: e.parents()), ref.parents())
        for e in hip.eng:
            e.step(obs[t:t + 1])
        ref.step(obs[t:t + 1])
        assert np.array_equal(hip.cat(lambda e: e.log_weights()), ref.log_weights())
        if (k + t) % 3 == 0:   # the parents of that resample, read only now: they come out of the exchange rows / the flagged entries
            assert np.array_equal(hip.cat(lambda e: e.parents()), ref.parents())
    assert np.array_equal(hip.cat(lambda e: e.states()), ref.states())


@pytest.mark.parametrize("n,seed", [(2047, 1), (2049, 2), (3 * 2048 + 1, 3), (65536 + 63, 4), (1 << 17, 5)])
@pytest.mark.parametrize("scheme", [0, 1, 2])
def test_unsharded_filter_edges(n, seed, scheme):
    import modppl_amd

    T = 5
    ys = O.lgssm_observations(T)
    pf = modppl_amd.ParticleSystem(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), n, seed)
    ref = O.OraclePF(1, 1, 1, O.LGSSM_PARAMS, n, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA)
    pf.init_step(None, ys[:1].reshape(1, 1))
    ref.init_step(ys[:1].reshape(1, 1))
    for t in range(1, T):
        assert pf.resample(scheme=scheme) == ref.resample(scheme)
        assert np.array_equal(pf.parents, ref.parents())
        pf.step(ys[t:t + 1].reshape(1, 1))
        ref.step(ys[t:t + 1].reshape(1, 1))
        assert np.array_equal(pf.log_weights, ref.log_weights())
    assert pf.log_marginal_likelihood_estimate() == ref.log_marginal_likelihood_estimate()
