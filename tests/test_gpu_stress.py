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


def _random_program(rng, length):
    """a random sequence of filter operations: what may follow what in the reference's API"""
    ops = []
    for _ in range(length):
        r = rng.random()
        if r < 0.45:
            ops.append(("resample_async", int(rng.choice([0, 0, 1, 2]))))
        elif r < 0.55:
            ops.append(("resample_sync", int(rng.choice([0, 1, 2]))))
        elif r < 0.65:
            ops.append(("read", str(rng.choice(["parents", "states", "log_weights", "ess", "lml"]))))
        ops.append(("step", None))
    return ops


@pytest.mark.parametrize("rep", [0, 1, 2])
@pytest.mark.parametrize("name,n,seed", [("lgssm1", 70001, 1), ("lgssm1", (1 << 20) + 4096 + 5, 2), ("lgssm1", (1 << 21) + 2048, 3),
                                         ("bearings", 50001, 4), ("band2", 2 * 2048 * 1024 + 2048, 5), ("spiral", 9000, 6)])
def test_random_programs_fused_against_unfused(name, n, seed, rep, monkeypatch, diag):
    """The state machine around the draws a step may make for itself (pending scheme, flush on a read, synchronous resamples in
    between, both table forms, the lattice instantiation): a random program of steps, asynchronous / synchronous resamples of
    all three schemes and reads, run on a handle whose k_propagate draws and on one that always launches k_draw_slots
    (MP_FUSED_DRAWS=0).  Every read and the final state must agree bit for bit."""
    import modppl_amd
    from tests.test_gpu_deferred import _wide_case

    rng = np.random.default_rng(900 + seed + 100 * rep)
    prog = _random_program(rng, 14)
    T = 1 + sum(1 for op, _ in prog if op == "step")
    if name == "lgssm1":
        model, obs = modppl_amd.lgssm_model(*O.LGSSM_PARAMS), O.lgssm_observations(T).reshape(T, 1)
    else:
        model, obs = _wide_case(name, T)
    fused = modppl_amd.ParticleSystem(model, n, 40 + seed)
    monkeypatch.setenv("MP_FUSED_DRAWS", "0")
    plain = modppl_amd.ParticleSystem(model, n, 40 + seed)
    monkeypatch.delenv("MP_FUSED_DRAWS")
    args0 = [0.0, 0.0] if name == "spiral" else None
    for pf in (fused, plain):
        pf.init_step(args0, obs[:1])
    t = 1
    resampled = False
    for op, arg in prog:
        got = []
        for pf in (fused, plain):
            if op == "step":
                pf.step(obs[t:t + 1])
            elif op == "resample_async":
                pf.resample(arg, sync=False)
            elif op == "resample_sync":
                got.append(pf.resample(arg, sync=True))
            elif arg == "parents":
                got.append(pf.parents.copy() if resampled else None)
            elif arg == "states":
                got.append(pf.states())
            elif arg == "log_weights":
                got.append(pf.log_weights.copy())
            elif arg == "ess":
                got.append(pf.effective_sample_size(fresh=True))
            else:
                got.append(pf.log_marginal_likelihood_estimate())
        if op == "step":
            t += 1
        if op.startswith("resample"):
            resampled = True
        if got and got[0] is not None:
            assert np.array_equal(np.asarray(got[0]), np.asarray(got[1])), (name, op, arg, t)
    assert np.array_equal(fused.states(), plain.states())
    assert np.array_equal(fused.log_weights, plain.log_weights)
    if resampled:
        assert np.array_equal(fused.parents, plain.parents)
    assert fused.log_marginal_likelihood_estimate() == plain.log_marginal_likelihood_estimate()
