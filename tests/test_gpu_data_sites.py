"""GPU: declared data sites (modppl_amd/csrc/mp_genfn.h; kind 105 = the reference's `hierarchical_model`, hierarchical.rs:33-47, with its
"(y, j)" sites declared as data).  An observation is then no register-resident site: nothing is stored per observation, the handlers
recompute the previous trace's log-density where `sample_at` (dyngenfn.rs:143-273) reads the stored one, and the 64-site cap of a
static trace does not apply.  Held here:
  * 11 observations: the chains of kind 105 against the hand-written kernels (k_mh_iterate) bit for bit;
  * 200 observations — beyond any static trace — against the checker's trie engine interpreting the same functor (real "(y, j)" trie
    entries with their stored weights): mh with both proposals, regen_mh with single / joint / cycled masks, standalone update /
    regenerate / assess / propose, importance sampling;
  * what a declared data site cannot do is an error, not a silent difference."""
import numpy as np
import pytest

from tests import oracle_lib as O

pytestmark = pytest.mark.gpu
KIND = 105


def _data(n_obs, seed=0):
    xs = np.linspace(-5.0, 5.0, n_obs) if n_obs != 11 else np.arange(-5.0, 6.0)
    ys = 0.3 + 0.4 * xs + 0.5 * xs * xs + 0.1 * np.random.default_rng(seed).normal(size=n_obs)
    return xs, ys, {4 + j: float(y) for j, y in enumerate(ys)}


def _same_trace(g, r):
    gv, gp = g.trace()
    rv, rp = r.trace()
    assert np.array_equal(gp, rp)
    assert np.array_equal(gv, rv)


@pytest.mark.parametrize("constrain", [None, False, True])
def test_eleven_observations_equal_the_hand_written_kernels(constrain):
    import modppl_amd

    xs, ys, _ = _data(11)
    n, seed = 5000, 41
    hw = modppl_amd.HierarchicalChains(xs, ys, n, seed, constrain_is_linear=constrain)
    dt = modppl_amd.HierarchicalChains(xs, ys, n, seed, constrain_is_linear=constrain, functor="data")
    assert np.array_equal(hw.states(), dt.states())
    for rnd in range(3):
        assert hw.mh(0.05, 4) == dt.mh(0.05, 4)
        assert np.array_equal(hw.states(), dt.states())
        assert hw.mh_add_or_remove(3) == dt.mh_add_or_remove(3)
        assert np.array_equal(hw.states(), dt.states())
        masks = ["coeffs/a", "coeffs/b"] if constrain else ["coeffs/a", "coeffs/b", "coeffs/c"]
        if constrain is not True:   # (masking coeffs/c on a chain that is linear is the reference's panic)
            lin = hw.states()[:, 0]
            if np.all(lin == 0.0):
                assert hw.regen_mh(masks, 6, cycle=True) == dt.regen_mh(masks, 6, cycle=True)
        assert hw.regen_mh(["coeffs/a", "coeffs/b"], 2) == dt.regen_mh(["coeffs/a", "coeffs/b"], 2)
        assert np.array_equal(hw.states(), dt.states())
    assert np.allclose(hw.logjp(), dt.logjp(), rtol=1e-13, atol=1e-10)


@pytest.mark.parametrize("n_obs", [64, 200])
def test_two_hundred_observations_against_the_trie_engine(n_obs):
    import modppl_amd

    xs, ys, cons = _data(n_obs, seed=n_obs)
    n, seed = 320, 19
    g = modppl_amd.FunctionChains(KIND, xs, cons, n, seed)
    r = O.OracleFunctionChains(KIND, xs, cons, n, seed, canonical=True)
    assert g.num_sites == 4
    _same_trace(g, r)
    for rnd in range(3):
        assert g.mh(1, [0.01], 5) == r.mh(1, [0.01], 5)
        _same_trace(g, r)
        assert g.mh(2, [], 3) == r.mh(2, [], 3)
        _same_trace(g, r)
        assert g.regen_mh([1, 2], 3) == r.regen_mh([1, 2], 3)
        _same_trace(g, r)
        assert g.regen_mh([1], 2) == r.regen_mh([1], 2)
        _same_trace(g, r)
    # the GFI calls one at a time, with a step of their own
    step = 900
    for diff in (0, 1):
        gw, (gdv, gdp) = g.update({2: 0.41 + diff}, argdiff=diff, rng_step=step)
        rw, (rdv, rdp) = r.update({2: 0.41 + diff}, argdiff=diff, rng_step=step)
        assert np.array_equal(gw, rw) and np.array_equal(gdp, rdp) and np.array_equal(gdv, rdv)
        _same_trace(g, r)
        step += 1
    gw, rw = g.regenerate([1, 3], rng_step=step), r.regenerate([1, 3], rng_step=step)
    assert np.array_equal(gw, rw)
    _same_trace(g, r)
    (gcv, gcp), gf = g.propose(1, [0.02], rng_step=step + 1)
    (rcv, rcp), rf = r.propose(1, [0.02], rng_step=step + 1)
    assert np.array_equal(gf, rf) and np.array_equal(gcv, rcv) and np.array_equal(gcp, rcp)
    gw, gd = g.update((gcv, gcp), rng_step=step + 1)
    rw, rd = r.update((rcv, rcp), rng_step=step + 1)
    assert np.array_equal(gw, rw)
    assert np.array_equal(g.assess(gd, proposal_kind=1, proposal_args=[0.02], rng_step=step + 1),
                          r.assess(rd, proposal_kind=1, proposal_args=[0.02], rng_step=step + 1))
    _same_trace(g, r)
    assert np.allclose(g.logjp(), r.logjp(), rtol=1e-12, atol=1e-9)


def test_importance_sampling_with_two_hundred_observations():
    import modppl_amd

    xs, ys, cons = _data(200, seed=3)
    n, m, seed = 2048, 45, 11
    tr, idx, lml = modppl_amd.fn_importance_resampling(KIND, xs, cons, n, m, seed)
    rtr, rlnw, rlml, ridx = O.OracleFunctionChains.importance(KIND, xs, cons, n, m, seed, canonical=True)
    assert lml == rlml and np.array_equal(idx, ridx)
    _same_trace(tr, rtr)


def test_what_a_declared_data_site_cannot_do_is_an_error():
    import modppl_amd

    xs, ys, cons = _data(40)
    g = modppl_amd.FunctionChains(KIND, xs, cons, 128, 1)
    E = modppl_amd.capi.ModpplError
    with pytest.raises(E):
        g.regen_mh([], 1)                       # the empty mask = the whole schema: would re-simulate the observations
    with pytest.raises(E):
        g.regenerate([], rng_step=5)
    with pytest.raises(E):
        g.simulate(rng_step=5)
    with pytest.raises(E):
        g.update({4: 0.0})                      # an observation's site id is not a site of the trace
    missing = dict(cons)
    del missing[4 + 17]
    with pytest.raises(E):
        modppl_amd.FunctionChains(KIND, xs, missing, 128, 1)
    with pytest.raises(E):
        modppl_amd.FunctionChains(KIND, xs, {}, 128, 1, simulate=True)
    assert g.mh(1, [0.01], 2) >= 0               # the handle is as it was
