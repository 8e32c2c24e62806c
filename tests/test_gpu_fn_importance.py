"""GPU: `GenFn::generate` / `simulate` as batched calls and importance sampling over a REGISTERED generative function
(modppl/src/gfi.rs:51-55, modppl/src/inference/importance.rs:12-50) — mp_fn_generate, mp_fn_simulate, mp_fn_generate_create,
mp_fn_simulate_create, mp_fn_importance_sampling, mp_fn_importance_resampling of the C ABI.

`importance_sampling` is generic over `impl GenFn`; the reference's third importance test runs it on `hierarchical_model` with 11
observations and 10 000 samples, resampling sqrt(N) of them, under the comment "this works with ~1,000,000 particles"
(modppl/tests/importance.rs:89-139).  Here: that shape bit for bit against the checker's generic `importance_resampling` over real
tries (oracle/src/inference.hpp:469-518 driven through oracle/src/mh_functor_adapter.hpp), then the 2^20-sample run the comment asks
for, held to the posterior it should find."""
import numpy as np
import pytest

from tests import oracle_lib as O

pytestmark = pytest.mark.gpu
HIER, Y0 = 101, 4   # the hierarchical model as a registered functor; "(y, k)" = site 4 + k


def _hier_data(seed=20241008):
    xs = np.arange(-5.0, 6.0)                                       # tests/importance.rs:98
    ys = 0.3 + 0.4 * xs + 0.5 * xs * xs + np.random.default_rng(seed).normal(0.0, 0.1, xs.size)   # :101-105
    return xs, {Y0 + k: float(y) for k, y in enumerate(ys)}


def _same_trace(g, r):
    gv, gp = g.trace()
    rv, rp = r.trace()
    assert np.array_equal(gp, rp)
    assert np.array_equal(gv, rv)


@pytest.mark.parametrize("kind", [101, 102, 103, 113])
def test_generate_and_simulate_against_the_dynamic_interpretation(kind):
    """generate (shared constraints, per-chain constraints, none at all) and simulate on four registered models, the two-level nested one
    included: weights, logjp and every trace value against the checker's `DynGenFn::generate / simulate` over tries."""
    import modppl_amd

    rng = np.random.default_rng(kind)
    if kind == 101:
        xs, cons = _hier_data(3)
    elif kind == 102:
        xs = np.linspace(-2, 2, 8)
        cons = {2 + 12 + k: v for k, v in enumerate(0.7 * xs - 0.2 + rng.normal(0, 0.4, xs.size))}
    elif kind == 113:
        xs = np.array([0.5, -1.0, 1.5, 2.0])
        cons = {6 + k: v for k, v in enumerate(1.3 * xs + rng.normal(0, 0.2, xs.size))}
    else:
        xs = np.linspace(-1, 3, 6)
        cons = {3 + k: v for k, v in enumerate(1.1 * xs + 0.3 + rng.normal(0, 0.6, xs.size))}
    n, seed = 513, 29
    g = modppl_amd.FunctionChains(kind, xs, cons, n, seed)
    r = O.OracleFunctionChains(kind, xs, cons, n, seed, canonical=True)
    _same_trace(g, r)
    # the constructor IS a generate at Philox step 0: its weights are what a second generate of the same step returns
    w0 = r.generate(cons, rng_step=0 + 7)   # (any explicit step: the checker's handle has no step-0 entry point after creation)
    assert np.array_equal(g.generate(cons, rng_step=7), w0)
    _same_trace(g, r)
    # assess = generate's weight (gfi.rs:85-90), traces untouched
    assert np.array_equal(g.assess(cons, rng_step=7), w0)
    # simulate: every site drawn; logjp = the trace's score (the device sums the choices' log-densities in visit order, the reference's
    # value is the trie's running weight — a sub-call's total enters as one term —: the same real number, last bits apart, as for
    # mp_mh_read_logjp)
    assert np.allclose(g.simulate(rng_step=8), r.simulate(rng_step=8), rtol=1e-13, atol=1e-12)
    _same_trace(g, r)
    assert np.allclose(g.logjp(), r.logjp(), rtol=1e-13, atol=1e-12)
    # per-chain constraints: every choice of the simulated traces -> weight = logjp, same traces back
    tv, tp = g.trace()
    gw, rw = g.generate((tv, tp), rng_step=9), r.generate((tv, tp), rng_step=9)
    assert np.array_equal(gw, rw)
    assert np.allclose(gw, g.logjp(), rtol=1e-13, atol=1e-12)
    _same_trace(g, r)
    # no constraints at all: a prior draw with weight 0 (dyngenfn.rs:132-140)
    gw, rw = g.generate({}, rng_step=10), r.generate({}, rng_step=10)
    assert np.all(gw == 0.0) and np.array_equal(gw, rw)
    _same_trace(g, r)
    # the two constructors
    gs = modppl_amd.FunctionChains(kind, xs, {}, n, seed + 1, simulate=True)
    rs = O.OracleFunctionChains(kind, xs, {}, n, seed + 1, canonical=True)
    assert np.allclose(gs.simulate(rng_step=3), rs.simulate(rng_step=3), rtol=1e-13, atol=1e-12)
    _same_trace(gs, rs)
    assert gs.initial_weights.shape == (n,) and np.all(np.isfinite(gs.initial_weights))


def test_a_constraint_nobody_consumes_is_an_error_and_leaves_the_chains_alone():
    """generate with a constraint on a site the model does not visit is the reference's panic (dyngenfn.rs:526-529): MP_ERR_STATE, the
    chains keep their traces, and the rejected call does not consume a Philox step."""
    import modppl_amd

    g = modppl_amd.FunctionChains(110, [], {0: 0.0}, 64, 1)   # b = false: x is not visited
    before, it0 = g.trace(), g.iterations
    with pytest.raises(modppl_amd.capi.ModpplError):
        g.generate({0: 0.0, 1: 0.3})
    after = g.trace()
    assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1])
    with pytest.raises(modppl_amd.capi.ModpplError):
        g.generate({7: 0.3})                                   # not a site of the model: rejected before anything is touched
    assert g.iterations == it0 + 1                             # (only the call that ran consumed a step)


@pytest.mark.parametrize("n", [1000, 4096])
def test_importance_hierarchical_against_the_trie_engine(n):
    """tests/importance.rs:89-139 (11 observations, sqrt(N) resampled) at sizes the trie engine walks in seconds: log-ML, every
    normalised weight, every resampled index and every trace value bit-equal to the canonical checker; the literal checker (libm,
    sequential sums, the linear-scan categorical) agrees on every index and to 1e-12 on log-ML."""
    import modppl_amd

    xs, cons = _hier_data()
    m, seed = int(np.sqrt(n)), 77
    tr, idx, lml = modppl_amd.fn_importance_resampling(HIER, xs, cons, n, m, seed)
    _, lnw, lml2 = modppl_amd.fn_importance_sampling(HIER, xs, cons, n, seed, traces=False)
    rtr, rlnw, rlml, ridx = O.OracleFunctionChains.importance(HIER, xs, cons, n, m, seed, canonical=True)
    assert lml == rlml == lml2
    assert np.array_equal(lnw, rlnw)
    assert np.array_equal(idx, ridx)
    _same_trace(tr, rtr)
    _, llnw, llml, lidx = O.OracleFunctionChains.importance(HIER, xs, cons, n, m, seed, canonical=False)
    assert abs(lml - llml) <= 1e-12 * abs(llml)
    assert int(np.count_nonzero(idx != lidx)) == 0
    assert np.allclose(lnw, llnw, rtol=0, atol=1e-9)
    # the traces handle is an ordinary chain handle: an MH move on the N samples
    assert tr.mh(1, [0.1], 1) >= 0
    assert abs(np.exp(lnw).sum() - 1.0) < 1e-9


def test_importance_hierarchical_at_a_million_samples():
    """"this works with ~1,000,000 particles" (tests/importance.rs:90-92): 2^20 samples, 1024 resampled; size-independent properties —
    determinism, normalised weights, indices in range — and the posterior the test prints: quadratic structure, coefficients near
    (0.3, 0.4, 0.5)."""
    import modppl_amd

    xs, cons = _hier_data()
    n, m, seed = 1 << 20, 1024, 5
    tr, idx, lml = modppl_amd.fn_importance_resampling(HIER, xs, cons, n, m, seed)
    tr2, idx2, lml2 = modppl_amd.fn_importance_resampling(HIER, xs, cons, n, m, seed, traces=False)
    assert lml == lml2 and np.array_equal(idx, idx2) and np.isfinite(lml)
    assert idx.max() < n
    vals, present = tr.trace()
    sel = vals[idx.astype(np.int64)]
    is_linear = sel[:, 0]
    assert np.mean(is_linear) < 0.05                       # the data are quadratic
    quad = sel[is_linear == 0.0]
    assert abs(np.mean(quad[:, 3]) - 0.5) < 0.05           # c (the sharpest: 0.1 noise against x^2 up to 25)
    assert abs(np.mean(quad[:, 2]) - 0.4) < 0.1            # b
    assert abs(np.mean(quad[:, 1]) - 0.3) < 0.8            # a ("especially the intercept" is poor)
