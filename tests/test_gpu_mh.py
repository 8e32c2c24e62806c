"""GPU parity of mh / regen_mh (modppl/src/inference/mh.rs) over the reference's hierarchical model:
chain states bit-exact against the structure-faithful oracle (dynamic tries, restated Update /
Regenerate handlers) in canonical arithmetic, accept counts equal, logjp to 1e-12."""
import numpy as np
import pytest

from tests import oracle_lib as O

pytestmark = pytest.mark.gpu

XS = np.arange(-5.0, 6.0)  # tests/mh.rs:81


def make_ys(seed=0):
    rng = np.random.default_rng(seed)
    return 0.3 + 0.4 * XS + 0.5 * XS * XS + 0.1 * rng.normal(size=XS.size)  # tests/mh.rs:84-87


# Every test below runs twice: over the hand-written kernels (k_mh_iterate) and over the SAME model and proposals written as
# registered functors (csrc/mp_mh_models.h) and run by the generic Update / Regenerate / Simulate / Generate handlers of
# csrc/mp_genfn.h (k_fn_mh, k_fn_regen) — both against the oracle's dynamic-trie engine, bit for bit.
@pytest.fixture(params=[False, True], ids=["handwritten", "functor"], autouse=True)
def engine(request):
    global FUNCTOR
    FUNCTOR = request.param
    yield request.param


FUNCTOR = False


def pair(n, seed, constrain):
    import modppl_amd

    ys = make_ys()
    g = modppl_amd.HierarchicalChains(XS, ys, n, seed, constrain_is_linear=constrain, functor=FUNCTOR)
    o = O.OracleMH(XS, ys, n, seed, -1 if constrain is None else int(constrain), canonical=True)
    assert np.array_equal(g.states(), o.state())
    return g, o


def check(g, o):
    assert np.array_equal(g.states(), o.state())
    assert np.allclose(g.logjp(), o.logjp(), rtol=1e-12, atol=1e-9)


def test_initial_generate_free_branch():
    g, o = pair(3000, 5, None)
    st = g.states()
    assert 0.6 < st[:, 0].mean() < 0.8  # is_linear ~ bernoulli(0.7)
    check(g, o)


def test_regen_mh_cycle_quadratic_branch():
    """C4 of BASELINE.json: masks cycle {coeffs/a},{coeffs/b},{coeffs/c} with the chain held in the quadratic branch."""
    g, o = pair(2000, 11, False)
    for _ in range(3):
        assert g.regen_mh(["coeffs/a", "coeffs/b", "coeffs/c"], n_iters=7, cycle=True) == o.regen_mh([1, 2, 3], n_iters=7, cycle=True)
        check(g, o)
    assert g.iterations == 21


def test_regen_mh_joint_masks_and_mixed_branches():
    g, o = pair(2000, 12, None)
    assert g.regen_mh(["coeffs/a"], 3) == o.regen_mh([1], 3)
    check(g, o)
    assert g.regen_mh(["coeffs/b", "coeffs/c"], 4) == o.regen_mh([2, 3], 4)
    check(g, o)
    assert g.regen_mh(["coeffs/c"], 2) == o.regen_mh([3], 2)   # on linear chains: nothing changes, always accepted
    check(g, o)


def test_mh_drift_proposal():
    """tests/mh.rs:96-106: hierarchical_drift_proposal with std 0.1 then 0.01."""
    g, o = pair(2000, 13, None)
    for std, k in ((0.1, 3), (0.01, 10), (0.1, 3)):
        assert g.mh(std, k) == o.mh(std, k)
        check(g, o)


@pytest.mark.parametrize("constrain", [None, False, True])
def test_mh_add_or_remove_reference_loop(constrain):
    """The loop of the reference's own test (tests/mh.rs:93-106): one structure-changing add_or_remove_param_proposal
    move, three drift moves at 0.1, ten at 0.01 — chains switch between the linear and the quadratic branch."""
    g, o = pair(1500, 21, constrain)
    flips = 0
    for _ in range(6):
        before = g.states()[:, 0].copy()
        assert g.mh_add_or_remove(1) == o.mh_add_or_remove(1)
        check(g, o)
        flips += int((g.states()[:, 0] != before).sum())
        assert g.mh(0.1, 3) == o.mh(0.1, 3)
        check(g, o)
        assert g.mh(0.01, 10) == o.mh(0.01, 10)
        check(g, o)
    assert flips > 0
    st = g.states()
    assert np.all(st[st[:, 0] == 1.0][:, 3] == 0.0)   # linear traces carry no coeffs/c


def test_interleaved_kernels():
    g, o = pair(500, 14, False)
    for r in range(4):
        assert g.mh(0.1, 2) == o.mh(0.1, 2)
        assert g.regen_mh(["coeffs/a", "coeffs/b", "coeffs/c"], 3, cycle=True) == o.regen_mh([1, 2, 3], 3, cycle=True)
    check(g, o)


def test_unsupported_masks():
    import modppl_amd
    from modppl_amd import capi

    g = modppl_amd.HierarchicalChains(XS, make_ys(), 16, 1, functor=FUNCTOR)
    with pytest.raises(modppl_amd.ModpplError) as e:
        g.regen_mh(["is_linear"])
    # hand-written kernels: refused up front; generic handlers: the move runs generate(args, sub) on the old `coeffs` sub-trace,
    # and a quadratic chain that redraws linear leaves its c unconsumed — the reference's panic (dyngenfn.rs:526-529): reported
    assert e.value.code == (capi.MP_ERR_STATE if FUNCTOR else capi.MP_ERR_UNSUPPORTED)


def test_regen_mh_empty_mask_regenerates_every_site():
    """dyngenfn.rs:571: an empty mask is the trace's whole schema — is_linear, the coefficients AND the observed "(y, i)"
    sites are redrawn from their distributions, the weight is 0 and every move is accepted (mh.rs:62).  Chain states and
    the re-simulated observations equal the trie engine's bit for bit; later moves then run on each chain's own ys."""
    g, o = pair(1500, 13, None)
    assert np.array_equal(g.observations(), np.tile(make_ys(), (1500, 1)))
    for n_iters in (1, 3):
        assert g.regen_mh([], n_iters=n_iters) == o.regen_mh([], n_iters=n_iters) == 1500 * n_iters
        assert np.array_equal(g.states(), o.state())
        assert np.array_equal(g.observations(), o.observations(len(XS)))
        assert np.allclose(g.logjp(), o.logjp(), rtol=1e-12, atol=1e-9)
    st = g.states()
    assert 0.6 < st[:, 0].mean() < 0.8            # a fresh prior draw of is_linear
    # the chains carry their own observations from here on: drift MH and masked regenerate agree with the trie engine
    assert g.mh(0.1, n_iters=5) == o.mh(0.1, n_iters=5)
    check(g, o)
    assert g.regen_mh(["coeffs/a", "coeffs/b"], n_iters=4) == o.regen_mh([1, 2], n_iters=4)
    check(g, o)
    assert np.array_equal(g.observations(), o.observations(len(XS)))


def test_c4_full_size_posterior():
    """BASELINE config 4 size: 2^20 chains x 100 drift sweeps + regen cycles; the chains concentrate on the
    least-squares coefficients (data noise 0.1, 11 points)."""
    import modppl_amd

    ys = make_ys()
    n = 1 << 20
    g = modppl_amd.HierarchicalChains(XS, ys, n, 3, constrain_is_linear=False, functor=FUNCTOR)
    g.regen_mh(["coeffs/a", "coeffs/b", "coeffs/c"], 30, cycle=True)
    acc = 0
    for std in (0.5, 0.1, 0.02):
        acc += g.mh(std, 100)
    acc += g.mh(0.02, 200)
    st = g.states()
    A = np.stack([np.ones_like(XS), XS, XS * XS], axis=1)
    ols = np.linalg.lstsq(A, ys, rcond=None)[0]
    med = np.median(st[:, 1:], axis=0)
    assert np.all(np.abs(med - ols) < 0.05), (med, ols)
    assert 0 < acc < n * 500
    assert g.iterations == 530


BOUNDS = [-5.0, 5.0, -5.0, 5.0]            # tests/mh.rs:55
OBS_COV = [[1.0, -0.6], [-0.6, 2.0]]       # :61
NOISE = [[0.25, 0.0], [0.0, 0.25]]         # :64


@pytest.mark.parametrize("noise", [NOISE, [[0.3, 0.1], [0.1, 0.2]], [[9.0, 0.0], [0.0, 9.0]]])
def test_pointed_2d_mh_reference_loop(noise):
    """test_metropolis_hastings_dyngenfn (tests/mh.rs:50-68): uniform_2d prior, dense mvnormal likelihood and proposal
    (mvnormal.random = L z + mu).  Chain states and accept counts bit-exact; the wide proposal leaves the bounds
    (prior -inf -> rejected, never NaN)."""
    import modppl_amd

    n, seed = 3000, 4
    g = modppl_amd.PointedChains(BOUNDS, OBS_COV, [0.0, 0.0], n, seed, functor=FUNCTOR)
    o = O.OraclePointedMH(BOUNDS, OBS_COV, [0.0, 0.0], n, seed, canonical=True)
    assert np.array_equal(g.states(), o.state())
    for it in (1, 4, 20):
        assert g.mh(noise, it) == o.mh(noise, it)
        assert np.array_equal(g.states(), o.state())
        assert np.allclose(g.logjp(), o.logjp(), rtol=1e-12, atol=1e-12)
    st = g.states()
    assert np.all(np.abs(st) <= 5.0) and np.all(np.isfinite(st))


def test_pointed_2d_posterior_full_size():
    """2^20 chains, 200 sweeps: the posterior of latent given obs = 0 under a flat prior on the box is (nearly) the
    mvnormal(0, obs_cov) likelihood itself."""
    import modppl_amd

    g = modppl_amd.PointedChains(BOUNDS, OBS_COV, [0.0, 0.0], 1 << 20, 9, functor=FUNCTOR)
    acc = g.mh(NOISE, 200)
    assert 0.3 < acc / (200 * (1 << 20)) < 0.95
    st = g.states()
    cov = np.cov(st.T)
    assert np.allclose(st.mean(axis=0), 0.0, atol=0.01)
    assert np.allclose(cov, np.array(OBS_COV), atol=0.03)


def test_function_chains_argument_checks():
    """mp_mh_create_fn / registered proposals: unknown kinds, bad constraints and a constraint the model never visits."""
    import modppl_amd
    from modppl_amd import capi

    ys = make_ys()
    cons = {capi.MP_SITE_Y0 + k: y for k, y in enumerate(ys)}
    with pytest.raises(modppl_amd.ModpplError) as e:
        modppl_amd.FunctionChains(999, XS, cons, 8, 1)
    assert e.value.code == capi.MP_ERR_UNSUPPORTED
    with pytest.raises(modppl_amd.ModpplError) as e:
        modppl_amd.FunctionChains(capi.MP_MH_MODEL_HIERARCHICAL_FN, XS, {25: 0.0}, 8, 1)
    assert e.value.code == capi.MP_ERR_INVALID_ARG
    with pytest.raises(modppl_amd.ModpplError) as e:   # "(y, 12)" with 11 data points: generate leaves the constraint unconsumed
        modppl_amd.FunctionChains(capi.MP_MH_MODEL_HIERARCHICAL_FN, XS, {**cons, capi.MP_SITE_Y0 + 12: 1.0}, 8, 1)
    assert e.value.code == capi.MP_ERR_STATE
    g = modppl_amd.FunctionChains(capi.MP_MH_MODEL_HIERARCHICAL_FN, XS, cons, 64, 1)
    assert g.num_sites == 20
    with pytest.raises(modppl_amd.ModpplError) as e:
        g.mh(77)
    assert e.value.code == capi.MP_ERR_UNSUPPORTED
    with pytest.raises(modppl_amd.ModpplError) as e:
        g.mh(capi.MP_MH_PROPOSAL_HIERARCHICAL_DRIFT, [-1.0])
    assert e.value.code == capi.MP_ERR_INVALID_ARG
    vals, present = g.trace()
    assert np.array_equal(vals[:, capi.MP_SITE_Y0:capi.MP_SITE_Y0 + 11], np.tile(ys, (64, 1)))
    assert np.all((present >> capi.MP_SITE_Y0) == (1 << 11) - 1)


# ---- a model that exists ONLY as a registered functor (kind 102, robust regression with outlier indicators): the device's
# generic handlers (mp_genfn.h) against the checker's dynamic interpretation of the SAME functor bodies (tries, sample_at /
# trace_at / gc: oracle/src/mh_functor_adapter.hpp) ---------------------------------------------------------------------------
RL_OUT0, RL_Y0 = 2, 14


def robust_line_pair(n, seed, n_data=10):
    import modppl_amd

    xs = np.linspace(-3, 3, n_data)
    rng = np.random.default_rng(1)
    ys = 0.7 * xs - 0.4 + 0.3 * rng.normal(size=n_data)
    ys[2] += 9.0
    ys[n_data - 3] -= 8.0
    cons = {RL_Y0 + k: y for k, y in enumerate(ys)}
    g = modppl_amd.FunctionChains(102, xs, cons, n, seed)
    o = O.OracleFunctionChains(102, xs, cons, n, seed)
    return g, o, n_data


def check_fn(g, o):
    gv, gp = g.trace()
    ov, op = o.trace()
    assert np.array_equal(gp, op)
    assert np.array_equal(gv, ov)
    assert np.allclose(g.logjp(), o.logjp(), rtol=1e-12, atol=1e-9)


def test_registered_only_model_against_the_dynamic_interpretation():
    if FUNCTOR:
        pytest.skip("one engine: the model has no hand-written kernel")
    g, o, nd = robust_line_pair(1500, 17)
    check_fn(g, o)
    for sweep in range(3):
        assert g.mh(1, [0.3], 2) == o.mh(1, [0.3], 2)                       # drift of the line (a sub-call's sites, constrained)
        check_fn(g, o)
        for k in (0, 2, nd - 3, nd - 1):
            assert g.mh(2, [k], 1) == o.mh(2, [k], 1)                       # flip of one indicator: a bernoulli site constrained
            check_fn(g, o)
        assert g.regen_mh([RL_OUT0 + k for k in range(nd)], nd, cycle=True) == o.regen_mh([RL_OUT0 + k for k in range(nd)], nd, cycle=True)
        check_fn(g, o)
        assert g.regen_mh([0], 2) == o.regen_mh([0], 2)                     # slope alone: masked site inside the sub-call
        check_fn(g, o)
        assert g.regen_mh([1, RL_OUT0 + 1, RL_OUT0 + 4], 2) == o.regen_mh([1, RL_OUT0 + 1, RL_OUT0 + 4], 2)   # sub-call and top-level sites at once
        check_fn(g, o)
    assert g.regen_mh([RL_OUT0 + 3], 3) == o.regen_mh([RL_OUT0 + 3], 3)     # masked top-level site AFTER the untouched sub-call: replayed
    check_fn(g, o)
    assert g.regen_mh([], 2) == o.regen_mh([], 2) == 3000                   # empty mask: the whole schema, observed sites included
    check_fn(g, o)
    assert g.mh(1, [0.2], 3) == o.mh(1, [0.2], 3)
    check_fn(g, o)
    assert g.iterations == 3 * (2 + 4 + nd + 2 + 2) + 3 + 2 + 3


def test_registered_only_model_full_size():
    """2^20 chains of the functor-only model: planted outliers are found by the flip moves"""
    if FUNCTOR:
        pytest.skip("one engine: the model has no hand-written kernel")
    import modppl_amd

    n_data = 10
    xs = np.linspace(-3, 3, n_data)
    rng = np.random.default_rng(1)
    ys = 0.7 * xs - 0.4 + 0.3 * rng.normal(size=n_data)
    ys[2] += 9.0
    ys[7] -= 8.0
    g = modppl_amd.FunctionChains(102, xs, {RL_Y0 + k: y for k, y in enumerate(ys)}, 1 << 20, 5)
    for sweep in range(40):
        g.mh(1, [0.4 if sweep < 20 else 0.1], 2)
        for k in range(n_data):
            g.mh(2, [k], 1)
    vals, present = g.trace()
    out = vals[:, RL_OUT0:RL_OUT0 + n_data].mean(axis=0)
    assert out[2] > 0.9 and out[7] > 0.9 and out[3:7].max() < 0.5, out
    assert np.isfinite(g.logjp()).all()


# ---- a change UPSTREAM of an untouched sub-call (kind 103: `big` is drawn before the line's sub-call): regenerate takes
# trace_at's generate(args, sub) arm — weight += new_weight - sub.weight(), the sub-trie's RUNNING weight (dyngenfn.rs:424-428) —
# and update its update(sub, args, Unknown, {}) arm (:371-381).  Device (mp_genfn.h: `subw`) against the dynamic interpretation. --
SL_BIG, SL_SLOPE, SL_ICPT, SL_Y0 = 0, 1, 2, 3


def test_change_upstream_of_an_untouched_sub_call():
    if FUNCTOR:
        pytest.skip("one engine: the model has no hand-written kernel")
    import modppl_amd

    n, nd = 2000, 9
    xs = np.linspace(-2, 2, nd)
    ys = -0.6 * xs + 0.8 + 0.7 * np.random.default_rng(4).normal(size=nd)
    cons = {SL_Y0 + k: y for k, y in enumerate(ys)}
    g = modppl_amd.FunctionChains(103, xs, cons, n, 29)
    o = O.OracleFunctionChains(103, xs, cons, n, 29)
    check_fn(g, o)
    for sweep in range(3):
        assert g.regen_mh([SL_BIG], 2) == o.regen_mh([SL_BIG], 2)            # masked upstream site, the sub-call unmasked: generate(args, sub)
        check_fn(g, o)
        assert g.mh(2, [0.2], 2) == o.mh(2, [0.2], 2)                        # drift inside the sub-call: its running weight moves (remove / observe)
        check_fn(g, o)
        assert g.mh(1, [], 2) == o.mh(1, [], 2)                              # toggle upstream: update(sub, Unknown, {}) rescoring inside the sub-call
        check_fn(g, o)
        assert g.regen_mh([SL_BIG, SL_SLOPE], 2) == o.regen_mh([SL_BIG, SL_SLOPE], 2)   # both: the inner regenerate arm
        check_fn(g, o)
        assert g.regen_mh([SL_BIG, SL_ICPT, SL_SLOPE], 3, cycle=True) == o.regen_mh([SL_BIG, SL_ICPT, SL_SLOPE], 3, cycle=True)
        check_fn(g, o)
    assert g.regen_mh([], 1) == o.regen_mh([], 1) == n
    check_fn(g, o)
    assert g.regen_mh([SL_BIG], 3) == o.regen_mh([SL_BIG], 3)
    check_fn(g, o)


def test_two_levels_of_sub_calls():
    """kind 113: a sub-call inside a sub-call — constraints that land in the inner call, a structure change in the middle one, a
    change upstream of both, and every mask placement — the device's static handlers (mp_genfn.h: `call` as a frame) against the
    checker's recursive trace_at over real nested tries, through mh and regen_mh.
    WHAT THIS PINS AND WHAT IT DOES NOT: the checker interprets the product's own functor source (oracle/src/mh_functor_adapter.hpp), so
    the HANDLER rules — sample_at / trace_at / gc at depth, dyngenfn.rs:100-486 — are held to an independent restatement, but the model
    BODY is shared: a mis-written body would be invisible here.  The model is synthetic (no counterpart in the reference), so there is
    no independent statement of it to hold it to; kinds 102 / 103 have one (tests/test_gpu_functor_logjoint.py: numpy log-joints)."""
    if FUNCTOR:
        pytest.skip("one engine: the model has no hand-written kernel")
    import modppl_amd
    from tests.test_oracle_mh_functor import nested_moves, nested_problem

    xs, cons = nested_problem()
    n = 3000
    g = modppl_amd.FunctionChains(113, xs, cons, n, 13)
    o = O.OracleFunctionChains(113, xs, cons, n, 13)
    nested_moves(g, o, check_fn, sweeps=2)
    vals, present = g.trace()
    assert 0 < ((present >> 4) & 1).sum() < n


def test_more_than_32_sites():
    """kind 114: 41 sites — presence, masks, constraints and the discard are 64-bit words inside the kernels and two 32-bit words per
    chain through the C ABI; the sub-call, its optional choice and six observations live above bit 32.
    WHAT THIS PINS AND WHAT IT DOES NOT: the checker interprets the product's own functor source (oracle/src/mh_functor_adapter.hpp), so
    the HANDLER rules — sample_at / trace_at / gc at depth, dyngenfn.rs:100-486 — are held to an independent restatement, but the model
    BODY is shared: a mis-written body would be invisible here.  The model is synthetic (no counterpart in the reference), so there is
    no independent statement of it to hold it to; kinds 102 / 103 have one (tests/test_gpu_functor_logjoint.py: numpy log-joints)."""
    if FUNCTOR:
        pytest.skip("one engine: the model has no hand-written kernel")
    import modppl_amd
    from tests.test_oracle_mh_functor import WF_NS, WF_O1, WF_O2, wide_moves, wide_problem

    xs, cons = wide_problem()
    n = 1500
    g = modppl_amd.FunctionChains(114, xs, cons, n, 5)
    o = O.OracleFunctionChains(114, xs, cons, n, 5)
    assert g.num_sites == WF_NS
    wide_moves(g, o, check_fn)
    # the GFI calls one at a time with per-chain tables: choices, discard and constraints above bit 32 through the two-word layout
    (gcv, gcp), gf = g.propose(2, [], rng_step=77)
    (ocv, ocp), of = o.propose(2, [], rng_step=77)
    assert gcp.dtype == np.uint64 and np.array_equal(gcp, ocp) and np.array_equal(gcv, ocv) and np.array_equal(gf, of)
    gw, gd = g.update((gcv, gcp), rng_step=77)
    ow, od = o.update((ocv, ocp), rng_step=77)
    assert np.array_equal(gw, ow) and np.array_equal(gd[1], od[1]) and np.array_equal(gd[0], od[0])
    assert ((gd[1] >> np.uint64(WF_O2)) & np.uint64(1)).any()              # some chains dropped o2: a discard bit in the high word
    assert np.array_equal(g.assess(gd, proposal_kind=2, rng_step=77), o.assess(od, proposal_kind=2, rng_step=77))
    assert np.array_equal(g.regenerate([WF_O1, 0], rng_step=78), o.regenerate([WF_O1, 0], rng_step=78))
    check_fn(g, o)


def test_masking_is_linear_where_the_reference_does_not_panic():
    """hierarchical model as a functor, all chains linear, ONE regen move with mask {is_linear}: linear -> linear and linear ->
    quadratic go through generate(args, sub) on the old `coeffs` sub-trace (c is drawn when the new branch wants it); only
    quadratic -> linear leaves a constraint behind and panics (dyngenfn.rs:526-529) — there is no such chain yet."""
    if not FUNCTOR:
        pytest.skip("the hand-written kernels refuse the mask up front")
    g, o = pair(1500, 33, True)
    assert g.regen_mh(["is_linear"], 1) == o.regen_mh([0], 1)
    check(g, o)
    assert 0.02 < (g.states()[:, 0] == 0.0).mean() < 0.4    # some chains moved to the quadratic branch (30 % proposed it)
    import modppl_amd
    from modppl_amd import capi

    with pytest.raises(modppl_amd.ModpplError) as e:          # now some chains are quadratic: quadratic -> linear is the reference's panic
        g.regen_mh(["is_linear"], 1)
    assert e.value.code == capi.MP_ERR_STATE
