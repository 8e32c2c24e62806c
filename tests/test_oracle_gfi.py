"""The reference's known answers for `update` (modppl/tests/dyngenfn.rs:55-114) through the checker's dynamic machinery on the
REGISTERED forms of its regression functions (modppl_amd/csrc/mp_mh_models.h kinds 110-112): these pin the checker's
update / regenerate / assess / propose entry points, which tests/test_gpu_gfi.py then holds the device's mp_fn_* against.
Constants quoted from the reference's tests."""
import numpy as np
import pytest

from tests import oracle_lib as O

B, X = 0, 1          # kind 110 / 112 site ids
M_, X_, Y_ = 0, 1, 2  # kind 111
UNKNOWN = 1


@pytest.mark.parametrize("canonical", [False, True])
def test_reference_update_known_answers(canonical):
    # test_sample_at_update_prev_and_constrained (:55-66): generate {b: true, x: 0.0}; update {x: 1.0} -> -0.5 exactly
    f = O.OracleFunctionChains(110, [], {B: 1.0, X: 0.0}, 3, 1, canonical=canonical)
    w, (dv, dp) = f.update({X: 1.0}, argdiff=UNKNOWN)
    assert np.all(w == -0.5)
    assert np.all(dp == (1 << X)) and np.all(dv[:, X] == 0.0)   # the discard holds the old x
    # test_sample_at_update_no_prev_and_constrained (:68-79): generate {b: false}; update {b: true, x: 1.0} -> -2.517551
    f = O.OracleFunctionChains(110, [], {B: 0.0}, 3, 2, canonical=canonical)
    w, (dv, dp) = f.update({B: 1.0, X: 1.0}, argdiff=UNKNOWN)
    assert np.allclose(w, -2.517551, atol=1e-6, rtol=0)
    assert np.all(dp == (1 << B)) and np.all(dv[:, B] == 0.0)
    # test_update_sample_at_prev_and_unconstrained (:81-93): generate {m: 1, x: 1, y: -0.3}; update {m: 0.5} -> 0.4
    f = O.OracleFunctionChains(111, [], {M_: 1.0, X_: 1.0, Y_: -0.3}, 3, 3, canonical=canonical)
    w, _ = f.update({M_: 0.5}, argdiff=UNKNOWN)
    assert np.allclose(w, 0.4, atol=1e-6, rtol=0)
    # test_update_no_prev_and_unconstrained (:95-114): generate {b: false}; update {b: true} -> ln 0.25 - ln 0.75, with a drawn x ...
    f = O.OracleFunctionChains(110, [], {B: 0.0}, 5, 4, canonical=canonical)
    w, _ = f.update({B: 1.0}, argdiff=UNKNOWN)
    assert np.allclose(w, -1.098612, atol=1e-6, rtol=0)
    v, p = f.trace()
    assert np.all(p == 0b11) and len(set(v[:, X])) == 5   # x was drawn, chain by chain
    # ... and with a new, unconstrained sub-call (trace_at)
    f = O.OracleFunctionChains(112, [], {B: 0.0}, 5, 5, canonical=canonical)
    w, _ = f.update({B: 1.0}, argdiff=UNKNOWN)
    assert np.allclose(w, -1.098612, atol=1e-6, rtol=0)
    assert np.all(f.trace()[1] == 0b1111)


def test_residual_constraints_are_the_reference_panic():
    # test_update_residual_constraints_panic (:125-131): a constraint nobody consumes
    f = O.OracleFunctionChains(110, [], {B: 0.0}, 2, 1)
    with pytest.raises(O.OracleError):
        f.update({X: 0.3})   # b stays false: "x" is never visited


def test_hand_composed_mh_equals_the_fused_move():
    """mh.rs:9-40 composed from propose / update / assess with ONE Philox step equals metropolis_hastings itself (the checker's own
    consistency; the device's is tests/test_gpu_gfi.py)."""
    xs = np.arange(-3.0, 4.0)
    ys = 0.2 + 0.5 * xs + 0.3 * xs * xs
    cons = {4 + k: y for k, y in enumerate(ys)}
    n, seed = 40, 9
    a = O.OracleFunctionChains(101, xs, cons, n, seed)
    b = O.OracleFunctionChains(101, xs, cons, n, seed)
    for it in range(1, 5):
        kind, args = (1, [0.2]) if it % 2 else (2, [])
        old_v, old_p = b.trace()
        choices, fwd = b.propose(kind, args, rng_step=it)
        w, discard = b.update(choices, rng_step=it)
        bwd = b.assess(discard, proposal_kind=kind, proposal_args=args, rng_step=it)
        alpha = w - fwd + bwd
        u = np.empty(n)
        for i in range(n):
            tmp = np.empty(1)
            O.load().oracle_u01_stream(seed, i, it, 2, 0, 1, O.dptr(tmp))
            u[i] = tmp[0]
        lnu = np.empty(n)
        O.load().oracle_mp_log(O.dptr(u), n, O.dptr(lnu))
        acc = lnu < alpha
        a.mh(kind, args, 1)
        new_v, new_p = b.trace()
        av, ap = a.trace()
        assert np.array_equal(ap, np.where(acc, new_p, old_p))
        assert np.array_equal(av, np.where(acc[:, None], new_v, old_v))
        assert 0 < acc.sum() < n or it > 1
        # b goes on from a's state: rejected chains back to their old traces (a caller would have kept them)
        b = O.OracleFunctionChains(101, xs, cons, n, seed)
        for k in range(1, it + 1):
            kk, aa = (1, [0.2]) if k % 2 else (2, [])
            b.mh(kk, aa, 1)


@pytest.mark.parametrize("argdiff", [0, UNKNOWN])
def test_product_handlers_on_the_host_update_across_a_structure_change(argdiff):
    """The product's static UPDATE handler (mp_genfn.h, compiled for the host: what k_fn_update runs per lane) against the trie
    engine where a constraint changes the model's STRUCTURE: is_linear false -> true drops coeffs/c inside the sub-call, whose gc
    subtracts that choice's log-density ONCE (dyngenfn.rs:453-470) — until round 4 the outer gc took it a second time; the fused mh
    tests only ever rejected such moves.  And true -> false, where coeffs/c is drawn."""
    xs = np.arange(-3.0, 4.0)
    ys = 0.2 + 0.5 * xs + 0.3 * xs * xs
    for start, flip in ((0.0, 1.0), (1.0, 0.0)):
        cons = {4 + k: y for k, y in enumerate(ys)}
        cons[0] = start
        n, seed = 50, 12
        st = O.HostStaticFunctionChains(101, xs, cons, n, seed)
        dy = O.OracleFunctionChains(101, xs, cons, n, seed, canonical=True)
        w_s, dp_s = st.update({0: flip, 1: 0.25}, argdiff=argdiff, rng_step=3)
        w_d, (dv_d, dp_d) = dy.update({0: flip, 1: 0.25}, argdiff=argdiff, rng_step=3)
        assert np.array_equal(w_s, w_d)
        assert np.array_equal(dp_s, dp_d)
        sv, sp = st.trace(dy.num_sites)
        dv, dp = dy.trace()
        assert np.array_equal(sp, dp) and np.array_equal(sv, dv)
        assert st.panics == 0


def test_generate_simulate_and_importance_over_a_functor_model():
    """The checker's entry points behind mp_fn_generate / mp_fn_simulate / mp_fn_importance_* (gfi.rs:51-55, importance.rs:12-50): assess is
    generate's weight (gfi.rs:85-90), a generate whose constraints are every choice of a simulated trace returns that trace and its
    logjp, and the generic importance_resampling over the hierarchical functor gives the same indices under the literal arithmetic
    (libm, sequential sums, linear-scan categorical) as under the canonical one the GPU evaluates."""
    xs = np.arange(-5.0, 6.0)                                   # modppl/tests/importance.rs:98
    ys = 0.3 + 0.4 * xs + 0.5 * xs * xs + np.random.default_rng(1).normal(0.0, 0.1, xs.size)
    cons = {4 + k: float(y) for k, y in enumerate(ys)}
    n = 400
    r = O.OracleFunctionChains(101, xs, cons, n, 9, canonical=True)
    w = r.generate(cons, rng_step=3)
    assert np.array_equal(w, r.assess(cons, rng_step=3))
    lj = r.simulate(rng_step=4)
    assert np.array_equal(lj, r.logjp())
    tv, tp = r.trace()
    assert np.allclose(r.generate((tv, tp), rng_step=5), lj, rtol=1e-13, atol=1e-12)
    tv2, tp2 = r.trace()
    assert np.array_equal(tp, tp2) and np.array_equal(tv, tv2)
    assert np.all(r.generate({}, rng_step=6) == 0.0)            # nothing constrained: a prior draw, weight 0 (dyngenfn.rs:132-140)
    tr, lnw, lml, idx = O.OracleFunctionChains.importance(101, xs, cons, n, 20, 9, canonical=True)
    tr2, lnw2, lml2, idx2 = O.OracleFunctionChains.importance(101, xs, cons, n, 20, 9, canonical=False)
    assert abs(lml - lml2) <= 1e-12 * abs(lml2) and np.array_equal(idx, idx2)
    assert abs(np.exp(lnw2).sum() - 1.0) < 1e-9
    # the N traces are generate's: chain i of a plain constructor with the same seed
    assert np.array_equal(tr.trace()[0], O.OracleFunctionChains(101, xs, cons, n, 9, canonical=True).trace()[0])
