"""GPU: GenFn::update / regenerate / assess / propose ONE AT A TIME on the device (mp_fn_* of the C ABI, modppl/src/gfi.rs:57-90),
for chains of registered generative functions:
  * the reference's own known answers for `update` (modppl/tests/dyngenfn.rs:55-114) on the device;
  * every weight, discard, choice table and resulting trace against the checker's dynamic machinery (tries, sample_at / trace_at /
    gc: tests/oracle_lib.py OracleFunctionChains), bit for bit;
  * mh.rs:9-40 composed by hand from propose / update / assess equals the fused mp_mh_step."""
import numpy as np
import pytest

from tests import oracle_lib as O

pytestmark = pytest.mark.gpu
B, X = 0, 1
UNKNOWN = 1


def _pair(kind, params, cons, n, seed):
    import modppl_amd

    return modppl_amd.FunctionChains(kind, params, cons, n, seed), O.OracleFunctionChains(kind, params, cons, n, seed, canonical=True)


def _same_trace(g, r):
    gv, gp = g.trace()
    rv, rp = r.trace()
    assert np.array_equal(gp, rp)
    assert np.array_equal(gv, rv)


def test_reference_update_known_answers_on_the_device():
    n = 257
    g, r = _pair(110, [], {B: 1.0, X: 0.0}, n, 1)
    w, (dv, dp) = g.update({X: 1.0}, argdiff=UNKNOWN)
    assert np.all(w == -0.5)                                   # dyngenfn.rs:65: assert_eq!(w, -0.5)
    assert np.all(dp == (1 << X)) and np.all(dv[:, X] == 0.0)
    g, r = _pair(110, [], {B: 0.0}, n, 2)
    w, (dv, dp) = g.update({B: 1.0, X: 1.0}, argdiff=UNKNOWN)
    assert np.allclose(w, -2.517551, atol=1e-6, rtol=0)        # :78
    assert np.array_equal(w, r.update({B: 1.0, X: 1.0}, argdiff=UNKNOWN)[0])
    g, r = _pair(111, [], {0: 1.0, 1: 1.0, 2: -0.3}, n, 3)
    w, _ = g.update({0: 0.5}, argdiff=UNKNOWN)
    assert np.allclose(w, 0.4, atol=1e-6, rtol=0)              # :92
    assert np.array_equal(w, r.update({0: 0.5}, argdiff=UNKNOWN)[0])
    for kind in (110, 112):                                    # :104 (sample_at), :113 (trace_at)
        g, r = _pair(kind, [], {B: 0.0}, n, 4)
        w, _ = g.update({B: 1.0}, argdiff=UNKNOWN)
        assert np.allclose(w, -1.098612, atol=1e-6, rtol=0)
        assert np.array_equal(w, r.update({B: 1.0}, argdiff=UNKNOWN)[0])
        _same_trace(g, r)                                      # the drawn x / sub-call choices too


def test_residual_constraints_are_an_error_on_the_device():
    import modppl_amd

    g, _ = _pair(110, [], {B: 0.0}, 64, 1)
    with pytest.raises(modppl_amd.capi.ModpplError):
        g.update({X: 0.3})


@pytest.mark.parametrize("kind", [101, 102, 103, 113])
def test_gfi_calls_against_the_dynamic_interpretation(kind):
    """A scripted sequence of update (shared and per-chain constraints, both ArgDiffs), regenerate (masks inside and outside
    sub-calls, the empty mask), propose and assess on three registered models: weights, discards, choices, traces."""
    rng = np.random.default_rng(kind)
    if kind == 101:
        xs = np.arange(-3.0, 4.0); y0 = 4
        cons = {y0 + k: v for k, v in enumerate(0.2 + 0.5 * xs + 0.3 * xs * xs)}
        free, props = [1, 2], [(1, [0.15]), (2, [])]   # (coeffs/a, coeffs/b: coeffs/c leaves the trace when a move turns a chain linear, and a
                                                       # constraint on a site the model does not visit is the reference's panic)
    elif kind == 102:
        xs = np.linspace(-2, 2, 8); y0 = 2 + 12
        ys = 0.7 * xs - 0.2 + rng.normal(0, 0.4, xs.size); ys[3] += 6.0
        cons = {y0 + k: v for k, v in enumerate(ys)}
        free, props = [0, 1, 2, 5], [(1, [0.2]), (2, [3.0])]
    elif kind == 113:   # two levels of sub-calls: a | mid { b | inner { c, f } | d if f | e } | y
        xs = np.array([0.5, -1.0, 1.5, 2.0]); y0 = 6
        cons = {y0 + k: v for k, v in enumerate(1.3 * xs + rng.normal(0, 0.2, xs.size))}
        free, props = [0, 2, 1, 5], [(1, [0.2]), (2, []), (3, [0.4])]
    else:
        xs = np.linspace(-1, 3, 6); y0 = 3
        cons = {y0 + k: v for k, v in enumerate(1.1 * xs + 0.3 + rng.normal(0, 0.6, xs.size))}
        free, props = [0, 1, 2], [(1, []), (2, [0.25])]
    n, seed = 300, 17
    g, r = _pair(kind, xs, cons, n, seed)
    _same_trace(g, r)
    step = 1
    for rnd in range(3):
        for diff in (0, UNKNOWN):
            # update: one free site pinned to a shared value
            site = free[(rnd + diff) % len(free)]
            val = 0.0 if kind in (102, 103) and site in (2, 5) and kind == 102 else float(rng.normal())
            if kind == 103 and site == 0:
                val = float(rnd % 2)
            gw, (gdv, gdp) = g.update({site: val}, argdiff=diff, rng_step=step)
            rw, (rdv, rdp) = r.update({site: val}, argdiff=diff, rng_step=step)
            assert np.array_equal(gw, rw), (kind, "update", site, diff)
            assert np.array_equal(gdp, rdp) and np.array_equal(gdv, rdv)
            _same_trace(g, r)
            step += 1
        # regenerate: a site, two sites, the whole schema
        for mask in ([free[rnd % len(free)]], free[:2], []):
            gw = g.regenerate(mask, rng_step=step)
            rw = r.regenerate(mask, rng_step=step)
            assert np.array_equal(gw, rw), (kind, "regenerate", mask)
            _same_trace(g, r)
            step += 1
            if not mask:   # (the whole schema re-simulates the observed sites too: put the observations back)
                gw, _ = g.update(cons, rng_step=step)
                rw, _ = r.update(cons, rng_step=step)
                assert np.array_equal(gw, rw)
                _same_trace(g, r)
                step += 1
        # propose -> update with the choices -> assess the discard under the proposal (the three calls of mh.rs:17-27)
        for pk, pa in props:
            (gcv, gcp), gf = g.propose(pk, pa, rng_step=step)
            (rcv, rcp), rf = r.propose(pk, pa, rng_step=step)
            assert np.array_equal(gf, rf) and np.array_equal(gcp, rcp) and np.array_equal(gcv, rcv)
            gw, gd = g.update((gcv, gcp), rng_step=step)
            rw, rd = r.update((rcv, rcp), rng_step=step)
            assert np.array_equal(gw, rw) and np.array_equal(gd[1], rd[1]) and np.array_equal(gd[0], rd[0])
            gb = g.assess(gd, proposal_kind=pk, proposal_args=pa, rng_step=step)
            rb = r.assess(rd, proposal_kind=pk, proposal_args=pa, rng_step=step)
            assert np.array_equal(gb, rb)
            _same_trace(g, r)
            step += 1
    # the model's own assess: every choice of the current traces as constraints = trace.logjp
    tv, tp = g.trace()
    assert np.array_equal(g.assess((tv, tp), rng_step=step), r.assess((tv, tp), rng_step=step))
    assert np.allclose(g.assess((tv, tp), rng_step=step), g.logjp(), rtol=1e-13, atol=1e-12)


def test_a_sub_call_skipped_on_a_later_visit_leaves_with_its_running_weight_on_the_device():
    """kind 112 (modppl/tests/dyngenfn.rs:44-53), b flipped true -> false -> true -> false through mp_fn_update with the sub-trace given
    a history in between: the enclosing gc removes a sub-call the body no longer reaches with that sub-trie's RUNNING weight
    (trie.rs:161-184, dyngenfn.rs:454-470), as the trie engine does — weights, discards and traces bit for bit (the CPU suite holds the
    host-compiled handlers to the same script: tests/test_oracle_mh_functor.py)."""
    n, seed = 3000, 3
    g, r = _pair(112, [], {0: 1.0}, n, seed)
    _same_trace(g, r)
    script = [({1: 0.7}, 1), ({2: -0.4}, 0), ({0: 0.0}, 0), ({0: 1.0}, 0), ({3: 1.9}, 1), ({0: 0.0}, 1), ({0: 1.0}, 1), ({1: 0.1, 3: -0.2}, 0), ({0: 0.0}, 0)]
    for k, (cons, diff) in enumerate(script):
        gw, (gdv, gdp) = g.update(cons, argdiff=diff, rng_step=20 + k)
        rw, (rdv, rdp) = r.update(cons, argdiff=diff, rng_step=20 + k)
        assert np.array_equal(gw, rw), (k, cons, diff)
        assert np.array_equal(gdp, rdp) and np.array_equal(gdv, rdv)
        _same_trace(g, r)


def test_hand_composed_mh_equals_mp_mh_step():
    import modppl_amd

    xs = np.arange(-3.0, 4.0)
    cons = {4 + k: y for k, y in enumerate(0.2 + 0.5 * xs + 0.3 * xs * xs)}
    n, seed = 4096, 23
    fused = modppl_amd.FunctionChains(101, xs, cons, n, seed)
    L = O.load()
    for it in range(1, 6):
        kind, args = (1, [0.2]) if it % 2 else (2, [])
        parts = modppl_amd.FunctionChains(101, xs, cons, n, seed)   # replay the fused chain up to here, then take the move apart
        for k in range(1, it):
            kk, aa = (1, [0.2]) if k % 2 else (2, [])
            parts.mh(kk, aa, 1)
        old_v, old_p = parts.trace()
        choices, fwd = parts.propose(kind, args, rng_step=it)
        w, discard = parts.update(choices, rng_step=it)
        bwd = parts.assess(discard, proposal_kind=kind, proposal_args=args, rng_step=it)
        alpha = w - fwd + bwd                                      # mh.rs:34
        u = np.empty(n)
        tmp = np.empty(1)
        for i in range(n):
            L.oracle_u01_stream(seed, i, it, 2, 0, 1, O.dptr(tmp))   # the accept uniform: (DOM_ACCEPT, site 0) of iteration `it`
            u[i] = tmp[0]
        lnu = np.empty(n)
        L.oracle_mp_log(O.dptr(u), n, O.dptr(lnu))
        acc = lnu < alpha
        assert fused.mh(kind, args, 1) == int(acc.sum())
        new_v, new_p = parts.trace()
        fv, fp = fused.trace()
        assert np.array_equal(fp, np.where(acc, new_p, old_p))
        assert np.array_equal(fv, np.where(acc[:, None], new_v, old_v))
