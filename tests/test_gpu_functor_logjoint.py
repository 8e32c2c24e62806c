"""GPU: an INDEPENDENT statement of the two functor-only MH models (kinds 102 and 103 of modppl_amd/csrc/mp_mh_models.h) —
their log-joint written in numpy from the model's prose, not from the functor — against the device: mp_mh_read_logjp of every
chain, and the accept decision of mh (mh.rs:34-36: accept iff ln u < alpha) on a few hundred chains, with the proposed values
taken from mp_fn_propose and the accept uniform from the Philox stream.  The checker interprets the SAME functor source as the
device for these models (oracle/src/mh_functor_adapter.hpp), so a wrong model BODY is invisible to it; here it is not."""
import numpy as np
import pytest

from tests import oracle_lib as O

pytestmark = pytest.mark.gpu
LN_2PI = np.log(2.0 * np.pi)


def nlp(x, mu, sd):
    return -0.5 * ((x - mu) / sd) ** 2 - 0.5 * LN_2PI - np.log(sd)


def logjoint_robust_line(v, xs, ys):
    """kind 102: slope, intercept ~ normal(0, 2); per point k: is_outlier_k ~ bernoulli(0.1); y_k ~ normal(slope x_k + intercept, is_outlier_k ? 5 : 0.5)
    sites: 0 slope, 1 intercept, 2 + k is_outlier_k (12 slots), 14 + k y_k"""
    slope, icpt = v[:, 0], v[:, 1]
    lj = nlp(slope, 0.0, 2.0) + nlp(icpt, 0.0, 2.0)
    for k, (x, y) in enumerate(zip(xs, ys)):
        out = v[:, 2 + k] != 0.0
        lj = lj + np.where(out, np.log(0.1), np.log(0.9)) + nlp(y, slope * x + icpt, np.where(out, 5.0, 0.5))
    return lj


def logjoint_scaled_line(v, xs, ys):
    """kind 103: big ~ bernoulli(0.3); slope, intercept ~ normal(0, 2); y_k ~ normal(slope x_k + intercept, big ? 2 : 0.5)
    sites: 0 big, 1 slope, 2 intercept, 3 + k y_k"""
    big = v[:, 0] != 0.0
    slope, icpt = v[:, 1], v[:, 2]
    lj = np.where(big, np.log(0.3), np.log(0.7)) + nlp(slope, 0.0, 2.0) + nlp(icpt, 0.0, 2.0)
    for x, y in zip(xs, ys):
        lj = lj + nlp(y, slope * x + icpt, np.where(big, 2.0, 0.5))
    return lj


def accept_uniforms(seed, n, step):
    L = O.load()
    u, tmp = np.empty(n), np.empty(1)
    for i in range(n):
        L.oracle_u01_stream(seed, i, step, 2, 0, 1, O.dptr(tmp))   # (DOM_ACCEPT, site 0) of MH iteration `step`
        u[i] = tmp[0]
    return u


@pytest.mark.parametrize("kind", [102, 103])
def test_logjoint_and_accept_decisions_against_numpy(kind):
    import modppl_amd

    rng = np.random.default_rng(kind)
    if kind == 102:
        xs = np.linspace(-2.5, 2.5, 9)
        ys = 0.8 * xs - 0.3 + rng.normal(0, 0.4, xs.size)
        ys[2] += 7.0
        y0, lj, moves = 14, logjoint_robust_line, [(1, [0.25]), (2, [2.0]), (1, [0.1]), (2, [5.0])]
    else:
        xs = np.linspace(-1.0, 3.0, 7)
        ys = 1.2 * xs + 0.4 + rng.normal(0, 1.1, xs.size)
        y0, lj, moves = 3, logjoint_scaled_line, [(1, []), (2, [0.3]), (1, []), (2, [0.1])]
    n, seed = 600, 41
    g = modppl_amd.FunctionChains(kind, xs, {y0 + k: y for k, y in enumerate(ys)}, n, seed)
    v, p = g.trace()
    assert np.allclose(g.logjp(), lj(v, xs, ys), rtol=1e-12, atol=1e-10)
    g.regen_mh([0, 1, 2], 6, cycle=True)   # move the chains off their prior draws
    n_acc = n_rej = 0
    for it_move, (pk, pa) in enumerate(moves):
        step = g.iterations + 1
        old, _ = g.trace()
        (cv, cp), fwd = g.propose(pk, pa, rng_step=step)
        new = old.copy()
        for k in range(old.shape[1]):
            sel = ((cp >> k) & 1).astype(bool)
            new[sel, k] = cv[sel, k]
        # both proposals are symmetric (a normal drift; a flip proposed with the same probability from either side): alpha = the
        # difference of the log-joints, evaluated here without any of the library's arithmetic
        alpha = lj(new, xs, ys) - lj(old, xs, ys)
        lnu = np.log(accept_uniforms(seed, n, step))
        clear = np.abs(lnu - alpha) > 1e-9   # (a tie within rounding of two different evaluations decides nothing)
        want = lnu < alpha
        got_count = g.mh(pk, pa, 1)
        now, _ = g.trace()
        moved = np.any(now != old, axis=1)
        same_proposal = np.all(new == old, axis=1)   # (a flip that proposes the current value: accepted or not, nothing moves)
        check = clear & ~same_proposal
        assert np.array_equal(moved[check], want[check]), (kind, pk)
        assert np.array_equal(now[moved], new[moved])
        assert abs(got_count - int(want.sum())) <= int((~clear).sum())
        assert np.allclose(g.logjp(), lj(now, xs, ys), rtol=1e-12, atol=1e-10)
        n_acc += int(want[check].sum()); n_rej += int((~want[check]).sum())
    assert n_acc > 50 and n_rej > 50   # both branches of the decision were exercised
