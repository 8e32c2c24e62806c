"""Rank counts of the split multinomial resample: the binomial variates behind them.

* bits: the checker's restatement (oracle/src/inference.hpp: canonical_binomial) against the product's own header
  (modppl_amd/csrc/mp_binomial.h) compiled for the host — same integers, case by case;
* law: against scipy's exact binomial pmf (chi-square over the support, both branches of the sampler: the search from 0
  for n p < 10 and the transformed rejection beyond), the Stirling tails against lgamma, and the multinomial the splitting
  tree yields (its means and covariances over many resample counts).
Runs without a GPU."""
import ctypes as C

import numpy as np
import pytest
from scipy import special, stats

from tests import oracle_lib as O


def _both(n, a, b, node, seed, rc):
    L = O.load()
    n, a, b = (np.ascontiguousarray(v, dtype=np.uint64) for v in (n, a, b))
    node = np.ascontiguousarray(node, dtype=np.uint32)
    r, p = np.zeros(len(n), dtype=np.uint64), np.zeros(len(n), dtype=np.uint64)
    u64p, u32p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
    L.oracle_binomial_both(n.ctypes.data_as(u64p), a.ctypes.data_as(u64p), b.ctypes.data_as(u64p), node.ctypes.data_as(u32p), len(n), seed, rc,
                           r.ctypes.data_as(u64p), p.ctypes.data_as(u64p))
    return r, p


def test_stirling_tails_are_the_tails_of_log_factorial():
    L = O.load()
    for k in list(range(0, 40)) + [100, 1000]:
        want = special.gammaln(k + 1.0) - (0.5 * np.log(2 * np.pi) + (k + 0.5) * np.log(k + 1.0) - (k + 1.0))
        got = L.oracle_stirling_tail(float(k))
        assert abs(got - want) < 2e-9, (k, got, want)
    for k in (10 ** 6, 2 ** 24, 2 ** 32):   # (lgamma cancels here: the leading term of the series instead)
        assert abs(L.oracle_stirling_tail(float(k)) * 12 * (k + 1.0) - 1) < 1e-9


def test_restatement_and_product_header_agree_bit_for_bit():
    rng = np.random.default_rng(7)
    cases = 20000
    n = rng.integers(0, 1 << 24, cases).astype(np.uint64)
    n[:2000] = rng.integers(0, 64, 2000)                       # few trials
    b = rng.integers(1, 1 << 62, cases).astype(np.uint64)
    frac = rng.random(cases)
    frac[2000:4000] = rng.random(2000) * 1e-6                  # n p below 10: the search branch
    frac[4000:4200] = 0.0
    frac[4200:4400] = 1.0
    a = np.minimum((frac * b.astype(np.float64)).astype(np.uint64), b)
    node = rng.integers(1, 128, cases).astype(np.uint32)
    r, p = _both(n, a, b, node, seed=0x1234_5678_9ABC, rc=11)
    assert np.array_equal(r, p)
    assert (r <= n).all()
    assert (r[a == 0] == 0).all() and (r[a == b] == n[a == b]).all()


def test_lane_wise_form_is_the_sequential_sampler():
    """mp_binomial_ratio_lanes (mp_binomial.h): eight attempts side by side, the expensive acceptance test piece by piece and only in
    front of the first attempt the squeeze accepted — what the table kernel's lanes do for the split counts — returns the sequential
    sampler's variate, i.e. the checker's, in every case (including the ones where all eight attempts are rejected: forced below by
    counting how often the test is needed at all)"""
    L = O.load()
    rng = np.random.default_rng(11)
    cases = 200000
    n = rng.integers(1, 1 << 33, cases).astype(np.uint64)
    n[:20000] = rng.integers(0, 4096, 20000)
    b = rng.integers(1, 1 << 62, cases).astype(np.uint64)
    frac = rng.random(cases)
    frac[20000:30000] = rng.random(10000) * 1e-7
    frac[30000:30100] = 0.0
    frac[30100:30200] = 1.0
    a = np.minimum((frac * b.astype(np.float64)).astype(np.uint64), b)
    node = rng.integers(1, 64, cases).astype(np.uint32)
    seed, rc = 0xBEEF_0000_1234, 5
    r, _ = _both(n, a, b, node, seed, rc)
    out = np.zeros(cases, dtype=np.uint64)
    u64p, u32p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
    L.oracle_binomial_lanes(n.ctypes.data_as(u64p), a.ctypes.data_as(u64p), b.ctypes.data_as(u64p), node.ctypes.data_as(u32p), cases, seed, rc,
                            out.ctypes.data_as(u64p))
    assert np.array_equal(out, r)


@pytest.mark.parametrize("n,a,b", [(40, 1, 10), (1 << 20, 1, 1 << 19), (1000, 3, 10), (1 << 20, 1, 8), (1 << 24, 7, 8), (300, 1, 2), (25, 2, 5)])
def test_law_against_the_exact_pmf(n, a, b):
    draws = 40000
    r, _ = _both(np.full(draws, n), np.full(draws, a), np.full(draws, b), np.arange(1, draws + 1), seed=99, rc=3)
    p = a / b
    mean, sd = n * p, np.sqrt(n * p * (1 - p))
    assert abs(r.mean() - mean) < 5 * sd / np.sqrt(draws)
    assert abs(r.std() / sd - 1) < 0.03
    # chi-square over bins of at least ~50 expected draws
    lo, hi = int(max(0, np.floor(mean - 5 * sd))), int(min(n, np.ceil(mean + 5 * sd)))
    edges = [lo]
    acc = 0.0
    for k in range(lo, hi + 1):
        acc += stats.binom.pmf(k, n, p) * draws
        if acc >= 50:
            edges.append(k + 1)
            acc = 0.0
    edges[-1] = hi + 1
    obs, exp = [], []
    cdf = lambda k: stats.binom.cdf(k - 1, n, p)   # noqa: E731  P(X < k)
    obs.append((r < edges[0]).sum()); exp.append(cdf(edges[0]) * draws)
    for e0, e1 in zip(edges[:-1], edges[1:]):
        obs.append(((r >= e0) & (r < e1)).sum()); exp.append((cdf(e1) - cdf(e0)) * draws)
    obs.append((r >= edges[-1]).sum()); exp.append((1 - cdf(edges[-1])) * draws)
    obs, exp = np.array(obs, dtype=float), np.array(exp)
    keep = exp > 5
    chi2 = ((obs[keep] - exp[keep]) ** 2 / exp[keep]).sum() + ((obs[~keep].sum() - exp[~keep].sum()) ** 2 / max(exp[~keep].sum(), 1.0))
    dof = keep.sum()
    assert chi2 < stats.chi2.ppf(1 - 1e-6, dof), (chi2, dof)


@pytest.mark.parametrize("world", [1, 2, 3, 5, 8])
def test_split_counts_are_a_multinomial(world):
    L = O.load()
    rng = np.random.default_rng(world)
    mass = rng.integers(1 << 40, 1 << 44, world).astype(np.uint64)
    if world >= 3:
        mass[1] = 0   # an empty rank gets nothing
    N = 2048 * world
    reps = 3000
    out = np.zeros((reps, world), dtype=np.uint64)
    u64p = C.POINTER(C.c_uint64)
    for rc in range(reps):
        L.oracle_split_counts(mass.ctypes.data_as(u64p), world, N, 4242, rc, out[rc].ctypes.data_as(u64p))
    assert (out.sum(axis=1) == N).all()
    p = mass.astype(np.float64) / mass.astype(np.float64).sum()
    c = out.astype(np.float64)
    for r in range(world):
        sd = np.sqrt(N * p[r] * (1 - p[r]))
        if sd == 0:
            assert (out[:, r] == (0 if p[r] == 0 else N)).all()
            continue
        assert abs(c[:, r].mean() - N * p[r]) < 5 * sd / np.sqrt(reps)
        assert abs(c[:, r].std() / sd - 1) < 0.08
    if world >= 2:   # cov(c_0, c_last) = -N p_0 p_last
        cov = np.cov(c[:, 0], c[:, -1])[0, 1]
        want = -N * p[0] * p[-1]
        assert abs(cov - want) < 0.15 * abs(want) + 1.0
