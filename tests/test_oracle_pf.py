"""CPU: internal consistency of the oracle's particle filter variants and the canonical
(order-free, fixed-point) resampling spec the GPU implements."""
import ctypes as C

import numpy as np
import pytest

from tests import oracle_lib as O


def run(variant, n, T, seed, kind=1, ds=1, do=1, params=O.LGSSM_PARAMS, ys=None, args0=None):
    ys = O.lgssm_observations(T) if ys is None else ys
    pf = O.OraclePF(kind, ds, do, params, n, seed, variant)
    pf.init_step(ys[:1], args0)
    Ls, pars = [], []
    for t in range(1, T):
        Ls.append(pf.resample())
        pars.append(pf.parents().copy())
        pf.step(ys[t:t + 1])
    return dict(L=np.array(Ls), par=np.array(pars), x=pf.state().copy(), lw=pf.log_weights().copy(),
                lml=pf.log_marginal_likelihood_estimate(), pf=pf)


def test_structure_faithful_equals_algorithm_faithful():
    """Dynamic-handler ParticleSystem (string-keyed tries, O(N) categorical scan) vs the SoA engine."""
    for canon in (0, 1):
        a = run(canon, 300, 12, 5)
        b = run(canon | O.VARIANT_SOA, 300, 12, 5)
        assert np.array_equal(a["par"], b["par"])
        assert np.array_equal(a["x"], b["x"]) and np.array_equal(a["lw"], b["lw"])
        assert a["lml"] == b["lml"] and np.array_equal(a["L"], b["L"])


def test_fast_search_is_index_identical_to_linear_scan():
    a = run(0, 500, 10, 9)
    b = run(O.VARIANT_FAST_SEARCH, 500, 10, 9)
    assert np.array_equal(a["par"], b["par"]) and a["lml"] == b["lml"]


def test_literal_vs_canonical_agree():
    """libm + sequential fp64 sums (the reference's arithmetic) vs mp_math + fixed-point CDF."""
    a = run(O.VARIANT_SOA, 20000, 20, 77)
    b = run(O.VARIANT_SOA | O.VARIANT_CANONICAL, 20000, 20, 77)
    assert abs(a["lml"] - b["lml"]) <= 1e-12 * abs(a["lml"])
    mism = int((a["par"] != b["par"]).sum())
    assert mism <= 2  # expected ~ N*sqrt(N)*eps per draw: essentially never at this size
    assert np.allclose(a["x"], b["x"], rtol=0, atol=1e-13) or mism > 0


def test_c1_config_against_kalman():
    """BASELINE.json configs[0]: LGSSM d=1, T=50, 1k particles on the CPU path; Kalman ground truth."""
    ys = O.lgssm_observations(50)
    exact = O.kalman_log_ml(ys)
    est = [run(O.VARIANT_SOA, 1000, 50, s, ys=ys)["lml"] for s in range(8)]
    assert abs(np.mean(est) - exact) < 0.15
    assert np.std(est) < 0.3
    big = run(O.VARIANT_SOA | O.VARIANT_CANONICAL, 100000, 50, 3, ys=ys)["lml"]
    assert abs(big - exact) < 0.05


def test_stale_ess_quirk():
    """particle_filter.rs:98-100 reads buffers only refreshed inside resample(): 1/N before any."""
    for variant in (0, O.VARIANT_SOA, O.VARIANT_SOA | O.VARIANT_CANONICAL):
        pf = O.OraclePF(1, 1, 1, O.LGSSM_PARAMS, 64, 1, variant)
        assert pf.effective_sample_size() == pytest.approx(1.0 / 64, rel=1e-15)
        pf.init_step([0.5])
        assert pf.effective_sample_size() == pytest.approx(1.0 / 64, rel=1e-15)
        pf.resample()
        ess = pf.effective_sample_size()
        assert 1.0 <= ess <= 64.0
        pf.step([0.1])
        assert pf.effective_sample_size() == ess  # still the value of the last resample


def test_canonical_spec_pieces(oracle):
    """level 0 per tile of 2048 (51-bit fixed point relative to the tile max), level 1 over tiles."""
    rng = np.random.default_rng(0)
    n = 5000
    lw = rng.normal(-1.0, 2.0, n)
    L, ess, Q = C.c_double(), C.c_double(), C.c_uint64()
    cum = np.empty(n, dtype=np.uint64)
    rc = oracle.oracle_canonical_normalize(O.dptr(lw), n, n, C.byref(L), C.byref(ess), C.byref(Q), cum.ctypes.data_as(C.POINTER(C.c_uint64)))
    assert rc == 0
    assert Q.value < 2 ** 63
    for b in range(3):   # tiles [0,2048), [2048,4096), [4096,5000): tile-local inclusive prefixes
        seg = cum[b * 2048:(b + 1) * 2048].astype(object)
        assert np.all(np.diff(seg) >= 0)
        q = np.diff(np.concatenate([[0], seg]))
        assert int(q.max()) == 2 ** 51                     # the tile's max particle gets exactly 2^51
        w = np.exp(lw[b * 2048:(b + 1) * 2048] - lw[b * 2048:(b + 1) * 2048].max())
        assert np.allclose(q.astype(float) / 2 ** 51, w, rtol=0, atol=2 ** -50)
    ref = float(np.logaddexp.reduce(lw))
    assert abs(L.value - ref) < 1e-12
    w = np.exp(lw - ref)
    assert abs(ess.value - 1.0 / np.sum(w * w)) < 1e-9 * ess.value
    # target rule: max(1, ceil(k Q / 2^52)); k = 0 -> 1 (first positive-weight particle), k = 2^52-1 -> <= Q
    assert oracle.oracle_canonical_target(0, Q.value) == 1
    assert oracle.oracle_canonical_target(2 ** 52 - 1, Q.value) <= Q.value
    for k in (1, 12345, 2 ** 51, 2 ** 52 - 1):
        assert oracle.oracle_canonical_target(k, Q.value) == max(1, -((-k * Q.value) // 2 ** 52))
    # the scale S = 62 - ceil(log2 N_global) only enters level 1: a larger job changes Q, not L
    rc = oracle.oracle_canonical_normalize(O.dptr(lw), n, 8 * n, C.byref(L), C.byref(ess), C.byref(Q), None)
    assert abs(L.value - ref) < 1e-11


def test_degenerate_weights_status():
    pf = O.OraclePF(1, 1, 1, O.LGSSM_PARAMS, 32, 1, O.VARIANT_SOA | O.VARIANT_CANONICAL)
    pf.init_step([np.inf])
    with pytest.raises(O.OracleError) as e:
        pf.resample()
    assert e.value.code == 4
    pf2 = O.OraclePF(1, 1, 1, O.LGSSM_PARAMS, 32, 1, 0)
    with pytest.raises(O.OracleError) as e:
        pf2.step([0.0])
    assert e.value.code == 2


def test_spiral_model_runs_like_smc_rs():
    """modppl/tests/smc.rs:48-91 shape: spiral Unfold, 500 particles x 20 steps, resample each step."""
    T, n = 20, 500
    ang = 0.7
    obs = np.array([[0.4 * np.cos(2 * np.pi * t / T + ang), 0.4 * np.sin(2 * np.pi * t / T + ang)] for t in range(T)])
    a = run(0 | O.VARIANT_FAST_SEARCH, n, T, 3, kind=2, ds=2, do=2, params=np.zeros(0), ys=obs, args0=[0.0, 0.0])
    b = run(O.VARIANT_SOA, n, T, 3, kind=2, ds=2, do=2, params=np.zeros(0), ys=obs, args0=[0.0, 0.0])
    assert np.array_equal(a["par"], b["par"]) and np.array_equal(a["x"], b["x"])
    assert np.isfinite(a["lml"])
    traj = a["pf"].trajectory(0)
    assert traj.shape == (T, 2)  # traces[i].retv: one state per step (smc.rs:67 reads .last())


@pytest.mark.parametrize("scheme", [1, 2])
def test_lattice_schemes_properties(scheme):
    """Systematic (1) / stratified (2) resampling of the canonical checker (extensions, no reference counterpart): sorted
    parents, offspring counts within 1 (2) of N w_i, unbiased-by-construction targets in [1, Q]; the literal engine
    refuses them."""
    n, seed = 5000, 12
    ys = O.lgssm_observations(5)
    pf = O.OraclePF(1, 1, 1, O.LGSSM_PARAMS, n, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA)
    pf.init_step(ys[:1])
    for t in range(1, 5):
        w = pf.log_weights().copy()
        pf.resample(scheme)
        par = pf.parents().astype(np.int64)
        assert np.all(np.diff(par) >= 0) and par.min() >= 0 and par.max() < n
        p = np.exp(w - np.logaddexp.reduce(w))
        counts = np.bincount(par, minlength=n)
        assert np.all(np.abs(counts - n * p) < (1.0 if scheme == 1 else 2.0) + 1e-9 * n)
        pf.step(ys[t:t + 1])
    lit = O.OraclePF(1, 1, 1, O.LGSSM_PARAMS, 100, seed, O.VARIANT_SOA)
    lit.init_step(ys[:1])
    with pytest.raises(O.OracleError):
        lit.resample(scheme)


def test_full_size_literal_vs_canonical_indices():
    """BASELINE config 2 size on the CPU: the reference's arithmetic (libm, sequential fp64 running-sum CDF, binary search
    over it) and the canonical fixed-point spec the GPU implements choose the same parents for every one of 3 x 2^20
    draws, and agree on the log total weight to 1e-14 relative.  (States differ in their last bits: libm vs mp_*.)"""
    n, T = 1 << 20, 4
    ys = O.lgssm_observations(T)
    lit = O.OraclePF(1, 1, 1, O.LGSSM_PARAMS, n, 20241008, O.VARIANT_SOA | O.VARIANT_FAST_SEARCH)
    can = O.OraclePF(1, 1, 1, O.LGSSM_PARAMS, n, 20241008, O.VARIANT_SOA | O.VARIANT_CANONICAL)
    lit.init_step(ys[:1])
    can.init_step(ys[:1])
    for t in range(1, T):
        La, Lb = lit.resample(), can.resample()
        assert abs(La - Lb) <= 1e-14 * abs(La)
        assert np.array_equal(lit.parents(), can.parents())
        lit.step(ys[t:t + 1])
        can.step(ys[t:t + 1])
    a, b = lit.log_marginal_likelihood_estimate(), can.log_marginal_likelihood_estimate()
    assert abs(a - b) <= 1e-12 * abs(a)


def test_simulate_moments():
    """DynUnfold::simulate of the checker: every site sampled, the LGSSM's stationary structure in the moments."""
    xs, ys = O.unfold_simulate(1, 1, 1, O.LGSSM_PARAMS, 4, 40000, 3)
    assert xs.shape == (40000, 4, 1) and ys.shape == (40000, 4, 1)
    assert abs(xs[:, 0, 0].std() - 1.0) < 0.02 and abs((ys[:, 0, 0] - xs[:, 0, 0]).std() - 1.0) < 0.02
    assert abs((xs[:, 1, 0] - 0.9 * xs[:, 0, 0]).std() - 0.5) < 0.01
    a, b = O.unfold_simulate(1, 1, 1, O.LGSSM_PARAMS, 4, 100, 3)
    assert np.array_equal(a, xs[:100]) and np.array_equal(b, ys[:100])   # trace i depends on (seed, i) only
