"""GPU parity of the resample whose row lookups are deferred (k_draw_slots leaves {target, start row} per slot; the next
k_propagate, or k_resolve_slots when the host asks first, looks the parents up — mp_pf.hip `deferred` / `parents_deferred`).
Every order in which a caller can interleave resample / step / reads must give the checker's values, bit for bit:
particle_filter.rs:103-116 (`resample`), :73-96 (`step` keeps `parents`)."""
import numpy as np
import pytest

from modppl_amd import capi
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu


def _pair(d, n, seed, T):
    import modppl_amd

    if d == 1:
        model, obs, kind, params = modppl_amd.lgssm_model(*O.LGSSM_PARAMS), O.lgssm_observations(T).reshape(T, 1), 1, O.LGSSM_PARAMS
    else:
        model, obs = modppl_amd.lgssm_band_model(d), np.random.default_rng(11).normal(0, 1.2, size=(T, d))
        kind, params = 5, np.array([d, 0.9, 0.05, 1.0, 0.5, 1.0])
    pf = modppl_amd.ParticleSystem(model, n, seed)
    ref = O.OraclePF(kind, d, d, params, n, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA)
    pf.init_step(None, obs[:1])
    ref.init_step(obs[:1])
    return pf, ref, obs


@pytest.mark.parametrize("d,n", [(1, 6001), (1, 2048), (4, 5000), (16, 3000)])
def test_two_resamples_in_a_row_then_step(d, n):
    """resample; resample (the second one needs level 0 of the zero weights: the first one's draws are looked up by
    k_resolve_slots, not by a propagate); then a step consumes the second one's draws."""
    pf, ref, obs = _pair(d, n, 21, 4)
    for t in range(1, 4):
        pf.resample(sync=False)
        ref.resample()
        pf.resample(sync=False)
        ref.resample()
        assert np.array_equal(pf.parents, ref.parents())
        pf.step(obs[t:t + 1])
        ref.step(obs[t:t + 1])
        assert np.array_equal(pf.states(), ref.state())
        assert np.array_equal(pf.log_weights, ref.log_weights())
    assert pf.log_marginal_likelihood_estimate() == ref.log_marginal_likelihood_estimate()


@pytest.mark.parametrize("d", [1, 4])
def test_schemes_alternate_without_a_read_in_between(d):
    """multinomial (deferred lookups) and systematic / stratified (single kernel) resamples alternate; nothing is read until the end
    of each round, so every hand-over between the two forms goes through the library's own bookkeeping."""
    pf, ref, obs = _pair(d, 7000, 5, 7)
    schemes = [0, 1, 0, 2, 0, 0]
    for t, sc in zip(range(1, 7), schemes):
        pf.resample(scheme=sc, sync=False)
        ref.resample(sc)
        pf.step(obs[t:t + 1])
        ref.step(obs[t:t + 1])
        assert np.array_equal(pf.parents, ref.parents()), f"t={t} scheme={sc}"
    assert np.array_equal(pf.states(), ref.state())
    assert np.array_equal(pf.log_weights, ref.log_weights())
    assert pf.log_marginal_likelihood_estimate() == ref.log_marginal_likelihood_estimate()


@pytest.mark.parametrize("d", [1, 4])
def test_several_steps_in_one_call_after_a_resample(d):
    """step(obs of 3 time steps) after a resample: only the first of the three launches looks draws up; the row tables swap once."""
    pf, ref, obs = _pair(d, 4500, 9, 8)
    pf.resample(sync=False)
    ref.resample()
    pf.step(obs[1:4])
    ref.step(obs[1:4])
    want = ref.parents().copy()
    pf.resample(sync=False)
    ref.resample()
    pf.step(obs[4:6])
    ref.step(obs[4:6])
    assert np.array_equal(pf.states(), ref.state())
    assert np.array_equal(pf.log_weights, ref.log_weights())
    assert np.array_equal(pf.parents, ref.parents())
    assert not np.array_equal(ref.parents(), want)   # (the second resample's parents, not the first's)


def test_reads_between_resample_and_step_do_not_disturb_the_step():
    """states / log-weights / ESS / log-ML read right after a resample materialise slot order (k_resolve_slots); the step that
    follows must then run as a plain step on those states."""
    pf, ref, obs = _pair(1, 9000, 2, 6)
    for t in range(1, 6):
        pf.resample(sync=False)
        ref.resample()
        if t % 2:
            assert np.array_equal(pf.states(), ref.state())
            assert np.all(pf.log_weights == 0.0)
        else:
            assert pf.log_marginal_likelihood_estimate() == ref.log_marginal_likelihood_estimate()
        pf.step(obs[t:t + 1])
        ref.step(obs[t:t + 1])
        assert np.array_equal(pf.parents, ref.parents())
    assert np.array_equal(pf.states(), ref.state())
    assert np.array_equal(pf.log_weights, ref.log_weights())


def test_draws_made_by_the_step_equal_draws_made_by_the_resample(monkeypatch, diag):
    """An asynchronous multinomial resample of a filter whose k_propagate can draw (two adjacent slots per lane, d = 1) enqueues
    nothing: the next step makes the draws for its own slots.  MP_FUSED_DRAWS=0 (read at creation) keeps the k_draw_slots
    launch.  Same Philox blocks, same targets, same walk: every value must agree, bit for bit, and with the checker."""
    import modppl_amd

    n, seed, T = 70001, 31, 6
    obs = O.lgssm_observations(T).reshape(T, 1)
    fused = modppl_amd.ParticleSystem(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), n, seed)
    monkeypatch.setenv("MP_FUSED_DRAWS", "0")
    plain = modppl_amd.ParticleSystem(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), n, seed)
    monkeypatch.delenv("MP_FUSED_DRAWS")
    ref = O.OraclePF(1, 1, 1, O.LGSSM_PARAMS, n, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA)
    for pf in (fused, plain):
        pf.init_step(None, obs[:1])
    ref.init_step(obs[:1])
    for t in range(1, T):
        for pf in (fused, plain):
            pf.resample(sync=False)
            pf.step(obs[t:t + 1])
        ref.resample()
        ref.step(obs[t:t + 1])
        assert np.array_equal(fused.parents, plain.parents)
        assert np.array_equal(fused.parents, ref.parents())
    assert np.array_equal(fused.states(), plain.states())
    assert np.array_equal(fused.log_weights, ref.log_weights())
    assert fused.effective_sample_size() == plain.effective_sample_size() == ref.effective_sample_size(0)
    assert fused.log_marginal_likelihood_estimate() == plain.log_marginal_likelihood_estimate() == ref.log_marginal_likelihood_estimate()


def _wide_case(name, T):
    """(model, observations) of the two-slot-lane kernels with wider states / more sites: fused draws since round 3"""
    import modppl_amd
    from tests.test_gpu_pf_models import bearings_obs, spiral_obs

    if name == "bearings":      # d = 4, four normal sites at t = 0: the lane-local queue path of k_propagate
        return modppl_amd.bearings_model(), bearings_obs(T).reshape(T, 1)
    if name == "band4":         # d = 4, wave-cooperative deviates... four sites: the queue path again, dense-ish weights
        return modppl_amd.lgssm_band_model(4), np.random.default_rng(11).normal(0, 1.2, size=(T, 4))
    if name == "band2":         # d = 2, two sites: the straight-line path with a gathered parent state
        return modppl_amd.lgssm_band_model(2), np.random.default_rng(12).normal(0, 1.2, size=(T, 2))
    if name == "spiral":        # d = 2, uniform sites at t = 0, an mvnormal observation
        return modppl_amd.spiral_model(), spiral_obs(T)
    raise KeyError(name)


@pytest.mark.parametrize("name,n", [("bearings", 70001), ("band4", 70001), ("band2", 70001), ("spiral", 70001),
                                    ("bearings", (1 << 21) + 4096 + 7), ("band2", (1 << 22))])
def test_wide_models_draw_for_themselves_too(name, n, monkeypatch, diag):
    """Every kernel whose lanes own two adjacent slots (1024 threads per 2048-slot tile: d <= 4, at most four normal sites) makes
    the previous resample's draws itself, up to 2048 tiles (two table entries per thread beyond 1024: the TAB2 instantiation).
    Against MP_FUSED_DRAWS=0 (k_draw_slots + the deferred lookups): same Philox blocks, same targets, same walk — parents, states,
    log-weights, ESS and log-ML bit for bit; a read between resample and step (flush_draws) included."""
    import modppl_amd

    T, seed = 5, 17
    model, obs = _wide_case(name, T)
    fused = modppl_amd.ParticleSystem(model, n, seed)
    monkeypatch.setenv("MP_FUSED_DRAWS", "0")
    plain = modppl_amd.ParticleSystem(model, n, seed)
    monkeypatch.delenv("MP_FUSED_DRAWS")
    args0 = [0.0, 0.0] if name == "spiral" else None
    for pf in (fused, plain):
        pf.init_step(args0, obs[:1])
    for t in range(1, T):
        for pf in (fused, plain):
            pf.resample(sync=False)
            if t == 3:
                pf.parents   # something other than a step comes first: the draws are flushed into k_draw_slots
            pf.step(obs[t:t + 1])
        assert np.array_equal(fused.parents, plain.parents), (name, t)
    assert np.array_equal(fused.states(), plain.states())
    assert np.array_equal(fused.log_weights, plain.log_weights)
    assert fused.effective_sample_size(fresh=True) == plain.effective_sample_size(fresh=True)
    assert fused.log_marginal_likelihood_estimate() == plain.log_marginal_likelihood_estimate()


@pytest.mark.parametrize("scheme", [1, 2])
@pytest.mark.parametrize("name,n", [("lgssm1", 70001), ("lgssm1", 1 << 20), ("lgssm1", (1 << 21) + 2048), ("bearings", 70001), ("band2", 40000),
                                    ("bearings", (1 << 21) + 6144)])
def test_lattice_draws_made_by_the_step(name, n, scheme, monkeypatch, diag):
    """Systematic (1) and stratified (2) resampling, asynchronous: the next k_propagate makes these draws too (its LAT
    instantiation: targets from the lattice function, no Philox block per lane; both table forms).  Against
    MP_FUSED_DRAWS=0 (k_draw_slots<., scheme>) bit for bit, and for the LGSSM against the canonical checker; schemes alternate
    with multinomial in one run."""
    import modppl_amd

    T, seed = 6, 23
    if name == "lgssm1":
        model, obs = modppl_amd.lgssm_model(*O.LGSSM_PARAMS), O.lgssm_observations(T).reshape(T, 1)
    else:
        model, obs = _wide_case(name, T)
    fused = modppl_amd.ParticleSystem(model, n, seed)
    monkeypatch.setenv("MP_FUSED_DRAWS", "0")
    plain = modppl_amd.ParticleSystem(model, n, seed)
    monkeypatch.delenv("MP_FUSED_DRAWS")
    ref = O.OraclePF(1, 1, 1, O.LGSSM_PARAMS, n, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA) if name == "lgssm1" else None
    for pf in (fused, plain):
        pf.init_step(None, obs[:1])
    if ref:
        ref.init_step(obs[:1])
    for t in range(1, T):
        sch = scheme if t != 3 else 0   # a multinomial resample in between: the same handle switches kernels
        for pf in (fused, plain):
            pf.resample(sch, sync=False)
            pf.step(obs[t:t + 1])
        assert np.array_equal(fused.parents, plain.parents), (name, t)
        if ref:
            ref.resample(sch)
            want = ref.parents().copy()
            ref.step(obs[t:t + 1])
            assert np.array_equal(fused.parents, want), (name, t)
    assert np.array_equal(fused.states(), plain.states())
    assert np.array_equal(fused.log_weights, plain.log_weights)
    assert fused.log_marginal_likelihood_estimate() == plain.log_marginal_likelihood_estimate()
    if ref:
        assert np.array_equal(fused.states(), ref.state())
        assert fused.log_marginal_likelihood_estimate() == ref.log_marginal_likelihood_estimate()


@pytest.mark.parametrize("n", [1 << 20, 1 << 21, (1 << 21) + 2048, (1 << 22), (1 << 22) + 2048])
def test_timed_path_at_the_timed_size(n):
    """The path bench.py TIMES — step; resample(sync=False) back to back, nothing read in between — against the canonical checker
    at the bench's population (2^20: 512 tiles), on both sides of 1024 tiles (up to there one table entry per thread of the
    drawing k_propagate, beyond it two: the TAB2 instantiation) and on both sides of 2048 tiles: up to 2^22 particles an
    asynchronous multinomial resample enqueues nothing and the next k_propagate makes the draws for its own slots from a tile
    table it builds in LDS; beyond, k_draw_slots makes them against the table one workgroup builds (mp_pf.hip: local_table,
    launch_draws).  Parents and states are read only after the step that consumed the draws."""
    pf, ref, obs = _pair(1, n, 20241008, 4)
    for t in range(1, 4):
        pf.resample(sync=False)
        pf.step(obs[t:t + 1])
        ref.resample()
        want = ref.parents().copy()
        ref.step(obs[t:t + 1])
        assert np.array_equal(pf.parents, want), f"t={t}"
    assert np.array_equal(pf.states(), ref.state())
    assert np.array_equal(pf.log_weights, ref.log_weights())
    assert pf.effective_sample_size() == ref.effective_sample_size(0)
    assert pf.log_marginal_likelihood_estimate() == ref.log_marginal_likelihood_estimate()


def test_degenerate_population_on_the_fused_path():
    """All log-weights -inf (an observation no particle can explain): the synchronous resample raises where the reference's
    categorical asserts (categorical.rs:23); an ASYNCHRONOUS resample of a filter that makes its draws inside the next step has
    enqueued nothing yet, so `synchronize()` right after it has nothing to report — the step that consumes the draws does
    (include/modppl_hip.h, mp_pf_resample)."""
    import modppl_amd
    from modppl_amd import capi

    obs = O.lgssm_observations(3).reshape(3, 1)
    bad = np.array([[1e200]])   # (y - x)^2 overflows: every log-weight is -inf
    pf = modppl_amd.ParticleSystem(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), 70001, 3)
    pf.init_step(None, obs[:1])
    pf.resample(sync=False)
    pf.step(bad)
    pf.resample(sync=False)
    pf.synchronize()                 # nothing of that resample has run
    pf.step(obs[1:2])                # its draws are made here
    with pytest.raises(capi.ModpplError):
        pf.synchronize()
    pf2 = modppl_amd.ParticleSystem(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), 70001, 3)
    pf2.init_step(None, obs[:1])
    pf2.resample(sync=False)
    pf2.step(bad)
    with pytest.raises(capi.ModpplError):
        pf2.resample()               # synchronous: reported at once


@pytest.mark.parametrize("n", [4097, 6143, 2048 * 5, 2048 * 6 + 1, 70001])
def test_two_tiles_per_workgroup_kernel_at_small_sizes(monkeypatch, n, diag):
    """k_propagate_mt (mp_pf_k1mt.h: one workgroup walks two tiles, row gathers under the other tile's arithmetic) is picked for
    jobs of at least two tiles per CU; MP_K1_MT_GRID=1 (read at creation) lowers that bar to two tiles, so that odd tile counts
    — a last workgroup without a second tile —, ragged last tiles and reads in every position meet it at sizes the checker
    walks in milliseconds.  Same Philox blocks, targets and walks as k_propagate: every value against the one-workgroup-per-tile
    kernel (MP_K1_MT=0) and against the checker, bit for bit."""
    import modppl_amd

    seed, T = 97, 7
    obs = O.lgssm_observations(T).reshape(T, 1)
    monkeypatch.setenv("MP_K1_MT_GRID", "1")
    mt = modppl_amd.ParticleSystem(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), n, seed)
    monkeypatch.delenv("MP_K1_MT_GRID")
    monkeypatch.setenv("MP_K1_MT", "0")
    one = modppl_amd.ParticleSystem(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), n, seed)
    monkeypatch.delenv("MP_K1_MT")
    ref = O.OraclePF(1, 1, 1, O.LGSSM_PARAMS, n, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA)
    for pf in (mt, one):
        pf.init_step(None, obs[:1])
    ref.init_step(obs[:1])
    for t in range(1, T):
        for pf in (mt, one):
            pf.resample(sync=False)
            pf.step(obs[t:t + 1])
        ref.resample()
        ref.step(obs[t:t + 1])
        if t % 2:   # (reads only every other round: the rounds in between hand over from one launch to the next untouched)
            assert np.array_equal(mt.parents, ref.parents())
            assert np.array_equal(mt.parents, one.parents)
            assert np.array_equal(mt.log_weights, ref.log_weights())
            assert np.array_equal(mt.states(), ref.state())
    assert np.array_equal(mt.states(), one.states())
    assert np.array_equal(mt.states(), ref.state())
    assert np.array_equal(mt.log_weights, ref.log_weights())
    assert mt.effective_sample_size(fresh=True) == ref.effective_sample_size(1)
    assert mt.log_marginal_likelihood_estimate() == ref.log_marginal_likelihood_estimate() == one.log_marginal_likelihood_estimate()


@pytest.mark.parametrize("n", [2048 * 4 + 77, 2048 * 7])
def test_log_weights_and_parents_left_out_by_the_step_are_reproduced_on_demand(monkeypatch, n, diag):
    """A drawing k_propagate_mt launch stores neither the log-weights nor the parents (round 5: `resample` zeroes the one and replaces
    the other, particle_filter.rs:109-114; 12 B per particle-step of dead stores in a step / resample loop).  Whoever reads them
    before the next resample gets them from a replay of that launch (mp_pf.hip ensure_lazy, MP_MT_REPLAY): every way of reaching that —
    a read of either, a FURTHER step without a resample in between (it accumulates onto the log-weights, :81), nothing at all — must
    leave the checker's values, and a filter that stores eagerly (MP_K1_LAZY=0) must agree."""
    import modppl_amd

    seed, T = 31, 12
    obs = O.lgssm_observations(T).reshape(T, 1)
    monkeypatch.setenv("MP_K1_MT_GRID", "1")
    lazy = modppl_amd.ParticleSystem(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), n, seed)
    monkeypatch.setenv("MP_K1_LAZY", "0")
    eager = modppl_amd.ParticleSystem(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), n, seed)
    monkeypatch.delenv("MP_K1_LAZY")
    monkeypatch.delenv("MP_K1_MT_GRID")
    ref = O.OraclePF(1, 1, 1, O.LGSSM_PARAMS, n, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA)
    for pf in (lazy, eager):
        pf.init_step(None, obs[:1])
    ref.init_step(obs[:1])
    # what happens between a step and the next resample, round by round
    plan = ["none", "logw", "parents", "step", "none", "both", "step+logw", "states", "none", "step", "parents"]
    for t, what in zip(range(1, T), plan):
        for pf in (lazy, eager):
            pf.resample(sync=False)
            pf.step(obs[t:t + 1])
            assert pf.last_propagate_form() == capi.MP_K1_FORM_TWO_TILES
        ref.resample()
        ref.step(obs[t:t + 1])
        if what in ("logw", "both"):
            assert np.array_equal(lazy.log_weights, ref.log_weights())
        if what in ("parents", "both"):
            assert np.array_equal(lazy.parents, ref.parents())
        if what == "states":
            assert np.array_equal(lazy.states(), ref.state())
        if what.startswith("step"):   # a second observation of the same generation: no resample, the log-weights accumulate
            for pf in (lazy, eager):
                pf.step(obs[t:t + 1])
            ref.step(obs[t:t + 1])
            if what == "step+logw":
                assert np.array_equal(lazy.log_weights, ref.log_weights())
                assert np.array_equal(lazy.parents, ref.parents())
    assert np.array_equal(lazy.log_weights, ref.log_weights())
    assert np.array_equal(lazy.log_weights, eager.log_weights)
    assert np.array_equal(lazy.parents, eager.parents)
    assert np.array_equal(lazy.states(), ref.state())
    assert lazy.log_marginal_likelihood_estimate() == ref.log_marginal_likelihood_estimate() == eager.log_marginal_likelihood_estimate()


@pytest.mark.parametrize("n", [2048 * 4 + 77, 1 << 20])
def test_run_owns_the_loop_and_skips_the_dead_stores(monkeypatch, n, diag):
    """mp_pf_run knows that a resample follows every step: its k_propagate_mt launches store neither the log-weights (zeroed by
    `resample`, particle_filter.rs:114) nor the parents of any resample but the last, which is still pending when it returns.
    Whatever is read afterwards — parents, states, (zero) log-weights, ESS, log-ML, a further step — equals the checker's loop."""
    import modppl_amd

    seed, T = 5, 6
    obs = O.lgssm_observations(T + 1).reshape(T + 1, 1)
    if n < (1 << 20):
        monkeypatch.setenv("MP_K1_MT_GRID", "1")
    pf = modppl_amd.ParticleSystem(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), n, seed)
    monkeypatch.delenv("MP_K1_MT_GRID", raising=False)
    ref = O.OraclePF(1, 1, 1, O.LGSSM_PARAMS, n, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA)
    pf.run(None, obs[:T])
    ref.init_step(obs[:1])
    ref.resample()
    for t in range(1, T):
        ref.step(obs[t:t + 1])
        ref.resample()
    assert np.array_equal(pf.parents, ref.parents())
    assert np.array_equal(pf.states(), ref.state())
    assert np.all(pf.log_weights == 0.0)
    assert pf.effective_sample_size() == ref.effective_sample_size(0)
    assert pf.log_marginal_likelihood_estimate() == ref.log_marginal_likelihood_estimate()
    pf.step(obs[T:T + 1])
    ref.step(obs[T:T + 1])
    assert np.array_equal(pf.log_weights, ref.log_weights())
    assert np.array_equal(pf.states(), ref.state())


@pytest.mark.parametrize("force", [None, "0", "1"])
@pytest.mark.parametrize("scheme", [0, 1, 2])
def test_collapsed_weights_one_double_state(monkeypatch, scheme, force, diag):
    """An observation noise of 1e-4 leaves ONE particle with all the weight: every draw lands in one tile, a thousand of them start in
    the guide cell that holds the light rows in front of the heavy one and walk up to two thousand rows.  The propagate kernels have
    an instantiation whose long walks finish by bisection (k_propagate<…, WALKB>, k_propagate_mt<…, true>): the default for these
    kernels (MP_WALK_BISECT = 0 / 1 forces never / always).  Same parents
    either way — the checker finds them by binary search — at a size where both the one-workgroup-per-tile kernel and the two-tile
    kernel run."""
    import modppl_amd

    if force is not None:
        monkeypatch.setenv("MP_WALK_BISECT", force)
    for n in (40000, 1 << 20):
        T, seed = 6, 3
        params = (0.0, 1.0, 0.9, 0.5, 1e-4)
        obs = np.random.default_rng(3).normal(0, 1.0, size=(T, 1))
        pf = modppl_amd.ParticleSystem(modppl_amd.lgssm_model(*params), n, seed)
        ref = O.OraclePF(1, 1, 1, params, n, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA, threads=8)
        pf.init_step(None, obs[:1])
        ref.init_step(obs[:1])
        for t in range(1, T):
            if t % 2:
                assert pf.resample(scheme) == ref.resample(scheme)
                assert np.array_equal(pf.parents, ref.parents())
            else:   # asynchronous: the draws and their lookups inside the next step's kernel
                pf.resample(scheme, sync=False)
                ref.resample(scheme)
            pf.step(obs[t:t + 1])
            ref.step(obs[t:t + 1])
            assert np.array_equal(pf.log_weights, ref.log_weights())
            if t % 2 == 0:
                assert np.array_equal(pf.parents, ref.parents())
        assert pf.effective_sample_size(fresh=True) < n / 64        # collapsed indeed
        assert np.array_equal(pf.states(), ref.state())


@pytest.mark.parametrize("scheme", [0, 1, 2])
def test_collapsed_weights_single_kernel_resampler(monkeypatch, scheme, diag):
    """MP_DEFERRED_LOOKUPS=0: k_resample_gather (also the kernel behind importance_resampling) with ONE particle carrying the weight —
    its long row walks finish by bisection and its tile walk is budgeted; same parents as the checker's binary searches."""
    import modppl_amd

    monkeypatch.setenv("MP_DEFERRED_LOOKUPS", "0")
    n, T, seed = 50000, 6, 4
    params = (0.0, 1.0, 0.9, 0.5, 1e-4)
    obs = np.random.default_rng(3).normal(0, 1.0, size=(T, 1))
    pf = modppl_amd.ParticleSystem(modppl_amd.lgssm_model(*params), n, seed)
    ref = O.OraclePF(1, 1, 1, params, n, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA, threads=8)
    pf.init_step(None, obs[:1])
    ref.init_step(obs[:1])
    for t in range(1, T):
        assert pf.resample(scheme) == ref.resample(scheme)
        assert np.array_equal(pf.parents, ref.parents())
        assert np.array_equal(pf.states(), ref.state())
        pf.step(obs[t:t + 1])
        ref.step(obs[t:t + 1])
        assert np.array_equal(pf.log_weights, ref.log_weights())
    assert pf.effective_sample_size(fresh=True) < n / 64
