#!/usr/bin/env python3
"""bench.py — particle-steps/sec of the LGSSM SMC hot path on MI355X (BASELINE.json metric).

A "step" is one SMC time step over the whole particle population: `ParticleSystem::step`
(propagate + weight) followed by `ParticleSystem::resample` (normalise, multinomial draw, gather)
— the loop body of modppl/tests/smc.rs:79-84.  Workload at N=1: BASELINE.json configs[1]
(LGSSM d=1, 2^20 particles, synthetic observations simulated from the model with numpy seed
20241008).  State is resident in HBM before the timed region; the timed region is K steps,
bracketed by barrier + synchronize, max over ranks.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

TRAFFIC_JSON = os.path.join(ROOT, "profiles", "r05", "traffic.json")  # tools/collect_traffic.py over the rocprofv3 --pmc passes of this command;
# it carries the content hash of the kernel sources it was measured with: a summary of another build is refused
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)
N_PER_GPU = 1 << 20
DIM = 1
# algorithmic bytes per particle per kernel family (B(d) = 32d + 64 per particle-step):
#   propagate       = propagate+weight (16d+16) and level 0 of the normalisation fused into it (LSE read 8 + normalise/scan 8+8)
#   normalize_scan  = the standalone form of that level 0 (only launched when the weights changed without a propagate)
#   bin_draws       = the search half of the resample (8 + 4): k_draw_slots
#   resample_gather = the gather half (4 + 16d, weight reset 8) where a kernel of its own does it (the single-kernel resampler,
#                     the sharded path); in the unsharded multinomial step the next k_propagate looks its slots' parents up
#                     itself, so that launch carries these bytes too (FUSED_GATHER below)
BYTES_K = {"propagate": 16 * DIM + 16 + 24, "normalize_scan": 8 + 8 + 8, "bin_draws": 8 + 4, "resample_gather": 4 + 16 * DIM + 8}
FUSED_GATHER = 4 + 16 * DIM + 8
SHARD_NOTE = {}   # how the sharded filter's collectives were issued (N > 1)
KERNEL_OF = {"propagate": "k_propagate<mp_lgssm1, 1024, false, false, false, true>", "normalize_scan": "k_normalize_tiles", "bin_draws": "k_draw_slots<1, 0>", "resample_gather": "k_resample_gather<0>"}
BYTES_STEP = 32 * DIM + 64


LGSSM_PARAMS = (0.0, 1.0, 0.9, 0.5, 1.0)   # mu0, sig0, a, sig_x, sig_y


def lgssm_observations(T, seed=20241008):
    """y_0..y_{T-1} simulated once from the model (x_0 ~ N(mu0, sig0), x_t ~ N(a x_{t-1}, sig_x), y_t ~ N(x_t, sig_y))."""
    mu0, sig0, a, sig_x, sig_y = LGSSM_PARAMS
    rng = np.random.default_rng(seed)
    ys = np.empty(T)
    x = mu0 + sig0 * rng.normal()
    for t in range(T):
        if t > 0:
            x = a * x + sig_x * rng.normal()
        ys[t] = x + sig_y * rng.normal()
    return ys


def kalman_log_ml(ys):
    """closed-form log marginal likelihood of the same model (scalar Kalman filter): the ground truth of log_ml"""
    mu0, sig0, a, sig_x, sig_y = LGSSM_PARAMS
    m, P, ll = mu0, sig0 * sig0, 0.0
    for t, y in enumerate(ys):
        if t > 0:
            m, P = a * m, a * a * P + sig_x * sig_x
        S = P + sig_y * sig_y
        ll += -0.5 * (np.log(2.0 * np.pi * S) + (y - m) ** 2 / S)
        K = P / S
        m, P = m + K * (y - m), (1.0 - K) * P
    return float(ll)


def _cpu_run(ys, n, variant, threads, target_seconds):
    from tests import oracle_lib as O

    pf = O.OraclePF(1, 1, 1, np.array(LGSSM_PARAMS), n, 20241008, variant, threads=threads)
    pf.init_step(ys[:1])
    pf.resample()
    t0 = time.perf_counter()
    steps = 0
    while steps < len(ys) - 1 and (time.perf_counter() - t0) < target_seconds:
        pf.step(ys[steps + 1:steps + 2])
        pf.resample()
        steps += 1
    dt = time.perf_counter() - t0
    return n * steps / dt, steps, dt


def cpu_baseline(ys, n):
    """The CPU restatement (oracle/), on bounded samples of the same workload:
      value               SoA engine, literal libm arithmetic, sequential fp64 CDF + binary search, 1 core (the
                          reference is single-threaded), full N, as many SMC steps as fit in ~10 s;
      all_cores           the same with the element-wise loops on this GPU's share of the host cores (<= 16; sums stay
                          sequential), ~5 s;
      structure_faithful  the reference's own structure — trie-addressed traces and its O(N^2) multinomial — at
                          N = 10^4 (at 2^20 one resample would take hours), a few steps."""
    from tests import oracle_lib as O

    v, steps, dt = _cpu_run(ys, n, O.VARIANT_SOA, 1, 10.0)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 16)   # one GPU's share of the host (the box shows every core of the node)
    va, sa, dta = _cpu_run(ys, n, O.VARIANT_SOA, cores, 5.0)
    n_sf = 10000
    vs, ss, dts = _cpu_run(ys, n_sf, 0, 1, 2.0)
    return {"value": v, "unit": "particle-steps/s", "cores": 1, "kind": "port",
            "sample": f"{steps} SMC steps (step+resample) at N={n}, C++ restatement of modppl's CPU path "
                      f"(SoA engine, literal libm arithmetic, sequential fp64 CDF + binary search), {dt:.1f} s",
            "all_cores": {"value": va, "cores": cores, "sample": f"{sa} steps at N={n}, std::thread over the element-wise loops, {dta:.1f} s"},
            "structure_faithful": {"value": vs, "cores": 1,
                                   "sample": f"{ss} steps at N={n_sf}, trie-addressed traces + the reference's O(N^2) multinomial, {dts:.1f} s"}}


def cpu_baseline_parity(model_fn, seed=20241008):
    """Part of the cpu_baseline leg (the only place bench.py touches oracle/, as the checker).
    BASELINE.md §3: the GPU filter beside the LITERAL CPU restatement (libm, sequential fp64 sums, the reference's
    `while t < u` scan) on the same seed and observations — (i) the C1 shape, N = 1000, T = 50, every resample compared;
    (ii) a 3-resample slice of the bench workload at N = 2^20.  Returns the relative log-ML difference and the number
    of resample indices that differ (the GPU evaluates the canonical fixed-point CDF)."""
    import modppl_amd
    from tests import oracle_lib as O

    out = {}
    worst_rel, mism, draws = 0.0, 0, 0
    for tag, n, T in (("c1_n1000_t50", 1000, 50), ("c2_slice_n2e20_t4", 1 << 20, 4)):
        ys = lgssm_observations(T)
        pf = modppl_amd.ParticleSystem(model_fn(), n, seed)
        ref = O.OraclePF(1, 1, 1, np.array(LGSSM_PARAMS), n, seed, O.VARIANT_SOA | O.VARIANT_FAST_SEARCH if n > 4096 else 0)
        pf.init_step(None, ys[:1])
        ref.init_step(ys[:1])
        m = 0
        # the 2^20 slice runs the TIMED path: resample(sync=False); step back to back, parents read after the step that consumed
        # the draws (k_propagate makes them itself there); the C1 shape keeps the synchronous resample (k_draw_slots makes them)
        timed_path = n >= (1 << 20)
        for t in range(1, T):
            ref.resample()
            want = ref.parents().copy()
            if timed_path:
                pf.resample(sync=False)
                pf.step(ys[t:t + 1])
                m += int((pf.parents != want).sum())
            else:
                pf.resample()
                m += int((pf.parents != want).sum())
                pf.step(ys[t:t + 1])
            ref.step(ys[t:t + 1])
        a, b = pf.log_marginal_likelihood_estimate(), ref.log_marginal_likelihood_estimate()
        rel = abs(a - b) / abs(b)
        out[tag] = {"log_ml_gpu": a, "log_ml_cpu": b, "log_ml_rel_err": rel, "index_mismatches": m, "draws": n * (T - 1),
                    "draws_made_by": "k_propagate (asynchronous resample: the timed path)" if timed_path else "k_draw_slots (synchronous resample)"}
        worst_rel = max(worst_rel, rel)
        mism += m
        draws += n * (T - 1)
    return worst_rel, mism, draws, out


def sub_benches(steps, warmup, which):
    """The other BASELINE.json configurations, from the same process, as driver-run numbers (not `value`):
    c3 bearings-only d = 4, 2^22 particles; c5_shard LGSSM d = 16, one GPU's share 2^21 of the 2^24-particle job;
    c4 regen-MH, 2^20 chains.  One step = step + multinomial resample; bytes per particle-step B(d) = 32 d + 64."""
    import modppl_amd

    rng = np.random.default_rng(20241008)
    T = 1 + warmup + steps
    res = {}
    capi_MULTINOMIAL, capi_SYSTEMATIC = 0, 1   # MP_RESAMPLE_*

    def pf_case(model, n, obs, d, also_systematic=False):
        def run(scheme):
            pf = modppl_amd.ParticleSystem(model, n, 20241008)
            pf.init_step(None, obs[:1])
            pf.resample(scheme, sync=False)
            for t in range(1, 1 + warmup):
                pf.step(obs[t:t + 1])
                pf.resample(scheme, sync=False)
            pf.synchronize()
            t0 = time.perf_counter()
            for t in range(1 + warmup, T):
                pf.step(obs[t:t + 1])
                pf.resample(scheme, sync=False)
            pf.synchronize()
            return time.perf_counter() - t0, pf.log_marginal_likelihood_estimate()

        dt, lml = run(capi_MULTINOMIAL)
        b = 32 * d + 64
        out = {"particles": n, "dim_state": d, "steps": steps, "us_per_step": dt / steps * 1e6, "particle_steps_per_s": n * steps / dt,
               "step_bytes_per_particle": b, "step_hbm_frac": b * n * steps / dt / 1e9 / HBM_PEAK_GBPS, "log_ml": lml}
        if also_systematic:   # supplementary (the reference has multinomial only): the same steps with the systematic lattice
            dts, _ = run(capi_SYSTEMATIC)
            out["systematic_resampling"] = {"us_per_step": dts / steps * 1e6, "step_hbm_frac": b * n * steps / dts / 1e9 / HBM_PEAK_GBPS}
        return out

    # The wide models' kernels are bound by VALU issue (sixteen polar normals per particle at d = 16), not by the bytes they move: next to
    # step_hbm_frac each carries the share of the chip's VALU issue slots its propagate kernel uses — from the rocprofv3 --pmc passes of
    # the profile round (profiles/r05/valu_issue.json, tools/collect_valu_issue.py), accepted only if they belong to THIS build
    def valu_issue(kernel_substr):
        try:
            from modppl_amd import build as _b

            vj = json.load(open(os.path.join(ROOT, "profiles", "r05", "valu_issue.json")))
            if vj.get("_measured", {}).get("source_hash") != _b.source_hash():
                return {"valu_issue_frac": None, "valu_issue_note": "profiles/r05/valu_issue.json is from another build"}
            hits = [(k, v) for k, v in vj.items() if k != "_measured" and kernel_substr in k]
            if not hits:
                return {"valu_issue_frac": None, "valu_issue_note": "no kernel matching %r in profiles/r05/valu_issue.json" % kernel_substr}
            k, v = max(hits, key=lambda kv: kv[1]["mean_launch_us"])
            return {"valu_issue_frac": v["valu_issue_frac"], "valu_issue_kernel": k, "valu_issue_kernel_us": v["mean_launch_us"],
                    "valu_issue_formula": vj["_measured"]["formula"], "bound": "VALU issue (instruction-bound: the bytes-based fraction measures the wrong resource)"}
        except Exception as e:   # noqa: BLE001
            return {"valu_issue_frac": None, "valu_issue_note": "no PMC summary for this build (%s)" % type(e).__name__}

    if "c3" in which:
        th = np.arctan2(1.0 + 0.05 * np.arange(T), 1.0 + 0.1 * np.arange(T)) + rng.normal(0, 0.02, T)
        res["c3"] = dict(pf_case(modppl_amd.bearings_model(), 1 << 22, th.reshape(T, 1), 4, also_systematic=True), workload="bearings-only tracker d=4, 2^22 particles (BASELINE.json configs[2])",
                         **valu_issue("k_propagate<mp_bearings"))
    if "c5" in which:
        res["c5_shard"] = dict(pf_case(modppl_amd.lgssm_band_model(16), 1 << 21, rng.normal(0, 1.2, size=(T, 16)), 16),
                               workload="LGSSM d=16, 2^21 particles = one GPU's share of configs[4] (16M over 8 GPUs), unsharded code path",
                               **valu_issue("k_propagate<mp_lgssm_band<16>"))
    if "c5" in which:
        # the same shard with a DENSE transition (dense A, Q, R: two mvnormal sites): the 16 x 16 products on the matrix cores
        r5 = np.random.default_rng(5)
        A = 0.9 * np.eye(16) + 0.08 * r5.normal(size=(16, 16)) / 4.0
        m = r5.normal(size=(16, 16)); Q = 0.25 * (m @ m.T / 16 + 0.5 * np.eye(16))
        m = r5.normal(size=(16, 16)); R = 4.0 * (m @ m.T / 16 + 0.5 * np.eye(16))
        res["c5_dense_shard"] = dict(pf_case(modppl_amd.lgssm_dense_model(A, Q, R, 1.0), 1 << 21, rng.normal(0, 1.5, size=(T, 16)), 16),
                                     workload="dense-transition LGSSM d=16 (mvnormal(A x, Q), mvnormal(x, R)), 2^21 particles, v_mfma_f64_16x16x4_f64 kernel",
                                     **valu_issue("k_propagate_dense16"))
    if "c4" in which:
        xs = np.arange(-5, 6, dtype=np.float64)
        ys = 0.3 + 0.4 * xs + 0.5 * xs * xs + rng.normal(0, 0.1, xs.size)
        ch = modppl_amd.HierarchicalChains(xs, ys, 1 << 20, 20241008, constrain_is_linear=False)
        ch.regen_mh([1, 2, 3], n_iters=3, cycle=True)
        sweeps = max(4, steps // 5)
        t0 = time.perf_counter()
        ch.regen_mh([1, 2, 3], n_iters=3 * sweeps, cycle=True)
        dt = time.perf_counter() - t0
        rate = (1 << 20) * 3 * sweeps / dt
        # this kernel is bound by fp64 vector arithmetic, not by HBM — both figures.  Arithmetic per chain-iteration
        # (quadratic branch, 11 observations; counted on k_mh_iterate<0>): one polar normal (1.27 attempts of
        # 9 operations, then ln, divide, sqrt: 71), eleven normal log-densities with their means (20 each) and weight updates (2 each),
        # the accept test's ln (37): ~350 fp64 instructions, ~400 flop with their fused multiply-adds counted twice; a division
        # or a square root counts as the ~10 instructions it expands to.  State: (a, b, c, is_linear) = 8 k + 8 bytes with k = 3,
        # read and written once per LAUNCH of 3 * sweeps iterations (it lives in registers in between).
        # MEASURED where profiles/r05/c4_flops.json belongs to this build (tools/collect_c4_flops.py over the SQ_INSTS_VALU_*_F64 counters of
        # k_mh_iterate<0>: 64 lanes x (ADD + MUL + TRANS + 2 FMA) per chain-iteration); the hand count (400) otherwise, and the line says which
        flop_per_it, flop_source = 400.0, "hand count of k_mh_iterate<0> (no PMC summary for this build)"
        try:
            from modppl_amd import build as _b

            fj = json.load(open(os.path.join(ROOT, "profiles", "r05", "c4_flops.json")))
            if fj.get("_measured", {}).get("source_hash") == _b.source_hash() and "k_mh_iterate<0>" in fj:
                flop_per_it = float(fj["k_mh_iterate<0>"]["flop_per_chain_iteration"])
                flop_source = "profiles/r05/c4_flops.json: " + fj["_measured"]["formula"] + " (rocprofv3 --pmc passes of tools/mh_bench.py: profiles/r05/pmc_mh_summary.txt)"
            else:
                flop_source += "; profiles/r05/c4_flops.json is from another build"
        except Exception as e:   # noqa: BLE001
            flop_source += f" ({type(e).__name__})"
        res["c4"] = {"workload": "regen-MH on the hierarchical model, 2^20 chains, masks cycling a, b, c (BASELINE.json configs[3])",
                     "chains": 1 << 20, "chain_iterations": 3 * sweeps, "chain_iterations_per_s": rate,
                     "roofline": {"bound": "fp64 vector arithmetic", "flop_per_chain_iteration": flop_per_it, "flop_source": flop_source,
                                  #  the PMC figure counts 64 lanes per wave-instruction, masked-off lanes (is_linear divergence, rejection
                                  # loops) and the fp64 operations inside mp_exp / mp_log / the division corrections included: an UPPER bound of the
                                  # useful work — the hand count of the model's own arithmetic stays beside it
                                  "flop_per_chain_iteration_is": "an upper bound (lane-inclusive wave-instruction counts)" if flop_per_it != 400.0 else "a hand count",
                                  "flop_per_chain_iteration_hand_count": 400.0, "frac_at_the_hand_count": rate * 400.0 / 1e12 / 78.6,
                                  "achieved": rate * flop_per_it / 1e12,
                                  "peak": 78.6, "unit": "TFLOP/s", "frac": rate * flop_per_it / 1e12 / 78.6,
                                  "hbm_bytes_per_chain_per_launch": 2 * (8 * 3 + 8), "iterations_per_launch": 3 * sweeps,
                                  "hbm_GBps": rate / (3 * sweeps) * 2 * (8 * 3 + 8) / 1e9, "hbm_frac": rate / (3 * sweeps) * 2 * (8 * 3 + 8) / 1e9 / HBM_PEAK_GBPS}}
        # the same model and moves as a REGISTERED functor run by the generic Update / Regenerate handlers (mp_genfn.h): same results, bit for bit.
        # `registered_functor`: its observations DECLARED as data sites (kind 105: four sites of trace, any number of observations — what a
        # model with data should be written as); `registered_functor_all_sites_in_registers`: every "(y, j)" an ordinary site (kind 101,
        # rounds 2-4's form: 20 sites of trace in registers)
        for key, fk in (("registered_functor", "data"), ("registered_functor_all_sites_in_registers", True)):
            try:
                fch = modppl_amd.HierarchicalChains(xs, ys, 1 << 20, 20241008, constrain_is_linear=False, functor=fk)
                fch.regen_mh([1, 2, 3], n_iters=3, cycle=True)
                t0 = time.perf_counter()
                fch.regen_mh([1, 2, 3], n_iters=3 * sweeps, cycle=True)
                dtf = time.perf_counter() - t0
                res["c4"][key] = {"chain_iterations_per_s": (1 << 20) * 3 * sweeps / dtf, "ratio_to_hand_written": dt / dtf,
                                  "same_states": bool(np.array_equal(fch.states(), ch.states())), "model_kind": 105 if fk == "data" else 101}
                del fch
            except Exception as e:   # noqa: BLE001
                res["c4"][key] = {"error": repr(e)[:200]}
    return res


def reference_shaped_loop(model, n, ys, steps, warmup):
    """The loop a drop-in caller of the reference writes (modppl/tests/smc.rs:64-90 over particle_filter.rs:73-116): every call
    SYNCHRONOUS — `step`; `effective_sample_size()` (the reference's stale value, a host f64); `L = resample()` (a host f64) —
    against bench.py's own loop, whose resample only enqueues.  The synchronous resample's L comes out of host-mapped memory written by
    the LAST workgroup of the step's own launch (round 5: mt_peek_tail; round 4 launched k_peek_level1 behind the step), the draws and
    lookups are left to the next step's k_propagate, the ESS comes out of host-mapped memory written by that launch's first workgroup:
    one launch and one host poll per step.  Also: the same loop with every particle's state copied to the host after each resample, as the reference's
    test does (PCIe-inclusive: 8 MB per step; never `value`), and the per-kernel durations of an instrumented repeat."""
    import modppl_amd
    from modppl_amd import capi

    pf = modppl_amd.ParticleSystem(model, n, 20241008)
    T = len(ys)
    pf.init_step(None, ys[:1])
    pf.resample()
    for t in range(1, 1 + warmup):
        pf.step(ys[t:t + 1])
        pf.effective_sample_size()
        pf.resample()
    dts_loop = []
    for rep in range(3):   # (the median of three loops, like `value`: the first loop of a process pays for first-touch effects)
        pf.synchronize()
        t0 = time.perf_counter()
        Ls = 0.0
        for t in range(1 + warmup, 1 + warmup + steps):
            pf.step(ys[t % T:t % T + 1])
            pf.effective_sample_size()
            Ls += pf.resample()
        pf.synchronize()
        dts_loop.append(time.perf_counter() - t0)
    dt = float(np.median(dts_loop))
    pf.set_timing(True)
    for t in range(1 + warmup, 1 + warmup + steps):
        pf.step(ys[t % T:t % T + 1])
        pf.effective_sample_size()
        pf.resample()
    pf.synchronize()
    fam = {k: pf.get_timing(v) for k, v in (("propagate", capi.MP_K_PROPAGATE), ("normalize_scan", capi.MP_K_NORMALIZE_SCAN),
                                            ("bin_draws", capi.MP_K_BIN_DRAWS), ("resample_gather", capi.MP_K_RESAMPLE_GATHER))}
    pf.set_timing(False)
    k_states = max(3, min(steps, 10))
    t0 = time.perf_counter()
    for t in range(1, 1 + k_states):
        pf.step(ys[t:t + 1])
        pf.effective_sample_size()
        pf.resample()
        pf.states()
    dts = time.perf_counter() - t0
    dts_pinned = None
    try:   # the same with the states landing in ONE pinned host buffer (what a caller who keeps every state would hand over)
        import torch
        pinned = torch.empty((n, model.dim_state), dtype=torch.float64).pin_memory().numpy()
        pf.states(out=pinned)
        t0 = time.perf_counter()
        for t in range(1, 1 + k_states):
            pf.step(ys[t:t + 1])
            pf.effective_sample_size()
            pf.resample()
            pf.states(out=pinned)
        dts_pinned = time.perf_counter() - t0
    except Exception:   # noqa: BLE001  (no pinned memory here: the figure is simply absent)
        dts_pinned = None
    return {"what": "step; effective_sample_size() -> f64; resample() -> f64 (every call synchronous, as in particle_filter.rs:73-116)",
            "steps": steps, "loops": len(dts_loop), "us_per_step": dt / steps * 1e6, "us_per_step_min": min(dts_loop) / steps * 1e6,
            "us_per_step_max": max(dts_loop) / steps * 1e6, "particle_steps_per_s": n * steps / dt,
            "step_hbm_frac": BYTES_STEP * n * steps / dt / 1e9 / HBM_PEAK_GBPS,
            "kernel_launches_per_step": {k: v[1] / steps for k, v in fam.items()},
            "kernel_avg_us": {k: (v[0] / v[1] * 1e3 if v[1] else 0.0) for k, v in fam.items()},
            "resample_return_value": "computed by the LAST workgroup of the step's own launch (mt_peek_tail: level 1 of the new tile scalars into host-mapped memory) once the loop is known to be synchronous; k_peek_level1, a launch of its own, only for the first synchronous resample — the difference between us_per_step and the kernel above is a launch on an idle queue plus one host poll",
            "with_states_copied_to_host_each_step": {"steps": k_states, "us_per_step": dts / k_states * 1e6,
                                                     "us_per_step_into_a_pinned_buffer": (dts_pinned / k_states * 1e6) if dts_pinned else None,
                                                     "note": "PCIe-inclusive: k_draw_slots + k_resolve_slots + 8 MB device-to-host per step (tests/smc.rs:64-90 writes every state to disk); us_per_step: into a fresh pageable array each step"},
            "sum_of_log_total_weights": Ls}


def _finite(obj):
    """JSON has no NaN / inf: a non-finite number anywhere (a degenerate sub-bench's log-ML, a 0 / 0 ratio) becomes null and its
    path is listed under "non_finite", so that one bad leg cannot cost the driver the whole line.  -> (clean object, paths)"""
    bad = []

    def walk(o, path):
        if isinstance(o, dict):
            return {k: walk(v, f"{path}.{k}" if path else str(k)) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [walk(v, f"{path}[{i}]") for i, v in enumerate(o)]
        if isinstance(o, (float, np.floating)) and not np.isfinite(o):
            bad.append(path)
            return None
        if isinstance(o, np.generic):
            return o.item()
        return o

    return walk(obj, ""), bad


def _library_identity():
    """Which libmodppl_hip.so this process runs: the tree's own build (content hash checked at load) or an override
    (MODPPL_HIP_LIB: A/B and diagnostics builds — a line measured with one is marked, and takes no traffic summary)."""
    from modppl_amd import build as _b

    override = os.environ.get("MODPPL_HIP_LIB")
    path = override or _b.SO
    stamp = None
    try:
        with open(path + ".srchash") as f:
            stamp = f.read().strip()
    except OSError:
        pass
    return {"path": os.path.relpath(path, ROOT) if path.startswith(ROOT) else path, "srchash": stamp, "tree_source_hash": _b.source_hash(),
            "override": bool(override), "product_build": (not override) and stamp == _b.source_hash()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--particles", type=int, default=N_PER_GPU, help="particles per GPU")
    ap.add_argument("--repeats", type=int, default=9, help="the timed region of --steps steps is repeated this many times; `value` is the median region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not record hipEvents around each launch in the timed region")
    ap.add_argument("--no-sub-benches", action="store_true", help="skip the c3 / c4 / c5 sub-objects")
    ap.add_argument("--no-systematic-leg", action="store_true", help="skip the supplementary systematic-resampling leg (profiles: only the headline step's kernels)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    # stdout carries exactly one JSON line: RCCL and the HIP runtime print banners on fd 1, so park it on stderr until then
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # MP_BENCH_REHEARSE=1 (never set by the driver): the N > 1 control flow on ONE GPU — all ranks share device 0 and the
    # collectives go through gloo with host-staged buffers.  For checking this script's multi-rank legs, not for numbers.
    rehearse = os.environ.get("MP_BENCH_REHEARSE", "0") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    shard_kw = {"host_staging": True} if rehearse else {}
    # Several GPUs: `value` is the OWNER-KEEPS exchange ("owned", the library's default: the parents are the single filter's multiset for any
    # number of GPUs — the algorithm of the single-GPU line and of rounds 1-3's lines), unless MP_SHARD_EXCHANGE says otherwise.  The split
    # multinomial (same law, another seeded stream: offspring per GPU drawn first, then every GPU its own parents — O(n) per GPU where
    # "owned" enumerates all N draws on every GPU) is timed as a supplementary leg beside it (`split_multinomial`), never as `value`
    # (round 4's N > 1 line had silently become a different algorithm).

    import modppl_amd
    from modppl_amd import capi

    n = args.particles
    K, W = args.steps, args.warmup
    T = 1 + W + K
    ys = lgssm_observations(T)     # nothing under oracle/ is touched outside cpu_baseline()
    kalman = kalman_log_ml(ys)

    model = modppl_amd.lgssm_model(*LGSSM_PARAMS)
    force_sharded = os.environ.get("MP_BENCH_FORCE_SHARDED", "0") == "1"  # diagnostics: the sharded code path in a world of one
    if force_sharded and dist is None:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
    if world == 1 and not force_sharded:
        pf = modppl_amd.ParticleSystem(model, n, 20241008, device=local_rank)
        timer = pf
    else:
        # ONE filter of world*n particles sharded over the ranks: RCCL all-gather of the tile totals (the weight
        # "all-reduce"), all-to-all of the parents' states (the particle exchange over xGMI; MP_SHARD_EXCHANGE=owned, the
        # default, moves only each rank's surplus, =exact keeps slot order and moves (G-1)/G of the particles)
        from modppl_amd.distributed import ShardedParticleSystem

        # A trial resample on a throw-away filter first: if the library's own RCCL path raises on ANY rank (it has run on one GPU
        # only: no multi-GPU node was ever available to the builder), every rank switches to the round-2 protocol over
        # torch.distributed (MP_SHARD_NATIVE=0) and the line says so — rather than no line at all.
        if world > 1 and os.environ.get("MP_SHARD_NATIVE", "1") == "1":
            ok, why = 1, ""
            try:
                trial = ShardedParticleSystem(model, 2048 * 4 * world, 1, engine_kwargs={"device_index": local_rank}, **shard_kw)
                trial.init_step(None, ys[:1])
                trial.resample(sync=True)
                trial.step(ys[1:2])
                trial.log_marginal_likelihood_estimate()
                trial.close()
            except Exception as e:   # noqa: BLE001
                ok, why = 0, repr(e)[:200]
            flag = torch.tensor([ok], dtype=torch.int32, device="cpu" if rehearse else "cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                os.environ["MP_SHARD_NATIVE"] = "0"
                SHARD_NOTE["transport"] = "torch.distributed collectives from Python (the native RCCL path failed its trial%s)" % (": " + why if why else " on another rank")
        pf = ShardedParticleSystem(model, n * world, 20241008, engine_kwargs={"device_index": local_rank}, **shard_kw)
        timer = pf.engine

    def barrier():
        # the filter's own stream first, by polling (mp_pf_synchronize spins on hipStreamQuery: it returns within a microsecond of
        # the last kernel, where a blocking synchronize wakes the host tens of microseconds late — 3-5 % of a 20-step region);
        # then the device-wide synchronize the contract asks for, which finds nothing left to wait for
        pf.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # what a plain device-to-device copy reaches on this box (read + write bytes), for scale next to the 8 TB/s spec peak
    # (measured first: 2.7 GB of copies also bring the card out of its idle clocks before anything is timed)
    copy_gbps = None
    if True:   # (every rank: each card is its own)
        a_ = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
        b_ = torch.empty_like(a_)
        b_.copy_(a_)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            b_.copy_(a_)
        e1.record()
        torch.cuda.synchronize()
        copy_gbps = 2 * a_.numel() * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del a_, b_
    pf.init_step(None, ys[:1])
    pf.resample(sync=False)
    for t in range(1, 1 + W):
        pf.step(ys[t:t + 1])
        pf.resample(sync=False)
    barrier()

    # ---- timed region: EXACTLY K steps, nothing but the hot path enqueued; one hipEvent pair on the kernels' stream around the
    # whole region (mp_pf_region_begin / _end: no per-launch instrumentation), max over ranks ----
    def timed_region():
        barrier()
        t0 = time.perf_counter()
        if hasattr(timer, "region_begin"):
            timer.region_begin()
        for t in range(1 + W, T):
            pf.step(ys[t:t + 1])
            pf.resample(sync=False)
        ev = timer.region_end() if hasattr(timer, "region_end") else (None, 0)
        barrier()
        d = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([d], device="cuda", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            d = float(tt.item())
        return d, ev

    dts, evs = [], []
    d0, ev0 = timed_region()
    dts.append(d0); evs.append(ev0)
    lml = pf.log_marginal_likelihood_estimate()   # of the first region: y_0 .. y_{T-1}, the run the Kalman value below belongs to
    # the same K steps again, R - 1 times (the filter simply goes on; the query above made the pending draws, so one untimed step
    # first puts the loop back into its steady state): `value` is the MEDIAN region, min / max beside it
    for _ in range(max(0, args.repeats - 1)):
        pf.step(ys[1 + W:2 + W])
        pf.resample(sync=False)
        d, ev = timed_region()
        dts.append(d); evs.append(ev)
    dt = float(np.median(dts))
    # ---- supplementary: the same K steps with systematic resampling (named next to multinomial in the north star; not `value`)
    dt_sys = None
    if not args.no_systematic_leg:   # (at N > 1 too: the sharded filter's lattice schemes need no enumeration of all N draws)
        sys_dts = []
        for rep in range(1 + min(4, max(0, args.repeats - 1))):   # (a first, untimed-in-effect region switches the scheme; median of the rest)
            barrier()
            t0 = time.perf_counter()
            for t in range(1 + W, T):
                pf.step(ys[t:t + 1])
                pf.resample(scheme=1, sync=False)
            barrier()
            d = time.perf_counter() - t0
            if dist is not None:
                tt = torch.tensor([d], device="cuda", dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                d = float(tt.item())
            sys_dts.append(d)
        dt_sys = float(np.median(sys_dts[1:])) if len(sys_dts) > 1 else sys_dts[0]
    # ---- per-kernel durations: the same K steps again with a hipEvent pair around every launch, recorded on the
    # stream the kernels run on (the pairs cost ~20 us per step, so they stay out of the region `value` is taken from)
    fam = {"propagate": (0.0, 0), "normalize_scan": (0.0, 0), "bin_draws": (0.0, 0), "resample_gather": (0.0, 0)}
    if not args.no_kernel_timing:
        timer.set_timing(True)
        for t in range(1 + W, T):
            pf.step(ys[t:t + 1])
            pf.resample(sync=False)
        barrier()
        fam = {"propagate": timer.get_timing(capi.MP_K_PROPAGATE), "normalize_scan": timer.get_timing(capi.MP_K_NORMALIZE_SCAN),
               "bin_draws": timer.get_timing(capi.MP_K_BIN_DRAWS), "resample_gather": timer.get_timing(capi.MP_K_RESAMPLE_GATHER)}
        timer.set_timing(False)

    # ---- N > 1: the workload that CAN scale (BASELINE.json configs[4]): LGSSM d = 16, 2^21 particles per GPU, same K steps,
    # through the same sharded filter (every rank runs it; rank 0 reports).  Not `value`.
    c5 = None
    if world > 1:
        from modppl_amd.distributed import ShardedParticleSystem

        n5 = 1 << 21
        obs5 = np.random.default_rng(20241008).normal(0, 1.2, size=(T, 16))
        pf5 = ShardedParticleSystem(modppl_amd.lgssm_band_model(16), n5 * world, 20241008, engine_kwargs={"device_index": local_rank}, **shard_kw)
        pf5.init_step(None, obs5[:1])
        pf5.resample(sync=False)
        for t in range(1, 1 + W):
            pf5.step(obs5[t:t + 1])
            pf5.resample(sync=False)
        dist.barrier(); torch.cuda.synchronize(); pf5.synchronize()
        t0 = time.perf_counter()
        for t in range(1 + W, T):
            pf5.step(obs5[t:t + 1])
            pf5.resample(sync=False)
        dist.barrier(); torch.cuda.synchronize(); pf5.synchronize()
        dt5 = time.perf_counter() - t0
        tt = torch.tensor([dt5], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt5 = float(tt.item())
        pf5.step(obs5[T - 1:T])
        pf5.resample(sync=True)          # outside the timed region: a synchronous resample reports how many rows travelled
        surplus = getattr(pf5, "last_exchange_rows", None)
        c5 = {"workload": "LGSSM d=16 bootstrap SMC, 2^21 particles per GPU, one sharded filter (BASELINE.json configs[4] at 8 GPUs)",
              "particles_per_gpu": n5, "particles_total": n5 * world, "steps": K, "ms_per_step": dt5 / K * 1e3,
              "particle_steps_per_s": n5 * world * K / dt5, "step_bytes_per_particle": 32 * 16 + 64,
              "step_hbm_frac_per_gpu": (32 * 16 + 64) * n5 * K / dt5 / 1e9 / HBM_PEAK_GBPS, "exchange": getattr(pf5, "exchange", None),
              "exchange_rows_last_step": surplus, "exchange_bytes_per_row": 8 * 17, "log_ml": pf5.log_marginal_likelihood_estimate()}
        del pf5
    # ---- N > 1, supplementary: the same K steps with the split multinomial exchange (not `value`).  Every rank first agrees that a small
    # trial of it works here (it has run on one GPU and over gloo only), so that a failure on one rank cannot leave the others in a collective.
    split_leg = None
    if world > 1 and getattr(pf, "exchange", "") == "owned" and os.environ.get("MP_BENCH_SPLIT_LEG", "1") == "1":
        from modppl_amd.distributed import ShardedParticleSystem

        kw2 = dict(shard_kw, exchange="split")
        ok, why = 1, ""
        try:
            trial = ShardedParticleSystem(model, 2048 * 4 * world, 1, engine_kwargs={"device_index": local_rank}, **kw2)
            trial.init_step(None, ys[:1])
            trial.resample(sync=True)
            trial.step(ys[1:2])
            trial.log_marginal_likelihood_estimate()
            trial.close()
        except Exception as e:   # noqa: BLE001
            ok, why = 0, repr(e)[:200]
        flag = torch.tensor([ok], dtype=torch.int32, device="cpu" if rehearse else "cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            split_leg = {"error": "the trial of exchange='split' failed on a rank" + (": " + why if why else "")}
        else:
            pfs = ShardedParticleSystem(model, n * world, 20241008, engine_kwargs={"device_index": local_rank}, **kw2)
            pfs.init_step(None, ys[:1])
            pfs.resample(sync=False)
            for t in range(1, 1 + W):
                pfs.step(ys[t:t + 1])
                pfs.resample(sync=False)
            pfs.synchronize(); torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in range(1 + W, T):
                pfs.step(ys[t:t + 1])
                pfs.resample(sync=False)
            pfs.synchronize(); torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            dsp = time.perf_counter() - t0
            tt = torch.tensor([dsp], device="cpu" if rehearse else "cuda", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dsp = float(tt.item())
            split_leg = {"exchange": "split", "ms_per_step": dsp / K * 1e3, "particle_steps_per_s": n * world * K / dsp,
                         "log_ml": pfs.log_marginal_likelihood_estimate(),
                         "note": "same law as `value`'s resample, another seeded stream (parents are not the single filter's multiset): supplementary, never `value`"}
            del pfs
    if rank == 0:
        # mean duration of one launch of each kernel (single GPU: one launch of each per step)
        timed = any(v[1] for v in fam.values())
        avg_us = {k: ((v[0] / v[1]) * 1e3 if v[1] else 0.0) for k, v in fam.items()} if timed else None
        sharded_path = world > 1 or force_sharded
        if hasattr(timer, "last_propagate_form") and timer.last_propagate_form() == 1:   # (the library says which form of K1 its steps launched)
            KERNEL_OF["propagate"] = "k_propagate_mt<mp_lgssm1, %s>" % ("false" if os.environ.get("MP_WALK_BISECT") == "0" else "true")   # (true: long walks finish by bisection)
        if sharded_path:   # the sharded filter runs other kernels for the resample
            if getattr(pf, "exchange", "") == "owned":
                KERNEL_OF.update({"bin_draws": "k_shard_table + k_shard_own_draw + k_shard_own_plan", "resample_gather": "k_shard_own_place"})
            else:
                KERNEL_OF.update({"bin_draws": "k_shard_route_fused", "resample_gather": "k_shard_resolve_binned"})
        roofline = None
        if timed:
            dom = max(fam, key=lambda k: fam[k][0])   # the kernel with the largest total time
            bytes_k = dict(BYTES_K)
            if not sharded_path and fam["resample_gather"][1] == 0:
                bytes_k["propagate"] += FUSED_GATHER   # the step's k_propagate also looked up / cloned the parents (no launch of its own did)
                if fam["bin_draws"][1] == 0:
                    bytes_k["propagate"] += BYTES_K["bin_draws"]   # ... and made the draws: the whole step is that one launch
            # the dominant kernel's mean launch duration: from the ONE event pair around each timed region (elapsed / launches of that
            # region, median over the regions) where every step is one launch of it — un-perturbed, and including the gaps between
            # launches, i.e. an upper bound; the per-launch event pairs of the instrumented repeat (kernel_avg_us) cost each launch
            # ~0.5 us and are kept for the breakdown by kernel
            dom_us, dom_src = avg_us[dom], "hipEvent pair around every launch (instrumented repeat)"
            reg = [e[0] / e[1] * 1e3 for e in evs if e[0] is not None and e[1] == K]
            if dom == "propagate" and reg and fam["bin_draws"][1] == 0 and fam["resample_gather"][1] == 0 and fam["normalize_scan"][1] == 0:
                dom_us, dom_src = float(np.median(reg)), f"one hipEvent pair around each timed region of {K} launches, median of {len(reg)} regions (no per-launch instrumentation)"
            achieved = bytes_k[dom] * n / (dom_us * 1e-6) / 1e9
            # HBM-side bytes per launch of that kernel: PMC passes cannot run inside this process, so this is the committed
            # summary of the same single-GPU command — accepted only if it names this kernel AND was measured with this build
            # (content hash of the kernel sources), otherwise null with the reason
            traffic, traffic_note = None, None
            lib_id = _library_identity()
            if not lib_id["product_build"]:
                traffic_note = "no traffic summary for a library that is not this tree's own build (MODPPL_HIP_LIB override)"
            elif n == N_PER_GPU and os.path.exists(TRAFFIC_JSON) and not sharded_path:
                try:
                    from modppl_amd import build as _b

                    tj = json.load(open(TRAFFIC_JSON))
                    meta = tj.get("_measured", {})
                    if meta.get("source_hash") != _b.source_hash():
                        traffic_note = f"profiles summary is from another build (commit {meta.get('commit')}): re-run tools/profile_round.sh"
                    elif KERNEL_OF.get(dom) not in meta.get("kernels", {}).get(dom, ""):
                        traffic_note = f"profiles summary names {meta.get('kernels', {}).get(dom)!r}, the dominant kernel is {KERNEL_OF.get(dom)!r}"
                    else:
                        traffic = tj[dom]["traffic_bytes"]
                        traffic_note = f"rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE per launch, commit {meta.get('commit')}; fetch correction: {tj[dom].get('fetch_correction')}"
                except Exception as e:   # noqa: BLE001
                    traffic_note = f"unreadable profiles summary: {e}"
            roofline = {"bound": "hbm", "kernel": KERNEL_OF.get(dom, dom), "kernel_us": dom_us, "kernel_us_source": dom_src,
                        "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_note,
                        "bytes_per_launch": bytes_k[dom] * n, "bytes_per_particle": bytes_k[dom],
                        "bytes_are": "ALGORITHMIC bytes of the reference step this launch stands for (SURVEY.md 8d: 96 B per particle-step at d = 1), "
                                     "not measured traffic: `traffic` is the measured figure"}
        out = {
            "metric": "particle-steps/sec, 1M-particle LGSSM SMC (step + multinomial resample per time step)",
            "value": n * world * K / dt,
            "unit": "particle-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": dt / K * 1e3,
            "repeats": len(dts),
            "ms_per_step_min": min(dts) / K * 1e3, "ms_per_step_max": max(dts) / K * 1e3,
            "value_is": f"median of {len(dts)} timed regions of {K} steps each (min / max: ms_per_step_min / _max)",
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic (observations simulated from the model, seed 20241008)",
            "config": {"workload": "LGSSM d=1 bootstrap SMC, resample every step (BASELINE.json configs[1])",
                       "particles_per_gpu": n, "time_steps_timed": K, "particles_total": n * world,
                       "parallelism": "1 GPU" if world == 1 else f"one filter sharded over {world} GPUs (RCCL all-gather of tile totals + all-to-all particle exchange)",
                       "exchange": getattr(pf, "exchange", None),
                       "collectives": None if world == 1 and not force_sharded else SHARD_NOTE.get(
                           "transport", ("through the library's mp_transport callbacks, staged on the host (rehearsal / gloo)" if getattr(pf, "_staged", None) is not None
                                         else "issued by the library on the filter's stream (mp_pf_shard_resample_rccl)") if getattr(pf, "_native", False)
                           else "torch.distributed collectives from Python (MP_SHARD_NATIVE=0)")},
            "log_ml": lml,
            "systematic_resampling_particle_steps_per_s": (n * world * K / dt_sys) if dt_sys else None,
            # (supplementary, not `value`: the same K steps with the systematic lattice — the reference has multinomial only; its draws
            # are made by the step's k_propagate as well, and its sorted parents make the row lookups nearly sequential)
            "systematic_resampling": ({"us_per_step": dt_sys / K * 1e6, "step_hbm_frac": BYTES_STEP * n * K / dt_sys / 1e9 / HBM_PEAK_GBPS}   # (per GPU)
                                      if dt_sys else None),
            "log_ml_abs_err_vs_kalman": abs(lml - kalman),
            "step_bytes_per_particle": BYTES_STEP,
            "step_hbm_frac": BYTES_STEP * n * K / dt / 1e9 / HBM_PEAK_GBPS,
            "hbm_copy_measured_GBps": copy_gbps,
            "kernel_avg_us": avg_us,
            "kernel_launches_per_step": {k: v[1] / K for k, v in fam.items()} if timed else None,
        }
        if roofline is not None:
            out["roofline"] = roofline
        if c5 is not None:
            out["c5"] = c5
        if split_leg is not None:
            out["split_multinomial"] = split_leg
        out["library"] = _library_identity()
        if world == 1 and not force_sharded and not args.no_sub_benches:
            try:
                out["reference_shaped_loop"] = reference_shaped_loop(model, n, ys, 50, 5)   # (its own 50 steps x 3 loops whatever K: a supplementary figure, not `value`)
            except Exception as e:   # noqa: BLE001
                out["reference_shaped_loop_error"] = f"{type(e).__name__}: {e}"
        if world == 1 and not force_sharded and not args.no_sub_benches:
            del pf
            for leg in ("c3", "c5", "c4"):   # (one leg's failure is that leg's, not the headline's)
                try:
                    out.update(sub_benches(max(10, min(K, 40)), min(W, 5), (leg,)))
                except Exception as e:   # noqa: BLE001
                    out[leg + "_error"] = f"{type(e).__name__}: {e}"
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ys, n)
            # parity beside the CPU restatement: relative log-ML difference, resample-index mismatches
            rel, mism, draws, detail = cpu_baseline_parity(lambda: modppl_amd.lgssm_model(*LGSSM_PARAMS))
            out["log_ml_rel_err_vs_cpu"] = rel
            out["resample_index_mismatches_vs_cpu"] = mism
            out["parity_vs_cpu"] = dict(detail, resample_draws_compared=draws,
                                        cpu="literal arithmetic (libm, sequential fp64 sums, linear `while t < u` scan at N=1000; binary search over the same running sum at 2^20)")
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        out, bad = _finite(out)
        if bad:
            out["non_finite"] = bad
        print(json.dumps(out, allow_nan=False), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
