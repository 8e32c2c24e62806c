#!/usr/bin/env python3
"""Per-model step times on one MI355X (BASELINE.json configs C2, C3, C5-per-GPU shard, C4): not the contract bench
(bench.py is), just the numbers DESIGN.md quotes for the other configurations.

    python tools/model_bench.py [--steps 30] [--which c2,c3,c5,c4]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def pf_case(name, model, n, obs, steps, warmup, bytes_per_particle, sharded=False, scheme=0):
    import modppl_amd
    from modppl_amd import capi

    if sharded:   # the sharded code path in a world of one (no collectives): what the exchange machinery costs locally
        from modppl_amd.distributed import ShardedParticleSystem

        sp = ShardedParticleSystem(model, n, 20241008)

        class _Wrap:
            def __getattr__(self, k):
                return getattr(sp, k)

            def set_timing(self, on):
                sp.engine.set_timing(on)

            def get_timing(self, fam):
                return sp.engine.get_timing(fam)

        pf = _Wrap()
        name += " [sharded path, world of one]"
    else:
        pf = modppl_amd.ParticleSystem(model, n, 20241008)
    pf.init_step(None, obs[:1])
    pf.resample(scheme, sync=False)
    for t in range(1, 1 + warmup):
        pf.step(obs[t:t + 1])
        pf.resample(scheme, sync=False)
    pf.synchronize()
    t0 = time.perf_counter()
    for t in range(1 + warmup, 1 + warmup + steps):
        pf.step(obs[t:t + 1])
        pf.resample(scheme, sync=False)
    pf.synchronize()
    dt = time.perf_counter() - t0
    pf.set_timing(True)
    for t in range(1 + warmup, 1 + warmup + steps):
        pf.step(obs[t:t + 1])
        pf.resample(scheme, sync=False)
    pf.synchronize()
    fam = {k: pf.get_timing(v) for k, v in (("propagate", capi.MP_K_PROPAGATE), ("bin_draws", capi.MP_K_BIN_DRAWS),
                                            ("resample_gather", capi.MP_K_RESAMPLE_GATHER))}
    us = dt / steps * 1e6
    return {"case": name, "particles": n, "us_per_step": us, "particle_steps_per_s": n * steps / dt,
            "hbm_frac_of_8TBs": bytes_per_particle * n / (us * 1e-6) / 8e12,
            "kernel_avg_us": {k: (v[0] / v[1] * 1e3 if v[1] else 0.0) for k, v in fam.items()},
            "log_ml": pf.log_marginal_likelihood_estimate()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--which", default="c2,c3,c5,c4")
    ap.add_argument("--scheme", type=int, default=0, help="0 multinomial, 1 systematic, 2 stratified")
    ap.add_argument("--sharded", action="store_true", help="run the filters through ShardedParticleSystem (world of one)")
    args = ap.parse_args()
    import modppl_amd
    import bench as B   # observations only; nothing under oracle/ is used by the tools

    T = 1 + args.warmup + args.steps
    rng = np.random.default_rng(20241008)
    out = []
    which = args.which.split(",")
    if "c2" in which:
        out.append(pf_case("C2 LGSSM d=1", modppl_amd.lgssm_model(*B.LGSSM_PARAMS), 1 << 20, B.lgssm_observations(T).reshape(T, 1),
                           args.steps, args.warmup, 96, args.sharded, args.scheme))
    if "c3" in which:
        th = np.arctan2(1.0 + 0.05 * np.arange(T), 1.0 + 0.1 * np.arange(T)) + rng.normal(0, 0.02, T)
        out.append(pf_case("C3 bearings d=4", modppl_amd.bearings_model(), 1 << 22, th.reshape(T, 1), args.steps, args.warmup, 192, args.sharded, args.scheme))
    if "mid" in which:   # the two-slot-lane kernels of wider states at 2^20 particles (512 tiles: the guide and the tile table sit in L2 / LDS)
        th = np.arctan2(1.0 + 0.05 * np.arange(T), 1.0 + 0.1 * np.arange(T)) + rng.normal(0, 0.02, T)
        out.append(pf_case("bearings d=4, 2^20", modppl_amd.bearings_model(), 1 << 20, th.reshape(T, 1), args.steps, args.warmup, 192, args.sharded, args.scheme))
        out.append(pf_case("LGSSM band d=4, 2^20", modppl_amd.lgssm_band_model(4), 1 << 20, rng.normal(0, 1.2, size=(T, 4)), args.steps, args.warmup,
                           192, args.sharded, args.scheme))
        out.append(pf_case("LGSSM band d=2, 2^21", modppl_amd.lgssm_band_model(2), 1 << 21, rng.normal(0, 1.2, size=(T, 2)), args.steps, args.warmup,
                           128, args.sharded, args.scheme))
    if "c5" in which:
        out.append(pf_case("C5 LGSSM band d=16 (one GPU's shard of 2^24 / 8)", modppl_amd.lgssm_band_model(16), 1 << 21,
                           rng.normal(0, 1.2, size=(T, 16)), args.steps, args.warmup, 576, args.sharded, args.scheme))
    if "c4" in which:
        xs = np.arange(-5, 6, dtype=np.float64)
        ys = 0.3 + 0.4 * xs + 0.5 * xs * xs + rng.normal(0, 0.1, xs.size)
        ch = modppl_amd.HierarchicalChains(xs, ys, 1 << 20, 20241008, constrain_is_linear=False)
        ch.regen_mh([1, 2, 3], n_iters=3, cycle=True)
        t0 = time.perf_counter()
        sweeps = 20
        ch.regen_mh([1, 2, 3], n_iters=3 * sweeps, cycle=True)
        dt = time.perf_counter() - t0
        out.append({"case": "C4 regen-MH hierarchical, masks cycle a,b,c", "chains": 1 << 20, "chain_iterations_per_s": (1 << 20) * 3 * sweeps / dt,
                    "us_per_sweep_of_3": dt / sweeps * 1e6})
        t0 = time.perf_counter()
        ch.mh(0.1, n_iters=60)
        dt = time.perf_counter() - t0
        out.append({"case": "C4 drift-proposal MH sigma=0.1", "chains": 1 << 20, "chain_iterations_per_s": (1 << 20) * 60 / dt})
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
