#!/usr/bin/env python3
"""Diagnostics: where the time of one SMC step goes inside K1 / K3a / K3b, which clock the chip holds, and whether every
workgroup of a launch is resident at once.  Uses the -DMP_STAMPS build (modppl_amd/csrc/libmodppl_hip_stamps.so): wave 0 of
every workgroup stamps the shader clock (s_memtime) and the 100 MHz real-time counter (s_memrealtime) at fixed points.
Never part of the product; numbers from this build are SHARES and clocks, not kernel times.

    python tools/stamp_probe.py [--particles N] [--steps K]
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

KERNELS, MAX_WG, SLOTS = 4, 16384, 32


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=int, default=1 << 20)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--raw", default=None, help="write the raw stamp array here (.npy)")
    ap.add_argument("--sync", action="store_true", help="the stamped launches are steps of the synchronous loop (step; ESS; L = resample()): a launch on an idle queue, with the peek tail")
    args = ap.parse_args()
    from modppl_amd import build as B

    # (MP_STAMPS_LIB: another -DMP_STAMPS build, e.g. one of tools/build_variant.py's)
    os.environ["MODPPL_HIP_LIB"] = os.environ.get("MP_STAMPS_LIB") or B.build_stamps()
    import modppl_amd
    from modppl_amd import capi
    from bench import LGSSM_PARAMS, lgssm_observations

    L = capi.load()
    L.mp_debug_stamps.argtypes = [C.c_void_p]
    n = args.particles
    ys = lgssm_observations(args.steps + 2)
    pf = modppl_amd.ParticleSystem(modppl_amd.lgssm_model(*LGSSM_PARAMS), n, 20241008)
    pf.init_step(None, ys[:1])
    pf.resample(sync=False)
    for t in range(1, args.steps):
        pf.step(ys[t:t + 1])
        pf.resample(sync=False)
    pf.synchronize()
    capi.check(L.mp_debug_stamps(None))   # arm
    # several steps back to back: every launch overwrites the stamps, the last one's remain — a launch in the middle of a busy
    # queue, as in the bench (a launch behind an idle queue starts its workgroups up to 3 us apart, XCD by XCD)
    ys2 = lgssm_observations(args.steps + 8)
    if args.sync:
        pf.resample()
    for t in range(args.steps, args.steps + 5):
        pf.step(ys2[t:t + 1])
        if args.sync:
            pf.effective_sample_size()
            pf.resample()
        else:
            pf.resample(sync=False)
    pf.synchronize()
    buf = np.zeros((KERNELS, MAX_WG, SLOTS), dtype=np.uint64)
    capi.check(L.mp_debug_stamps(buf.ctypes.data_as(C.c_void_p)))
    if args.raw:
        np.save(args.raw, buf)
    out = {"particles": n}
    names = {0: "k_propagate", 1: "k_draw_slots"}
    for k, name in names.items():
        b = buf[k]
        live = b[:, 1] != 0
        if not live.any():
            continue
        b = b[live].astype(np.int64)
        rt0, rt1 = b[:, 1], b[:, 5]
        c0, c1 = b[:, 0], b[:, 4]
        t_first = rt0.min()
        dur_rt = (rt1 - rt0) / 100.0        # us per workgroup (wave 0)
        clk = (c1 - c0) / np.maximum(rt1 - rt0, 1) * 100.0   # MHz
        rec = {
            "workgroups": int(live.sum()),
            "kernel_span_us": float((rt1.max() - t_first) / 100.0),
            "wg_start_us_p0_p50_p90_p100": [float(np.percentile((rt0 - t_first) / 100.0, q)) for q in (0, 50, 90, 100)],
            "wg_end_us_p0_p50_p90_p100": [float(np.percentile((rt1 - t_first) / 100.0, q)) for q in (0, 50, 90, 100)],
            "wg_lifetime_us_mean": float(dur_rt.mean()),
            "shader_clock_MHz_median": float(np.median(clk)),
            "xcc_ids": sorted(set(int(x) for x in (b[:, 6] >> 32) & 0xF)),
        }
        tot = (c1 - c0).astype(np.float64)
        if k == 0:
            rec["share_rejection_loop"] = float(np.mean((b[:, 2] - c0) / tot))
            rec["share_model"] = float(np.mean((b[:, 3] - b[:, 2]) / tot))
            rec["share_norm_to_scan"] = float(np.mean((b[:, 7] - b[:, 3]) / tot))
            rec["share_rows_guide"] = float(np.mean((c1 - b[:, 7]) / tot))
        if k == 1:
            rec["share_table"] = float(np.mean((b[:, 2] - c0) / tot))
            rec["share_draws_ranks"] = float(np.mean((b[:, 3] - b[:, 2]) / tot))
            rec["share_offsets_stores"] = float(np.mean((c1 - b[:, 3]) / tot))
        out[name] = rec
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
