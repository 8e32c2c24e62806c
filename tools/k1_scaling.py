#!/usr/bin/env python3
"""k_propagate<mp_lgssm1> as a function of the number of resident workgroups per CU (tiles = n / 2048; 256 CUs):
plain steps (no lookups), step + asynchronous resample with the draws made by K1, and with k_draw_slots.
    python tools/k1_scaling.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench as B  # noqa: E402  (observations only)
import modppl_amd  # noqa: E402
from modppl_amd import capi  # noqa: E402

ys = B.lgssm_observations(64).reshape(64, 1)
for logn in (17, 18, 19, 20, 21):
    n = 1 << logn
    row = []
    for mode in ("plain", "resample"):
        pf = modppl_amd.ParticleSystem(modppl_amd.lgssm_model(*B.LGSSM_PARAMS), n, 7)
        pf.init_step(None, ys[:1])
        for t in range(1, 12):
            pf.step(ys[t:t + 1])
            if mode == "resample":
                pf.resample(sync=False)
        pf.synchronize()
        pf.set_timing(True)
        for t in range(12, 62):
            pf.step(ys[t:t + 1])
            if mode == "resample":
                pf.resample(sync=False)
        pf.synchronize()
        k1 = pf.get_timing(capi.MP_K_PROPAGATE)
        dr = pf.get_timing(capi.MP_K_BIN_DRAWS)
        row.append(f"{mode}: K1 {k1[0] / k1[1] * 1e3:6.2f} us" + (f" + draws {dr[0] / dr[1] * 1e3:5.2f}" if dr[1] else ""))
        pf.close()
    print(f"n = 2^{logn} ({n // 2048:5d} workgroups, {n / 2048 / 256:4.2f} per CU): " + "; ".join(row), flush=True)
