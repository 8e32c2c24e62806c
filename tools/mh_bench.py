#!/usr/bin/env python3
"""C4 (BASELINE configs[3]: hierarchical model, 2^20 chains, regen_mh cycling the coefficient masks + drift mh) over the
hand-written kernels and over the same model as a registered functor run by the generic handlers (mp_genfn.h).
    python tools/mh_bench.py [chains] [iters]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import modppl_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
xs = np.arange(-5.0, 6.0)
rng = np.random.default_rng(0)
ys = 0.3 + 0.4 * xs + 0.5 * xs * xs + 0.1 * rng.normal(size=xs.size)
out = {}
for functor in (False, True, "data"):   # hand-written kernels; registered functor, every site in registers (kind 101); the same with declared data sites (kind 105)
    g = modppl_amd.HierarchicalChains(xs, ys, n, 3, constrain_is_linear=False, functor=functor)
    g.regen_mh(["coeffs/a", "coeffs/b", "coeffs/c"], 30, cycle=True)
    g.mh(0.1, 10)
    res = {}
    for name, fn in (("regen_mh_cycle", lambda: g.regen_mh(["coeffs/a", "coeffs/b", "coeffs/c"], iters, cycle=True)),
                     ("mh_drift", lambda: g.mh(0.05, iters)), ("mh_add_or_remove", lambda: g.mh_add_or_remove(iters))):
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        res[name] = {"chain_iters_per_s": n * iters / dt, "ms": dt * 1e3}
    out["functor_data_sites" if functor == "data" else ("functor" if functor else "handwritten")] = res
    g.close()
out["ratio_functor_over_handwritten"] = {k: out["functor"][k]["chain_iters_per_s"] / out["handwritten"][k]["chain_iters_per_s"] for k in out["functor"]}
out["ratio_functor_data_sites_over_handwritten"] = {k: out["functor_data_sites"][k]["chain_iters_per_s"] / out["handwritten"][k]["chain_iters_per_s"]
                                                    for k in out["functor_data_sites"]}
print(json.dumps(out, indent=1))
