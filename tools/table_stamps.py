#!/usr/bin/env python3
"""Diagnostics: the timeline of ONE sharded resample + step on one rank of an emulated world (tools/route_scale.py's set-up), from
the -DMP_STAMPS build: k_shard_table_mw's phases (real-time counter, 100 MHz), k_shard_self_place, and the propagate kernel that
follows, all on one clock — so the gaps between the launches show too.  Shares and gaps, not kernel times.

    python tools/table_stamps.py [--world 8] [--schemes 3,1,2] [--particles N]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

KERNELS, MAX_WG, SLOTS = 4, 16384, 32
PHASES = ["loads + max", "exp / quantise / scan", "ticket round trip", "prefix over ranks + stores", "counts / ranges (fold alongside)", "plan + verdict", "placement"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=int, default=1 << 20)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--schemes", default="3,1,2,0")
    ap.add_argument("--two-calls", action="store_true", help="count, then expand (the placement as a launch of its own)")
    args = ap.parse_args()
    from modppl_amd import build as B

    os.environ["MODPPL_HIP_LIB"] = os.environ.get("MP_STAMPS_LIB") or B.build_stamps()
    import torch
    import modppl_amd
    from modppl_amd import capi
    from modppl_amd.distributed import HipShardEngine
    import bench as BN

    L = capi.load()
    L.mp_debug_stamps.argtypes = [C.c_void_p]
    n, world = args.particles, args.world
    model = modppl_amd.lgssm_model(*BN.LGSSM_PARAMS)
    ys = BN.lgssm_observations(80).reshape(80, 1)
    eng = HipShardEngine(model, n, n * world, 0, 7)
    nt = n // 2048
    tiles = torch.zeros(3 * nt, dtype=torch.int64, device=eng.device)
    eng.shard_bind_tiles(C.c_void_p(tiles.data_ptr()))
    eng.init_step(None, ys[:1])
    eng.shard_tiles_packed(C.c_void_p(tiles.data_ptr()))
    eng.synchronize()
    tiles_all = tiles.repeat(world).contiguous()
    torch.cuda.synchronize()
    ocap = max(4096, n // 128)
    send = torch.zeros(world * ocap * 2, dtype=torch.float64, device=eng.device)
    orow = torch.zeros((world * ocap + n) * 2, dtype=torch.float64, device=eng.device)
    names = {0: "multinomial (owned)", 1: "systematic", 2: "stratified", 3: "split multinomial"}

    def one(scheme, t):
        if args.two_calls:
            eng.shard_owned_count(scheme, C.c_void_p(tiles_all.data_ptr()), world, 0, ocap, want_counts=False)
            eng.shard_owned_expand(world, 0, ocap, C.c_void_p(send.data_ptr()), C.c_void_p(orow.data_ptr()), world * ocap)
        else:
            eng.shard_owned_count_expand(scheme, C.c_void_p(tiles_all.data_ptr()), world, 0, ocap, C.c_void_p(send.data_ptr()),
                                         C.c_void_p(orow.data_ptr()), world * ocap)
        eng.shard_owned_commit(C.c_void_p(orow.data_ptr()), world * ocap, False, want_counts=False)
        eng.step(ys[1 + t % 60:2 + t % 60])
        eng.synchronize()
        tiles_all.copy_(tiles.repeat(world))
        torch.cuda.synchronize()

    for scheme in [int(s) for s in args.schemes.split(",")]:
        for t in range(6):
            one(scheme, t)
        acc = []
        for rep in range(8):
            capi.check(L.mp_debug_stamps(None))
            one(scheme, 6 + rep)
            buf = np.zeros((KERNELS, MAX_WG, SLOTS), dtype=np.uint64)
            capi.check(L.mp_debug_stamps(buf.ctypes.data_as(C.c_void_p)))
            tb = buf[2][:world].astype(np.int64)
            if not tb[0, 0]:
                continue
            t0 = tb[:, 0].min()
            row = {"wg_start_spread": (tb[:, 0].max() - t0) / 100.0}
            w0 = tb[0]
            last = 0
            for k in range(7):
                if w0[k + 1]:
                    row[PHASES[k]] = (w0[k + 1] - w0[k]) / 100.0
                    last = k + 1
            if w0[9]:   # inside the split counts: what does not depend on a node's draw count, then the levels of the tree
                row["    split counts: uniforms, masses"] = (w0[9] - w0[4]) / 100.0
                for lv in range(6):
                    if w0[10 + lv]:
                        row[f"    split counts: level {lv}"] = (w0[10 + lv] - w0[9 + lv]) / 100.0
            row["table launch, leader's end"] = (w0[last] - t0) / 100.0
            end = w0[last]
            pl = buf[3][:64].astype(np.int64)
            if pl[0, 0]:
                row["gap table -> place launch"] = (pl[:, 0].min() - w0[last]) / 100.0
                row["place launch"] = (pl[:, 2].max() - pl[:, 0].min()) / 100.0
                end = pl[:, 2].max()
            k1 = buf[0].astype(np.int64)
            live = k1[:, 1] != 0
            if live.any():
                row["gap -> propagate (host: verdict poll, commit, launch)"] = (k1[live, 1].min() - end) / 100.0
                row["propagate span"] = (k1[live, 5].max() - k1[live, 1].min()) / 100.0
                row["table start -> propagate end"] = (k1[live, 5].max() - t0) / 100.0
            acc.append(row)
        if not acc:
            print(f"world {world}: {names[scheme]}: no stamps (k_shard_table_mw did not run)")
            continue
        print(f"world {world}: {names[scheme]}: medians over {len(acc)} resamples, us", flush=True)
        for k in acc[0]:
            vals = [r[k] for r in acc if k in r]
            print(f"    {k:56s} {np.median(vals):7.2f}   (min {min(vals):.2f}, max {max(vals):.2f})")
    eng.close()


if __name__ == "__main__":
    main()
