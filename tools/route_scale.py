#!/usr/bin/env python3
"""What the sharded route costs on ONE rank as the world grows (the tile table every workgroup builds covers the whole job):
a filter created as rank 0 of `world`, fed a gathered-tiles buffer made of `world` copies of its own tiles."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    import argparse
    import numpy as np
    import modppl_amd
    from modppl_amd import capi
    from modppl_amd.distributed import HipShardEngine
    import bench as B   # observations only; nothing under oracle/ is used by the tools

    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="lgssm1", choices=["lgssm1", "band16"], help="band16: the C5 model (its propagate kernel cannot make the draws: k_shard_self_draw does)")
    ap.add_argument("--particles", type=int, default=1 << 20, help="per rank")
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--two-calls", action="store_true", help="count, then expand (before round 5: the placement as a launch of its own)")
    args = ap.parse_args()
    n = args.particles
    if args.model == "band16":
        model, dim = modppl_amd.lgssm_band_model(16), 16
        obs_of = lambda T: np.random.default_rng(3).normal(0, 1.2, size=(T, 16))   # noqa: E731
    else:
        model, dim = modppl_amd.lgssm_model(*B.LGSSM_PARAMS), 1
        obs_of = lambda T: B.lgssm_observations(T).reshape(T, 1)   # noqa: E731
    ys = obs_of(3)
    for world in [int(w) for w in args.worlds.split(",")]:
        eng = HipShardEngine(model, n, n * world, 0, 7)
        nt = n // 2048
        dev = eng.device
        tiles = torch.zeros(3 * nt, dtype=torch.int64, device=dev)
        eng.shard_bind_tiles(C.c_void_p(tiles.data_ptr()))
        eng.init_step(None, ys[:1])
        eng.shard_tiles_packed(C.c_void_p(tiles.data_ptr()))
        eng.synchronize()
        tiles_all = tiles if world == 1 else tiles.repeat(world).contiguous()   # (a world of one gathers nothing: its own, bound buffer)
        torch.cuda.synchronize()   # torch's stream made it; the engine's kernels run on a stream of their own
        cap = int(n // (8 * world) * 1.25) + 512
        req = torch.zeros(world * 8 * (cap + 1) * 2, dtype=torch.int64, device=dev)
        rows = torch.zeros(world * 8 * cap * (dim + 1), dtype=torch.float64, device=dev)
        eng.set_timing(True)
        for _ in range(12):
            eng.shard_route_fixed(0, C.c_void_p(tiles_all.data_ptr()), world, 0, cap, C.c_void_p(req.data_ptr()))
            eng.shard_resolve_fixed(C.c_void_p(req.data_ptr()), world, cap, C.c_void_p(rows.data_ptr()))
        eng.synchronize()
        r = eng.get_timing(capi.MP_K_BIN_DRAWS)
        g = eng.get_timing(capi.MP_K_RESAMPLE_GATHER)
        print(f"world {world}: route (unpack + route + finalize) {r[0] / r[1] * 1e3:.1f} us, resolve + publish {g[0] / g[1] * 1e3:.1f} us", flush=True)
        # the owner-keeps form: multinomial = every rank enumerates all world * n draws and keeps its own; lattice schemes = a
        # rank's own draws are a range found by two searches
        ocap = max(4096, n // 128)
        send = torch.zeros(world * ocap * (dim + 1), dtype=torch.float64, device=dev)
        orow = torch.zeros((world * ocap + n) * (dim + 1), dtype=torch.float64, device=dev)

        def count_expand(scheme):
            # what mp_pf_shard_resample issues: count + expand of the equal-split form as one call (a self-drawn resample: one launch)
            if args.two_calls:
                eng.shard_owned_count(scheme, C.c_void_p(tiles_all.data_ptr()), world, 0, ocap, want_counts=False)
                eng.shard_owned_expand(world, 0, ocap, C.c_void_p(send.data_ptr()), C.c_void_p(orow.data_ptr()), world * ocap)
            else:
                eng.shard_owned_count_expand(scheme, C.c_void_p(tiles_all.data_ptr()), world, 0, ocap, C.c_void_p(send.data_ptr()),
                                             C.c_void_p(orow.data_ptr()), world * ocap)

        for scheme, name in ((0, "multinomial"), (1, "systematic"), (2, "stratified"), (3, "split multinomial")):
            eng.synchronize()
            eng.set_timing(False)
            eng.set_timing(True)
            for _ in range(12):
                count_expand(scheme)
            eng.synchronize()
            r = eng.get_timing(capi.MP_K_BIN_DRAWS)
            g = eng.get_timing(capi.MP_K_RESAMPLE_GATHER)
            place = g[0] / g[1] * 1e3 if g[1] else 0.0   # (a world of one launches nothing there: the next k_propagate looks its draws up)
            count = r[0] / r[1] * 1e3 if r[1] else 0.0   # (a self-drawn resample in a world of one launches nothing at all)
            print(f"world {world}: owner-keeps {name}: count (table + own draws + plan [+ placement, self-drawn]) {count:.1f} us, place + surplus lookups {place:.1f} us",
                  flush=True)
        # a whole step of this rank, kernel by kernel (HIP events around every launch): count + expand + commit (asynchronous) + the
        # next propagate, which makes the kept draws of a self-drawn resample itself (the rows "received" for the deficit slots are
        # whatever the buffer holds: timing only)
        ys2 = obs_of(64)
        for scheme, name in ((0, "multinomial"), (1, "systematic"), (2, "stratified"), (3, "split multinomial")):
            K = 30
            for rep in range(2):
                eng.synchronize()
                eng.set_timing(False)
                eng.set_timing(True)      # (the first pass warms up)
                for t in range(K):
                    count_expand(scheme)
                    eng.shard_owned_commit(C.c_void_p(orow.data_ptr()), world * ocap, False, want_counts=False)
                    eng.step(ys2[1 + t % 60:2 + t % 60])
                    if world > 1:   # the tiles this rank would contribute to the next all-gather (its own, `world` times over)
                        eng.synchronize()
                        tiles_all.copy_(tiles.repeat(world))
                        torch.cuda.synchronize()
                eng.synchronize()
            k1 = eng.get_timing(capi.MP_K_PROPAGATE)
            r = eng.get_timing(capi.MP_K_BIN_DRAWS)
            g = eng.get_timing(capi.MP_K_RESAMPLE_GATHER)
            us = lambda v: v[0] / v[1] * 1e3 if v[1] else 0.0   # noqa: E731
            print(f"world {world}: {name}: kernels of one resample + step: propagate {us(k1):.1f} + count {us(r):.1f} + place {us(g):.1f} "
                  f"= {us(k1) + us(r) + us(g):.1f} us", flush=True)
        eng.close()


if __name__ == "__main__":
    main()
