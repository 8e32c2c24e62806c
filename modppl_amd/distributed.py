"""Sharded particle filter: one process per GPU, particles split into contiguous global-slot ranges.

Collectives per resample (torch.distributed; backend "nccl" IS RCCL on ROCm, over xGMI on one node):

    all_gather       24 B/tile  per-tile max log-weight + fixed-point totals of every shard: the weight "all-reduce";
                                every rank builds the same tile table from them
    all_to_all       u64        the draws (tile, local target), routed to the rank that owns their CDF range
    all_to_all       f64 rows   the parents' states + global ids back to the asking rank (the particle exchange)

Device-resident groups (RCCL) use fixed-capacity, equal-split all-to-alls (counts travel in the segment headers), so a
resample enqueues 4 library calls + 3 collectives and touches the host once, at the end.  If one pair of ranks needs
more than the capacity (collapsed weights) that resample is repeated with exact split sizes, which costs one more
all-to-all of counts and a host round trip in the middle.  Host-staged groups (gloo) always use the exact-size form.

Because Philox is keyed by GLOBAL slot and the fixed-point scale is global, the filter's results do not
depend on the number of shards (tests/test_distributed_cpu.py checks this bit for bit with 2 ranks): exchange="exact".

exchange="owned" (the default) is the form that fits point-to-point xGMI: the same N draws, but an offspring stays on the rank
that owns its parent and only each rank's surplus over its n slots travels (include/modppl_hip.h, "owner keeps"):

    all_gather       24 B/tile  as above
    all_to_all       f64 rows   the surplus: O(sqrt(N)) rows, fixed capacity per pair (exact sizes after an overflow)

The multiset of parents is the single filter's; which slot an offspring occupies depends on the world size.
"""
import contextlib
import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist

from . import capi


class HipShardEngine:
    """The product's local engine: the gfx950 kernels through the C ABI (device pointers)."""

    def __init__(self, model, n_local, n_global, slot_offset, seed, device_index=0):
        self._L = capi.load()
        self.model = model
        self.n = int(n_local)
        self.device = torch.device("cuda", device_index)
        self._desc = model.desc()
        sh = capi.Shard(int(n_global), int(slot_offset))
        h = C.c_void_p()
        # One dedicated (non-default) stream shared by the kernels, torch's copies and the collectives: torch's
        # default stream has handle 0, which the C ABI reads as "create your own".
        self._stream = torch.cuda.Stream(self.device)
        capi.check(self._L.mp_pf_create(C.byref(self._desc), self.n, int(seed), C.byref(sh), 0, device_index,
                                        C.c_void_p(self._stream.cuda_stream), C.byref(h)))
        self._h = h

    def stream_ctx(self):
        return torch.cuda.stream(self._stream)

    def init_step(self, args0, obs):
        a = None if args0 is None else np.ascontiguousarray(args0, dtype=np.float64)
        capi.check(self._L.mp_pf_init_step(self._h, a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None,
                                           obs.ctypes.data_as(C.POINTER(C.c_double)), obs.shape[0]))

    def step(self, obs):
        capi.check(self._L.mp_pf_step(self._h, obs.ctypes.data_as(C.POINTER(C.c_double)), obs.shape[0]))

    def shard_tiles(self, tm_ptr, tw_ptr, tw2_ptr):
        capi.check(self._L.mp_pf_shard_tiles(self._h, tm_ptr, tw_ptr, tw2_ptr))

    def shard_route(self, scheme, tm_ptr, tw_ptr, tw2_ptr, nt_all, world, rank, req_ptr):
        counts = (C.c_int64 * world)()
        capi.check(self._L.mp_pf_shard_route(self._h, scheme, tm_ptr, tw_ptr, tw2_ptr, world, rank, req_ptr, counts))
        return list(counts)

    def shard_resolve(self, req_ptr, n_req, rows_ptr):
        capi.check(self._L.mp_pf_shard_resolve(self._h, req_ptr, n_req, rows_ptr))

    def shard_scatter(self, rows_ptr, want_value):
        out = C.c_double()
        capi.check(self._L.mp_pf_shard_scatter(self._h, rows_ptr, C.byref(out) if want_value else None))
        return out.value if want_value else None

    # fixed-capacity form (no host round trip until the scatter): see include/modppl_hip.h
    supports_fixed = True

    def shard_bind_tiles(self, tiles_ptr):
        capi.check(self._L.mp_pf_shard_bind_tiles(self._h, tiles_ptr))

    def shard_tiles_packed(self, tiles_ptr):
        capi.check(self._L.mp_pf_shard_tiles_packed(self._h, tiles_ptr))

    def shard_route_fixed(self, scheme, tiles_all_ptr, world, rank, cap, req_ptr):
        capi.check(self._L.mp_pf_shard_route_fixed(self._h, scheme, tiles_all_ptr, world, rank, cap, req_ptr))

    def shard_resolve_fixed(self, req_ptr, world, cap, rows_ptr):
        capi.check(self._L.mp_pf_shard_resolve_fixed(self._h, req_ptr, world, cap, rows_ptr))

    def shard_commit_fixed(self, rows_ptr, want_value):
        """-> (committed, log total weight or None)"""
        out = C.c_double()
        code = self._L.mp_pf_shard_commit_fixed(self._h, rows_ptr, C.byref(out) if want_value else None)
        if code == capi.MP_ERR_CAPACITY:
            return False, None
        capi.check(code)
        return True, (out.value if want_value else None)

    # "owner keeps" form
    supports_owned = True
    # ... and the whole resample as ONE library call that issues the collectives itself (mp_pf_shard_resample, include/modppl_hip.h)
    supports_native = True

    def shard_resample_native(self, transport, world, rank, scheme, force, want_value):
        out = C.c_double()
        capi.check(self._L.mp_pf_shard_resample(self._h, C.byref(transport) if transport is not None else None, world, rank, scheme, int(force),
                                                C.byref(out) if want_value else None))
        return out.value if want_value else None

    def shard_query_native(self, transport, world, force):
        lml, ess = C.c_double(), C.c_double()
        capi.check(self._L.mp_pf_shard_query_native(self._h, C.byref(transport) if transport is not None else None, world, int(force),
                                                    C.byref(lml), C.byref(ess)))
        return lml.value, ess.value

    def shard_native_stats(self, world):
        fb, rows, cap = C.c_uint64(), C.c_uint64(), C.c_uint64()
        counts = (C.c_uint64 * 64)()
        capi.check(self._L.mp_pf_shard_resample_stats(self._h, C.byref(fb), C.byref(rows), counts, C.byref(cap)))
        return fb.value, (None if rows.value == 2 ** 64 - 1 else rows.value), list(counts)[:world], cap.value

    def stream_copy(self, dst_ptr, src_ptr, nbytes, to_host):
        capi.check(self._L.mp_pf_stream_copy(self._h, dst_ptr, src_ptr, nbytes, int(to_host)))

    def shard_owned_count(self, scheme, tiles_all_ptr, world, rank, cap=0, want_counts=True):
        counts = (C.c_uint64 * world)()
        capi.check(self._L.mp_pf_shard_owned_count(self._h, scheme, tiles_all_ptr, world, rank, cap, counts if want_counts else None))
        return list(counts) if want_counts else None

    def shard_owned_expand(self, world, rank, cap, send_ptr, rows_ptr, recv_rows):
        capi.check(self._L.mp_pf_shard_owned_expand(self._h, world, rank, cap, send_ptr, rows_ptr, recv_rows))

    def shard_owned_count_expand(self, scheme, tiles_all_ptr, world, rank, cap, send_ptr, rows_ptr, recv_rows):
        """count + expand of the equal-split form in one call (a self-drawn resample then needs one launch for both)"""
        capi.check(self._L.mp_pf_shard_owned_count_expand(self._h, scheme, tiles_all_ptr, world, rank, cap, send_ptr, rows_ptr, recv_rows))

    def shard_owned_commit(self, rows_ptr, recv_rows, want_value, want_counts=True):
        """-> (committed, log total weight or None, offspring per rank or None).  Without value and counts the call waits for
        the plan's verdict word only (and not at all in a world of one); with either, for the stream."""
        out = C.c_double()
        counts = (C.c_uint64 * 64)() if want_counts else None   # SH_MAX_WORLD entries: the library writes the world's
        code = self._L.mp_pf_shard_owned_commit(self._h, rows_ptr, C.byref(out) if want_value else None, counts)
        if code == capi.MP_ERR_CAPACITY:
            return False, None, (list(counts) if want_counts else None)
        capi.check(code)
        return True, (out.value if want_value else None), (list(counts) if want_counts else None)

    def shard_query_packed(self, tiles_all_ptr, world):
        lml, ess = C.c_double(), C.c_double()
        capi.check(self._L.mp_pf_shard_query_packed(self._h, tiles_all_ptr, world, C.byref(lml), C.byref(ess)))
        return lml.value, ess.value

    def shard_query(self, tm_ptr, tw_ptr, tw2_ptr, nt_all):
        lml, ess = C.c_double(), C.c_double()
        capi.check(self._L.mp_pf_shard_query(self._h, tm_ptr, tw_ptr, tw2_ptr, self._world_of(nt_all), C.byref(lml), C.byref(ess)))
        return lml.value, ess.value

    def _world_of(self, nt_all):
        nt_local = (self.n + 2047) // 2048
        return nt_all // nt_local

    def ess_reference(self):
        out = C.c_double()
        capi.check(self._L.mp_pf_effective_sample_size(self._h, capi.MP_ESS_REFERENCE, C.byref(out)))
        return out.value

    def states(self):
        x = np.empty((self.n, self.model.dim_state))
        capi.check(self._L.mp_pf_read_state(self._h, x.ctypes.data_as(C.POINTER(C.c_double))))
        return x

    def log_weights(self):
        w = np.empty(self.n)
        capi.check(self._L.mp_pf_read_log_weights(self._h, w.ctypes.data_as(C.POINTER(C.c_double))))
        return w

    def parents(self):
        p = np.empty(self.n, dtype=np.uint32)
        capi.check(self._L.mp_pf_read_parents(self._h, p.ctypes.data_as(C.POINTER(C.c_uint32))))
        return p

    def synchronize(self):
        capi.check(self._L.mp_pf_synchronize(self._h))

    def set_timing(self, on):
        capi.check(self._L.mp_pf_set_timing(self._h, int(on)))

    def get_timing(self, family):
        ms, n = C.c_double(), C.c_uint64()
        capi.check(self._L.mp_pf_get_timing(self._h, family, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def region_begin(self):
        capi.check(self._L.mp_pf_region_begin(self._h))

    def region_end(self):
        ms, n = C.c_double(), C.c_uint64()
        capi.check(self._L.mp_pf_region_end(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def last_propagate_form(self):
        out = C.c_int32()
        capi.check(self._L.mp_pf_last_propagate_form(self._h, C.byref(out)))
        return out.value

    def close(self):
        if getattr(self, "_h", None):
            self._L.mp_pf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rccl_transport(L, group, world, rank, device_index):
    """A communicator of the LIBRARY's own over the ranks of `group` (mp_rccl_*): rank 0 makes the id, torch.distributed only
    carries its 128 bytes to the others.  -> (Transport, comm handle)"""
    ident = (C.c_ubyte * 128)()
    err = None
    if rank == 0:
        try:
            capi.check(L.mp_rccl_unique_id(ident))
        except capi.ModpplError as e:
            err = e
    # rank 0 ALWAYS broadcasts (None = "no id"), so every rank runs the same sequence of collectives whatever failed where
    box = [bytes(ident) if err is None else None]
    if world > 1:
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if box[0] is None:
        raise err or capi.ModpplError(capi.MP_ERR_UNSUPPORTED, "rank 0 could not make an RCCL id")
    ident = (C.c_ubyte * 128).from_buffer_copy(box[0])
    comm = C.c_void_p()
    capi.check(L.mp_rccl_comm_create(world, rank, ident, device_index, C.byref(comm)))
    t = capi.Transport()
    capi.check(L.mp_transport_rccl(comm, C.byref(t)))
    return t, comm


class HostStagedTransport:
    """mp_transport over a process group that cannot take device buffers (gloo; ranks sharing one GPU in tests): the library
    calls back with device pointers, the callbacks stage through host memory (mp_pf_stream_copy) and run the collective on CPU
    tensors.  Only for tests: every collective costs two copies and a stream wait."""

    def __init__(self, engine, group, world):
        self.engine, self.group, self.world = engine, group, world
        self._ag = capi.ALL_GATHER_FN(self._all_gather)
        self._aa = capi.ALL_TO_ALL_FN(self._all_to_all)
        self.struct = capi.Transport(None, self._ag, self._aa)
        self.calls = {"all_gather": 0, "all_to_all": 0}

    def _all_gather(self, ctx, d_send, d_recv, nbytes, stream):
        try:
            self.calls["all_gather"] += 1
            mine = torch.empty(nbytes, dtype=torch.uint8)
            self.engine.stream_copy(C.c_void_p(mine.data_ptr()), C.c_void_p(d_send), nbytes, True)
            parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(self.world)]
            dist.all_gather(parts, mine, group=self.group)
            allp = torch.cat(parts)
            self.engine.stream_copy(C.c_void_p(d_recv), C.c_void_p(allp.data_ptr()), nbytes * self.world, False)
            return 0
        except Exception:   # noqa: BLE001 (an exception must not unwind through the C frames)
            return capi.MP_ERR_HIP

    def _all_to_all(self, ctx, d_send, send_off, send_bytes, d_recv, recv_off, recv_bytes, world, stream):
        try:
            self.calls["all_to_all"] += 1
            outs, ins = [], []
            for q in range(world):
                o = torch.empty(int(send_bytes[q]), dtype=torch.uint8)
                if send_bytes[q]:
                    self.engine.stream_copy(C.c_void_p(o.data_ptr()), C.c_void_p(d_send + int(send_off[q])), int(send_bytes[q]), True)
                outs.append(o)
                ins.append(torch.empty(int(recv_bytes[q]), dtype=torch.uint8))
            dist.all_to_all(ins, outs, group=self.group) if dist.get_backend(self.group) != "gloo" else self._gloo_all_to_all(ins, outs)
            for q in range(world):
                if recv_bytes[q]:
                    self.engine.stream_copy(C.c_void_p(d_recv + int(recv_off[q])), C.c_void_p(ins[q].data_ptr()), int(recv_bytes[q]), False)
            return 0
        except Exception:   # noqa: BLE001
            return capi.MP_ERR_HIP

    def _gloo_all_to_all(self, ins, outs):
        # gloo has no all_to_all: pairwise send / recv, lower rank first
        me = dist.get_rank(self.group)
        ins[me].copy_(outs[me])
        for q in range(self.world):
            if q == me:
                continue
            peer = dist.get_global_rank(self.group, q) if self.group is not None else q
            first, second = ("send", "recv") if me < q else ("recv", "send")
            for op in (first, second):   # (empty pieces are skipped on both sides alike: the sizes come from the same plan)
                if op == "send" and outs[q].numel():
                    dist.send(outs[q], peer, group=self.group)
                if op == "recv" and ins[q].numel():
                    dist.recv(ins[q], peer, group=self.group)


class ShardedParticleSystem:
    """`ParticleSystem` (modppl/src/inference/particle_filter.rs) over a process group: same methods, same results
    as one filter with `num_particles` particles, whatever the world size.

    `host_staging=True` moves the exchanged buffers through host memory (for process groups whose backend
    cannot take device tensors, e.g. gloo with ranks sharing one GPU in tests)."""

    def __init__(self, model, num_particles, seed, *, group=None, engine_cls=HipShardEngine, engine_kwargs=None, host_staging=False,
                 exchange=None):
        self.group = group
        self.exchange = exchange or os.environ.get("MP_SHARD_EXCHANGE", "owned")
        if self.exchange not in ("owned", "exact", "split"):
            raise capi.ModpplError(capi.MP_ERR_INVALID_ARG, "exchange must be 'owned', 'split' or 'exact'")
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # MP_SHARD_ALWAYS_COLLECTIVE=1: issue the collectives even in a world of one (exercises the backend's API path)
        self._always = dist.is_initialized() and os.environ.get("MP_SHARD_ALWAYS_COLLECTIVE", "0") == "1"
        if num_particles % self.world:
            raise capi.ModpplError(capi.MP_ERR_INVALID_ARG, "num_particles must be divisible by the world size")
        self.model = model
        self.num_particles = int(num_particles)
        self.n = self.num_particles // self.world
        self.slot_offset = self.rank * self.n
        self.engine = engine_cls(model, self.n, self.num_particles, self.slot_offset, seed, **(engine_kwargs or {}))
        self.dev = torch.device(self.engine.device)
        self.comm_dev = torch.device("cpu") if host_staging else self.dev
        self._ctx = getattr(self.engine, "stream_ctx", contextlib.nullcontext)
        d = model.dim_state
        if self.world > 1 and self.n % 2048:
            raise capi.ModpplError(capi.MP_ERR_INVALID_ARG, "shards must be tile-aligned: num_particles / world_size must be a multiple of 2048")
        self.nt = (self.n + 2047) // 2048
        # level-0 tiles of this shard, packed [3][nt] as int64 (row 0 = bits of the f64 tile maxima) so ONE all-gather moves them
        self._tiles = torch.zeros(3 * self.nt, dtype=torch.int64, device=self.dev)
        self._tiles_all = torch.zeros(self.world * 3 * self.nt, dtype=torch.int64, device=self.dev)   # rank-major [world][3][nt]
        self._tm_all = torch.zeros(self.world * self.nt, dtype=torch.int64, device=self.dev)           # field-major [world*nt] each
        self._tw_all = torch.zeros(self.world * self.nt, dtype=torch.int64, device=self.dev)
        self._tw2_all = torch.zeros(self.world * self.nt, dtype=torch.int64, device=self.dev)
        self._req = torch.zeros(2 * self.n, dtype=torch.int64, device=self.dev)
        self._rows = torch.zeros(self.n * (d + 1), dtype=torch.float64, device=self.dev)
        # fixed-capacity exchange (device-resident process groups): equal-split all-to-alls, no host round trip until the
        # scatter; a pair of ranks needing more than `cap` draws falls back to the variable-size phases for that resample
        self._fixed = bool(getattr(self.engine, "supports_fixed", False)) and not host_staging \
            and os.environ.get("MP_SHARD_FIXED", "1") == "1"
        self.fallbacks = 0
        # "split": the owner-keeps exchange with the multinomial draws made rank by rank (MP_RESAMPLE_MULTINOMIAL_SPLIT: counts per
        # rank first, then every rank its own — O(n) per rank, the same law, another seeded stream than the single filter's)
        self._owned = self.exchange in ("owned", "split")
        if self._owned and not getattr(self.engine, "supports_owned", False):
            raise capi.ModpplError(capi.MP_ERR_UNSUPPORTED, "this engine has no owner-keeps exchange")
        # The owner-keeps resample of the product's engine is ONE library call (mp_pf_shard_resample): buffers, phases, fallback
        # and collectives live in the library; this class only names the transport.  MP_SHARD_NATIVE=0, or an engine without the
        # entry point (the CPU checker in tests/test_distributed_cpu.py), runs the same protocol from _resample_owned below.
        self._native = self._owned and getattr(self.engine, "supports_native", False) and os.environ.get("MP_SHARD_NATIVE", "1") == "1"
        self._transport, self._rccl_comm, self._staged = None, None, None
        if self._native and (self.world > 1 or self._always):
            if host_staging or dist.get_backend(group) == "gloo":
                self._staged = HostStagedTransport(self.engine, group, self.world)
                self._transport = self._staged.struct
            else:
                # A communicator of the library's own.  If ANY rank cannot resolve an RCCL (no librccl.so.1), EVERY rank takes the
                # round-2 protocol below over torch.distributed instead.  The ranks agree on that from a LOCAL probe
                # (mp_rccl_available: no collective, no rendezvous inside it) before any of them enters the id broadcast or
                # ncclCommInitRank, so every rank runs the same sequence of collectives.  A failure after that agreement (rank 0
                # cannot make an id: every rank learns it from the broadcast; a communicator that does not come up) is an error.
                ok = int(self.engine._L.mp_rccl_available()) if os.environ.get("MP_SHARD_RCCL", "1") == "1" else 0
                if self.world > 1:
                    flag = torch.tensor([ok], dtype=torch.int32, device=self.dev)
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
                    ok = int(flag.item())
                if ok:
                    self._transport, self._rccl_comm = rccl_transport(self.engine._L, group, self.world, self.rank, self.dev.index or 0)
                else:
                    self._native = False
                    if self.rank == 0:
                        import warnings
                        warnings.warn("native RCCL transport unavailable (librccl.so.1 does not resolve on this or another rank): the "
                                      "sharded resample runs its collectives through torch.distributed")
        self._host_staging = host_staging
        self._ow_keep = None
        self._ox_cache, self._ox_flip = {}, 0   # exact-size exchange buffers
        self.last_counts = None          # offspring per rank of the last synchronous owner-keeps resample
        self.last_exchange_rows = None   # surplus rows it moved between ranks
        if self._owned and not self._native:
            # surplus rows per pair of ranks in the equal-split all-to-all: the surplus of a rank is the spread of a
            # Binomial(N, ~1/world) count plus the imbalance of the shard masses, both O(sqrt) of n
            self._ow_fixed = self._fixed
            # (std of a rank's surplus ~ sqrt(2 n): 1400 rows at 2^20 particles per rank, so 8192 rows per pair is ~6 sigma;
            # a pair that needs more falls back once and the capacity doubles)
            ocap = max(4096, self.n // 128)
            self._ow_cap_forced = "MP_SHARD_OWNED_CAP" in os.environ   # tests: force the overflow path
            self._ow_cap = min(self.n, int(os.environ.get("MP_SHARD_OWNED_CAP", ocap)))
            # The equal-split all-to-all moves `cap` rows to every peer whatever the surplus is: fine for 16-byte rows, not
            # for wide states whose shard masses differ by percents (LGSSM d = 16: 3 % of 2^21 particles, 136-byte rows ->
            # 125 MB of padding per rank and step at 8 ranks).  Beyond this many padded bytes per rank the exchange uses exact
            # sizes (one host round trip per resample, a few percent of such a step).
            self._ow_fixed_max_bytes = int(os.environ.get("MP_SHARD_OWNED_FIXED_MAX_BYTES", 4 << 20))
            if self._ow_fixed and self._ow_padded_bytes(self._ow_cap) > self._ow_fixed_max_bytes and not self._ow_cap_forced:
                self._ow_fixed = False
            if self._ow_fixed:
                self._alloc_owned(self._ow_cap)
        if self._fixed and not self._native:
            slack = float(os.environ.get("MP_SHARD_SLACK", "1.25"))
            per = self.n // (min(8, self.nt) * self.world)   # draws per (owner, eighth of the owner's tiles) sub-segment, on average
            cap = min(self.n, int(per * slack) + 512)
            self._cap_forced = "MP_SHARD_CAP" in os.environ       # tests: force the overflow path
            cap = int(os.environ.get("MP_SHARD_CAP", cap))
            self._p_tiles = C.c_void_p(self._tiles.data_ptr())
            self.engine.shard_bind_tiles(self._p_tiles)   # the filter keeps its tiles in the tensor the all-gather reads
            w = self.world
            self._p_tiles_all = C.c_void_p(self._tiles.data_ptr() if (w == 1 and not self._always) else self._tiles_all.data_ptr())
            if not self._owned:
                self._alloc_fixed(cap)
        if self.dev.type == "cuda":
            torch.cuda.synchronize(self.dev)   # the buffers above were zero-filled on torch's current stream; the filter runs on its own

    def _alloc_fixed(self, cap):
        """exchange buffers for `cap` draws per (owner, eighth) sub-segment"""
        w, d = self.world, self.model.dim_state
        self.cap = int(cap)
        self._fx_req_out = torch.zeros(w * 8 * (cap + 1) * 2, dtype=torch.int64, device=self.dev)
        self._fx_req_in = self._fx_req_out if (w == 1 and not self._always) else torch.zeros_like(self._fx_req_out)
        self._fx_rows_out = torch.zeros(w * 8 * cap * (d + 1), dtype=torch.float64, device=self.dev)
        self._fx_rows_in = self._fx_rows_out if (w == 1 and not self._always) else torch.zeros_like(self._fx_rows_out)
        self._p_req_out, self._p_req_in = C.c_void_p(self._fx_req_out.data_ptr()), C.c_void_p(self._fx_req_in.data_ptr())
        self._p_rows_out, self._p_rows_in = C.c_void_p(self._fx_rows_out.data_ptr()), C.c_void_p(self._fx_rows_in.data_ptr())

    def _ox_buffer(self, name, numel):
        key = (name, self._ox_flip)
        buf = self._ox_cache.get(key)
        if buf is None or buf.numel() < numel:
            buf = torch.empty(int(numel * 1.25) + 64, dtype=torch.float64, device=self.dev)
            self._ox_cache[key] = buf
        return buf

    def _ow_padded_bytes(self, cap):
        return self.world * int(cap) * (self.model.dim_state + 1) * 8

    def _alloc_owned(self, cap):
        w, d = self.world, self.model.dim_state
        self._ow_cap = int(cap)
        self._ow_send = torch.zeros(w * cap * (d + 1), dtype=torch.float64, device=self.dev)
        # two row buffers, used alternately: the one the previous resample filled is still read by the propagate in flight
        self._ow_rows = [torch.zeros((w * cap + self.n) * (d + 1), dtype=torch.float64, device=self.dev) for _ in range(2)]
        self._ow_flip = 0

    # ---- collectives (identical for nccl/device tensors and gloo/CPU tensors) ----
    def _c(self, t):
        return t if t.device == self.comm_dev else t.to(self.comm_dev)

    def _all_gather(self, out, t):
        if self.world == 1 and not self._always:
            out.copy_(t)
            return
        c_in, c_out = self._c(t), self._c(out)
        dist.all_gather_into_tensor(c_out, c_in, group=self.group) if c_out.device.type != "cpu" else \
            dist.all_gather(list(c_out.view(self.world, -1).unbind(0)), c_in, group=self.group)
        if c_out is not out:
            out.copy_(c_out)

    def _all_to_all(self, send, send_counts, recv_counts, width):
        """variable all-to-all of rows of `width` elements; returns the receive buffer (on self.dev)."""
        n_recv = int(sum(recv_counts))
        if self.world == 1 and not self._always:
            return send[: n_recv * width]
        c_send = self._c(send[: int(sum(send_counts)) * width].contiguous())
        c_recv = torch.empty(n_recv * width, dtype=send.dtype, device=self.comm_dev)
        dist.all_to_all_single(c_recv, c_send, output_split_sizes=[int(c) * width for c in recv_counts],
                               input_split_sizes=[int(c) * width for c in send_counts], group=self.group)
        return c_recv if c_recv.device == self.dev else c_recv.to(self.dev)

    # ---- ParticleSystem methods ----
    def _obs(self, constraints):
        obs = np.ascontiguousarray(constraints, dtype=np.float64)
        if obs.size == 0 or obs.size % self.model.dim_obs:
            raise capi.ModpplError(capi.MP_ERR_CONSTRAINTS, "constraints must hold dim_obs values per time step")
        return obs.reshape(-1, self.model.dim_obs)

    def init_step(self, args, constraints):
        self.engine.init_step(args, self._obs(constraints))

    def step(self, constraints):
        self.engine.step(self._obs(constraints))
        return self

    def _normalize(self):
        """level 0 on every shard, then ONE all-gather of the tiles (24 B each): the all-reduce of log-weights."""
        nt, w = self.nt, self.world
        t = self._tiles
        self.engine.shard_tiles(C.c_void_p(t.data_ptr()), C.c_void_p(t.data_ptr() + 8 * nt), C.c_void_p(t.data_ptr() + 16 * nt))
        self._all_gather(self._tiles_all, t)
        g = self._tiles_all.view(w, 3, nt)
        self._tm_all.copy_(g[:, 0, :].reshape(-1))
        self._tw_all.copy_(g[:, 1, :].reshape(-1))
        self._tw2_all.copy_(g[:, 2, :].reshape(-1))

    def _tile_ptrs(self):
        return (C.c_void_p(self._tm_all.data_ptr()), C.c_void_p(self._tw_all.data_ptr()), C.c_void_p(self._tw2_all.data_ptr()), self.world * self.nt)

    def resample(self, scheme=capi.MP_RESAMPLE_MULTINOMIAL, sync=True):
        """resample() -> log total weight (particle_filter.rs:103-116), multinomial over ALL shards."""
        if self.exchange == "split" and scheme == capi.MP_RESAMPLE_MULTINOMIAL:
            scheme = capi.MP_RESAMPLE_MULTINOMIAL_SPLIT
        with self._ctx():
            return self._resample(scheme, sync)

    def _gather_tiles_packed(self):
        self.engine.shard_tiles_packed(self._p_tiles)
        if self.world > 1 or self._always:
            dist.all_gather_into_tensor(self._tiles_all, self._tiles, group=self.group)

    def _resample_fixed(self, scheme, sync):
        """3 collectives (all-gather of tiles, all-to-all of draws, all-to-all of rows), 4 library calls, one host wait
        (for the owner-side resolve, while the rows travel)."""
        e, w, cap = self.engine, self.world, self.cap
        self._gather_tiles_packed()
        e.shard_route_fixed(scheme, self._p_tiles_all, w, self.rank, cap, self._p_req_out)
        if w > 1 or self._always:
            dist.all_to_all_single(self._fx_req_in, self._fx_req_out, group=self.group)
        e.shard_resolve_fixed(self._p_req_in, w, cap, self._p_rows_out)
        if w > 1 or self._always:
            dist.all_to_all_single(self._fx_rows_in, self._fx_rows_out, group=self.group)
        return e.shard_commit_fixed(self._p_rows_in, sync)   # waits for the resolve only; the rows may still be in flight

    @staticmethod
    def owned_plan(c_all, n):
        """amount[r][s]: rows rank r sends to rank s when unit u of the surplus fills unit u of the deficit"""
        w = len(c_all)
        S = [max(int(c) - n, 0) for c in c_all]
        D = [max(n - int(c), 0) for c in c_all]
        PS = [sum(S[:r]) for r in range(w)]
        PD = [sum(D[:r]) for r in range(w)]
        return [[max(0, min(PS[r] + S[r], PD[s] + D[s]) - max(PS[r], PD[s])) for s in range(w)] for r in range(w)]

    def _resample_owned(self, scheme, sync):
        if self._native:
            value = self.engine.shard_resample_native(self._transport, self.world, self.rank, scheme, self._always, sync)
            self.fallbacks, rows, counts, _ = self.engine.shard_native_stats(self.world)
            # (an asynchronous resample that kept to its capacity never reads its counts back: None, not an earlier resample's)
            self.last_counts, self.last_exchange_rows = (counts, rows) if rows is not None else (None, None)
            return value
        e, w, d = self.engine, self.world, self.model.dim_state
        p_tiles = C.c_void_p(self._tiles.data_ptr())
        e.shard_tiles_packed(p_tiles)
        solo = w == 1 and not self._always
        if solo:
            p_tiles_all = p_tiles
        else:
            self._all_gather(self._tiles_all, self._tiles)
            p_tiles_all = C.c_void_p(self._tiles_all.data_ptr())
        counts = None
        if self._ow_fixed:
            # 2 collectives, 3 library calls, one host wait (for the plan, while the rows are written and travel)
            cap = self._ow_cap
            rows = self._ow_rows[self._ow_flip]
            self._ow_flip ^= 1
            if hasattr(e, "shard_owned_count_expand"):   # one call (and, for a self-drawn resample, one launch): the HIP engine
                e.shard_owned_count_expand(scheme, p_tiles_all, w, self.rank, cap, C.c_void_p(self._ow_send.data_ptr()), C.c_void_p(rows.data_ptr()), w * cap)
            else:
                e.shard_owned_count(scheme, p_tiles_all, w, self.rank, cap, want_counts=False)
                e.shard_owned_expand(w, self.rank, cap, C.c_void_p(self._ow_send.data_ptr()), C.c_void_p(rows.data_ptr()), w * cap)
            if not solo:
                dist.all_to_all_single(rows[: w * cap * (d + 1)], self._ow_send, group=self.group)
            p_rows = C.c_void_p(rows.data_ptr())
            done, value, cnts = e.shard_owned_commit(p_rows, w * cap, sync, want_counts=sync)   # a synchronous resample waits for the stream anyway
            if done:
                if cnts is not None:
                    self.last_counts = cnts[:w]
                    self.last_exchange_rows = sum(max(int(c) - self.n, 0) for c in self.last_counts)   # rows that travelled, job-wide
                return value
            self.fallbacks += 1      # some pair of ranks exchanges more than cap rows: exact sizes this time
            _, _, counts = e.shard_owned_commit(p_rows, w * cap, False, want_counts=True)   # same verdict, now with the counts
        else:
            counts = e.shard_owned_count(scheme, p_tiles_all, w, self.rank, 0, want_counts=True)
        self.last_counts = [int(c) for c in counts[:w]]
        self.last_exchange_rows = sum(max(c - self.n, 0) for c in self.last_counts)
        amount = self.owned_plan(counts[:w], self.n)
        send_counts = amount[self.rank]
        recv_counts = [amount[r][self.rank] for r in range(w)]
        n_send, n_recv = sum(send_counts), sum(recv_counts)
        # grow-only buffers, two sets used alternately (the rows of the previous resample are still read by the propagate in
        # flight); with states wider than one double the library copies no kept offspring, so only the received rows need room
        keep_rows = self.n if d == 1 else 0
        send = self._ox_buffer("send", max(n_send, 1) * (d + 1))
        rows = self._ox_buffer("rows", max(n_recv + keep_rows, 1) * (d + 1))
        self._ox_flip ^= 1
        e.shard_owned_expand(w, self.rank, 0, C.c_void_p(send.data_ptr()), C.c_void_p(rows.data_ptr()), n_recv)
        if solo or sum(map(sum, amount)) == 0:
            recv = send          # nobody has a surplus (every rank sees the same counts, so every rank skips the collective)
        else:
            recv = self._all_to_all(send, send_counts, recv_counts, d + 1)
        if n_recv:
            rows[: n_recv * (d + 1)].copy_(recv[: n_recv * (d + 1)])
        done, value, _ = e.shard_owned_commit(C.c_void_p(rows.data_ptr()), n_recv, sync, want_counts=False)
        self._ow_keep = (rows, send)   # the next propagate reads its parents' states from `rows`
        if self._ow_fixed and not self._ow_cap_forced and self._ow_cap < self.n:
            self.synchronize()
            grown = min(self.n, max(2 * self._ow_cap, 2 * max(max(a) for a in amount)))
            if self._ow_padded_bytes(grown) > self._ow_fixed_max_bytes:
                self._ow_fixed = False       # every rank sees the same counts, so every rank switches alike
            else:
                self._alloc_owned(grown)
        return value

    def _resample(self, scheme, sync):
        if self._owned:
            return self._resample_owned(scheme, sync)
        if self._fixed:
            done, value = self._resample_fixed(scheme, sync)
            if done:
                return value
            self.fallbacks += 1   # collapsed weights: one owner serves (nearly) everybody — exact sizes this time
            value = self._resample_variable(scheme, sync)
            if not self._cap_forced and self.cap < self.n:
                # every rank took this branch (the overflow flag is global), so every rank grows alike; nothing refers to the
                # old buffers any more (the exact-size path scatters eagerly)
                self.synchronize()
                self._alloc_fixed(min(self.n, self.cap + self.cap // 2))
            return value
        return self._resample_variable(scheme, sync)

    def _resample_variable(self, scheme, sync):
        d = self.model.dim_state
        self._normalize()
        tm, tw, tw2, nt_all = self._tile_ptrs()
        send_counts = self.engine.shard_route(scheme, tm, tw, tw2, nt_all, self.world, self.rank, C.c_void_p(self._req.data_ptr()))
        if self.world > 1 or self._always:
            sc = torch.tensor(send_counts, dtype=torch.int64, device=self.comm_dev)
            rc = torch.empty(self.world, dtype=torch.int64, device=self.comm_dev)
            dist.all_to_all_single(rc, sc, group=self.group)
            recv_counts = rc.tolist()
        else:
            recv_counts = list(send_counts)
        req_in = self._all_to_all(self._req, send_counts, recv_counts, 2)           # draws (tile, target) -> owners
        n_req = int(sum(recv_counts))
        rows_out = torch.empty(max(n_req, 1) * (d + 1), dtype=torch.float64, device=self.dev)
        self.engine.shard_resolve(C.c_void_p(req_in.data_ptr()), n_req, C.c_void_p(rows_out.data_ptr()))
        rows_in = self._all_to_all(rows_out, recv_counts, send_counts, d + 1)       # parents' states -> askers (particle exchange)
        if rows_in.data_ptr() != self._rows.data_ptr():
            self._rows[: rows_in.numel()].copy_(rows_in)
        return self.engine.shard_scatter(C.c_void_p(self._rows.data_ptr()), sync)

    def _query(self):
        if self._native:
            return self.engine.shard_query_native(self._transport, self.world, self._always)
        with self._ctx():
            if self._fixed:
                self._gather_tiles_packed()
                return self.engine.shard_query_packed(self._p_tiles_all, self.world)
            self._normalize()
            return self.engine.shard_query(*self._tile_ptrs())

    def log_marginal_likelihood_estimate(self):
        return self._query()[0]

    def effective_sample_size(self, fresh=False):
        if not fresh:
            return self.engine.ess_reference()
        return self._query()[1]

    def maybe_resample(self, ess_fraction=0.5, scheme=capi.MP_RESAMPLE_MULTINOMIAL):
        """ESS-triggered resampling (extension, as ParticleSystem.maybe_resample): resample iff the ESS of the current
        weights over ALL shards is below ess_fraction * N.  Every rank computes the same ESS from the gathered tiles, so
        every rank takes the same branch.  -> (resampled, ess, log total weight or None)."""
        ess = self.effective_sample_size(fresh=True)
        if ess < float(ess_fraction) * self.num_particles:
            return True, ess, self.resample(scheme)
        return False, ess, None

    def states(self):
        return self.engine.states()

    @property
    def log_weights(self):
        return self.engine.log_weights()

    @property
    def parents(self):
        return self.engine.parents()

    def synchronize(self):
        self.engine.synchronize()

    def close(self):
        """the library's own RCCL communicator goes before the filter whose stream it was used on"""
        if getattr(self, "_rccl_comm", None):
            self.engine.synchronize()
            self.engine._L.mp_rccl_comm_destroy(self._rccl_comm)
            self._rccl_comm = None
        if getattr(self, "engine", None) is not None and hasattr(self.engine, "close"):
            self.engine.close()

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001
            pass
