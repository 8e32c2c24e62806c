"""Reference runs of the "owner keeps" sharded resample (include/modppl_hip.h) for the tests: all shards of a job in ONE
process, the exchange done by hand with numpy.  Test infrastructure: the local engines are the CPU checker's."""
import ctypes as C

import numpy as np

from modppl_amd.distributed import ShardedParticleSystem
from tests import oracle_lib as O


def owned_placement(parents_global, x_before, n, world):
    """Where the owner-keeps rule puts the offspring of ONE filter's resample, restated from its definition with numpy only:
    parents_global[g] = parent of draw g in the single filter (N = n * world draws), x_before = [N][d] states before the
    resample.  -> (x_after [N][d], parent ids [N]) in rank-major slot order."""
    P = np.asarray(parents_global, dtype=np.int64)
    owner = P // n
    lists = [P[owner == r] for r in range(world)]              # a rank's offspring in the order of their draws
    c = [len(l) for l in lists]
    surplus = np.concatenate([l[n:] for l in lists]) if world > 1 else np.empty(0, dtype=np.int64)   # unit order: donors by rank
    out = np.empty(n * world, dtype=np.int64)
    u = 0
    for r in range(world):
        keep = min(c[r], n)
        out[r * n: r * n + keep] = lists[r][:keep]
        need = n - keep
        out[r * n + keep: (r + 1) * n] = surplus[u: u + need]   # receivers by rank, slots c_r, c_r + 1, ...
        u += need
    assert u == len(surplus)
    return x_before[out], out.astype(np.uint32)


class OwnedReference:
    """`world` shards of one job, each an OracleShardEngine; resample() is the owner-keeps protocol with the all-gather and
    the all-to-all replaced by numpy copies."""

    def __init__(self, model, num_particles, seed, world):
        self.model, self.world, self.n = model, world, num_particles // world
        self.nt = (self.n + 2047) // 2048
        self.eng = [O.OracleShardEngine(model, self.n, num_particles, r * self.n, seed) for r in range(world)]

    def init_step(self, args0, obs):
        for e in self.eng:
            e.init_step(args0, obs)

    def step(self, obs):
        for e in self.eng:
            e.step(obs)

    def resample(self, scheme=0):
        w, n, d = self.world, self.n, self.model.dim_state
        tiles = np.zeros((w, 3 * self.nt), dtype=np.int64)
        for r, e in enumerate(self.eng):
            e.shard_tiles_packed(C.c_void_p(tiles[r].ctypes.data))
        p_all = C.c_void_p(tiles.ctypes.data)
        counts = [e.shard_owned_count(scheme, p_all, w, r) for r, e in enumerate(self.eng)]
        assert all(c == counts[0] for c in counts)
        amount = ShardedParticleSystem.owned_plan(counts[0], n)
        sends = []
        for r, e in enumerate(self.eng):
            buf = np.zeros((max(sum(amount[r]), 1), d + 1))
            sent = e.shard_owned_expand(w, r, 0, C.c_void_p(buf.ctypes.data), None, 0)
            assert sent == sum(amount[r])
            sends.append(buf)
        L = None
        for s, e in enumerate(self.eng):
            parts = []
            for r in range(w):
                off = sum(amount[r][:s])
                parts.append(sends[r][off: off + amount[r][s]])
            recv = np.ascontiguousarray(np.concatenate(parts)) if parts else np.zeros((0, d + 1))
            n_recv = recv.shape[0]
            if n_recv == 0:
                recv = np.zeros((1, d + 1))
            _, L, _ = e.shard_owned_commit(C.c_void_p(recv.ctypes.data), n_recv, True)
        self.counts = counts[0]
        return L

    def states(self):
        return np.concatenate([e.states() for e in self.eng])

    def parents(self):
        return np.concatenate([e.parents() for e in self.eng])

    def log_weights(self):
        return np.concatenate([e.log_weights() for e in self.eng])
