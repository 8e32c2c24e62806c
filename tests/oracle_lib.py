"""ctypes loader for oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
SO = os.path.join(ORACLE_DIR, "liboracle.so")

VARIANT_CANONICAL = 1
VARIANT_SOA = 2
VARIANT_FAST_SEARCH = 4


class ModelDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("dim_state", C.c_int32), ("dim_obs", C.c_int32), ("n_params", C.c_int32),
                ("params", C.POINTER(C.c_double))]


class Shard(C.Structure):
    _fields_ = [("n_global", C.c_uint64), ("slot_offset", C.c_uint64)]


def _source_hash():
    import hashlib

    srcs = sorted(os.path.join(ORACLE_DIR, "src", f) for f in os.listdir(os.path.join(ORACLE_DIR, "src")))
    srcs += [os.path.join(ORACLE_DIR, "Makefile")]
    csrc = os.path.join(ROOT, "modppl_amd", "csrc")
    srcs += sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h"))   # the oracle includes mp_math.h & co
    srcs += [os.path.join(ROOT, "include", "modppl_hip.h")]
    h = hashlib.sha256()
    for s in srcs:
        h.update(os.path.basename(s).encode())
        with open(s, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force=False):
    """content-hash staleness (a checkout can give edited files older mtimes): the hash of every source is kept next to the .so"""
    stamp = SO + ".srchash"
    want = _source_hash()
    have = open(stamp).read().strip() if os.path.exists(stamp) and os.path.exists(SO) else None
    if force or have != want:
        subprocess.run(["make", "-B", "-C", ORACLE_DIR, "liboracle.so"], check=True, capture_output=True)
        with open(stamp, "w") as f:
            f.write(want + "\n")
    return SO


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    # On the GPU box the prebuilt .so travels with the snapshot (with its source hash); a mismatch rebuilds it (g++ is there).
    build()
    L = C.CDLL(SO)
    d, i32, u32, i64, u64, p = C.c_double, C.c_int32, C.c_uint32, C.c_int64, C.c_uint64, C.c_void_p
    dp = C.POINTER(C.c_double)
    L.oracle_last_error.restype = C.c_char_p
    L.oracle_pf_create.argtypes = [C.POINTER(ModelDesc), u64, u64, C.POINTER(Shard), u32, i32, C.POINTER(p)]
    L.oracle_pf_set_threads.argtypes = [p, i32]
    L.oracle_pf_init_step.argtypes = [p, dp, dp, i32]
    L.oracle_pf_step.argtypes = [p, dp, i32]
    L.oracle_pf_effective_sample_size.argtypes = [p, i32, dp]
    L.oracle_pf_resample.argtypes = [p, i32, dp]
    L.oracle_pf_log_marginal_likelihood_estimate.argtypes = [p, dp]
    L.oracle_pf_read_state.argtypes = [p, dp]
    L.oracle_pf_read_log_weights.argtypes = [p, dp]
    L.oracle_pf_read_parents.argtypes = [p, C.POINTER(u32)]
    L.oracle_pf_read_trajectory.argtypes = [p, u64, dp, C.POINTER(i32)]
    L.oracle_pf_time.argtypes = [p, C.POINTER(i64)]
    L.oracle_pf_destroy.argtypes = [p]
    L.oracle_importance_resampling.argtypes = [C.POINTER(ModelDesc), dp, dp, i32, u64, u64, u64, i32, dp, dp, C.POINTER(u64), dp]
    L.oracle_pf_shard_tiles.argtypes = [p, p, p, p]
    L.oracle_pf_shard_route.argtypes = [p, i32, p, p, p, u64, i32, i32, p, C.POINTER(i64)]
    L.oracle_pf_shard_resolve.argtypes = [p, p, u64, p]
    L.oracle_pf_shard_scatter.argtypes = [p, p, dp]
    L.oracle_pf_shard_query.argtypes = [p, p, p, p, u64, dp, dp]
    L.oracle_pf_shard_owned_count.argtypes = [p, i32, p, p, p, u64, i32, i32, C.POINTER(u64)]
    L.oracle_pf_shard_owned_expand.argtypes = [p, i32, p, C.POINTER(u64)]
    L.oracle_pf_shard_owned_adopt.argtypes = [p, i32, p, u64, dp]
    L.oracle_mh_create.argtypes = [dp, dp, i32, i32, u64, u64, i32, C.POINTER(p)]
    L.oracle_mh_step.argtypes = [p, d, i32, C.POINTER(u64)]
    L.oracle_unfold_simulate.argtypes = [C.POINTER(ModelDesc), dp, i32, u64, u64, i32, dp, dp]
    L.oracle_mh_step_add_or_remove.argtypes = [p, i32, C.POINTER(u64)]
    L.oracle_mh_pointed_create.argtypes = [dp, dp, dp, u64, u64, i32, C.POINTER(p)]
    L.oracle_mh_pointed_step.argtypes = [p, dp, i32, C.POINTER(u64)]
    L.oracle_mh_pointed_read_state.argtypes = [p, dp]
    L.oracle_mh_pointed_read_logjp.argtypes = [p, dp]
    L.oracle_mh_pointed_destroy.argtypes = [p]
    L.oracle_regen_mh_step.argtypes = [p, C.POINTER(i32), i32, i32, i32, C.POINTER(u64)]
    L.oracle_mh_read_state.argtypes = [p, dp]
    L.oracle_mh_read_logjp.argtypes = [p, dp]
    L.oracle_mh_read_observations.argtypes = [p, i32, dp]
    L.oracle_mh_destroy.argtypes = [p]
    L.oracle_mp_exp.argtypes = [dp, i64, dp]
    L.oracle_mp_log.argtypes = [dp, i64, dp]
    L.oracle_binomial_both.argtypes = [C.POINTER(u64), C.POINTER(u64), C.POINTER(u64), C.POINTER(u32), i64, u64, u32, C.POINTER(u64), C.POINTER(u64)]
    L.oracle_binomial_both.restype = None
    L.oracle_binomial_lanes.argtypes = [C.POINTER(u64), C.POINTER(u64), C.POINTER(u64), C.POINTER(u32), i64, u64, u32, C.POINTER(u64)]
    L.oracle_binomial_lanes.restype = None
    L.oracle_split_counts.argtypes = [C.POINTER(u64), i32, u64, u64, u32, C.POINTER(u64)]
    L.oracle_split_counts.restype = None
    L.oracle_stirling_tail.argtypes = [C.c_double]
    L.oracle_stirling_tail.restype = C.c_double
    L.oracle_mp_div_hoisted.argtypes = [dp, dp, i64, dp]
    L.oracle_mp_div_hoisted.restype = None
    L.oracle_mp_normal_logpdf_both.argtypes = [dp, dp, dp, i64, dp, dp]
    L.oracle_mp_normal_logpdf_both.restype = None
    L.oracle_mp_exp.restype = None
    for f in (L.oracle_mp_sin, L.oracle_mp_cos):
        f.argtypes = [dp, i64, dp]
        f.restype = None
    L.oracle_mp_atan2.argtypes = [dp, dp, i64, dp]
    L.oracle_mp_atan2.restype = None
    L.oracle_mp_log.restype = None
    L.oracle_philox.argtypes = [C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    L.oracle_philox.restype = None
    L.oracle_u01_stream.argtypes = [u64, u32, u32, u32, u32, i64, dp]
    L.oracle_u01_stream.restype = None
    L.oracle_logsumexp.argtypes = [dp, i64, i32]
    L.oracle_logsumexp.restype = d
    L.oracle_normal_logpdf.argtypes = [d, d, d, i32]
    L.oracle_normal_logpdf.restype = d
    L.oracle_normal_random.argtypes = [u64, u32, u32, u32, u32, d, d, i32]
    L.oracle_normal_random.restype = d
    L.oracle_uniform_logpdf.argtypes = [d, d, d]
    L.oracle_uniform_logpdf.restype = d
    L.oracle_bernoulli_logpdf.argtypes = [i32, d]
    L.oracle_bernoulli_logpdf.restype = d
    L.oracle_uniform2d_logpdf.argtypes = [d] * 6
    L.oracle_uniform2d_logpdf.restype = d
    L.oracle_mvnormal_logpdf.argtypes = [i32, dp, dp, dp]
    L.oracle_mvnormal_logpdf.restype = d
    L.oracle_mvnormal_random.argtypes = [u64, u32, i32, dp, dp, dp]
    L.oracle_mvnormal_random.restype = None
    L.oracle_mvnormal_logpdf_dense.argtypes = [i32, dp, dp, dp, i32]
    L.oracle_mvnormal_logpdf_dense.restype = d
    L.oracle_mvnormal_random_dense.argtypes = [u64, u32, u32, u32, u32, i32, dp, dp, i32, dp]
    L.oracle_mvnormal_random_dense.restype = None
    L.oracle_categorical_scan.argtypes = [d, dp, i64]
    L.oracle_categorical_scan.restype = i64
    L.oracle_canonical_normalize.argtypes = [dp, i64, u64, dp, dp, C.POINTER(u64), C.POINTER(u64)]
    L.oracle_canonical_target.argtypes = [u64, u64]
    L.oracle_canonical_target.restype = u64
    L.oracle_kalman_log_ml.argtypes = [dp, dp, i32]
    L.oracle_kalman_log_ml.restype = d
    L.oracle_lgssm_simulate_observations.argtypes = [dp, u64, i32, i32, dp]
    L.oracle_lgssm_simulate_observations.restype = None
    L.oracle_hmm_forward.argtypes = [dp, i32, dp, i32]
    L.oracle_hmm_forward.restype = d
    L.oracle_kat_update_weights.argtypes = [u64, dp]
    L.oracle_kat_residual_panics.argtypes = [u64]
    L.oracle_kat_update.argtypes = [u64, dp]
    L.oracle_kat_regenerate.argtypes = [u64, dp]
    L.oracle_kat_simulate.argtypes = [u64]
    L.oracle_kat_simulate.restype = d
    _lib = L
    return L


def dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class OracleError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"oracle status {code}: {msg}")
        self.code = code


class OraclePF:
    """Drives the restated ParticleSystem with the same call sequence as modppl_amd.ParticleSystem."""

    def __init__(self, kind, dim_state, dim_obs, params, n, seed, variant, shard=None, threads=1):
        self.L = load()
        self.n, self.dim_state, self.dim_obs = n, dim_state, dim_obs
        self._params = np.ascontiguousarray(params, dtype=np.float64)
        desc = ModelDesc(kind, dim_state, dim_obs, len(self._params), dptr(self._params))
        sh = Shard(*shard) if shard else None
        h = C.c_void_p()
        self._ck(self.L.oracle_pf_create(C.byref(desc), n, seed, C.byref(sh) if sh else None, 0, variant, C.byref(h)))
        self.h = h
        if threads > 1:
            self._ck(self.L.oracle_pf_set_threads(self.h, threads))

    def _ck(self, code):
        if code != 0:
            raise OracleError(code, self.L.oracle_last_error().decode())

    def init_step(self, obs, args0=None):
        obs = np.ascontiguousarray(obs, dtype=np.float64).reshape(-1, self.dim_obs)
        a = None if args0 is None else dptr(np.ascontiguousarray(args0, dtype=np.float64))
        self._ck(self.L.oracle_pf_init_step(self.h, a, dptr(obs), obs.shape[0]))

    def step(self, obs):
        obs = np.ascontiguousarray(obs, dtype=np.float64).reshape(-1, self.dim_obs)
        self._ck(self.L.oracle_pf_step(self.h, dptr(obs), obs.shape[0]))

    def effective_sample_size(self, mode=0):
        out = C.c_double()
        self._ck(self.L.oracle_pf_effective_sample_size(self.h, mode, C.byref(out)))
        return out.value

    def resample(self, scheme=0):
        out = C.c_double()
        self._ck(self.L.oracle_pf_resample(self.h, scheme, C.byref(out)))
        return out.value

    def log_marginal_likelihood_estimate(self):
        out = C.c_double()
        self._ck(self.L.oracle_pf_log_marginal_likelihood_estimate(self.h, C.byref(out)))
        return out.value

    def state(self):
        x = np.empty((self.n, self.dim_state))
        self._ck(self.L.oracle_pf_read_state(self.h, dptr(x)))
        return x

    def log_weights(self):
        w = np.empty(self.n)
        self._ck(self.L.oracle_pf_read_log_weights(self.h, dptr(w)))
        return w

    def parents(self):
        p = np.empty(self.n, dtype=np.uint32)
        self._ck(self.L.oracle_pf_read_parents(self.h, p.ctypes.data_as(C.POINTER(C.c_uint32))))
        return p

    def trajectory(self, i, max_t=4096):
        out = np.empty((max_t, self.dim_state))
        t = C.c_int32()
        self._ck(self.L.oracle_pf_read_trajectory(self.h, i, dptr(out), C.byref(t)))
        return out[: t.value].copy()

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.oracle_pf_destroy(self.h)
        except Exception:
            pass


LGSSM_PARAMS = np.array([0.0, 1.0, 0.9, 0.5, 1.0])  # mu0, sig0, a, sig_x, sig_y
DATA_SEED = 20241008


def lgssm_observations(T=50, seed=DATA_SEED, canonical=True, params=LGSSM_PARAMS):
    L = load()
    ys = np.empty(T)
    L.oracle_lgssm_simulate_observations(dptr(np.ascontiguousarray(params)), seed, T, int(canonical), dptr(ys))
    return ys


def kalman_log_ml(ys, params=LGSSM_PARAMS):
    L = load()
    ys = np.ascontiguousarray(ys, dtype=np.float64)
    return L.oracle_kalman_log_ml(dptr(np.ascontiguousarray(params)), dptr(ys), len(ys))


def importance_resampling(kind, dim_state, dim_obs, params, obs, num_samples, num_ret, seed, variant, args0=None):
    """oracle importance_resampling -> (log_ml, log_normalized_weights, indices, final_states)."""
    L = load()
    params = np.ascontiguousarray(params, dtype=np.float64)
    desc = ModelDesc(kind, dim_state, dim_obs, len(params), dptr(params))
    obs = np.ascontiguousarray(obs, dtype=np.float64).reshape(-1, dim_obs)
    lml = C.c_double()
    lnw = np.empty(num_samples)
    idx = np.empty(num_ret, dtype=np.uint64)
    xs = np.empty((num_samples, dim_state))
    a = None if args0 is None else dptr(np.ascontiguousarray(args0, dtype=np.float64))
    rc = L.oracle_importance_resampling(C.byref(desc), a, dptr(obs), obs.shape[0], num_samples, num_ret, seed, variant, C.byref(lml),
                                        dptr(lnw), idx.ctypes.data_as(C.POINTER(C.c_uint64)), dptr(xs))
    if rc != 0:
        raise OracleError(rc, L.oracle_last_error().decode())
    return lml.value, lnw, idx, xs


def unfold_simulate(kind, dim_state, dim_obs, params, n_steps, n, seed, args0=None, canonical=True):
    """oracle DynUnfold::simulate over n traces -> (states[n][T][d], obs[n][T][dobs])."""
    L = load()
    params = np.ascontiguousarray(params, dtype=np.float64)
    desc = ModelDesc(kind, dim_state, dim_obs, len(params), dptr(params))
    xs = np.empty((n, n_steps, dim_state))
    ys = np.empty((n, n_steps, dim_obs))
    a = None if args0 is None else dptr(np.ascontiguousarray(args0, dtype=np.float64))
    rc = L.oracle_unfold_simulate(C.byref(desc), a, n_steps, n, seed, int(canonical), dptr(xs), dptr(ys))
    if rc != 0:
        raise OracleError(rc, L.oracle_last_error().decode())
    return xs, ys


class OracleMH:
    """N independent chains of the restated hierarchical_model driven by the restated mh / regen_mh."""

    def __init__(self, xs, ys, n_chains, seed, constrain_is_linear=-1, canonical=True):
        self.L = load()
        self.n = n_chains
        xs = np.ascontiguousarray(xs, dtype=np.float64)
        ys = np.ascontiguousarray(ys, dtype=np.float64)
        h = C.c_void_p()
        self._ck(self.L.oracle_mh_create(dptr(xs), dptr(ys), len(xs), constrain_is_linear, n_chains, seed, int(canonical), C.byref(h)))
        self.h = h

    def _ck(self, code):
        if code != 0:
            raise OracleError(code, self.L.oracle_last_error().decode())

    def mh(self, drift_std, n_iters=1):
        acc = C.c_uint64()
        self._ck(self.L.oracle_mh_step(self.h, drift_std, n_iters, C.byref(acc)))
        return acc.value

    def mh_add_or_remove(self, n_iters=1):
        acc = C.c_uint64()
        self._ck(self.L.oracle_mh_step_add_or_remove(self.h, n_iters, C.byref(acc)))
        return acc.value

    def regen_mh(self, mask_sites, n_iters=1, cycle=False):
        m = (C.c_int32 * len(mask_sites))(*mask_sites)
        acc = C.c_uint64()
        self._ck(self.L.oracle_regen_mh_step(self.h, m, len(mask_sites), int(cycle), n_iters, C.byref(acc)))
        return acc.value

    def state(self):
        out = np.empty((self.n, 4))
        self._ck(self.L.oracle_mh_read_state(self.h, dptr(out)))
        return out

    def logjp(self):
        out = np.empty(self.n)
        self._ck(self.L.oracle_mh_read_logjp(self.h, dptr(out)))
        return out

    def observations(self, n_data):
        out = np.empty((self.n, n_data))
        self._ck(self.L.oracle_mh_read_observations(self.h, n_data, dptr(out)))
        return out

    def __del__(self):
        try:
            self.L.oracle_mh_destroy(self.h)
        except Exception:
            pass


class OracleShardEngine:
    """Local engine for modppl_amd.distributed.ShardedParticleSystem backed by the CPU checker:
    used by the CPU (gloo) tests of the multi-rank orchestration.  Never used by the product."""

    device = "cpu"

    def __init__(self, model, n_local, n_global, slot_offset, seed, **_):
        self.L = load()
        self.model = model
        self.n = n_local
        self._params = np.ascontiguousarray(model.params, dtype=np.float64)
        desc = ModelDesc(model.kind, model.dim_state, model.dim_obs, len(self._params), dptr(self._params))
        sh = Shard(n_global, slot_offset)
        h = C.c_void_p()
        self._ck(self.L.oracle_pf_create(C.byref(desc), n_local, seed, C.byref(sh), 0, VARIANT_CANONICAL | VARIANT_SOA, C.byref(h)))
        self.h = h

    def _ck(self, code):
        if code != 0:
            raise OracleError(code, self.L.oracle_last_error().decode())

    def init_step(self, args0, obs):
        a = None if args0 is None else dptr(np.ascontiguousarray(args0, dtype=np.float64))
        self._ck(self.L.oracle_pf_init_step(self.h, a, dptr(obs), obs.shape[0]))

    def step(self, obs):
        self._ck(self.L.oracle_pf_step(self.h, dptr(obs), obs.shape[0]))

    def shard_tiles(self, tm_ptr, tw_ptr, tw2_ptr):
        self._ck(self.L.oracle_pf_shard_tiles(self.h, tm_ptr, tw_ptr, tw2_ptr))

    def shard_route(self, scheme, tm_ptr, tw_ptr, tw2_ptr, nt_all, world, rank, req_ptr):
        counts = (C.c_int64 * world)()
        self._ck(self.L.oracle_pf_shard_route(self.h, scheme, tm_ptr, tw_ptr, tw2_ptr, nt_all, world, rank, req_ptr, counts))
        return list(counts)

    def shard_resolve(self, req_ptr, n_req, rows_ptr):
        self._ck(self.L.oracle_pf_shard_resolve(self.h, req_ptr, n_req, rows_ptr))

    def shard_scatter(self, rows_ptr, want_value):
        out = C.c_double()
        self._ck(self.L.oracle_pf_shard_scatter(self.h, rows_ptr, C.byref(out) if want_value else None))
        return out.value if want_value else None

    def shard_query(self, tm_ptr, tw_ptr, tw2_ptr, nt_all):
        lml, ess = C.c_double(), C.c_double()
        self._ck(self.L.oracle_pf_shard_query(self.h, tm_ptr, tw_ptr, tw2_ptr, nt_all, C.byref(lml), C.byref(ess)))
        return lml.value, ess.value

    # "owner keeps" form: same method names and arguments as HipShardEngine's (packed = exact-size buffers only here)
    supports_owned = True

    def shard_tiles_packed(self, tiles_ptr):
        nt = (self.n + 2047) // 2048
        b = tiles_ptr.value
        self._ck(self.L.oracle_pf_shard_tiles(self.h, C.c_void_p(b), C.c_void_p(b + 8 * nt), C.c_void_p(b + 16 * nt)))

    def shard_owned_count(self, scheme, tiles_all_ptr, world, rank, cap=0, want_counts=True):
        nt = (self.n + 2047) // 2048
        g = np.ctypeslib.as_array((C.c_int64 * (world * 3 * nt)).from_address(tiles_all_ptr.value)).reshape(world, 3, nt)
        tm = np.ascontiguousarray(g[:, 0, :]).reshape(-1)
        tw = np.ascontiguousarray(g[:, 1, :]).reshape(-1)
        tw2 = np.ascontiguousarray(g[:, 2, :]).reshape(-1)
        counts = (C.c_uint64 * world)()
        self._rank = rank
        self._ck(self.L.oracle_pf_shard_owned_count(self.h, scheme, C.c_void_p(tm.ctypes.data), C.c_void_p(tw.ctypes.data),
                                                    C.c_void_p(tw2.ctypes.data), world * nt, world, rank, counts))
        return list(counts)

    def shard_owned_expand(self, world, rank, cap, send_ptr, rows_ptr, recv_rows):
        assert cap == 0, "the checker only has the exact-size form"
        sent = C.c_uint64()
        self._ck(self.L.oracle_pf_shard_owned_expand(self.h, rank, send_ptr, C.byref(sent)))
        return sent.value

    def shard_owned_commit(self, rows_ptr, recv_rows, want_value, want_counts=True):
        out = C.c_double()
        self._ck(self.L.oracle_pf_shard_owned_adopt(self.h, self._rank, rows_ptr, recv_rows, C.byref(out)))
        return True, (out.value if want_value else None), None

    def ess_reference(self):
        out = C.c_double()
        self._ck(self.L.oracle_pf_effective_sample_size(self.h, 0, C.byref(out)))
        return out.value

    def states(self):
        x = np.empty((self.n, self.model.dim_state))
        self._ck(self.L.oracle_pf_read_state(self.h, dptr(x)))
        return x

    def log_weights(self):
        w = np.empty(self.n)
        self._ck(self.L.oracle_pf_read_log_weights(self.h, dptr(w)))
        return w

    def parents(self):
        p_ = np.empty(self.n, dtype=np.uint32)
        self._ck(self.L.oracle_pf_read_parents(self.h, p_.ctypes.data_as(C.POINTER(C.c_uint32))))
        return p_

    def synchronize(self):
        pass


class OraclePointedMH:
    """N chains of the restated pointed_2d_model under mh with pointed_2d_drift_proposal (tests/mh.rs:50-68)."""

    def __init__(self, bounds, obs_cov, obs, n_chains, seed, canonical=True):
        self.L = load()
        self.n = n_chains
        h = C.c_void_p()
        b, c, o = (np.ascontiguousarray(v, dtype=np.float64).reshape(-1) for v in (bounds, obs_cov, obs))
        self._ck(self.L.oracle_mh_pointed_create(dptr(b), dptr(c), dptr(o), n_chains, seed, int(canonical), C.byref(h)))
        self.h = h

    def _ck(self, code):
        if code != 0:
            raise OracleError(code, self.L.oracle_last_error().decode())

    def mh(self, noise, n_iters=1):
        nz = np.ascontiguousarray(noise, dtype=np.float64).reshape(-1)
        acc = C.c_uint64()
        self._ck(self.L.oracle_mh_pointed_step(self.h, dptr(nz), n_iters, C.byref(acc)))
        return acc.value

    def state(self):
        out = np.empty((self.n, 2))
        self._ck(self.L.oracle_mh_pointed_read_state(self.h, dptr(out)))
        return out

    def logjp(self):
        out = np.empty(self.n)
        self._ck(self.L.oracle_mh_pointed_read_logjp(self.h, dptr(out)))
        return out

    def __del__(self):
        try:
            self.L.oracle_mh_pointed_destroy(self.h)
        except Exception:
            pass


class OracleFunctionChains:
    """N chains of a REGISTERED MH functor model (modppl_amd/csrc/mp_mh_models.h) run by the checker's dynamic machinery — tries,
    sample_at / trace_at / gc, mh / regen_mh — through oracle/src/mh_functor_adapter.hpp: no hand-written restatement."""

    def __init__(self, kind, params, constraints, n_chains, seed, canonical=True):
        self.L = load()
        self.n = n_chains
        params = np.ascontiguousarray(params, dtype=np.float64).ravel()
        sites = np.array(sorted(constraints), dtype=np.int32)
        vals = np.array([constraints[int(k)] for k in sites], dtype=np.float64)
        h = C.c_void_p()
        self._ck(self.L.oracle_mhfn_create(int(kind), dptr(params), int(params.size), sites.ctypes.data_as(C.POINTER(C.c_int32)), dptr(vals), int(sites.size),
                                           C.c_uint64(n_chains), C.c_uint64(seed), int(canonical), C.byref(h)))
        self.h = h
        ns = C.c_int32()
        self._ck(self.L.oracle_mhfn_n_sites(self.h, C.byref(ns)))
        self.num_sites = ns.value

    def _ck(self, code):
        if code != 0:
            raise OracleError(code, self.L.oracle_last_error().decode())

    def mh(self, proposal_kind, proposal_args=(), n_iters=1):
        a = np.ascontiguousarray(proposal_args, dtype=np.float64).ravel()
        acc = C.c_uint64()
        self._ck(self.L.oracle_mhfn_step(self.h, int(proposal_kind), dptr(a), int(a.size), int(n_iters), C.byref(acc)))
        return acc.value

    def regen_mh(self, mask_sites, n_iters=1, cycle=False):
        m = (C.c_int32 * max(len(mask_sites), 1))(*mask_sites)
        acc = C.c_uint64()
        self._ck(self.L.oracle_mhfn_regen(self.h, m, len(mask_sites), int(cycle), int(n_iters), C.byref(acc)))
        return acc.value

    def trace(self):
        vals = np.empty((self.n, self.num_sites))
        present = np.empty(self.n, dtype=np.uint64)
        self._ck(self.L.oracle_mhfn_read_trace(self.h, dptr(vals), present.ctypes.data_as(C.POINTER(C.c_uint64))))
        return vals, present

    # ---- GenFn::update / regenerate / assess / propose one at a time (gfi.rs:57-90): the checker of mp_fn_* ----
    def _cons(self, constraints):
        ip, up = C.POINTER(C.c_int32), C.POINTER(C.c_uint64)
        if isinstance(constraints, dict):
            sites = np.array(sorted(constraints), dtype=np.int32)
            vals = np.array([constraints[int(k)] for k in sites], dtype=np.float64)
            return (sites, vals), (sites.ctypes.data_as(ip), dptr(vals), int(sites.size), None, None)
        cv = np.ascontiguousarray(constraints[0], dtype=np.float64).reshape(self.n, self.num_sites)
        cp = np.ascontiguousarray(constraints[1], dtype=np.uint64).reshape(self.n)   # (one 64-bit presence word per chain, whatever the product's table uses)
        return (cv, cp), (None, None, 0, dptr(cv), cp.ctypes.data_as(up))

    def update(self, constraints, argdiff=0, rng_step=0):
        keep, c = self._cons(constraints)
        w, dv, dp_ = np.empty(self.n), np.zeros((self.n, self.num_sites)), np.zeros(self.n, dtype=np.uint64)
        self._ck(self.L.oracle_mhfn_update(self.h, int(argdiff), C.c_uint32(rng_step), c[0], c[1], c[2], c[3], c[4], dptr(w), dptr(dv),
                                           dp_.ctypes.data_as(C.POINTER(C.c_uint64))))
        return w, (dv, dp_)

    def regenerate(self, mask_sites, argdiff=0, rng_step=0):
        m = (C.c_int32 * max(len(mask_sites), 1))(*mask_sites)
        w = np.empty(self.n)
        self._ck(self.L.oracle_mhfn_regenerate(self.h, int(argdiff), C.c_uint32(rng_step), m, len(mask_sites), dptr(w)))
        return w

    def assess(self, constraints, proposal_kind=-1, proposal_args=(), rng_step=0):
        keep, c = self._cons(constraints)
        a = np.ascontiguousarray(proposal_args, dtype=np.float64).ravel()
        w = np.empty(self.n)
        self._ck(self.L.oracle_mhfn_assess(self.h, int(proposal_kind), dptr(a), int(a.size), C.c_uint32(rng_step), c[0], c[1], c[2], c[3], c[4], dptr(w)))
        return w

    def propose(self, proposal_kind, proposal_args=(), rng_step=0):
        a = np.ascontiguousarray(proposal_args, dtype=np.float64).ravel()
        cv, cp, w = np.zeros((self.n, self.num_sites)), np.zeros(self.n, dtype=np.uint64), np.empty(self.n)
        self._ck(self.L.oracle_mhfn_propose(self.h, int(proposal_kind), dptr(a), int(a.size), C.c_uint32(rng_step), dptr(cv),
                                            cp.ctypes.data_as(C.POINTER(C.c_uint64)), dptr(w)))
        return (cv, cp), w

    def generate(self, constraints, rng_step=0):
        keep, c = self._cons(constraints)
        w = np.empty(self.n)
        self._ck(self.L.oracle_mhfn_generate(self.h, C.c_uint32(rng_step), c[0], c[1], c[2], c[3], c[4], dptr(w)))
        return w

    def simulate(self, rng_step=0):
        w = np.empty(self.n)
        self._ck(self.L.oracle_mhfn_simulate(self.h, C.c_uint32(rng_step), dptr(w)))
        return w

    @classmethod
    def importance(cls, kind, params, constraints, num_samples, num_ret, seed, canonical=True):
        """importance_resampling(model, args, constraints, N, M) (importance.rs:37-50) through the checker's generic functions
        -> (traces: OracleFunctionChains, log_normalized_weights, log_ml_estimate, resampled_indices)"""
        self = cls.__new__(cls)
        self.L = load()
        self.n = int(num_samples)
        params = np.ascontiguousarray(params, dtype=np.float64).ravel()
        sites = np.array(sorted(constraints), dtype=np.int32)
        vals = np.array([constraints[int(k)] for k in sites], dtype=np.float64)
        lnw, lml, idx = np.empty(self.n), C.c_double(), np.empty(max(int(num_ret), 1), dtype=np.uint64)
        h = C.c_void_p()
        self._ck(self.L.oracle_mhfn_importance(int(kind), dptr(params), int(params.size), sites.ctypes.data_as(C.POINTER(C.c_int32)), dptr(vals), int(sites.size),
                                               C.c_uint64(num_samples), C.c_uint64(num_ret), C.c_uint64(seed), int(canonical), C.byref(lml), dptr(lnw),
                                               idx.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(h)))
        self.h = h
        ns = C.c_int32()
        self._ck(self.L.oracle_mhfn_n_sites(self.h, C.byref(ns)))
        self.num_sites = ns.value
        return self, lnw, lml.value, idx[:int(num_ret)]

    def logjp(self):
        out = np.empty(self.n)
        self._ck(self.L.oracle_mhfn_read_logjp(self.h, dptr(out)))
        return out

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.oracle_mhfn_destroy(self.h)
                self.h = None
        except Exception:
            pass


class HostStaticFunctionChains:
    """The PRODUCT's static MH handlers (modppl_amd/csrc/mp_genfn.h) compiled for the host inside the test library and run chain
    by chain — a checkee, not a checker: what the k_fn_* kernels execute per lane, available without a GPU so that the CPU
    suite holds the handler rules against the dynamic machinery (OracleFunctionChains) too."""

    def __init__(self, kind, params, constraints, n_chains, seed):
        self.L = load()
        self.n = n_chains
        params = np.ascontiguousarray(params, dtype=np.float64).ravel()
        sites = np.array(sorted(constraints), dtype=np.int32)
        vals = np.array([constraints[int(k)] for k in sites], dtype=np.float64)
        h = C.c_void_p()
        self._ck(self.L.oracle_mhfn_static_create(int(kind), dptr(params), int(params.size), sites.ctypes.data_as(C.POINTER(C.c_int32)), dptr(vals),
                                                  int(sites.size), C.c_uint64(n_chains), C.c_uint64(seed), C.byref(h)))
        self.h = h
        self.num_sites = None

    def _ck(self, code):
        if code != 0:
            raise OracleError(code, self.L.oracle_last_error().decode())

    def mh(self, proposal_kind, proposal_args=(), n_iters=1):
        a = np.ascontiguousarray(proposal_args, dtype=np.float64).ravel()
        acc = C.c_uint64()
        self._ck(self.L.oracle_mhfn_static_step(self.h, int(proposal_kind), dptr(a), int(a.size), int(n_iters), C.byref(acc)))
        return acc.value

    def regen_mh(self, mask_sites, n_iters=1, cycle=False):
        m = (C.c_int32 * max(len(mask_sites), 1))(*mask_sites)
        acc = C.c_uint64()
        self._ck(self.L.oracle_mhfn_static_regen(self.h, m, len(mask_sites), int(cycle), int(n_iters), C.byref(acc)))
        return acc.value

    def update(self, constraints, argdiff=0, rng_step=1):
        """k_fn_update's per-lane work with constraints shared by all chains -> (weights, discard presence)"""
        sites = np.array(sorted(constraints), dtype=np.int32)
        vals = np.array([constraints[int(k)] for k in sites], dtype=np.float64)
        w, dp_ = np.empty(self.n), np.zeros(self.n, dtype=np.uint64)
        self._ck(self.L.oracle_mhfn_static_update(self.h, sites.ctypes.data_as(C.POINTER(C.c_int32)), dptr(vals), int(sites.size), int(argdiff),
                                                  C.c_uint32(rng_step), dptr(w), dp_.ctypes.data_as(C.POINTER(C.c_uint64))))
        return w, dp_

    def trace(self, num_sites):
        vals = np.empty((self.n, num_sites))
        present = np.empty(self.n, dtype=np.uint64)
        pan = C.c_uint64()
        self._ck(self.L.oracle_mhfn_static_read(self.h, dptr(vals), present.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(pan)))
        self.panics = pan.value
        return vals, present

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.oracle_mhfn_static_destroy(self.h)
                self.h = None
        except Exception:
            pass
