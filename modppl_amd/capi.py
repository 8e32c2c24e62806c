"""ctypes binding of include/modppl_hip.h (the C ABI of the MI355X hot path).

There is no CPU fallback: if the library is missing, or no GPU is visible when a handle is
created, this raises.
"""
import ctypes as C
import os

from . import build as _build

MP_OK = 0
MP_ERR_INVALID_ARG, MP_ERR_STATE, MP_ERR_CONSTRAINTS, MP_ERR_DEGENERATE, MP_ERR_HIP, MP_ERR_UNSUPPORTED, MP_ERR_CAPACITY = 1, 2, 3, 4, 5, 6, 7

MP_MODEL_LGSSM1, MP_MODEL_SPIRAL, MP_MODEL_HMM, MP_MODEL_BEARINGS, MP_MODEL_LGSSM_BAND, MP_MODEL_POINTED_2D, MP_MODEL_LINE = 1, 2, 3, 4, 5, 6, 7
MP_MODEL_LGSSM_DENSE = 8
MP_RESAMPLE_MULTINOMIAL, MP_RESAMPLE_SYSTEMATIC, MP_RESAMPLE_STRATIFIED, MP_RESAMPLE_MULTINOMIAL_SPLIT = 0, 1, 2, 3
MP_ESS_REFERENCE, MP_ESS_FRESH = 0, 1
MP_PF_RECORD_HISTORY = 1
MP_K_PROPAGATE, MP_K_NORMALIZE_SCAN, MP_K_RESAMPLE_GATHER, MP_K_BIN_DRAWS = 0, 1, 2, 3
MP_K1_FORM_TILE, MP_K1_FORM_TWO_TILES, MP_K1_FORM_DENSE16 = 0, 1, 2   # mp_pf_last_propagate_form
MP_SITE_IS_LINEAR, MP_SITE_A, MP_SITE_B, MP_SITE_C, MP_SITE_Y0 = 0, 1, 2, 3, 4
MP_MH_MODEL_HIERARCHICAL = 1
MP_MH_MODEL_POINTED_2D = 2
MP_ARGDIFF_NOCHANGE, MP_ARGDIFF_UNKNOWN = 0, 1   # gfi.rs:94-111
MP_MH_MODEL_HIERARCHICAL_FN = 101
MP_MH_MODEL_HIERARCHICAL_DATA_FN = 105   # the same model with its observations declared as data sites (any number of them)
MP_MH_MODEL_POINTED_FN = 120        # pointed_2d_model as a registered functor with vector-valued sites   # the hierarchical model as a registered functor (mp_mh_create_fn)
MP_MH_PROPOSAL_HIERARCHICAL_DRIFT = 1
MP_MH_PROPOSAL_HIERARCHICAL_ADD_OR_REMOVE = 2
MP_MH_PROPOSAL_POINTED_DRIFT = 3

# every symbol include/modppl_hip.h declares (tests/test_capi_symbols.py checks the export table)
SYMBOLS = [
    "mp_last_error", "mp_device_count", "mp_pf_create", "mp_pf_init_step", "mp_pf_step", "mp_pf_effective_sample_size",
    "mp_pf_resample", "mp_pf_resample_if_ess_below", "mp_pf_log_marginal_likelihood_estimate", "mp_pf_read_state", "mp_pf_read_log_weights",
    "mp_pf_read_parents", "mp_pf_read_trajectory", "mp_pf_read_trajectories", "mp_pf_time", "mp_pf_run", "mp_pf_synchronize", "mp_pf_destroy",
    "mp_pf_set_timing", "mp_pf_get_timing", "mp_pf_region_begin", "mp_pf_region_end", "mp_pf_last_propagate_form", "mp_unfold_simulate", "mp_importance_resampling", "mp_importance_sampling",
    "mp_pf_shard_bind_tiles", "mp_pf_shard_tiles_packed", "mp_pf_shard_route_fixed", "mp_pf_shard_resolve_fixed", "mp_pf_shard_commit_fixed", "mp_pf_shard_query_packed",
    "mp_pf_shard_owned_count", "mp_pf_shard_owned_expand", "mp_pf_shard_owned_commit", "mp_pf_shard_owned_count_expand",
    "mp_pf_shard_resample", "mp_pf_shard_resample_rccl", "mp_pf_shard_query_native", "mp_pf_shard_resample_stats", "mp_transport_rccl",
    "mp_rccl_available", "mp_rccl_unique_id", "mp_rccl_comm_create", "mp_rccl_comm_destroy", "mp_pf_stream_copy",
    "mp_pf_shard_tiles", "mp_pf_shard_route", "mp_pf_shard_resolve", "mp_pf_shard_scatter", "mp_pf_shard_query",
    "mp_mh_create", "mp_mh_create_pointed", "mp_mh_step", "mp_regen_mh_step", "mp_mh_read_state", "mp_mh_read_logjp", "mp_mh_read_observations", "mp_mh_iterations", "mp_mh_destroy",
    "mp_mh_create_fn", "mp_mh_n_sites", "mp_mh_read_trace", "mp_fn_update", "mp_fn_regenerate", "mp_fn_assess", "mp_fn_propose",
    "mp_fn_generate", "mp_fn_simulate", "mp_fn_generate_create", "mp_fn_simulate_create", "mp_fn_importance_sampling", "mp_fn_importance_resampling",
    # include/modppl_hip_probe.h
    "mp_probe_math", "mp_probe_normal_sample", "mp_probe_u01", "mp_probe_mfma_f64", "mp_probe_mvnormal",
]


class ModelDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("dim_state", C.c_int32), ("dim_obs", C.c_int32), ("n_params", C.c_int32),
                ("params", C.POINTER(C.c_double))]


class Shard(C.Structure):
    _fields_ = [("n_global", C.c_uint64), ("slot_offset", C.c_uint64)]


class ModpplError(RuntimeError):
    """Stands for a reference `panic!` (the reference has no Result type anywhere)."""

    def __init__(self, code, msg):
        super().__init__(f"modppl_hip status {code}: {msg}")
        self.code = code


_lib = None


# mp_transport (include/modppl_hip.h): the two collectives of the sharded resample as C function pointers
ALL_GATHER_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)
ALL_TO_ALL_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64),
                            C.POINTER(C.c_uint64), C.c_int32, C.c_void_p)


class Transport(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("all_gather", ALL_GATHER_FN), ("all_to_all", ALL_TO_ALL_FN)]


def _preload_hip_runtime():
    """One HIP/HSA runtime per process.  PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64.so.1
    (same SONAMEs as /opt/rocm's); if this library pulled in /opt/rocm's copy first, a later `import torch` in the
    same process would find "No HIP GPUs".  So when torch is installed its bundled runtime is loaded first (without
    importing torch) and libmodppl_hip.so binds to it."""
    import importlib.util

    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    _lib = _load_so(os.environ.get("MODPPL_HIP_LIB") or _build.SO)   # the override is for the diagnostics builds (tools/, tests' subprocesses)
    return _lib


_diag = None


def load_diag():
    """The DIAGNOSTICS build of the library (-DMP_DIAGNOSTICS, csrc/mp_diag.h): the same sources with their A/B and test switches (MP_K1_MT,
    MP_DEFERRED_LOOKUPS, ...) enabled — for tests and tools; the product library reads no environment variable."""
    global _diag
    if _diag is None:
        if _build.is_stale(_build.SO_DIAG, ("-DMP_DIAGNOSTICS",)):
            try:
                _build.build_diag()
            except RuntimeError as e:
                raise ModpplError(MP_ERR_HIP, f"{_build.SO_DIAG} is missing or stale and could not be rebuilt: {e}")
        _diag = _load_so(_build.SO_DIAG)
    return _diag


def _load_so(so):
    if so == _build.SO and _build.is_stale():
        # never load a binary that does not correspond to the checked-out sources (content hash, not mtime): rebuild it
        # where hipcc exists, fail loudly where it does not — there is no CPU fallback either way
        try:
            _build.build()
        except RuntimeError as e:
            raise ModpplError(MP_ERR_HIP, f"{so} is missing or stale and could not be rebuilt: {e}")
    if not os.path.exists(so):
        raise ModpplError(MP_ERR_HIP, f"{so} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                      "(the gfx950 path has no CPU fallback)")
    _preload_hip_runtime()
    L = C.CDLL(so)
    d, i32, u32, i64, u64, p = C.c_double, C.c_int32, C.c_uint32, C.c_int64, C.c_uint64, C.c_void_p
    dp = C.POINTER(C.c_double)
    L.mp_last_error.restype = C.c_char_p
    L.mp_device_count.restype = i32
    L.mp_pf_create.argtypes = [C.POINTER(ModelDesc), u64, u64, C.POINTER(Shard), u32, i32, p, C.POINTER(p)]
    L.mp_pf_init_step.argtypes = [p, dp, dp, i32]
    L.mp_pf_step.argtypes = [p, dp, i32]
    L.mp_pf_effective_sample_size.argtypes = [p, i32, dp]
    L.mp_pf_resample.argtypes = [p, i32, dp]
    L.mp_pf_resample_if_ess_below.argtypes = [p, i32, C.c_double, C.POINTER(C.c_int32), dp, dp]
    L.mp_pf_log_marginal_likelihood_estimate.argtypes = [p, dp]
    L.mp_pf_read_state.argtypes = [p, dp]
    L.mp_pf_read_log_weights.argtypes = [p, dp]
    L.mp_pf_read_parents.argtypes = [p, C.POINTER(u32)]
    L.mp_pf_read_trajectory.argtypes = [p, u64, dp, C.POINTER(i32)]
    L.mp_pf_read_trajectories.argtypes = [p, u64, u64, dp, C.POINTER(i32)]
    L.mp_pf_time.argtypes = [p, C.POINTER(i64)]
    L.mp_pf_run.argtypes = [p, dp, dp, i32, i32]
    L.mp_pf_synchronize.argtypes = [p]
    L.mp_pf_destroy.argtypes = [p]
    L.mp_pf_set_timing.argtypes = [p, i32]
    L.mp_pf_get_timing.argtypes = [p, i32, dp, C.POINTER(u64)]
    L.mp_pf_last_propagate_form.argtypes = [p, C.POINTER(i32)]
    L.mp_pf_region_begin.argtypes = [p]
    L.mp_pf_region_end.argtypes = [p, dp, C.POINTER(u64)]
    L.mp_unfold_simulate.argtypes = [C.POINTER(ModelDesc), dp, i32, u64, u64, i32, dp, dp]
    L.mp_importance_resampling.argtypes = [C.POINTER(ModelDesc), dp, dp, i32, u64, u64, u64, i32, dp, dp, C.POINTER(u64), dp]
    L.mp_importance_sampling.argtypes = [C.POINTER(ModelDesc), dp, dp, i32, u64, u64, i32, dp, dp, dp]
    L.mp_pf_shard_bind_tiles.argtypes = [p, p]
    L.mp_pf_shard_tiles_packed.argtypes = [p, p]
    L.mp_pf_shard_route_fixed.argtypes = [p, i32, p, i32, i32, u64, p]
    L.mp_pf_shard_resolve_fixed.argtypes = [p, p, i32, u64, p]
    L.mp_pf_shard_commit_fixed.argtypes = [p, p, dp]
    L.mp_pf_shard_query_packed.argtypes = [p, p, i32, dp, dp]
    L.mp_pf_shard_owned_count.argtypes = [p, i32, p, i32, i32, u64, C.POINTER(u64)]
    L.mp_pf_shard_owned_expand.argtypes = [p, i32, i32, u64, p, p, u64]
    L.mp_pf_shard_owned_count_expand.argtypes = [p, i32, p, i32, i32, u64, p, p, u64]
    L.mp_pf_shard_owned_commit.argtypes = [p, p, dp, C.POINTER(u64)]
    L.mp_pf_shard_resample.argtypes = [p, C.POINTER(Transport), i32, i32, i32, i32, dp]
    L.mp_pf_shard_resample_rccl.argtypes = [p, p, i32, i32, i32, i32, dp]
    L.mp_pf_shard_query_native.argtypes = [p, C.POINTER(Transport), i32, i32, dp, dp]
    L.mp_pf_shard_resample_stats.argtypes = [p, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    L.mp_transport_rccl.argtypes = [p, C.POINTER(Transport)]
    L.mp_rccl_available.argtypes = []
    L.mp_rccl_unique_id.argtypes = [p]
    L.mp_rccl_comm_create.argtypes = [i32, i32, p, i32, C.POINTER(p)]
    L.mp_rccl_comm_destroy.argtypes = [p]
    L.mp_pf_stream_copy.argtypes = [p, p, p, u64, i32]
    L.mp_pf_shard_tiles.argtypes = [p, p, p, p]
    L.mp_pf_shard_route.argtypes = [p, i32, p, p, p, i32, i32, p, C.POINTER(i64)]
    L.mp_pf_shard_resolve.argtypes = [p, p, u64, p]
    L.mp_pf_shard_scatter.argtypes = [p, p, dp]
    L.mp_pf_shard_query.argtypes = [p, p, p, p, i32, dp, dp]
    L.mp_mh_create.argtypes = [i32, dp, dp, i32, i32, u64, u64, i32, p, C.POINTER(p)]
    L.mp_mh_create_pointed.argtypes = [dp, dp, dp, u64, u64, i32, p, C.POINTER(p)]
    L.mp_mh_step.argtypes = [p, i32, dp, i32, i32, C.POINTER(u64)]
    L.mp_regen_mh_step.argtypes = [p, C.POINTER(i32), i32, i32, i32, C.POINTER(u64)]
    L.mp_mh_read_state.argtypes = [p, dp]
    L.mp_mh_read_logjp.argtypes = [p, dp]
    L.mp_mh_read_observations.argtypes = [p, dp]
    L.mp_mh_iterations.argtypes = [p, C.POINTER(u64)]
    L.mp_mh_destroy.argtypes = [p]
    L.mp_mh_create_fn.argtypes = [i32, dp, i32, C.POINTER(i32), dp, i32, u64, u64, i32, p, C.POINTER(p)]
    L.mp_mh_n_sites.argtypes = [p, C.POINTER(i32)]
    L.mp_mh_read_trace.argtypes = [p, dp, C.POINTER(u32)]
    L.mp_fn_update.argtypes = [p, i32, u32, C.POINTER(i32), dp, i32, dp, C.POINTER(u32), dp, dp, C.POINTER(u32)]
    L.mp_fn_regenerate.argtypes = [p, i32, u32, C.POINTER(i32), i32, dp]
    L.mp_fn_assess.argtypes = [p, i32, dp, i32, u32, C.POINTER(i32), dp, i32, dp, C.POINTER(u32), dp]
    L.mp_fn_propose.argtypes = [p, i32, dp, i32, u32, dp, C.POINTER(u32), dp]
    L.mp_fn_generate.argtypes = [p, u32, C.POINTER(i32), dp, i32, dp, C.POINTER(u32), dp]
    L.mp_fn_simulate.argtypes = [p, u32, dp]
    L.mp_fn_generate_create.argtypes = [i32, dp, i32, C.POINTER(i32), dp, i32, u64, u64, i32, p, dp, C.POINTER(p)]
    L.mp_fn_simulate_create.argtypes = [i32, dp, i32, u64, u64, i32, p, dp, C.POINTER(p)]
    L.mp_fn_importance_sampling.argtypes = [i32, dp, i32, C.POINTER(i32), dp, i32, u64, u64, i32, dp, dp, C.POINTER(p)]
    L.mp_fn_importance_resampling.argtypes = [i32, dp, i32, C.POINTER(i32), dp, i32, u64, u64, u64, i32, dp, dp, C.POINTER(u64), C.POINTER(p)]
    L.mp_probe_math.argtypes = [i32, dp, dp, dp, i64, dp, i32]
    L.mp_probe_normal_sample.argtypes = [u64, u32, u32, u32, u32, d, d, i64, dp, i32]
    L.mp_probe_u01.argtypes = [u64, u32, u32, u32, u32, u32, i64, dp, i32]
    L.mp_probe_mfma_f64.argtypes = [dp, dp, dp, dp, i32]
    L.mp_probe_mvnormal.argtypes = [i32, i32, dp, dp, dp, i64, dp, u64, u32, u32, u32, u32, dp, i32]
    return L


def check(code):
    if code != MP_OK:
        raise ModpplError(code, load().mp_last_error().decode())
