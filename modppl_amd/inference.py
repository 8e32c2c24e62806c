"""Host-side mirror of modppl's inference entry points for the MI355X path.

Same names, argument meaning and error behaviour as the reference (a reference `panic!` becomes a
`ModpplError`), calling the HIP kernels through the C ABI of include/modppl_hip.h:

    ParticleSystem.{new, init_step, step, effective_sample_size, resample,
                    log_marginal_likelihood_estimate}      modppl/src/inference/particle_filter.rs:44-121
"""
import ctypes as C

import numpy as np

from . import capi

_DP = C.POINTER(C.c_double)


_F64 = np.dtype(np.float64)


def _dptr(a):
    # (a float64 array's address for a `double*` argument: through the buffer protocol this costs 0.4 us, `a.ctypes.data_as` 2.1 us —
    # 4 % of a synchronous step; read-only or empty arrays take the slow way)
    try:
        return C.byref(C.c_double.from_buffer(a))
    except (TypeError, ValueError):
        return a.ctypes.data_as(_DP)


class ParticleSystem:
    """`ParticleSystem<Args,Data,Ret,F>` over an Unfold model (alias `DynParticles`, dynunfold.rs:20).

    `seed` replaces `rng: ThreadRng` (which cannot be seeded in the reference).
    """

    def __init__(self, model, num_particles, seed, *, device=0, stream=None, flags=0):
        self._L = capi.load()
        self.model = model
        self.num_particles = int(num_particles)
        self._desc = model.desc()
        h = C.c_void_p()
        capi.check(self._L.mp_pf_create(C.byref(self._desc), self.num_particles, int(seed), None, int(flags), int(device),
                                        C.c_void_p(stream) if stream else None, C.byref(h)))
        self._h = h
        # the per-step calls of a synchronous caller's loop (step; ESS; L = resample()) are 45 us of GPU time: bound functions, the
        # model's width and two reusable out-parameters instead of an attribute chain and a fresh ctypes object per call
        self._dim_obs = int(model.dim_obs)
        self._c_step, self._c_ess, self._c_resample = self._L.mp_pf_step, self._L.mp_pf_effective_sample_size, self._L.mp_pf_resample
        self._out_ess, self._out_L = C.c_double(), C.c_double()
        self._ref_ess, self._ref_L = C.byref(self._out_ess), C.byref(self._out_L)

    # ParticleSystem::new
    @classmethod
    def new(cls, model, num_particles, seed, **kw):
        return cls(model, num_particles, seed, **kw)

    def _obs(self, constraints):
        obs = np.ascontiguousarray(constraints, dtype=np.float64)
        if obs.size == 0 or obs.size % self.model.dim_obs:
            raise capi.ModpplError(capi.MP_ERR_CONSTRAINTS, "constraints must hold dim_obs values per time step")
        return obs.reshape(-1, self.model.dim_obs)

    def init_step(self, args, constraints):
        """init_step(args, constraints): N x generate((1, args), constraints) — particle_filter.rs:60-70."""
        obs = self._obs(constraints)
        a = None
        if args is not None:
            a = np.ascontiguousarray(args, dtype=np.float64).reshape(-1)
            if a.size != self.model.dim_state:
                raise capi.ModpplError(capi.MP_ERR_INVALID_ARG, "args must hold dim_state values")
        capi.check(self._L.mp_pf_init_step(self._h, _dptr(a) if a is not None else None, _dptr(obs), obs.shape[0]))

    def step(self, constraints):
        """step(constraints) -> Self: N x update(.., ArgDiff::Extend, constraints) — particle_filter.rs:73-96."""
        obs = constraints
        if type(obs) is np.ndarray and obs.dtype == _F64 and obs.flags.c_contiguous and obs.size and not obs.size % self._dim_obs:
            steps = obs.size // self._dim_obs    # (already what _obs would make of it)
        else:
            obs = self._obs(constraints)
            steps = obs.shape[0]
        code = self._c_step(self._h, _dptr(obs), steps)
        if code:
            capi.check(code)
        return self

    def effective_sample_size(self, fresh=False):
        """particle_filter.rs:98-100.  The reference reads the weights normalised by the LAST resample
        (1/N before any); fresh=True evaluates the current log-weights instead."""
        code = self._c_ess(self._h, capi.MP_ESS_FRESH if fresh else capi.MP_ESS_REFERENCE, self._ref_ess)
        if code:
            capi.check(code)
        return self._out_ess.value

    def resample(self, scheme=capi.MP_RESAMPLE_MULTINOMIAL, sync=True):
        """resample() -> log total weight — particle_filter.rs:103-116.  sync=False only enqueues."""
        if not sync:
            code = self._c_resample(self._h, scheme, None)
            if code:
                capi.check(code)
            return None
        code = self._c_resample(self._h, scheme, self._ref_L)
        if code:
            capi.check(code)
        return self._out_L.value

    def maybe_resample(self, ess_fraction=0.5, scheme=capi.MP_RESAMPLE_MULTINOMIAL):
        """ESS-triggered resampling (extension): resample iff ESS(current weights) < ess_fraction * N.
        -> (resampled, ess, log total weight or None)."""
        did, ess, ltw = C.c_int32(), C.c_double(), C.c_double()
        capi.check(self._L.mp_pf_resample_if_ess_below(self._h, scheme, float(ess_fraction), C.byref(did), C.byref(ess), C.byref(ltw)))
        return bool(did.value), ess.value, (ltw.value if did.value else None)

    def log_marginal_likelihood_estimate(self):
        """particle_filter.rs:119-121."""
        out = C.c_double()
        capi.check(self._L.mp_pf_log_marginal_likelihood_estimate(self._h, C.byref(out)))
        return out.value

    def run(self, args, constraints, scheme=capi.MP_RESAMPLE_MULTINOMIAL):
        """The loop of modppl/tests/smc.rs:64-90 (init_step; resample; {step; resample}*) enqueued in one call."""
        obs = self._obs(constraints)
        a = None if args is None else np.ascontiguousarray(args, dtype=np.float64).reshape(-1)
        capi.check(self._L.mp_pf_run(self._h, _dptr(a) if a is not None else None, _dptr(obs), obs.shape[0], scheme))
        return self

    def synchronize(self):
        capi.check(self._L.mp_pf_synchronize(self._h))

    # the pub `traces` field, flattened: traces[i].retv.last()
    def states(self, out=None):
        """-> [num_particles, dim_state].  `out`: a float64 C-contiguous array of that size to fill instead of a fresh one — a PINNED one
        (e.g. `torch.empty(..., pin_memory=True).numpy()`) takes the copy at the link's rate instead of through a staging buffer."""
        if out is None:
            x = np.empty((self.num_particles, self.model.dim_state))
        else:
            x = out
            if not (isinstance(x, np.ndarray) and x.dtype == _F64 and x.flags.c_contiguous and x.size == self.num_particles * self.model.dim_state):
                raise capi.ModpplError(capi.MP_ERR_INVALID_ARG, "states(out=): a C-contiguous float64 array of num_particles * dim_state values")
        capi.check(self._L.mp_pf_read_state(self._h, _dptr(x)))
        return x

    @property
    def log_weights(self):
        w = np.empty(self.num_particles)
        capi.check(self._L.mp_pf_read_log_weights(self._h, _dptr(w)))
        return w

    @property
    def parents(self):
        p = np.empty(self.num_particles, dtype=np.uint32)
        capi.check(self._L.mp_pf_read_parents(self._h, p.ctypes.data_as(C.POINTER(C.c_uint32))))
        return p

    @property
    def time(self):
        t = C.c_int64()
        capi.check(self._L.mp_pf_time(self._h, C.byref(t)))
        return t.value

    def trajectory(self, i):
        """traces[i].retv — the Vec<State> of particle i's lineage (needs flags=MP_PF_RECORD_HISTORY)."""
        out = np.empty((max(self.time, 1), self.model.dim_state))
        t = C.c_int32()
        capi.check(self._L.mp_pf_read_trajectory(self._h, int(i), _dptr(out), C.byref(t)))
        return out[: t.value]

    def trajectories(self, first=0, count=None):
        """traces[first .. first + count).retv at once: [count, t, dim_state] (one kernel walks all the lineages;
        modppl/tests/smc.rs:67 reads every particle's)."""
        count = self.num_particles - first if count is None else int(count)
        T = max(self.time, 1)
        out = np.empty((count, T, self.model.dim_state))
        t = C.c_int32()
        capi.check(self._L.mp_pf_read_trajectories(self._h, int(first), count, _dptr(out), C.byref(t)))
        return out[:, : t.value]

    def set_timing(self, enabled):
        capi.check(self._L.mp_pf_set_timing(self._h, int(enabled)))

    def last_propagate_form(self):
        """0 one workgroup per tile (k_propagate), 1 two tiles per workgroup (k_propagate_mt), 2 the dense d = 16 MFMA kernel; -1 before any step"""
        out = C.c_int32()
        capi.check(self._L.mp_pf_last_propagate_form(self._h, C.byref(out)))
        return out.value

    def region_begin(self):
        """ONE hipEvent pair around a region of launches (un-perturbed device time): region_begin(); ...; region_end()."""
        capi.check(self._L.mp_pf_region_begin(self._h))

    def region_end(self):
        """-> (elapsed ms on the handle's stream, k_propagate-family launches enqueued since region_begin); waits for the stream"""
        ms, n = C.c_double(), C.c_uint64()
        capi.check(self._L.mp_pf_region_end(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def get_timing(self, family):
        ms, n = C.c_double(), C.c_uint64()
        capi.check(self._L.mp_pf_get_timing(self._h, family, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def close(self):
        if getattr(self, "_h", None):
            self._L.mp_pf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def importance_sampling(model, model_args, constraints, num_samples, seed, *, device=0, full_traces=False):
    """`importance_sampling(model, model_args, constraints, num_samples)` — modppl/src/inference/importance.rs:12-28.

    Returns (traces, log_normalized_weights, log_ml_estimate); `traces` is the array of final states
    (`traces[i].retv.last()`) or, with full_traces=True, every sample's states at every step — `traces[i].retv`,
    [num_samples, n_steps, dim_state] — which is what the reference's `Vec<Trace>` holds."""
    if full_traces:
        L = capi.load()
        obs = np.ascontiguousarray(constraints, dtype=np.float64)
        if obs.size == 0 or obs.size % model.dim_obs:
            raise capi.ModpplError(capi.MP_ERR_CONSTRAINTS, "constraints must hold dim_obs values per time step")
        obs = obs.reshape(-1, model.dim_obs)
        a = None if model_args is None else np.ascontiguousarray(model_args, dtype=np.float64).reshape(-1)
        desc = model.desc()
        lml = C.c_double()
        lnw = np.empty(int(num_samples))
        traj = np.empty((int(num_samples), obs.shape[0], model.dim_state))
        capi.check(L.mp_importance_sampling(C.byref(desc), _dptr(a) if a is not None else None, _dptr(obs), obs.shape[0], int(num_samples), int(seed),
                                            int(device), C.byref(lml), _dptr(lnw), _dptr(traj)))
        return traj, lnw, lml.value
    states, lnw, lml, _ = _importance(model, model_args, constraints, num_samples, 0, seed, device)
    return states, lnw, lml


def importance_resampling(model, model_args, constraints, num_samples, num_ret_samples, seed, *, device=0):
    """`importance_resampling(model, model_args, constraints, num_samples, num_ret_samples)` — importance.rs:37-50.

    Returns (traces, resampled_indices, log_ml_estimate) like the reference (ALL traces plus M indices)."""
    states, _, lml, idx = _importance(model, model_args, constraints, num_samples, num_ret_samples, seed, device)
    return states, idx, lml


def _importance(model, model_args, constraints, num_samples, num_ret, seed, device):
    L = capi.load()
    obs = np.ascontiguousarray(constraints, dtype=np.float64)
    if obs.size == 0 or obs.size % model.dim_obs:
        raise capi.ModpplError(capi.MP_ERR_CONSTRAINTS, "constraints must hold dim_obs values per time step")
    obs = obs.reshape(-1, model.dim_obs)
    a = None if model_args is None else np.ascontiguousarray(model_args, dtype=np.float64).reshape(-1)
    desc = model.desc()
    lml = C.c_double()
    lnw = np.empty(int(num_samples))
    states = np.empty((int(num_samples), model.dim_state))
    idx = np.empty(int(num_ret), dtype=np.uint64)
    capi.check(L.mp_importance_resampling(C.byref(desc), _dptr(a) if a is not None else None, _dptr(obs), obs.shape[0], int(num_samples),
                                          int(num_ret), int(seed), int(device), C.byref(lml), _dptr(lnw),
                                          idx.ctypes.data_as(C.POINTER(C.c_uint64)) if num_ret else None, _dptr(states)))
    return states, lnw, lml.value, idx


def simulate(model, args, n_steps, num_traces, seed, *, device=0):
    """`model.simulate((n_steps, args))` for `num_traces` independent traces — DynUnfold::simulate
    (modppl/src/modeling/dynunfold.rs:22-39): every site sampled, the observation sites too.
    -> (states [num_traces, n_steps, dim_state], observations [num_traces, n_steps, dim_obs])."""
    L = capi.load()
    desc = model.desc()
    xs = np.empty((int(num_traces), int(n_steps), model.dim_state))
    ys = np.empty((int(num_traces), int(n_steps), model.dim_obs))
    a = None if args is None else np.ascontiguousarray(args, dtype=np.float64)
    capi.check(L.mp_unfold_simulate(C.byref(desc), _dptr(a) if a is not None else None, int(n_steps), int(num_traces), int(seed), int(device),
                                    _dptr(xs), _dptr(ys)))
    return xs, ys


class HierarchicalChains:
    """N independent MH chains over the reference's `hierarchical_model`
    (modppl/tests/dyngenfns/hierarchical.rs:33-47), advanced by the reference's MH entry points:

        mh(model, trace, proposal, proposal_args)      modppl/src/inference/mh.rs:9-51
        regen_mh(model, trace, mask)                   modppl/src/inference/mh.rs:54-75

    Creation runs `hierarchical_model.generate(xs, observations)` per chain (modppl/tests/mh.rs:91)."""

    A, B, C_ = capi.MP_SITE_A, capi.MP_SITE_B, capi.MP_SITE_C
    _ADDR = {"coeffs/a": capi.MP_SITE_A, "coeffs / a": capi.MP_SITE_A, "coeffs/b": capi.MP_SITE_B, "coeffs / b": capi.MP_SITE_B,
             "coeffs/c": capi.MP_SITE_C, "coeffs / c": capi.MP_SITE_C, "is_linear": capi.MP_SITE_IS_LINEAR}

    def __init__(self, xs, ys, num_chains, seed, *, constrain_is_linear=None, device=0, stream=None, functor=False):
        """functor=True: the same model as a REGISTERED generative function (csrc/mp_mh_models.h, kind 101) run by the
        generic Update / Regenerate handlers instead of the hand-written kernels — same calls, same results bit for bit.
        functor="data": the registered function whose "(y, j)" sites are DECLARED data sites (kind 105): any number of observations
        (the hand-written kernels and kind 101 stop at 16), same latents bit for bit; regen_mh with the empty mask and
        observations() do not apply (an observation is not part of a chain's state there)."""
        self._L = capi.load()
        xs = np.ascontiguousarray(xs, dtype=np.float64)
        ys = np.ascontiguousarray(ys, dtype=np.float64)
        if xs.shape != ys.shape or xs.ndim != 1:
            raise capi.ModpplError(capi.MP_ERR_INVALID_ARG, "xs and ys must be 1-D arrays of equal length")
        self.num_chains = int(num_chains)
        self._n_data = len(xs)
        self._fn = None
        if functor:
            cons = {capi.MP_SITE_Y0 + k: float(y) for k, y in enumerate(ys)}
            if constrain_is_linear is not None:
                cons[capi.MP_SITE_IS_LINEAR] = float(bool(constrain_is_linear))
            kind = capi.MP_MH_MODEL_HIERARCHICAL_DATA_FN if functor == "data" else capi.MP_MH_MODEL_HIERARCHICAL_FN
            self._fn = FunctionChains(kind, xs, cons, num_chains, seed, device=device, stream=stream)
            self._h = self._fn._h
            return
        c = -1 if constrain_is_linear is None else int(bool(constrain_is_linear))
        h = C.c_void_p()
        capi.check(self._L.mp_mh_create(capi.MP_MH_MODEL_HIERARCHICAL, _dptr(xs), _dptr(ys), len(xs), c, self.num_chains, int(seed), int(device),
                                        C.c_void_p(stream) if stream else None, C.byref(h)))
        self._h = h

    def mh(self, drift_std, n_iters=1):
        """n_iters x mh(&hierarchical_model, trace, &hierarchical_drift_proposal, drift_std); returns accepted moves."""
        a = np.array([drift_std], dtype=np.float64)
        acc = C.c_uint64()
        capi.check(self._L.mp_mh_step(self._h, capi.MP_MH_PROPOSAL_HIERARCHICAL_DRIFT, _dptr(a), 1, int(n_iters), C.byref(acc)))
        return acc.value

    def mh_add_or_remove(self, n_iters=1):
        """n_iters x mh(&hierarchical_model, trace, &add_or_remove_param_proposal, ()) (tests/mh.rs:94); returns accepted moves."""
        acc = C.c_uint64()
        capi.check(self._L.mp_mh_step(self._h, capi.MP_MH_PROPOSAL_HIERARCHICAL_ADD_OR_REMOVE, None, 0, int(n_iters), C.byref(acc)))
        return acc.value

    def regen_mh(self, mask, n_iters=1, cycle=False):
        """n_iters x regen_mh(&hierarchical_model, trace, &mask); mask = addresses ("coeffs/a", ...) or site ids."""
        sites = [self._ADDR[m] if isinstance(m, str) else int(m) for m in mask]
        m = (C.c_int32 * max(len(sites), 1))(*sites)
        acc = C.c_uint64()
        capi.check(self._L.mp_regen_mh_step(self._h, m if sites else None, len(sites), int(cycle), int(n_iters), C.byref(acc)))
        return acc.value

    def states(self):
        """[num_chains, 4] = is_linear, a, b, c  (read_coeffs of hierarchical.rs:5-16)."""
        if self._fn is not None:
            vals, present = self._fn.trace()
            out = vals[:, :4].copy()
            assert np.all(present & 0b111 == 0b111) and np.all(((present >> 3) & 1) == (vals[:, 0] == 0.0))   # c is there iff quadratic
            return out
        out = np.empty((self.num_chains, 4))
        capi.check(self._L.mp_mh_read_state(self._h, _dptr(out)))
        return out

    def logjp(self):
        out = np.empty(self.num_chains)
        capi.check(self._L.mp_mh_read_logjp(self._h, _dptr(out)))
        return out

    def observations(self):
        """[num_chains, n_data]: the "(y, i)" choices of every chain's trace (the data, until regen_mh with an empty mask
        — the trace's whole schema, dyngenfn.rs:571 — re-simulated them)."""
        if self._fn is not None:
            return self._fn.trace()[0][:, capi.MP_SITE_Y0:capi.MP_SITE_Y0 + self._n_data].copy()
        out = np.empty((self.num_chains, self._n_data))
        capi.check(self._L.mp_mh_read_observations(self._h, _dptr(out)))
        return out

    @property
    def iterations(self):
        it = C.c_uint64()
        capi.check(self._L.mp_mh_iterations(self._h, C.byref(it)))
        return it.value

    def close(self):
        if getattr(self, "_fn", None) is not None:
            self._fn.close()
            self._h = None
        if getattr(self, "_h", None):
            self._L.mp_mh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _fn_constraints(constraints):
    sites = np.array(sorted(constraints), dtype=np.int32)
    vals = np.array([constraints[int(k)] for k in sites], dtype=np.float64)
    return sites, vals


def fn_importance_sampling(model_kind, params, constraints, num_samples, seed, *, device=0, traces=True):
    """`importance_sampling(model, model_args, constraints, num_samples)` (modppl/src/inference/importance.rs:12-28) for a REGISTERED
    generative function (any model written for csrc/mp_genfn.h — the reference's call is generic over `impl GenFn`, and its third test
    runs it on `hierarchical_model`, tests/importance.rs:89-139).  -> (traces: FunctionChains whose chain i is sample i (or None),
    log_normalized_weights [num_samples], log_ml_estimate)."""
    return _fn_importance(model_kind, params, constraints, num_samples, 0, seed, device, traces)[:3]


def fn_importance_resampling(model_kind, params, constraints, num_samples, num_ret_samples, seed, *, device=0, traces=True):
    """`importance_resampling(...)` (importance.rs:37-50) for a registered generative function.
    -> (traces, resampled_indices [num_ret_samples], log_ml_estimate)   — the reference's return triple."""
    tr, lnw, lml, idx = _fn_importance(model_kind, params, constraints, num_samples, num_ret_samples, seed, device, traces)
    return tr, idx, lml


def _fn_importance(model_kind, params, constraints, num_samples, num_ret, seed, device, traces):
    L = capi.load()
    params = np.ascontiguousarray(params, dtype=np.float64).ravel()
    sites, vals = _fn_constraints(constraints)
    lnw = np.empty(int(num_samples))
    lml = C.c_double()
    idx = np.empty(max(int(num_ret), 1), dtype=np.uint64)
    h = C.c_void_p()
    common = (int(model_kind), _dptr(params) if params.size else None, int(params.size), sites.ctypes.data_as(C.POINTER(C.c_int32)) if sites.size else None,
              _dptr(vals) if sites.size else None, int(sites.size), int(num_samples))
    if num_ret:
        capi.check(L.mp_fn_importance_resampling(*common, int(num_ret), int(seed), int(device), C.byref(lml), _dptr(lnw),
                                                 idx.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(h) if traces else None))
    else:
        capi.check(L.mp_fn_importance_sampling(*common, int(seed), int(device), C.byref(lml), _dptr(lnw), C.byref(h) if traces else None))
    tr = FunctionChains(model_kind, params, {}, num_samples, seed, _handle=h) if traces else None
    return tr, lnw, lml.value, idx[:int(num_ret)]


class FunctionChains:
    """N independent MH chains over a REGISTERED generative function (csrc/mp_mh_models.h: a functor over the static
    handler of csrc/mp_genfn.h, `MP_REGISTER_MH_MODEL`), the counterpart of calling the reference's `mh` / `regen_mh`
    (modppl/src/inference/mh.rs:9-75) with a model and proposal of one's own.  Sites are integer ids.

        constraints = {site: value}: creation runs model.generate(params, constraints) per chain."""

    def __init__(self, model_kind, params, constraints, num_chains, seed, *, device=0, stream=None, simulate=False, _handle=None):
        """constraints = {site: value}: the chains start as `model.generate(params, constraints)` (gfi.rs:53-55; `initial_weights` keeps
        that call's weights); simulate=True: as `model.simulate(params)` instead (gfi.rs:51; `initial_weights` = each trace's logjp)."""
        self._L = capi.load()
        self.num_chains = int(num_chains)
        self.initial_weights = None
        if _handle is not None:   # (a handle the library returned: the traces of mp_fn_importance_*)
            self._h = _handle
        else:
            params = np.ascontiguousarray(params, dtype=np.float64).ravel()
            w = np.empty(self.num_chains)
            h = C.c_void_p()
            if simulate:
                capi.check(self._L.mp_fn_simulate_create(int(model_kind), _dptr(params) if params.size else None, int(params.size), self.num_chains, int(seed),
                                                         int(device), C.c_void_p(stream) if stream else None, _dptr(w), C.byref(h)))
            else:
                sites, vals = _fn_constraints(constraints)
                capi.check(self._L.mp_fn_generate_create(int(model_kind), _dptr(params) if params.size else None, int(params.size),
                                                         sites.ctypes.data_as(C.POINTER(C.c_int32)) if sites.size else None, _dptr(vals) if sites.size else None,
                                                         int(sites.size), self.num_chains, int(seed), int(device), C.c_void_p(stream) if stream else None,
                                                         _dptr(w), C.byref(h)))
            self._h = h
            self.initial_weights = w
        ns = C.c_int32()
        capi.check(self._L.mp_mh_n_sites(self._h, C.byref(ns)))
        self.num_sites = ns.value

    def generate(self, constraints, rng_step=0):
        """(trace, weight) = model.generate(args, constraints) on every chain (gfi.rs:53-55): the chains' traces are replaced; -> weights"""
        keep, c = self._constraints(constraints)
        w = np.empty(self.num_chains)
        capi.check(self._L.mp_fn_generate(self._h, int(rng_step), c[0], c[1], c[2], c[3], c[4], _dptr(w)))
        return w

    def simulate(self, rng_step=0):
        """trace = model.simulate(args) on every chain (gfi.rs:51): every site drawn; -> each new trace's logjp"""
        w = np.empty(self.num_chains)
        capi.check(self._L.mp_fn_simulate(self._h, int(rng_step), _dptr(w)))
        return w

    def mh(self, proposal_kind, proposal_args=(), n_iters=1):
        a = np.ascontiguousarray(proposal_args, dtype=np.float64).ravel()
        acc = C.c_uint64()
        capi.check(self._L.mp_mh_step(self._h, int(proposal_kind), _dptr(a) if a.size else None, int(a.size), int(n_iters), C.byref(acc)))
        return acc.value

    def regen_mh(self, mask, n_iters=1, cycle=False):
        sites = [int(m) for m in mask]
        m = (C.c_int32 * max(len(sites), 1))(*sites)
        acc = C.c_uint64()
        capi.check(self._L.mp_regen_mh_step(self._h, m if sites else None, len(sites), int(cycle), int(n_iters), C.byref(acc)))
        return acc.value

    # Presence words: the C ABI carries [num_chains][W] 32-bit words, W = (num_sites + 31) // 32.  Here: one integer per chain, bit k =
    # site k — uint32 for models of up to 32 sites, uint64 beyond.
    def _words(self):
        return (self.num_sites + 31) // 32

    def _present_buf(self):
        return np.empty((self.num_chains, self._words()), dtype=np.uint32)

    def _present_out(self, words):
        if words.shape[1] == 1:
            return words.reshape(self.num_chains)
        return words[:, 0].astype(np.uint64) | (words[:, 1].astype(np.uint64) << np.uint64(32))

    def _present_in(self, present):
        p = np.asarray(present).reshape(self.num_chains).astype(np.uint64)
        w = np.empty((self.num_chains, self._words()), dtype=np.uint32)
        w[:, 0] = (p & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        if w.shape[1] > 1:
            w[:, 1] = (p >> np.uint64(32)).astype(np.uint32)
        elif np.any(p >> np.uint64(32)):
            raise capi.ModpplError(capi.MP_ERR_INVALID_ARG, "present names a site the model does not have")
        return np.ascontiguousarray(w)

    def trace(self):
        """(values [num_chains, num_sites], present [num_chains]: bit k = site k is in the chain's trace)"""
        vals = np.empty((self.num_chains, self.num_sites))
        present = self._present_buf()
        capi.check(self._L.mp_mh_read_trace(self._h, _dptr(vals), present.ctypes.data_as(C.POINTER(C.c_uint32))))
        return vals, self._present_out(present)

    # ---- the GFI operations one at a time (modppl/src/gfi.rs:57-90), every chain per call ------------------------------------
    # constraints: {site: value} shared by all chains, or a (values [num_chains, num_sites], present [num_chains]) pair per chain
    # (what propose() and update()'s discard return).  rng_step = 0: the next MH iteration's Philox step, which the call consumes.
    def _constraints(self, constraints):
        none_i, none_d, none_u = None, None, None
        if isinstance(constraints, dict):
            sites = np.array(sorted(constraints), dtype=np.int32)
            vals = np.array([constraints[int(k)] for k in sites], dtype=np.float64)
            keep = (sites, vals)
            return keep, (sites.ctypes.data_as(C.POINTER(C.c_int32)) if sites.size else none_i, _dptr(vals) if sites.size else none_d, int(sites.size), none_d, none_u)
        cv, cp = constraints
        cv = np.ascontiguousarray(cv, dtype=np.float64).reshape(self.num_chains, self.num_sites)
        cp = self._present_in(cp)
        return (cv, cp), (none_i, none_d, 0, _dptr(cv), cp.ctypes.data_as(C.POINTER(C.c_uint32)))

    def update(self, constraints, argdiff=capi.MP_ARGDIFF_NOCHANGE, rng_step=0, want_discard=True):
        """(new_trace, discard, weight) = model.update(trace, args, argdiff, constraints) on every chain (gfi.rs:57-64); the chains'
        traces are replaced.  -> (weights [num_chains], (discard_values, discard_present) or None)"""
        keep, c = self._constraints(constraints)
        w = np.empty(self.num_chains)
        dv = np.empty((self.num_chains, self.num_sites)) if want_discard else None
        dp_ = self._present_buf() if want_discard else None
        capi.check(self._L.mp_fn_update(self._h, int(argdiff), int(rng_step), c[0], c[1], c[2], c[3], c[4], _dptr(w), _dptr(dv) if want_discard else None,
                                        dp_.ctypes.data_as(C.POINTER(C.c_uint32)) if want_discard else None))
        return w, ((dv, self._present_out(dp_)) if want_discard else None)

    def regenerate(self, mask, argdiff=capi.MP_ARGDIFF_NOCHANGE, rng_step=0):
        """(new_trace, weight) = model.regenerate(trace, args, argdiff, mask) on every chain (gfi.rs:66-73); -> weights"""
        sites = [int(m) for m in mask]
        m = (C.c_int32 * max(len(sites), 1))(*sites)
        w = np.empty(self.num_chains)
        capi.check(self._L.mp_fn_regenerate(self._h, int(argdiff), int(rng_step), m if sites else None, len(sites), _dptr(w)))
        return w

    def assess(self, constraints, proposal_kind=-1, proposal_args=(), rng_step=0):
        """weight = f.assess(args, constraints) (gfi.rs:85-90): f = the model (proposal_kind < 0) or a registered proposal applied to
        each chain's current trace (mh.rs:25-27); -> weights.  Traces are not modified."""
        keep, c = self._constraints(constraints)
        a = np.ascontiguousarray(proposal_args, dtype=np.float64).ravel()
        w = np.empty(self.num_chains)
        capi.check(self._L.mp_fn_assess(self._h, int(proposal_kind), _dptr(a) if a.size else None, int(a.size), int(rng_step), c[0], c[1], c[2], c[3], c[4], _dptr(w)))
        return w

    def propose(self, proposal_kind, proposal_args=(), rng_step=0):
        """(choices, weight) = proposal.propose((trace, args)) on every chain (gfi.rs:78-83, mh.rs:17-19);
        -> ((choice_values, choice_present), weights).  Traces are not modified."""
        a = np.ascontiguousarray(proposal_args, dtype=np.float64).ravel()
        cv = np.empty((self.num_chains, self.num_sites))
        cp = self._present_buf()
        w = np.empty(self.num_chains)
        capi.check(self._L.mp_fn_propose(self._h, int(proposal_kind), _dptr(a) if a.size else None, int(a.size), int(rng_step), _dptr(cv),
                                         cp.ctypes.data_as(C.POINTER(C.c_uint32)), _dptr(w)))
        return (cv, self._present_out(cp)), w

    def logjp(self):
        out = np.empty(self.num_chains)
        capi.check(self._L.mp_mh_read_logjp(self._h, _dptr(out)))
        return out

    @property
    def iterations(self):
        it = C.c_uint64()
        capi.check(self._L.mp_mh_iterations(self._h, C.byref(it)))
        return it.value

    def close(self):
        if getattr(self, "_h", None):
            self._L.mp_mh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PointedChains:
    """N independent MH chains over the reference's `pointed_2d_model` (modppl/tests/dyngenfns/simple.rs:27-34):
    latent ~ uniform_2d(bounds), obs ~ mvnormal(latent, obs_cov) observed — the model of `test_metropolis_hastings_dyngenfn`
    (tests/mh.rs:50-68).  `mh(noise)` = mh(&pointed_2d_model, trace, &pointed_2d_drift_proposal, noise)."""

    def __init__(self, bounds, obs_cov, obs, num_chains, seed, *, device=0, stream=None, functor=False):
        self._L = capi.load()
        b = np.ascontiguousarray(bounds, dtype=np.float64).reshape(4)
        c = np.ascontiguousarray(obs_cov, dtype=np.float64).reshape(4)
        o = np.ascontiguousarray(obs, dtype=np.float64).reshape(2)
        self.num_chains = int(num_chains)
        self._fn = None
        if functor:
            # the same model and proposal as a REGISTERED functor with vector-valued sites (csrc/mp_mh_models.h kind 120: latent = slots
            # 1, 2; obs = slots 3, 4), run by the generic handlers of csrc/mp_genfn.h: same results, bit for bit
            self._fn = FunctionChains(capi.MP_MH_MODEL_POINTED_FN, np.concatenate([b, c]), {3: o[0], 4: o[1]}, num_chains, seed, device=device, stream=stream)
            self._h = None
            return
        h = C.c_void_p()
        capi.check(self._L.mp_mh_create_pointed(_dptr(b), _dptr(c), _dptr(o), self.num_chains, int(seed), int(device),
                                                C.c_void_p(stream) if stream else None, C.byref(h)))
        self._h = h

    def mh(self, noise, n_iters=1):
        nz = np.ascontiguousarray(noise, dtype=np.float64).reshape(4)
        if self._fn is not None:
            return self._fn.mh(1, nz, n_iters)
        acc = C.c_uint64()
        capi.check(self._L.mp_mh_step(self._h, capi.MP_MH_PROPOSAL_POINTED_DRIFT, _dptr(nz), 4, int(n_iters), C.byref(acc)))
        return acc.value

    def states(self):
        """[num_chains, 2] = latent."""
        if self._fn is not None:
            return np.ascontiguousarray(self._fn.trace()[0][:, 1:3])
        out = np.empty((self.num_chains, 2))
        capi.check(self._L.mp_mh_read_state(self._h, _dptr(out)))
        return out

    def logjp(self):
        if self._fn is not None:
            return self._fn.logjp()
        out = np.empty(self.num_chains)
        capi.check(self._L.mp_mh_read_logjp(self._h, _dptr(out)))
        return out

    def close(self):
        if getattr(self, "_fn", None) is not None:
            self._fn.close()
            self._fn = None
        if getattr(self, "_h", None):
            self._L.mp_mh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
