//! Drop-in for modppl's `ParticleSystem` over `DynUnfold` models on an MI355X.  SOURCE ONLY (no rustc in the build image).
//!
//! Same method names and ownership as `modppl::inference::ParticleSystem` (modppl/src/inference/particle_filter.rs):
//! `step(self) -> Self` consumes the filter, `resample(&mut self) -> f64`; a non-zero status from the library is
//! re-raised as a panic, the reference's only error convention.
use modppl_hip_sys as sys;
use std::ffi::{c_void, CStr};
use std::ptr;

fn check(rc: i32) {
    if rc != sys::MP_OK {
        let msg = unsafe { CStr::from_ptr(sys::mp_last_error()) }.to_string_lossy().into_owned();
        panic!("modppl-hip status {}: {}", rc, msg);
    }
}

/// Counterpart of `DynUnfold<State>`: selects a kernel compiled into libmodppl_hip.so.
#[derive(Clone)]
pub struct UnfoldModel { pub kind: i32, pub dim_state: i32, pub dim_obs: i32, pub params: Vec<f64> }

impl UnfoldModel {
    pub fn lgssm(mu0: f64, sig0: f64, a: f64, sig_x: f64, sig_y: f64) -> Self {
        UnfoldModel { kind: sys::MP_MODEL_LGSSM1, dim_state: 1, dim_obs: 1, params: vec![mu0, sig0, a, sig_x, sig_y] }
    }
    /// `spiral_model` of modppl/tests/dyngenfns/unfold.rs
    pub fn spiral() -> Self { UnfoldModel { kind: sys::MP_MODEL_SPIRAL, dim_state: 2, dim_obs: 2, params: vec![] } }
}

pub struct ParticleSystem { h: *mut sys::mp_pf, model: UnfoldModel, num_particles: usize }

impl ParticleSystem {
    /// `ParticleSystem::new(model, num_particles, rng)`; `seed` replaces the unseedable `ThreadRng`.
    pub fn new(model: UnfoldModel, num_particles: usize, seed: u64) -> Self {
        let desc = sys::mp_model_desc { kind: model.kind, dim_state: model.dim_state, dim_obs: model.dim_obs,
                                        n_params: model.params.len() as i32, params: model.params.as_ptr() };
        let mut h = ptr::null_mut();
        check(unsafe { sys::mp_pf_create(&desc, num_particles as u64, seed, ptr::null(), 0, 0, ptr::null_mut(), &mut h) });
        ParticleSystem { h, model, num_particles }
    }
    /// `init_step(args, constraints)`: constraints = one `dim_obs` vector per time step.
    pub fn init_step(&mut self, args: &[f64], constraints: &[f64]) {
        let a = if args.is_empty() { ptr::null() } else { args.as_ptr() };
        check(unsafe { sys::mp_pf_init_step(self.h, a, constraints.as_ptr(), (constraints.len() / self.model.dim_obs as usize) as i32) });
    }
    /// `step(self, constraints) -> Self`
    pub fn step(self, constraints: &[f64]) -> Self {
        check(unsafe { sys::mp_pf_step(self.h, constraints.as_ptr(), (constraints.len() / self.model.dim_obs as usize) as i32) });
        self
    }
    pub fn effective_sample_size(&self) -> f64 {
        let mut v = 0.0;
        check(unsafe { sys::mp_pf_effective_sample_size(self.h, sys::MP_ESS_REFERENCE, &mut v) });
        v
    }
    pub fn resample(&mut self) -> f64 {
        let mut v = 0.0;
        check(unsafe { sys::mp_pf_resample(self.h, sys::MP_RESAMPLE_MULTINOMIAL, &mut v) });
        v
    }
    pub fn log_marginal_likelihood_estimate(&self) -> f64 {
        let mut v = 0.0;
        check(unsafe { sys::mp_pf_log_marginal_likelihood_estimate(self.h, &mut v) });
        v
    }
    /// `traces[i].retv.last()` for every particle, row-major `[num_particles][dim_state]`.
    pub fn states(&self) -> Vec<f64> {
        let mut x = vec![0.0; self.num_particles * self.model.dim_state as usize];
        check(unsafe { sys::mp_pf_read_state(self.h, x.as_mut_ptr()) });
        x
    }
}

impl ParticleSystem {
    /// Extension: systematic / stratified resampling (`sys::MP_RESAMPLE_*`); the reference only has multinomial.
    pub fn resample_with(&mut self, scheme: i32) -> f64 {
        let mut v = 0.0;
        check(unsafe { sys::mp_pf_resample(self.h, scheme, &mut v) });
        v
    }
    /// Extension: resample iff the ESS of the current weights is below `ess_fraction * N`.  -> (resampled, ess)
    pub fn maybe_resample(&mut self, ess_fraction: f64, scheme: i32) -> (bool, f64) {
        let (mut did, mut ess, mut ltw) = (0i32, 0.0f64, 0.0f64);
        check(unsafe { sys::mp_pf_resample_if_ess_below(self.h, scheme, ess_fraction, &mut did, &mut ess, &mut ltw) });
        (did != 0, ess)
    }
    /// `log_weights` (pub field of the reference's struct)
    pub fn log_weights(&self) -> Vec<f64> {
        let mut w = vec![0.0; self.num_particles];
        check(unsafe { sys::mp_pf_read_log_weights(self.h, w.as_mut_ptr()) });
        w
    }
    /// `parents` (pub field): the indices of the last resample
    pub fn parents(&self) -> Vec<u32> {
        let mut p = vec![0u32; self.num_particles];
        check(unsafe { sys::mp_pf_read_parents(self.h, p.as_mut_ptr()) });
        p
    }
}

impl Drop for ParticleSystem {
    fn drop(&mut self) { unsafe { sys::mp_pf_destroy(self.h); } }
}

/// `importance_resampling(model, args, constraints, num_samples)` (modppl/src/inference/importance.rs:37-50) with
/// `num_ret_samples` draws: -> (log_ml_estimate, log_normalized_weights, resampled indices, the resampled final states).
pub fn importance_resampling(model: &UnfoldModel, args: &[f64], constraints: &[f64], num_samples: usize, num_ret_samples: usize,
                             seed: u64) -> (f64, Vec<f64>, Vec<u64>, Vec<f64>) {
    let desc = sys::mp_model_desc { kind: model.kind, dim_state: model.dim_state, dim_obs: model.dim_obs,
                                    n_params: model.params.len() as i32, params: model.params.as_ptr() };
    let a = if args.is_empty() { ptr::null() } else { args.as_ptr() };
    let n_steps = (constraints.len() / model.dim_obs as usize) as i32;
    let mut lml = 0.0;
    let mut lnw = vec![0.0; num_samples];
    let mut idx = vec![0u64; num_ret_samples];
    let mut xs = vec![0.0; num_ret_samples * model.dim_state as usize];
    check(unsafe {
        sys::mp_importance_resampling(&desc, a, constraints.as_ptr(), n_steps, num_samples as u64, num_ret_samples as u64, seed, 0,
                                      &mut lml, lnw.as_mut_ptr(), idx.as_mut_ptr(), xs.as_mut_ptr())
    });
    (lml, lnw, idx, xs)
}

/// `DynUnfold::simulate` for `n` independent traces of `n_steps` kernel calls (modppl/src/modeling/dynunfold.rs:22-39):
/// -> (states `[n][n_steps][dim_state]`, observations `[n][n_steps][dim_obs]`).
pub fn simulate(model: &UnfoldModel, args: &[f64], n_steps: usize, n: usize, seed: u64) -> (Vec<f64>, Vec<f64>) {
    let desc = sys::mp_model_desc { kind: model.kind, dim_state: model.dim_state, dim_obs: model.dim_obs,
                                    n_params: model.params.len() as i32, params: model.params.as_ptr() };
    let a = if args.is_empty() { ptr::null() } else { args.as_ptr() };
    let mut xs = vec![0.0; n * n_steps * model.dim_state as usize];
    let mut ys = vec![0.0; n * n_steps * model.dim_obs as usize];
    check(unsafe { sys::mp_unfold_simulate(&desc, a, n_steps as i32, n as u64, seed, 0, xs.as_mut_ptr(), ys.as_mut_ptr()) });
    (xs, ys)
}

/// `n_chains` independent chains over the reference's `hierarchical_model` (modppl/tests/dyngenfns/hierarchical.rs:18-47):
/// `mh` / `regen_mh` (modppl/src/inference/mh.rs:9-75) advance every chain by `n_iters` iterations per call.
pub struct HierarchicalChains { h: *mut sys::mp_mh, n_chains: usize }

impl HierarchicalChains {
    /// `constrain_is_linear`: -1 leaves `is_linear` free, 0 / 1 constrain it (tests/mh.rs:81-89 constrains the data only)
    pub fn new(xs: &[f64], ys: &[f64], constrain_is_linear: i32, n_chains: usize, seed: u64) -> Self {
        assert_eq!(xs.len(), ys.len());
        let mut h = ptr::null_mut();
        check(unsafe {
            sys::mp_mh_create(sys::MP_MH_MODEL_HIERARCHICAL, xs.as_ptr(), ys.as_ptr(), xs.len() as i32, constrain_is_linear,
                              n_chains as u64, seed, 0, ptr::null_mut(), &mut h)
        });
        HierarchicalChains { h, n_chains }
    }
    /// `mh(model, trace, hierarchical_drift_proposal, (drift_std,))` x n_iters per chain -> accepted moves
    pub fn mh(&mut self, drift_std: f64, n_iters: i32) -> u64 {
        let mut acc = 0u64;
        check(unsafe { sys::mp_mh_step(self.h, sys::MP_MH_PROPOSAL_HIERARCHICAL_DRIFT, &drift_std, 1, n_iters, &mut acc) });
        acc
    }
    /// `mh(model, trace, add_or_remove_param_proposal, ())` x n_iters per chain (the structure-changing move of tests/mh.rs:94)
    pub fn mh_add_or_remove(&mut self, n_iters: i32) -> u64 {
        let mut acc = 0u64;
        check(unsafe { sys::mp_mh_step(self.h, sys::MP_MH_PROPOSAL_HIERARCHICAL_ADD_OR_REMOVE, ptr::null(), 0, n_iters, &mut acc) });
        acc
    }
    /// `regen_mh(model, trace, mask)` x n_iters per chain; `mask` = `sys::MP_SITE_*`; `cycle`: one site of the mask per
    /// iteration, in turn, instead of all of them at once
    pub fn regen_mh(&mut self, mask: &[i32], cycle: bool, n_iters: i32) -> u64 {
        let mut acc = 0u64;
        check(unsafe { sys::mp_regen_mh_step(self.h, mask.as_ptr(), mask.len() as i32, cycle as i32, n_iters, &mut acc) });
        acc
    }
    /// per chain `[is_linear, a, b, c]`
    pub fn states(&self) -> Vec<f64> {
        let mut x = vec![0.0; self.n_chains * 4];
        check(unsafe { sys::mp_mh_read_state(self.h, x.as_mut_ptr()) });
        x
    }
    /// `trace.logjp` per chain
    pub fn logjp(&self) -> Vec<f64> {
        let mut x = vec![0.0; self.n_chains];
        check(unsafe { sys::mp_mh_read_logjp(self.h, x.as_mut_ptr()) });
        x
    }
}

impl Drop for HierarchicalChains {
    fn drop(&mut self) { unsafe { sys::mp_mh_destroy(self.h); } }
}

/// N chains of a REGISTERED generative function (csrc/mp_mh_models.h: `MP_REGISTER_MH_MODEL` / `MP_REGISTER_MH_PROPOSAL`) —
/// `importance_sampling(model, model_args, constraints, num_samples)` (importance.rs:12-28) for a registered generative function:
/// -> (traces: chain i = sample i, log_normalized_weights, log_ml_estimate)
pub fn fn_importance_sampling(model_kind: i32, params: &[f64], constraints: &[(i32, f64)], num_samples: usize, seed: u64) -> (FunctionChains, Vec<f64>, f64) {
    let sites: Vec<i32> = constraints.iter().map(|c| c.0).collect();
    let vals: Vec<f64> = constraints.iter().map(|c| c.1).collect();
    let (mut h, mut lml, mut lnw) = (ptr::null_mut(), 0.0f64, vec![0.0; num_samples]);
    check(unsafe {
        sys::mp_fn_importance_sampling(model_kind, params.as_ptr(), params.len() as i32, sites.as_ptr(), vals.as_ptr(), sites.len() as i32, num_samples as u64, seed, 0,
                                       &mut lml, lnw.as_mut_ptr(), &mut h)
    });
    (FunctionChains::adopt(h, num_samples), lnw, lml)
}
/// `importance_resampling(model, model_args, constraints, num_samples, num_ret_samples)` (importance.rs:37-50)
/// -> (traces, resampled_indices, log_ml_estimate)
pub fn fn_importance_resampling(model_kind: i32, params: &[f64], constraints: &[(i32, f64)], num_samples: usize, num_ret_samples: usize,
                                seed: u64) -> (FunctionChains, Vec<usize>, f64) {
    let sites: Vec<i32> = constraints.iter().map(|c| c.0).collect();
    let vals: Vec<f64> = constraints.iter().map(|c| c.1).collect();
    let (mut h, mut lml, mut idx) = (ptr::null_mut(), 0.0f64, vec![0u64; num_ret_samples]);
    check(unsafe {
        sys::mp_fn_importance_resampling(model_kind, params.as_ptr(), params.len() as i32, sites.as_ptr(), vals.as_ptr(), sites.len() as i32, num_samples as u64,
                                         num_ret_samples as u64, seed, 0, &mut lml, ptr::null_mut(), idx.as_mut_ptr(), &mut h)
    });
    (FunctionChains::adopt(h, num_samples), idx.into_iter().map(|i| i as usize).collect(), lml)
}

/// what `mh(&model, trace, &proposal, args)` / `regen_mh(&model, trace, &mask)` (mh.rs:9-75) become for a model and proposal
/// of one's own: sites are integer ids, `constraints` the `(site, value)` pairs of the initial `model.generate`.
pub struct FunctionChains { h: *mut sys::mp_mh, n_chains: usize, n_sites: usize }

impl FunctionChains {
    pub fn new(model_kind: i32, params: &[f64], constraints: &[(i32, f64)], n_chains: usize, seed: u64) -> Self {
        let sites: Vec<i32> = constraints.iter().map(|c| c.0).collect();
        let vals: Vec<f64> = constraints.iter().map(|c| c.1).collect();
        let mut h = ptr::null_mut();
        check(unsafe {
            sys::mp_mh_create_fn(model_kind, params.as_ptr(), params.len() as i32, sites.as_ptr(), vals.as_ptr(), sites.len() as i32,
                                 n_chains as u64, seed, 0, ptr::null_mut(), &mut h)
        });
        let mut ns = 0i32;
        check(unsafe { sys::mp_mh_n_sites(h, &mut ns) });
        FunctionChains { h, n_chains, n_sites: ns as usize }
    }
    /// `model.simulate(args)` per chain (gfi.rs:51) -> (chains, each trace's logjp)
    pub fn simulate_new(model_kind: i32, params: &[f64], n_chains: usize, seed: u64) -> (Self, Vec<f64>) {
        let mut h = ptr::null_mut();
        let mut lj = vec![0.0; n_chains];
        check(unsafe { sys::mp_fn_simulate_create(model_kind, params.as_ptr(), params.len() as i32, n_chains as u64, seed, 0, ptr::null_mut(), lj.as_mut_ptr(), &mut h) });
        (Self::adopt(h, n_chains), lj)
    }
    /// `model.generate(args, constraints)` per chain (gfi.rs:53-55) -> (chains, weights)
    pub fn generate_new(model_kind: i32, params: &[f64], constraints: &[(i32, f64)], n_chains: usize, seed: u64) -> (Self, Vec<f64>) {
        let sites: Vec<i32> = constraints.iter().map(|c| c.0).collect();
        let vals: Vec<f64> = constraints.iter().map(|c| c.1).collect();
        let mut h = ptr::null_mut();
        let mut w = vec![0.0; n_chains];
        check(unsafe {
            sys::mp_fn_generate_create(model_kind, params.as_ptr(), params.len() as i32, sites.as_ptr(), vals.as_ptr(), sites.len() as i32, n_chains as u64, seed, 0,
                                       ptr::null_mut(), w.as_mut_ptr(), &mut h)
        });
        (Self::adopt(h, n_chains), w)
    }
    fn adopt(h: *mut sys::mp_mh, n_chains: usize) -> Self {
        let mut ns = 0i32;
        check(unsafe { sys::mp_mh_n_sites(h, &mut ns) });
        FunctionChains { h, n_chains, n_sites: ns as usize }
    }
    /// `(trace, weight) = model.generate(args, constraints)` on every chain, shared constraints; the traces are replaced -> weights
    pub fn generate(&mut self, constraints: &[(i32, f64)], rng_step: u32) -> Vec<f64> {
        let sites: Vec<i32> = constraints.iter().map(|c| c.0).collect();
        let vals: Vec<f64> = constraints.iter().map(|c| c.1).collect();
        let mut w = vec![0.0; self.n_chains];
        check(unsafe { sys::mp_fn_generate(self.h, rng_step, sites.as_ptr(), vals.as_ptr(), sites.len() as i32, ptr::null(), ptr::null(), w.as_mut_ptr()) });
        w
    }
    /// `trace = model.simulate(args)` on every chain; the traces are replaced -> each new trace's logjp
    pub fn simulate(&mut self, rng_step: u32) -> Vec<f64> {
        let mut lj = vec![0.0; self.n_chains];
        check(unsafe { sys::mp_fn_simulate(self.h, rng_step, lj.as_mut_ptr()) });
        lj
    }
    /// `mh(model, trace, proposal, args)` x n_iters per chain -> accepted moves
    pub fn mh(&mut self, proposal_kind: i32, args: &[f64], n_iters: i32) -> u64 {
        let mut acc = 0u64;
        check(unsafe { sys::mp_mh_step(self.h, proposal_kind, args.as_ptr(), args.len() as i32, n_iters, &mut acc) });
        acc
    }
    /// `regen_mh(model, trace, mask)` x n_iters per chain; an empty mask is the trace's whole schema (dyngenfn.rs:571)
    pub fn regen_mh(&mut self, mask: &[i32], cycle: bool, n_iters: i32) -> u64 {
        let mut acc = 0u64;
        check(unsafe { sys::mp_regen_mh_step(self.h, mask.as_ptr(), mask.len() as i32, cycle as i32, n_iters, &mut acc) });
        acc
    }
    /// 32-bit presence words per chain: one up to 32 sites, two beyond (`present[chain * words() + w]`)
    pub fn words(&self) -> usize { (self.n_sites + 31) / 32 }
    /// `(values[chain][site], present[chain][words()])`: bit k of a chain's words = site k is in the chain's trace
    pub fn trace(&self) -> (Vec<f64>, Vec<u32>) {
        let mut v = vec![0.0; self.n_chains * self.n_sites];
        let mut p = vec![0u32; self.n_chains * self.words()];
        check(unsafe { sys::mp_mh_read_trace(self.h, v.as_mut_ptr(), p.as_mut_ptr()) });
        (v, p)
    }
    // ---- `GenFn::update / regenerate / assess / propose` (gfi.rs:57-90) one at a time, every chain per call ----
    /// `(new_trace, discard, weight) = model.update(trace, args, diff, constraints)` with per-chain constraints
    /// `(values[chain][site], present[chain])`; the chains' traces are replaced.  -> (weights, discard)
    pub fn update(&mut self, constraints: &(Vec<f64>, Vec<u32>), diff: &ArgDiff, rng_step: u32) -> (Vec<f64>, (Vec<f64>, Vec<u32>)) {
        let mut w = vec![0.0; self.n_chains];
        let mut dv = vec![0.0; self.n_chains * self.n_sites];
        let mut dp = vec![0u32; self.n_chains * self.words()];
        let d = match diff { ArgDiff::NoChange => 0, ArgDiff::Unknown => 1, ArgDiff::Extend => panic!("a DynGenFn's update takes NoChange or Unknown") };
        check(unsafe { sys::mp_fn_update(self.h, d, rng_step, ptr::null(), ptr::null(), 0, constraints.0.as_ptr(), constraints.1.as_ptr(), w.as_mut_ptr(),
                                         dv.as_mut_ptr(), dp.as_mut_ptr()) });
        (w, (dv, dp))
    }
    /// `(new_trace, weight) = model.regenerate(trace, args, diff, mask)`; an empty mask is the trace's whole schema
    pub fn regenerate(&mut self, mask: &[i32], diff: &ArgDiff, rng_step: u32) -> Vec<f64> {
        let mut w = vec![0.0; self.n_chains];
        let d = match diff { ArgDiff::NoChange => 0, ArgDiff::Unknown => 1, ArgDiff::Extend => panic!("a DynGenFn's regenerate takes NoChange or Unknown") };
        check(unsafe { sys::mp_fn_regenerate(self.h, d, rng_step, mask.as_ptr(), mask.len() as i32, w.as_mut_ptr()) });
        w
    }
    /// `weight = f.assess(args, constraints)`: `proposal = None` -> the model; `Some((kind, args))` -> that proposal on each chain's trace
    pub fn assess(&mut self, constraints: &(Vec<f64>, Vec<u32>), proposal: Option<(i32, &[f64])>, rng_step: u32) -> Vec<f64> {
        let mut w = vec![0.0; self.n_chains];
        let (kind, args): (i32, &[f64]) = proposal.unwrap_or((-1, &[]));
        check(unsafe { sys::mp_fn_assess(self.h, kind, args.as_ptr(), args.len() as i32, rng_step, ptr::null(), ptr::null(), 0, constraints.0.as_ptr(),
                                         constraints.1.as_ptr(), w.as_mut_ptr()) });
        w
    }
    /// `(choices, weight) = proposal.propose((trace, args))`
    pub fn propose(&mut self, proposal_kind: i32, args: &[f64], rng_step: u32) -> ((Vec<f64>, Vec<u32>), Vec<f64>) {
        let mut w = vec![0.0; self.n_chains];
        let mut cv = vec![0.0; self.n_chains * self.n_sites];
        let mut cp = vec![0u32; self.n_chains * self.words()];
        check(unsafe { sys::mp_fn_propose(self.h, proposal_kind, args.as_ptr(), args.len() as i32, rng_step, cv.as_mut_ptr(), cp.as_mut_ptr(), w.as_mut_ptr()) });
        ((cv, cp), w)
    }
}

impl Drop for FunctionChains {
    fn drop(&mut self) { unsafe { sys::mp_mh_destroy(self.h); } }
}

/// One rank of a filter sharded over `world` GPUs (one process per GPU).  `resample` is ONE library call: the library issues
/// its RCCL all-gather and grouped send/recv exchange itself, on the filter's stream (include/modppl_hip.h:
/// `mp_pf_shard_resample_rccl`).  `comm` is the host's `ncclComm_t` (e.g. from an `rccl-sys` binding).
pub struct ShardedParticleSystem { h: *mut sys::mp_pf, model: UnfoldModel, num_particles: usize, world: i32, rank: i32, comm: *mut c_void }

impl ShardedParticleSystem {
    pub fn new(model: UnfoldModel, particles_per_rank: usize, seed: u64, world: i32, rank: i32, comm: *mut c_void, device: i32) -> Self {
        let d = sys::mp_model_desc { kind: model.kind, dim_state: model.dim_state, dim_obs: model.dim_obs, n_params: model.params.len() as i32,
                                     params: model.params.as_ptr() };
        let sh = sys::mp_shard { n_global: (particles_per_rank * world as usize) as u64, slot_offset: (particles_per_rank * rank as usize) as u64 };
        let mut h = ptr::null_mut();
        check(unsafe { sys::mp_pf_create(&d, particles_per_rank as u64, seed, &sh, 0, device, ptr::null_mut(), &mut h) });
        ShardedParticleSystem { h, model, num_particles: particles_per_rank, world, rank, comm }
    }
    pub fn init_step(&mut self, args: &[f64], constraints: &[f64]) {
        check(unsafe { sys::mp_pf_init_step(self.h, if args.is_empty() { ptr::null() } else { args.as_ptr() }, constraints.as_ptr(),
                                            (constraints.len() / self.model.dim_obs as usize) as i32) });
    }
    pub fn step(self, constraints: &[f64]) -> Self {
        check(unsafe { sys::mp_pf_step(self.h, constraints.as_ptr(), (constraints.len() / self.model.dim_obs as usize) as i32) });
        self
    }
    /// log total weight of the whole job
    pub fn resample(&mut self, scheme: i32) -> f64 {
        let mut v = 0.0;
        check(unsafe { sys::mp_pf_shard_resample_rccl(self.h, self.comm, self.world, self.rank, scheme, 0, &mut v) });
        v
    }
    /// asynchronous: nothing is waited for but one polled word
    pub fn resample_async(&mut self, scheme: i32) {
        check(unsafe { sys::mp_pf_shard_resample_rccl(self.h, self.comm, self.world, self.rank, scheme, 0, ptr::null_mut()) });
    }
    /// `log_marginal_likelihood_estimate` of the whole job (one small all-gather of the ranks' tile scalars)
    pub fn log_marginal_likelihood_estimate(&mut self) -> f64 {
        let mut t = sys::mp_transport { ctx: ptr::null_mut(), all_gather: None, all_to_all: None };
        let (mut lml, mut ess) = (0.0, 0.0);
        if !self.comm.is_null() { check(unsafe { sys::mp_transport_rccl(self.comm, &mut t) }); }
        check(unsafe { sys::mp_pf_shard_query_native(self.h, if self.comm.is_null() { ptr::null() } else { &t }, self.world, 0, &mut lml, &mut ess) });
        lml
    }
    /// this rank's slots (offspring stay on the rank that owns their parent: slot numbers carry no meaning across ranks)
    pub fn states(&self) -> Vec<f64> {
        let mut x = vec![0.0; self.num_particles * self.model.dim_state as usize];
        check(unsafe { sys::mp_pf_read_state(self.h, x.as_mut_ptr()) });
        x
    }
}

impl Drop for ShardedParticleSystem {
    fn drop(&mut self) { unsafe { sys::mp_pf_destroy(self.h); } }
}

// ---------------------------------------------------------------------------------------------------------------------
// The GenFn surface (modppl/src/gfi.rs:49-92), batched.
//
// `trait GenFn<Args, Data, Ret>` is per trace: `generate(&self, args, constraints) -> (Trace, f64)`.  A device library cannot
// sit behind it — `ParticleSystem` calls it once per particle (particle_filter.rs:65,76), i.e. one launch per particle — so the
// drop-in sits one level up (ParticleSystem, importance_*, mh / regen_mh above).  What CAN keep the trait's shape is its
// batched counterpart: the same five methods with the same argument meaning, over N traces that live on the device.
// `DeviceTraces` plays `Trace<(i64, State), Vec<DynTrie>, Vec<State>>` for all N at once; `UnfoldModel` implements it the
// way `DynUnfold` implements `GenFn` (dynunfold.rs:22-100): `update` accepts `ArgDiff::Extend` only, `regenerate` keeps the
// trait's default panic (gfi.rs:66-73), `propose` / `assess` are the trait's provided methods.
// ---------------------------------------------------------------------------------------------------------------------
pub enum ArgDiff { NoChange, Unknown, Extend }   // gfi.rs:25-31

/// N traces of an Unfold model on the device: `args.0` steps taken, choices = the states, `retv.last()` readable.
pub struct DeviceTraces { pf: ParticleSystem, log_weights: Vec<f64> }

impl DeviceTraces {
    pub fn num_traces(&self) -> usize { self.pf.num_particles }
    /// `traces[i].retv.last()` for every i, particle-major
    pub fn last_states(&self) -> Vec<f64> { self.pf.states() }
    /// hand the traces to the particle filter API (they ARE a `ParticleSystem`'s `traces` field)
    pub fn into_particle_system(self) -> ParticleSystem { self.pf }
}

pub trait BatchedGenFn {
    /// `simulate(args)`: N x `GenFn::simulate((t, state0))` — every site sampled, the observation sites too (dynunfold.rs:22-39).
    /// Returns (states [n][t][dim_state], observations [n][t][dim_obs]).
    fn simulate(&self, n: usize, seed: u64, t: usize, state0: &[f64]) -> (Vec<f64>, Vec<f64>);
    /// `generate(args, constraints)`: N x `GenFn::generate((t, state0), constraints)` -> (traces, weights) (dynunfold.rs:41-64)
    fn generate(&self, n: usize, seed: u64, state0: &[f64], constraints: &[f64]) -> (DeviceTraces, Vec<f64>);
    /// `update(trace, args, ArgDiff::Extend, constraints)` -> (traces, discard = (), incremental weights) (dynunfold.rs:66-100)
    fn update(&self, traces: DeviceTraces, diff: ArgDiff, constraints: &[f64]) -> (DeviceTraces, (), Vec<f64>);
    /// gfi.rs:66-73: the default panics, and `DynUnfold` does not override it
    fn regenerate(&self, _traces: DeviceTraces, _diff: ArgDiff, _mask: &[i32]) -> (DeviceTraces, Vec<f64>) {
        panic!("regenerate: not implemented for Unfold models (gfi.rs:66-73 default)")
    }
    /// gfi.rs:81-84: `propose` = simulate -> (choices, logjp); an Unfold trace's logjp is 0 in the reference (dynunfold.rs:51,60)
    fn propose(&self, n: usize, seed: u64, t: usize, state0: &[f64]) -> ((Vec<f64>, Vec<f64>), Vec<f64>) {
        (self.simulate(n, seed, t, state0), vec![0.0; n])
    }
    /// gfi.rs:87-90: `assess` = generate(..).1
    fn assess(&self, n: usize, seed: u64, state0: &[f64], constraints: &[f64]) -> Vec<f64> { self.generate(n, seed, state0, constraints).1 }
}

impl BatchedGenFn for UnfoldModel {
    fn simulate(&self, n: usize, seed: u64, t: usize, state0: &[f64]) -> (Vec<f64>, Vec<f64>) { simulate(self, state0, t, n, seed) }
    fn generate(&self, n: usize, seed: u64, state0: &[f64], constraints: &[f64]) -> (DeviceTraces, Vec<f64>) {
        let mut pf = ParticleSystem::new(self.clone(), n, seed);
        pf.init_step(state0, constraints);                    // N x generate over all the steps the constraints cover
        let w = pf.log_weights();
        (DeviceTraces { pf, log_weights: w.clone() }, w)
    }
    fn update(&self, traces: DeviceTraces, diff: ArgDiff, constraints: &[f64]) -> (DeviceTraces, (), Vec<f64>) {
        match diff { ArgDiff::Extend => {}, _ => panic!("update: only ArgDiff::Extend is supported (dynunfold.rs:72)") }
        let before = traces.log_weights;
        let pf = traces.pf.step(constraints);
        let after = pf.log_weights();
        let inc: Vec<f64> = after.iter().zip(before.iter()).map(|(a, b)| a - b).collect();
        (DeviceTraces { pf, log_weights: after }, (), inc)
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// `dyngen!` on this path (sketch, source only).  modppl's proc-macro (modppl-macros/src/lib.rs:20-113) rewrites
// `dist(args) %= addr` into `__g.sample_at(&dist, (args), addr)` inside a closure over the dynamic handler; a closure cannot
// cross to the GPU, so here a model body lives ONCE as a C++ functor (`g.template normal<SITE>(mu, sd)`: the same rewrite,
// with the address turned into a compile-time site id) in modppl_amd/csrc/mp_models_extra.h, where
// `MP_REGISTER_UNFOLD_MODEL(kind, Type, parse)` makes it known to the device library AND to the CPU checker.  What the Rust
// side needs per model is only its descriptor; this macro declares one, so that a new model is one block of C++ plus:
//
//     unfold_model!(stochastic_volatility, kind = 100, dim_state = 1, dim_obs = 1, params = [mu, phi, sigma, sig0]);
//     let m = UnfoldModel::stochastic_volatility(-1.0, 0.95, 0.25, 0.8);
// ---------------------------------------------------------------------------------------------------------------------
#[macro_export]
macro_rules! unfold_model {
    ($name:ident, kind = $kind:expr, dim_state = $ds:expr, dim_obs = $dobs:expr, params = [$($p:ident),*]) => {
        impl $crate::UnfoldModel {
            pub fn $name($($p: f64),*) -> Self {
                $crate::UnfoldModel { kind: $kind, dim_state: $ds, dim_obs: $dobs, params: vec![$($p),*] }
            }
        }
    };
}
unfold_model!(stochastic_volatility, kind = 100, dim_state = 1, dim_obs = 1, params = [mu, phi, sigma, sig0]);
