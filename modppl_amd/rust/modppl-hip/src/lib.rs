//! Drop-in for modppl's `ParticleSystem` over `DynUnfold` models on an MI355X.  SOURCE ONLY (no rustc in the build image).
//!
//! Same method names and ownership as `modppl::inference::ParticleSystem` (modppl/src/inference/particle_filter.rs):
//! `step(self) -> Self` consumes the filter, `resample(&mut self) -> f64`; a non-zero status from the library is
//! re-raised as a panic, the reference's only error convention.
use modppl_hip_sys as sys;
use std::ffi::CStr;
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

impl Drop for ParticleSystem {
    fn drop(&mut self) { unsafe { sys::mp_pf_destroy(self.h); } }
}
