//! Raw declarations of include/modppl_hip.h.  SOURCE ONLY (no rustc in the build image).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_void};

pub const MP_OK: i32 = 0;
pub const MP_ERR_INVALID_ARG: i32 = 1;
pub const MP_ERR_STATE: i32 = 2;
pub const MP_ERR_CONSTRAINTS: i32 = 3;
pub const MP_ERR_DEGENERATE: i32 = 4;
pub const MP_ERR_HIP: i32 = 5;
pub const MP_ERR_UNSUPPORTED: i32 = 6;
pub const MP_ERR_CAPACITY: i32 = 7;

pub const MP_MODEL_LGSSM1: i32 = 1;
pub const MP_MODEL_SPIRAL: i32 = 2;
pub const MP_MODEL_HMM: i32 = 3;
pub const MP_MODEL_BEARINGS: i32 = 4;
pub const MP_MODEL_LGSSM_BAND: i32 = 5;
pub const MP_MODEL_POINTED_2D: i32 = 6;
pub const MP_MODEL_LINE: i32 = 7;
pub const MP_MODEL_LGSSM_DENSE: i32 = 8;
pub const MP_RESAMPLE_MULTINOMIAL: i32 = 0;
pub const MP_RESAMPLE_SYSTEMATIC: i32 = 1;
pub const MP_RESAMPLE_STRATIFIED: i32 = 2;
pub const MP_RESAMPLE_MULTINOMIAL_SPLIT: i32 = 3;
pub const MP_MH_MODEL_HIERARCHICAL: i32 = 1;
pub const MP_MH_MODEL_POINTED_2D: i32 = 2;
pub const MP_MH_MODEL_HIERARCHICAL_FN: i32 = 101;   // the hierarchical model as a registered functor (mp_mh_create_fn)
pub const MP_MH_MODEL_HIERARCHICAL_DATA_FN: i32 = 105;   // the same model with its observations declared as data sites (any number)
pub const MP_MH_PROPOSAL_POINTED_DRIFT: i32 = 3;
pub const MP_SITE_IS_LINEAR: i32 = 0;
pub const MP_SITE_A: i32 = 1;
pub const MP_SITE_B: i32 = 2;
pub const MP_SITE_C: i32 = 3;
pub const MP_SITE_Y0: i32 = 4;
pub const MP_MH_PROPOSAL_HIERARCHICAL_DRIFT: i32 = 1;
pub const MP_MH_PROPOSAL_HIERARCHICAL_ADD_OR_REMOVE: i32 = 2;
pub const MP_ESS_REFERENCE: i32 = 0;
pub const MP_ESS_FRESH: i32 = 1;
pub const MP_PF_RECORD_HISTORY: u32 = 1;

#[repr(C)]
pub struct mp_model_desc {
    pub kind: i32,
    pub dim_state: i32,
    pub dim_obs: i32,
    pub n_params: i32,
    pub params: *const f64,
}
#[repr(C)]
pub struct mp_shard {
    pub n_global: u64,
    pub slot_offset: u64,
}
/// The two collectives of the sharded resample as plain C function pointers (include/modppl_hip.h: `mp_transport`).
#[repr(C)]
pub struct mp_transport {
    pub ctx: *mut c_void,
    pub all_gather: Option<unsafe extern "C" fn(ctx: *mut c_void, d_send: *const c_void, d_recv: *mut c_void, bytes_per_rank: u64,
                                               stream: *mut c_void) -> i32>,
    pub all_to_all: Option<unsafe extern "C" fn(ctx: *mut c_void, d_send: *const c_void, send_off: *const u64, send_bytes: *const u64,
                                               d_recv: *mut c_void, recv_off: *const u64, recv_bytes: *const u64, world: i32,
                                               stream: *mut c_void) -> i32>,
}
#[repr(C)]
pub struct mp_pf { _private: [u8; 0] }
#[repr(C)]
pub struct mp_mh { _private: [u8; 0] }

extern "C" {
    pub fn mp_last_error() -> *const c_char;
    pub fn mp_device_count() -> i32;
    pub fn mp_pf_create(model: *const mp_model_desc, n_particles: u64, seed: u64, shard: *const mp_shard, flags: u32,
                        device: i32, stream: *mut c_void, out: *mut *mut mp_pf) -> i32;
    pub fn mp_pf_init_step(h: *mut mp_pf, args0: *const f64, obs: *const f64, n_steps: i32) -> i32;
    pub fn mp_pf_step(h: *mut mp_pf, obs: *const f64, n_steps: i32) -> i32;
    pub fn mp_pf_effective_sample_size(h: *mut mp_pf, ess_mode: i32, out: *mut f64) -> i32;
    pub fn mp_pf_resample(h: *mut mp_pf, scheme: i32, log_total_weight: *mut f64) -> i32;
    pub fn mp_pf_resample_if_ess_below(h: *mut mp_pf, scheme: i32, ess_fraction: f64, resampled: *mut i32, ess_out: *mut f64,
                                       log_total_weight: *mut f64) -> i32;
    pub fn mp_pf_log_marginal_likelihood_estimate(h: *mut mp_pf, out: *mut f64) -> i32;
    pub fn mp_pf_read_state(h: *mut mp_pf, x_out: *mut f64) -> i32;
    pub fn mp_pf_read_log_weights(h: *mut mp_pf, out: *mut f64) -> i32;
    pub fn mp_pf_read_parents(h: *mut mp_pf, out: *mut u32) -> i32;
    pub fn mp_pf_read_trajectory(h: *mut mp_pf, i: u64, out: *mut f64, t_steps: *mut i32) -> i32;
    pub fn mp_pf_read_trajectories(h: *mut mp_pf, first: u64, count: u64, out: *mut f64, t_steps: *mut i32) -> i32;
    pub fn mp_pf_time(h: *mut mp_pf, out: *mut i64) -> i32;
    pub fn mp_pf_run(h: *mut mp_pf, args0: *const f64, obs: *const f64, n_steps: i32, scheme: i32) -> i32;
    pub fn mp_pf_synchronize(h: *mut mp_pf) -> i32;
    pub fn mp_pf_destroy(h: *mut mp_pf) -> i32;
    // sharded filter: device pointers; the caller runs the collectives between the phases (include/modppl_hip.h)
    pub fn mp_pf_shard_tiles(h: *mut mp_pf, d_tile_m: *mut f64, d_tile_w: *mut u64, d_tile_w2: *mut u64) -> i32;
    pub fn mp_pf_shard_route(h: *mut mp_pf, scheme: i32, d_tm_all: *const f64, d_tw_all: *const u64, d_tw2_all: *const u64,
                             world: i32, rank: i32, d_req_out: *mut u64, send_counts: *mut i64) -> i32;
    pub fn mp_pf_shard_resolve(h: *mut mp_pf, d_req_in: *const u64, n_req: u64, d_rows_out: *mut f64) -> i32;
    pub fn mp_pf_shard_scatter(h: *mut mp_pf, d_rows_in: *const f64, log_total_weight: *mut f64) -> i32;
    pub fn mp_pf_shard_query(h: *mut mp_pf, d_tm_all: *const f64, d_tw_all: *const u64, d_tw2_all: *const u64, world: i32,
                             log_ml: *mut f64, ess: *mut f64) -> i32;
    pub fn mp_pf_shard_bind_tiles(h: *mut mp_pf, d_tiles: *mut u64) -> i32;
    pub fn mp_pf_shard_tiles_packed(h: *mut mp_pf, d_tiles_out: *mut u64) -> i32;
    pub fn mp_pf_shard_route_fixed(h: *mut mp_pf, scheme: i32, d_tiles_all: *const u64, world: i32, rank: i32, capacity: u64,
                                   d_req_out: *mut u64) -> i32;
    pub fn mp_pf_shard_resolve_fixed(h: *mut mp_pf, d_req_in: *const u64, world: i32, capacity: u64, d_rows_out: *mut f64) -> i32;
    pub fn mp_pf_shard_commit_fixed(h: *mut mp_pf, d_rows_in: *const f64, log_total_weight: *mut f64) -> i32;
    pub fn mp_pf_shard_query_packed(h: *mut mp_pf, d_tiles_all: *const u64, world: i32, log_ml: *mut f64, ess: *mut f64) -> i32;
    pub fn mp_pf_shard_owned_count(h: *mut mp_pf, scheme: i32, d_tiles_all: *const u64, world: i32, rank: i32, capacity: u64, counts_out: *mut u64) -> i32;
    pub fn mp_pf_shard_owned_expand(h: *mut mp_pf, world: i32, rank: i32, capacity: u64, d_send_out: *mut f64, d_rows: *mut f64, recv_rows: u64) -> i32;
    pub fn mp_pf_shard_owned_count_expand(h: *mut mp_pf, scheme: i32, d_tiles_all: *const u64, world: i32, rank: i32, capacity: u64, d_send_out: *mut f64, d_rows: *mut f64, recv_rows: u64) -> i32;
    pub fn mp_pf_shard_owned_commit(h: *mut mp_pf, d_rows: *const f64, log_total_weight: *mut f64, counts_out: *mut u64) -> i32;
    // the whole sharded resample behind one call (the collectives go through `mp_transport`, or RCCL directly)
    pub fn mp_pf_shard_resample(h: *mut mp_pf, t: *const mp_transport, world: i32, rank: i32, scheme: i32, force_collectives: i32,
                                log_total_weight: *mut f64) -> i32;
    pub fn mp_pf_shard_resample_rccl(h: *mut mp_pf, nccl_comm: *mut c_void, world: i32, rank: i32, scheme: i32, force_collectives: i32,
                                     log_total_weight: *mut f64) -> i32;
    pub fn mp_pf_shard_query_native(h: *mut mp_pf, t: *const mp_transport, world: i32, force_collectives: i32, log_ml: *mut f64,
                                    ess: *mut f64) -> i32;
    pub fn mp_pf_shard_resample_stats(h: *mut mp_pf, fallbacks: *mut u64, exchange_rows: *mut u64, counts_out: *mut u64,
                                      capacity: *mut u64) -> i32;
    pub fn mp_transport_rccl(nccl_comm: *mut c_void, out: *mut mp_transport) -> i32;
    pub fn mp_rccl_available() -> i32;
    pub fn mp_rccl_unique_id(out128: *mut c_void) -> i32;
    pub fn mp_rccl_comm_create(world: i32, rank: i32, id128: *const c_void, device: i32, comm_out: *mut *mut c_void) -> i32;
    pub fn mp_rccl_comm_destroy(comm: *mut c_void) -> i32;
    pub fn mp_pf_stream_copy(h: *mut mp_pf, dst: *mut c_void, src: *const c_void, bytes: u64, to_host: i32) -> i32;
    // per-kernel-family hipEvent timing (bench)
    pub fn mp_pf_set_timing(h: *mut mp_pf, enabled: i32) -> i32;
    pub fn mp_pf_get_timing(h: *mut mp_pf, family: i32, total_ms: *mut f64, launches: *mut u64) -> i32;
    pub fn mp_pf_last_propagate_form(h: *mut mp_pf, out: *mut i32) -> i32;
    pub fn mp_pf_region_begin(h: *mut mp_pf) -> i32;
    pub fn mp_pf_region_end(h: *mut mp_pf, elapsed_ms: *mut f64, propagate_launches: *mut u64) -> i32;
    pub fn mp_unfold_simulate(model: *const mp_model_desc, args0: *const f64, n_steps: i32, n: u64, seed: u64, device: i32,
                              states_out: *mut f64, obs_out: *mut f64) -> i32;
    pub fn mp_importance_sampling(model: *const mp_model_desc, args0: *const f64, obs: *const f64, n_steps: i32, num_samples: u64, seed: u64,
                                  device: i32, log_ml_estimate: *mut f64, log_normalized_weights: *mut f64, trajectories_out: *mut f64) -> i32;
    pub fn mp_importance_resampling(model: *const mp_model_desc, args0: *const f64, obs: *const f64, n_steps: i32,
                                    num_samples: u64, num_ret_samples: u64, seed: u64, device: i32,
                                    log_ml_estimate: *mut f64, log_normalized_weights: *mut f64,
                                    resampled_indices: *mut u64, final_states: *mut f64) -> i32;
    pub fn mp_mh_create(model_kind: i32, xs: *const f64, ys: *const f64, n_data: i32, constrain_is_linear: i32,
                        n_chains: u64, seed: u64, device: i32, stream: *mut c_void, out: *mut *mut mp_mh) -> i32;
    pub fn mp_mh_create_pointed(bounds: *const f64, obs_cov: *const f64, obs: *const f64, n_chains: u64, seed: u64, device: i32,
                                stream: *mut c_void, out: *mut *mut mp_mh) -> i32;
    pub fn mp_mh_step(h: *mut mp_mh, proposal_kind: i32, proposal_args: *const f64, n_proposal_args: i32, n_iters: i32,
                      accepted: *mut u64) -> i32;
    pub fn mp_regen_mh_step(h: *mut mp_mh, mask_sites: *const i32, n_mask: i32, cycle: i32, n_iters: i32, accepted: *mut u64) -> i32;
    pub fn mp_mh_read_state(h: *mut mp_mh, out: *mut f64) -> i32;
    pub fn mp_mh_read_logjp(h: *mut mp_mh, out: *mut f64) -> i32;
    pub fn mp_mh_read_observations(h: *mut mp_mh, out: *mut f64) -> i32;
    // chains of a registered generative function (csrc/mp_mh_models.h)
    pub fn mp_mh_create_fn(model_kind: i32, params: *const f64, n_params: i32, constraint_sites: *const i32, constraint_values: *const f64,
                           n_constraints: i32, n_chains: u64, seed: u64, device: i32, stream: *mut c_void, out: *mut *mut mp_mh) -> i32;
    pub fn mp_mh_n_sites(h: *mut mp_mh, out: *mut i32) -> i32;
    pub fn mp_mh_read_trace(h: *mut mp_mh, values: *mut f64, present: *mut u32) -> i32;
    pub fn mp_fn_update(h: *mut mp_mh, argdiff: i32, rng_step: u32, sites: *const i32, values: *const f64, n_constraints: i32,
                        chain_values: *const f64, chain_present: *const u32, weights_out: *mut f64, discard_values_out: *mut f64,
                        discard_present_out: *mut u32) -> i32;
    pub fn mp_fn_regenerate(h: *mut mp_mh, argdiff: i32, rng_step: u32, mask_sites: *const i32, n_mask: i32, weights_out: *mut f64) -> i32;
    pub fn mp_fn_assess(h: *mut mp_mh, proposal_kind: i32, proposal_args: *const f64, n_proposal_args: i32, rng_step: u32, sites: *const i32,
                        values: *const f64, n_constraints: i32, chain_values: *const f64, chain_present: *const u32, weights_out: *mut f64) -> i32;
    pub fn mp_fn_propose(h: *mut mp_mh, proposal_kind: i32, proposal_args: *const f64, n_proposal_args: i32, rng_step: u32,
                         choice_values_out: *mut f64, choice_present_out: *mut u32, weights_out: *mut f64) -> i32;
    pub fn mp_fn_generate(h: *mut mp_mh, rng_step: u32, sites: *const i32, values: *const f64, n_constraints: i32, chain_values: *const f64,
                          chain_present: *const u32, weights_out: *mut f64) -> i32;
    pub fn mp_fn_simulate(h: *mut mp_mh, rng_step: u32, logjp_out: *mut f64) -> i32;
    pub fn mp_fn_generate_create(model_kind: i32, params: *const f64, n_params: i32, constraint_sites: *const i32, constraint_values: *const f64,
                                 n_constraints: i32, n_chains: u64, seed: u64, device: i32, stream: *mut c_void, weights_out: *mut f64,
                                 out: *mut *mut mp_mh) -> i32;
    pub fn mp_fn_simulate_create(model_kind: i32, params: *const f64, n_params: i32, n_chains: u64, seed: u64, device: i32, stream: *mut c_void,
                                 logjp_out: *mut f64, out: *mut *mut mp_mh) -> i32;
    // importance sampling over a registered generative function (importance.rs:12-50)
    pub fn mp_fn_importance_sampling(model_kind: i32, params: *const f64, n_params: i32, constraint_sites: *const i32, constraint_values: *const f64,
                                     n_constraints: i32, num_samples: u64, seed: u64, device: i32, log_ml_estimate: *mut f64,
                                     log_normalized_weights: *mut f64, traces_out: *mut *mut mp_mh) -> i32;
    pub fn mp_fn_importance_resampling(model_kind: i32, params: *const f64, n_params: i32, constraint_sites: *const i32, constraint_values: *const f64,
                                       n_constraints: i32, num_samples: u64, num_ret_samples: u64, seed: u64, device: i32, log_ml_estimate: *mut f64,
                                       log_normalized_weights: *mut f64, resampled_indices: *mut u64, traces_out: *mut *mut mp_mh) -> i32;
    pub fn mp_mh_iterations(h: *mut mp_mh, out: *mut u64) -> i32;
    pub fn mp_mh_destroy(h: *mut mp_mh) -> i32;
}
