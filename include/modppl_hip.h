/* modppl_hip.h — C ABI of the MI355X (gfx950) particle-filter / importance / MH hot path.
 *
 * agarret7/modppl has no FFI or plugin seam; its inference library is generic Rust over the
 * `GenFn` trait and owns the per-particle loops itself (modppl/src/inference/particle_filter.rs:65,76,110;
 * importance.rs:18-19).  The drop-in boundary therefore sits at the level of `ParticleSystem`
 * and of the free functions `importance_sampling`/`importance_resampling`/`mh`/`regen_mh`
 * (modppl/src/inference/mod.rs:8-10).  Each entry point below names the reference item it
 * replaces.  A Rust `-sys` binding for these symbols is shown in INTEGRATION.md.
 *
 * Conventions
 *  - one opaque handle per filter / chain set; a handle is NOT thread-safe (the reference's
 *    ParticleSystem is !Send because it owns a ThreadRng: particle_filter.rs:21).
 *  - every call returns int32 status; 0 = ok.  A non-zero status stands for a reference
 *    `panic!`/`assert!` (the reference has no Result anywhere); mp_last_error() gives the text.
 *  - all in/out buffers are caller-owned HOST memory unless a parameter says "device";
 *    device memory is owned by the handle.  No callbacks cross the ABI: a model is chosen by
 *    descriptor {kind, dims, params[]} and is compiled into the library as a static
 *    handler-polymorphic kernel (the GPU stand-in for a `dyngen!` function).
 *  - randomness: Philox4x32-10 keyed by `seed`, counter = (slot, step, domain|site, attempt);
 *    see modppl_amd/csrc/mp_philox.h.  The reference's ThreadRng cannot be seeded.
 *  - `stream`: a hipStream_t passed as void*, or NULL for the handle's own stream.  Work is
 *    enqueued asynchronously; only calls that return a value to the host synchronise.
 */
#ifndef MODPPL_HIP_H
#define MODPPL_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes -------------------------------------------------------------------- */
#define MP_OK 0
#define MP_ERR_INVALID_ARG 1 /* bad descriptor, null pointer, size mismatch                          */
#define MP_ERR_STATE 2       /* call order the reference panics on (step/resample before init_step)  */
#define MP_ERR_CONSTRAINTS 3 /* wrong number of constraints: dynunfold.rs:57,75 asserts               */
#define MP_ERR_DEGENERATE 4  /* all log-weights -inf: NaN probabilities, categorical.rs:23 assert    */
#define MP_ERR_HIP 5         /* HIP runtime failure (no device, out of memory, launch failure)       */
#define MP_ERR_UNSUPPORTED 6 /* model / option not compiled into the library                         */
#define MP_ERR_CAPACITY 7    /* fixed-capacity sharded exchange overflowed: nothing committed, repeat variable-size */

/* ---- models -------------------------------------------------------------------------- */
enum mp_model_kind {
    /* LGSSM d=1 (BASELINE.json configs 1-2).  params = {mu0, sig0, a, sig_x, sig_y}:
     *   t==0: x ~ normal(mu0, sig0);  t>0: x ~ normal(a*x_prev, sig_x);  y ~ normal(x, sig_y) observed */
    MP_MODEL_LGSSM1 = 1,
    /* modppl/tests/dyngenfns/unfold.rs:14-32 spiral_kernel.  no params; state = (r, theta), obs = 2 */
    MP_MODEL_SPIRAL = 2,
    /* modppl/tests/hmm/model.rs:33-80.  params = {n_states, n_obs, prior[S], emission[O][S], transition[S][S]}
     * (column-stochastic, as tests/particle_filter.rs:41-51 builds them); state = 1, obs = 1 (integer valued) */
    MP_MODEL_HMM = 3,
    /* bearings-only tracker d=4 (BASELINE.json config 3).  params = {p0x,p0y,sig_p0,sig_v0,sig_a,sig_theta} */
    MP_MODEL_BEARINGS = 4,
    /* LGSSM d=D (config 5): x' = A x + sig_x z, y = x + sig_y e.  params = {D, a, band, sig0, sig_x, sig_y} */
    MP_MODEL_LGSSM_BAND = 5,
    /* static (single-step) models of modppl/tests/importance.rs, for mp_importance_resampling with n_steps = 1:
     * pointed_2d_model (tests/dyngenfns/simple.rs:27-34): params = {xmin,xmax,ymin,ymax, cov[4] row-major}; state = latent (2), obs = 2 */
    MP_MODEL_POINTED_2D = 6,
    /* line_model (simple.rs:9-24): params = xs[11]; state = (slope, intercept), obs = ys[11] */
    MP_MODEL_LINE = 7,
    /* Dense LGSSM, dim_state = dim_obs = D (16 compiled in): t==0: x ~ mvnormal(0, sig0^2 I); t>0: x ~ mvnormal(A x_prev, Q);
     * mvnormal(x, R) observed — two mvnormal sites (mvnormal.rs:14-38; Q may be singular: eigen transform, :30-33).
     * params = {D, sig0, A[D*D], Q[D*D], R[D*D]} row-major.  The matrix products run on the matrix cores. */
    MP_MODEL_LGSSM_DENSE = 8,
};

typedef struct mp_model_desc {
    int32_t kind;       /* enum mp_model_kind                     */
    int32_t dim_state;  /* doubles per particle state             */
    int32_t dim_obs;    /* doubles per time step of constraints   */
    int32_t n_params;
    const double* params;
} mp_model_desc;

/* Which global slots this process owns (one process per GPU).  NULL = all of them. */
typedef struct mp_shard {
    uint64_t n_global;    /* particles in the whole job                          */
    uint64_t slot_offset; /* first global slot owned by this handle              */
} mp_shard;

enum mp_resample_scheme {
    MP_RESAMPLE_MULTINOMIAL = 0, /* particle_filter.rs:37-41 (the only scheme the reference has) */
    MP_RESAMPLE_SYSTEMATIC = 1,  /* extension named by the north star; no reference counterpart: one uniform u0 per resample,
                                  * u_g = (g + u0) / N; parents come out sorted, so the gather is coalesced and the multi-GPU
                                  * exchange near-neighbour */
    MP_RESAMPLE_STRATIFIED = 2,  /* extension: the same lattice with one uniform per output slot, u_g = (g + u_g') / N; sorted parents too */
    MP_RESAMPLE_MULTINOMIAL_SPLIT = 3, /* particle_filter.rs:37-41's law for SHARDED filters at O(n) work per rank (opt-in): the offspring
                                  * per rank are drawn first — Multinomial(N; rank masses), by binary splitting with one binomial variate
                                  * per node (csrc/mp_binomial.h) — then every rank draws its own c_r parents i.i.d. from its own weights.
                                  * Same law as MP_RESAMPLE_MULTINOMIAL, another seeded stream: the parents differ from the single
                                  * filter's unless the world is one rank (then they are the same, bit for bit; an unsharded filter
                                  * takes this value as MP_RESAMPLE_MULTINOMIAL).  DESIGN.md §8.3 */
};
enum mp_ess_mode {
    MP_ESS_REFERENCE = 0, /* particle_filter.rs:98-100: from the weights normalised by the LAST resample() (1/N before any) */
    MP_ESS_FRESH = 1,     /* from the current log-weights                                                                   */
};
#define MP_PF_RECORD_HISTORY 1u /* keep x[t] and parent[r] so that `traces[i].retv` can be rebuilt */

typedef struct mp_pf mp_pf;

const char* mp_last_error(void);
/* Number of visible gfx950 devices (0 when none): lets callers fail loudly before creating anything. */
int32_t mp_device_count(void);

/* ParticleSystem::new — particle_filter.rs:44-57.  `seed` replaces `rng: ThreadRng`. */
int32_t mp_pf_create(const mp_model_desc* model, uint64_t n_particles, uint64_t seed, const mp_shard* shard,
                     uint32_t flags, int32_t device, void* stream, mp_pf** out);
/* ParticleSystem::init_step — :60-70 (N x DynUnfold::generate, dynunfold.rs:41-64).
 * `args0` = the Unfold's initial state argument (dim_state doubles, may be NULL = zeros);
 * `obs` = constraints of n_steps >= 1 time steps, [n_steps][dim_obs]. */
int32_t mp_pf_init_step(mp_pf* h, const double* args0, const double* obs, int32_t n_steps);
/* ParticleSystem::step — :73-96 (N x DynUnfold::update with ArgDiff::Extend, dynunfold.rs:66-100). */
int32_t mp_pf_step(mp_pf* h, const double* obs, int32_t n_steps);
/* ParticleSystem::effective_sample_size — :98-100. */
int32_t mp_pf_effective_sample_size(mp_pf* h, int32_t ess_mode, double* out);
/* ParticleSystem::resample — :103-116.  Returns the log total weight through `log_total_weight`;
 * pass NULL to enqueue without synchronising (the value still feeds the log-ML estimate).
 * Degenerate weights (all -inf / NaN: where the reference's categorical asserts, categorical.rs:23) are MP_ERR_DEGENERATE from
 * this call when it synchronises; from an asynchronous one the error is sticky and surfaces at the next synchronising call
 * AFTER the work that normalises has run — for filters whose next mp_pf_step makes the resample's draws itself (dim_state <= 4
 * and at most four normal sites, at most 2^22 particles; any of the three schemes) that is the first synchronising call after that step, not an
 * mp_pf_synchronize directly behind the resample: nothing of the resample has been enqueued by then. */
int32_t mp_pf_resample(mp_pf* h, int32_t scheme, double* log_total_weight);
/* ESS-triggered resampling (extension named by the north star): resample with `scheme` iff the effective sample size of the
 * CURRENT weights is below ess_fraction * n_particles.  *resampled = 0 / 1; ess_out and log_total_weight may be NULL. */
int32_t mp_pf_resample_if_ess_below(mp_pf* h, int32_t scheme, double ess_fraction, int32_t* resampled, double* ess_out,
                                    double* log_total_weight);
/* ParticleSystem::log_marginal_likelihood_estimate — :119-121. */
int32_t mp_pf_log_marginal_likelihood_estimate(mp_pf* h, double* out);
/* The pub `traces` field (:13): traces[i].retv.last() for all i -> x_out[n_local][dim_state]. */
int32_t mp_pf_read_state(mp_pf* h, double* x_out);
/* log_weights (:15) -> out[n_local]. */
int32_t mp_pf_read_log_weights(mp_pf* h, double* out);
/* parents (:20) of the most recent resample -> out[n_local] (global slot ids). */
int32_t mp_pf_read_parents(mp_pf* h, uint32_t* out);
/* traces[i].retv (Vec<State>) rebuilt from the recorded ancestry -> out[t_steps][dim_state].
 * Needs MP_PF_RECORD_HISTORY. */
int32_t mp_pf_read_trajectory(mp_pf* h, uint64_t i, double* out, int32_t* t_steps);
/* The same for the particles first .. first + count - 1 at once (one kernel walks the recorded ancestry of all of them):
 * out[count][t][dim_state].  tests/smc.rs:67 reads `retv` of every particle. */
int32_t mp_pf_read_trajectories(mp_pf* h, uint64_t first, uint64_t count, double* out, int32_t* t_steps);
/* Number of Unfold steps taken so far (trace.args.0). */
int32_t mp_pf_time(mp_pf* h, int64_t* out);
/* Whole filter in one call: init_step on obs[0], then for t=1..T-1 {step; resample} with a
 * resample after the first step too — the loop of modppl/tests/smc.rs:64-90.  Enqueues
 * everything without host synchronisation.  resample_every = 1 reproduces smc.rs. */
int32_t mp_pf_run(mp_pf* h, const double* args0, const double* obs, int32_t n_steps, int32_t scheme);
/* Wait for everything enqueued on the handle; surfaces sticky device-side errors. */
int32_t mp_pf_synchronize(mp_pf* h);
int32_t mp_pf_destroy(mp_pf* h);

/* ---- sharded filter: one process per GPU, particles split into contiguous global-slot ranges ----
 * A handle created with a `shard` {n_global, slot_offset} owns n_particles of n_global slots (equal, tile-aligned
 * shards: slot_offset and n_particles multiples of 2048).  Philox is keyed by GLOBAL slot and the normalisation is
 * hierarchical over tiles of 2048 global slots, so results do not depend on the number of shards.
 * init_step / step / read_* work unchanged; resample is split into phases with the collectives (RCCL via
 * torch.distributed, or any transport) run by the caller in between.  All pointers in this group are in the handle's
 * memory space (DEVICE pointers for this library), enqueued on the handle's stream:
 *
 *   shard_tiles    -> [all-gather of 24 B per tile]    per-tile max log-weight + fixed-point totals: every rank then
 *                                                      builds the same tile table (no separate max all-reduce)
 *   shard_route    -> [all-to-all of request pairs]    each draw goes to the rank that owns its tile
 *   shard_resolve  -> [all-to-all of f64 rows back]    the particle exchange over xGMI
 *   shard_scatter
 */
/* level 0 of the normalisation of this shard: d_tile_m / d_tile_W / d_tile_W2 [n_particles / 2048 (rounded up)] */
int32_t mp_pf_shard_tiles(mp_pf* h, double* d_tile_m, uint64_t* d_tile_W, uint64_t* d_tile_W2);
/* Draw this shard's n targets against the gathered tiles of ALL ranks (rank-major, world * tiles-per-shard entries each),
 * find tile and owner, and pack the requests grouped by owner (stable): d_req_out[n][2] = {tile inside the owner's
 * shard, tile-local target}.  send_counts[world] (HOST, synchronises) = requests per owner.  Also folds the global log
 * total weight into the log-ML estimate. */
int32_t mp_pf_shard_route(mp_pf* h, int32_t scheme, const double* d_tile_m_all, const uint64_t* d_tile_W_all, const uint64_t* d_tile_W2_all,
                          int32_t world, int32_t rank, uint64_t* d_req_out, int64_t* send_counts);
/* Owner side: resolve n_req request pairs to parents; d_rows_out[n_req][dim_state + 1] = parent state, then the
 * parent's global slot id as a double. */
int32_t mp_pf_shard_resolve(mp_pf* h, const uint64_t* d_req_in, uint64_t n_req, double* d_rows_out);
/* Requester side: rows (in the order the requests were packed) -> new states / parents; log-weights = 0; returns the
 * global log total weight through log_total_weight if non-NULL (synchronises). */
int32_t mp_pf_shard_scatter(mp_pf* h, const double* d_rows_in, double* log_total_weight);
/* log_marginal_likelihood_estimate / fresh ESS of the whole job from the gathered tiles (after shard_tiles). */
int32_t mp_pf_shard_query(mp_pf* h, const double* d_tile_m_all, const uint64_t* d_tile_W_all, const uint64_t* d_tile_W2_all, int32_t world,
                          double* log_ml, double* ess);

/* Fixed-capacity form of the same phases: nothing synchronises with the host until the commit and the all-to-alls
 * have equal splits.  Tiles travel packed, [3][tiles] 8-byte words per rank (bits of the f64 maxima, W, W2), gathered
 * rank-major.  Draws are grouped by (owner rank, eighth of the owner's tiles) so the owner resolves each group on one XCD:
 *   d_req [world][8][capacity + 1][2] u64   entry 0 of a sub-segment = {count, overflow flag}, then {tile, local target}
 *   d_rows[world][8][capacity][dim_state + 1] f64   the parents' states and global ids, in request order
 * mp_pf_shard_commit_fixed makes the one host wait: if any sub-segment needed more than `capacity` draws (every
 * rank sees the same answer) it commits nothing and returns MP_ERR_CAPACITY; the caller then repeats the resample with
 * the variable-size phases (mp_pf_shard_route/resolve/scatter), which read the same rows, tiles and Philox counters.
 * Sub-segment loads are equal up to O(cv sqrt(8 world / n)), so a capacity of 1.25 n / (8 world) is only exceeded by
 * collapsed weights.  The order of requests inside a sub-segment is not deterministic; the filter's results are. */
int32_t mp_pf_shard_bind_tiles(mp_pf* h, uint64_t* d_tiles);   /* from now on the filter keeps its packed tiles in this caller-owned
                                                                  * buffer ([3][tiles] words): the all-gather reads it in place  */
int32_t mp_pf_shard_tiles_packed(mp_pf* h, uint64_t* d_tiles_out); /* normalise if needed; copy unless d_tiles_out is the bound one */
int32_t mp_pf_shard_route_fixed(mp_pf* h, int32_t scheme, const uint64_t* d_tiles_all, int32_t world, int32_t rank, uint64_t capacity,
                                uint64_t* d_req_out);
int32_t mp_pf_shard_resolve_fixed(mp_pf* h, const uint64_t* d_req_in, int32_t world, uint64_t capacity, double* d_rows_out);
/* Waits for the resolve only (not for the all-to-all that is carrying the rows into d_rows_in) and commits: the next
 * mp_pf_step reads the parents' states straight from d_rows_in, which must therefore be filled by work enqueued on the
 * filter's stream before that step and stay untouched until it has run; any other reader first copies them into slot order. */
int32_t mp_pf_shard_commit_fixed(mp_pf* h, const double* d_rows_in, double* log_total_weight);
int32_t mp_pf_shard_query_packed(mp_pf* h, const uint64_t* d_tiles_all, int32_t world, double* log_ml, double* ess);

/* ---- "owner keeps" form of the sharded resample: the exchange that fits point-to-point xGMI ------------------------
 * `traces[i] = traces[parents[i]].clone()` (particle_filter.rs:109-113) treats particles as exchangeable: WHICH slot an
 * offspring lands in carries no meaning.  With iid parents, keeping slot order across G ranks moves (G-1)/G of all
 * particles over the links at every resample.  Here every rank enumerates all N draws of the job (the draws of the single
 * filter: same Philox counters and targets, hence the same parent for every draw g), keeps the ones that land in its own
 * rows, and places the offspring on the rank that owns their parent, in the order of their draws; only each rank's surplus
 * over its n slots travels, to the ranks that drew fewer than n: unit u of the job's surplus (donors in rank order, each
 * donor's offspring n, n+1, ... in order) fills unit u of the job's deficit (receivers in rank order, slots c_s, c_s+1,
 * ...).  The traffic is O(sqrt(N)) rows instead of O(N); one all-to-all (of rows) instead of two.  The price: slot contents
 * depend on the number of ranks (same parents, different places; a world of one IS the single filter); results are
 * deterministic for a given (seed, world).
 *
 *   mp_pf_shard_owned_count   all-gathered packed tiles -> tile table, log-ML fold, every rank's offspring count, this
 *                             rank's own draws in draw order, the exchange plan and — capacity > 0 — the verdict "some pair
 *                             of ranks exchanges more than `capacity` rows".  counts_out (host, [world]) non-null: wait and
 *                             return the offspring per rank (exact-size form).
 *   mp_pf_shard_owned_expand  own draws -> rows of d_rows ([recv_rows + n][dim_state + 1]: the first recv_rows rows are
 *                             where the received surplus arrives, the rest are this rank's own offspring), surplus rows ->
 *                             d_send_out.  d_rows belongs to the handle until the next resample: with dim_state > 1 a kept
 *                             offspring is NOT copied there (its parent is local: the next step gathers the parent's state
 *                             itself), only received rows are read from it — read states through mp_pf_read_state.  capacity > 0: equal splits, d_send_out = [world][capacity] rows, recv_rows =
 *                             world * capacity, pair (r -> s) at [s][j]; capacity == 0: exact sizes, surplus in unit order
 *                             (= destination order), received rows in unit order (= source order).
 *   mp_pf_shard_owned_commit  waits for the plan of the count (not for the expand or the exchange: they are ordered before
 *                             the next step on the filter's stream) and commits as mp_pf_shard_commit_fixed does (the next
 *                             step reads states from d_rows).  With log_total_weight and counts_out both NULL the wait is for
 *                             one polled word of host-mapped memory (the verdict), and in a world of one — nothing to exchange,
 *                             nothing to overflow — there is no wait at all (degenerate weights then surface at the next
 *                             synchronising call, as for mp_pf_resample without a log_total_weight); with either, the call
 *                             waits for the filter's stream.  MP_ERR_CAPACITY: the verdict above (all ranks reach the same
 *                             one); nothing was committed; call again with counts_out to get the offspring per rank (same
 *                             verdict), then repeat the expand with capacity 0 and exact-size buffers, exchange, commit. */
int32_t mp_pf_shard_owned_count(mp_pf* h, int32_t scheme, const uint64_t* d_tiles_all, int32_t world, int32_t rank, uint64_t capacity,
                                uint64_t* counts_out);
int32_t mp_pf_shard_owned_expand(mp_pf* h, int32_t world, int32_t rank, uint64_t capacity, double* d_send_out, double* d_rows, uint64_t recv_rows);
int32_t mp_pf_shard_owned_commit(mp_pf* h, const double* d_rows, double* log_total_weight, uint64_t* counts_out);
/* count + expand of the equal-split form (capacity > 0) as one call — what mp_pf_shard_resample issues.  Same results as the two
 * calls; knowing the send buffer when the count is launched lets a self-drawn resample (lattice schemes, split multinomial) do the
 * table, the offspring counts, the plan, the verdict and the placement of the surplus in ONE kernel launch instead of two. */
int32_t mp_pf_shard_owned_count_expand(mp_pf* h, int32_t scheme, const uint64_t* d_tiles_all, int32_t world, int32_t rank, uint64_t capacity,
                                       double* d_send_out, double* d_rows, uint64_t recv_rows);

/* ---- the sharded resample as ONE call: the library runs the collectives itself ----------------------------------------
 * ParticleSystem::resample (particle_filter.rs:103-116) of a filter sharded over `world` devices, owner-keeps exchange: tiles
 * packed -> all-gather -> mp_pf_shard_owned_count / _expand -> all-to-all of the surplus rows -> mp_pf_shard_owned_commit, with
 * the fallback to exact sizes when a pair of ranks exceeds the equal-split capacity, all buffers owned by the handle and every
 * collective enqueued on the filter's stream.  A C, C++, Rust or Python host calls this (or the _rccl form) once per
 * resample; nothing else of the protocol lives outside the library.
 *
 * mp_transport: the two collectives, as plain C function pointers (device pointers, byte counts, the filter's hipStream_t):
 *   all_gather(ctx, d_send, d_recv, bytes_per_rank, stream)              rank-major result
 *   all_to_all(ctx, d_send, send_off[world], send_bytes[world], d_recv, recv_off[world], recv_bytes[world], world, stream)
 * Both return 0 or an MP_ERR_* code.  mp_transport_rccl fills one over an ncclComm_t: ncclAllGather, and ONE group of
 * ncclSend / ncclRecv per exchange (xGMI is point-to-point).  librccl.so.1 is resolved at first use, by SONAME — a process
 * that has one mapped already (PyTorch) gets that copy — and never for a single-GPU filter.
 * force_collectives != 0: issue them even in a world of one (exercises the transport on one GPU).
 * log_total_weight NULL: asynchronous (one polled word per resample in a world > 1, nothing in a world of one). */
typedef struct mp_transport {
    void* ctx;
    int32_t (*all_gather)(void* ctx, const void* d_send, void* d_recv, uint64_t bytes_per_rank, void* stream);
    int32_t (*all_to_all)(void* ctx, const void* d_send, const uint64_t* send_off, const uint64_t* send_bytes, void* d_recv,
                          const uint64_t* recv_off, const uint64_t* recv_bytes, int32_t world, void* stream);
} mp_transport;
int32_t mp_pf_shard_resample(mp_pf* h, const mp_transport* t, int32_t world, int32_t rank, int32_t scheme, int32_t force_collectives,
                             double* log_total_weight);
int32_t mp_pf_shard_resample_rccl(mp_pf* h, void* nccl_comm, int32_t world, int32_t rank, int32_t scheme, int32_t force_collectives,
                                  double* log_total_weight);
/* log_marginal_likelihood_estimate (:119-121) / fresh ESS of the whole job: level 0 of this shard, all-gather, level 1 */
int32_t mp_pf_shard_query_native(mp_pf* h, const mp_transport* t, int32_t world, int32_t force_collectives, double* log_ml, double* ess);
/* what the last native resamples did: fallbacks to exact sizes so far, surplus rows moved job-wide and offspring per rank of the
 * last resample whose counts reached the host (a synchronous one, or one that fell back), the current equal-split capacity
 * (0: exact sizes).  Any pointer may be NULL. */
int32_t mp_pf_shard_resample_stats(mp_pf* h, uint64_t* fallbacks, uint64_t* exchange_rows, uint64_t* counts_out, uint64_t* capacity);
int32_t mp_transport_rccl(void* nccl_comm, mp_transport* out);
/* a communicator of the library's own: rank 0 makes the 128-byte id and hands it to the other ranks by whatever channel the
 * host has (MPI, a file, torch.distributed's store); every rank then creates its communicator on its device */
int32_t mp_rccl_available(void);   /* 1 if this process can resolve an RCCL; local, no collective: agree on it before any rendezvous */
int32_t mp_rccl_unique_id(void* out128);
int32_t mp_rccl_comm_create(int32_t world, int32_t rank, const void* id128, int32_t device, void** comm_out);
int32_t mp_rccl_comm_destroy(void* comm);
/* device <-> host copy on the filter's stream, waited for: for transports that stage through the host */
int32_t mp_pf_stream_copy(mp_pf* h, void* dst, const void* src, uint64_t bytes, int32_t to_host);

/* ---- profiling hooks (bench.py: HIP-event timing on the stream the kernels run on) ------ */
/* Accumulated GPU time (ms) and launch count of kernel family `which` since the last reset,
 * measured with hipEvents recorded around each launch when timing is enabled. */
enum mp_kernel_family {
    MP_K_PROPAGATE = 0,       /* k_propagate: model kernel + level 0 of the normalisation            */
    MP_K_NORMALIZE_SCAN = 1,  /* k_normalize_tiles: level 0 alone (weights changed without a propagate) */
    MP_K_RESAMPLE_GATHER = 2, /* the gather half of a resample where a kernel of its own does it: k_resample_gather, shard kernels */
    MP_K_BIN_DRAWS = 3,       /* the draw half: k_draw_slots, or the sharded route (k_shard_table / own_bin / route_fused) */
    MP_K_COUNT = 4
};
int32_t mp_pf_set_timing(mp_pf* h, int32_t enabled);
int32_t mp_pf_get_timing(mp_pf* h, int32_t which, double* total_ms, uint64_t* launches);
/* ONE hipEvent pair around a whole region of launches on the handle's stream (begin ... end): the elapsed device time and the
 * k_propagate-family launches enqueued in between.  Unlike the per-launch pairs above this costs the region nothing per step, so
 * elapsed / launches is the un-perturbed mean duration of a launch (including the gaps between launches) — what bench.py's
 * roofline uses.  mp_pf_region_end waits for the stream. */
/* which form of K1 the handle's last step launched (profiles name kernels; a bench line should name the one it measured) */
enum mp_k1_form {
    MP_K1_FORM_TILE = 0,       /* k_propagate<Model, ...>: one workgroup per 2048-slot tile                         */
    MP_K1_FORM_TWO_TILES = 1,  /* k_propagate_mt<Model>: one workgroup per CU over two tiles (mp_pf_k1mt.h)         */
    MP_K1_FORM_DENSE16 = 2     /* k_propagate_dense16: the dense d = 16 model on the matrix cores                   */
};
int32_t mp_pf_last_propagate_form(mp_pf* h, int32_t* out);   /* -1 before the first step */
int32_t mp_pf_region_begin(mp_pf* h);
int32_t mp_pf_region_end(mp_pf* h, double* elapsed_ms, uint64_t* propagate_launches);

/* ---- GenFn::simulate over an Unfold model — DynUnfold::simulate, modppl/src/modeling/dynunfold.rs:22-39 ---- */
/* n independent traces of n_steps kernel calls with EVERY site sampled (the sites that are observations on the filtering
 * path too): states_out[n][n_steps][dim_state], obs_out[n][n_steps][dim_obs] (host).  Trace i uses Philox slot i, step t.
 * MP_MODEL_HMM: MP_ERR_UNSUPPORTED (the reference's HMM leaves simulate unimplemented). */
int32_t mp_unfold_simulate(const mp_model_desc* model, const double* args0, int32_t n_steps, uint64_t n, uint64_t seed, int32_t device,
                           double* states_out, double* obs_out);

/* ---- importance sampling — modppl/src/inference/importance.rs:12-50 -------------------- */
/* importance_resampling(model, args, constraints, num_samples, num_ret_samples):
 * N x generate over all n_steps constraints, logsumexp, log_ml = L - ln N, lnw_i = w_i - L,
 * M categorical draws.  Any of the out pointers may be NULL. */
int32_t mp_importance_resampling(const mp_model_desc* model, const double* args0, const double* obs, int32_t n_steps,
                                 uint64_t num_samples, uint64_t num_ret_samples, uint64_t seed, int32_t device,
                                 double* log_ml_estimate, double* log_normalized_weights, uint64_t* resampled_indices,
                                 double* final_states);

/* importance_sampling(model, args, constraints, num_samples) — importance.rs:12-31 — with what it returns first: ALL N traces.
 * A trace of an Unfold model is its states at every step: trajectories_out[num_samples][n_steps][dim_state] (host; may be NULL),
 * next to the log normalised weights [num_samples] and the log marginal likelihood estimate.  Sample i = Philox slot i. */
int32_t mp_importance_sampling(const mp_model_desc* model, const double* args0, const double* obs, int32_t n_steps, uint64_t num_samples, uint64_t seed,
                               int32_t device, double* log_ml_estimate, double* log_normalized_weights, double* trajectories_out);

/* ---- Metropolis-Hastings — modppl/src/inference/mh.rs:9-75 -------------------------------- */
/* N independent chains advanced in lockstep (chains never communicate: replicas only).  The model
 * is the reference's `hierarchical_model` (modppl/tests/dyngenfns/hierarchical.rs:33-47):
 *   is_linear ~ bernoulli(0.7); coeffs = linear() | quadratic() (a, b[, c] ~ normal(0,1));
 *   y_i ~ normal(a + b x_i [+ c x_i^2], 0.1) observed.
 * Static site ids (the stand-in for trie addresses): */
enum mp_mh_site { MP_SITE_IS_LINEAR = 0, MP_SITE_A = 1, MP_SITE_B = 2, MP_SITE_C = 3, MP_SITE_Y0 = 4 /* "(y, k)" = MP_SITE_Y0 + k */ };
enum mp_mh_model_kind { MP_MH_MODEL_HIERARCHICAL = 1, MP_MH_MODEL_POINTED_2D = 2, MP_MH_MODEL_HIERARCHICAL_FN = 101 /* mp_mh_create_fn */,
                       MP_MH_MODEL_HIERARCHICAL_DATA_FN = 105 /* mp_mh_create_fn: the same model with its "(y, j)" sites DECLARED as data — any number of
                                                                 observations (params = xs[n_obs]; observation j = constraint on site id 4 + j), four
                                                                 sites of trace; see "declared data sites" below */,
                       MP_MH_MODEL_POINTED_FN = 120 /* mp_mh_create_fn: pointed_2d_model as a registered functor (vector-valued sites: latent = slots 1, 2; obs = 3, 4) */ };
enum mp_mh_proposal_kind {
    MP_MH_PROPOSAL_HIERARCHICAL_DRIFT = 1, /* hierarchical_drift_proposal(tr, drift_std): hierarchical.rs:62-70; args = {drift_std} */
    MP_MH_PROPOSAL_HIERARCHICAL_ADD_OR_REMOVE = 2, /* add_or_remove_param_proposal(tr): hierarchical.rs:48-61, the structure-changing move of
                                                    * tests/mh.rs:94 (is_linear re-proposed, coeffs/c added or removed); no args */
    MP_MH_PROPOSAL_POINTED_DRIFT = 3, /* pointed_2d_drift_proposal(tr, noise): simple.rs:36-41; args = the 2x2 noise covariance, row-major */
};
typedef struct mp_mh mp_mh;

/* Initial traces: model.generate(xs, observations) per chain (modppl/tests/mh.rs:91), Philox step 0.
 * constrain_is_linear: -1 = sampled from its prior, 0 / 1 = constrained to false / true.  n_data <= 16. */
int32_t mp_mh_create(int32_t model_kind, const double* xs, const double* ys, int32_t n_data, int32_t constrain_is_linear,
                     uint64_t n_chains, uint64_t seed, int32_t device, void* stream, mp_mh** out);
/* The reference's other MH model, `pointed_2d_model` (modppl/tests/dyngenfns/simple.rs:27-34, driven by tests/mh.rs:50-68):
 *   latent ~ uniform_2d(bounds); obs ~ mvnormal(latent, obs_cov), observed.
 * bounds = {xmin, xmax, ymin, ymax}; obs_cov 2x2 row-major; obs[2].  Chains of this model take MP_MH_PROPOSAL_POINTED_DRIFT
 * (latent' ~ mvnormal(latent, noise): mvnormal.random = L z + mu, L the lower Cholesky factor, mvnormal.rs:24-37);
 * mp_mh_read_state then writes out[n_chains][2] = latent. */
int32_t mp_mh_create_pointed(const double* bounds, const double* obs_cov, const double* obs, uint64_t n_chains, uint64_t seed, int32_t device,
                             void* stream, mp_mh** out);
/* n_iters x `mh(model, trace, proposal, proposal_args)` per chain (mh.rs:9-51).  `accepted` (nullable)
 * receives the total number of accepted moves over all chains and iterations of this call. */
int32_t mp_mh_step(mp_mh* h, int32_t proposal_kind, const double* proposal_args, int32_t n_proposal_args, int32_t n_iters,
                   uint64_t* accepted);
/* n_iters x `regen_mh(model, trace, mask)` per chain (mh.rs:54-75; DynGenFn::regenerate dyngenfn.rs:563-583).
 * mask_sites: subset of {MP_SITE_A, MP_SITE_B, MP_SITE_C} ("coeffs/a" ...).  cycle = 0: every iteration
 * masks all listed sites at once; cycle = 1: iteration k masks only mask_sites[k % n_mask].
 * n_mask = 0 (empty mask): as in the reference the mask is then the trace's whole schema (dyngenfn.rs:571): every site —
 * is_linear, the coefficients and the "(y, i)" sites that were observations — is redrawn, the weight is 0 and every move is
 * accepted; from then on a chain's observations are part of its state (mp_mh_read_observations).
 * Masking is_linear with a non-empty mask is MP_ERR_UNSUPPORTED: the reference panics on the quadratic->linear structure
 * change (dyngenfn.rs:425,526-529). */
int32_t mp_regen_mh_step(mp_mh* h, const int32_t* mask_sites, int32_t n_mask, int32_t cycle, int32_t n_iters, uint64_t* accepted);
/* Chain states -> out[n_chains][4] = {is_linear (0/1), a, b, c}  (c = 0 when is_linear: such a trace has no coeffs/c);
 * pointed model: out[n_chains][2]. */
int32_t mp_mh_read_state(mp_mh* h, double* out);
/* trace.logjp per chain, summed in site order (the reference's value is the trie's running weight: same to ~1e-15 rel). */
int32_t mp_mh_read_logjp(mp_mh* h, double* out);
/* The "(y, i)" choices of every chain's trace -> out[n_chains][n_data]: the data passed to mp_mh_create until an empty-mask
 * regenerate re-simulated them (hierarchical model only). */
int32_t mp_mh_read_observations(mp_mh* h, double* out);
/* ---- mh / regen_mh for REGISTERED generative functions (modppl_amd/csrc/mp_mh_models.h) ----
 * The reference's mh / regen_mh take ANY model and proposal written with `dyngen!` (mh.rs:9-75 over DynGenFn::update /
 * regenerate / propose / assess, dyngenfn.rs:143-273, 321-446, 453-483, 523-583).  The static counterpart: a model or proposal is one
 * functor over the handler of mp_genfn.h (sites = compile-time ids, sub-calls = `call<SITES>`), registered with
 * MP_REGISTER_MH_MODEL / MP_REGISTER_MH_PROPOSAL; the library instantiates the Simulate / Generate / Update / Regenerate
 * kernels for it.  Kind 101 is the hierarchical model again, written that way (params = xs[0 .. n_data); proposal kinds 1, 2 as
 * above): it must and does reproduce the hand-written kernels bit for bit (tests/test_gpu_mh.py).
 * Initial traces: model.generate(args, constraints) per chain, constraints = (site id, value) pairs, Philox step 0.
 * A constraint on a site the model does not visit is the reference's panic: MP_ERR_STATE (also from the step functions when
 * a move reaches such a case; the chains keep whatever the kernel left). */
int32_t mp_mh_create_fn(int32_t model_kind, const double* params, int32_t n_params, const int32_t* constraint_sites, const double* constraint_values,
                        int32_t n_constraints, uint64_t n_chains, uint64_t seed, int32_t device, void* stream, mp_mh** out);
/* Declared data sites (modppl_amd/csrc/mp_genfn.h).  A registered model may declare its observations as DATA instead of giving each a
 * site id of its own: they are then not part of the register-resident trace (no value / log-density / presence bit per observation, no
 * MP_FN_MAX_SITES = 64 cap: `hierarchical_model`'s loop over `xs`, hierarchical.rs:33-47, has no bound), their values live in one array
 * shared by all chains, and the handlers apply `sample_at`'s rules to each of them in the reference's order, recomputing the previous
 * trace's log-density where the reference reads the stored one (same bits).  Across this ABI observation j of such a model is site id
 * n_sites + j, accepted ONLY among the constraints of the creating call (mp_mh_create_fn / mp_fn_generate_create / mp_fn_importance_*),
 * where every one of them must appear.  What cannot be done with a declared data site — because it would turn an observation into a
 * chain's own state — is MP_ERR_UNSUPPORTED: mp_fn_simulate[_create], and the empty mask of mp_regen_mh_step / mp_fn_regenerate (the
 * whole schema, dyngenfn.rs:571, re-simulates observed sites too).  mp_mh_n_sites / mp_mh_read_trace cover the ordinary sites only. */
/* Number of site ids of the model (the row width of mp_mh_read_trace). */
int32_t mp_mh_n_sites(mp_mh* h, int32_t* out);
/* Every chain's trace: values[n_chains][n_sites] (0 where the site is absent) and present[n_chains][W] — 32-bit words, W =
 * (n_sites + 31) / 32: ONE word per chain for models of up to 32 sites (every model the reference's tests need), two up to the
 * 64 sites a model may have; bit k of a chain's words = site k is in the trace.  Every `*present*` array of the mp_fn_* calls
 * below has this layout.  mp_mh_step (proposal kinds registered for the model), mp_regen_mh_step (mask_sites = any site ids; cycle and the
 * empty mask as above), mp_mh_read_logjp, mp_mh_iterations and mp_mh_destroy apply to these handles; mp_mh_read_state and
 * mp_mh_read_observations do not. */
int32_t mp_mh_read_trace(mp_mh* h, double* values, uint32_t* present);

/* ---- The GFI operations ONE AT A TIME, batched over the chains of a registered generative function ---------------------------
 * `GenFn::update / regenerate / propose / assess` are public in the reference (modppl/src/gfi.rs:57-90) and its tests call them
 * directly (modppl/tests/dyngenfn.rs:55-114, 303-388); mp_mh_step / mp_regen_mh_step above fuse them into whole MH moves, the
 * calls below expose the pieces so that a caller can compose inference moves of its own.  One call = the operation on EVERY
 * chain of the handle (each chain's trace is the `trace` argument; args = the model's parameters).
 *   constraints  shared by all chains: sites[n_constraints] + values[n_constraints]; or per chain: chain_values[n_chains][n_sites]
 *                + chain_present[n_chains][W] (the layout of mp_mh_read_trace — what mp_fn_propose and a discard come out as), with
 *                sites = values = NULL, n_constraints = 0
 *   argdiff      ArgDiff::NoChange / ::Unknown (gfi.rs:94-111): under Unknown every revisited choice is re-scored
 *   rng_step     the Philox step of whatever the call draws (free sites); 0 = the next MH iteration's, which the call then
 *                consumes (mp_mh_iterations advances by one).  A caller composing `mh` by hand passes the same step to
 *                propose / update / assess, as mh.rs:9-40 uses one rng for all three.
 *   weights_out  [n_chains] (host), or NULL
 * Errors: MP_ERR_STATE when a chain reaches a case the reference panics on (constraints nobody consumed, dyngenfn.rs:526-529). */
enum mp_argdiff { MP_ARGDIFF_NOCHANGE = 0, MP_ARGDIFF_UNKNOWN = 1 };
/* (new_trace, discard, weight) = model.update(trace, args, argdiff, constraints) — gfi.rs:57-64, dyngenfn.rs:536-560.  Every
 * chain's trace is REPLACED by its new trace; the discard (previous values of the choices that were replaced or dropped) comes back
 * as discard_values_out[n_chains][n_sites] + discard_present_out[n_chains] (either may be NULL). */
int32_t mp_fn_update(mp_mh* h, int32_t argdiff, uint32_t rng_step, const int32_t* sites, const double* values, int32_t n_constraints,
                     const double* chain_values, const uint32_t* chain_present, double* weights_out, double* discard_values_out,
                     uint32_t* discard_present_out);
/* (new_trace, weight) = model.regenerate(trace, args, argdiff, mask) — gfi.rs:66-73, dyngenfn.rs:562-583; mask_sites = the
 * masked sites (n_mask = 0: the trace's whole schema, :571).  Every chain's trace is replaced. */
int32_t mp_fn_regenerate(mp_mh* h, int32_t argdiff, uint32_t rng_step, const int32_t* mask_sites, int32_t n_mask, double* weights_out);
/* weight = f.assess(args, constraints) = f.generate(args, constraints).1 — gfi.rs:85-90.  proposal_kind < 0: f = the model (the
 * chains' traces play no part); otherwise f = that registered proposal applied to each chain's CURRENT trace, as in mh.rs:25-27
 * (`proposal.assess((new_trace, args), discard)` after an update).  Traces are not modified. */
int32_t mp_fn_assess(mp_mh* h, int32_t proposal_kind, const double* proposal_args, int32_t n_proposal_args, uint32_t rng_step, const int32_t* sites,
                     const double* values, int32_t n_constraints, const double* chain_values, const uint32_t* chain_present, double* weights_out);
/* (choices, weight) = proposal.propose((trace, args)) — gfi.rs:78-83, mh.rs:17-19: the registered proposal simulated on each
 * chain's current trace; choice_values_out[n_chains][n_sites] + choice_present_out[n_chains], weights_out = the choices' logjp.
 * Traces are not modified. */
int32_t mp_fn_propose(mp_mh* h, int32_t proposal_kind, const double* proposal_args, int32_t n_proposal_args, uint32_t rng_step,
                      double* choice_values_out, uint32_t* choice_present_out, double* weights_out);

/* (trace, weight) = model.generate(args, constraints) — gfi.rs:53-55, dyngenfn.rs:513-521: constrained sites take their constraint
 * and score into the weight, every other site is drawn from its prior (the internal proposal importance sampling is made of).
 * Every chain's trace is REPLACED by its new trace; constraints shared or per chain as above. */
int32_t mp_fn_generate(mp_mh* h, uint32_t rng_step, const int32_t* sites, const double* values, int32_t n_constraints, const double* chain_values,
                       const uint32_t* chain_present, double* weights_out);
/* trace = model.simulate(args) — gfi.rs:51, dyngenfn.rs:503-511: every site drawn, the observed ones included; every chain's trace
 * is replaced; logjp_out[n_chains] = trace.logjp (nullable). */
int32_t mp_fn_simulate(mp_mh* h, uint32_t rng_step, double* logjp_out);
/* The same two as constructors: n_chains traces of a registered function from nothing, Philox step 0.  mp_fn_generate_create with
 * weights_out = NULL is mp_mh_create_fn. */
int32_t mp_fn_generate_create(int32_t model_kind, const double* params, int32_t n_params, const int32_t* constraint_sites, const double* constraint_values,
                              int32_t n_constraints, uint64_t n_chains, uint64_t seed, int32_t device, void* stream, double* weights_out, mp_mh** out);
int32_t mp_fn_simulate_create(int32_t model_kind, const double* params, int32_t n_params, uint64_t n_chains, uint64_t seed, int32_t device, void* stream,
                              double* logjp_out, mp_mh** out);

/* ---- importance sampling over a REGISTERED generative function — modppl/src/inference/importance.rs:12-50 -------------------
 * `importance_sampling(model, model_args, constraints, num_samples)` is generic over `impl GenFn`; the reference's third test runs it
 * on `hierarchical_model` (modppl/tests/importance.rs:89-139: 11 observations, 10 000 samples, "works with ~1,000,000 particles").
 * mp_importance_sampling above takes Unfold-style model descriptors only; these take any registered function:
 *   num_samples x model.generate(args, constraints) (Philox slot = sample, step 0)             importance.rs:18-20
 *   log_total_weight = logsumexp(weights); log_ml_estimate = log_total_weight - ln N;  lnw_i = w_i - log_total_weight   :21-25
 *   importance_resampling: num_ret_samples categorical draws over exp(lnw) (one sequential stream, Philox domain IS)      :44-47
 * The normalisation is the filters' canonical one, so log_ml and the indices are bit-equal to the checker's.
 * log_normalized_weights[num_samples], resampled_indices[num_ret_samples] (host; the former nullable).  traces_out (nullable): the
 * num_samples traces as a handle — mp_mh_read_trace returns them, and every mp_mh_* / mp_fn_* call applies (sample i = chain i);
 * the caller destroys it. */
int32_t mp_fn_importance_sampling(int32_t model_kind, const double* params, int32_t n_params, const int32_t* constraint_sites, const double* constraint_values,
                                  int32_t n_constraints, uint64_t num_samples, uint64_t seed, int32_t device, double* log_ml_estimate,
                                  double* log_normalized_weights, mp_mh** traces_out);
int32_t mp_fn_importance_resampling(int32_t model_kind, const double* params, int32_t n_params, const int32_t* constraint_sites, const double* constraint_values,
                                    int32_t n_constraints, uint64_t num_samples, uint64_t num_ret_samples, uint64_t seed, int32_t device,
                                    double* log_ml_estimate, double* log_normalized_weights, uint64_t* resampled_indices, mp_mh** traces_out);

/* MH iterations applied so far (the Philox step of the next iteration is this + 1). */
int32_t mp_mh_iterations(mp_mh* h, uint64_t* out);
int32_t mp_mh_destroy(mp_mh* h);

#ifdef __cplusplus
}
#endif
#endif /* MODPPL_HIP_H */
