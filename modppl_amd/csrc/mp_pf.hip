// mp_pf.hip — gfx950 kernels and C-ABI implementation of the particle-filter hot path
// (include/modppl_hip.h).  Written for MI355X only: 64-lane wavefronts, LDS-staged reductions,
// particle-major state in HBM.  Device code: mp_pf_kernels.h, mp_pf_shard_kernels.h; this file: the host side.
//
// Data layout in HBM (per handle, n = local particles, d = dim_state, tiles of 2048 rows):
//   x[2][n][d]  f64   particle states (slot order, particle-major)
//   logw[n]     f64   log-weights                              (particle_filter.rs:15)
//   cx[n]       {u64,f64}  resampling table rows: tile-local inclusive prefix of the fixed-point weights + x[0]
//               (+ cx_alt: a k_propagate that looks up the last resample's draws in cx writes the new table there)
//   guide[nt][2048] u16  per-tile bucketed inverse CDF (first row of each bucket)
//   tile_m/W/W2[nt]      per-tile max log-weight and fixed-point totals (level 0 of the normalisation)
//   dfr_lt/dfr_row[n]  u64 / u32  draws of the last resample not looked up yet: tile-local target and start row per output slot
//   parent[n]   u32   parents of the last resample             (particle_filter.rs:20)
//   scal        mp_dev_scalars   log_ml_estimate, last log total weight, ESS ... (device-resident so
//                                that a whole filter run needs no host round trip)
//
//   tab_incl/tab_ratio/tab_head  the job's tile table (level 1), built once per normalisation by the last level-0 workgroup
//
// Kernels of one SMC time step (results never depend on dispatch order or XCD placement: all cross-workgroup sums are
// integer; the one hand-off inside a launch is the atomic ticket that lets the LAST workgroup of the level-0 launch build
// the tile table from every workgroup's tile scalars — whichever workgroup that is):
//   K1  k_propagate       ParticleSystem::init_step/step: one workgroup per 2048-row tile runs the model kernel in
//                         Generate mode (logw (+)= weight) and, in the same pass, level 0 of normalize_weights
//                         (:27-35): tile max, exp, 51-bit fixed point, tile-local scan, rows, guide; level 1 (the tile
//                         table) by the last workgroup
//   K3a k_draw_slots      multinomial_resampling (:37-41), the draws: Philox (two adjacent draws per block), target, tile +
//                         guide lookup -> {target, start row} per output slot
//       the row lookups + the clone loop of resample (:109-114) run inside the NEXT k_propagate (each workgroup looks up
//       the parents of its own slots, under its arithmetic), or in k_resolve_slots when the host asks for states / parents first
//   (K3 k_resample_gather single-kernel form: importance_resampling's M draws, systematic / stratified resampling)
// The normalisation spec (hierarchical fixed point) is stated in DESIGN.md §4 and restated on the CPU in
// oracle/src/inference.hpp.
#include <hip/hip_runtime.h>
#include <cstring>

#include <algorithm>
#include <cstdlib>
#include <memory>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/modppl_hip.h"
#include "mp_diag.h"
#include "mp_linalg.h"
// models register themselves (mp_models.h, MP_REGISTER_UNFOLD_MODEL): here a registration creates the device factory
template <class M>
int mp_register_model_hip(int kind, bool (*parse)(const mp_model_desc&, M&, std::string&));
#define MP_MODEL_REGISTRAR mp_register_model_hip
#include "mp_models.h"

typedef unsigned long long u64;

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int32_t mp_fail(int32_t code, const std::string& msg) {
    g_err = msg;
    return code;
}
int32_t mp_set_error(int32_t code, const std::string& msg) { return mp_fail(code, msg); }  // shared with mp_mh.hip
#define HIPCK(call)                                                                                  \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return mp_fail(MP_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));           \
    } while (0)

#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes for mp_shard_native.h (resolved with dlsym at first use: no link-time dependency)
#include "mp_pf_kernels.h"
#include "mp_pf_k1mt.h"
#include "mp_pf_shard_kernels.h"

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int ceil_log2_u64(u64 n) {
    int b = 0;
    while ((1ull << b) < n) ++b;
    return b;
}
static size_t table_lds(int nt, int threads) {  // s_incl[nt] + s_W[nt] + s_red + s_wtot
    return sizeof(u64) * 2 * (size_t)nt + (sizeof(double) + sizeof(u64)) * (size_t)(threads / 64);
}

// MP_K1_THREADS (A/B measurements: other lane shapes of k_propagate<mp_lgssm1>), read ONCE per process: what a model may do
// (can_draw) and the launch shape come from this one value — read at two different times they could disagree, and a kernel
// that cannot draw would be handed draws to make
static int k1_threads_override() {
    static const int v = [] { const char* e = mp_diag_env("MP_K1_THREADS"); return e ? atoi(e) : 0; }();
    return v;
}

struct PropagateArgs {
    u64 n, slot_offset;
    uint32_t k0, k1;
    long long t;
    const double* x_in;
    double* x_out;
    double* logw;
    mp_obs obs;
    mp_state0 s0;
    int overwrite;
    int grid;
    hipStream_t stream;
    const uint32_t* dfr_row;  // draws of the last resample not looked up yet: start rows / targets per slot, and the table they
    const u64* dfr_lt;        // refer to (cx_old); `cx` below is then the OTHER table buffer
    const mp_cx* cx_old;
    const double* inv_rows;   // exchange rows of the last sharded resample (slot i reads row inv[i])
    mp_cx* cx;
    unsigned short* guide;
    double* tile_m;
    u64* tile_W;
    u64* tile_W2;
    double* tile_m_new;       // (drawing launches of local_table handles) the tile scalars this launch WRITES: the other buffer pair (= what `tail` names)
    u64* tile_W_new;
    u64* tile_W2_new;
    const uint32_t* inv;
    mp_k1_draw drw_v;         // ... what it reads for that (by value: kernel arguments)
    int drw;                  // non-zero: this launch also MAKES the previous resample's draws (resample counter rc), into dfr_row / dfr_lt
    uint32_t rc;
    size_t dyn_lds;           // LDS for that phase's copy of the tile table
    const mp_k1_tail* tail;   // device copy of {cx, guide, tile_*, tab} (k_propagate reads them there)
    mp_k1_aux aux;
    bool walk_bisect;         // the draws this launch looks up are a lattice's (consecutive slots, consecutive targets): wide-state kernels then
                              // finish long walks by bisection (the WALKB instantiation)
    int mt_grid;              // > 0: a drawing launch may run as k_propagate_mt with this many workgroups (one per CU, several tiles each)
    int mt_flags;             // MP_MT_SKIP_* (mp_pf_k1mt.h)
    unsigned int* peek_ticket;          // MP_MT_PEEK: the launch's last workgroup hands L / ESS of the NEW tile scalars to the host (mp_k1mt)
    mp_host_mirror* peek_mirror;
    unsigned long long peek_seq;
};
struct ModelOps {
    int dim_state = 0, dim_obs = 0;
    void* owned_device_mem = nullptr;   // model constants that do not fit kernel arguments (freed with the model)
    int max_normals = 0;
    bool can_draw = false;   // its k_propagate can make the previous resample's draws itself (lanes of two adjacent slots: the 1024-thread launch shape)
    virtual ~ModelOps() { if (owned_device_mem) (void)hipFree(owned_device_mem); }
    virtual int propagate(const PropagateArgs& a) const = 0;   // -> MP_K1_FORM_* of the kernel it launched
    virtual int n_normals(long long t) const = 0;
    virtual void simulate(u64 n, uint32_t k0, uint32_t k1, int n_steps, const mp_state0& s0, double* states, double* obs, hipStream_t st) const = 0;
};
template <class Model>
struct ModelOpsT : ModelOps {
    Model model;
    explicit ModelOpsT(const Model& m) : model(m) {
        dim_state = Model::DIM_STATE;
        dim_obs = Model::DIM_OBS;
        max_normals = Model::MAX_NORMALS;
        // (the conditions of CAN_DRAW in k_propagate, for the launch configuration `propagate` below picks)
        can_draw = Model::MAX_NORMALS <= 4 && Model::DIM_STATE <= 4 &&   // = the 1024-thread, two-slot-lane launch shape of `propagate` below
                   !std::is_same<Model, mp_lgssm_dense<16>>::value && k1_threads_override() == 0;   // (that override launches other lane shapes)
        static_assert(Model::DIM_STATE <= MP_MAX_STATE && Model::DIM_OBS <= MP_MAX_OBS, "model too wide for mp_obs / mp_state0");
        static_assert(TILE_ITEMS % k1_items<Model>() == 0, "rounds of k_propagate");
    }
    void simulate(u64 n, uint32_t k0, uint32_t k1, int n_steps, const mp_state0& s0, double* states, double* obs, hipStream_t st) const override {
        hipLaunchKernelGGL(k_simulate<Model>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, model, n, k0, k1, n_steps, s0, states, obs);
    }
    int propagate(const PropagateArgs& a) const override {
        // light kernels (few registers) run 1024 threads x 2 particles per tile: twice the waves in flight for the same 2048-slot tile
        constexpr int THREADS = (Model::MAX_NORMALS <= 4 && Model::DIM_STATE <= 4) ? 1024 : TILE_THREADS;
        const int k1t = k1_threads_override();   // A/B measurements
        if constexpr (std::is_same<Model, mp_lgssm1>::value) {
            if (k1t == 256) {
                hipLaunchKernelGGL((k_propagate<Model, 256>), dim3(a.grid), dim3(256), a.dyn_lds, a.stream, a.drw_v.tile_m_old, a.drw_v.tile_W_old, a.drw_v.tile_W2_old,
                                   a.drw_v.nt, a.drw, model, a.n, a.slot_offset, a.k0, a.k1, a.t,
                                   a.x_in, a.x_out, a.logw, a.obs, a.s0, a.overwrite, a.dfr_row, a.inv_rows, a.cx_old, a.tail,
                                   a.inv, a.dfr_lt, a.aux, a.drw_v, a.rc);
                return MP_K1_FORM_TILE;
            }
            if (k1t == 512) {
                hipLaunchKernelGGL((k_propagate<Model, 512>), dim3(a.grid), dim3(512), a.dyn_lds, a.stream, a.drw_v.tile_m_old, a.drw_v.tile_W_old, a.drw_v.tile_W2_old,
                                   a.drw_v.nt, a.drw, model, a.n, a.slot_offset, a.k0, a.k1, a.t,
                                   a.x_in, a.x_out, a.logw, a.obs, a.s0, a.overwrite, a.dfr_row, a.inv_rows, a.cx_old, a.tail,
                                   a.inv, a.dfr_lt, a.aux, a.drw_v, a.rc);
                return MP_K1_FORM_TILE;
            }
        }
        if constexpr (std::is_same<Model, mp_lgssm_dense<16>>::value) {
            // the dense transition's products on the matrix cores (k_propagate_dense16); MP_DENSE_MFMA=0 keeps the scalar
            // interpretation of the same functor (same bits)
            static const bool mfma = [] { const char* e = mp_diag_env("MP_DENSE_MFMA"); return !(e && e[0] == '0'); }();
            if (mfma) {
                hipLaunchKernelGGL(a.walk_bisect ? k_propagate_dense16<true> : k_propagate_dense16<false>, dim3(a.grid), dim3(DENSE_THREADS), 0, a.stream, model, a.n, a.slot_offset, a.k0, a.k1, a.t, a.x_in,
                                   a.x_out, a.logw, a.obs, a.overwrite, a.dfr_row, a.dfr_lt, a.cx_old, a.cx, a.guide, a.tile_m, a.tile_W, a.tile_W2, a.aux,
                                   a.inv, a.inv_rows);
                return MP_K1_FORM_DENSE16;
            }
        }
        if constexpr (THREADS == 1024 && Model::DIM_STATE == 1 && 2 * Model::MAX_NORMALS <= 4) {
            // one workgroup per CU walking several tiles (mp_pf_k1mt.h): drawing launches of unsharded filters with at most one table
            // entry per thread
            if (a.drw == 1 && a.mt_grid > 0 && a.drw_v.nt <= 1024 && a.drw_v.nt >= 2 * a.mt_grid && a.cx_old && !a.inv && !a.inv_rows) {
                // (multinomial draws, jobs of at least two tiles per CU: below that one workgroup per tile spreads over more CUs —
                // 2^17 particles 23 against 32 us —, and a lattice's sorted lookups leave nothing to hide: 30.0 against 31.5 us;
                // 2^20: 38.6 against 38.9, 2^21: 80.9 against 85.9, profiles/r04/k1_scaling.txt)
                mp_k1mt m;
                m.n = a.n; m.slot_offset = a.slot_offset; m.n_global = a.drw_v.n_global; m.t = a.t; m.k0 = a.k0; m.k1 = a.k1; m.rc = a.rc;
                m.S = a.drw_v.S; m.flags = a.mt_flags; m.logw = a.logw; m.cx_old = a.cx_old; m.guide_old = a.drw_v.guide_old;
                m.cx_new = a.cx; m.guide_new = a.guide; m.tm_new = a.tile_m_new; m.tW_new = a.tile_W_new; m.tW2_new = a.tile_W2_new;
                m.parent = a.drw_v.parent; m.scal = a.drw_v.scal;
                m.peek_ticket = a.peek_ticket; m.mirror = a.peek_mirror; m.peek_seq = a.peek_seq;
                mp_obs_n<Model::DIM_OBS> ob;
                for (int j = 0; j < Model::DIM_OBS; ++j) ob.v[j] = a.obs.v[j];
                const int grid = (a.drw_v.nt + 1) / 2;   // two tiles per workgroup: b and b + grid
                if (a.walk_bisect)
                    hipLaunchKernelGGL((k_propagate_mt<Model, true>), dim3(grid), dim3(1024), a.dyn_lds, a.stream, a.drw_v.tile_m_old, a.drw_v.tile_W_old,
                                       a.drw_v.tile_W2_old, a.drw_v.nt, a.drw, model, m, ob);
                else
                    hipLaunchKernelGGL((k_propagate_mt<Model, false>), dim3(grid), dim3(1024), a.dyn_lds, a.stream, a.drw_v.tile_m_old, a.drw_v.tile_W_old,
                                       a.drw_v.tile_W2_old, a.drw_v.nt, a.drw, model, m, ob);
                return MP_K1_FORM_TWO_TILES;
            }
        }
        // the form of k_propagate: TAB2 (more tiles than threads in a drawing launch), LAT (the pending draws are a lattice's), SHD (a
        // sharded filter's self-drawn resample); WALKB (long row walks finish by bisection) for the forms that have such an instantiation
        const int sch = a.drw >> 1;
        const bool shd = a.drw && a.drw_v.shd_range;
        const bool lat = a.drw && (sch == 1 || sch == 2);
        const bool tab2 = a.drw && !shd && a.drw_v.nt > THREADS;
        if constexpr (THREADS == 1024) {
            if (shd) return lat ? (a.walk_bisect ? k1<false, true, true, true>(a) : k1<false, true, true, false>(a))
                                : (a.walk_bisect ? k1<false, false, true, true>(a) : k1<false, false, true, false>(a));
            if (tab2) return lat ? k1<true, true, false, false>(a) : k1<true, false, false, false>(a);
            if (lat) return a.walk_bisect ? k1<false, true, false, true>(a) : k1<false, true, false, false>(a);
        }
        return a.walk_bisect ? k1<false, false, false, true>(a) : k1<false, false, false, false>(a);
    }
    template <bool TAB2, bool LAT, bool SHD, bool WALKB>
    int k1(const PropagateArgs& a) const {
        constexpr int THREADS = (Model::MAX_NORMALS <= 4 && Model::DIM_STATE <= 4) ? 1024 : TILE_THREADS;
        hipLaunchKernelGGL((k_propagate<Model, THREADS, TAB2, LAT, SHD, WALKB>), dim3(a.grid), dim3(THREADS), a.dyn_lds, a.stream, a.drw_v.tile_m_old,
                           a.drw_v.tile_W_old, a.drw_v.tile_W2_old, a.drw_v.nt, a.drw, model, a.n, a.slot_offset, a.k0, a.k1, a.t,
                           a.x_in, a.x_out, a.logw, a.obs, a.s0, a.overwrite, a.dfr_row, a.inv_rows, a.cx_old, a.tail,
                           a.inv, a.dfr_lt, a.aux, a.drw_v, a.rc);
        return MP_K1_FORM_TILE;
    }
    int n_normals(long long t) const override { return model.n_normals(t); }
};

// kind -> factory of the registered models
#include <atomic>
#include <functional>
#include <map>
static std::map<int, std::function<int32_t(const mp_model_desc*, std::unique_ptr<ModelOps>&)>>& model_registry() {
    static std::map<int, std::function<int32_t(const mp_model_desc*, std::unique_ptr<ModelOps>&)>> r;
    return r;
}
template <class M>
int mp_register_model_hip(int kind, bool (*parse)(const mp_model_desc&, M&, std::string&)) {
    model_registry()[kind] = [parse](const mp_model_desc* m, std::unique_ptr<ModelOps>& out) -> int32_t {
        M k{};
        std::string err;
        if (!parse(*m, k, err)) return mp_fail(MP_ERR_INVALID_ARG, err);
        out.reset(new ModelOpsT<M>(k));   // instantiates k_propagate / k_simulate / the resample kernels for M
        return MP_OK;
    };
    return kind;
}

static int32_t make_model(const mp_model_desc* m, std::unique_ptr<ModelOps>& out) {
    if (!m) return mp_fail(MP_ERR_INVALID_ARG, "model descriptor is null");
    switch (m->kind) {
    case MP_MODEL_LGSSM1: {
        if (m->n_params != 5 || !m->params) return mp_fail(MP_ERR_INVALID_ARG, "MP_MODEL_LGSSM1 takes 5 params {mu0,sig0,a,sig_x,sig_y}");
        if (m->dim_state != 1 || m->dim_obs != 1) return mp_fail(MP_ERR_INVALID_ARG, "MP_MODEL_LGSSM1: dim_state = dim_obs = 1");
        mp_lgssm1 k{m->params[0], m->params[1], m->params[2], m->params[3], m->params[4], 0.};
        if (!(k.sig0 > 0.) || !(k.sig_x > 0.) || !(k.sig_y > 0.)) return mp_fail(MP_ERR_INVALID_ARG, "MP_MODEL_LGSSM1: standard deviations must be > 0");
        k.ln_sig_y = mp_log(k.sig_y);
        out.reset(new ModelOpsT<mp_lgssm1>(k));
        return MP_OK;
    }
    case MP_MODEL_SPIRAL: {
        if (m->dim_state != 2 || m->dim_obs != 2) return mp_fail(MP_ERR_INVALID_ARG, "MP_MODEL_SPIRAL: dim_state = dim_obs = 2");
        mp_spiral k;
        const std::vector<double> cov = {0.001, 0., 0., 0.001};  // unfold.rs:29
        std::vector<double> inv;
        if (!mp_host_inverse(cov, 2, inv)) return mp_fail(MP_ERR_INVALID_ARG, "covariance not invertible");
        for (int i = 0; i < 4; ++i) k.cov_inv[i] = inv[i];
        k.ln_det = mp_log(mp_host_det(cov, 2));
        std::vector<double> L;
        if (!mp_host_cholesky(cov, 2, L)) return mp_fail(MP_ERR_INVALID_ARG, "covariance not positive definite");
        for (int i = 0; i < 4; ++i) { k.chol[i] = L[i]; k.cov[i] = cov[i]; }
        out.reset(new ModelOpsT<mp_spiral>(k));
        return MP_OK;
    }
    case MP_MODEL_HMM: {
        if (m->n_params < 2 || !m->params) return mp_fail(MP_ERR_INVALID_ARG, "MP_MODEL_HMM: params = {S, O, prior[S], emission[O][S], transition[S][S]}");
        const int S = (int)m->params[0], O = (int)m->params[1];
        if (S < 1 || O < 1 || S > MP_HMM_MAX || O > MP_HMM_MAX) return mp_fail(MP_ERR_UNSUPPORTED, "MP_MODEL_HMM: 1..8 states / observations");
        if (m->n_params != 2 + S + O * S + S * S) return mp_fail(MP_ERR_INVALID_ARG, "MP_MODEL_HMM: params length mismatch");
        if (m->dim_state != 1 || m->dim_obs != 1) return mp_fail(MP_ERR_INVALID_ARG, "MP_MODEL_HMM: dim_state = dim_obs = 1");
        mp_hmm k{};
        k.n_states = S; k.n_obs = O;
        const double* prior = m->params + 2;
        const double* emis = prior + S;
        const double* trans = emis + O * S;
        auto sums_to_one = [](const double* p, int n, int stride) {  // categorical.rs:13,23: assert |sum - 1| <= 1e-8
            double s_ = 0.;
            for (int i = 0; i < n; ++i) s_ += p[i * stride];
            return std::fabs(s_ - 1.0) <= 1e-8;
        };
        if (!sums_to_one(prior, S, 1)) return mp_fail(MP_ERR_INVALID_ARG, "MP_MODEL_HMM: prior does not sum to 1 (eps 1e-8)");
        for (int s_ = 0; s_ < S; ++s_) {
            if (!sums_to_one(emis + s_, O, S)) return mp_fail(MP_ERR_INVALID_ARG, "MP_MODEL_HMM: emission column does not sum to 1");
            if (!sums_to_one(trans + s_, S, S)) return mp_fail(MP_ERR_INVALID_ARG, "MP_MODEL_HMM: transition column does not sum to 1");
        }
        for (int s_ = 0; s_ < S; ++s_) k.prior[s_] = prior[s_];
        for (int s_ = 0; s_ < S; ++s_) {
            for (int o = 0; o < O; ++o) k.emission_col[s_][o] = emis[o * S + s_];
            for (int s2 = 0; s2 < S; ++s2) k.transition_col[s_][s2] = trans[s2 * S + s_];
        }
        out.reset(new ModelOpsT<mp_hmm>(k));
        return MP_OK;
    }
    case MP_MODEL_BEARINGS: {
        if (m->n_params != 6 || !m->params) return mp_fail(MP_ERR_INVALID_ARG, "MP_MODEL_BEARINGS takes 6 params {p0x,p0y,sig_p0,sig_v0,sig_a,sig_theta}");
        if (m->dim_state != 4 || m->dim_obs != 1) return mp_fail(MP_ERR_INVALID_ARG, "MP_MODEL_BEARINGS: dim_state = 4, dim_obs = 1");
        mp_bearings k{m->params[0], m->params[1], m->params[2], m->params[3], m->params[4], m->params[5], 0.};
        if (!(k.sig_p0 > 0.) || !(k.sig_v0 > 0.) || !(k.sig_a > 0.) || !(k.sig_theta > 0.)) return mp_fail(MP_ERR_INVALID_ARG, "MP_MODEL_BEARINGS: standard deviations must be > 0");
        k.ln_sig_theta = mp_log(k.sig_theta);
        out.reset(new ModelOpsT<mp_bearings>(k));
        return MP_OK;
    }
    case MP_MODEL_LGSSM_BAND: {
        if (m->n_params != 6 || !m->params) return mp_fail(MP_ERR_INVALID_ARG, "MP_MODEL_LGSSM_BAND takes 6 params {D,a,band,sig0,sig_x,sig_y}");
        const int D = (int)m->params[0];
        if (m->dim_state != D || m->dim_obs != D) return mp_fail(MP_ERR_INVALID_ARG, "MP_MODEL_LGSSM_BAND: dim_state = dim_obs = D");
        if (!(m->params[3] > 0.) || !(m->params[4] > 0.) || !(m->params[5] > 0.)) return mp_fail(MP_ERR_INVALID_ARG, "MP_MODEL_LGSSM_BAND: standard deviations must be > 0");
        const double ln_sy = mp_log(m->params[5]);
        if (D == 16) { out.reset(new ModelOpsT<mp_lgssm_band<16>>(mp_lgssm_band<16>{m->params[1], m->params[2], m->params[3], m->params[4], m->params[5], ln_sy})); return MP_OK; }
        if (D == 4) { out.reset(new ModelOpsT<mp_lgssm_band<4>>(mp_lgssm_band<4>{m->params[1], m->params[2], m->params[3], m->params[4], m->params[5], ln_sy})); return MP_OK; }
        if (D == 2) { out.reset(new ModelOpsT<mp_lgssm_band<2>>(mp_lgssm_band<2>{m->params[1], m->params[2], m->params[3], m->params[4], m->params[5], ln_sy})); return MP_OK; }
        return mp_fail(MP_ERR_UNSUPPORTED, "MP_MODEL_LGSSM_BAND: D in {2, 4, 16} is compiled in");
    }
    case MP_MODEL_POINTED_2D: {
        if (m->n_params != 8 || !m->params) return mp_fail(MP_ERR_INVALID_ARG, "MP_MODEL_POINTED_2D takes 8 params {xmin,xmax,ymin,ymax, cov row-major}");
        if (m->dim_state != 2 || m->dim_obs != 2) return mp_fail(MP_ERR_INVALID_ARG, "MP_MODEL_POINTED_2D: dim_state = dim_obs = 2");
        if (!(m->params[1] > m->params[0]) || !(m->params[3] > m->params[2])) return mp_fail(MP_ERR_INVALID_ARG, "MP_MODEL_POINTED_2D: xmax > xmin and ymax > ymin");
        mp_pointed2d k{};
        k.xmin = m->params[0]; k.xmax = m->params[1]; k.ymin = m->params[2]; k.ymax = m->params[3];
        const std::vector<double> cov(m->params + 4, m->params + 8);
        std::vector<double> inv;
        const double det = mp_host_det(cov, 2);
        if (!(det > 0.) || !mp_host_inverse(cov, 2, inv)) return mp_fail(MP_ERR_INVALID_ARG, "MP_MODEL_POINTED_2D: covariance must be invertible with a positive determinant");
        for (int i = 0; i < 4; ++i) k.cov_inv[i] = inv[i];
        k.ln_det = mp_log(det);
        std::vector<double> L;
        if (!mp_host_cholesky(cov, 2, L)) return mp_fail(MP_ERR_UNSUPPORTED, "MP_MODEL_POINTED_2D: covariance without a Cholesky factor (the reference's eigen fallback is not built)");
        for (int i = 0; i < 4; ++i) k.chol[i] = L[i];
        out.reset(new ModelOpsT<mp_pointed2d>(k));
        return MP_OK;
    }
    case MP_MODEL_LINE: {
        if (m->n_params != 11 || !m->params || m->dim_obs != 11 || m->dim_state != 2)
            return mp_fail(MP_ERR_UNSUPPORTED, "MP_MODEL_LINE: the 11-point design of tests/importance.rs:62 is compiled in (params = xs[11], dim_state = 2, dim_obs = 11)");
        mp_line<11> k{};
        for (int i = 0; i < 11; ++i) k.xs[i] = m->params[i];
        k.ln_noise = mp_log(0.1);
        out.reset(new ModelOpsT<mp_line<11>>(k));
        return MP_OK;
    }
    case MP_MODEL_LGSSM_DENSE: {
        if (!m->params || m->n_params < 2) return mp_fail(MP_ERR_INVALID_ARG, "MP_MODEL_LGSSM_DENSE: params = {D, sig0, A[D*D], Q[D*D], R[D*D]}");
        const int D = (int)m->params[0];
        if (D != 16) return mp_fail(MP_ERR_UNSUPPORTED, "MP_MODEL_LGSSM_DENSE: D = 16 is compiled in (one matrix-core tile)");
        if (m->n_params != 2 + 3 * D * D || m->dim_state != D || m->dim_obs != D)
            return mp_fail(MP_ERR_INVALID_ARG, "MP_MODEL_LGSSM_DENSE: n_params = 2 + 3 D^2, dim_state = dim_obs = D");
        const double sig0 = m->params[1];
        if (!(sig0 > 0.)) return mp_fail(MP_ERR_INVALID_ARG, "MP_MODEL_LGSSM_DENSE: sig0 must be > 0");
        const double* q = m->params + 2;
        const std::vector<double> A(q, q + D * D), Q(q + D * D, q + 2 * D * D), R(q + 2 * D * D, q + 3 * D * D);
        // the per-call nalgebra work of mvnormal.rs:17-18,27-33, once: transform of Q / sig0^2 I / R, inverse and ln det of R
        std::vector<double> cov0((size_t)D * D, 0.), TQ, T0, TR, Rinv;
        for (int i = 0; i < D; ++i) cov0[i * D + i] = sig0 * sig0;
        mp_host_mvnormal_transform(Q, D, TQ);
        mp_host_mvnormal_transform(cov0, D, T0);
        mp_host_mvnormal_transform(R, D, TR);
        const double detR = mp_host_det(R, D);
        if (!mp_host_inverse(R, D, Rinv)) return mp_fail(MP_ERR_INVALID_ARG, "MP_MODEL_LGSSM_DENSE: R is not invertible (try_inverse().unwrap() panics, mvnormal.rs:18)");
        std::vector<double> mats;
        mats.insert(mats.end(), A.begin(), A.end());
        mats.insert(mats.end(), TQ.begin(), TQ.end());
        mats.insert(mats.end(), T0.begin(), T0.end());
        mats.insert(mats.end(), Rinv.begin(), Rinv.end());
        mats.insert(mats.end(), TR.begin(), TR.end());
        double* d_mats = nullptr;
        HIPCK(hipMalloc(&d_mats, sizeof(double) * mats.size()));
        if (hipMemcpy(d_mats, mats.data(), sizeof(double) * mats.size(), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(d_mats);
            return mp_fail(MP_ERR_HIP, "MP_MODEL_LGSSM_DENSE: copying the model matrices failed");
        }
        mp_lgssm_dense<16> k{d_mats, mp_log(detR)};
        auto* ops = new ModelOpsT<mp_lgssm_dense<16>>(k);
        ops->owned_device_mem = d_mats;
        out.reset(ops);
        return MP_OK;
    }
    default: {
        auto it = model_registry().find(m->kind);   // models added through MP_REGISTER_UNFOLD_MODEL (mp_models_extra.h)
        if (it != model_registry().end()) return it->second(m, out);
        return mp_fail(MP_ERR_UNSUPPORTED, "model kind " + std::to_string(m->kind) + " is not compiled into this library");
    }
    }
}

struct TimedLaunch {
    hipEvent_t start, stop;
    int family;
};

struct mp_shard_native_state;   // mp_shard_native.h
struct mp_pf {
    std::unique_ptr<ModelOps> ops;
    mp_shard_native_state* native = nullptr;   // buffers and bookkeeping of mp_pf_shard_resample
    u64 n = 0, n_global = 0, slot_offset = 0, seed = 0;
    uint32_t flags = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int S = 0;
    int nt = 0;  // tiles of this handle (== K1 grid)
    int k3_grid = 0;
    // device buffers
    double* x[2] = {nullptr, nullptr};
    int cur = 0;
    double* logw = nullptr;
    mp_cx* cx = nullptr;
    unsigned short* guide = nullptr;
    uint32_t* parent = nullptr;
    double* tile_m = nullptr;
    u64* tile_W = nullptr;
    u64* tile_W2 = nullptr;
    mp_dev_scalars* scal = nullptr;
    double* aos = nullptr;             // staging for read_state
    mp_dev_scalars* h_scal = nullptr;  // pinned
    int* h_flag = nullptr;             // host-mapped: the device-side sticky error (mp_dev_scalars::host_flag points here)
    mp_host_mirror* h_mirror = nullptr;   // host-mapped: L / ESS / log-ML of every fold, and k_peek_level1's answers (mp_pf_kernels.h)
    mp_host_mirror* d_mirror = nullptr;   // its device address
    unsigned long long peek_seq = 0;      // k_peek_level1 launches (and MP_MT_PEEK tails) so far
    bool sync_loop = false;               // the last resample was synchronous (`L = resample()`): the next k_propagate_mt launch peeks for the one to come
    bool peek_valid = false;              // ... and did: peek number peek_seq belongs to the current tile scalars
    int use_mirror = 1;                   // MP_HOST_MIRROR=0: synchronous calls copy mp_dev_scalars back as before (A/B measurements)
    // the draws of a multinomial resample, per output slot (k_draw_slots)
    u64* dfr_lt = nullptr;              // [n] tile-local target
    uint32_t* dfr_row = nullptr;        // [n] table row where the forward scan starts
    int nchunks = 0;
    int use_deferred = 1;               // MP_DEFERRED_LOOKUPS=0 selects the single-kernel resampler (A/B measurements)
    bool sh_parents_lazy = false;       // ... or in column D of the exchange rows sh_rows / sh_req_slot
    // level-1 table built by the last workgroup of the level-0 launch (mp_tab)
    mp_k1_tail* k1_tail = nullptr;      // device copy of what k_propagate's last phase needs (update_k1_tail)
    mp_k1_tail* k1_tail_alt = nullptr;  // the same with cx_alt / guide_alt for cx / guide
    unsigned short* guide_alt = nullptr; // second guide buffer (swaps with guide together with the row tables)
    u64* tab_W = nullptr;               // the tile table's copy of tile_W (mp_tab::W)
    bool draw_pending = false;          // with `deferred`: not even the draws of the last resample have been made (counter pending_rc);
    uint32_t pending_rc = 0;            // the next k_propagate makes them, or flush_draws() when anything else needs them first
    int pending_scheme = 0;             // (their resampling scheme)
    int use_fused_draws = 1;            // MP_FUSED_DRAWS=0: a resample always launches k_draw_slots (A/B measurements)
    int use_k1_mt = 1;                  // MP_K1_MT=0: drawing launches stay one workgroup per tile (k_propagate) instead of k_propagate_mt (A/B measurements)
    int cus = 0;                        // compute units of the device (= workgroups of a k_propagate_mt launch)
    int mt_flags = 0;                   // (diagnostics) MP_MT_SKIP_* forced for every k_propagate_mt launch
    // Lazy log-weights and parents (round 5).  A drawing k_propagate_mt launch stores neither: `resample` zeroes the one and replaces the
    // other (particle_filter.rs:109-114), so in a step / resample loop they are dead stores — 12 B per particle-step of fabric writes in a
    // kernel that is bound by its fabric transactions (-1.1 us per step at 2^20).  Whoever does read them before the next resample
    // (mp_pf_read_log_weights, mp_pf_read_parents, a further step without a resample) first REPLAYS that launch with the same arguments
    // and MP_MT_REPLAY: same Philox blocks, tables of the resampled generation (cx_alt, guide_alt, tiles_alt: untouched until the next
    // drawing launch), same parents and log-weights, nothing else stored (ensure_lazy).
    int use_lazy = 1;                   // MP_K1_LAZY=0 (diagnostics): every launch stores them
    bool lazy_pending = false;          // logw[] and parent[] are those of an EARLIER launch: lazy_args reproduces the current ones
    PropagateArgs lazy_args;
    mp_cx* cx_alt = nullptr;            // second row-table buffer: a k_propagate that looks up deferred draws in cx writes the new table here
    bool deferred = false;              // the last resample only drew: {dfr_lt, dfr_row}[slot] against the table in cx; x[cur] is the pre-resample state
    bool parents_deferred = false;      // ... and a step has consumed the draws since: its parents are still {dfr_lt, dfr_row} against cx_alt
    unsigned int* tab_ticket = nullptr;
    u64* tab_incl = nullptr;
    double* tab_ratio = nullptr;
    mp_tab_head* tab_head = nullptr;
    int use_k1_table = 1;               // MP_K1_TABLE=0: every k_draw_slots workgroup builds the table itself (A/B measurements)
    // local_table: the level-0 launches of this handle build NO job table (no ticket, no workgroup left behind at the end of every
    // step); a drawing k_propagate builds it per workgroup in LDS from the previous generation's tile scalars, which are therefore
    // double-buffered like the row table (tiles_alt), and k_build_table makes the global one when something else asks for it
    bool local_table = false;
    bool table_fresh = false;           // tab_* (the job's tile table in global memory) describe the current tile scalars
    u64* tiles_alt = nullptr;           // second [3][nt] tile-scalar buffer
    bool rows_fresh = false;            // cx / guide / tile_* describe the current log-weights
    bool x_in_rows = false;             // (dim_state 1) the current states are the x0 of the rows of cx; x[cur] is stale (ensure_x)
    // sharded-resample scratch (allocated on first use)
    unsigned char* sh_dest = nullptr;
    u64* sh_lt = nullptr;
    uint32_t* sh_tile = nullptr;
    uint32_t* sh_req_slot = nullptr;
    uint32_t* sh_blockcount = nullptr;
    uint32_t* sh_blockoff = nullptr;
    long long* sh_counts = nullptr;
    bool ow_placed = false;               // the last count's table launch also placed (self-drawn, equal-split capacity): expand launches nothing
    uint64_t ow_placed_cap = 0;
    double* ow_placed_send = nullptr;
    mp_tab_part* sh_tab_part = nullptr;   // [SH_MAX_WORLD] k_shard_table_mw: every rank's {sum T, sum T2}
    unsigned int* sh_tab_ticket = nullptr;   // counts up by `world` per launch
    unsigned int sh_tab_seq = 0;
    long long* h_counts = nullptr;  // pinned
    int sh_world = 0;
    u64 sh_cap = 0;                 // fixed-capacity exchange: request slots per (src, dst) pair
    double* sh_tm_all = nullptr;    // unpacked gathered tiles
    u64* sh_tW_all = nullptr;
    u64* sh_tW2_all = nullptr;
    double* sh_ratio_all = nullptr;   // and its per-tile (double)W / (double)T
    u64* sh_incl_all = nullptr;     // the job's tile table (inclusive prefix of T_b), built once per resample by k_shard_table
    int* sh_overflow = nullptr;
    unsigned int* sh_done = nullptr;   // [2] tickets of the route / resolve workgroups (the last one writes headers / publishes)
    u64* tiles_own = nullptr;       // the allocation behind tile_m / tile_W / tile_W2 unless the caller bound its own buffer
    mp_shard_pub* h_pub = nullptr;  // pinned, host-mapped
    mp_shard_pub* d_pub = nullptr;  // its device address
    hipEvent_t ev_resolved = nullptr;
    bool sh_lazy = false;           // the states of the last sharded resample still sit in the exchange buffer sh_rows, slot i at row sh_req_slot[i]
    const double* sh_rows = nullptr;
    bool sh_recv = false;           // (owner-keeps) the draws of the last resample hold MP_DRAW_RECV entries: rows of sh_rows
    u64 sh_rows_cap = 0;
    bool logw_zero = false;         // log-weights are all zero (after a sharded resample) and the buffer has not been cleared
    mp_dev_scalars* scal_undo = nullptr;  // the scalars before a fixed-capacity route folded this resample in
    // "owner keeps" form: per super-chunk of R * 1024 draws a window of entries (k_shard_own_draw), [ow_nsc][R * 1024]
    u64* ow_seg_lt = nullptr;             // tile-local target
    uint32_t* ow_seg_row = nullptr;       // start row of the forward scan
    uint32_t* ow_sccnt = nullptr;         // own draws per super-chunk, and their exclusive scan
    uint32_t* ow_base = nullptr;
    unsigned long long* ow_call = nullptr;   // offspring per rank [SH_MAX_WORLD]
    mp_owned_plan* ow_plan = nullptr;
    mp_own_range* ow_range = nullptr;
    u64* ow_kthr = nullptr;               // [SH_MAX_WORLD] rank boundaries as thresholds on the 52-bit uniforms (multinomial)
    u64 ow_last_cap = 0;                  // capacity of the last mp_pf_shard_owned_expand (0 = exact sizes: nothing can overflow)
    unsigned long long ow_seq = 0;        // owner-keeps resamples planned so far: the plan of number k writes pub->seq = k last
    int ow_nsc = 0, ow_R = 0, ow_wgs = 0;
    int ow_world = 0;
    int ow_scheme = 0;
    // self-drawn form (lattice schemes, split multinomial; mp_pf_shard_kernels.h): own offspring have their targets in closed form
    int use_shard_self = 1;               // MP_SHARD_SELF=0: the lattice schemes keep the window form (k_shard_own_draw / _place) — A/B and tests
    bool ow_self = false;                 // the resample being run (count .. commit) is self-drawn
    bool ow_solo_folded = false;          // (world of one) its scalars have been folded already (a synchronous commit): the next k_propagate must not
    int ow_rank = 0;
    mp_own_range* ow_range_solo = nullptr;   // {0, n}: a world of one owns every draw
    bool draws_lattice = false;           // the draws of the last resample are a lattice's (systematic / stratified)
    int walk_bisect_force = -1;           // MP_WALK_BISECT: 0 = the plain-walk kernels always, 1 = the bisecting ones always, unset = by the rule in launch_propagate
    bool pending_shard = false;           // with draw_pending: the pending draws are a sharded filter's self-drawn ones (world ps_world, rank ps_rank)
    int ps_world = 1, ps_rank = 0;
    bool sharded = false;
    // ancestry record (MP_PF_RECORD_HISTORY): the event log from which `traces[i].retv` is rebuilt
    struct HistEvent { int kind; void* buf; };  // kind 0: states after an Unfold step ([n][d] f64); 1: parents of a resample ([n] u32)
    std::vector<HistEvent> hist;
    // the event buffers come out of slabs that grow geometrically, so a step or a resample does not call hipMalloc
    std::vector<void*> hist_slabs;
    unsigned char* hist_slab_cur = nullptr;
    size_t hist_slab_left = 0, hist_slab_next = 0;
    mp_hist_event* d_hist_events = nullptr;   // device copy of the log for k_trajectories
    size_t d_hist_events_cap = 0;
    // host-side filter state
    long long t = 0;  // Unfold steps taken (trace.args.0)
    uint32_t resample_count = 0;
    bool initialised = false;
    // timing
    bool timing = false;
    std::vector<TimedLaunch> timed;
    std::vector<hipEvent_t> event_pool;
    int last_k1_form = -1;                          // MP_K1_FORM_* of the last k_propagate-family launch
    hipEvent_t region_ev[2] = {nullptr, nullptr};   // mp_pf_region_begin / _end
    bool region_open = false;
    uint64_t region_launches = 0;
    double fam_ms[MP_K_COUNT] = {0, 0, 0, 0};
    uint64_t fam_launches[MP_K_COUNT] = {0, 0, 0, 0};
};

static mp_tab tab_of(const mp_pf* h) {
    mp_tab t;
    const bool on = h->use_k1_table && !h->sharded && h->tab_ticket && !h->local_table;
    t.ticket = on ? h->tab_ticket : nullptr;
    t.incl = h->tab_incl;
    t.ratio = h->tab_ratio;
    t.head = h->tab_head;
    t.S = h->S;
    t.W = h->tab_W;
    return t;
}

static int32_t update_k1_tail(mp_pf* h) {   // after anything that changes one of these pointers
    mp_k1_tail t;
    t.cx = h->cx; t.guide = h->guide; t.tile_m = h->tile_m; t.tile_W = h->tile_W; t.tile_W2 = h->tile_W2; t.tab = tab_of(h);
    if (!h->k1_tail) HIPCK(hipMalloc(&h->k1_tail, sizeof(mp_k1_tail)));
    HIPCK(hipMemcpyAsync(h->k1_tail, &t, sizeof(t), hipMemcpyHostToDevice, h->stream));
    HIPCK(hipStreamSynchronize(h->stream));   // `t` is a stack object
    if (h->cx_alt) {
        t.cx = h->cx_alt;
        t.guide = h->guide_alt;
        if (h->local_table) {   // (the launch that draws reads the old generation's tile scalars while it writes the new one's)
            t.tile_m = reinterpret_cast<double*>(h->tiles_alt); t.tile_W = h->tiles_alt + h->nt; t.tile_W2 = h->tiles_alt + 2 * (size_t)h->nt;
        }
        if (!h->k1_tail_alt) HIPCK(hipMalloc(&h->k1_tail_alt, sizeof(mp_k1_tail)));
        HIPCK(hipMemcpyAsync(h->k1_tail_alt, &t, sizeof(t), hipMemcpyHostToDevice, h->stream));
        HIPCK(hipStreamSynchronize(h->stream));
    }
    return MP_OK;
}

static hipEvent_t get_event(mp_pf* h) {
    if (!h->event_pool.empty()) {
        hipEvent_t e = h->event_pool.back();
        h->event_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
struct LaunchTimer {
    mp_pf* h;
    TimedLaunch tl;
    bool on;
    LaunchTimer(mp_pf* h_, int family) : h(h_), on(h_->timing) {
        if (on) {
            tl.family = family;
            tl.start = get_event(h);
            tl.stop = get_event(h);
            (void)hipEventRecord(tl.start, h->stream);
        }
    }
    ~LaunchTimer() {
        if (on) {
            (void)hipEventRecord(tl.stop, h->stream);
            h->timed.push_back(tl);
        }
    }
};
static int32_t drain_timing(mp_pf* h) {
    if (h->timed.empty()) return MP_OK;
    HIPCK(hipStreamSynchronize(h->stream));
    for (auto& tl : h->timed) {
        float ms = 0.f;
        HIPCK(hipEventElapsedTime(&ms, tl.start, tl.stop));
        h->fam_ms[tl.family] += (double)ms;
        h->fam_launches[tl.family] += 1;
        h->event_pool.push_back(tl.start);
        h->event_pool.push_back(tl.stop);
    }
    h->timed.clear();
    return MP_OK;
}

static int32_t check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mp_fail(MP_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    return MP_OK;
}

// Wait for the handle's stream by polling: a blocking hipStreamSynchronize wakes the host tens of microseconds after the
// last kernel has finished, which is several SMC steps' worth at 2^20 particles (the handle is single-threaded: nothing
// else wants this core meanwhile).
static hipError_t stream_wait(hipStream_t s) {
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return e;
    }
}
// Poll one host-mapped sequence word until the device has written `want` (data words are written before it, release order).
// If the stream drains without it, whoever was to write it never ran: an error of the library, reported rather than spun on.
static int32_t wait_seq(mp_pf* h, const volatile unsigned long long* word, unsigned long long want, const char* what) {
    for (unsigned spins = 1;; ++spins) {
        if (*word == want) break;
        if ((spins & 255u) == 0u) {
            const hipError_t e = hipStreamQuery(h->stream);
            if (e == hipErrorNotReady) continue;
            if (e != hipSuccess) return mp_fail(MP_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
            if (*word == want) break;
            return mp_fail(MP_ERR_STATE, std::string(what) + ": the stream drained without the value (sequence " + std::to_string(*word) + ", expected " + std::to_string(want) + ")");
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (*(volatile int*)h->h_flag)
        return mp_fail(MP_ERR_DEGENERATE, "all log-weights are -inf: normalized weights are NaN (categorical.rs:23 assert in the reference)");
    return MP_OK;
}
static int32_t flush_draws(mp_pf* h);
// the log-weights and parents a drawing k_propagate_mt launch did not store (mp_pf::lazy_pending), now
static int32_t ensure_lazy(mp_pf* h) {
    if (!h->lazy_pending) return MP_OK;
    h->lazy_pending = false;
    PropagateArgs a = h->lazy_args;
    a.mt_flags = MP_MT_REPLAY;
    (void)h->ops->propagate(a);
    return check_launch("k_propagate_mt (replay for the log-weights and parents)");
}
static int32_t fetch_scalars(mp_pf* h) {
    {   // (the scalars of a resample are folded by whoever makes its draws)
        int32_t rcf = flush_draws(h);
        if (rcf != MP_OK) return rcf;
    }
    HIPCK(hipMemcpyAsync(h->h_scal, h->scal, sizeof(mp_dev_scalars), hipMemcpyDeviceToHost, h->stream));
    HIPCK(stream_wait(h->stream));
    if (h->h_scal->degenerate)
        return mp_fail(MP_ERR_DEGENERATE, "all log-weights are -inf: normalized weights are NaN (categorical.rs:23 assert in the reference)");
    return MP_OK;
}

// dim_state 1: k_propagate leaves the new states in the row table only; whoever wants x[cur] gets it here
static int32_t ensure_x(mp_pf* h) {
    if (!h->x_in_rows) return MP_OK;
    hipLaunchKernelGGL(k_rows_to_x, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, h->stream, h->n, (const mp_cx*)h->cx, h->x[h->cur]);
    h->x_in_rows = false;
    return check_launch("k_rows_to_x");
}

// Slot-order x / parent / logw after a resample that only drew (only when something other than the next step needs them).
static int32_t materialize(mp_pf* h) {
    if (h->sh_lazy) {
        hipLaunchKernelGGL(k_shard_adopt_rows, dim3((unsigned)((h->n + SH_THREADS - 1) / SH_THREADS)), dim3(SH_THREADS), 0, h->stream, h->n,
                           h->ops->dim_state, h->sh_rows, h->sh_req_slot, (const double*)h->x[h->cur ^ 1], h->slot_offset, h->x[h->cur], h->parent);
        h->sh_lazy = false;
        h->sh_parents_lazy = false;
        h->x_in_rows = false;   // (every slot's state was just written)
        int32_t rc = check_launch("k_shard_adopt_rows");
        if (rc != MP_OK) return rc;
    }
    if (h->logw_zero) {
        HIPCK(hipMemsetAsync(h->logw, 0, sizeof(double) * h->n, h->stream));
        h->logw_zero = false;
    }
    if (h->deferred) {
        {
            int32_t rcf = flush_draws(h);
            if (rcf != MP_OK) return rcf;
        }
        const int d = h->ops->dim_state;
        hipLaunchKernelGGL(k_resolve_slots<true>, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, h->stream, h->n, d, (const u64*)h->dfr_lt,
                           (const uint32_t*)h->dfr_row, (const mp_cx*)h->cx, (const double*)h->x[h->cur], d == 1 ? h->x[h->cur] : h->x[h->cur ^ 1],
                           h->parent, h->logw, h->sh_recv ? h->sh_rows : (const double*)nullptr, h->slot_offset);
        if (d > 1) h->cur ^= 1;   // wider states were gathered from the pre-resample buffer into the other one
        h->deferred = false;
        h->parents_deferred = false;   // k_resolve_slots wrote parent[] too
        h->x_in_rows = false;          // ... and every slot's state
        return check_launch("k_resolve_slots");
    }
    return MP_OK;   // (x[cur] itself: ensure_x, for the callers that read it)
}

// a buffer of `bytes` for one history event, out of the current slab (a new slab is twice the last one, at least 8 events)
static int32_t hist_alloc(mp_pf* h, size_t bytes, void** out) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (h->hist_slab_left < bytes) {
        size_t want = h->hist_slab_next ? h->hist_slab_next : 8 * bytes;
        if (want < bytes) want = bytes;
        void* slab = nullptr;
        HIPCK(hipMalloc(&slab, want));
        h->hist_slabs.push_back(slab);
        h->hist_slab_cur = static_cast<unsigned char*>(slab);
        h->hist_slab_left = want;
        h->hist_slab_next = 2 * want;
    }
    *out = h->hist_slab_cur;
    h->hist_slab_cur += bytes;
    h->hist_slab_left -= bytes;
    return MP_OK;
}

static int32_t launch_propagate(mp_pf* h, const double* args0, const double* obs, bool overwrite) {
    // (still pending here = no resample since: this is a further step of the same generation, which accumulates onto the log-weights)
    { int32_t rcl = ensure_lazy(h); if (rcl != MP_OK) return rcl; }
    PropagateArgs a;
    a.n = h->n; a.slot_offset = h->slot_offset;
    a.k0 = (uint32_t)h->seed; a.k1 = (uint32_t)(h->seed >> 32);
    a.t = h->t;
    // with deferred draws and dim_state > 1 this propagate gathers the parents' states itself, from the pre-resample buffer
    // into the other one
    const bool gather_here = h->deferred && !h->sh_lazy && h->ops->dim_state > 1;
    // (after a sharded commit x[cur ^ 1] still holds the pre-resample states: kept offspring of the owner-keeps exchange read
    // their parents there when states are wider than one double)
    a.x_in = h->sh_lazy ? h->x[h->cur ^ 1] : h->x[h->cur]; a.x_out = gather_here ? h->x[h->cur ^ 1] : h->x[h->cur];
    a.logw = h->logw;
    for (int j = 0; j < MP_MAX_OBS; ++j) a.obs.v[j] = (j < h->ops->dim_obs) ? obs[j] : 0.;
    for (int j = 0; j < MP_MAX_STATE; ++j) a.s0.v[j] = (args0 && j < h->ops->dim_state) ? args0[j] : 0.;
    a.overwrite = overwrite ? 1 : ((h->deferred || h->logw_zero) ? 2 : 0);
    // draws of the last resample are looked up by this launch, in the table they were drawn against (cx); the new table goes
    // into the other buffer
    a.dfr_row = h->deferred ? h->dfr_row : nullptr;
    a.dfr_lt = h->deferred ? h->dfr_lt : nullptr;
    a.cx_old = h->deferred ? h->cx : nullptr;
    a.inv_rows = h->sh_lazy ? h->sh_rows : ((h->deferred && h->sh_recv) ? h->sh_rows : nullptr);
    a.inv = h->sh_lazy ? h->sh_req_slot : nullptr;
    a.cx = h->deferred ? h->cx_alt : h->cx; a.guide = h->deferred ? h->guide_alt : h->guide;
    a.tile_m = h->tile_m; a.tile_W = h->tile_W; a.tile_W2 = h->tile_W2;
    a.tile_m_new = h->tile_m; a.tile_W_new = h->tile_W; a.tile_W2_new = h->tile_W2;
    if (h->deferred && h->local_table && h->tiles_alt) {   // (what k1_tail_alt says: update_k1_tail)
        a.tile_m_new = reinterpret_cast<double*>(h->tiles_alt); a.tile_W_new = h->tiles_alt + h->nt; a.tile_W2_new = h->tiles_alt + 2 * (size_t)h->nt;
    }
    // Long row walks (collapsed weights: mp_resolve_draws' BISECT) — which instantiation looks the draws up.  One-double models: always
    // the long-walk one (with MP_WALK_LINEAR = 12 it measures the same as the plain walk on healthy weights — 37.3 us either way,
    // six alternations on one box — and an ESS that drops from 5 x 10^5 to 9 within one step no longer costs a 490 us step; a rule on
    // the LAST normalisation's ESS, tried first, is a step late for exactly that case).  Wide models: when the draws are a lattice's
    // (consecutive slots share a tile's long walks: the 40 % tail of DESIGN.md section 5; under multinomial draws the bisecting instantiation
    // is 4 % slower on C5 and buys nothing).  Both find the same parents: the choice changes time only.
    {
        const bool wide = !(h->ops->max_normals <= 4 && h->ops->dim_state <= 4);
        a.walk_bisect = h->deferred && (wide ? h->draws_lattice : true);
        if (h->walk_bisect_force >= 0) a.walk_bisect = h->walk_bisect_force != 0;   // MP_WALK_BISECT=0 / 1 (tests, A/B): never / always
    }
    a.mt_grid = (h->use_k1_mt && !h->sharded) ? h->cus : 0;
    a.mt_flags = h->mt_flags;
    const bool lazy = h->use_lazy && h->deferred && h->draw_pending && !h->pending_shard && h->pending_scheme == MP_RESAMPLE_MULTINOMIAL;
    if (lazy) a.mt_flags |= MP_MT_SKIP_LOGW | MP_MT_SKIP_PARENT;   // (read by k_propagate_mt only; below: whether that is what ran)
    // The synchronous loop (`step; ESS; L = resample()`, tests/smc.rs:64-90): when the last resample was synchronous the next one will
    // be — the launch's last workgroup computes its return value on the way out (mt_peek_tail: a ~3 us tail) instead of a k_peek_level1
    // launch behind the step (a launch on an idle queue + a dependency gap)
    const bool want_peek = h->sync_loop && h->d_mirror && h->tab_ticket && !h->sharded && h->local_table;
    a.peek_ticket = nullptr; a.peek_mirror = nullptr; a.peek_seq = 0;
    if (want_peek) { a.mt_flags |= MP_MT_PEEK; a.peek_ticket = h->tab_ticket; a.peek_mirror = h->d_mirror; a.peek_seq = h->peek_seq + 1; }
    // ... and made by it too, when the resample left them pending (kernels of two-slot lanes)
    a.drw = (h->deferred && h->draw_pending) ? (1 | (h->pending_scheme << 1)) : 0;   // bit 0: draw; bits 1..2: the scheme
    a.drw_v = mp_k1_draw{};
    if (a.drw) {
        mp_k1_draw& d = a.drw_v;
        d.tile_m_old = h->tile_m; d.tile_W_old = h->tile_W; d.tile_W2_old = h->tile_W2;   // (this launch writes the other set: k1_tail_alt)
        d.guide_old = h->guide; d.scal = h->scal; d.parent = h->parent;
        d.n_global = h->n_global; d.nt = h->nt; d.S = h->S;
    }
    if (a.drw && h->pending_shard) {
        // a sharded filter's self-drawn resample: the table is this rank's slice of the job's (k_shard_table's arrays; in a world of one
        // the table the last level-0 launch left), own offspring [0, c_me) of ow_range, the slots beyond read their received rows
        mp_k1_draw& d = a.drw_v;
        const bool solo = h->ps_world == 1;
        const size_t off = (size_t)h->ps_rank * (size_t)h->nt;
        d.tile_m_old = solo ? (const double*)h->tab_ratio : (const double*)h->sh_ratio_all + off;
        d.tile_W_old = solo ? (const u64*)h->tab_W : (const u64*)h->sh_tW_all + off;
        d.tile_W2_old = solo ? (const u64*)h->tab_incl : (const u64*)h->sh_incl_all + off;
        d.shd_range = solo ? h->ow_range_solo : h->ow_range;
        d.shd_head = (solo && !h->ow_solo_folded) ? (const mp_tab_head*)h->tab_head : nullptr;
        d.shd_rank = h->ps_rank; d.shd_world = h->ps_world;
        a.dfr_lt = nullptr;   // (dfr_row: the flags of the slots other ranks fill)
    } else if (a.drw) { a.dfr_row = nullptr; a.dfr_lt = nullptr; }   // (not read: the kernel writes them through the struct's pointers)
    a.rc = h->pending_rc;
    a.dyn_lds = a.drw ? 24 * (size_t)h->nt : 0;
    a.grid = h->nt;
    a.stream = h->stream;
    a.aux.tab = tab_of(h);
    a.aux.x_rows = 0;
    if (h->ops->dim_state == 1 && h->x_in_rows && !h->deferred && !h->sh_lazy) {   // a plain step: the previous states are the x0 of the current rows
        a.x_in = reinterpret_cast<const double*>(h->cx);
        a.aux.x_rows = 1;
    }
    a.tail = h->deferred ? h->k1_tail_alt : h->k1_tail;
    {
        LaunchTimer lt(h, MP_K_PROPAGATE);
        h->last_k1_form = h->ops->propagate(a);
    }
    h->region_launches += 1;
    if (lazy && h->last_k1_form == MP_K1_FORM_TWO_TILES) {
        h->lazy_pending = true;
        h->lazy_args = a;
    }
    h->peek_valid = want_peek && h->last_k1_form == MP_K1_FORM_TWO_TILES;
    if (h->peek_valid) h->peek_seq += 1;
    if (h->deferred) {   // the fresh table is the current one from here on; the old one stays intact for mp_pf_read_parents
        std::swap(h->cx, h->cx_alt);
        std::swap(h->guide, h->guide_alt);
        std::swap(h->k1_tail, h->k1_tail_alt);
        if (h->local_table) {
            std::swap(h->tiles_own, h->tiles_alt);
            h->tile_m = reinterpret_cast<double*>(h->tiles_own); h->tile_W = h->tiles_own + h->nt; h->tile_W2 = h->tiles_own + 2 * (size_t)h->nt;
        }
        h->draw_pending = false;
        h->pending_shard = false;
        h->parents_deferred = !a.drw;   // (a launch that made the draws itself wrote parent[] as well)
        h->deferred = false;
    }
    h->t += 1;
    if (gather_here) h->cur ^= 1;
    if (h->ops->dim_state == 1) h->x_in_rows = true;   // k_propagate left the new states in the row table only
    h->logw_zero = false;
    // parents of that resample stay where they are (the draws + the old table, or the exchange rows) until somebody asks for
    // them or the next resample replaces them: mp_pf_read_parents (particle_filter.rs:20 keeps `parents` across `step`)
    if (h->sh_lazy) h->sh_parents_lazy = true;
    h->sh_lazy = false;    // k_propagate wrote x[cur] and logw in slot order ...
    h->rows_fresh = true;  // ... and level 0 of their normalisation
    h->table_fresh = a.aux.tab.ticket != nullptr;   // (built by that launch's last workgroup, or not at all)
    int32_t rc_ = check_launch("k_propagate");
    if (rc_ != MP_OK) return rc_;
    if (h->flags & MP_PF_RECORD_HISTORY) {
        void* buf = nullptr;
        const size_t bytes = sizeof(double) * h->n * (size_t)h->ops->dim_state;
        int32_t rch = hist_alloc(h, bytes, &buf);
        if (rch != MP_OK) return rch;
        {
            int32_t rcx = ensure_x(h);
            if (rcx != MP_OK) return rcx;
        }
        HIPCK(hipMemcpyAsync(buf, h->x[h->cur], bytes, hipMemcpyDeviceToDevice, h->stream));
        h->hist.push_back({0, buf});
    }
    return MP_OK;
}

// level 0 for the current log-weights when no propagate produced it (after a resample: the weights are zero)
static int32_t ensure_rows(mp_pf* h) {
    int32_t rc = materialize(h);
    if (rc != MP_OK) return rc;
    if (h->rows_fresh) return MP_OK;
    rc = ensure_x(h);
    if (rc != MP_OK) return rc;
    {
        LaunchTimer lt(h, MP_K_NORMALIZE_SCAN);
        hipLaunchKernelGGL(k_normalize_tiles, dim3(h->nt), dim3(TILE_THREADS), 0, h->stream, h->logw, h->x[h->cur], h->ops->dim_state, h->n, h->cx, h->guide,
                           h->tile_m, h->tile_W, h->tile_W2, tab_of(h));
    }
    h->table_fresh = tab_of(h).ticket != nullptr;
    h->rows_fresh = true;
    h->peek_valid = false;   // (other tile scalars than the ones a k_propagate_mt tail peeked at)
    return check_launch("k_normalize_tiles");
}

static int32_t shard_scratch(mp_pf* h, int world, u64 cap);

extern "C" {

const char* mp_last_error(void) { return g_err.c_str(); }

int32_t mp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int32_t mp_pf_create(const mp_model_desc* model, uint64_t n_particles, uint64_t seed, const mp_shard* shard, uint32_t flags,
                     int32_t device, void* stream, mp_pf** out) {
    if (!out) return mp_fail(MP_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    if (n_particles == 0) return mp_fail(MP_ERR_INVALID_ARG, "n_particles must be > 0");
    if (n_particles > 0xFFFFFFFFull) return mp_fail(MP_ERR_INVALID_ARG, "n_particles must fit u32 parent indices");
    struct Cleanup { void operator()(mp_pf* p) const { (void)mp_pf_destroy(p); } };   // an early return frees what was allocated so far
    std::unique_ptr<mp_pf, Cleanup> h(new mp_pf());
    int32_t rc = make_model(model, h->ops);
    if (rc != MP_OK) return rc;
    h->n = n_particles;
    h->n_global = shard ? shard->n_global : n_particles;
    h->slot_offset = shard ? shard->slot_offset : 0;
    if (h->n_global < h->n + h->slot_offset || h->n_global > 0xFFFFFFFFull) return mp_fail(MP_ERR_INVALID_ARG, "shard does not fit n_global (< 2^32)");
    h->sharded = shard && (h->n_global != h->n);
    if (h->sharded && ((h->slot_offset % TILE) || (h->n % TILE)))
        return mp_fail(MP_ERR_INVALID_ARG, "shards must be tile-aligned: slot_offset and n_particles multiples of 2048");
    h->seed = seed;
    h->flags = flags;
    h->device = device;
    h->S = 62 - ceil_log2_u64(h->n_global);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        return mp_fail(MP_ERR_HIP, "no HIP device visible: the gfx950 path has no CPU fallback");
    }
    HIPCK(hipSetDevice(device));
    if (stream) {
        h->stream = (hipStream_t)stream;
    } else {
        HIPCK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->own_stream = true;
    }
    const u64 n = h->n;
    const int d = h->ops->dim_state;
    h->nt = (int)((n + TILE - 1) / TILE);
    if ((h->n_global + TILE - 1) / TILE > MAX_TILES) return mp_fail(MP_ERR_UNSUPPORTED, "at most 2^24 particles per job in this build (tile table in LDS)");
    h->k3_grid = (int)((n + KG_THREADS * KG_ITEMS - 1) / (KG_THREADS * KG_ITEMS));
    if (h->k3_grid > K3_MAX_BLOCKS) h->k3_grid = K3_MAX_BLOCKS;
    h->nchunks = (int)((n + DRAW_CHUNK - 1) / DRAW_CHUNK);
    {
        const char* env = mp_diag_env("MP_DEFERRED_LOOKUPS");
        if (env && env[0] == '0') h->use_deferred = 0;
        env = mp_diag_env("MP_K1_TABLE");
        if (env && env[0] == '0') h->use_k1_table = 0;
        env = mp_diag_env("MP_FUSED_DRAWS");
        if (env && env[0] == '0') h->use_fused_draws = 0;
        env = mp_diag_env("MP_WALK_BISECT");
        if (env && (env[0] == '0' || env[0] == '1')) h->walk_bisect_force = env[0] - '0';
        env = mp_diag_env("MP_SHARD_SELF");
        if (env && env[0] == '0') h->use_shard_self = 0;
        env = mp_diag_env("MP_K1_MT");
        if (env && env[0] == '0') h->use_k1_mt = 0;
        HIPCK(hipDeviceGetAttribute(&h->cus, hipDeviceAttributeMultiprocessorCount, device));
        env = mp_diag_env("MP_K1_LAZY");
        if (env && env[0] == '0') h->use_lazy = 0;
        env = mp_diag_env("MP_K1_MT_GRID");   // (A/B measurements: another number of workgroups)
        if (env && atoi(env) > 0) h->cus = atoi(env);
        env = mp_diag_env("MP_K1_MT_FLAGS");  // (A/B measurements: MP_MT_SKIP_* for every launch — results of reads are then undefined)
        if (env) h->mt_flags = atoi(env);
        // kernels that make their draws themselves build the job's tile table themselves too (MP_K1_LOCAL_TABLE=0: the last
        // workgroup of every level-0 launch builds it, as for every other kernel)
        env = mp_diag_env("MP_K1_LOCAL_TABLE");
        h->local_table = !h->sharded && h->use_k1_table && h->use_fused_draws && h->use_deferred && h->ops->can_draw && h->nt <= 2048 &&
                         !(h->flags & MP_PF_RECORD_HISTORY) && !(env && env[0] == '0');
    }
    HIPCK(hipMalloc(&h->x[0], sizeof(double) * n * d));
    HIPCK(hipMalloc(&h->x[1], sizeof(double) * n * d));
    HIPCK(hipMalloc(&h->logw, sizeof(double) * n));
    HIPCK(hipMalloc(&h->cx, sizeof(mp_cx) * (size_t)h->nt * TILE));
    HIPCK(hipMalloc(&h->guide, sizeof(unsigned short) * (size_t)h->nt * GUIDE_N));
    HIPCK(hipMalloc(&h->parent, sizeof(uint32_t) * n));
    HIPCK(hipMalloc(&h->tiles_own, sizeof(u64) * 3 * h->nt));   // packed [3][nt]: m (f64 bits), W, W2 — the unit the shards all-gather
    h->tile_m = reinterpret_cast<double*>(h->tiles_own);
    h->tile_W = h->tiles_own + h->nt;
    h->tile_W2 = h->tiles_own + 2 * (size_t)h->nt;
    HIPCK(hipMalloc(&h->scal, sizeof(mp_dev_scalars)));
    if (!h->sharded) {
        HIPCK(hipMalloc(&h->tab_ticket, 64));   // a line of its own
        HIPCK(hipMemsetAsync(h->tab_ticket, 0, 64, h->stream));
        HIPCK(hipMalloc(&h->tab_incl, sizeof(u64) * h->nt));
        HIPCK(hipMalloc(&h->tab_ratio, sizeof(double) * h->nt));
        HIPCK(hipMalloc(&h->tab_head, sizeof(mp_tab_head)));
        HIPCK(hipMalloc(&h->tab_W, sizeof(u64) * h->nt));
    }
    HIPCK(hipMalloc(&h->aos, sizeof(double) * n));   // scratch for importance sampling's normalised log-weights
    HIPCK(hipHostMalloc(&h->h_scal, sizeof(mp_dev_scalars)));
    if (!h->sharded) {   // (sharded handles resample through the mp_pf_shard_* phases)
        HIPCK(hipMalloc(&h->dfr_lt, sizeof(u64) * (size_t)h->nchunks * DRAW_CHUNK));
        HIPCK(hipMalloc(&h->dfr_row, sizeof(uint32_t) * (size_t)h->nchunks * DRAW_CHUNK));
    }
    // tile tables above 64 KiB of LDS need the limit raised once per kernel
    {
        const int nt_job = (int)((h->n_global + TILE - 1) / TILE);
        const size_t need = table_lds(nt_job, K3_THREADS) + 8192;   // + the per-key counters of the sharded route
        if (need > 48 * 1024) {
            HIPCK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_resample_gather<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
            HIPCK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_resample_gather<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
            HIPCK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_resample_gather<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
            HIPCK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_draw_slots<0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
            HIPCK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_draw_slots<0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
            HIPCK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_draw_slots<0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
            HIPCK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_finalize_tiles), hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
            HIPCK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_shard_targets), hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
        }
    }
    // ParticleSystem::new: log_weights = 0, parents = 0, log_ml_estimate = 0 (particle_filter.rs:44-57)
    HIPCK(hipMemsetAsync(h->x[0], 0, sizeof(double) * n * d, h->stream));
    HIPCK(hipMemsetAsync(h->x[1], 0, sizeof(double) * n * d, h->stream));
    HIPCK(hipMemsetAsync(h->logw, 0, sizeof(double) * n, h->stream));
    HIPCK(hipMemsetAsync(h->parent, 0, sizeof(uint32_t) * n, h->stream));
    HIPCK(hipHostMalloc(&h->h_flag, sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
    *h->h_flag = 0;
    mp_dev_scalars init{};
    init.ess_stale = 1.0 / (double)h->n_global;  // exp(-logsumexp(zeros)) before any resample
    HIPCK(hipHostGetDevicePointer((void**)&init.host_flag, h->h_flag, 0));
    {
        const char* env = mp_diag_env("MP_HOST_MIRROR");
        if (env && env[0] == '0') h->use_mirror = 0;
    }
    if (!h->sharded && h->use_mirror) {   // (sharded handles fold, undo and re-fold through mp_pf_shard_*: they keep the copy)
        HIPCK(hipHostMalloc(&h->h_mirror, sizeof(mp_host_mirror), hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(h->h_mirror, 0, sizeof(mp_host_mirror));
        h->h_mirror->ess_stale = init.ess_stale;
        HIPCK(hipHostGetDevicePointer((void**)&h->d_mirror, h->h_mirror, 0));
        init.mirror = h->d_mirror;
    }
    *h->h_scal = init;
    HIPCK(hipMemcpyAsync(h->scal, h->h_scal, sizeof(mp_dev_scalars), hipMemcpyHostToDevice, h->stream));
    HIPCK(hipStreamSynchronize(h->stream));
    {
        int32_t rct = update_k1_tail(h.get());
        if (rct != MP_OK) return rct;
    }
    *out = h.release();
    return MP_OK;
}

int32_t mp_pf_init_step(mp_pf* h, const double* args0, const double* obs, int32_t n_steps) {
    if (!h || !obs) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    if (n_steps < 1) return mp_fail(MP_ERR_CONSTRAINTS, "init_step needs the constraints of at least one time step");
    if (h->initialised) return mp_fail(MP_ERR_STATE, "init_step called twice");
    HIPCK(hipSetDevice(h->device));
    for (int k = 0; k < n_steps; ++k) {
        int32_t rc = launch_propagate(h, args0, obs + (size_t)k * h->ops->dim_obs, k == 0);
        if (rc != MP_OK) return rc;
    }
    h->initialised = true;
    return MP_OK;
}

int32_t mp_pf_step(mp_pf* h, const double* obs, int32_t n_steps) {
    if (!h || !obs) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    if (n_steps < 1) return mp_fail(MP_ERR_CONSTRAINTS, "step needs the constraints of at least one time step");
    if (!h->initialised) return mp_fail(MP_ERR_STATE, "step before init_step");
    HIPCK(hipSetDevice(h->device));
    for (int k = 0; k < n_steps; ++k) {
        int32_t rc = launch_propagate(h, nullptr, obs + (size_t)k * h->ops->dim_obs, false);
        if (rc != MP_OK) return rc;
    }
    return MP_OK;
}

// The job's tile table in global memory (tab_*), for the kernels that read it there: built by the last workgroup of the level-0
// launch when that launch was given the ticket, by one workgroup of its own otherwise (handles whose drawing k_propagates
// build the table per workgroup in LDS; a handle that changed modes).  false: this handle has no such buffers (sharded).
static bool ensure_table(mp_pf* h) {
    if (!h->tab_incl || h->sharded || !h->use_k1_table) return false;
    if (!h->table_fresh) {
        mp_tab tb = tab_of(h);
        tb.ticket = nullptr;
        hipLaunchKernelGGL(k_build_table, dim3(1), dim3(1024), 0, h->stream, (const double*)h->tile_m, (const u64*)h->tile_W, (const u64*)h->tile_W2, h->nt, tb);
        h->table_fresh = true;
    }
    return true;
}

// k_draw_slots for resample number `rc` of scheme `scheme`: {target, start row} per output slot
static int32_t launch_draws(mp_pf* h, int32_t scheme, uint32_t rc) {
    {
        LaunchTimer lt(h, MP_K_BIN_DRAWS);
        const size_t lds_tail = (sizeof(double) + sizeof(u64)) * (DRAW_THREADS / 64);
        // tile table: in global memory (ensure_table) and copied to LDS (1) or, beyond K1_TABLE_LDS_MAX_TILES, probed in L2 (2);
        // handles without such a table build it per workgroup (0)
        const int tabmode = ensure_table(h) ? (h->nt <= K1_TABLE_LDS_MAX_TILES ? 1 : 2) : 0;
        const size_t lds = (tabmode == 1 ? 24 * (size_t)h->nt : tabmode == 0 ? 16 * (size_t)h->nt : 0) + lds_tail;
        const u64* incl = tabmode ? (const u64*)h->tab_incl : nullptr;
        const double* ratio = tabmode ? (const double*)h->tab_ratio : nullptr;
        const mp_tab_head* head = tabmode ? (const mp_tab_head*)h->tab_head : nullptr;
        // (two workgroups per CU when the table is copied to LDS; the L2-probing form has no table to amortise)
        static const int draw_wgs = [] { const char* e = mp_diag_env("MP_DRAW_WGS"); return e ? atoi(e) : 512; }();
        // (measured: 2^22 particles / 2048 tiles 55 -> 45 us; at 1024 tiles one workgroup per chunk is the faster form, 26 against 29 us)
        const int draw_grid = (tabmode == 2 || h->nt <= 1024) ? h->nchunks : std::min(h->nchunks, std::max(1, draw_wgs));
#define MP_LAUNCH_DRAW(TM, SC)                                                                                                              \
        hipLaunchKernelGGL((k_draw_slots<TM, SC>), dim3(draw_grid), dim3(DRAW_THREADS), lds, h->stream, h->n, h->n_global, h->slot_offset,       \
                           (uint32_t)h->seed, (uint32_t)(h->seed >> 32), rc, h->S, h->tile_m, h->tile_W, h->tile_W2, h->nt,    \
                           h->guide, h->dfr_lt, h->dfr_row, h->scal, incl, ratio, head, h->nchunks)
#define MP_LAUNCH_DRAW_SCHEME(TM)                                                                                                           \
        do {                                                                                                                                \
            if (scheme == MP_RESAMPLE_MULTINOMIAL) MP_LAUNCH_DRAW(TM, 0);                                                                   \
            else if (scheme == MP_RESAMPLE_SYSTEMATIC) MP_LAUNCH_DRAW(TM, 1);                                                               \
            else MP_LAUNCH_DRAW(TM, 2);                                                                                                     \
        } while (0)
        if (tabmode == 1) MP_LAUNCH_DRAW_SCHEME(1);
        else if (tabmode == 2) MP_LAUNCH_DRAW_SCHEME(2);
        else MP_LAUNCH_DRAW_SCHEME(0);
#undef MP_LAUNCH_DRAW_SCHEME
#undef MP_LAUNCH_DRAW
    }
    return check_launch("k_draw_slots");
}

// the draws a resample left to the next k_propagate, made now because something else needs them first
static int32_t launch_self_draws(mp_pf* h, int scheme, uint32_t rc, int world, int rank);
static int32_t flush_draws(mp_pf* h) {
    if (!h->draw_pending) return MP_OK;
    h->draw_pending = false;
    if (h->pending_shard) {
        h->pending_shard = false;
        if (h->ps_world == 1 && !h->ow_solo_folded) {   // (what the next k_propagate would have done on the way)
            hipLaunchKernelGGL(k_shard_solo_fold, dim3(1), dim3(1), 0, h->stream, (const mp_tab_head*)h->tab_head, h->scal, h->S, h->n_global);
            h->ow_solo_folded = true;
        }
        return launch_self_draws(h, h->pending_scheme, h->pending_rc, h->ps_world, h->ps_rank);
    }
    return launch_draws(h, h->pending_scheme, h->pending_rc);
}

int32_t mp_pf_resample(mp_pf* h, int32_t scheme, double* log_total_weight) {
    if (!h) return mp_fail(MP_ERR_INVALID_ARG, "null handle");
    if (!h->initialised) return mp_fail(MP_ERR_STATE, "resample before init_step");
    if (scheme == MP_RESAMPLE_MULTINOMIAL_SPLIT) scheme = MP_RESAMPLE_MULTINOMIAL;   // (one rank: the same draws, include/modppl_hip.h)
    if (scheme != MP_RESAMPLE_MULTINOMIAL && scheme != MP_RESAMPLE_SYSTEMATIC && scheme != MP_RESAMPLE_STRATIFIED)
        return mp_fail(MP_ERR_INVALID_ARG, "unknown resampling scheme");
    if (h->sharded) return mp_fail(MP_ERR_STATE, "sharded handle: resample runs through the mp_pf_shard_* phases");
    HIPCK(hipSetDevice(h->device));
    h->lazy_pending = false;   // log_weights.fill(0.) and new parents (particle_filter.rs:109-114): what the last step did not store is dead
    h->sync_loop = log_total_weight != nullptr;
    int32_t rc = ensure_rows(h);
    if (rc != MP_OK) return rc;
    h->parents_deferred = false;   // this resample's parents replace whatever was still waiting to be read
    h->sh_parents_lazy = false;
    h->sh_recv = false;
    const int d = h->ops->dim_state;
    bool drawn_only = false;
    bool peek_L = false;   // the return value from k_peek_level1 (the draws stay pending)
    if (h->use_deferred) {
        // draws only: the lookups are done by whoever consumes the parents — the next k_propagate, under its arithmetic, or
        // k_resolve_slots when the host asks first
        if (!h->cx_alt) {
            HIPCK(hipMalloc(&h->cx_alt, sizeof(mp_cx) * (size_t)h->nt * TILE));
            HIPCK(hipMalloc(&h->guide_alt, sizeof(unsigned short) * (size_t)h->nt * GUIDE_N));
            if (h->local_table) HIPCK(hipMalloc(&h->tiles_alt, sizeof(u64) * 3 * h->nt));
            int32_t rct = update_k1_tail(h);
            if (rct != MP_OK) return rct;
        }
        const mp_tab tab = tab_of(h);
        // ... and for kernels whose lanes own one Philox block's two slots, not even the draws are made here: an asynchronous
        // multinomial resample enqueues NOTHING, the next k_propagate draws for its own slots (flush_draws() otherwise)
        // (up to 2048 tiles = 2^22 particles: the kernel's table, 24 B per tile, stays within the default dynamic-LDS limit)
        // (the lattice schemes too: their targets need no Philox block per lane at all)
        // (a SYNCHRONOUS resample too, when the handle has the host-mapped mirror: its return value comes from k_peek_level1 below)
        if (h->nt <= 2048 && (!log_total_weight || h->h_mirror) && h->use_fused_draws && h->ops->can_draw &&
            h->local_table && !(h->flags & MP_PF_RECORD_HISTORY)) {
            h->draw_pending = true;
            h->pending_rc = h->resample_count;
            h->pending_scheme = scheme;
            peek_L = log_total_weight != nullptr;
        } else {
            rc = launch_draws(h, scheme, h->resample_count);
            if (rc != MP_OK) return rc;
        }
        drawn_only = true;
    } else {
        LaunchTimer lt(h, MP_K_RESAMPLE_GATHER);
        if (scheme == MP_RESAMPLE_STRATIFIED) {
            hipLaunchKernelGGL(k_resample_gather<2>, dim3(h->k3_grid), dim3(KG_THREADS), table_lds(h->nt, KG_THREADS), h->stream, h->n, h->n,
                               h->n_global, h->slot_offset, (uint32_t)MP_DOM_RESAMPLE, (uint32_t)h->seed, (uint32_t)(h->seed >> 32), h->resample_count,
                               h->S, d, h->cx, h->guide, h->tile_m, h->tile_W, h->tile_W2, h->nt, h->x[h->cur], h->x[h->cur ^ 1], h->parent, h->logw,
                               h->scal);
        } else if (scheme == MP_RESAMPLE_SYSTEMATIC) {
            hipLaunchKernelGGL(k_resample_gather<1>, dim3(h->k3_grid), dim3(KG_THREADS), table_lds(h->nt, KG_THREADS), h->stream, h->n, h->n,
                               h->n_global, h->slot_offset, (uint32_t)MP_DOM_RESAMPLE, (uint32_t)h->seed, (uint32_t)(h->seed >> 32), h->resample_count,
                               h->S, d, h->cx, h->guide, h->tile_m, h->tile_W, h->tile_W2, h->nt, h->x[h->cur], h->x[h->cur ^ 1], h->parent, h->logw,
                               h->scal);
        } else {
            hipLaunchKernelGGL(k_resample_gather<0>, dim3(h->k3_grid), dim3(KG_THREADS), table_lds(h->nt, KG_THREADS), h->stream, h->n, h->n,
                               h->n_global, h->slot_offset, (uint32_t)MP_DOM_RESAMPLE, (uint32_t)h->seed, (uint32_t)(h->seed >> 32), h->resample_count,
                               h->S, d, h->cx, h->guide, h->tile_m, h->tile_W, h->tile_W2, h->nt, h->x[h->cur], h->x[h->cur ^ 1], h->parent, h->logw,
                               h->scal);
        }
    }
    rc = check_launch("resample kernels");
    if (rc != MP_OK) return rc;
    h->draws_lattice = scheme != MP_RESAMPLE_MULTINOMIAL;
    if (drawn_only) h->deferred = true;   // x[cur] is the (stale) pre-resample state until the draws are looked up
    else {
        h->cur ^= 1;
        h->x_in_rows = false;   // k_resample_gather wrote every slot's state into x[cur] (slot order): the rows are the OLD generation's
    }
    h->rows_fresh = false;            // the log-weights are now all zero
    h->resample_count += 1;
    if (h->flags & MP_PF_RECORD_HISTORY) {
        rc = materialize(h);
        if (rc != MP_OK) return rc;
        void* buf = nullptr;
        rc = hist_alloc(h, sizeof(uint32_t) * h->n, &buf);
        if (rc != MP_OK) return rc;
        HIPCK(hipMemcpyAsync(buf, h->parent, sizeof(uint32_t) * h->n, hipMemcpyDeviceToDevice, h->stream));
        h->hist.push_back({1, buf});
    }
    if (log_total_weight && peek_L) {
        // `resample() -> f64` without making the draws now: level 1 of the normalisation that is being resampled by one small
        // workgroup, its result in host-mapped memory; the draws, the lookups and the fold into log_ml happen inside the next
        // step's k_propagate as after an asynchronous resample (k_draw_slots + k_resolve_slots + a copy of the scalars before)
        if (!h->peek_valid) {   // (else: the step's own last workgroup has computed it, or is about to: mt_peek_tail)
            h->peek_seq += 1;
            hipLaunchKernelGGL(k_peek_level1, dim3(1), dim3(1024), 0, h->stream, (const double*)h->tile_m, (const u64*)h->tile_W, (const u64*)h->tile_W2, h->nt, h->S,
                               h->scal, h->d_mirror, h->peek_seq);
            rc = check_launch("k_peek_level1");
            if (rc != MP_OK) return rc;
        }
        h->peek_valid = false;
        rc = wait_seq(h, &h->h_mirror->peek_seq, h->peek_seq, "resample");
        if (rc != MP_OK) return rc;
        if (h->h_mirror->peek_degenerate)
            return mp_fail(MP_ERR_DEGENERATE, "all log-weights are -inf: normalized weights are NaN (categorical.rs:23 assert in the reference)");
        *log_total_weight = h->h_mirror->peek_L;
    } else if (log_total_weight) {
        rc = fetch_scalars(h);
        if (rc != MP_OK) return rc;
        *log_total_weight = h->h_scal->L;
    }
    return MP_OK;
}

static int32_t query(mp_pf* h) {
    if (h->sharded) return mp_fail(MP_ERR_STATE, "sharded handle: use mp_pf_shard_query");
    int32_t rc = ensure_rows(h);
    if (rc != MP_OK) return rc;
    hipLaunchKernelGGL(k_finalize_tiles, dim3(1), dim3(K3_THREADS), table_lds(h->nt, K3_THREADS), h->stream, h->tile_m, h->tile_W, h->tile_W2, h->nt,
                       h->S, h->n_global, 1, h->scal);
    rc = check_launch("k_finalize_tiles");
    if (rc != MP_OK) return rc;
    return fetch_scalars(h);
}

int32_t mp_pf_effective_sample_size(mp_pf* h, int32_t ess_mode, double* out) {
    if (!h || !out) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    HIPCK(hipSetDevice(h->device));
    if (ess_mode == MP_ESS_REFERENCE) {
        if (h->h_mirror) {
            // the value of the last resample's normalisation (particle_filter.rs:98-100 reads buffers only `resample` refreshes): in
            // host-mapped memory as soon as whoever makes that resample's draws has folded it — the first workgroup of the step
            // that follows an asynchronous resample does so in its first microseconds, so this does not wait for the step to end
            int32_t rc = flush_draws(h);
            if (rc != MP_OK) return rc;
            rc = wait_seq(h, &h->h_mirror->fold_seq, h->resample_count, "effective_sample_size");
            if (rc != MP_OK) return rc;
            *out = h->h_mirror->ess_stale;
            return MP_OK;
        }
        int32_t rc = fetch_scalars(h);
        if (rc != MP_OK) return rc;
        *out = h->h_scal->ess_stale;
        return MP_OK;
    }
    if (ess_mode != MP_ESS_FRESH) return mp_fail(MP_ERR_INVALID_ARG, "unknown ess mode");
    if (!h->initialised) return mp_fail(MP_ERR_STATE, "effective_sample_size(FRESH) before init_step");
    int32_t rc = query(h);
    if (rc != MP_OK) return rc;
    *out = h->h_scal->ess_fresh;
    return MP_OK;
}

// ESS-triggered (adaptive) resampling: the usual SMC policy on top of the reference's unconditional `resample`
// (tests/smc.rs:79-84 resamples every step).  Uses the CURRENT weights (MP_ESS_FRESH), one host round trip.
int32_t mp_pf_resample_if_ess_below(mp_pf* h, int32_t scheme, double ess_fraction, int32_t* resampled, double* ess_out, double* log_total_weight) {
    if (!h || !resampled) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    if (!(ess_fraction >= 0.) || ess_fraction > 1.) return mp_fail(MP_ERR_INVALID_ARG, "ess_fraction must be in [0, 1]");
    if (h->sharded) return mp_fail(MP_ERR_UNSUPPORTED, "sharded filters decide from mp_pf_shard_query_packed on every rank");
    double ess = 0.;
    int32_t rc = mp_pf_effective_sample_size(h, MP_ESS_FRESH, &ess);
    if (rc != MP_OK) return rc;
    if (ess_out) *ess_out = ess;
    *resampled = (ess < ess_fraction * (double)h->n) ? 1 : 0;
    if (!*resampled) return MP_OK;
    return mp_pf_resample(h, scheme, log_total_weight);
}

int32_t mp_pf_log_marginal_likelihood_estimate(mp_pf* h, double* out) {
    if (!h || !out) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    HIPCK(hipSetDevice(h->device));
    int32_t rc = query(h);
    if (rc != MP_OK) return rc;
    *out = h->h_scal->lml_fresh;
    return MP_OK;
}

int32_t mp_pf_read_state(mp_pf* h, double* x_out) {
    if (!h || !x_out) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    HIPCK(hipSetDevice(h->device));
    { int32_t rcm = materialize(h); if (rcm != MP_OK) return rcm; }
    { int32_t rcx = ensure_x(h); if (rcx != MP_OK) return rcx; }
    const int d = h->ops->dim_state;   // device layout = host layout: particle-major x[i][d]
    HIPCK(hipMemcpyAsync(x_out, h->x[h->cur], sizeof(double) * h->n * d, hipMemcpyDeviceToHost, h->stream));
    HIPCK(hipStreamSynchronize(h->stream));
    return MP_OK;
}

int32_t mp_pf_read_log_weights(mp_pf* h, double* out) {
    if (!h || !out) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    HIPCK(hipSetDevice(h->device));
    { int32_t rcm = materialize(h); if (rcm != MP_OK) return rcm; }
    { int32_t rcl = ensure_lazy(h); if (rcl != MP_OK) return rcl; }
    HIPCK(hipMemcpyAsync(out, h->logw, sizeof(double) * h->n, hipMemcpyDeviceToHost, h->stream));
    HIPCK(hipStreamSynchronize(h->stream));
    return MP_OK;
}

int32_t mp_pf_read_parents(mp_pf* h, uint32_t* out) {
    if (!h || !out) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    HIPCK(hipSetDevice(h->device));
    { int32_t rcm = materialize(h); if (rcm != MP_OK) return rcm; }
    { int32_t rcl = ensure_lazy(h); if (rcl != MP_OK) return rcl; }
    // a step after a lazy resample moved the states on but left the parents where the resample put them
    if (h->parents_deferred) {
        hipLaunchKernelGGL(k_resolve_slots<false>, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, h->stream, h->n, h->ops->dim_state,
                           (const u64*)h->dfr_lt, (const uint32_t*)h->dfr_row, (const mp_cx*)h->cx_alt, (const double*)nullptr, (double*)nullptr, h->parent,
                           (double*)nullptr, h->sh_recv ? h->sh_rows : (const double*)nullptr, h->slot_offset);
        h->parents_deferred = false;
        int32_t rcp = check_launch("k_resolve_slots");
        if (rcp != MP_OK) return rcp;
    }
    if (h->sh_parents_lazy) {
        hipLaunchKernelGGL(k_shard_adopt_parents, dim3((unsigned)((h->n + SH_THREADS - 1) / SH_THREADS)), dim3(SH_THREADS), 0, h->stream, h->n,
                           h->ops->dim_state, h->sh_rows, h->sh_req_slot, h->slot_offset, h->parent);
        h->sh_parents_lazy = false;
        int32_t rcp = check_launch("k_shard_adopt_parents");
        if (rcp != MP_OK) return rcp;
    }
    HIPCK(hipMemcpyAsync(out, h->parent, sizeof(uint32_t) * h->n, hipMemcpyDeviceToHost, h->stream));
    HIPCK(hipStreamSynchronize(h->stream));
    return MP_OK;
}

// ---- sharded phases ------------------------------------------------------------------------------

int32_t mp_pf_shard_tiles(mp_pf* h, double* d_tile_m, uint64_t* d_tile_W, uint64_t* d_tile_W2) {
    if (!h || !d_tile_m || !d_tile_W || !d_tile_W2) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    if (!h->initialised) return mp_fail(MP_ERR_STATE, "resample before init_step");
    HIPCK(hipSetDevice(h->device));
    int32_t rc = ensure_rows(h);
    if (rc != MP_OK) return rc;
    HIPCK(hipMemcpyAsync(d_tile_m, h->tile_m, sizeof(double) * h->nt, hipMemcpyDeviceToDevice, h->stream));
    HIPCK(hipMemcpyAsync(d_tile_W, h->tile_W, sizeof(u64) * h->nt, hipMemcpyDeviceToDevice, h->stream));
    HIPCK(hipMemcpyAsync(d_tile_W2, h->tile_W2, sizeof(u64) * h->nt, hipMemcpyDeviceToDevice, h->stream));
    return MP_OK;
}

int32_t mp_pf_shard_route(mp_pf* h, int32_t scheme, const double* d_tm_all, const uint64_t* d_tW_all, const uint64_t* d_tW2_all, int32_t world,
                          int32_t rank, uint64_t* d_req_out, int64_t* send_counts) {
    if (!h || !d_tm_all || !d_tW_all || !d_tW2_all || !d_req_out || !send_counts) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    if (scheme != MP_RESAMPLE_MULTINOMIAL && scheme != MP_RESAMPLE_SYSTEMATIC && scheme != MP_RESAMPLE_STRATIFIED)
        return mp_fail(MP_ERR_INVALID_ARG, "unknown resampling scheme");
    if (world < 1 || world > SH_MAX_WORLD || rank < 0 || rank >= world) return mp_fail(MP_ERR_INVALID_ARG, "1 <= world <= 64, 0 <= rank < world");
    if ((u64)world * h->n != h->n_global) return mp_fail(MP_ERR_INVALID_ARG, "equal tile-aligned shards: world * n_particles must equal n_global");
    HIPCK(hipSetDevice(h->device));
    const int nblk = (int)((h->n + SH_THREADS - 1) / SH_THREADS);
    const int nt_all = h->nt * world;
    {
        int32_t rcs = shard_scratch(h, world, h->sh_cap ? h->sh_cap : 1);
        if (rcs != MP_OK) return rcs;
    }
    {
        LaunchTimer lt(h, MP_K_RESAMPLE_GATHER);
        const size_t lds = table_lds(nt_all, SH_THREADS) + sizeof(uint32_t) * SH_MAX_WORLD;
        hipLaunchKernelGGL(k_shard_targets, dim3(nblk), dim3(SH_THREADS), lds, h->stream, h->n, h->n_global, h->slot_offset, (uint32_t)h->seed,
                           (uint32_t)(h->seed >> 32), h->resample_count, (int)scheme, h->S, d_tm_all, (const u64*)d_tW_all,
                           nt_all, h->nt, world, h->sh_dest, h->sh_lt, h->sh_tile, h->sh_blockcount);
        hipLaunchKernelGGL(k_shard_offsets, dim3(world), dim3(SH_THREADS), 0, h->stream, h->sh_blockcount, nblk, world, h->sh_blockoff, h->sh_counts);
        hipLaunchKernelGGL(k_shard_pack, dim3(nblk), dim3(SH_THREADS), 0, h->stream, h->n, h->sh_dest, h->sh_lt, h->sh_tile, h->sh_blockoff, h->sh_counts,
                           world, (u64*)d_req_out, h->sh_req_slot);
    }
    int32_t rc = check_launch("k_shard_targets/offsets/pack");
    if (rc != MP_OK) return rc;
    // the finalisation of this normalisation (L, ESS, log-ML) only needs the gathered tiles
    hipLaunchKernelGGL(k_finalize_tiles, dim3(1), dim3(K3_THREADS), table_lds(nt_all, K3_THREADS), h->stream, d_tm_all, (const u64*)d_tW_all,
                       (const u64*)d_tW2_all, nt_all, h->S, h->n_global, 0, h->scal);
    rc = check_launch("k_finalize_tiles");
    if (rc != MP_OK) return rc;
    HIPCK(hipMemcpyAsync(h->h_counts, h->sh_counts, sizeof(long long) * world, hipMemcpyDeviceToHost, h->stream));
    HIPCK(hipStreamSynchronize(h->stream));
    for (int r = 0; r < world; ++r) send_counts[r] = h->h_counts[r];
    return MP_OK;
}

int32_t mp_pf_shard_resolve(mp_pf* h, const uint64_t* d_req_in, uint64_t n_req, double* d_rows_out) {
    if (!h) return mp_fail(MP_ERR_INVALID_ARG, "null handle");
    HIPCK(hipSetDevice(h->device));
    // (before the early return: mp_pf_shard_scatter flips `cur` whether or not this rank was asked for rows, and a dim_state-1
    // handle's current states may live in the row table only)
    { int32_t rcx = ensure_x(h); if (rcx != MP_OK) return rcx; }
    if (n_req == 0) return MP_OK;
    if (!d_req_in || !d_rows_out) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    int grid = (int)std::min<u64>((n_req + K3_THREADS - 1) / K3_THREADS, (u64)K3_MAX_BLOCKS);
    {
        LaunchTimer lt(h, MP_K_RESAMPLE_GATHER);
        hipLaunchKernelGGL(k_shard_resolve, dim3(grid), dim3(K3_THREADS), 0, h->stream, h->n, (u64)n_req, h->slot_offset, h->ops->dim_state,
                           (const u64*)d_req_in, h->cx, h->guide, h->tile_W, h->x[h->cur], d_rows_out);
    }
    return check_launch("k_shard_resolve");
}

int32_t mp_pf_shard_scatter(mp_pf* h, const double* d_rows_in, double* log_total_weight) {
    if (!h || !d_rows_in) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    if (!h->sh_req_slot) return mp_fail(MP_ERR_STATE, "shard_scatter before shard_route");
    HIPCK(hipSetDevice(h->device));
    {
        LaunchTimer lt(h, MP_K_RESAMPLE_GATHER);
        hipLaunchKernelGGL(k_shard_scatter, dim3((unsigned)((h->n + SH_THREADS - 1) / SH_THREADS)), dim3(SH_THREADS), 0, h->stream, h->n,
                           h->ops->dim_state, d_rows_in, h->sh_req_slot, h->x[h->cur ^ 1], h->parent, h->logw);
    }
    int32_t rc = check_launch("k_shard_scatter");
    if (rc != MP_OK) return rc;
    h->parents_deferred = false;   // k_shard_scatter wrote parent[]
    h->sh_parents_lazy = false;
    h->cur ^= 1;
    h->x_in_rows = false;          // ... and every slot's state into x[cur]
    h->rows_fresh = false;
    h->resample_count += 1;
    if (log_total_weight) {
        rc = fetch_scalars(h);
        if (rc != MP_OK) return rc;
        *log_total_weight = h->h_scal->L;
    }
    return MP_OK;
}

int32_t mp_pf_shard_query(mp_pf* h, const double* d_tm_all, const uint64_t* d_tW_all, const uint64_t* d_tW2_all, int32_t world, double* log_ml,
                          double* ess) {
    if (!h || !d_tm_all || !d_tW_all || !d_tW2_all) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    HIPCK(hipSetDevice(h->device));
    const int nt_all = h->nt * world;
    hipLaunchKernelGGL(k_finalize_tiles, dim3(1), dim3(K3_THREADS), table_lds(nt_all, K3_THREADS), h->stream, d_tm_all, (const u64*)d_tW_all,
                       (const u64*)d_tW2_all, nt_all, h->S, h->n_global, 1, h->scal);
    int32_t rc = check_launch("k_finalize_tiles");
    if (rc != MP_OK) return rc;
    rc = fetch_scalars(h);
    if (rc != MP_OK) return rc;
    if (log_ml) *log_ml = h->h_scal->lml_fresh;
    if (ess) *ess = h->h_scal->ess_fresh;
    return MP_OK;
}


// ---- fixed-capacity phases: nothing here synchronises with the host -------------------------------------
static int32_t shard_scratch(mp_pf* h, int world, u64 cap) {
    const int nblk = (int)((h->n + SH_THREADS - 1) / SH_THREADS);
    if (h->sh_dest && h->sh_world >= world && h->sh_cap >= cap && h->sh_tm_all) return MP_OK;
    (void)hipFree(h->sh_dest); (void)hipFree(h->sh_lt); (void)hipFree(h->sh_tile); (void)hipFree(h->sh_req_slot); (void)hipFree(h->sh_blockcount);
    (void)hipFree(h->sh_blockoff); (void)hipFree(h->sh_counts); (void)hipFree(h->sh_tm_all); (void)hipFree(h->sh_tW_all); (void)hipFree(h->sh_tW2_all); (void)hipFree(h->sh_incl_all); (void)hipFree(h->sh_ratio_all);
    (void)hipFree(h->sh_overflow); (void)hipFree(h->sh_done); (void)hipFree(h->sh_tab_part); (void)hipFree(h->sh_tab_ticket);
    if (h->h_counts) (void)hipHostFree(h->h_counts);
    const u64 slots = std::max<u64>(h->n, (u64)world * SH_BINS * cap);
    HIPCK(hipMalloc(&h->sh_dest, h->n));
    HIPCK(hipMalloc(&h->sh_lt, sizeof(u64) * h->n));
    HIPCK(hipMalloc(&h->sh_tile, sizeof(uint32_t) * h->n));
    HIPCK(hipMalloc(&h->sh_req_slot, sizeof(uint32_t) * slots));
    HIPCK(hipMemsetAsync(h->sh_req_slot, 0, sizeof(uint32_t) * slots, h->stream));   // never an out-of-range row index, whatever path leaves it unwritten
    HIPCK(hipMalloc(&h->sh_blockcount, sizeof(uint32_t) * (size_t)nblk * world));
    HIPCK(hipMalloc(&h->sh_blockoff, sizeof(uint32_t) * (size_t)nblk * world));
    HIPCK(hipMalloc(&h->sh_counts, sizeof(long long) * SH_MAX_KEYS));
    HIPCK(hipMalloc(&h->sh_tm_all, sizeof(double) * (size_t)h->nt * world));
    HIPCK(hipMalloc(&h->sh_tW_all, sizeof(u64) * (size_t)h->nt * world));
    HIPCK(hipMalloc(&h->sh_tW2_all, sizeof(u64) * (size_t)h->nt * world));
    HIPCK(hipMalloc(&h->sh_incl_all, sizeof(u64) * (size_t)h->nt * world));
    HIPCK(hipMalloc(&h->sh_ratio_all, sizeof(double) * (size_t)h->nt * world));
    HIPCK(hipMalloc(&h->sh_overflow, sizeof(int)));
    HIPCK(hipMalloc(&h->sh_done, 2 * sizeof(unsigned int)));
    HIPCK(hipMalloc(&h->sh_tab_part, sizeof(mp_tab_part) * SH_MAX_WORLD));
    HIPCK(hipMalloc(&h->sh_tab_ticket, sizeof(unsigned int)));
    HIPCK(hipMemsetAsync(h->sh_tab_ticket, 0, sizeof(unsigned int), h->stream));
    h->sh_tab_seq = 0;
    HIPCK(hipMemsetAsync(h->sh_done, 0, 2 * sizeof(unsigned int), h->stream));
    if (!h->scal_undo) HIPCK(hipMalloc(&h->scal_undo, sizeof(mp_dev_scalars)));
    HIPCK(hipMemsetAsync(h->sh_overflow, 0, sizeof(int), h->stream));
    HIPCK(hipHostMalloc(&h->h_counts, sizeof(long long) * SH_MAX_WORLD));
    if (!h->h_pub) {
        HIPCK(hipHostMalloc(&h->h_pub, sizeof(mp_shard_pub), hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(h->h_pub, 0, sizeof(mp_shard_pub));
        HIPCK(hipHostGetDevicePointer((void**)&h->d_pub, h->h_pub, 0));
        HIPCK(hipEventCreateWithFlags(&h->ev_resolved, hipEventDisableTiming));
    }
    h->sh_world = world;
    h->sh_cap = cap;
    return MP_OK;
}

int32_t mp_pf_shard_bind_tiles(mp_pf* h, uint64_t* d_tiles) {
    if (!h || !d_tiles) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    HIPCK(hipSetDevice(h->device));
    HIPCK(hipMemcpyAsync(d_tiles, h->tile_m, sizeof(u64) * 3 * h->nt, hipMemcpyDeviceToDevice, h->stream));
    h->local_table = false;   // (a handle driven through the shard phases: its tile scalars live in the caller's buffer, not in a pair of the library's)
    h->tile_m = reinterpret_cast<double*>(d_tiles);
    h->tile_W = (u64*)d_tiles + h->nt;
    h->tile_W2 = (u64*)d_tiles + 2 * (size_t)h->nt;
    return update_k1_tail(h);
}

int32_t mp_pf_shard_tiles_packed(mp_pf* h, uint64_t* d_tiles_out) {
    if (!h || !d_tiles_out) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    if (!h->initialised) return mp_fail(MP_ERR_STATE, "resample before init_step");
    HIPCK(hipSetDevice(h->device));
    int32_t rc = ensure_rows(h);
    if (rc != MP_OK) return rc;
    if ((void*)d_tiles_out != (void*)h->tile_m)   // a bound buffer already holds them
        HIPCK(hipMemcpyAsync(d_tiles_out, h->tile_m, sizeof(u64) * 3 * h->nt, hipMemcpyDeviceToDevice, h->stream));
    return MP_OK;
}

// the job's tile table from the gathered tiles: one workgroup per rank in a world of several (k_shard_table_mw), one in all otherwise;
// `plan` / `place`: what the table's leading workgroup does on its way out for a self-drawn resample (mp_pf_shard_kernels.h)
static void launch_shard_table(mp_pf* h, const u64* d_tiles_all, int world, unsigned long long* c_all, int scheme, int rank, mp_own_range* range,
                               u64* kthr, const mp_own_plan_args* plan = nullptr, const mp_self_place_args* place = nullptr) {
    mp_table_tail tail{};
    tail.c_all = c_all; tail.scheme = scheme; tail.rank = rank;
    tail.k0 = (uint32_t)h->seed; tail.k1 = (uint32_t)(h->seed >> 32); tail.rc = h->resample_count;
    tail.range = range; tail.kthr = kthr;
    if (plan) { tail.plan = *plan; tail.do_plan = 1; }
    if (plan && place) { tail.place = *place; tail.do_place = 1; }
    static const bool one_wg = mp_diag_env("MP_SHARD_TABLE_ONE_WG") && atoi(mp_diag_env("MP_SHARD_TABLE_ONE_WG")) != 0;   // A/B
    // (measured, 512 tiles per rank: one workgroup 8.3 us up to 4 ranks and 16.9 us at 8; one per rank 9.8 - 10.8 us at 2 .. 8)
    // (MP_SHARD_TABLE_MW_TILES: the job size from which one workgroup per rank builds the table — tests lower it to reach that
    // kernel with a few thousand particles)
    const char* mw_env = mp_diag_env("MP_SHARD_TABLE_MW_TILES");   // (read per call: a resample, not a hot loop)
    const size_t mw_tiles = mw_env ? (size_t)atoll(mw_env) : (size_t)2048;
    if (world > 1 && !one_wg && (size_t)world * (size_t)h->nt > mw_tiles) {
        h->sh_tab_seq += (unsigned)world;
        auto kern = h->nt <= SHT_THREADS ? k_shard_table_mw<1> : k_shard_table_mw<SHT_PER>;
        hipLaunchKernelGGL(kern, dim3(world), dim3(SHT_THREADS), 0, h->stream, d_tiles_all, world, h->nt, h->S, h->n_global, h->sh_tm_all,
                           h->sh_tW_all, h->sh_tW2_all, h->sh_incl_all, h->sh_ratio_all, h->sh_counts, h->scal, h->scal_undo, h->sh_tab_part,
                           h->sh_tab_ticket, h->sh_tab_seq, tail);
    } else {
        auto kern = (size_t)world * (size_t)h->nt <= 2 * (size_t)SHT_THREADS ? k_shard_table<2> : k_shard_table<SHT_PER>;
        hipLaunchKernelGGL(kern, dim3(1), dim3(SHT_THREADS), 0, h->stream, d_tiles_all, world, h->nt, h->S, h->n_global, h->sh_tm_all,
                           h->sh_tW_all, h->sh_tW2_all, h->sh_incl_all, h->sh_ratio_all, h->sh_counts, h->scal, h->scal_undo, tail);
    }
}

int32_t mp_pf_shard_route_fixed(mp_pf* h, int32_t scheme, const uint64_t* d_tiles_all, int32_t world, int32_t rank, uint64_t capacity,
                                uint64_t* d_req_out) {
    if (!h || !d_tiles_all || !d_req_out) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    if (scheme != MP_RESAMPLE_MULTINOMIAL && scheme != MP_RESAMPLE_SYSTEMATIC && scheme != MP_RESAMPLE_STRATIFIED)
        return mp_fail(MP_ERR_INVALID_ARG, "unknown resampling scheme");
    if (world < 1 || world > SH_MAX_WORLD || rank < 0 || rank >= world) return mp_fail(MP_ERR_INVALID_ARG, "1 <= world <= 64, 0 <= rank < world");
    if ((u64)world * h->n != h->n_global) return mp_fail(MP_ERR_INVALID_ARG, "equal tile-aligned shards: world * n_particles must equal n_global");
    if (capacity == 0) return mp_fail(MP_ERR_INVALID_ARG, "capacity must be > 0");
    if ((u64)world * SH_BINS * capacity >= (1ull << 31)) return mp_fail(MP_ERR_INVALID_ARG, "exchange buffer rows must be < 2^31 (row indices carry a flag bit)");
    HIPCK(hipSetDevice(h->device));
    int32_t rc = shard_scratch(h, world, capacity);
    if (rc != MP_OK) return rc;
    const int nt_all = h->nt * world;
    {
        LaunchTimer lt(h, MP_K_BIN_DRAWS);
        launch_shard_table(h, (const u64*)d_tiles_all, world, nullptr, 0, 0, nullptr, nullptr);
        const int nblk_f = (int)((h->n + SH_THREADS * SHF_ITEMS - 1) / (SH_THREADS * SHF_ITEMS));
        hipLaunchKernelGGL(k_shard_route_fused, dim3(nblk_f), dim3(SH_THREADS), 0, h->stream, h->n, h->n_global, h->slot_offset, (uint32_t)h->seed,
                           (uint32_t)(h->seed >> 32), h->resample_count, (int)scheme, (const u64*)h->sh_incl_all, (const u64*)h->sh_tW_all, (const double*)h->sh_ratio_all,
                           nt_all, h->nt, world, (u64)capacity, (unsigned long long*)h->sh_counts, (u64*)d_req_out, h->sh_req_slot, h->sh_done,
                           h->sh_overflow);
    }
    return check_launch("shard_route_fixed kernels");
}

int32_t mp_pf_shard_resolve_fixed(mp_pf* h, const uint64_t* d_req_in, int32_t world, uint64_t capacity, double* d_rows_out) {
    if (!h || !d_req_in || !d_rows_out) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    if (world < 1 || world > SH_MAX_WORLD || capacity == 0) return mp_fail(MP_ERR_INVALID_ARG, "1 <= world <= 64, capacity > 0");
    if (!h->h_pub) return mp_fail(MP_ERR_STATE, "shard_resolve_fixed before shard_route_fixed");
    HIPCK(hipSetDevice(h->device));
    { int32_t rcx = ensure_x(h); if (rcx != MP_OK) return rcx; }
    // workgroups per (asking rank, eighth): enough to fill the chip, each takes K3_THREADS * K3_ITEMS requests per round
    const u64 per = (u64)K3_THREADS * SHR_ITEMS;
    const unsigned groups = (unsigned)std::max<u64>(1, std::min<u64>((capacity + per - 1) / per, std::max<u64>(1, 512 / (u64)world)));
    {
        LaunchTimer lt(h, MP_K_RESAMPLE_GATHER);
        hipLaunchKernelGGL(k_shard_resolve_binned, dim3(groups * SH_BINS, world), dim3(K3_THREADS), 0, h->stream, h->n, (u64)capacity, h->slot_offset,
                           h->ops->dim_state, (const u64*)d_req_in, h->cx, h->guide, h->tile_W, h->x[h->cur], d_rows_out, h->sh_overflow);
        hipLaunchKernelGGL(k_shard_publish, dim3(1), dim3(1), 0, h->stream, h->sh_overflow, h->scal, h->d_pub);
    }
    int32_t rc = check_launch("k_shard_resolve_binned");
    if (rc != MP_OK) return rc;
    HIPCK(hipEventRecord(h->ev_resolved, h->stream));   // mp_pf_shard_commit_fixed waits for this, not for what follows on the stream
    return MP_OK;
}

int32_t mp_pf_shard_commit_fixed(mp_pf* h, const double* d_rows_in, double* log_total_weight) {
    if (!h || !d_rows_in) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    if (!h->h_pub) return mp_fail(MP_ERR_STATE, "shard_commit_fixed before shard_resolve_fixed");
    HIPCK(hipSetDevice(h->device));
    // The one host wait of this form, and only for the resolve: the all-to-all that carries the rows back may still be
    // running.  Did any sub-segment exceed the capacity?  (Every rank reaches the same answer: senders flag it in the
    // request headers.)  If so nothing is committed; the caller repeats this resample with the variable-size phases,
    // which read the same rows, tiles and Philox counters.
    HIPCK(hipEventSynchronize(h->ev_resolved));
    if (h->h_pub->overflow) {
        HIPCK(hipMemsetAsync(h->sh_overflow, 0, sizeof(int), h->stream));
        HIPCK(hipMemcpyAsync(h->scal, h->scal_undo, sizeof(mp_dev_scalars), hipMemcpyDeviceToDevice, h->stream));  // un-fold the log-ML increment
        return mp_fail(MP_ERR_CAPACITY, "sharded exchange: a sub-segment needs more than `capacity` draws; repeat with the variable-size phases");
    }
    // all weights -inf: no rank owns a draw, the request slots were never written — nothing may be committed (the next
    // propagate would read rows[inv[i]] out of bounds); the reference panics here (categorical.rs:23)
    if (h->h_pub->degenerate)
        return mp_fail(MP_ERR_DEGENERATE, "all log-weights are -inf: normalized weights are NaN (categorical.rs:23 assert in the reference)");
    // traces[i] = traces[parents[i]].clone() (particle_filter.rs:109-113), lazily: the next propagate reads slot i's state
    // from row sh_req_slot[i] of the exchange buffer; anything else first copies states and parents into slot order.
    h->sh_rows = d_rows_in;
    h->sh_lazy = true;
    h->sh_recv = false;
    h->sh_parents_lazy = false;
    h->logw_zero = true;   // log_weights.fill(0.) (:114): the next propagate does not re-read them; anything else clears the buffer first
    h->cur ^= 1;
    h->rows_fresh = false;
    h->resample_count += 1;
    if (log_total_weight) {
        if (h->h_pub->degenerate)
            return mp_fail(MP_ERR_DEGENERATE, "all log-weights are -inf: normalized weights are NaN (categorical.rs:23 assert in the reference)");
        *log_total_weight = h->h_pub->L;
    }
    return MP_OK;
}

// ---- "owner keeps" form: offspring stay with the rank that owns their parent; only the surplus travels ----
static void owned_free(mp_pf* h) {
    (void)hipFree(h->ow_seg_lt); (void)hipFree(h->ow_seg_row);
    (void)hipFree(h->ow_sccnt); (void)hipFree(h->ow_base); (void)hipFree(h->ow_call); (void)hipFree(h->ow_plan);
    (void)hipFree(h->ow_range); (void)hipFree(h->ow_kthr); (void)hipFree(h->ow_range_solo);
    h->ow_range_solo = nullptr;
    h->ow_kthr = nullptr;
    h->ow_seg_lt = nullptr; h->ow_seg_row = nullptr;
    h->ow_sccnt = nullptr; h->ow_base = nullptr; h->ow_call = nullptr; h->ow_plan = nullptr; h->ow_range = nullptr;
}
// super-chunk shape of one resample: the multinomial draws of a rank are spread over all N draws, so a workgroup takes
// min(world, 4) rounds of 1024 to collect ~1024 own ones; under a lattice scheme a rank's own draws are one contiguous range
static void owned_shape(mp_pf* h, int world, int scheme) {
    static const int r_env = mp_diag_env("MP_OWN_ROUNDS") ? atoi(mp_diag_env("MP_OWN_ROUNDS")) : 0;   // A/B
    h->ow_R = scheme ? 1 : (r_env > 0 ? std::min(r_env, mp_own_rounds(world)) : mp_own_rounds(world));
    const u64 Wd = (u64)h->ow_R * OWN_ROUND;
    h->ow_nsc = (int)((h->n_global + Wd - 1) / Wd);
}
static int32_t owned_scratch(mp_pf* h, int world) {
    if (h->ow_seg_lt && h->ow_world == world) return MP_OK;
    if (h->ow_seg_lt) {
        HIPCK(hipStreamSynchronize(h->stream));
        owned_free(h);
    }
    // Windows cover every draw of the job (a rank may own any of them), touched only where this rank owns draws:
    // 16 B x n_global of address space per rank, ~16 B x n_local of it used per resample.
    // (sized for single-round super-chunks, the most there can be; owned_shape() picks the rounds per resampling scheme)
    const int nsc1 = (int)((h->n_global + OWN_ROUND - 1) / OWN_ROUND);
    const size_t ent = (size_t)nsc1 * OWN_ROUND;
    h->ow_nsc = nsc1;
    HIPCK(hipMalloc(&h->ow_seg_lt, sizeof(u64) * (world == 1 ? 2 : ent)));        // (a world of one writes its draws straight into the slot-order arrays)
    HIPCK(hipMalloc(&h->ow_seg_row, sizeof(uint32_t) * (world == 1 ? 2 : ent)));
    // what the next k_propagate (or k_resolve_slots) looks the kept offspring up from, and the second table it writes meanwhile:
    // the deferred-lookup machinery of the unsharded resample
    if (!h->dfr_lt) {
        HIPCK(hipMalloc(&h->dfr_lt, sizeof(u64) * (size_t)h->nchunks * DRAW_CHUNK));
        HIPCK(hipMalloc(&h->dfr_row, sizeof(uint32_t) * (size_t)h->nchunks * DRAW_CHUNK));
    }
    if (!h->cx_alt) {
        HIPCK(hipMalloc(&h->cx_alt, sizeof(mp_cx) * (size_t)h->nt * TILE));
        HIPCK(hipMalloc(&h->guide_alt, sizeof(unsigned short) * (size_t)h->nt * GUIDE_N));
        if (h->local_table) HIPCK(hipMalloc(&h->tiles_alt, sizeof(u64) * 3 * h->nt));
        int32_t rct = update_k1_tail(h);
        if (rct != MP_OK) return rct;
    }
    HIPCK(hipMalloc(&h->ow_sccnt, sizeof(uint32_t) * (size_t)h->ow_nsc));
    HIPCK(hipMalloc(&h->ow_base, sizeof(uint32_t) * (size_t)h->ow_nsc));
    HIPCK(hipMalloc(&h->ow_call, sizeof(unsigned long long) * SH_MAX_WORLD));
    HIPCK(hipMalloc(&h->ow_plan, sizeof(mp_owned_plan)));
    HIPCK(hipMalloc(&h->ow_range, sizeof(mp_own_range)));
    HIPCK(hipMalloc(&h->ow_kthr, sizeof(u64) * SH_MAX_WORLD));
    HIPCK(hipMemsetAsync(h->ow_call, 0, sizeof(unsigned long long) * SH_MAX_WORLD, h->stream));
    HIPCK(hipMemsetAsync(h->ow_plan, 0, sizeof(mp_owned_plan), h->stream));
    if (world == 1) {   // what k_shard_own_plan would find, every time (it is not launched in a world of one)
        const mp_own_range all{0ull, h->n};
        HIPCK(hipMalloc(&h->ow_range_solo, sizeof(mp_own_range)));
        HIPCK(hipMemcpyAsync(h->ow_range_solo, &all, sizeof(all), hipMemcpyHostToDevice, h->stream));
        std::vector<uint32_t> base((size_t)nsc1);
        for (int k = 0; k < nsc1; ++k) base[(size_t)k] = (uint32_t)((u64)k * OWN_ROUND);
        HIPCK(hipMemcpyAsync(h->ow_base, base.data(), sizeof(uint32_t) * (size_t)nsc1, hipMemcpyHostToDevice, h->stream));
        const unsigned long long c0 = h->n;
        HIPCK(hipMemcpyAsync(h->ow_call, &c0, sizeof(c0), hipMemcpyHostToDevice, h->stream));
        HIPCK(hipStreamSynchronize(h->stream));   // the host buffers above go out of scope
    }
    h->ow_world = world;
    return MP_OK;
}

// The host's one wait per owner-keeps resample: until the plan of resample number ow_seq has written its verdict word
// (host-mapped memory, polled: no event, so no end-of-kernel cache write-back in the stream and no driver wake-up latency).
static int32_t owned_wait_plan(mp_pf* h, unsigned* flags) {
    volatile unsigned long long* v = &h->h_pub->verdict;
    for (unsigned spins = 0;; ++spins) {
        const unsigned long long w = *v;
        if ((w >> 8) == h->ow_seq) { *flags = (unsigned)(w & 0xFFu); return MP_OK; }
        if ((spins & 1023u) == 1023u) {
            const hipError_t e = hipStreamQuery(h->stream);
            if (e == hipSuccess) {   // everything has run: the verdict is there, or it never will be
                const unsigned long long w2 = *v;
                if ((w2 >> 8) == h->ow_seq) { *flags = (unsigned)(w2 & 0xFFu); return MP_OK; }
                return mp_fail(MP_ERR_HIP, "owner-keeps plan: the stream drained without a verdict");
            }
            if (e != hipErrorNotReady) return mp_fail(MP_ERR_HIP, std::string("owner-keeps plan: ") + hipGetErrorString(e));
        }
    }
}

// the next k_propagate can make a self-drawn resample's kept draws itself (its SHD form: lanes of two adjacent slots, one table
// entry per thread); MP_FUSED_DRAWS=0 keeps them in a launch of their own
static bool self_k1_draws(const mp_pf* h) {
    return h->use_fused_draws && h->ops->can_draw && h->nt <= 1024 && !(h->flags & MP_PF_RECORD_HISTORY);
}
// the kept draws of a self-drawn resample (number rc) into the slot-order arrays: {tile-local target, start row} for slots [0, min(c_me, n))
static int32_t launch_self_draws(mp_pf* h, int scheme, uint32_t rc, int world, int rank) {
    const bool solo = world == 1;
    const size_t off = (size_t)rank * (size_t)h->nt;
    const u64* incl = solo ? (const u64*)h->tab_incl : (const u64*)h->sh_incl_all + off;
    const u64* W = solo ? (const u64*)h->tab_W : (const u64*)h->sh_tW_all + off;
    const double* ratio = solo ? (const double*)h->tab_ratio : (const double*)h->sh_ratio_all + off;
    const mp_own_range* range = solo ? h->ow_range_solo : h->ow_range;
    const bool tab_lds = h->nt <= 2048;
    const size_t lds = tab_lds ? (size_t)h->nt * 24 : 0;
    const bool lat = scheme != MP_RESAMPLE_MULTINOMIAL_SPLIT;
    auto kern = lat ? (tab_lds ? k_shard_self_draw<1, true> : k_shard_self_draw<2, true>) : (tab_lds ? k_shard_self_draw<1, false> : k_shard_self_draw<2, false>);
    const int grid = (int)std::min<u64>((h->n + 2 * SELF_THREADS - 1) / (2 * SELF_THREADS), 1024);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(SELF_THREADS), lds, h->stream, h->n, h->n_global, (uint32_t)h->seed, (uint32_t)(h->seed >> 32), rc, scheme,
                       incl, W, ratio, h->nt, world, rank, (const unsigned short*)h->guide, range, h->dfr_lt, h->dfr_row);
    return check_launch("k_shard_self_draw");
}

// what the placement of a self-drawn resample reads and writes (k_shard_self_place, or the table kernel's leading workgroup)
static mp_self_place_args self_place_args(const mp_pf* h, int rank, uint64_t capacity, double* d_send_out) {
    const size_t off = (size_t)rank * (size_t)h->nt;
    mp_self_place_args a{};
    a.n = h->n; a.slot_offset = h->slot_offset; a.cap = (u64)capacity; a.D = h->ops->dim_state;
    a.incl_sl = (const u64*)h->sh_incl_all + off; a.W_sl = (const u64*)h->sh_tW_all + off; a.ratio_sl = (const double*)h->sh_ratio_all + off;
    a.guide = (const unsigned short*)h->guide; a.cx = (const mp_cx*)h->cx; a.x = (const double*)h->x[h->cur];
    a.send = d_send_out; a.dfr_lt = h->dfr_lt; a.dfr_row = h->dfr_row;
    return a;
}

// fuse_send: the caller will expand with this equal-split send buffer next (mp_pf_shard_owned_count_expand) — a self-drawn resample
// then places in the table's launch
static int32_t shard_owned_count_impl(mp_pf* h, int32_t scheme, const uint64_t* d_tiles_all, int32_t world, int32_t rank, uint64_t capacity,
                                      uint64_t* counts_out, double* fuse_send) {
    if (!h || !d_tiles_all) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    if (!h->initialised) return mp_fail(MP_ERR_STATE, "resample before init_step");
    if (scheme != MP_RESAMPLE_MULTINOMIAL && scheme != MP_RESAMPLE_SYSTEMATIC && scheme != MP_RESAMPLE_STRATIFIED && scheme != MP_RESAMPLE_MULTINOMIAL_SPLIT)
        return mp_fail(MP_ERR_INVALID_ARG, "unknown resampling scheme");
    if (world < 1 || world > SH_MAX_WORLD || rank < 0 || rank >= world) return mp_fail(MP_ERR_INVALID_ARG, "1 <= world <= 64, 0 <= rank < world");
    if ((u64)world * h->n != h->n_global) return mp_fail(MP_ERR_INVALID_ARG, "equal tile-aligned shards: world * n_particles must equal n_global");
    if (h->n_global >= (1ull << 32)) return mp_fail(MP_ERR_INVALID_ARG, "owner-keeps exchange: n_global must be < 2^32");
    HIPCK(hipSetDevice(h->device));
    int32_t rc = ensure_rows(h);
    if (rc != MP_OK) return rc;
    rc = shard_scratch(h, world, h->sh_cap ? h->sh_cap : 1);
    if (rc != MP_OK) return rc;
    rc = owned_scratch(h, world);
    if (rc != MP_OK) return rc;
    h->ow_rank = rank;
    h->ow_self = false;
    h->ow_placed = false;
    h->ow_solo_folded = false;
    // Self-drawn form (mp_pf_shard_kernels.h): the lattice schemes (unless MP_SHARD_SELF=0) and the split multinomial.  A world of
    // one needs the table the last level-0 launch left (ensure_table); without it the lattice schemes keep the window form and
    // the split multinomial runs as the plain one (in a world of one the two make the same draws, bit for bit).
    bool self = (scheme == MP_RESAMPLE_MULTINOMIAL_SPLIT) || (scheme != MP_RESAMPLE_MULTINOMIAL && h->use_shard_self);
    if (self && world == 1 && !((const void*)d_tiles_all == (const void*)h->tile_m && ensure_table(h))) self = false;
    if (!self && scheme == MP_RESAMPLE_MULTINOMIAL_SPLIT) {
        if (world > 1) return mp_fail(MP_ERR_STATE, "split multinomial resample: the self-drawn form is not available on this handle");
        scheme = MP_RESAMPLE_MULTINOMIAL;
    }
    if (self) {
        h->ow_self = true;
        h->ow_scheme = scheme;
        if (world > 1) {
            LaunchTimer lt(h, MP_K_BIN_DRAWS);
            // the job's table, the offspring per rank (lattice: closed form; split: the binomial tree) and this rank's own range; then
            // the exchange plan from the counts alone
            mp_own_plan_args pa;
            pa.n = h->n; pa.n_global = h->n_global; pa.cap = (u64)capacity;
            pa.world = world; pa.nsc = 0; pa.lattice = 0; pa.S = h->S;
            pa.sccnt = h->ow_sccnt; pa.c_all = h->ow_call;
            pa.scal = h->scal; pa.undo = h->scal_undo; pa.head = nullptr;
            pa.base = h->ow_base; pa.plan_out = h->ow_plan; pa.pub = h->d_pub; pa.seq = ++h->ow_seq;
            pa.range = h->ow_range; pa.Wd = (u64)OWN_ROUND;
            // (the plan needs nothing but the counts: the table kernel's leading workgroup makes it on its way out — no launch of its
            // own; with an equal-split send buffer known now, and the kept draws left to the next k_propagate, it places too)
            const bool fuse = fuse_send && capacity > 0 && self_k1_draws(h);
            const mp_self_place_args pl = self_place_args(h, rank, capacity, fuse_send);
            launch_shard_table(h, (const u64*)d_tiles_all, world, h->ow_call, (int)scheme, rank, h->ow_range, h->ow_kthr, &pa, fuse ? &pl : nullptr);
            h->ow_placed = fuse;
            h->ow_placed_cap = capacity;
            h->ow_placed_send = fuse_send;
        }
        rc = check_launch("self-drawn resample: table / plan");
        if (rc != MP_OK) return rc;
        if (counts_out && world == 1) {
            counts_out[0] = h->n;
        } else if (counts_out) {
            HIPCK(stream_wait(h->stream));
            if (h->h_pub->degenerate)
                return mp_fail(MP_ERR_DEGENERATE, "all log-weights are -inf: normalized weights are NaN (categorical.rs:23 assert in the reference)");
            for (int r = 0; r < world; ++r) counts_out[r] = h->h_pub->counts[r];
        }
        return MP_OK;
    }
    owned_shape(h, world, scheme);
    {
        LaunchTimer lt(h, MP_K_BIN_DRAWS);
        // A world of one: the job's tile table is the one the last workgroup of k_propagate / k_normalize_tiles built (as for
        // the unsharded resample); otherwise one workgroup builds it from the gathered tiles.
        const bool solo_tab = world == 1 && scheme == MP_RESAMPLE_MULTINOMIAL && (const void*)d_tiles_all == (const void*)h->tile_m && ensure_table(h);
        if (!solo_tab) launch_shard_table(h, (const u64*)d_tiles_all, world, h->ow_call, (int)scheme, rank, h->ow_range, h->ow_kthr);
        const u64* t_incl = solo_tab ? (const u64*)h->tab_incl : (const u64*)h->sh_incl_all;
        const u64* t_W = solo_tab ? (const u64*)h->tile_W : (const u64*)h->sh_tW_all;
        const double* t_ratio = solo_tab ? (const double*)h->tab_ratio : (const double*)h->sh_ratio_all;
        mp_own_plan_args pa;
        pa.n = h->n; pa.n_global = h->n_global; pa.cap = (u64)capacity;
        pa.world = world; pa.nsc = h->ow_nsc; pa.lattice = scheme ? 1 : 0; pa.S = h->S;
        pa.sccnt = h->ow_sccnt; pa.c_all = h->ow_call;
        pa.scal = h->scal; pa.undo = h->scal_undo; pa.head = solo_tab ? (const mp_tab_head*)h->tab_head : nullptr;
        pa.base = h->ow_base; pa.plan_out = h->ow_plan; pa.pub = h->d_pub; pa.seq = ++h->ow_seq;
        pa.range = h->ow_range; pa.Wd = (u64)h->ow_R * OWN_ROUND;
        const bool tab_lds = h->nt <= 1024;   // (with the 32 KB of own targets at R = 4 this stays under the 64 KB a launch may ask for)
        const size_t lds = sizeof(u64) * (size_t)h->ow_R * OWN_ROUND + (tab_lds ? (size_t)h->nt * 24 : 0);
        auto kern = scheme ? (tab_lds ? k_shard_own_draw<1, true> : k_shard_own_draw<2, true>) : (tab_lds ? k_shard_own_draw<1, false> : k_shard_own_draw<2, false>);
        // a world of one: every draw is this rank's own, super-chunks are 1024 consecutive slots — the window IS slot order
        u64* win_lt = world == 1 ? h->dfr_lt : h->ow_seg_lt;
        uint32_t* win_row = world == 1 ? h->dfr_row : h->ow_seg_row;
        // lattice: a rank's own draws are ~n consecutive ones, wherever they start: that many workgroups (a rank that owns more takes turns)
        // multinomial: resident workgroups (3 per CU with the 44 KB of LDS at R = 4) take turns over the super-chunks, so the LDS table is
        // filled once per workgroup and the per-rank counts leave as `world` atomics per workgroup
        static const int mn_wgs = mp_diag_env("MP_OWN_WGS") ? atoi(mp_diag_env("MP_OWN_WGS")) : 768;
        const int own_wgs = scheme ? (int)std::min<u64>((u64)h->ow_nsc, 2 * ((h->n + OWN_ROUND - 1) / OWN_ROUND) + 2) : std::min(h->ow_nsc, mn_wgs);
        h->ow_wgs = own_wgs;
        hipLaunchKernelGGL(kern, dim3(own_wgs), dim3(OWN_THREADS), lds, h->stream, h->n, h->n_global, (uint32_t)h->seed, (uint32_t)(h->seed >> 32),
                           h->resample_count, (int)scheme, h->ow_R, t_incl, t_W, t_ratio, h->nt, world, rank, (const unsigned short*)h->guide,
                           (const mp_own_range*)h->ow_range, win_lt, win_row, pa,
                           (!solo_tab && scheme == MP_RESAMPLE_MULTINOMIAL && world > 1) ? (const u64*)h->ow_kthr : (const u64*)nullptr);
        // A world of one has nothing to plan: every draw is this rank's own, super-chunk sc starts at offspring sc * 1024,
        // nothing is sent or received (base[], the plan and c_all[0] = n were set when the scratch was allocated).
        if (world > 1) hipLaunchKernelGGL(k_shard_own_plan, dim3(1), dim3(SHP_THREADS), 0, h->stream, pa);
    }
    h->ow_scheme = scheme;
    rc = check_launch("shard_owned_count kernels");
    if (rc != MP_OK) return rc;
    if (counts_out && world == 1) {
        rc = fetch_scalars(h);   // degenerate weights are reported here, as by the plan's verdict in larger worlds
        if (rc != MP_OK) return rc;
        counts_out[0] = h->n;
    } else if (counts_out) {   // exact sizes: the caller sizes its buffers from the counts (the stream drains: every field of h_pub is out)
        HIPCK(stream_wait(h->stream));
        if (h->h_pub->degenerate)
            return mp_fail(MP_ERR_DEGENERATE, "all log-weights are -inf: normalized weights are NaN (categorical.rs:23 assert in the reference)");
        for (int r = 0; r < world; ++r) counts_out[r] = h->h_pub->counts[r];
    }
    return MP_OK;
}

int32_t mp_pf_shard_owned_count(mp_pf* h, int32_t scheme, const uint64_t* d_tiles_all, int32_t world, int32_t rank, uint64_t capacity,
                                uint64_t* counts_out) {
    return shard_owned_count_impl(h, scheme, d_tiles_all, world, rank, capacity, counts_out, nullptr);
}

int32_t mp_pf_shard_owned_count_expand(mp_pf* h, int32_t scheme, const uint64_t* d_tiles_all, int32_t world, int32_t rank, uint64_t capacity,
                                       double* d_send_out, double* d_rows, uint64_t recv_rows) {
    if (!h || !d_tiles_all || !d_send_out || !d_rows) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    if (capacity == 0) return mp_fail(MP_ERR_INVALID_ARG, "count_expand is the equal-split form: capacity must be > 0 (exact sizes: count, then expand)");
    if (world >= 1 && recv_rows != (uint64_t)world * capacity) return mp_fail(MP_ERR_INVALID_ARG, "fixed capacity: recv_rows must be world * capacity");
    if (recv_rows + h->n >= (1ull << 31)) return mp_fail(MP_ERR_INVALID_ARG, "exchange buffer rows must be < 2^31 (row indices carry a flag bit)");
    int32_t rc = shard_owned_count_impl(h, scheme, d_tiles_all, world, rank, capacity, nullptr, d_send_out);
    if (rc != MP_OK) return rc;
    return mp_pf_shard_owned_expand(h, world, rank, capacity, d_send_out, d_rows, recv_rows);
}

int32_t mp_pf_shard_owned_expand(mp_pf* h, int32_t world, int32_t rank, uint64_t capacity, double* d_send_out, double* d_rows, uint64_t recv_rows) {
    if (!h || !d_send_out || !d_rows) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    if (!h->ow_seg_lt || h->ow_world != world) return mp_fail(MP_ERR_STATE, "shard_owned_expand before shard_owned_count");
    if (rank < 0 || rank >= world) return mp_fail(MP_ERR_INVALID_ARG, "0 <= rank < world");
    if (capacity && recv_rows != (uint64_t)world * capacity) return mp_fail(MP_ERR_INVALID_ARG, "fixed capacity: recv_rows must be world * capacity");
    if (recv_rows + h->n >= (1ull << 31)) return mp_fail(MP_ERR_INVALID_ARG, "exchange buffer rows must be < 2^31 (row indices carry a flag bit)");
    HIPCK(hipSetDevice(h->device));
    h->ow_last_cap = capacity;
    if (h->ow_self) {
        // the kept offspring are drawn by the next k_propagate itself where its kernel can (self_k1_draws), here otherwise; the
        // surplus is looked up now (it travels), the deficit slots are flagged with the rows that will arrive
        if (world == 1 && self_k1_draws(h)) return MP_OK;   // (nothing to launch: no surplus, no deficit, the draws are the next step's)
        // nothing left to launch: the table's leading workgroup has placed — into THIS buffer with THIS capacity (the exact-size repeat
        // after an overflow verdict comes with another buffer and capacity 0: placed again below)
        if (h->ow_placed && self_k1_draws(h) && capacity == h->ow_placed_cap && d_send_out == h->ow_placed_send) return MP_OK;
        LaunchTimer lt(h, MP_K_RESAMPLE_GATHER);
        if (!self_k1_draws(h)) {
            int32_t rcs = launch_self_draws(h, h->ow_scheme, h->resample_count, world, rank);
            if (rcs != MP_OK) return rcs;
        }
        if (world > 1) {
            auto kern = (h->ow_scheme == MP_RESAMPLE_MULTINOMIAL_SPLIT) ? k_shard_self_place<false> : k_shard_self_place<true>;
            hipLaunchKernelGGL(kern, dim3(64), dim3(256), 0, h->stream, self_place_args(h, rank, capacity, d_send_out), h->n_global, (uint32_t)h->seed,
                               (uint32_t)(h->seed >> 32), h->resample_count, h->ow_scheme, world, rank, h->nt, (const mp_own_range*)h->ow_range,
                               (const mp_owned_plan*)h->ow_plan);
        }
        return check_launch("k_shard_self_place");
    }
    if (world > 1) {   // (a world of one: k_shard_own_draw wrote the slot-order arrays itself; nothing is sent or received)
        LaunchTimer lt(h, MP_K_RESAMPLE_GATHER);
        const unsigned grid = (unsigned)std::max(1, std::min(h->ow_scheme ? h->ow_wgs : h->ow_nsc, 2048));
        auto kern = h->ow_scheme ? k_shard_own_place<true> : k_shard_own_place<false>;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, h->stream, h->n, h->slot_offset, h->ops->dim_state, world, rank, h->ow_R, h->ow_nsc,
                           (u64)capacity, (const u64*)h->ow_seg_lt, (const uint32_t*)h->ow_seg_row, (const uint32_t*)h->ow_sccnt,
                           (const uint32_t*)h->ow_base, (const mp_cx*)h->cx, (const double*)h->x[h->cur], (const mp_owned_plan*)h->ow_plan,
                           (const unsigned long long*)h->ow_call, d_send_out, h->dfr_lt, h->dfr_row,
                           h->ow_scheme ? (const mp_own_range*)h->ow_range : (const mp_own_range*)nullptr);
    }
    return check_launch("k_shard_own_place");
}

int32_t mp_pf_shard_owned_commit(mp_pf* h, const double* d_rows, double* log_total_weight, uint64_t* counts_out) {
    if (!h || !d_rows) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    if (!h->ow_seg_lt || !h->h_pub) return mp_fail(MP_ERR_STATE, "shard_owned_commit before shard_owned_expand");
    HIPCK(hipSetDevice(h->device));
    // The one host wait of a resample, and only for the plan's verdict word: the rows may still be written / travelling.  A
    // world of one exchanges nothing, so an asynchronous resample there has no verdict to wait for (degenerate weights
    // surface at the next synchronising call, as for mp_pf_resample without a log_total_weight).
    const bool slow = log_total_weight || counts_out;   // these read more than the verdict word: the stream drains first
    unsigned flags = 0;
    double L_one = 0.;
    if (h->ow_world == 1) {
        if (slow && h->ow_self && !h->ow_solo_folded) {   // nobody has folded this normalisation yet (the next k_propagate would have)
            hipLaunchKernelGGL(k_shard_solo_fold, dim3(1), dim3(1), 0, h->stream, (const mp_tab_head*)h->tab_head, h->scal, h->S, h->n_global);
            h->ow_solo_folded = true;
        }
        if (slow) {   // no plan ran: the scalars themselves
            int32_t rcs = fetch_scalars(h);
            if (rcs != MP_OK) return rcs;
            L_one = h->h_scal->L;
            if (counts_out) counts_out[0] = h->n;
        }
    } else if (slow) {
        HIPCK(stream_wait(h->stream));
        flags = (unsigned)(h->h_pub->verdict & 0xFFu);
        if ((h->h_pub->verdict >> 8) != h->ow_seq) return mp_fail(MP_ERR_HIP, "owner-keeps plan: the stream drained without a verdict");
    } else if (h->ow_world > 1) {
        int32_t rcw = owned_wait_plan(h, &flags);
        if (rcw != MP_OK) return rcw;
    }
    if (counts_out && h->ow_world > 1)
        for (int r = 0; r < h->ow_world; ++r) counts_out[r] = h->h_pub->counts[r];
    if (flags & 2u)   // before anything is committed: with Q == 0 no rank owns a draw and the donor never writes its request slots
        return mp_fail(MP_ERR_DEGENERATE, "all log-weights are -inf: normalized weights are NaN (categorical.rs:23 assert in the reference)");
    if ((flags & 1u) && h->ow_last_cap) {
        // a pair of ranks exchanges more than `capacity` rows (every rank reaches this verdict from the same counts): nothing
        // is committed; the entries stay queued, the caller repeats the expand with exact sizes (capacity 0)
        return mp_fail(MP_ERR_CAPACITY, "owner-keeps exchange: a pair of ranks needs more than `capacity` rows; repeat the expand with exact sizes");
    }
    // from here on the filter is in the state an unsharded resample that only DREW leaves it in: slot i's draw {target, start row}
    // in dfr_lt / dfr_row against the table cx, x[cur] the pre-resample states — the next k_propagate (or k_resolve_slots, if the
    // host reads first) looks the kept offspring up; the slots filled from other ranks read their rows of d_rows (MP_DRAW_RECV)
    h->sh_rows = d_rows;
    h->sh_recv = h->ow_world > 1;
    h->sh_lazy = false;
    h->sh_parents_lazy = false;
    h->deferred = true;
    h->draws_lattice = h->ow_scheme == MP_RESAMPLE_SYSTEMATIC || h->ow_scheme == MP_RESAMPLE_STRATIFIED;
    h->draw_pending = false;
    h->pending_shard = false;
    if (h->ow_self && self_k1_draws(h)) {   // the kept draws are left to the next k_propagate (flush_draws() when something else needs them first)
        h->draw_pending = true;
        h->pending_shard = true;
        h->pending_rc = h->resample_count;
        h->pending_scheme = h->ow_scheme;
        h->ps_world = h->ow_world; h->ps_rank = h->ow_rank;
    }
    h->parents_deferred = false;
    h->rows_fresh = false;
    h->resample_count += 1;
    if (log_total_weight) {
        if (h->ow_world == 1) {
            *log_total_weight = L_one;
        } else {
            if (h->h_pub->degenerate)
                return mp_fail(MP_ERR_DEGENERATE, "all log-weights are -inf: normalized weights are NaN (categorical.rs:23 assert in the reference)");
            *log_total_weight = h->h_pub->L;
        }
    }
    return MP_OK;
}

int32_t mp_pf_shard_query_packed(mp_pf* h, const uint64_t* d_tiles_all, int32_t world, double* log_ml, double* ess) {
    if (!h || !d_tiles_all) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    HIPCK(hipSetDevice(h->device));
    int32_t rc = shard_scratch(h, world, h->sh_cap ? h->sh_cap : 1);
    if (rc != MP_OK) return rc;
    const int nt_all = h->nt * world;
    hipLaunchKernelGGL(k_unpack_tiles, dim3((nt_all + K3_THREADS - 1) / K3_THREADS), dim3(K3_THREADS), 0, h->stream, (const u64*)d_tiles_all, world, h->nt,
                       h->sh_tm_all, h->sh_tW_all, h->sh_tW2_all);
    hipLaunchKernelGGL(k_finalize_tiles, dim3(1), dim3(K3_THREADS), table_lds(nt_all, K3_THREADS), h->stream, h->sh_tm_all, h->sh_tW_all, h->sh_tW2_all,
                       nt_all, h->S, h->n_global, 1, h->scal);
    rc = check_launch("k_finalize_tiles");
    if (rc != MP_OK) return rc;
    rc = fetch_scalars(h);
    if (rc != MP_OK) return rc;
    if (log_ml) *log_ml = h->h_scal->lml_fresh;
    if (ess) *ess = h->h_scal->ess_fresh;
    return MP_OK;
}

int32_t mp_pf_read_trajectories(mp_pf* h, uint64_t first, uint64_t count, double* out, int32_t* t_steps) {
    if (!h || !out || !t_steps) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    if (!(h->flags & MP_PF_RECORD_HISTORY)) return mp_fail(MP_ERR_STATE, "read_trajectory needs a filter created with MP_PF_RECORD_HISTORY");
    if (h->sharded) return mp_fail(MP_ERR_UNSUPPORTED, "read_trajectory: ancestors of a sharded filter live on other ranks");
    if (count == 0 || first >= h->n || count > h->n - first) return mp_fail(MP_ERR_INVALID_ARG, "particle range out of bounds");
    HIPCK(hipSetDevice(h->device));
    const int d = h->ops->dim_state;
    const int T = (int)h->t;
    *t_steps = (int32_t)h->t;
    if (T == 0) return MP_OK;
    // the event log as the kernel reads it
    std::vector<mp_hist_event> ev(h->hist.size());
    for (size_t e = 0; e < h->hist.size(); ++e) { ev[e].buf = h->hist[e].buf; ev[e].kind = h->hist[e].kind; ev[e].pad = 0; }
    if (h->d_hist_events_cap < ev.size()) {
        (void)hipFree(h->d_hist_events);
        h->d_hist_events = nullptr;
        h->d_hist_events_cap = std::max<size_t>(64, 2 * ev.size());
        HIPCK(hipMalloc(&h->d_hist_events, sizeof(mp_hist_event) * h->d_hist_events_cap));
    }
    HIPCK(hipMemcpyAsync(h->d_hist_events, ev.data(), sizeof(mp_hist_event) * ev.size(), hipMemcpyHostToDevice, h->stream));
    double* d_out = nullptr;
    const size_t bytes = sizeof(double) * (size_t)count * (size_t)T * (size_t)d;
    HIPCK(hipMalloc(&d_out, bytes));
    hipLaunchKernelGGL(k_trajectories, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, h->stream, (u64)first, (u64)count, d, T, (int)ev.size(),
                       (const mp_hist_event*)h->d_hist_events, d_out);
    hipError_t e1 = hipGetLastError();
    if (e1 == hipSuccess) e1 = hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, h->stream);
    if (e1 == hipSuccess) e1 = hipStreamSynchronize(h->stream);   // (ev / d_out stay alive until here)
    (void)hipFree(d_out);
    if (e1 != hipSuccess) return mp_fail(MP_ERR_HIP, std::string("mp_pf_read_trajectories: ") + hipGetErrorString(e1));
    return MP_OK;
}

int32_t mp_pf_read_trajectory(mp_pf* h, uint64_t i, double* out, int32_t* t_steps) {
    return mp_pf_read_trajectories(h, i, 1, out, t_steps);
}

int32_t mp_pf_time(mp_pf* h, int64_t* out) {
    if (!h || !out) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    *out = h->t;
    return MP_OK;
}

int32_t mp_pf_run(mp_pf* h, const double* args0, const double* obs, int32_t n_steps, int32_t scheme) {
    if (!h || !obs) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    if (n_steps < 1) return mp_fail(MP_ERR_CONSTRAINTS, "run needs at least one time step");
    int32_t rc = mp_pf_init_step(h, args0, obs, 1);
    if (rc != MP_OK) return rc;
    rc = mp_pf_resample(h, scheme, nullptr);
    if (rc != MP_OK) return rc;
    // (the log-weights a step would store are dead here — `resample` zeroes them, particle_filter.rs:114 — and so are the parents of
    // every resample but the last: k_propagate_mt launches leave both out and reproduce them on demand, mp_pf::lazy_pending; round 4 did
    // that for this loop only, by flags that a failing resample left behind)
    for (int t = 1; t < n_steps && rc == MP_OK; ++t) {
        rc = mp_pf_step(h, obs + (size_t)t * h->ops->dim_obs, 1);
        if (rc == MP_OK) rc = mp_pf_resample(h, scheme, nullptr);
    }
    return rc;
}

int32_t mp_pf_synchronize(mp_pf* h) {
    if (!h) return mp_fail(MP_ERR_INVALID_ARG, "null handle");
    HIPCK(hipSetDevice(h->device));
    // polling the stream and one host-mapped word: copying the scalars back for their `degenerate` field cost 15 us of every call
    HIPCK(stream_wait(h->stream));
    if (*(volatile int*)h->h_flag)
        return mp_fail(MP_ERR_DEGENERATE, "all log-weights are -inf: normalized weights are NaN (categorical.rs:23 assert in the reference)");
    return MP_OK;
}

int32_t mp_pf_set_timing(mp_pf* h, int32_t enabled) {
    if (!h) return mp_fail(MP_ERR_INVALID_ARG, "null handle");
    int32_t rc = drain_timing(h);
    if (rc != MP_OK) return rc;
    h->timing = enabled != 0;
    for (int f = 0; f < MP_K_COUNT; ++f) {
        h->fam_ms[f] = 0.;
        h->fam_launches[f] = 0;
    }
    return MP_OK;
}

int32_t mp_pf_get_timing(mp_pf* h, int32_t which, double* total_ms, uint64_t* launches) {
    if (!h || which < 0 || which >= MP_K_COUNT) return mp_fail(MP_ERR_INVALID_ARG, "bad argument");
    int32_t rc = drain_timing(h);
    if (rc != MP_OK) return rc;
    if (total_ms) *total_ms = h->fam_ms[which];
    if (launches) *launches = h->fam_launches[which];
    return MP_OK;
}

int32_t mp_pf_last_propagate_form(mp_pf* h, int32_t* out) {
    if (!h || !out) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    *out = h->last_k1_form;
    return MP_OK;
}

int32_t mp_pf_region_begin(mp_pf* h) {
    if (!h) return mp_fail(MP_ERR_INVALID_ARG, "null handle");
    HIPCK(hipSetDevice(h->device));
    if (!h->region_ev[0]) {
        HIPCK(hipEventCreate(&h->region_ev[0]));
        HIPCK(hipEventCreate(&h->region_ev[1]));
    }
    h->region_launches = 0;
    HIPCK(hipEventRecord(h->region_ev[0], h->stream));
    h->region_open = true;
    return MP_OK;
}

int32_t mp_pf_region_end(mp_pf* h, double* elapsed_ms, uint64_t* propagate_launches) {
    if (!h) return mp_fail(MP_ERR_INVALID_ARG, "null handle");
    if (!h->region_open) return mp_fail(MP_ERR_STATE, "region_end without region_begin");
    HIPCK(hipSetDevice(h->device));
    HIPCK(hipEventRecord(h->region_ev[1], h->stream));
    HIPCK(stream_wait(h->stream));
    float ms = 0.f;
    HIPCK(hipEventElapsedTime(&ms, h->region_ev[0], h->region_ev[1]));
    h->region_open = false;
    if (elapsed_ms) *elapsed_ms = (double)ms;
    if (propagate_launches) *propagate_launches = h->region_launches;
    return MP_OK;
}

static void shard_native_free(mp_shard_native_state* s);   // mp_shard_native.h
int32_t mp_pf_destroy(mp_pf* h) {
    if (!h) return MP_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    shard_native_free(h->native);
    for (auto& tl : h->timed) {
        (void)hipEventDestroy(tl.start);
        (void)hipEventDestroy(tl.stop);
    }
    for (auto e : h->event_pool) (void)hipEventDestroy(e);
    for (auto e : h->region_ev) if (e) (void)hipEventDestroy(e);
    for (void* slab : h->hist_slabs) (void)hipFree(slab);
    (void)hipFree(h->d_hist_events);
    (void)hipFree(h->x[0]); (void)hipFree(h->x[1]); (void)hipFree(h->logw); (void)hipFree(h->cx); (void)hipFree(h->cx_alt); (void)hipFree(h->k1_tail_alt); (void)hipFree(h->guide);
    (void)hipFree(h->guide_alt); (void)hipFree(h->tab_W);
    (void)hipFree(h->parent); (void)hipFree(h->tiles_own); (void)hipFree(h->tiles_alt); (void)hipFree(h->scal);
    (void)hipFree(h->aos);
    (void)hipFree(h->k1_tail);
    (void)hipFree(h->tab_ticket); (void)hipFree(h->tab_incl); (void)hipFree(h->tab_ratio); (void)hipFree(h->tab_head);
    (void)hipFree(h->dfr_lt); (void)hipFree(h->dfr_row);
    (void)hipFree(h->sh_dest); (void)hipFree(h->sh_lt); (void)hipFree(h->sh_tile); (void)hipFree(h->sh_req_slot); (void)hipFree(h->sh_blockcount);
    (void)hipFree(h->sh_blockoff); (void)hipFree(h->sh_counts); (void)hipFree(h->sh_tm_all); (void)hipFree(h->sh_tW_all); (void)hipFree(h->sh_tW2_all); (void)hipFree(h->sh_incl_all); (void)hipFree(h->sh_ratio_all);
    (void)hipFree(h->sh_overflow); (void)hipFree(h->sh_done); (void)hipFree(h->scal_undo);
    owned_free(h);
    if (h->h_counts) (void)hipHostFree(h->h_counts);
    if (h->h_pub) (void)hipHostFree(h->h_pub);
    if (h->ev_resolved) (void)hipEventDestroy(h->ev_resolved);
    (void)hipHostFree(h->h_scal);
    if (h->h_flag) (void)hipHostFree(h->h_flag);
    if (h->h_mirror) (void)hipHostFree(h->h_mirror);
    if (h->own_stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return MP_OK;
}

// GenFn::simulate over an Unfold model — DynUnfold::simulate (modppl/src/modeling/dynunfold.rs:22-39): n independent traces of
// n_steps kernel calls, every site sampled (the observation sites too).
int32_t mp_unfold_simulate(const mp_model_desc* model, const double* args0, int32_t n_steps, uint64_t n, uint64_t seed, int32_t device,
                           double* states_out, double* obs_out) {
    if (!model || !states_out || !obs_out) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    if (n_steps < 1) return mp_fail(MP_ERR_INVALID_ARG, "assert final_t >= 1 (dynunfold.rs:24)");
    if (n == 0 || n > 0xFFFFFFFFull) return mp_fail(MP_ERR_INVALID_ARG, "n must be in [1, 2^32)");
    if (model->kind == MP_MODEL_HMM) return mp_fail(MP_ERR_UNSUPPORTED, "the reference's HMM has no simulate (tests/hmm/model.rs: unimplemented)");
    std::unique_ptr<ModelOps> ops;
    int32_t rc = make_model(model, ops);
    if (rc != MP_OK) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        return mp_fail(MP_ERR_HIP, "no HIP device visible: the gfx950 path has no CPU fallback");
    }
    HIPCK(hipSetDevice(device));
    const size_t nx = (size_t)n * n_steps * ops->dim_state, ny = (size_t)n * n_steps * ops->dim_obs;
    double *dx = nullptr, *dy = nullptr;
    HIPCK(hipMalloc(&dx, sizeof(double) * nx));
    if (hipMalloc(&dy, sizeof(double) * ny) != hipSuccess) { (void)hipFree(dx); return mp_fail(MP_ERR_HIP, "hipMalloc failed"); }
    mp_state0 s0{};
    for (int j = 0; j < MP_MAX_STATE; ++j) s0.v[j] = (args0 && j < ops->dim_state) ? args0[j] : 0.;
    ops->simulate(n, (uint32_t)seed, (uint32_t)(seed >> 32), n_steps, s0, dx, dy, nullptr);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(states_out, dx, sizeof(double) * nx, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(obs_out, dy, sizeof(double) * ny, hipMemcpyDeviceToHost);
    (void)hipFree(dx); (void)hipFree(dy);
    if (e != hipSuccess) return mp_fail(MP_ERR_HIP, std::string("mp_unfold_simulate: ") + hipGetErrorString(e));
    return MP_OK;
}

static int32_t importance_run(const mp_model_desc* model, const double* args0, const double* obs, int32_t n_steps, uint64_t num_samples,
                              uint64_t num_ret_samples, uint64_t seed, int32_t device, double* log_ml_estimate, double* log_normalized_weights,
                              uint64_t* resampled_indices, double* final_states, double* trajectories);
int32_t mp_importance_resampling(const mp_model_desc* model, const double* args0, const double* obs, int32_t n_steps, uint64_t num_samples,
                                 uint64_t num_ret_samples, uint64_t seed, int32_t device, double* log_ml_estimate,
                                 double* log_normalized_weights, uint64_t* resampled_indices, double* final_states) {
    return importance_run(model, args0, obs, n_steps, num_samples, num_ret_samples, seed, device, log_ml_estimate, log_normalized_weights,
                          resampled_indices, final_states, nullptr);
}
int32_t mp_importance_sampling(const mp_model_desc* model, const double* args0, const double* obs, int32_t n_steps, uint64_t num_samples, uint64_t seed,
                               int32_t device, double* log_ml_estimate, double* log_normalized_weights, double* trajectories_out) {
    return importance_run(model, args0, obs, n_steps, num_samples, 0, seed, device, log_ml_estimate, log_normalized_weights, nullptr, nullptr, trajectories_out);
}
static int32_t importance_run(const mp_model_desc* model, const double* args0, const double* obs, int32_t n_steps, uint64_t num_samples,
                              uint64_t num_ret_samples, uint64_t seed, int32_t device, double* log_ml_estimate, double* log_normalized_weights,
                              uint64_t* resampled_indices, double* final_states, double* trajectories) {
    if (!model || !obs) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    if (num_ret_samples > 0xFFFFFFFFull) return mp_fail(MP_ERR_INVALID_ARG, "num_ret_samples must fit u32");
    mp_pf* h = nullptr;
    // (the traces themselves — importance.rs:26-27 returns all N of them — are every sample's states at every step: the ancestry
    // record, which without a resample is each slot's own history)
    int32_t rc = mp_pf_create(model, num_samples, seed, nullptr, trajectories ? MP_PF_RECORD_HISTORY : 0, device, nullptr, &h);
    if (rc != MP_OK) return rc;
    struct Guard { mp_pf* h; ~Guard() { mp_pf_destroy(h); } } guard{h};
    // importance_sampling: N x generate(model_args, constraints) over all n_steps constraints (importance.rs:18-20)
    rc = mp_pf_init_step(h, args0, obs, n_steps);
    if (rc != MP_OK) return rc;
    rc = ensure_rows(h);
    if (rc != MP_OK) return rc;
    hipLaunchKernelGGL(k_finalize_tiles, dim3(1), dim3(K3_THREADS), table_lds(h->nt, K3_THREADS), h->stream, h->tile_m, h->tile_W, h->tile_W2, h->nt,
                       h->S, h->n, 2, h->scal);
    rc = check_launch("k_finalize_tiles");
    if (rc != MP_OK) return rc;
    rc = fetch_scalars(h);
    if (rc != MP_OK) return rc;
    if (log_ml_estimate) *log_ml_estimate = h->h_scal->lml_fresh;
    if (log_normalized_weights) {
        double* tmp = h->aos;  // n doubles of scratch (dim_state >= 1)
        hipLaunchKernelGGL(k_sub_scalar, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, h->stream, h->logw, &h->scal->L, h->n, tmp);
        rc = check_launch("k_sub_scalar");
        if (rc != MP_OK) return rc;
        HIPCK(hipMemcpyAsync(log_normalized_weights, tmp, sizeof(double) * h->n, hipMemcpyDeviceToHost, h->stream));
        HIPCK(hipStreamSynchronize(h->stream));
    }
    if (final_states) {
        rc = mp_pf_read_state(h, final_states);
        if (rc != MP_OK) return rc;
    }
    if (trajectories) {
        int32_t t_steps = 0;
        rc = mp_pf_read_trajectories(h, 0, num_samples, trajectories, &t_steps);
        if (rc != MP_OK) return rc;
    }
    if (resampled_indices && num_ret_samples > 0) {
        // importance_resampling: M categorical draws over exp(lnw) (importance.rs:44-47), slot j of DOM_IS
        uint32_t* d_idx = nullptr;
        HIPCK(hipMalloc(&d_idx, sizeof(uint32_t) * num_ret_samples));
        const int grid = (int)std::min<u64>((num_ret_samples + KG_THREADS * KG_ITEMS - 1) / (KG_THREADS * KG_ITEMS), (u64)K3_MAX_BLOCKS);
        hipLaunchKernelGGL(k_resample_gather<0>, dim3(grid), dim3(KG_THREADS), table_lds(h->nt, KG_THREADS), h->stream, h->n, (u64)num_ret_samples,
                           h->n_global, (u64)0, (uint32_t)MP_DOM_IS, (uint32_t)h->seed, (uint32_t)(h->seed >> 32), 0u, h->S, h->ops->dim_state, h->cx,
                           h->guide, h->tile_m, h->tile_W, h->tile_W2, h->nt, (const double*)nullptr, (double*)nullptr, d_idx, (double*)nullptr,
                           (mp_dev_scalars*)nullptr);
        rc = check_launch("k_resample_gather(IS)");
        std::vector<uint32_t> idx(num_ret_samples);
        hipError_t e1 = hipMemcpyAsync(idx.data(), d_idx, sizeof(uint32_t) * num_ret_samples, hipMemcpyDeviceToHost, h->stream);
        hipError_t e2 = hipStreamSynchronize(h->stream);
        (void)hipFree(d_idx);
        if (rc != MP_OK) return rc;
        if (e1 != hipSuccess || e2 != hipSuccess) return mp_fail(MP_ERR_HIP, "importance_resampling: index copy failed");
        for (uint64_t j = 0; j < num_ret_samples; ++j) resampled_indices[j] = idx[j];
    }
    return MP_OK;
}

#include "mp_shard_native.h"

#ifdef MP_STAMPS
// diagnostics only (libmodppl_hip_stamps.so, tools/stamp_probe.py): enable / read the per-workgroup stamps
int32_t mp_debug_stamps(unsigned long long* host_out /* [KERNELS][MAX_WG][SLOTS] or NULL to (re)arm */) {
    static unsigned long long* d_buf = nullptr;
    const size_t bytes = sizeof(unsigned long long) * MP_STAMP_KERNELS * MP_STAMP_MAX_WG * MP_STAMP_SLOTS;
    if (!d_buf) {
        HIPCK(hipMalloc(&d_buf, bytes));
        HIPCK(hipMemcpyToSymbol(HIP_SYMBOL(g_mp_stamp_buf), &d_buf, sizeof(d_buf)));
    }
    HIPCK(hipDeviceSynchronize());
    if (host_out) HIPCK(hipMemcpy(host_out, d_buf, bytes, hipMemcpyDeviceToHost));
    else HIPCK(hipMemset(d_buf, 0, bytes));
    return MP_OK;
}
#endif

}  // extern "C"

// The tail of importance_sampling / importance_resampling (importance.rs:21-27, 44-47) over ANY device array of log-weights — for
// callers inside the library whose samples are not an Unfold model's (mp_mh.hip: importance sampling over a registered generative
// function).  The same canonical normalisation and the same M categorical draws (Philox domain IS) as importance_run above: level 0 by
// k_normalize_tiles, level 1 by k_finalize_tiles (mode 2: L and log_ml = L - ln N), lnw = lw - L, draws by k_resample_gather<0>.
int32_t mp_is_finish_device(const double* d_logw, uint64_t n, uint64_t num_ret, uint64_t seed, int32_t device, void* stream, double* log_ml_estimate,
                            double* log_normalized_weights, uint64_t* resampled_indices) {
    if (!d_logw || n == 0 || n > 0xFFFFFFFFull) return mp_fail(MP_ERR_INVALID_ARG, "importance sampling: num_samples must be in [1, 2^32)");
    if (num_ret > 0xFFFFFFFFull) return mp_fail(MP_ERR_INVALID_ARG, "num_ret_samples must fit u32");
    const int nt = (int)((n + TILE - 1) / TILE);
    if (nt > MAX_TILES) return mp_fail(MP_ERR_UNSUPPORTED, "at most 2^24 samples in this build (tile table in LDS)");
    HIPCK(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    const int S = 62 - ceil_log2_u64(n);
    struct Bufs {
        mp_cx* cx = nullptr; unsigned short* guide = nullptr; u64* tiles = nullptr; mp_dev_scalars* scal = nullptr; double* tmp = nullptr; uint32_t* idx = nullptr;
        ~Bufs() { (void)hipFree(cx); (void)hipFree(guide); (void)hipFree(tiles); (void)hipFree(scal); (void)hipFree(tmp); (void)hipFree(idx); }
    } b;
    HIPCK(hipMalloc(&b.cx, sizeof(mp_cx) * (size_t)nt * TILE));
    HIPCK(hipMalloc(&b.guide, sizeof(unsigned short) * (size_t)nt * GUIDE_N));
    HIPCK(hipMalloc(&b.tiles, sizeof(u64) * 3 * (size_t)nt));
    HIPCK(hipMalloc(&b.scal, sizeof(mp_dev_scalars)));
    HIPCK(hipMemsetAsync(b.scal, 0, sizeof(mp_dev_scalars), st));
    double* tile_m = reinterpret_cast<double*>(b.tiles);
    u64 *tile_W = b.tiles + nt, *tile_W2 = b.tiles + 2 * (size_t)nt;
    const size_t lds = table_lds(nt, K3_THREADS);
    if (lds > 48 * 1024) {
        HIPCK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_finalize_tiles), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_resample_gather<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    mp_tab tab{};
    tab.S = S;
    hipLaunchKernelGGL(k_normalize_tiles, dim3(nt), dim3(TILE_THREADS), 0, st, d_logw, d_logw /* x0: the rows' second halves are not read here */, 1, (u64)n, b.cx,
                       b.guide, tile_m, tile_W, tile_W2, tab);
    hipLaunchKernelGGL(k_finalize_tiles, dim3(1), dim3(K3_THREADS), lds, st, tile_m, tile_W, tile_W2, nt, S, (u64)n, 2, b.scal);
    int32_t rc = check_launch("importance sampling: normalisation");
    if (rc != MP_OK) return rc;
    mp_dev_scalars hs;
    HIPCK(hipMemcpyAsync(&hs, b.scal, sizeof(hs), hipMemcpyDeviceToHost, st));
    HIPCK(hipStreamSynchronize(st));
    if (hs.degenerate) return mp_fail(MP_ERR_DEGENERATE, "all log-weights are -inf: normalized weights are NaN (categorical.rs:23 assert in the reference)");
    if (log_ml_estimate) *log_ml_estimate = hs.lml_fresh;
    if (log_normalized_weights) {
        HIPCK(hipMalloc(&b.tmp, sizeof(double) * n));
        hipLaunchKernelGGL(k_sub_scalar, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_logw, &b.scal->L, (u64)n, b.tmp);
        rc = check_launch("k_sub_scalar");
        if (rc != MP_OK) return rc;
        HIPCK(hipMemcpyAsync(log_normalized_weights, b.tmp, sizeof(double) * n, hipMemcpyDeviceToHost, st));
        HIPCK(hipStreamSynchronize(st));
    }
    if (resampled_indices && num_ret > 0) {
        HIPCK(hipMalloc(&b.idx, sizeof(uint32_t) * num_ret));
        const int grid = (int)std::min<u64>((num_ret + KG_THREADS * KG_ITEMS - 1) / (KG_THREADS * KG_ITEMS), (u64)K3_MAX_BLOCKS);
        hipLaunchKernelGGL(k_resample_gather<0>, dim3(grid), dim3(KG_THREADS), lds, st, (u64)n, (u64)num_ret, (u64)n, (u64)0, (uint32_t)MP_DOM_IS, (uint32_t)seed,
                           (uint32_t)(seed >> 32), 0u, S, 1, b.cx, b.guide, tile_m, tile_W, tile_W2, nt, (const double*)nullptr, (double*)nullptr, b.idx,
                           (double*)nullptr, (mp_dev_scalars*)nullptr);
        rc = check_launch("k_resample_gather(IS)");
        if (rc != MP_OK) return rc;
        std::vector<uint32_t> idx(num_ret);
        HIPCK(hipMemcpyAsync(idx.data(), b.idx, sizeof(uint32_t) * num_ret, hipMemcpyDeviceToHost, st));
        HIPCK(hipStreamSynchronize(st));
        for (uint64_t j = 0; j < num_ret; ++j) resampled_indices[j] = idx[j];
    }
    return MP_OK;
}
