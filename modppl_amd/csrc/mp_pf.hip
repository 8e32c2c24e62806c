// mp_pf.hip — gfx950 kernels and C-ABI implementation of the particle-filter hot path
// (include/modppl_hip.h).  Written for MI355X only: 64-lane wavefronts, LDS-staged reductions,
// SoA particle state in HBM.
//
// Data layout in HBM (per handle, n = local particles, d = dim_state, tiles of 2048 rows):
//   x[2][d][n]  f64   particle states (slot order)
//   logw[n]     f64   log-weights                              (particle_filter.rs:15)
//   cx[n]       {u64,f64}  resampling table rows: tile-local inclusive prefix of the fixed-point weights + x[0]
//   guide[nt][2048] u16  per-tile bucketed inverse CDF (first row of each bucket)
//   tile_m/W/W2[nt]      per-tile max log-weight and fixed-point totals (level 0 of the normalisation)
//   seg_lt/seg_row/perm/res_x/res_parent  the XCD-binned resampler's segments and segment-ordered results
//   parent[n]   u32   parents of the last resample             (particle_filter.rs:20)
//   scal        mp_dev_scalars   log_ml_estimate, last log total weight, ESS ... (device-resident so
//                                that a whole filter run needs no host round trip)
//
// Kernels of one SMC time step (no inter-workgroup communication inside a launch, so nothing depends on
// dispatch order or XCD placement; all cross-workgroup sums are integer):
//   K1  k_propagate       ParticleSystem::init_step/step: one workgroup per 2048-row tile runs the model kernel in
//                         Generate mode (logw (+)= weight) and, in the same pass, level 0 of normalize_weights
//                         (:27-35): tile max, exp, 51-bit fixed point, tile-local scan, rows, guide
//   K3a k_bin_draws       multinomial_resampling (:37-41), part 1: tile table (level 1), Philox draw, target,
//                         tile + guide lookup, stable split of the draws into 8 CDF-eighth bins
//   K3b k_resolve_bins    part 2 + the clone loop of resample (:109-114): row lookups per bin on one XCD
//   (K3 k_resample_gather single-kernel form: importance_resampling's M draws and systematic resampling)
// The normalisation spec (hierarchical fixed point) is stated in DESIGN.md §4 and restated on the CPU in
// oracle/src/inference.hpp.
#include <hip/hip_runtime.h>
#include <cstring>

#include <algorithm>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "../../include/modppl_hip.h"
#include "mp_linalg.h"
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

// ---------------------------------------------------------------------------------------------
// device scalars
// ---------------------------------------------------------------------------------------------
struct mp_dev_scalars {
    double m;          // global max log-weight of the last level-1 combine
    double L;          // log total weight of the last resample (resample()'s return value)
    double log_ml;     // log_ml_estimate (particle_filter.rs:24)
    double ess_stale;  // ESS of the weights normalised by the last resample (:98-100 semantics)
    double ess_fresh;  // outputs of the query path (k_finalize_tiles mode 1)
    double lml_fresh;
    u64 Q, Q2;
    int degenerate;    // sticky: all log-weights were -inf (or +inf) at a normalisation
    int pad;
};

constexpr int TILE_THREADS = 512;
constexpr int TILE_ITEMS = 4;
constexpr int TILE = TILE_THREADS * TILE_ITEMS;  // 2048 rows per tile: a constant of the normalisation spec
constexpr int GUIDE_BITS = 11;                   // one guide bucket per table row (GUIDE_N == TILE): 2 B per particle
constexpr int GUIDE_N = 1 << GUIDE_BITS;
static_assert(GUIDE_N == TILE, "normalize_tile zeroes/stores the guide with one 8-byte word per thread");
constexpr int GUIDE_DIRECT = 8;                  // bucket runs longer than this are filled by the whole wave
constexpr int FIX_BITS = 51;                     // level-0 fixed point: q = rint(exp(lw - m_tile) * 2^51), 2048 * 2^51 < 2^63
constexpr int BIN_CHUNK = 1024;                  // output slots per chunk of the binned resampler
constexpr int BIN_THREADS = 256;
constexpr int BIN_ITEMS = BIN_CHUNK / BIN_THREADS;
constexpr int BIN_GROUP = 8;                     // chunks per k_resolve_bins workgroup
// Position of entry e of segment (bin, chunk) in the sparse segment arrays [bin][chunk][1024].  Only ~128 entries of
// each 1024-entry window are used; the start is rotated by (chunk % 8) * 128 so the used parts spread over the address space.
#define MP_SEG_POS(bin, c, e, nchunks) ((((u64)(bin) * (u64)(nchunks) + (u64)(c)) * BIN_CHUNK) + (u64)((((uint32_t)(e)) + (((uint32_t)(c)) & 7u) * 128u) & 1023u))
constexpr int K3_THREADS = 256;
constexpr int K3_ITEMS = 4;
constexpr int K3_MAX_BLOCKS = 4096;
constexpr int MAX_TILES = 8192;                  // LDS tile table: 16 B per tile

// particles per predraw round in k_propagate: 4 pre-drawn (u, r) pairs per lane whatever the model's number of normal sites
template <class Model>
constexpr int k1_items() { return Model::MAX_NORMALS >= 4 ? 1 : 4 / Model::MAX_NORMALS; }

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ u64 wave_sum_u64(u64 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// inclusive scan across the 64 lanes of a wave
__device__ __forceinline__ u64 wave_incl_scan_u64(u64 v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u64 t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}
__device__ __forceinline__ u64 mp_quantize(double e, double scale) {
    const double r = rint(e * scale);
    return (r >= 0.) ? (u64)r : 0ull;  // NaN -> 0
}

// One row of the resampling table: tile-local inclusive fixed-point CDF value and the first state
// component of the same particle, so that the probe that finds a parent also fetches its state.
struct __attribute__((aligned(16))) mp_cx {
    u64 cum;
    double x0;
};

// Guide table (bucketed inverse CDF, per tile): bucket g covers tile-local targets t with
// (t >> shift) == g, shift = max(0, bitlen(W) - 11) for the tile total W; guide[g] = first local
// index j with cum_j >= max(1, g << shift).  A draw then starts its scan at guide[t >> shift]
// and walks forward (expected < 2 rows).  Integer shifts only: no rounding anywhere.
__device__ __forceinline__ int mp_guide_shift(u64 W) {
    const int bits = 64 - __clzll((long long)W);  // W == 0 -> clz = 64 -> bits = 0
    return bits > GUIDE_BITS ? bits - GUIDE_BITS : 0;
}

// ---------------------------------------------------------------------------------------------
// level 0 of the normalisation for ONE tile, by the workgroup (TILE_THREADS threads) that owns it.
// Thread t holds rows tile*2048 + 4t .. 4t+3: log-weights lw[] and first state components xv[].
//   m_b = max lw;  a = mp_exp(lw - m_b);  q = rint(a * 2^51);  rows = tile-local inclusive prefix;  W_b, W2_b;  guide.
// ---------------------------------------------------------------------------------------------
template <int THREADS>
__device__ __forceinline__ void normalize_tile(const double (&lw)[TILE / THREADS], const double (&xv)[TILE / THREADS], u64 n, u64 tile,
                                               mp_cx* __restrict__ cx, unsigned short* __restrict__ guide,
                                               double* __restrict__ tile_m, u64* __restrict__ tile_W, u64* __restrict__ tile_W2) {
    constexpr int ITEMS_ = TILE / THREADS;   // 512 x 4 or 1024 x 2: a tile is always 2048 consecutive slots
    __shared__ double s_red[THREADS / 64];
    __shared__ u64 s_wsum[THREADS / 64];
    __shared__ u64 s_wsum2[THREADS / 64];
    __shared__ __attribute__((aligned(16))) unsigned short s_guide[GUIDE_N];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 base = tile * TILE + (u64)tid * ITEMS_;

    double m = MP_NEG_INF;
#pragma unroll
    for (int j = 0; j < ITEMS_; ++j)
        if (base + j < n) m = fmax(m, lw[j]);
    m = wave_max(m);
    if (lane == 0) s_red[wave] = m;
    if constexpr (THREADS == 512) reinterpret_cast<u64*>(s_guide)[tid] = 0ull;  // THREADS x (8 | 4) B = the whole guide
    else reinterpret_cast<uint32_t*>(s_guide)[tid] = 0u;
    __syncthreads();
    m = s_red[0];
#pragma unroll
    for (int w = 1; w < THREADS / 64; ++w) m = fmax(m, s_red[w]);
    const bool ok = (m > MP_NEG_INF) && (m < MP_INF);

    const double scale = mp_u2f((u64)(1023 + FIX_BITS) << 52);  // 2^51
    u64 c[ITEMS_];
    u64 run = 0, run2 = 0;
#pragma unroll
    for (int j = 0; j < ITEMS_; ++j) {
        const bool live = ok && (base + j < n);
        const double a = live ? mp_exp(lw[j] - m) : 0.;
        run += mp_quantize(a, scale);
        run2 += mp_quantize(a * a, scale);
        c[j] = run;
    }
    const u64 incl = wave_incl_scan_u64(run, lane);
    const u64 wtot2 = wave_sum_u64(run2);
    if (lane == 63) s_wsum[wave] = incl;
    if (lane == 0) s_wsum2[wave] = wtot2;
    __syncthreads();
    u64 woff = 0, W = 0;
#pragma unroll
    for (int k = 0; k < THREADS / 64; ++k) {
        const u64 v = s_wsum[k];
        if (k < wave) woff += v;
        W += v;
    }
    const u64 off = woff + (incl - run);
#pragma unroll
    for (int j = 0; j < ITEMS_; ++j) {
        if (base + j < n) {
            mp_cx row;
            row.cum = off + c[j];
            row.x0 = xv[j];
            cx[base + j] = row;
        }
    }
    if (tid == 0) {
        u64 t2 = 0;
#pragma unroll
        for (int k = 0; k < THREADS / 64; ++k) t2 += s_wsum2[k];
        tile_m[tile] = m;
        tile_W[tile] = W;
        tile_W2[tile] = t2;
    }

    // ---- guide table of this tile ------------------------------------------------------------
    const int shift = mp_guide_shift(W);
    u64 prev = off;
    int long_lo = 0, long_hi = -1, long_j = 0;  // at most one long run is kept per thread; extra ones fall back to direct writes
#pragma unroll
    for (int j = 0; j < ITEMS_; ++j) {
        const u64 cur = off + c[j];
        if (cur > prev) {
            const int g_lo = prev ? (int)(prev >> shift) + 1 : 0;
            const int g_hi = (int)(cur >> shift);
            const unsigned short idx = (unsigned short)(tid * ITEMS_ + j);
            if (g_hi - g_lo < GUIDE_DIRECT || long_hi >= long_lo) {
                for (int g = g_lo; g <= g_hi; ++g) s_guide[g] = idx;
            } else {
                long_lo = g_lo; long_hi = g_hi; long_j = idx;
            }
        }
        prev = cur;
    }
    // wave-cooperative fill of long runs (a particle holding a large share of the tile's weight)
    u64 pending = __ballot(long_hi >= long_lo);
    while (pending) {
        const int leader = __ffsll((long long)pending) - 1;
        const int lo = __shfl(long_lo, leader, 64), hi = __shfl(long_hi, leader, 64), jj = __shfl(long_j, leader, 64);
        for (int g = lo + lane; g <= hi; g += 64) s_guide[g] = (unsigned short)jj;
        pending &= pending - 1;
    }
    __syncthreads();
    if constexpr (THREADS == 512) reinterpret_cast<u64*>(guide + tile * GUIDE_N)[tid] = reinterpret_cast<const u64*>(s_guide)[tid];
    else reinterpret_cast<uint32_t*>(guide + tile * GUIDE_N)[tid] = reinterpret_cast<const uint32_t*>(s_guide)[tid];
}

// standalone form: used when the log-weights changed without a propagate (after a resample, before a query or a
// second resample)
__global__ __launch_bounds__(TILE_THREADS) void k_normalize_tiles(const double* __restrict__ logw, const double* __restrict__ x0, int D, u64 n,
                                                                  mp_cx* __restrict__ cx, unsigned short* __restrict__ guide,
                                                                  double* __restrict__ tile_m, u64* __restrict__ tile_W, u64* __restrict__ tile_W2) {
    const u64 base = (u64)blockIdx.x * TILE + (u64)threadIdx.x * TILE_ITEMS;
    double lw[TILE_ITEMS], xv[TILE_ITEMS];
    if (base + TILE_ITEMS <= n) {
        const double2 a = *reinterpret_cast<const double2*>(logw + base);
        const double2 b = *reinterpret_cast<const double2*>(logw + base + 2);
        lw[0] = a.x; lw[1] = a.y; lw[2] = b.x; lw[3] = b.y;
        if (D == 1) {
            const double2 xa = *reinterpret_cast<const double2*>(x0 + base);
            const double2 xb = *reinterpret_cast<const double2*>(x0 + base + 2);
            xv[0] = xa.x; xv[1] = xa.y; xv[2] = xb.x; xv[3] = xb.y;
        } else {
#pragma unroll
            for (int j = 0; j < TILE_ITEMS; ++j) xv[j] = x0[(base + j) * (u64)D];   // states are particle-major: x[i][d]
        }
    } else {
#pragma unroll
        for (int j = 0; j < TILE_ITEMS; ++j) {
            lw[j] = (base + j < n) ? logw[base + j] : MP_NEG_INF;
            xv[j] = (base + j < n) ? x0[(base + j) * (u64)D] : 0.;
        }
    }
    normalize_tile<TILE_THREADS>(lw, xv, n, blockIdx.x, cx, guide, tile_m, tile_W, tile_W2);
}

// ---------------------------------------------------------------------------------------------
// K1: propagate + weight + level 0 of the normalisation, one workgroup per tile
// ---------------------------------------------------------------------------------------------
// A lane owns the 4 consecutive particles 4*tid .. 4*tid+3 of its tile, processed in rounds of k1_items<Model>()
// particles.  Per round, phase 1 runs every polar rejection loop of the lane as ONE lane-local work queue over its
// (particle, normal site) items: a wave iterates max-over-lanes of the SUM of attempts instead of the sum over
// items of the max, i.e. ~1.9 Philox blocks per item at 4 items instead of ~3.6 (acceptance pi/4, 64 lanes).
// Phase 2 runs the model kernel per particle on the accepted pairs.
template <class Model, int THREADS>
__global__ __launch_bounds__(THREADS, (THREADS == 1024 ? 8 : 1)) void k_propagate(Model model, u64 n, u64 slot_offset, uint32_t k0, uint32_t k1,
                                                            long long t, const double* x_in, double* x_out, double* logw,
                                                            mp_obs obs, mp_state0 s0, int overwrite,
                                                            const unsigned short* __restrict__ perm, const double* __restrict__ res_x,
                                                            u64 res_stride, int nchunks, mp_cx* __restrict__ cx,
                                                            unsigned short* __restrict__ guide, double* __restrict__ tile_m,
                                                            u64* __restrict__ tile_W, u64* __restrict__ tile_W2,
                                                            const uint32_t* __restrict__ inv, const uint32_t* __restrict__ res_parent) {
    constexpr int D = Model::DIM_STATE;
    constexpr int NS = Model::MAX_NORMALS;
    constexpr int LANE_ITEMS = TILE / THREADS;
    constexpr int ITEMS = k1_items<Model>() < LANE_ITEMS ? k1_items<Model>() : LANE_ITEMS;
    constexpr int ROUNDS = LANE_ITEMS / ITEMS;
    constexpr int M = ITEMS * NS;
    const int ns = model.n_normals(t);  // wave-uniform
    const u64 base = (u64)blockIdx.x * TILE + (u64)threadIdx.x * LANE_ITEMS;
    double lw[LANE_ITEMS], xv[LANE_ITEMS];
#pragma unroll
    for (int j = 0; j < LANE_ITEMS; ++j) { lw[j] = MP_NEG_INF; xv[j] = 0.; }
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
        const u64 i0 = base + (u64)rd * ITEMS;
        // ---- phase 1: accepted (u, r) pairs for every (particle, normal site) of this round ----
        double pu[M], pr[M];
#pragma unroll
        for (int q = 0; q < M; ++q) { pu[q] = 0.; pr[q] = 1.; }
        {
            int p = 0, sidx = 0;  // current item: particle p of the round, normal site index sidx
            uint32_t att = 0;
            while (p < ITEMS && ns > 0) {
                const u64 i = i0 + (u64)p;
                if (i >= n) break;
                const mp_u64x2 b = mp_philox4x32_10((uint32_t)(slot_offset + i), (uint32_t)t,
                                                    ((uint32_t)MP_DOM_MODEL << 16) | model.normal_site(sidx), att, k0, k1);
                const double u = mp_u01(b.a) * 2. - 1.;
                const double v = mp_u01(b.b) * 2. - 1.;
                const double r = u * u + v * v;
                if (r == 0. || r > 1.) {  // normal.rs:22
                    ++att;
                } else {
                    const int q = p * NS + sidx;
#pragma unroll
                    for (int qq = 0; qq < M; ++qq) {
                        pu[qq] = (qq == q) ? u : pu[qq];
                        pr[qq] = (qq == q) ? r : pr[qq];
                    }
                    att = 0;
                    if (++sidx == ns) { sidx = 0; ++p; }
                }
            }
        }
        // ---- phase 2: the model kernel in Generate mode --------------------------------------
#pragma unroll
        for (int p = 0; p < ITEMS; ++p) {
            const u64 i = i0 + (u64)p;
            if (i < n) {
                double prev[D], next[D];
                if (inv) {
                    // the last (sharded) resample left the parents' states where the all-to-all put them: slot i's row
                    // {x[0..D), parent id} is row inv[i] of the exchange buffer (res_x here)
                    const double* row = res_x + (u64)inv[i] * (u64)(D + 1);
#pragma unroll
                    for (int d = 0; d < D; ++d) prev[d] = row[d];
                } else if (perm) {
                    // the last resample left the states in bin-segment order (k_resolve_bins): slot i's state sits at
                    // segment (bin, chunk of i) position rank, (bin << 10 | rank) = perm[i]
                    const uint32_t pr_ = perm[i];
                    const u64 pos = MP_SEG_POS(pr_ >> 10, i >> 10, pr_ & 1023u, nchunks);
                    if constexpr (D == 1) {
                        prev[0] = res_x[pos];              // k_resolve_bins had it in the table row it found
                    } else {
                        // wider states are gathered here, straight from the parent's (particle-major) row of the
                        // pre-resample buffer: one line per particle, hidden under this kernel's arithmetic
                        const double* src = x_in + (u64)res_parent[pos] * D;
#pragma unroll
                        for (int d = 0; d < D; ++d) prev[d] = src[d];
                    }
                } else {
#pragma unroll
                    for (int d = 0; d < D; ++d) prev[d] = (t == 0) ? s0.v[d] : x_in[i * D + d];
                }
                mp_stream rng;
                rng.k0 = k0; rng.k1 = k1; rng.slot = (uint32_t)(slot_offset + i); rng.step = (uint32_t)t;
                mp_generate_handler<Model> g(rng, obs.v, &pu[p * NS], &pr[p * NS]);
                model(g, t, prev, next);
#pragma unroll
                for (int d = 0; d < D; ++d) x_out[i * D + d] = next[d];
                // particle_filter.rs:68 (init: overwrite) / :81 (accumulate); overwrite == 2: the log-weights are known to
                // be all zero after a resample (log_weights.fill(0.), :114) and are not re-read
                const double w = overwrite == 1 ? g.weight : (overwrite == 2 ? 0. + g.weight : logw[i] + g.weight);
                logw[i] = w;
                lw[rd * ITEMS + p] = w;
                xv[rd * ITEMS + p] = next[0];
            }
        }
    }
    // ---- level 0 of normalize_weights for this tile, while everything is still in registers ----
    normalize_tile<THREADS>(lw, xv, n, blockIdx.x, cx, guide, tile_m, tile_W, tile_W2);
}

// ---------------------------------------------------------------------------------------------
// level 1: the tile table of a workgroup.  T_b = rint((double)W_b * mp_exp(m_b - m) * 2^(S-51)), inclusive prefix in
// s_incl[nt], W_b in s_W[nt]; returns the global max m (every thread).  s_red needs THREADS/64 doubles, s_wtot
// THREADS/64 u64.
// ---------------------------------------------------------------------------------------------
template <int THREADS>
__device__ __forceinline__ double block_tile_table(const double* __restrict__ tile_m, const u64* __restrict__ tile_W, int nt, int S,
                                                   u64* s_incl, u64* s_W, double* s_red, u64* s_wtot) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (nt + THREADS - 1) / THREADS;
    const int b0 = tid * per;
    double m = MP_NEG_INF;
    for (int j = 0; j < per; ++j)
        if (b0 + j < nt) m = fmax(m, tile_m[b0 + j]);
    m = wave_max(m);
    if (lane == 0) s_red[wave] = m;
    __syncthreads();
    m = s_red[0];
#pragma unroll
    for (int w = 1; w < THREADS / 64; ++w) m = fmax(m, s_red[w]);
    const bool ok = (m > MP_NEG_INF) && (m < MP_INF);
    const double sc = mp_u2f((u64)(1023 + S - FIX_BITS) << 52);  // 2^(S-51)
    u64 run = 0;
    for (int j = 0; j < per; ++j) {
        const int idx = b0 + j;
        if (idx < nt) {
            const u64 W = tile_W[idx];
            const double f = ok ? mp_exp(tile_m[idx] - m) : 0.;
            run += mp_quantize((double)W * f * sc, 1.0);
            s_incl[idx] = run;
            s_W[idx] = W;
        }
    }
    const u64 incl = wave_incl_scan_u64(run, lane);
    if (lane == 63) s_wtot[wave] = incl;
    __syncthreads();
    u64 woff = 0;
    for (int k = 0; k < wave; ++k) woff += s_wtot[k];
    const u64 off = woff + (incl - run);
    for (int j = 0; j < per; ++j) {
        const int idx = b0 + j;
        if (idx < nt) s_incl[idx] += off;
    }
    __syncthreads();
    return m;
}
// tile-local target of residual r in (0, T] of a tile with totals (W, T):
// lt = clamp((u64)ceil((double)r * ((double)W / (double)T)), 1, W)
__device__ __forceinline__ u64 mp_local_target(u64 r, u64 W, u64 T) {
    const double ratio = (double)W / (double)T;
    const double v = ceil((double)r * ratio);
    u64 x = (v >= 1.) ? (u64)v : 1ull;
    if (x > W) x = W;
    if (x < 1ull) x = 1ull;
    return x;
}
// Q2 = sum_b rint((double)W2_b * mp_exp(2 (m_b - m)) * 2^(S-51)) by one workgroup; result valid in thread 0
template <int THREADS>
__device__ __forceinline__ u64 block_sum_T2(const double* __restrict__ tile_m, const u64* __restrict__ tile_W2, int nt, int S, double m, u64* s_wtot) {
    const bool ok = (m > MP_NEG_INF) && (m < MP_INF);
    const double sc = mp_u2f((u64)(1023 + S - FIX_BITS) << 52);
    u64 q2 = 0;
    for (int j = threadIdx.x; j < nt; j += THREADS) {
        const double f2 = ok ? mp_exp(2. * (tile_m[j] - m)) : 0.;
        q2 += mp_quantize((double)tile_W2[j] * f2 * sc, 1.0);
    }
    q2 = wave_sum_u64(q2);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_wtot[threadIdx.x >> 6] = q2;
    __syncthreads();
    u64 Q2 = 0;
    if (threadIdx.x == 0)
        for (int k = 0; k < THREADS / 64; ++k) Q2 += s_wtot[k];
    return Q2;
}
__device__ __forceinline__ void finalize_scalars(u64 Q, u64 Q2, int S, double* L_out, double* ess_out, double m) {
    const double inv = mp_u2f((u64)(1023 - S) << 52);  // 2^-S
    const double Qs = (double)Q * inv, Q2s = (double)Q2 * inv;
    *L_out = m + mp_log(Qs);
    *ess_out = (Qs * Qs) / Q2s;
}
// thread 0 of a workgroup that has the tile table folds a normalisation into the filter scalars
__device__ __forceinline__ void fold_scalars(mp_dev_scalars* scal, u64 Q, u64 Q2, int S, double m, u64 n_global, int mode) {
    double L, ess;
    finalize_scalars(Q, Q2, S, &L, &ess, m);
    scal->m = m;
    if (!(m > MP_NEG_INF) || !(m < MP_INF) || Q == 0) scal->degenerate = 1;
    if (mode == 0) {  // resample (particle_filter.rs:104-105)
        scal->L = L;
        scal->ess_stale = ess;
        scal->Q = Q;
        scal->Q2 = Q2;
        scal->log_ml += L - mp_log((double)n_global);
    } else {          // query (particle_filter.rs:119-121; fresh ESS)
        scal->L = L;
        scal->ess_fresh = ess;
        scal->lml_fresh = scal->log_ml + L - mp_log((double)n_global);
    }
}

__device__ __forceinline__ mp_cx load_row_nt(const mp_cx* p) {
    typedef u64 u64x2 __attribute__((ext_vector_type(2)));
    const u64x2 v = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(p));
    mp_cx r;
    r.cum = v.x;
    r.x0 = __builtin_bit_cast(double, (u64)v.y);
    return r;
}

// target = max(1, ceil(k * Q / 2^52)), k < 2^52, Q < 2^63
__device__ __forceinline__ u64 mp_target(u64 k52, u64 Q) {
    u64 lo = k52 * Q;
    u64 hi = __umul64hi(k52, Q);
    const u64 add = (1ull << 52) - 1ull;
    const u64 lo2 = lo + add;
    hi += (lo2 < lo) ? 1ull : 0ull;
    const u64 t = (hi << 12) | (lo2 >> 52);
    return t < 1ull ? 1ull : t;
}

// Systematic resampling (extension; the reference only has multinomial): one uniform u0 = k32 / 2^32 per resample
// (Philox slot 0, site 1), u_g = (g + u0) / N for global output slot g; target = floor(u_g * Q) + 1, evaluated
// exactly in integers: p = g * 2^32 + k32, A = (p * Q) >> 32, target = A / N + 1   (1 <= target <= Q).
__device__ __forceinline__ u64 mp_target_systematic(u64 g, uint32_t k32, u64 Q, u64 n_global) {
    const u64 p = (g << 32) | (u64)k32;
    const u64 lo = p * Q;
    const u64 hi = __umul64hi(p, Q);
    const u64 a_lo = (lo >> 32) | (hi << 32);   // A = (hi:lo) >> 32, A < 2^95
    const u64 a_hi = hi >> 32;                  // < 2^31
    // long division of (a_hi : a_lo) by n_global < 2^32, base 2^32
    u64 r = a_hi % n_global;                    // a_hi / n_global contributes to bits >= 64 of the quotient: zero since A / N < Q < 2^63
    u64 cur = (r << 32) | (a_lo >> 32);
    const u64 q1 = cur / n_global;
    r = cur % n_global;
    cur = (r << 32) | (a_lo & 0xFFFFFFFFull);
    const u64 q0 = cur / n_global;
    return ((q1 << 32) | q0) + 1ull;
}
__device__ __forceinline__ uint32_t mp_systematic_k32(uint32_t rc, uint32_t k0, uint32_t k1) {
    const mp_u64x2 r = mp_philox4x32_10(0u, rc, ((uint32_t)MP_DOM_RESAMPLE << 16) | 1u, 0u, k0, k1);
    return (uint32_t)(r.a >> 32);
}
// Stratified resampling (extension): the same lattice with one uniform PER output slot, u_g = (g + k32_g / 2^32) / N
// (Philox slot g, site 2); parents still come out sorted.
__device__ __forceinline__ uint32_t mp_stratified_k32(u64 g, uint32_t rc, uint32_t k0, uint32_t k1) {
    const mp_u64x2 r = mp_philox4x32_10((uint32_t)g, rc, ((uint32_t)MP_DOM_RESAMPLE << 16) | 2u, 0u, k0, k1);
    return (uint32_t)(r.a >> 32);
}
// target of global output slot g under scheme 1 (systematic, shared k32) or 2 (stratified)
__device__ __forceinline__ u64 mp_target_lattice(int scheme, u64 g, uint32_t shared_k32, uint32_t rc, uint32_t k0, uint32_t k1, u64 Q, u64 n_global) {
    return mp_target_systematic(g, scheme == 2 ? mp_stratified_k32(g, rc, k0, k1) : shared_k32, Q, n_global);
}

// Tile of a global target: tile totals are nearly equal (each sums 2048 weights), so target * nt / Q lands within a
// tile or two of the answer; walk from there.  Same result as a lower_bound over s_incl, fewer LDS reads.
__device__ __forceinline__ uint32_t tile_of_target(const u64* s_incl, uint32_t nt, u64 target, double nt_over_Q) {
    int b = (int)((double)target * nt_over_Q);
    if (b > (int)nt - 1) b = (int)nt - 1;
    if (b < 0) b = 0;
    while (b > 0 && s_incl[b - 1] >= target) --b;          // first b with incl[b] >= target ...
    while (b < (int)nt - 1 && s_incl[b] < target) ++b;     // ... from either side
    return (uint32_t)b;
}
// global target -> (tile, tile-local target, guide slot)
__device__ __forceinline__ void mp_locate(const u64* s_incl, const u64* s_W, uint32_t nt, u64 target, double nt_over_Q, uint32_t* tile, u64* lt,
                                          uint32_t* gslot) {
    const uint32_t b = tile_of_target(s_incl, nt, target, nt_over_Q);
    const u64 excl = b ? s_incl[b - 1] : 0ull;
    const u64 T = s_incl[b] - excl;
    const u64 W = s_W[b];
    const u64 x = mp_local_target(target - excl, W, T);
    uint32_t g = (uint32_t)(x >> mp_guide_shift(W));
    if (g > GUIDE_N - 1) g = GUIDE_N - 1;
    *tile = b; *lt = x; *gslot = b * (uint32_t)GUIDE_N + g;
}

// ---------------------------------------------------------------------------------------------
// K3 (single-kernel form): draw, search, gather, reset.  Used for importance_resampling's M draws (domain IS) and
// for systematic resampling (sorted parents: coalesced by construction).  n_out draws over a table of n rows.
// ---------------------------------------------------------------------------------------------
template <int SCHEME>
__global__ __launch_bounds__(K3_THREADS) void k_resample_gather(u64 n, u64 n_out, u64 n_global, u64 slot_offset, uint32_t domain,
                                                                uint32_t k0, uint32_t k1, uint32_t rc, int S, int D,
                                                                const mp_cx* __restrict__ cx, const unsigned short* __restrict__ guide,
                                                                const double* __restrict__ tile_m, const u64* __restrict__ tile_W,
                                                                const u64* __restrict__ tile_W2, int nt,
                                                                const double* __restrict__ x_old, double* __restrict__ x_new,
                                                                uint32_t* __restrict__ parent, double* __restrict__ logw, mp_dev_scalars* scal) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* s_incl = reinterpret_cast<u64*>(smem);
    u64* s_W = s_incl + nt;
    double* s_red = reinterpret_cast<double*>(s_W + nt);
    u64* s_wtot = reinterpret_cast<u64*>(s_red + K3_THREADS / 64);
    const double m = block_tile_table<K3_THREADS>(tile_m, tile_W, nt, S, s_incl, s_W, s_red, s_wtot);
    const u64 Q = s_incl[nt - 1];
    if (blockIdx.x == 0 && scal != nullptr) {  // workgroup-uniform: fold this normalisation into the filter scalars
        const u64 Q2 = block_sum_T2<K3_THREADS>(tile_m, tile_W2, nt, S, m, s_wtot);
        if (threadIdx.x == 0) fold_scalars(scal, Q, Q2, S, m, n_global, 0);
    }
    const uint32_t sys_k32 = SCHEME == 1 ? mp_systematic_k32(rc, k0, k1) : 0u;
    const double nt_over_Q = (double)nt / (double)Q;  // only a starting guess for the tile walk: no effect on results
    for (u64 i0 = (u64)blockIdx.x * (K3_THREADS * K3_ITEMS) + threadIdx.x; i0 < n_out; i0 += (u64)gridDim.x * (K3_THREADS * K3_ITEMS)) {
        u64 lt[K3_ITEMS], tbase[K3_ITEMS];
        uint32_t tlen[K3_ITEMS], j[K3_ITEMS], gslot[K3_ITEMS];
#pragma unroll
        for (int k = 0; k < K3_ITEMS; ++k) {
            const u64 i = i0 + (u64)k * K3_THREADS;
            u64 target;
            if (SCHEME != 0) {
                target = mp_target_lattice(SCHEME, slot_offset + (i < n_out ? i : 0), sys_k32, rc, k0, k1, Q, n_global);
            } else {
                const mp_u64x2 r = mp_philox4x32_10((uint32_t)(slot_offset + i), rc, (domain << 16), 0u, k0, k1);
                target = mp_target(mp_u52(r.a), Q);
            }
            uint32_t b;
            mp_locate(s_incl, s_W, (uint32_t)nt, target, nt_over_Q, &b, &lt[k], &gslot[k]);
            tbase[k] = (u64)b * TILE;
            tlen[k] = (uint32_t)((n - tbase[k]) < (u64)TILE ? (n - tbase[k]) : (u64)TILE);
        }
#pragma unroll
        for (int k = 0; k < K3_ITEMS; ++k) j[k] = guide[gslot[k]];
        mp_cx r0[K3_ITEMS], r1[K3_ITEMS];
#pragma unroll
        for (int k = 0; k < K3_ITEMS; ++k) {
            if (j[k] > tlen[k] - 1) j[k] = tlen[k] - 1;
            const uint32_t j1 = (j[k] + 1 < tlen[k]) ? j[k] + 1 : j[k];
            r0[k] = load_row_nt(cx + tbase[k] + j[k]);
            r1[k] = load_row_nt(cx + tbase[k] + j1);
        }
#pragma unroll
        for (int k = 0; k < K3_ITEMS; ++k) {
            const u64 i = i0 + (u64)k * K3_THREADS;
            mp_cx row = r0[k];
            uint32_t jj = j[k];
            if (row.cum < lt[k] && jj + 1 < tlen[k]) {    // first row with cum >= lt
                row = r1[k];
                ++jj;
                while (row.cum < lt[k] && jj + 1 < tlen[k]) {
                    ++jj;
                    row = load_row_nt(cx + tbase[k] + jj);
                }
            }
            if (i < n_out) {
                const u64 p = tbase[k] + jj;
                parent[i] = (uint32_t)p;
                if (x_new) {
                    x_new[i * D] = row.x0;                // traces[i] = traces[parents[i]].clone()
                    for (int d = 1; d < D; ++d) x_new[i * D + d] = x_old[p * D + d];
                }
                if (logw) logw[i] = 0.;                   // log_weights.fill(0.)
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// XCD-binned multinomial resampling (same parents per slot as k_resample_gather, bit for bit)
// ---------------------------------------------------------------------------------------------
// The row table (16 B x N) does not fit one XCD's 4 MB L2, so random row reads cross the fabric a full line at a
// time.  But the top 3 bits of a draw's uniform say which EIGHTH of the CDF it lands in.  So:
//   K3a k_bin_draws     every chunk of 1024 output slots: tile table, Philox, target, tile, guide lookup (the 2 MB
//                       guide is L2-resident everywhere), stable split of the chunk's draws into the 8 bins:
//                       segment [bin][chunk][<=1024] of (tile-local target, start row) and perm[slot] = (bin << 10 | pos).
//   K3b k_resolve_bins  workgroup (group of 8 chunks, bin b) with blockIdx % 8 == b — workgroups are dealt
//                       round-robin over the 8 XCDs, so the row lookups of bin b run on one XCD whose L2 then holds
//                       that eighth of the table (speed only: any placement gives the same result).
// Measured (profiles/r01): L2 hit rate 0.58 -> 0.86, fabric traffic 107 -> 52 MB per resample of 2^20, 46 -> 30 us.
__global__ __launch_bounds__(BIN_THREADS) void k_bin_draws(u64 n, u64 n_global, u64 slot_offset, uint32_t k0, uint32_t k1, uint32_t rc, int S, int nchunks,
                                                           const double* __restrict__ tile_m, const u64* __restrict__ tile_W,
                                                           const u64* __restrict__ tile_W2, int nt,
                                                           const unsigned short* __restrict__ guide,
                                                           u64* __restrict__ seg_lt, uint32_t* __restrict__ seg_row,
                                                           unsigned short* __restrict__ perm, unsigned short* __restrict__ seg_cnt,
                                                           mp_dev_scalars* scal) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NW = BIN_THREADS / 64;
    u64* s_incl = reinterpret_cast<u64*>(smem);                        // [nt]
    u64* s_W = s_incl + nt;                                            // [nt]
    double* s_red = reinterpret_cast<double*>(s_W + nt);               // [NW]
    u64* s_wtot = reinterpret_cast<u64*>(s_red + NW);                  // [NW]
    uint32_t* s_wcnt = reinterpret_cast<uint32_t*>(s_wtot + NW);       // [BIN_ITEMS][NW][8] counts
    uint32_t* s_woff = s_wcnt + BIN_ITEMS * NW * 8;                    // same shape: exclusive offsets
    const double m = block_tile_table<BIN_THREADS>(tile_m, tile_W, nt, S, s_incl, s_W, s_red, s_wtot);
    const u64 Q = s_incl[nt - 1];
    const int c = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blockIdx.x == 0) {  // fold this normalisation into the filter scalars
        const u64 Q2 = block_sum_T2<BIN_THREADS>(tile_m, tile_W2, nt, S, m, s_wtot);
        if (threadIdx.x == 0) fold_scalars(scal, Q, Q2, S, m, n_global, 0);
    }

    u64 lt[BIN_ITEMS];
    uint32_t gslot[BIN_ITEMS], tile_of[BIN_ITEMS];
    int bin[BIN_ITEMS];
    uint32_t rank_in_wave[BIN_ITEMS];
    const double nt_over_Q = (double)nt / (double)Q;  // only a starting guess for the tile walk: no effect on results
#pragma unroll
    for (int q = 0; q < BIN_ITEMS; ++q) {
        const u64 i = (u64)c * BIN_CHUNK + (u64)q * BIN_THREADS + threadIdx.x;
        const mp_u64x2 r = mp_philox4x32_10((uint32_t)(slot_offset + i), rc, ((uint32_t)MP_DOM_RESAMPLE << 16), 0u, k0, k1);
        const u64 k52 = mp_u52(r.a);
        mp_locate(s_incl, s_W, (uint32_t)nt, mp_target(k52, Q), nt_over_Q, &tile_of[q], &lt[q], &gslot[q]);
        bin[q] = (i < n) ? (int)(k52 >> 49) : -1;
        rank_in_wave[q] = 0;
#pragma unroll
        for (int bb = 0; bb < 8; ++bb) {
            const u64 bal = __ballot(bin[q] == bb);
            if (bin[q] == bb) rank_in_wave[q] = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) s_wcnt[(q * NW + wave) * 8 + bb] = (uint32_t)__popcll(bal);
        }
    }
    // the guide lookups go out now (the guide is L2-resident on every XCD) and land while the offsets are built
    uint32_t j0[BIN_ITEMS];
#pragma unroll
    for (int q = 0; q < BIN_ITEMS; ++q) j0[q] = guide[gslot[q]];
    __syncthreads();
    // exclusive offsets in the stable order: item q-major (slots q*256 .. q*256+255), then wave, then lane == increasing slot
    if (threadIdx.x < 8) {
        uint32_t run = 0;
        for (int q = 0; q < BIN_ITEMS; ++q)
            for (int w = 0; w < NW; ++w) {
                s_woff[(q * NW + w) * 8 + threadIdx.x] = run;
                run += s_wcnt[(q * NW + w) * 8 + threadIdx.x];
            }
        seg_cnt[(u64)threadIdx.x * nchunks + c] = (unsigned short)run;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < BIN_ITEMS; ++q) {
        if (bin[q] >= 0) {
            const uint32_t pos = s_woff[(q * NW + wave) * 8 + bin[q]] + rank_in_wave[q];
            const u64 sp = MP_SEG_POS(bin[q], c, pos, nchunks);
            const u64 tbase = (u64)tile_of[q] * TILE;
            const uint32_t tlen = (uint32_t)((n - tbase) < (u64)TILE ? (n - tbase) : (u64)TILE);
            const uint32_t jj = j0[q] > tlen - 1 ? tlen - 1 : j0[q];
            seg_lt[sp] = lt[q];
            seg_row[sp] = (uint32_t)tbase + jj;   // row where the forward scan starts
            perm[(u64)c * BIN_CHUNK + q * BIN_THREADS + threadIdx.x] = (unsigned short)((bin[q] << 10) | pos);
        }
    }
}

// K3b: pure lookup, two dependent hops (segment entry -> table rows), all inside the bin's eighth of the table.
// Thread (quad, e) = (tid >> 7, tid & 127) owns entry e of the 4 segments of chunks group*8 + quad*4 + {0..3}; a
// segment holds 128 +- 11 entries, so nearly every lane is live and each has 4 independent chains in flight.
// Results stay in SEGMENT order (res_x[d][bin][chunk][pos], res_parent likewise: coalesced stores); the next
// k_propagate reads its inputs through perm[], k_unpermute materialises slot order when the host asks.
__global__ __launch_bounds__(K3_THREADS) void k_resolve_bins(u64 n, int D, int nchunks, const u64* __restrict__ seg_lt,
                                                             const uint32_t* __restrict__ seg_row, const unsigned short* __restrict__ seg_cnt,
                                                             const mp_cx* __restrict__ cx,
                                                             const double* __restrict__ x_old, double* __restrict__ res_x, u64 res_stride,
                                                             uint32_t* __restrict__ res_parent) {
    const int bin = blockIdx.x & 7;
    const int group = blockIdx.x >> 3;
    const int e0 = threadIdx.x & 127, quad = threadIdx.x >> 7;
    int cnt[K3_ITEMS], chunk_of[K3_ITEMS];
    u64 lt[K3_ITEMS], spos[K3_ITEMS];
    uint32_t row0[K3_ITEMS];
#pragma unroll
    for (int k = 0; k < K3_ITEMS; ++k) {
        const int c = group * BIN_GROUP + quad * K3_ITEMS + k;
        const bool ok = c < nchunks;
        chunk_of[k] = ok ? c : 0;
        cnt[k] = ok ? (int)seg_cnt[(u64)bin * nchunks + c] : 0;
        spos[k] = MP_SEG_POS(bin, chunk_of[k], e0, nchunks);
        lt[k] = seg_lt[spos[k]];        // in bounds for every thread; masked by cnt below
        row0[k] = seg_row[spos[k]];
    }
    // first row >= lt, walking forward from `row` inside its tile; r0/r1 = that row and the next one, already loaded
    auto finish = [&](u64 ltx, uint32_t row, u64 sp, mp_cx r0, mp_cx r1) {
        const u64 tend = (((u64)row / TILE) + 1) * TILE;
        const u64 last = (tend < n ? tend : n) - 1;      // last row of the tile
        mp_cx cur = r0;
        u64 p = row;
        if (cur.cum < ltx && p < last) {
            cur = r1;
            ++p;
            while (cur.cum < ltx && p < last) {
                ++p;
                cur = cx[p];
            }
        }
        res_parent[sp] = (uint32_t)p;
        res_x[sp] = cur.x0;
        // D > 1: the rest of the state is gathered by the next k_propagate (or k_unpermute) from res_parent
    };
    bool live[K3_ITEMS];
    mp_cx r0[K3_ITEMS], r1[K3_ITEMS];
#pragma unroll
    for (int k = 0; k < K3_ITEMS; ++k) {
        live[k] = e0 < cnt[k];
        if (live[k]) {
            const u64 tend = (((u64)row0[k] / TILE) + 1) * TILE;
            const u64 last = (tend < n ? tend : n) - 1;
            r0[k] = cx[row0[k]];
            r1[k] = cx[(u64)row0[k] + ((u64)row0[k] < last ? 1 : 0)];
        } else {
            r0[k].cum = ~0ull; r0[k].x0 = 0.; r1[k] = r0[k];
        }
    }
#pragma unroll
    for (int k = 0; k < K3_ITEMS; ++k)
        if (live[k]) finish(lt[k], row0[k], spos[k], r0[k], r1[k]);
    // entries 128.. of a segment (about 5 % of the entries: the upper tail of Binomial(1024, 1/8))
#pragma unroll
    for (int k = 0; k < K3_ITEMS; ++k) {
        for (int e = 128 + e0; e < cnt[k]; e += 128) {
            const u64 sp = MP_SEG_POS(bin, chunk_of[k], e, nchunks);
            const uint32_t row = seg_row[sp];
            const u64 tend = (((u64)row / TILE) + 1) * TILE;
            const u64 last = (tend < n ? tend : n) - 1;
            const mp_cx a = cx[row];
            const mp_cx bq = cx[(u64)row + ((u64)row < last ? 1 : 0)];
            finish(seg_lt[sp], row, sp, a, bq);
        }
    }
}

// slot order from segment order: traces[i] = traces[parents[i]].clone(); log_weights.fill(0.) (particle_filter.rs:109-114)
__global__ void k_unpermute(u64 n, int D, int nchunks, const unsigned short* __restrict__ perm, const double* __restrict__ res_x, u64 res_stride,
                            const uint32_t* __restrict__ res_parent, const double* __restrict__ x_old, double* __restrict__ x_new,
                            uint32_t* __restrict__ parent, double* __restrict__ logw) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t pr = perm[i];
    const u64 pos = MP_SEG_POS(pr >> 10, i >> 10, pr & 1023u, nchunks);
    const uint32_t p = res_parent[pos];
    if (D == 1) {
        x_new[i] = res_x[pos];
    } else {
        for (int d = 0; d < D; ++d) x_new[i * D + d] = x_old[(u64)p * D + d];
    }
    parent[i] = p;
    logw[i] = 0.;
}

// Level 1 on its own (one workgroup): mode 1 = query (log_marginal_likelihood_estimate / fresh ESS), mode 0 = fold a
// sharded resample, mode 2 = importance sampling (L and log_ml = L - ln N, importance.rs:21-22).
__global__ __launch_bounds__(K3_THREADS) void k_finalize_tiles(const double* __restrict__ tile_m, const u64* __restrict__ tile_W,
                                                               const u64* __restrict__ tile_W2, int nt, int S, u64 n_global, int mode,
                                                               mp_dev_scalars* scal, mp_dev_scalars* undo = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* s_incl = reinterpret_cast<u64*>(smem);
    u64* s_W = s_incl + nt;
    double* s_red = reinterpret_cast<double*>(s_W + nt);
    u64* s_wtot = reinterpret_cast<u64*>(s_red + K3_THREADS / 64);
    const double m = block_tile_table<K3_THREADS>(tile_m, tile_W, nt, S, s_incl, s_W, s_red, s_wtot);
    const u64 Q = s_incl[nt - 1];
    const u64 Q2 = block_sum_T2<K3_THREADS>(tile_m, tile_W2, nt, S, m, s_wtot);
    if (threadIdx.x == 0) {
        if (undo) *undo = *scal;   // a fixed-capacity exchange that overflows puts these back
        if (mode == 2) {
            double L, ess;
            finalize_scalars(Q, Q2, S, &L, &ess, m);
            scal->m = m;
            if (!(m > MP_NEG_INF) || !(m < MP_INF) || Q == 0) scal->degenerate = 1;
            scal->L = L;
            scal->lml_fresh = L - mp_log((double)n_global);
        } else {
            fold_scalars(scal, Q, Q2, S, m, n_global, mode);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// sharded filter phases (include/modppl_hip.h "sharded filter").  Shards are tile-aligned, so a shard's tiles are
// tiles of the job; every rank gathers all tiles' (m, W, W2) and builds the same table.
// ---------------------------------------------------------------------------------------------
constexpr int SH_THREADS = 256;
constexpr int SH_MAX_WORLD = 64;

// pass 1: target of every local slot -> owner rank, tile inside the owner's shard, tile-local target; owner histogram
__global__ __launch_bounds__(SH_THREADS) void k_shard_targets(u64 n, u64 n_global, u64 slot_offset, uint32_t k0, uint32_t k1, uint32_t rc,
                                                              int systematic, int S, const double* __restrict__ tm_all,
                                                              const u64* __restrict__ tW_all, int nt_all, int nt_local, int world,
                                                              unsigned char* __restrict__ dest, u64* __restrict__ lt_out,
                                                              uint32_t* __restrict__ tile_out, uint32_t* __restrict__ blockcount) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* s_incl = reinterpret_cast<u64*>(smem);
    u64* s_W = s_incl + nt_all;
    double* s_red = reinterpret_cast<double*>(s_W + nt_all);
    u64* s_wtot = reinterpret_cast<u64*>(s_red + SH_THREADS / 64);
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(s_wtot + SH_THREADS / 64);  // [SH_MAX_WORLD]
    if (threadIdx.x < SH_MAX_WORLD) s_cnt[threadIdx.x] = 0;
    block_tile_table<SH_THREADS>(tm_all, tW_all, nt_all, S, s_incl, s_W, s_red, s_wtot);
    const u64 Q = s_incl[nt_all - 1];
    const double nt_over_Q = (double)nt_all / (double)Q;
    const u64 i = (u64)blockIdx.x * SH_THREADS + threadIdx.x;
    if (i < n) {
        u64 target;
        if (systematic) {   // 1 systematic, 2 stratified
            target = mp_target_lattice(systematic, slot_offset + i, systematic == 1 ? mp_systematic_k32(rc, k0, k1) : 0u, rc, k0, k1, Q, n_global);
        } else {
            const mp_u64x2 r = mp_philox4x32_10((uint32_t)(slot_offset + i), rc, ((uint32_t)MP_DOM_RESAMPLE << 16), 0u, k0, k1);
            target = mp_target(mp_u52(r.a), Q);
        }
        uint32_t b, gs;
        u64 lt;
        mp_locate(s_incl, s_W, (uint32_t)nt_all, target, nt_over_Q, &b, &lt, &gs);
        const int s = (int)(b / (uint32_t)nt_local);
        dest[i] = (unsigned char)s;
        lt_out[i] = lt;
        tile_out[i] = b % (uint32_t)nt_local;
        atomicAdd(&s_cnt[s], 1u);
    }
    __syncthreads();
    if (threadIdx.x < world) blockcount[(u64)blockIdx.x * world + threadIdx.x] = s_cnt[threadIdx.x];
}
// pass 2 (one workgroup per owner): per-owner totals and exclusive per-workgroup offsets
__global__ __launch_bounds__(SH_THREADS) void k_shard_offsets(const uint32_t* __restrict__ blockcount, int nblk, int world,
                                                              uint32_t* __restrict__ blockoff, long long* __restrict__ counts) {
    __shared__ uint32_t s_wave[SH_THREADS / 64];
    const int r = blockIdx.x;
    const int per = (nblk + SH_THREADS - 1) / SH_THREADS;
    const int b0 = threadIdx.x * per, b1 = (b0 + per < nblk) ? b0 + per : nblk;
    uint32_t mine = 0;
    for (int b = b0; b < b1; ++b) mine += blockcount[(u64)b * world + r];
    // exclusive scan of `mine` over the workgroup
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(inc, o);
        if (lane >= o) inc += v;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (int w = 0; w < SH_THREADS / 64; ++w) {
        if (w < wave) base += s_wave[w];
        total += s_wave[w];
    }
    uint32_t run = base + inc - mine;
    for (int b = b0; b < b1; ++b) {
        blockoff[(u64)b * world + r] = run;
        run += blockcount[(u64)b * world + r];
    }
    if (threadIdx.x == 0) counts[r] = (long long)total;
}
// pass 3: stable pack of the requests (tile in owner, tile-local target) grouped by owner
__global__ __launch_bounds__(SH_THREADS) void k_shard_pack(u64 n, const unsigned char* __restrict__ dest, const u64* __restrict__ lt_in,
                                                           const uint32_t* __restrict__ tile_in, const uint32_t* __restrict__ blockoff,
                                                           const long long* __restrict__ counts, int world, u64* __restrict__ req_out,
                                                           uint32_t* __restrict__ req_slot) {
    __shared__ uint32_t s_wcnt[SH_THREADS / 64][SH_MAX_WORLD];
    __shared__ u64 s_gstart[SH_MAX_WORLD];
    const u64 i = (u64)blockIdx.x * SH_THREADS + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = (i < n) ? (int)dest[i] : -1;
    if (threadIdx.x == 0) {
        u64 run = 0;
        for (int r = 0; r < world; ++r) { s_gstart[r] = run; run += (u64)counts[r]; }
    }
    uint32_t my_rank_in_wave = 0;
    for (int r = 0; r < world; ++r) {
        const u64 bal = __ballot(s == r);
        if (s == r) my_rank_in_wave = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) s_wcnt[wave][r] = (uint32_t)__popcll(bal);
    }
    __syncthreads();
    if (s >= 0) {
        uint32_t before = 0;
        for (int w = 0; w < wave; ++w) before += s_wcnt[w][s];
        const u64 pos = s_gstart[s] + blockoff[(u64)blockIdx.x * world + s] + before + my_rank_in_wave;
        req_out[2 * pos] = (u64)tile_in[i];
        req_out[2 * pos + 1] = lt_in[i];
        req_slot[pos] = (uint32_t)i;
    }
}
// owner side: (tile, tile-local target) -> parent rows
__global__ __launch_bounds__(K3_THREADS) void k_shard_resolve(u64 n, u64 n_req, u64 slot_offset, int D, const u64* __restrict__ req,
                                                              const mp_cx* __restrict__ cx, const unsigned short* __restrict__ guide,
                                                              const u64* __restrict__ tile_W, const double* __restrict__ x,
                                                              double* __restrict__ rows) {
    for (u64 q = (u64)blockIdx.x * K3_THREADS + threadIdx.x; q < n_req; q += (u64)gridDim.x * K3_THREADS) {
        const u64 b = req[2 * q];
        const u64 lt = req[2 * q + 1];
        const int shift = mp_guide_shift(tile_W[b]);
        const u64 tbase = b * TILE;
        const uint32_t tlen = (uint32_t)((n - tbase) < (u64)TILE ? (n - tbase) : (u64)TILE);
        uint32_t g = (uint32_t)(lt >> shift);
        if (g > GUIDE_N - 1) g = GUIDE_N - 1;
        uint32_t j = guide[b * GUIDE_N + g];
        if (j > tlen - 1) j = tlen - 1;
        mp_cx row = load_row_nt(cx + tbase + j);
        while (row.cum < lt && j + 1 < tlen) {
            ++j;
            row = load_row_nt(cx + tbase + j);
        }
        const u64 p = tbase + j;
        double* out = rows + q * (u64)(D + 1);
        out[0] = row.x0;
        for (int d = 1; d < D; ++d) out[d] = x[p * D + d];
        out[D] = (double)(slot_offset + p);
    }
}
// requester side
__global__ __launch_bounds__(SH_THREADS) void k_shard_scatter(u64 n, int D, const double* __restrict__ rows, const uint32_t* __restrict__ req_slot,
                                                              double* __restrict__ x_new, uint32_t* __restrict__ parent, double* __restrict__ logw) {
    const u64 pos = (u64)blockIdx.x * SH_THREADS + threadIdx.x;
    if (pos < n) {
        const uint32_t i = req_slot[pos];
        const double* in = rows + pos * (u64)(D + 1);
        for (int d = 0; d < D; ++d) x_new[(u64)i * D + d] = in[d];
        parent[i] = (uint32_t)in[D];
        logw[i] = 0.;
    }
}


// ---- fixed-capacity exchange (no host round trip) -------------------------------------------------
// Gathered tiles arrive rank-major, [world][3][nt_local] 8-byte words (row 0: bits of the f64 tile maxima, row 1: W,
// row 2: W2).  Requests travel in fixed segments: req[dst][cap + 1][2], entry 0 = {count, 0}; rows likewise
// rows[src][cap][D + 1].  A pair (src, dst) exchanging more than `cap` draws sets the sticky overflow flag (the filter
// then reports MP_ERR_UNSUPPORTED at the next synchronising call instead of continuing with dropped draws).
__global__ __launch_bounds__(K3_THREADS) void k_unpack_tiles(const u64* __restrict__ packed, int world, int nt_local, double* __restrict__ tm,
                                                             u64* __restrict__ tW, u64* __restrict__ tW2, long long* zero_counts = nullptr) {
    const int i = blockIdx.x * K3_THREADS + threadIdx.x;
    if (zero_counts && i < 512) zero_counts[i] = 0;  // the per-(owner, eighth) request counters of the route that follows (SH_MAX_KEYS)
    if (i < world * nt_local) {
        const int r = i / nt_local, b = i % nt_local;
        const u64* base = packed + (u64)r * 3 * nt_local;
        tm[i] = mp_u2f(base[b]);
        tW[i] = base[nt_local + b];
        tW2[i] = base[2 * nt_local + b];
    }
}
// Fixed-capacity route in ONE pass: target -> owner / tile / local target, and the request is written straight into the
// sub-segment (owner, eighth of the owner's tiles).  Places come from one global atomic per (workgroup, sub-segment), so
// the order of requests inside a sub-segment varies from run to run; the results do not (inv[i] remembers where the
// request of slot i went, which is where its row comes back).  Grouping by eighth lets the owner resolve each group on one XCD, whose L2 then holds
// that eighth of its rows (the same trick as k_bin_draws / k_resolve_bins).
// what the owner-side resolve publishes to host-mapped memory when its last workgroup finishes
struct mp_shard_pub {
    int overflow;
    int degenerate;
    double L;
};
constexpr int SHF_ITEMS = 4;
constexpr int SH_BINS = 8;
constexpr int SH_MAX_KEYS = SH_MAX_WORLD * SH_BINS;
__global__ __launch_bounds__(SH_THREADS) void k_shard_route_fused(u64 n, u64 n_global, u64 slot_offset, uint32_t k0, uint32_t k1, uint32_t rc,
                                                                  int systematic, int S, const double* __restrict__ tm_all,
                                                                  const u64* __restrict__ tW_all, int nt_all, int nt_local, int world, u64 capb,
                                                                  unsigned long long* __restrict__ counts, u64* __restrict__ req_out,
                                                                  uint32_t* __restrict__ inv) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* s_incl = reinterpret_cast<u64*>(smem);
    u64* s_W = s_incl + nt_all;
    double* s_red = reinterpret_cast<double*>(s_W + nt_all);
    u64* s_wtot = reinterpret_cast<u64*>(s_red + SH_THREADS / 64);
    u64* s_base = s_wtot + SH_THREADS / 64;                                   // [keys] start inside the sub-segment
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(s_base + SH_MAX_KEYS);      // [keys] draws of this workgroup per sub-segment
    const int keys = world * SH_BINS;
    for (int k = threadIdx.x; k < keys; k += SH_THREADS) s_cnt[k] = 0;
    block_tile_table<SH_THREADS>(tm_all, tW_all, nt_all, S, s_incl, s_W, s_red, s_wtot);   // ends with a barrier
    const u64 Q = s_incl[nt_all - 1];
    const double nt_over_Q = (double)nt_all / (double)Q;
    const u64 i0 = (u64)blockIdx.x * (SH_THREADS * SHF_ITEMS) + threadIdx.x;
    const uint32_t k32 = systematic == 1 ? mp_systematic_k32(rc, k0, k1) : 0u;
    int key[SHF_ITEMS];
    uint32_t tl[SHF_ITEMS], place[SHF_ITEMS];
    u64 lt[SHF_ITEMS];
#pragma unroll
    for (int k = 0; k < SHF_ITEMS; ++k) {
        const u64 i = i0 + (u64)k * SH_THREADS;
        key[k] = -1;
        if (i < n) {
            u64 target;
            if (systematic) {   // 1 systematic, 2 stratified
                target = mp_target_lattice(systematic, slot_offset + i, k32, rc, k0, k1, Q, n_global);
            } else {
                const mp_u64x2 r = mp_philox4x32_10((uint32_t)(slot_offset + i), rc, ((uint32_t)MP_DOM_RESAMPLE << 16), 0u, k0, k1);
                target = mp_target(mp_u52(r.a), Q);
            }
            uint32_t b, gs;
            mp_locate(s_incl, s_W, (uint32_t)nt_all, target, nt_over_Q, &b, &lt[k], &gs);
            const uint32_t own = b / (uint32_t)nt_local;
            tl[k] = b - own * (uint32_t)nt_local;
            key[k] = (int)(own * SH_BINS + (tl[k] * SH_BINS) / (uint32_t)nt_local);
            place[k] = atomicAdd(&s_cnt[key[k]], 1u);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < keys; k += SH_THREADS)
        s_base[k] = s_cnt[k] ? atomicAdd(&counts[k], (unsigned long long)s_cnt[k]) : 0ull;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SHF_ITEMS; ++k) {
        if (key[k] >= 0) {
            const u64 j = s_base[key[k]] + place[k];
            if (j < capb) {
                ulonglong2* sub = reinterpret_cast<ulonglong2*>(req_out) + (u64)key[k] * (capb + 1);
                sub[j + 1] = make_ulonglong2((u64)tl[k], lt[k]);
                inv[i0 + (u64)k * SH_THREADS] = (uint32_t)((u64)key[k] * capb + j);
            }
        }
    }
}
// Level 1 of a sharded resample + the sub-segment headers {count, "some sub-segment of mine overflowed"}, once every
// workgroup of the route has reserved its places.  One workgroup.
__global__ __launch_bounds__(K3_THREADS) void k_shard_finalize(const double* __restrict__ tile_m, const u64* __restrict__ tile_W,
                                                               const u64* __restrict__ tile_W2, int nt, int S, u64 n_global, mp_dev_scalars* scal,
                                                               mp_dev_scalars* undo, const unsigned long long* __restrict__ counts, int world,
                                                               u64 capb, u64* __restrict__ req_out, int* overflow) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* s_incl = reinterpret_cast<u64*>(smem);
    u64* s_W = s_incl + nt;
    double* s_red = reinterpret_cast<double*>(s_W + nt);
    u64* s_wtot = reinterpret_cast<u64*>(s_red + K3_THREADS / 64);
    const int keys = world * SH_BINS;
    int mine = 0;
    for (int k = threadIdx.x; k < keys; k += K3_THREADS) mine |= (counts[k] > capb) ? 1 : 0;
    const int any = __syncthreads_or(mine);
    for (int k = threadIdx.x; k < keys; k += K3_THREADS) {
        const u64 c = counts[k];
        u64* sub = req_out + (u64)k * (capb + 1) * 2;
        sub[0] = c < capb ? c : capb;
        sub[1] = (u64)any;
    }
    if (threadIdx.x == 0 && any) *overflow = 1;
    const double m = block_tile_table<K3_THREADS>(tile_m, tile_W, nt, S, s_incl, s_W, s_red, s_wtot);
    const u64 Q = s_incl[nt - 1];
    const u64 Q2 = block_sum_T2<K3_THREADS>(tile_m, tile_W2, nt, S, m, s_wtot);
    if (threadIdx.x == 0) {
        *undo = *scal;   // a fixed-capacity exchange that overflows puts these back
        fold_scalars(scal, Q, Q2, S, m, n_global, 0);
    }
}
// owner side: blockIdx.x & 7 = eighth of this shard's tiles (workgroups are dealt round-robin to the 8 XCDs, gridDim.x is
// a multiple of 8), blockIdx.y = asking rank.  Per request: guide cell -> first row -> short forward walk, all inside
// the eighth.  Rows go back in the order the requests came.
__global__ __launch_bounds__(K3_THREADS) void k_shard_resolve_binned(u64 n, u64 capb, u64 slot_offset, int D, const u64* __restrict__ req,
                                                                     const mp_cx* __restrict__ cx, const unsigned short* __restrict__ guide,
                                                                     const u64* __restrict__ tile_W, const double* __restrict__ x,
                                                                     double* __restrict__ rows, int* overflow) {
    const int bin = blockIdx.x & (SH_BINS - 1), grp = blockIdx.x >> 3, ngrp = gridDim.x >> 3;
    const u64 key = (u64)blockIdx.y * SH_BINS + bin;
    const ulonglong2* sub = reinterpret_cast<const ulonglong2*>(req) + key * (capb + 1);
    const ulonglong2 head = sub[0];
    const u64 cnt = head.x < capb ? head.x : capb;
    if (grp == 0 && threadIdx.x == 0 && head.y) *overflow = 1;
    double* out_sub = rows + key * capb * (u64)(D + 1);
    for (u64 q0 = (u64)grp * (K3_THREADS * K3_ITEMS); q0 < cnt; q0 += (u64)ngrp * (K3_THREADS * K3_ITEMS)) {
        u64 lt[K3_ITEMS], tbase[K3_ITEMS], last[K3_ITEMS];
        uint32_t gi[K3_ITEMS];
        bool live[K3_ITEMS];
#pragma unroll
        for (int k = 0; k < K3_ITEMS; ++k) {   // hop 0: the requests (coalesced)
            const u64 q = q0 + (u64)k * K3_THREADS + threadIdx.x;
            live[k] = q < cnt;
            const ulonglong2 e = live[k] ? sub[q + 1] : make_ulonglong2(0ull, 1ull);
            const u64 b = e.x;
            lt[k] = e.y;
            tbase[k] = b * TILE;
            const u64 tend = tbase[k] + TILE;
            last[k] = (tend < n ? tend : n) - 1;
            uint32_t g = (uint32_t)(lt[k] >> mp_guide_shift(tile_W[b]));
            if (g > GUIDE_N - 1) g = GUIDE_N - 1;
            gi[k] = (uint32_t)b * (uint32_t)GUIDE_N + g;
        }
        u64 p[K3_ITEMS];
#pragma unroll
        for (int k = 0; k < K3_ITEMS; ++k) {   // hop 1: guide cells
            u64 j = tbase[k] + guide[gi[k]];
            p[k] = j < last[k] ? j : last[k];
        }
        mp_cx r0[K3_ITEMS], r1[K3_ITEMS];
#pragma unroll
        for (int k = 0; k < K3_ITEMS; ++k) {   // hop 2: the row and its successor
            r0[k] = cx[p[k]];
            r1[k] = cx[p[k] + (p[k] < last[k] ? 1 : 0)];
        }
#pragma unroll
        for (int k = 0; k < K3_ITEMS; ++k) {
            if (!live[k]) continue;
            mp_cx cur = r0[k];
            u64 pp = p[k];
            if (cur.cum < lt[k] && pp < last[k]) {
                cur = r1[k];
                ++pp;
                while (cur.cum < lt[k] && pp < last[k]) {
                    ++pp;
                    cur = cx[pp];
                }
            }
            const u64 q = q0 + (u64)k * K3_THREADS + threadIdx.x;
            double* out = out_sub + q * (u64)(D + 1);
            if (D == 1) {
                *reinterpret_cast<double2*>(out) = make_double2(cur.x0, (double)(slot_offset + pp));
            } else {
                out[0] = cur.x0;
                for (int d = 1; d < D; ++d) out[d] = x[pp * D + d];
                out[D] = (double)(slot_offset + pp);
            }
        }
    }
}
// after the resolve: "somebody overflowed" and the scalars of this normalisation, where the host reads them after waiting
// for ev_resolved (host-mapped memory: no copy command, no stream sync)
__global__ void k_shard_publish(const int* overflow, const mp_dev_scalars* scal, mp_shard_pub* pub) {
    pub->L = scal->L;
    pub->degenerate = scal->degenerate;
    pub->overflow = *overflow;
}
// requester side, only when something other than the next propagate needs slot order: x[i], parent[i] from row inv[i]
__global__ __launch_bounds__(SH_THREADS) void k_shard_adopt_rows(u64 n, int D, const double* __restrict__ rows, const uint32_t* __restrict__ inv,
                                                                 double* __restrict__ x_new, uint32_t* __restrict__ parent) {
    const u64 i = (u64)blockIdx.x * SH_THREADS + threadIdx.x;
    if (i >= n) return;
    const double* in = rows + (u64)inv[i] * (u64)(D + 1);
    for (int d = 0; d < D; ++d) x_new[i * D + d] = in[d];
    parent[i] = (uint32_t)in[D];
}

// out[i] = a[i] - *b  (log_normalized_weights = w_i - log_total_weight, importance.rs:23-25)
__global__ void k_sub_scalar(const double* __restrict__ a, const double* __restrict__ b, u64 n, double* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] - *b;
}

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
    const unsigned short* perm;
    const double* res_x;
    u64 res_stride;
    int nchunks;
    mp_cx* cx;
    unsigned short* guide;
    double* tile_m;
    u64* tile_W;
    u64* tile_W2;
    const uint32_t* inv;
    const uint32_t* res_parent;
};
struct ModelOps {
    int dim_state = 0, dim_obs = 0;
    virtual ~ModelOps() {}
    virtual void propagate(const PropagateArgs& a) const = 0;
};
template <class Model>
struct ModelOpsT : ModelOps {
    Model model;
    explicit ModelOpsT(const Model& m) : model(m) {
        dim_state = Model::DIM_STATE;
        dim_obs = Model::DIM_OBS;
        static_assert(Model::DIM_STATE <= MP_MAX_STATE && Model::DIM_OBS <= MP_MAX_OBS, "model too wide for mp_obs / mp_state0");
        static_assert(TILE_ITEMS % k1_items<Model>() == 0, "rounds of k_propagate");
    }
    void propagate(const PropagateArgs& a) const override {
        // light kernels (few registers) run 1024 threads x 2 particles per tile: twice the waves in flight for the same 2048-slot tile
        constexpr int THREADS = (Model::MAX_NORMALS <= 2 && Model::DIM_STATE <= 2) ? 1024 : TILE_THREADS;
        hipLaunchKernelGGL((k_propagate<Model, THREADS>), dim3(a.grid), dim3(THREADS), 0, a.stream, model, a.n, a.slot_offset, a.k0, a.k1, a.t,
                           a.x_in, a.x_out, a.logw, a.obs, a.s0, a.overwrite, a.perm, a.res_x, a.res_stride, a.nchunks, a.cx, a.guide,
                           a.tile_m, a.tile_W, a.tile_W2, a.inv, a.res_parent);
    }
};

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
    default:
        return mp_fail(MP_ERR_UNSUPPORTED, "model kind " + std::to_string(m->kind) + " is not compiled into this library");
    }
}

struct TimedLaunch {
    hipEvent_t start, stop;
    int family;
};

struct mp_pf {
    std::unique_ptr<ModelOps> ops;
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
    // binned resampling scratch: segments [bin][chunk][1024]
    u64* seg_lt = nullptr;              // tile-local target of every binned draw
    uint32_t* seg_row = nullptr;        // table row where the forward scan of every binned draw starts
    unsigned short* perm = nullptr;     // [n]: (bin << 10 | position) of every slot's draw
    unsigned short* seg_cnt = nullptr;
    double* res_x = nullptr;            // [d][8 * nchunks * 1024]: resampled states in segment order
    uint32_t* res_parent = nullptr;     // [8 * nchunks * 1024]
    u64 res_stride = 0;
    int nchunks = 0;
    int use_binned = 1;                 // MP_BINNED_RESAMPLE=0 selects the single-kernel path (A/B measurements)
    bool permuted = false;              // the current states / parents / (zero) log-weights live in res_* (lazy slot order)
    bool rows_fresh = false;            // cx / guide / tile_* describe the current log-weights
    // sharded-resample scratch (allocated on first use)
    unsigned char* sh_dest = nullptr;
    u64* sh_lt = nullptr;
    uint32_t* sh_tile = nullptr;
    uint32_t* sh_req_slot = nullptr;
    uint32_t* sh_blockcount = nullptr;
    uint32_t* sh_blockoff = nullptr;
    long long* sh_counts = nullptr;
    long long* h_counts = nullptr;  // pinned
    int sh_world = 0;
    u64 sh_cap = 0;                 // fixed-capacity exchange: request slots per (src, dst) pair
    double* sh_tm_all = nullptr;    // unpacked gathered tiles
    u64* sh_tW_all = nullptr;
    u64* sh_tW2_all = nullptr;
    int* sh_overflow = nullptr;
    u64* tiles_own = nullptr;       // the allocation behind tile_m / tile_W / tile_W2 unless the caller bound its own buffer
    mp_shard_pub* h_pub = nullptr;  // pinned, host-mapped
    mp_shard_pub* d_pub = nullptr;  // its device address
    hipEvent_t ev_resolved = nullptr;
    bool sh_lazy = false;           // the states of the last sharded resample still sit in the exchange buffer sh_rows, slot i at row sh_req_slot[i]
    const double* sh_rows = nullptr;
    u64 sh_rows_cap = 0;
    bool logw_zero = false;         // log-weights are all zero (after a sharded resample) and the buffer has not been cleared
    mp_dev_scalars* scal_undo = nullptr;  // the scalars before a fixed-capacity route folded this resample in
    bool sharded = false;
    // ancestry record (MP_PF_RECORD_HISTORY): the event log from which `traces[i].retv` is rebuilt
    struct HistEvent { int kind; void* buf; };  // kind 0: states after an Unfold step ([d][n] f64); 1: parents of a resample ([n] u32)
    std::vector<HistEvent> hist;
    // host-side filter state
    long long t = 0;  // Unfold steps taken (trace.args.0)
    uint32_t resample_count = 0;
    bool initialised = false;
    // timing
    bool timing = false;
    std::vector<TimedLaunch> timed;
    std::vector<hipEvent_t> event_pool;
    double fam_ms[MP_K_COUNT] = {0, 0, 0, 0};
    uint64_t fam_launches[MP_K_COUNT] = {0, 0, 0, 0};
};

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

static int32_t fetch_scalars(mp_pf* h) {
    HIPCK(hipMemcpyAsync(h->h_scal, h->scal, sizeof(mp_dev_scalars), hipMemcpyDeviceToHost, h->stream));
    HIPCK(hipStreamSynchronize(h->stream));
    if (h->h_scal->degenerate)
        return mp_fail(MP_ERR_DEGENERATE, "all log-weights are -inf: normalized weights are NaN (categorical.rs:23 assert in the reference)");
    return MP_OK;
}

// Slot-order x / parent / logw after a binned resample (only when something other than the next step needs them).
static int32_t materialize(mp_pf* h) {
    if (h->sh_lazy) {
        hipLaunchKernelGGL(k_shard_adopt_rows, dim3((unsigned)((h->n + SH_THREADS - 1) / SH_THREADS)), dim3(SH_THREADS), 0, h->stream, h->n,
                           h->ops->dim_state, h->sh_rows, h->sh_req_slot, h->x[h->cur], h->parent);
        h->sh_lazy = false;
        int32_t rc = check_launch("k_shard_adopt_rows");
        if (rc != MP_OK) return rc;
    }
    if (h->logw_zero) {
        HIPCK(hipMemsetAsync(h->logw, 0, sizeof(double) * h->n, h->stream));
        h->logw_zero = false;
    }
    if (!h->permuted) return MP_OK;
    hipLaunchKernelGGL(k_unpermute, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, h->stream, h->n, h->ops->dim_state, h->nchunks, h->perm,
                       h->res_x, h->res_stride, h->res_parent, h->x[h->cur], h->ops->dim_state == 1 ? h->x[h->cur] : h->x[h->cur ^ 1], h->parent,
                       h->logw);
    if (h->ops->dim_state > 1) h->cur ^= 1;   // wider states were gathered from the pre-resample buffer into the other one
    h->permuted = false;
    return check_launch("k_unpermute");
}

static int32_t launch_propagate(mp_pf* h, const double* args0, const double* obs, bool overwrite) {
    PropagateArgs a;
    a.n = h->n; a.slot_offset = h->slot_offset;
    a.k0 = (uint32_t)h->seed; a.k1 = (uint32_t)(h->seed >> 32);
    a.t = h->t;
    // after a binned resample of a filter with dim_state > 1 this propagate gathers the parents' states itself, from the
    // pre-resample buffer into the other one
    const bool gather_here = h->permuted && !h->sh_lazy && h->ops->dim_state > 1;
    a.x_in = h->x[h->cur]; a.x_out = gather_here ? h->x[h->cur ^ 1] : h->x[h->cur];
    a.res_parent = h->res_parent;
    a.logw = h->logw;
    for (int j = 0; j < MP_MAX_OBS; ++j) a.obs.v[j] = (j < h->ops->dim_obs) ? obs[j] : 0.;
    for (int j = 0; j < MP_MAX_STATE; ++j) a.s0.v[j] = (args0 && j < h->ops->dim_state) ? args0[j] : 0.;
    a.overwrite = overwrite ? 1 : ((h->permuted || h->logw_zero) ? 2 : 0);
    a.perm = h->permuted ? h->perm : nullptr;
    a.res_x = h->sh_lazy ? h->sh_rows : h->res_x;
    a.inv = h->sh_lazy ? h->sh_req_slot : nullptr;
    a.res_stride = h->res_stride;
    a.nchunks = h->nchunks;
    a.cx = h->cx; a.guide = h->guide; a.tile_m = h->tile_m; a.tile_W = h->tile_W; a.tile_W2 = h->tile_W2;
    a.grid = h->nt;
    a.stream = h->stream;
    {
        LaunchTimer lt(h, MP_K_PROPAGATE);
        h->ops->propagate(a);
    }
    h->t += 1;
    if (gather_here) h->cur ^= 1;
    h->logw_zero = false;
    h->sh_lazy = false;
    h->permuted = false;   // k_propagate wrote x[cur] and logw in slot order ...
    h->rows_fresh = true;  // ... and level 0 of their normalisation
    int32_t rc_ = check_launch("k_propagate");
    if (rc_ != MP_OK) return rc_;
    if (h->flags & MP_PF_RECORD_HISTORY) {
        double* buf = nullptr;
        const size_t bytes = sizeof(double) * h->n * (size_t)h->ops->dim_state;
        HIPCK(hipMalloc(&buf, bytes));
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
    {
        LaunchTimer lt(h, MP_K_NORMALIZE_SCAN);
        hipLaunchKernelGGL(k_normalize_tiles, dim3(h->nt), dim3(TILE_THREADS), 0, h->stream, h->logw, h->x[h->cur], h->ops->dim_state, h->n, h->cx, h->guide,
                           h->tile_m, h->tile_W, h->tile_W2);
    }
    h->rows_fresh = true;
    return check_launch("k_normalize_tiles");
}

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
    std::unique_ptr<mp_pf> h(new mp_pf());
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
    h->k3_grid = (int)((n + K3_THREADS * K3_ITEMS - 1) / (K3_THREADS * K3_ITEMS));
    if (h->k3_grid > K3_MAX_BLOCKS) h->k3_grid = K3_MAX_BLOCKS;
    h->nchunks = (int)((n + BIN_CHUNK - 1) / BIN_CHUNK);
    {
        const char* env = getenv("MP_BINNED_RESAMPLE");
        if (env && env[0] == '0') h->use_binned = 0;
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
    HIPCK(hipMalloc(&h->aos, sizeof(double) * n));   // scratch for importance sampling's normalised log-weights
    HIPCK(hipHostMalloc(&h->h_scal, sizeof(mp_dev_scalars)));
    h->res_stride = 8ull * (u64)h->nchunks * BIN_CHUNK;
    HIPCK(hipMalloc(&h->seg_lt, sizeof(u64) * h->res_stride));
    HIPCK(hipMalloc(&h->seg_row, sizeof(uint32_t) * h->res_stride));
    HIPCK(hipMalloc(&h->perm, sizeof(unsigned short) * (size_t)h->nchunks * BIN_CHUNK));
    HIPCK(hipMalloc(&h->seg_cnt, sizeof(unsigned short) * 8 * (size_t)h->nchunks));
    HIPCK(hipMalloc(&h->res_x, sizeof(double) * h->res_stride));   // first state component only (dim_state == 1 needs nothing else)
    HIPCK(hipMalloc(&h->res_parent, sizeof(uint32_t) * h->res_stride));
    // tile tables above 64 KiB of LDS need the limit raised once per kernel
    {
        const int nt_job = (int)((h->n_global + TILE - 1) / TILE);
        const size_t need = table_lds(nt_job, K3_THREADS) + 1024;
        if (need > 48 * 1024) {
            HIPCK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_resample_gather<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
            HIPCK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_resample_gather<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
            HIPCK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bin_draws), hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
            HIPCK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_finalize_tiles), hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
            HIPCK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_shard_targets), hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
        }
    }
    // ParticleSystem::new: log_weights = 0, parents = 0, log_ml_estimate = 0 (particle_filter.rs:44-57)
    HIPCK(hipMemsetAsync(h->x[0], 0, sizeof(double) * n * d, h->stream));
    HIPCK(hipMemsetAsync(h->x[1], 0, sizeof(double) * n * d, h->stream));
    HIPCK(hipMemsetAsync(h->logw, 0, sizeof(double) * n, h->stream));
    HIPCK(hipMemsetAsync(h->parent, 0, sizeof(uint32_t) * n, h->stream));
    mp_dev_scalars init{};
    init.ess_stale = 1.0 / (double)h->n_global;  // exp(-logsumexp(zeros)) before any resample
    *h->h_scal = init;
    HIPCK(hipMemcpyAsync(h->scal, h->h_scal, sizeof(mp_dev_scalars), hipMemcpyHostToDevice, h->stream));
    HIPCK(hipStreamSynchronize(h->stream));
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

int32_t mp_pf_resample(mp_pf* h, int32_t scheme, double* log_total_weight) {
    if (!h) return mp_fail(MP_ERR_INVALID_ARG, "null handle");
    if (!h->initialised) return mp_fail(MP_ERR_STATE, "resample before init_step");
    if (scheme != MP_RESAMPLE_MULTINOMIAL && scheme != MP_RESAMPLE_SYSTEMATIC && scheme != MP_RESAMPLE_STRATIFIED)
        return mp_fail(MP_ERR_INVALID_ARG, "unknown resampling scheme");
    if (h->sharded) return mp_fail(MP_ERR_STATE, "sharded handle: resample runs through the mp_pf_shard_* phases");
    HIPCK(hipSetDevice(h->device));
    int32_t rc = ensure_rows(h);
    if (rc != MP_OK) return rc;
    const int d = h->ops->dim_state;
    bool binned = false;
    if (scheme == MP_RESAMPLE_MULTINOMIAL && h->use_binned) {
        LaunchTimer lt(h, MP_K_BIN_DRAWS);
        const size_t lds_a = table_lds(h->nt, BIN_THREADS) + sizeof(uint32_t) * 2 * BIN_ITEMS * (BIN_THREADS / 64) * 8;
        hipLaunchKernelGGL(k_bin_draws, dim3(h->nchunks), dim3(BIN_THREADS), lds_a, h->stream, h->n, h->n_global, h->slot_offset, (uint32_t)h->seed,
                           (uint32_t)(h->seed >> 32), h->resample_count, h->S, h->nchunks, h->tile_m, h->tile_W, h->tile_W2, h->nt, h->guide,
                           h->seg_lt, h->seg_row, h->perm, h->seg_cnt, h->scal);
    }
    {
        LaunchTimer lt(h, MP_K_RESAMPLE_GATHER);
        if (scheme == MP_RESAMPLE_MULTINOMIAL && h->use_binned) {
            const int ngroups = (h->nchunks + BIN_GROUP - 1) / BIN_GROUP;
            hipLaunchKernelGGL(k_resolve_bins, dim3(ngroups * 8), dim3(K3_THREADS), 0, h->stream, h->n, d, h->nchunks, h->seg_lt, h->seg_row,
                               h->seg_cnt, h->cx, h->x[h->cur], h->res_x, h->res_stride, h->res_parent);
            binned = true;
        } else if (scheme == MP_RESAMPLE_STRATIFIED) {
            hipLaunchKernelGGL(k_resample_gather<2>, dim3(h->k3_grid), dim3(K3_THREADS), table_lds(h->nt, K3_THREADS), h->stream, h->n, h->n,
                               h->n_global, h->slot_offset, (uint32_t)MP_DOM_RESAMPLE, (uint32_t)h->seed, (uint32_t)(h->seed >> 32), h->resample_count,
                               h->S, d, h->cx, h->guide, h->tile_m, h->tile_W, h->tile_W2, h->nt, h->x[h->cur], h->x[h->cur ^ 1], h->parent, h->logw,
                               h->scal);
        } else if (scheme == MP_RESAMPLE_SYSTEMATIC) {
            hipLaunchKernelGGL(k_resample_gather<1>, dim3(h->k3_grid), dim3(K3_THREADS), table_lds(h->nt, K3_THREADS), h->stream, h->n, h->n,
                               h->n_global, h->slot_offset, (uint32_t)MP_DOM_RESAMPLE, (uint32_t)h->seed, (uint32_t)(h->seed >> 32), h->resample_count,
                               h->S, d, h->cx, h->guide, h->tile_m, h->tile_W, h->tile_W2, h->nt, h->x[h->cur], h->x[h->cur ^ 1], h->parent, h->logw,
                               h->scal);
        } else {
            hipLaunchKernelGGL(k_resample_gather<0>, dim3(h->k3_grid), dim3(K3_THREADS), table_lds(h->nt, K3_THREADS), h->stream, h->n, h->n,
                               h->n_global, h->slot_offset, (uint32_t)MP_DOM_RESAMPLE, (uint32_t)h->seed, (uint32_t)(h->seed >> 32), h->resample_count,
                               h->S, d, h->cx, h->guide, h->tile_m, h->tile_W, h->tile_W2, h->nt, h->x[h->cur], h->x[h->cur ^ 1], h->parent, h->logw,
                               h->scal);
        }
    }
    rc = check_launch("resample kernels");
    if (rc != MP_OK) return rc;
    if (binned) h->permuted = true;   // results stay in segment order; x[cur] is the (stale) pre-resample state
    else h->cur ^= 1;
    h->rows_fresh = false;            // the log-weights are now all zero
    h->resample_count += 1;
    if (h->flags & MP_PF_RECORD_HISTORY) {
        rc = materialize(h);
        if (rc != MP_OK) return rc;
        uint32_t* buf = nullptr;
        HIPCK(hipMalloc(&buf, sizeof(uint32_t) * h->n));
        HIPCK(hipMemcpyAsync(buf, h->parent, sizeof(uint32_t) * h->n, hipMemcpyDeviceToDevice, h->stream));
        h->hist.push_back({1, buf});
    }
    if (log_total_weight) {
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
    const int d = h->ops->dim_state;   // device layout = host layout: particle-major x[i][d]
    HIPCK(hipMemcpyAsync(x_out, h->x[h->cur], sizeof(double) * h->n * d, hipMemcpyDeviceToHost, h->stream));
    HIPCK(hipStreamSynchronize(h->stream));
    return MP_OK;
}

int32_t mp_pf_read_log_weights(mp_pf* h, double* out) {
    if (!h || !out) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    HIPCK(hipSetDevice(h->device));
    { int32_t rcm = materialize(h); if (rcm != MP_OK) return rcm; }
    HIPCK(hipMemcpyAsync(out, h->logw, sizeof(double) * h->n, hipMemcpyDeviceToHost, h->stream));
    HIPCK(hipStreamSynchronize(h->stream));
    return MP_OK;
}

int32_t mp_pf_read_parents(mp_pf* h, uint32_t* out) {
    if (!h || !out) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    HIPCK(hipSetDevice(h->device));
    { int32_t rcm = materialize(h); if (rcm != MP_OK) return rcm; }
    HIPCK(hipMemcpyAsync(out, h->parent, sizeof(uint32_t) * h->n, hipMemcpyDeviceToHost, h->stream));
    HIPCK(hipStreamSynchronize(h->stream));
    return MP_OK;
}

// ---- sharded phases ------------------------------------------------------------------------------
static int32_t shard_scratch(mp_pf* h, int world, u64 cap);

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
    if (n_req == 0) return MP_OK;
    if (!d_req_in || !d_rows_out) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    HIPCK(hipSetDevice(h->device));
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
    h->cur ^= 1;
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
    (void)hipFree(h->sh_blockoff); (void)hipFree(h->sh_counts); (void)hipFree(h->sh_tm_all); (void)hipFree(h->sh_tW_all); (void)hipFree(h->sh_tW2_all);
    (void)hipFree(h->sh_overflow);
    if (h->h_counts) (void)hipHostFree(h->h_counts);
    const u64 slots = std::max<u64>(h->n, (u64)world * SH_BINS * cap);
    HIPCK(hipMalloc(&h->sh_dest, h->n));
    HIPCK(hipMalloc(&h->sh_lt, sizeof(u64) * h->n));
    HIPCK(hipMalloc(&h->sh_tile, sizeof(uint32_t) * h->n));
    HIPCK(hipMalloc(&h->sh_req_slot, sizeof(uint32_t) * slots));
    HIPCK(hipMalloc(&h->sh_blockcount, sizeof(uint32_t) * (size_t)nblk * world));
    HIPCK(hipMalloc(&h->sh_blockoff, sizeof(uint32_t) * (size_t)nblk * world));
    HIPCK(hipMalloc(&h->sh_counts, sizeof(long long) * SH_MAX_KEYS));
    HIPCK(hipMalloc(&h->sh_tm_all, sizeof(double) * (size_t)h->nt * world));
    HIPCK(hipMalloc(&h->sh_tW_all, sizeof(u64) * (size_t)h->nt * world));
    HIPCK(hipMalloc(&h->sh_tW2_all, sizeof(u64) * (size_t)h->nt * world));
    HIPCK(hipMalloc(&h->sh_overflow, sizeof(int)));
    if (!h->scal_undo) HIPCK(hipMalloc(&h->scal_undo, sizeof(mp_dev_scalars)));
    HIPCK(hipMemsetAsync(h->sh_overflow, 0, sizeof(int), h->stream));
    HIPCK(hipHostMalloc(&h->h_counts, sizeof(long long) * SH_MAX_WORLD));
    if (!h->h_pub) {
        HIPCK(hipHostMalloc(&h->h_pub, sizeof(mp_shard_pub), hipHostMallocMapped));
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
    h->tile_m = reinterpret_cast<double*>(d_tiles);
    h->tile_W = (u64*)d_tiles + h->nt;
    h->tile_W2 = (u64*)d_tiles + 2 * (size_t)h->nt;
    return MP_OK;
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

int32_t mp_pf_shard_route_fixed(mp_pf* h, int32_t scheme, const uint64_t* d_tiles_all, int32_t world, int32_t rank, uint64_t capacity,
                                uint64_t* d_req_out) {
    if (!h || !d_tiles_all || !d_req_out) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    if (scheme != MP_RESAMPLE_MULTINOMIAL && scheme != MP_RESAMPLE_SYSTEMATIC && scheme != MP_RESAMPLE_STRATIFIED)
        return mp_fail(MP_ERR_INVALID_ARG, "unknown resampling scheme");
    if (world < 1 || world > SH_MAX_WORLD || rank < 0 || rank >= world) return mp_fail(MP_ERR_INVALID_ARG, "1 <= world <= 64, 0 <= rank < world");
    if ((u64)world * h->n != h->n_global) return mp_fail(MP_ERR_INVALID_ARG, "equal tile-aligned shards: world * n_particles must equal n_global");
    if (capacity == 0) return mp_fail(MP_ERR_INVALID_ARG, "capacity must be > 0");
    HIPCK(hipSetDevice(h->device));
    int32_t rc = shard_scratch(h, world, capacity);
    if (rc != MP_OK) return rc;
    const int nt_all = h->nt * world;
    {
        LaunchTimer lt(h, MP_K_BIN_DRAWS);
        const int cover = std::max(nt_all, SH_MAX_KEYS);
        hipLaunchKernelGGL(k_unpack_tiles, dim3((cover + K3_THREADS - 1) / K3_THREADS), dim3(K3_THREADS), 0, h->stream, (const u64*)d_tiles_all, world,
                           h->nt, h->sh_tm_all, h->sh_tW_all, h->sh_tW2_all, h->sh_counts);
        const size_t lds = table_lds(nt_all, SH_THREADS) + (sizeof(uint32_t) + sizeof(u64)) * SH_MAX_KEYS;
        const int nblk_f = (int)((h->n + SH_THREADS * SHF_ITEMS - 1) / (SH_THREADS * SHF_ITEMS));
        hipLaunchKernelGGL(k_shard_route_fused, dim3(nblk_f), dim3(SH_THREADS), lds, h->stream, h->n, h->n_global, h->slot_offset, (uint32_t)h->seed,
                           (uint32_t)(h->seed >> 32), h->resample_count, (int)scheme, h->S, h->sh_tm_all, h->sh_tW_all,
                           nt_all, h->nt, world, (u64)capacity, (unsigned long long*)h->sh_counts, (u64*)d_req_out, h->sh_req_slot);
        hipLaunchKernelGGL(k_shard_finalize, dim3(1), dim3(K3_THREADS), table_lds(nt_all, K3_THREADS), h->stream, h->sh_tm_all, h->sh_tW_all,
                           h->sh_tW2_all, nt_all, h->S, h->n_global, h->scal, h->scal_undo, (const unsigned long long*)h->sh_counts, world,
                           (u64)capacity, (u64*)d_req_out, h->sh_overflow);
    }
    return check_launch("shard_route_fixed kernels");
}

int32_t mp_pf_shard_resolve_fixed(mp_pf* h, const uint64_t* d_req_in, int32_t world, uint64_t capacity, double* d_rows_out) {
    if (!h || !d_req_in || !d_rows_out) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    if (world < 1 || world > SH_MAX_WORLD || capacity == 0) return mp_fail(MP_ERR_INVALID_ARG, "1 <= world <= 64, capacity > 0");
    if (!h->h_pub) return mp_fail(MP_ERR_STATE, "shard_resolve_fixed before shard_route_fixed");
    HIPCK(hipSetDevice(h->device));
    // workgroups per (asking rank, eighth): enough to fill the chip, each takes K3_THREADS * K3_ITEMS requests per round
    const u64 per = (u64)K3_THREADS * K3_ITEMS;
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
    // traces[i] = traces[parents[i]].clone() (particle_filter.rs:109-113), lazily: the next propagate reads slot i's state
    // from row sh_req_slot[i] of the exchange buffer; anything else first copies states and parents into slot order.
    h->sh_rows = d_rows_in;
    h->sh_lazy = true;
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

int32_t mp_pf_read_trajectory(mp_pf* h, uint64_t i, double* out, int32_t* t_steps) {
    if (!h || !out || !t_steps) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    if (!(h->flags & MP_PF_RECORD_HISTORY)) return mp_fail(MP_ERR_STATE, "read_trajectory needs a filter created with MP_PF_RECORD_HISTORY");
    if (h->sharded) return mp_fail(MP_ERR_UNSUPPORTED, "read_trajectory: ancestors of a sharded filter live on other ranks");
    if (i >= h->n) return mp_fail(MP_ERR_INVALID_ARG, "particle index out of range");
    HIPCK(hipSetDevice(h->device));
    HIPCK(hipStreamSynchronize(h->stream));
    // walk the event log backwards: a resample maps slot -> parent slot (traces[i] = traces[parents[i]].clone(),
    // particle_filter.rs:109-113); a step contributes the state of the current ancestor slot (retv.push, dynunfold.rs:58,92)
    const int d = h->ops->dim_state;
    int t = (int)h->t;
    uint64_t a = i;
    for (size_t e = h->hist.size(); e-- > 0;) {
        const auto& ev = h->hist[e];
        if (ev.kind == 1) {
            uint32_t p = 0;
            HIPCK(hipMemcpy(&p, (const uint32_t*)ev.buf + a, sizeof(uint32_t), hipMemcpyDeviceToHost));
            a = p;
        } else {
            --t;
            HIPCK(hipMemcpy(out + (size_t)t * d, (const double*)ev.buf + (size_t)a * d, sizeof(double) * d, hipMemcpyDeviceToHost));
        }
    }
    *t_steps = (int32_t)h->t;
    return MP_OK;
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
    for (int t = 1; t < n_steps; ++t) {
        rc = mp_pf_step(h, obs + (size_t)t * h->ops->dim_obs, 1);
        if (rc != MP_OK) return rc;
        rc = mp_pf_resample(h, scheme, nullptr);
        if (rc != MP_OK) return rc;
    }
    return MP_OK;
}

int32_t mp_pf_synchronize(mp_pf* h) {
    if (!h) return mp_fail(MP_ERR_INVALID_ARG, "null handle");
    HIPCK(hipSetDevice(h->device));
    return fetch_scalars(h);
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

int32_t mp_pf_destroy(mp_pf* h) {
    if (!h) return MP_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    for (auto& tl : h->timed) {
        (void)hipEventDestroy(tl.start);
        (void)hipEventDestroy(tl.stop);
    }
    for (auto e : h->event_pool) (void)hipEventDestroy(e);
    for (auto& ev : h->hist) (void)hipFree(ev.buf);
    (void)hipFree(h->x[0]); (void)hipFree(h->x[1]); (void)hipFree(h->logw); (void)hipFree(h->cx); (void)hipFree(h->guide);
    (void)hipFree(h->parent); (void)hipFree(h->tiles_own); (void)hipFree(h->scal);
    (void)hipFree(h->aos);
    (void)hipFree(h->seg_lt); (void)hipFree(h->seg_row); (void)hipFree(h->perm); (void)hipFree(h->seg_cnt); (void)hipFree(h->res_x); (void)hipFree(h->res_parent);
    (void)hipFree(h->sh_dest); (void)hipFree(h->sh_lt); (void)hipFree(h->sh_tile); (void)hipFree(h->sh_req_slot); (void)hipFree(h->sh_blockcount);
    (void)hipFree(h->sh_blockoff); (void)hipFree(h->sh_counts); (void)hipFree(h->sh_tm_all); (void)hipFree(h->sh_tW_all); (void)hipFree(h->sh_tW2_all);
    (void)hipFree(h->sh_overflow); (void)hipFree(h->scal_undo);
    if (h->h_counts) (void)hipHostFree(h->h_counts);
    if (h->h_pub) (void)hipHostFree(h->h_pub);
    if (h->ev_resolved) (void)hipEventDestroy(h->ev_resolved);
    (void)hipHostFree(h->h_scal);
    if (h->own_stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return MP_OK;
}

int32_t mp_importance_resampling(const mp_model_desc* model, const double* args0, const double* obs, int32_t n_steps, uint64_t num_samples,
                                 uint64_t num_ret_samples, uint64_t seed, int32_t device, double* log_ml_estimate,
                                 double* log_normalized_weights, uint64_t* resampled_indices, double* final_states) {
    if (!model || !obs) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    if (num_ret_samples > 0xFFFFFFFFull) return mp_fail(MP_ERR_INVALID_ARG, "num_ret_samples must fit u32");
    mp_pf* h = nullptr;
    int32_t rc = mp_pf_create(model, num_samples, seed, nullptr, 0, device, nullptr, &h);
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
    if (resampled_indices && num_ret_samples > 0) {
        // importance_resampling: M categorical draws over exp(lnw) (importance.rs:44-47), slot j of DOM_IS
        uint32_t* d_idx = nullptr;
        HIPCK(hipMalloc(&d_idx, sizeof(uint32_t) * num_ret_samples));
        const int grid = (int)std::min<u64>((num_ret_samples + K3_THREADS * K3_ITEMS - 1) / (K3_THREADS * K3_ITEMS), (u64)K3_MAX_BLOCKS);
        hipLaunchKernelGGL(k_resample_gather<0>, dim3(grid), dim3(K3_THREADS), table_lds(h->nt, K3_THREADS), h->stream, h->n, (u64)num_ret_samples,
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

}  // extern "C"
