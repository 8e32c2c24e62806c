// mp_pf.hip — gfx950 kernels and C-ABI implementation of the particle-filter hot path
// (include/modppl_hip.h).  Written for MI355X only: 64-lane wavefronts, LDS-staged reductions,
// SoA particle state in HBM.
//
// Data layout in HBM (per handle, n = local particles, d = dim_state):
//   x[2][d][n]  f64   particle states, double-buffered (resample gathers from one into the other)
//   logw[n]     f64   log-weights                              (particle_filter.rs:15)
//   cx[n]       {u64,f64}  resampling table rows: tile-local inclusive prefix sum of the fixed-point weights + x[0]
//   guide[nt][2048] u16  per-tile bucketed inverse CDF (first row of each bucket)
//   parent[n]   u32   parents of the last resample             (particle_filter.rs:20)
//   blockmax[]  f64   per-workgroup maxima of logw (written by every kernel that changes logw)
//   tilesum[], tilesum2[] u64  per-tile totals of q and q^2-weights
//   scal        mp_dev_scalars   log_ml_estimate, last log total weight, ESS ... (device-resident so
//                                that a whole filter run needs no host round trip)
//
// Kernels (one Unfold step + resample = K1, K2, K3; no inter-workgroup communication inside a
// launch, so nothing depends on dispatch order or XCD placement):
//   K1 k_propagate        ParticleSystem::init_step/step  : per particle run the model kernel in
//                         Generate mode, logw (+)= weight, per-workgroup max          [16d+16 B/particle]
//   K2 k_normalize_scan   normalize_weights (:27-35) as an order-free fixed-point CDF: m = max,
//                         q = rint(exp(lw-m) * 2^S), tile-local inclusive scan, tile totals [8+8 B]
//   K3 k_resample_gather  multinomial_resampling + the clone loop of resample (:37-41,:109-114):
//                         LDS scan of tile totals, Philox draw, two-level binary search, gather
//                         x[parent], logw = 0; workgroup 0 also folds L into log_ml  [8+4+4+16d+8 B]
// Fixed point: S = 62 - ceil(log2 N_global); integer sums are associative, so any reduction order
// gives the same bits (DESIGN.md §4 states the spec; oracle/src/inference.hpp restates it on the CPU).
#include <hip/hip_runtime.h>

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
    double m;          // max log-weight seen by the last K2
    double L;          // log total weight of the last resample (resample()'s return value)
    double log_ml;     // log_ml_estimate (particle_filter.rs:24)
    double ess_stale;  // ESS of the weights normalised by the last resample (:98-100 semantics)
    double ess_fresh;  // outputs of the query path (k_lse_finalize)
    double lml_fresh;
    u64 Q, Q2;
    int degenerate;    // sticky: all log-weights were -inf (or +inf) at a normalisation
    int pad;
};

constexpr int K1_THREADS = 256;
// particles per lane in k_propagate: 4 pre-drawn (u, r) pairs per lane whatever the model's number of normal sites
template <class Model>
constexpr int k1_items() { return Model::MAX_NORMALS >= 4 ? 1 : 4 / Model::MAX_NORMALS; }
constexpr int K1_MAX_BLOCKS = 2048;
constexpr int SCAN_THREADS = 512;
constexpr int SCAN_ITEMS = 4;
constexpr int TILE = SCAN_THREADS * SCAN_ITEMS;  // 2048 particles per scan tile
constexpr int BIN_CHUNK = 1024;       // output slots per chunk of the binned resampler
constexpr int BIN_THREADS = 256;
constexpr int BIN_ITEMS = BIN_CHUNK / BIN_THREADS;
constexpr int BIN_GROUP = 8;          // chunks per k_resolve_bins workgroup
// Position of entry e of segment (bin, chunk) in the sparse segment arrays [bin][chunk][1024].  Only ~128 entries of
// each 1024-entry window are used; rotating the start by (chunk % 8) * 128 spreads the used parts over all memory
// channels instead of the ones the first eighth of every 8 KB window maps to.
#define MP_SEG_POS(bin, c, e, nchunks) ((((u64)(bin) * (u64)(nchunks) + (u64)(c)) * BIN_CHUNK) + (u64)((((uint32_t)(e)) + (((uint32_t)(c)) & 7u) * 128u) & 1023u))
constexpr int K3_THREADS = 256;
constexpr int K3_ITEMS = 4;
constexpr int K3_MAX_BLOCKS = 4096;
constexpr int MAX_TILES = 8192;                  // LDS prefix of tile totals: 64 KiB

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

// ---------------------------------------------------------------------------------------------
// K1: propagate + weight + per-workgroup max
// ---------------------------------------------------------------------------------------------
// Each lane owns K1_ITEMS particles (slots base + p*K1_THREADS + tid: coalesced per p).
// Phase 1 runs every polar rejection loop of the lane as ONE lane-local work queue over its
// (particle, normal site) items: a wave iterates max-over-lanes of the SUM of attempts instead of
// the sum over items of the max, i.e. ~1.9 Philox blocks per item at 4 items instead of ~3.6
// (acceptance pi/4, 64 lanes).  Phase 2 runs the model kernel per particle on the accepted pairs.
template <class Model>
__global__ __launch_bounds__(K1_THREADS) void k_propagate(Model model, u64 n, u64 slot_offset, uint32_t k0, uint32_t k1,
                                                          long long t, const double* x_in, double* x_out, double* logw,
                                                          mp_obs obs, mp_state0 s0, int overwrite, double* __restrict__ blockmax,
                                                          const unsigned short* __restrict__ perm, const double* __restrict__ res_x,
                                                          u64 res_stride, int nchunks) {
    constexpr int D = Model::DIM_STATE;
    constexpr int NS = Model::MAX_NORMALS;
    constexpr int K1_ITEMS = k1_items<Model>();
    constexpr int M = K1_ITEMS * NS;
    double lmax = MP_NEG_INF;
    const int ns = model.n_normals(t);  // wave-uniform
    for (u64 i0 = (u64)blockIdx.x * (K1_THREADS * K1_ITEMS) + threadIdx.x; i0 < n; i0 += (u64)gridDim.x * (K1_THREADS * K1_ITEMS)) {
        // ---- phase 1: accepted (u, r) pairs for every (particle, normal site) of this lane ----
        double pu[M], pr[M];
#pragma unroll
        for (int q = 0; q < M; ++q) { pu[q] = 0.; pr[q] = 1.; }
        {
            int p = 0, sidx = 0;         // current item: particle p, normal site index sidx
            uint32_t att = 0;
            // skip particles past the end
            while (p < K1_ITEMS && ns > 0) {
                const u64 i = i0 + (u64)p * K1_THREADS;
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
        for (int p = 0; p < K1_ITEMS; ++p) {
            const u64 i = i0 + (u64)p * K1_THREADS;
            if (i < n) {
                double prev[D], next[D];
                if (perm) {
                    // the last resample left the states in bin-segment order (k_resolve_bins): slot i's state sits at
                    // segment (bin, chunk of i) position rank, (bin << 10 | rank) = perm[i]
                    const uint32_t pr = perm[i];
                    const u64 pos = MP_SEG_POS(pr >> 10, i >> 10, pr & 1023u, nchunks);
#pragma unroll
                    for (int d = 0; d < D; ++d) prev[d] = res_x[(u64)d * res_stride + pos];
                } else {
#pragma unroll
                    for (int d = 0; d < D; ++d) prev[d] = (t == 0) ? s0.v[d] : x_in[(u64)d * n + i];
                }
                mp_stream rng;
                rng.k0 = k0; rng.k1 = k1; rng.slot = (uint32_t)(slot_offset + i); rng.step = (uint32_t)t;
                mp_generate_handler<Model> g(rng, obs.v, &pu[p * NS], &pr[p * NS]);
                model(g, t, prev, next);
#pragma unroll
                for (int d = 0; d < D; ++d) x_out[(u64)d * n + i] = next[d];
                // particle_filter.rs:68 (init: overwrite) / :81 (accumulate); overwrite == 2: the log-weights are known to
                // be all zero after a resample (log_weights.fill(0.), :114) and are not re-read
                const double w = overwrite == 1 ? g.weight : (overwrite == 2 ? 0. + g.weight : logw[i] + g.weight);
                logw[i] = w;
                lmax = fmax(lmax, w);
            }
        }
    }
    __shared__ double s_max[K1_THREADS / 64];
    lmax = wave_max(lmax);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_max[wave] = lmax;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = s_max[0];
#pragma unroll
        for (int w = 1; w < K1_THREADS / 64; ++w) m = fmax(m, s_max[w]);
        blockmax[blockIdx.x] = m;
    }
}

// ---------------------------------------------------------------------------------------------
// K2: fixed-point normalisation + tile-local inclusive scan
// ---------------------------------------------------------------------------------------------
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
constexpr int GUIDE_BITS = 11;              // one guide bucket per table row (GUIDE_N == TILE): 2 B per particle
constexpr int GUIDE_N = 1 << GUIDE_BITS;
static_assert(GUIDE_N == TILE, "k_normalize_scan zeroes/stores the guide with one 8-byte word per thread");
constexpr int GUIDE_DIRECT = 8;             // bucket runs longer than this are filled by the whole wave

// Guide table (bucketed inverse CDF, per tile): bucket g covers tile-local targets t with
// (t >> shift) == g, shift = max(0, bitlen(W) - 11) for the tile total W; guide[g] = first local
// index j with cum_j >= max(1, g << shift).  A draw then starts its scan at guide[t >> shift]
// and walks forward (expected < 2 rows).  Integer shifts only: no rounding anywhere.
__device__ __forceinline__ int mp_guide_shift(u64 W) {
    const int bits = 64 - __clzll((long long)W);  // W == 0 -> clz = 64 -> bits = 0
    return bits > GUIDE_BITS ? bits - GUIDE_BITS : 0;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_normalize_scan(const double* __restrict__ logw, const double* __restrict__ x0, u64 n,
                                                                 const double* __restrict__ blockmax, int nb, int S,
                                                                 mp_cx* __restrict__ cx, unsigned short* __restrict__ guide,
                                                                 u64* __restrict__ tilesum, u64* __restrict__ tilesum2,
                                                                 mp_dev_scalars* scal) {
    __shared__ double s_red[SCAN_THREADS / 64];
    __shared__ u64 s_wsum[SCAN_THREADS / 64];
    __shared__ u64 s_wsum2[SCAN_THREADS / 64];
    __shared__ __attribute__((aligned(16))) unsigned short s_guide[GUIDE_N];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // issue the tile's loads first: they overlap the max reduction below
    const double scale = mp_u2f((u64)(1023 + S) << 52);  // 2^S
    const u64 base = (u64)blockIdx.x * TILE + (u64)tid * SCAN_ITEMS;
    const bool full = base + SCAN_ITEMS <= n;
    double w[SCAN_ITEMS], xv[SCAN_ITEMS];
    if (full) {
        const double2 a = *reinterpret_cast<const double2*>(logw + base);
        const double2 b = *reinterpret_cast<const double2*>(logw + base + 2);
        w[0] = a.x; w[1] = a.y; w[2] = b.x; w[3] = b.y;
        const double2 xa = *reinterpret_cast<const double2*>(x0 + base);
        const double2 xb = *reinterpret_cast<const double2*>(x0 + base + 2);
        xv[0] = xa.x; xv[1] = xa.y; xv[2] = xb.x; xv[3] = xb.y;
    } else {
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; ++j) {
            w[j] = (base + j < n) ? logw[base + j] : MP_NEG_INF;
            xv[j] = (base + j < n) ? x0[base + j] : 0.;
        }
    }
    // m = max over the per-workgroup maxima (exact in any order)
    double m = MP_NEG_INF;
    for (int j = tid; j < nb; j += SCAN_THREADS) m = fmax(m, blockmax[j]);
    m = wave_max(m);
    if (lane == 0) s_red[wave] = m;
    reinterpret_cast<u64*>(s_guide)[tid] = 0ull;  // SCAN_THREADS x 8 B = the whole guide
    __syncthreads();
    m = s_red[0];
#pragma unroll
    for (int w = 1; w < SCAN_THREADS / 64; ++w) m = fmax(m, s_red[w]);
    const bool ok = (m > MP_NEG_INF) && (m < MP_INF);
    if (blockIdx.x == 0 && tid == 0) {
        scal->m = m;
        if (!ok) scal->degenerate = 1;
    }

    u64 c[SCAN_ITEMS];
    u64 run = 0, run2 = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        const bool live = ok && (base + j < n);
        const double e = live ? mp_exp(w[j] - m) : 0.;
        run += mp_quantize(e, scale);
        run2 += mp_quantize(e * e, scale);
        c[j] = run;
    }
    const u64 incl = wave_incl_scan_u64(run, lane);
    const u64 wtot2 = wave_sum_u64(run2);
    if (lane == 63) s_wsum[wave] = incl;
    if (lane == 0) s_wsum2[wave] = wtot2;
    __syncthreads();
    u64 woff = 0, W = 0;
#pragma unroll
    for (int k = 0; k < SCAN_THREADS / 64; ++k) {
        const u64 v = s_wsum[k];
        if (k < wave) woff += v;
        W += v;
    }
    const u64 off = woff + (incl - run);
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        if (base + j < n) {
            mp_cx row;
            row.cum = off + c[j];
            row.x0 = xv[j];
            cx[base + j] = row;
        }
    }
    if (tid == 0) {
        tilesum[blockIdx.x] = W;
        u64 t2 = 0;
#pragma unroll
        for (int k = 0; k < SCAN_THREADS / 64; ++k) t2 += s_wsum2[k];
        tilesum2[blockIdx.x] = t2;
    }

    // ---- guide table of this tile ------------------------------------------------------------
    const int shift = mp_guide_shift(W);
    u64 prev = off;
    int long_lo = 0, long_hi = -1, long_j = 0;  // at most one long run is kept per thread; extra ones fall back to direct writes
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        const u64 cur = off + c[j];
        if (cur > prev) {
            const int g_lo = prev ? (int)(prev >> shift) + 1 : 0;
            const int g_hi = (int)(cur >> shift);
            const unsigned short idx = (unsigned short)(tid * SCAN_ITEMS + j);
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
    reinterpret_cast<u64*>(guide + (u64)blockIdx.x * GUIDE_N)[tid] = reinterpret_cast<const u64*>(s_guide)[tid];
}

// ---------------------------------------------------------------------------------------------
// shared helper: workgroup-wide inclusive scan of nt (<= MAX_TILES) u64 values into LDS
// ---------------------------------------------------------------------------------------------
template <int THREADS>
__device__ __forceinline__ void block_scan_tiles(const u64* __restrict__ tilesum, int nt, u64* s_incl, u64* s_wtot) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (nt + THREADS - 1) / THREADS;
    const int b0 = tid * per;
    u64 run = 0;
    for (int j = 0; j < per; ++j) {
        const int idx = b0 + j;
        if (idx < nt) {
            run += tilesum[idx];
            s_incl[idx] = run;
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
}

__device__ __forceinline__ void finalize_scalars(u64 Q, u64 Q2, int S, double* L_out, double* ess_out, double m) {
    const double inv = mp_u2f((u64)(1023 - S) << 52);  // 2^-S
    const double Qs = (double)Q * inv, Q2s = (double)Q2 * inv;
    *L_out = m + mp_log(Qs);
    *ess_out = (Qs * Qs) / Q2s;
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

// first index in [0, len) with a[idx] >= target (len if none)
template <class Ptr>
__device__ __forceinline__ uint32_t lower_bound_u64(Ptr a, uint32_t len, u64 target) {
    uint32_t lo = 0, hi = len;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a[mid] >= target) hi = mid;
        else lo = mid + 1;
    }
    return lo;
}

// ---------------------------------------------------------------------------------------------
// K3: draw, search, gather, reset
// ---------------------------------------------------------------------------------------------
// ABL > 0 are timing-only ablations used by tools/k3_ablate.hip (1: no global reads, 2: guide only).
template <int ABL, bool SYSTEMATIC = false>
__global__ __launch_bounds__(K3_THREADS) void k_resample_gather(u64 n, u64 n_out, u64 n_global, u64 slot_offset, uint32_t domain,
                                                                uint32_t k0, uint32_t k1,
                                                                uint32_t rc, int S, int D, const mp_cx* __restrict__ cx,
                                                                const unsigned short* __restrict__ guide,
                                                                const u64* __restrict__ tilesum, const u64* __restrict__ tilesum2,
                                                                int nt, const double* __restrict__ x_old, double* __restrict__ x_new,
                                                                uint32_t* __restrict__ parent, double* __restrict__ logw,
                                                                double* __restrict__ blockmax, int nb, mp_dev_scalars* scal) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* s_incl = reinterpret_cast<u64*>(smem);  // [nt]
    u64* s_wtot = s_incl + nt;                   // [K3_THREADS/64]
    if (ABL == 5) {  // timing-only: no tile-total loads / scan
        for (int jj = threadIdx.x; jj < nt; jj += K3_THREADS) s_incl[jj] = (u64)(jj + 1) << 40;
        __syncthreads();
    } else {
        block_scan_tiles<K3_THREADS>(tilesum, nt, s_incl, s_wtot);
    }
    const u64 Q = s_incl[nt - 1];

    if (blockIdx.x == 0 && ABL == 0 && scal != nullptr) {  // workgroup-uniform: fold this normalisation into the filter scalars
        u64 q2 = 0;
        for (int j = threadIdx.x; j < nt; j += K3_THREADS) q2 += tilesum2[j];
        q2 = wave_sum_u64(q2);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) s_wtot[threadIdx.x >> 6] = q2;
        __syncthreads();
        if (threadIdx.x == 0) {
            u64 Q2 = 0;
            for (int k = 0; k < K3_THREADS / 64; ++k) Q2 += s_wtot[k];
            double L, ess;
            finalize_scalars(Q, Q2, S, &L, &ess, scal->m);
            scal->L = L;
            scal->ess_stale = ess;
            scal->Q = Q;
            scal->Q2 = Q2;
            scal->log_ml += L - mp_log((double)n_global);  // particle_filter.rs:105
        }
        for (int j = threadIdx.x; j < nb; j += K3_THREADS) blockmax[j] = 0.;  // logw is 0 after a resample
    }

    const uint32_t sys_k32 = SYSTEMATIC ? mp_systematic_k32(rc, k0, k1) : 0u;
    // Each thread resolves K3_ITEMS draws with independent load chains (Philox -> LDS tile search ->
    // guide entry -> two table rows), so that K3_ITEMS x 64 cache-line requests per wave are in flight
    // at every hop instead of 64: the kernel is bound by the latency of these dependent hops.
    for (u64 i0 = (u64)blockIdx.x * (K3_THREADS * K3_ITEMS) + threadIdx.x; i0 < n_out; i0 += (u64)gridDim.x * (K3_THREADS * K3_ITEMS)) {
        u64 lt[K3_ITEMS], tbase[K3_ITEMS];
        uint32_t tlen[K3_ITEMS], j[K3_ITEMS];
        const unsigned short* gp[K3_ITEMS];
#pragma unroll
        for (int k = 0; k < K3_ITEMS; ++k) {
            const u64 i = i0 + (u64)k * K3_THREADS;
            u64 target;
            if (SYSTEMATIC) {
                target = mp_target_systematic(slot_offset + (i < n_out ? i : 0), sys_k32, Q, n_global);
            } else {
                mp_u64x2 r;
                if (ABL == 3) r.a = (i * 0x9E3779B97F4A7C15ull) ^ ((u64)rc << 20);  // timing-only: no Philox
                else r = mp_philox4x32_10((uint32_t)(slot_offset + i), rc, (domain << 16), 0u, k0, k1);
                u64 k52 = mp_u52(r.a);
                if (ABL == 6) k52 = (k52 >> 3) | ((u64)(blockIdx.x & 7u) << 49);  // timing-only: draws confined to the XCD's eighth of the CDF
                target = mp_target(k52, Q);
            }
            uint32_t b = (ABL == 4) ? (uint32_t)((target >> 7) % (u64)nt)  // timing-only: no LDS search
                                    : lower_bound_u64(s_incl, (uint32_t)nt, target);
            if (b > (uint32_t)(nt - 1)) b = (uint32_t)(nt - 1);
            const u64 incl_b = s_incl[b];
            const u64 excl = b ? s_incl[b - 1] : 0ull;
            lt[k] = target - excl;                        // tile-local target, 1 <= lt <= W
            const int shift = mp_guide_shift(incl_b - excl);
            tbase[k] = (u64)b * TILE;
            tlen[k] = (uint32_t)((n - tbase[k]) < (u64)TILE ? (n - tbase[k]) : (u64)TILE);
            uint32_t g = (uint32_t)(lt[k] >> shift);
            if (g > GUIDE_N - 1) g = GUIDE_N - 1;
            gp[k] = guide + (u64)b * GUIDE_N + g;
        }
#pragma unroll
        for (int k = 0; k < K3_ITEMS; ++k) j[k] = (ABL == 1 || ABL >= 3) ? (uint32_t)(lt[k] & (TILE - 1)) : *gp[k];
        mp_cx r0[K3_ITEMS], r1[K3_ITEMS];
#pragma unroll
        for (int k = 0; k < K3_ITEMS; ++k) {
            if (j[k] > tlen[k] - 1) j[k] = tlen[k] - 1;
            const uint32_t j1 = (j[k] + 1 < tlen[k]) ? j[k] + 1 : j[k];
            if (ABL == 0) {
                r0[k] = load_row_nt(cx + tbase[k] + j[k]);   // streamed: must not evict the guide from L2
                r1[k] = load_row_nt(cx + tbase[k] + j1);
            } else {
                r0[k].cum = lt[k]; r0[k].x0 = (double)j[k]; r1[k] = r0[k];
            }
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
                    x_new[i] = row.x0;                    // traces[i] = traces[parents[i]].clone()
                    for (int d = 1; d < D; ++d) x_new[(u64)d * n_out + i] = x_old[(u64)d * n + p];
                }
                if (logw) logw[i] = 0.;                   // log_weights.fill(0.)
            }
        }
    }
}


// Tile of a global target: tile totals are nearly equal (each sums 2048 weights), so target * nt / Q lands within a
// tile or two of the answer; walk from there.  Same result as lower_bound_u64(s_incl, nt, target), fewer LDS reads.
__device__ __forceinline__ uint32_t tile_of_target(const u64* s_incl, uint32_t nt, u64 target, double nt_over_Q) {
    int b = (int)((double)target * nt_over_Q);
    if (b > (int)nt - 1) b = (int)nt - 1;
    if (b < 0) b = 0;
    while (b > 0 && s_incl[b - 1] >= target) --b;          // first b with incl[b] >= target ...
    while (b < (int)nt - 1 && s_incl[b] < target) ++b;     // ... from either side
    return (uint32_t)b;
}

// ---------------------------------------------------------------------------------------------
// XCD-binned multinomial resampling (same parents per slot as k_resample_gather, bit for bit)
// ---------------------------------------------------------------------------------------------
// The row table (16 B x N) does not fit one XCD's 4 MB L2, so random row reads cross the fabric a full line at a
// time (profiles/r01/k3_ablation_n2e20.txt).  But the top 3 bits of a draw's uniform say which EIGHTH of the CDF
// it lands in, and they do not depend on the weights.  So:
//   K3a k_bin_draws     every chunk of 1024 output slots splits its draws into 8 bins by those bits (stable order):
//                       segment [bin][chunk][<=1024] of uniforms (sparse addressing, dense traffic) and perm[slot] =
//                       (bin << 10 | position in its segment).
//   K3b k_resolve_bins  workgroup (group of 8 chunks, bin b) with blockIdx % 8 == b — workgroups are dealt
//                       round-robin over the 8 XCDs, so all lookups of bin b run on one XCD whose L2 then holds
//                       that eighth of the table (speed only: any placement gives the same result).

// K3a: per draw Philox -> target -> tile (LDS search over the tile totals) -> tile-local target lt and guide slot;
// stable split of the chunk's draws into the 8 CDF-eighth bins.  No random global access here.
__global__ __launch_bounds__(BIN_THREADS) void k_bin_draws(u64 n, u64 n_global, u64 slot_offset, uint32_t k0, uint32_t k1, uint32_t rc, int S, int nchunks,
                                                           const u64* __restrict__ tilesum, const u64* __restrict__ tilesum2, int nt,
                                                           const unsigned short* __restrict__ guide,
                                                           u64* __restrict__ seg_lt, uint32_t* __restrict__ seg_row,
                                                           unsigned short* __restrict__ perm, unsigned short* __restrict__ seg_cnt,
                                                           double* __restrict__ blockmax, int nb, mp_dev_scalars* scal) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* s_incl = reinterpret_cast<u64*>(smem);                               // [nt]
    u64* s_wtot = s_incl + nt;                                                // [BIN_THREADS/64]
    uint32_t* s_wcnt = reinterpret_cast<uint32_t*>(s_wtot + BIN_THREADS / 64);  // [BIN_ITEMS][BIN_THREADS/64][8] counts
    uint32_t* s_woff = s_wcnt + BIN_ITEMS * (BIN_THREADS / 64) * 8;             // same shape: exclusive offsets (+ [8] totals)
    block_scan_tiles<BIN_THREADS>(tilesum, nt, s_incl, s_wtot);
    const u64 Q = s_incl[nt - 1];
    const int c = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    if (blockIdx.x == 0) {  // fold this normalisation into the filter scalars (as k_resample_gather does)
        u64 q2 = 0;
        for (int j = threadIdx.x; j < nt; j += BIN_THREADS) q2 += tilesum2[j];
        q2 = wave_sum_u64(q2);
        __syncthreads();
        if (lane == 0) s_wtot[wave] = q2;
        __syncthreads();
        if (threadIdx.x == 0) {
            u64 Q2 = 0;
            for (int k = 0; k < BIN_THREADS / 64; ++k) Q2 += s_wtot[k];
            double L, ess;
            finalize_scalars(Q, Q2, S, &L, &ess, scal->m);
            scal->L = L;
            scal->ess_stale = ess;
            scal->Q = Q;
            scal->Q2 = Q2;
            scal->log_ml += L - mp_log((double)n_global);  // particle_filter.rs:105
        }
        for (int j = threadIdx.x; j < nb; j += BIN_THREADS) blockmax[j] = 0.;
    }

    u64 lt[BIN_ITEMS];
    uint32_t gidx[BIN_ITEMS], tile_of[BIN_ITEMS];
    int bin[BIN_ITEMS];
    uint32_t rank_in_wave[BIN_ITEMS];
    const double nt_over_Q = (double)nt / (double)Q;  // only a starting guess for the tile walk: no effect on results
#pragma unroll
    for (int q = 0; q < BIN_ITEMS; ++q) {
        const u64 i = (u64)c * BIN_CHUNK + (u64)q * BIN_THREADS + threadIdx.x;
        const mp_u64x2 r = mp_philox4x32_10((uint32_t)(slot_offset + i), rc, ((uint32_t)MP_DOM_RESAMPLE << 16), 0u, k0, k1);
        const u64 k52 = mp_u52(r.a);
        const u64 target = mp_target(k52, Q);
        const uint32_t b = tile_of_target(s_incl, (uint32_t)nt, target, nt_over_Q);
        const u64 incl_b = s_incl[b];
        const u64 excl = b ? s_incl[b - 1] : 0ull;
        lt[q] = target - excl;                           // tile-local target, 1 <= lt <= W
        uint32_t g = (uint32_t)(lt[q] >> mp_guide_shift(incl_b - excl));
        if (g > GUIDE_N - 1) g = GUIDE_N - 1;
        gidx[q] = b * (uint32_t)GUIDE_N + g;
        bin[q] = (i < n) ? (int)(k52 >> 49) : -1;
        tile_of[q] = b;
        rank_in_wave[q] = 0;
#pragma unroll
        for (int bb = 0; bb < 8; ++bb) {
            const u64 bal = __ballot(bin[q] == bb);
            if (bin[q] == bb) rank_in_wave[q] = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) s_wcnt[(q * (BIN_THREADS / 64) + wave) * 8 + bb] = (uint32_t)__popcll(bal);
        }
    }
    // the guide lookups go out now (the 2 MB guide is L2-resident on every XCD) and land while the offsets are built
    uint32_t j0[BIN_ITEMS];
#pragma unroll
    for (int q = 0; q < BIN_ITEMS; ++q) j0[q] = guide[gidx[q]];
    __syncthreads();
    // exclusive offsets in the stable order: item q-major (slots q*256 .. q*256+255), then wave, then lane == increasing slot
    constexpr int NW = BIN_THREADS / 64;
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
// V: timing-only ablation bits (tools/k3_ablate.hip): 1 = no result stores, 4 = no row loads
template <int V = 0>
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
    // first row >= lt, walking forward from `row` inside its tile; r0/r1 = that row and the next one when already loaded
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
        if (!(V & 1) || cur.x0 == 1.2345e301) {
            res_parent[sp] = (uint32_t)p;
            res_x[sp] = cur.x0;
            for (int d = 1; d < D; ++d) res_x[(u64)d * res_stride + sp] = x_old[(u64)d * n + p];
        }
    };
    bool live[K3_ITEMS];
    mp_cx r0[K3_ITEMS], r1[K3_ITEMS];
#pragma unroll
    for (int k = 0; k < K3_ITEMS; ++k) {
        live[k] = e0 < cnt[k];
        if (live[k] && !(V & 4)) {
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
                            const uint32_t* __restrict__ res_parent, double* __restrict__ x_new, uint32_t* __restrict__ parent,
                            double* __restrict__ logw) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t pr = perm[i];
    const u64 pos = MP_SEG_POS(pr >> 10, i >> 10, pr & 1023u, nchunks);
    for (int d = 0; d < D; ++d) x_new[(u64)d * n + i] = res_x[(u64)d * res_stride + pos];
    parent[i] = res_parent[pos];
    logw[i] = 0.;
}

// query path: log_marginal_likelihood_estimate / fresh ESS from the current log-weights (after K2)
__global__ __launch_bounds__(K3_THREADS) void k_lse_finalize(const u64* __restrict__ tilesum, const u64* __restrict__ tilesum2, int nt, int S,
                                                             u64 n_global, mp_dev_scalars* scal) {
    __shared__ u64 s_a[K3_THREADS / 64], s_b[K3_THREADS / 64];
    u64 q = 0, q2 = 0;
    for (int j = threadIdx.x; j < nt; j += K3_THREADS) {
        q += tilesum[j];
        q2 += tilesum2[j];
    }
    q = wave_sum_u64(q);
    q2 = wave_sum_u64(q2);
    if ((threadIdx.x & 63) == 0) {
        s_a[threadIdx.x >> 6] = q;
        s_b[threadIdx.x >> 6] = q2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 Q = 0, Q2 = 0;
        for (int k = 0; k < K3_THREADS / 64; ++k) {
            Q += s_a[k];
            Q2 += s_b[k];
        }
        double L, ess;
        finalize_scalars(Q, Q2, S, &L, &ess, scal->m);
        scal->ess_fresh = ess;
        scal->lml_fresh = scal->log_ml + L - mp_log((double)n_global);  // particle_filter.rs:119-121
    }
}


// ---------------------------------------------------------------------------------------------
// sharded filter phases (include/modppl_hip.h "sharded filter")
// ---------------------------------------------------------------------------------------------
constexpr int SH_THREADS = 256;
constexpr int SH_MAX_WORLD = 64;

__global__ __launch_bounds__(SH_THREADS) void k_reduce_max(const double* __restrict__ v, int nv, double* __restrict__ out) {
    __shared__ double s_m[SH_THREADS / 64];
    double m = MP_NEG_INF;
    for (int j = threadIdx.x; j < nv; j += SH_THREADS) m = fmax(m, v[j]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < SH_THREADS / 64; ++k) m = fmax(m, s_m[k]);
        out[0] = fmax(m, s_m[0]);
    }
}
__global__ __launch_bounds__(SH_THREADS) void k_sum_tiles(const u64* __restrict__ tilesum, const u64* __restrict__ tilesum2, int nt,
                                                          u64* __restrict__ out) {
    __shared__ u64 s_a[SH_THREADS / 64], s_b[SH_THREADS / 64];
    u64 q = 0, q2 = 0;
    for (int j = threadIdx.x; j < nt; j += SH_THREADS) { q += tilesum[j]; q2 += tilesum2[j]; }
    q = wave_sum_u64(q); q2 = wave_sum_u64(q2);
    if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = q; s_b[threadIdx.x >> 6] = q2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 Q = 0, Q2 = 0;
        for (int k = 0; k < SH_THREADS / 64; ++k) { Q += s_a[k]; Q2 += s_b[k]; }
        out[0] = Q; out[1] = Q2;
    }
}

// pass 1: target of every local slot -> owner rank + shard-local target; per-workgroup owner histogram
__global__ __launch_bounds__(SH_THREADS) void k_shard_targets(u64 n, u64 n_global, u64 slot_offset, uint32_t k0, uint32_t k1, uint32_t rc,
                                                              int systematic, const u64* __restrict__ totals_all, int world,
                                                              unsigned char* __restrict__ dest, u64* __restrict__ lt_out,
                                                              uint32_t* __restrict__ blockcount) {
    __shared__ u64 s_incl[SH_MAX_WORLD];
    __shared__ uint32_t s_cnt[SH_MAX_WORLD];
    if (threadIdx.x < SH_MAX_WORLD) s_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        u64 run = 0;
        for (int r = 0; r < world; ++r) { run += totals_all[2 * r]; s_incl[r] = run; }
    }
    __syncthreads();
    const u64 Q = s_incl[world - 1];
    const u64 i = (u64)blockIdx.x * SH_THREADS + threadIdx.x;
    if (i < n) {
        u64 target;
        if (systematic) {
            target = mp_target_systematic(slot_offset + i, mp_systematic_k32(rc, k0, k1), Q, n_global);
        } else {
            const mp_u64x2 r = mp_philox4x32_10((uint32_t)(slot_offset + i), rc, ((uint32_t)MP_DOM_RESAMPLE << 16), 0u, k0, k1);
            target = mp_target(mp_u52(r.a), Q);
        }
        int s = 0;
        while (s < world - 1 && s_incl[s] < target) ++s;  // first rank whose inclusive total reaches the target
        const u64 excl = s ? s_incl[s - 1] : 0ull;
        dest[i] = (unsigned char)s;
        lt_out[i] = target - excl;
        atomicAdd(&s_cnt[s], 1u);
    }
    __syncthreads();
    if (threadIdx.x < world) blockcount[(u64)blockIdx.x * world + threadIdx.x] = s_cnt[threadIdx.x];
}
// pass 2 (one workgroup): per-owner totals and exclusive per-workgroup offsets
__global__ __launch_bounds__(SH_THREADS) void k_shard_offsets(const uint32_t* __restrict__ blockcount, int nblk, int world,
                                                              uint32_t* __restrict__ blockoff, long long* __restrict__ counts) {
    // thread r < world walks the workgroups sequentially (nblk <= 65536, world <= 64: a few tens of microseconds at worst)
    const int r = threadIdx.x;
    if (r < world) {
        uint32_t run = 0;
        for (int b = 0; b < nblk; ++b) {
            blockoff[(u64)b * world + r] = run;
            run += blockcount[(u64)b * world + r];
        }
        counts[r] = (long long)run;
    }
}
// pass 3: stable pack of the requests grouped by owner
__global__ __launch_bounds__(SH_THREADS) void k_shard_pack(u64 n, const unsigned char* __restrict__ dest, const u64* __restrict__ lt_in,
                                                           const uint32_t* __restrict__ blockoff, const long long* __restrict__ counts, int world,
                                                           u64* __restrict__ req_out, uint32_t* __restrict__ req_slot) {
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
        req_out[pos] = lt_in[i];
        req_slot[pos] = (uint32_t)i;
    }
}
// owner side: shard-local targets -> parent rows
__global__ __launch_bounds__(K3_THREADS) void k_shard_resolve(u64 n, u64 n_req, u64 slot_offset, int D, const u64* __restrict__ req,
                                                              const mp_cx* __restrict__ cx, const unsigned short* __restrict__ guide,
                                                              const u64* __restrict__ tilesum, int nt, const double* __restrict__ x,
                                                              double* __restrict__ rows) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* s_incl = reinterpret_cast<u64*>(smem);
    u64* s_wtot = s_incl + nt;
    block_scan_tiles<K3_THREADS>(tilesum, nt, s_incl, s_wtot);
    for (u64 q = (u64)blockIdx.x * K3_THREADS + threadIdx.x; q < n_req; q += (u64)gridDim.x * K3_THREADS) {
        const u64 target = req[q];
        uint32_t b = lower_bound_u64(s_incl, (uint32_t)nt, target);
        if (b > (uint32_t)(nt - 1)) b = (uint32_t)(nt - 1);
        const u64 incl_b = s_incl[b];
        const u64 excl = b ? s_incl[b - 1] : 0ull;
        const u64 lt = target - excl;
        const int shift = mp_guide_shift(incl_b - excl);
        const u64 tbase = (u64)b * TILE;
        const uint32_t tlen = (uint32_t)((n - tbase) < (u64)TILE ? (n - tbase) : (u64)TILE);
        uint32_t g = (uint32_t)(lt >> shift);
        if (g > GUIDE_N - 1) g = GUIDE_N - 1;
        uint32_t j = guide[(u64)b * GUIDE_N + g];
        if (j > tlen - 1) j = tlen - 1;
        mp_cx row = load_row_nt(cx + tbase + j);
        while (row.cum < lt && j + 1 < tlen) {
            ++j;
            row = load_row_nt(cx + tbase + j);
        }
        const u64 p = tbase + j;
        double* out = rows + q * (u64)(D + 1);
        out[0] = row.x0;
        for (int d = 1; d < D; ++d) out[d] = x[(u64)d * n + p];
        out[D] = (double)(slot_offset + p);
    }
}
// requester side
__global__ __launch_bounds__(SH_THREADS) void k_shard_scatter(u64 n, int D, const double* __restrict__ rows, const uint32_t* __restrict__ req_slot,
                                                              double* __restrict__ x_new, uint32_t* __restrict__ parent, double* __restrict__ logw,
                                                              double* __restrict__ blockmax, int nb) {
    const u64 pos = (u64)blockIdx.x * SH_THREADS + threadIdx.x;
    if (pos < n) {
        const uint32_t i = req_slot[pos];
        const double* in = rows + pos * (u64)(D + 1);
        for (int d = 0; d < D; ++d) x_new[(u64)d * n + i] = in[d];
        parent[i] = (uint32_t)in[D];
        logw[i] = 0.;
    }
    if (blockIdx.x == 0)
        for (int j = threadIdx.x; j < nb; j += SH_THREADS) blockmax[j] = 0.;
}
// scalars of a sharded normalisation from the gathered totals; mode 0 = resample (fold into log_ml), 1 = query
__global__ void k_shard_finalize(const u64* __restrict__ totals_all, int world, int S, u64 n_global, int mode, mp_dev_scalars* scal) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        u64 Q = 0, Q2 = 0;
        for (int r = 0; r < world; ++r) { Q += totals_all[2 * r]; Q2 += totals_all[2 * r + 1]; }
        double L, ess;
        finalize_scalars(Q, Q2, S, &L, &ess, scal->m);
        if (mode == 0) {
            scal->L = L;
            scal->ess_stale = ess;
            scal->Q = Q; scal->Q2 = Q2;
            scal->log_ml += L - mp_log((double)n_global);
        } else {
            scal->ess_fresh = ess;
            scal->lml_fresh = scal->log_ml + L - mp_log((double)n_global);
        }
    }
}

// transpose SoA [d][n] -> host-facing AoS [n][d]
__global__ void k_soa_to_aos(const double* __restrict__ x, u64 n, int D, double* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        for (int d = 0; d < D; ++d) out[i * D + d] = x[(u64)d * n + i];
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int ceil_log2_u64(u64 n) {
    int b = 0;
    while ((1ull << b) < n) ++b;
    return b;
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
    double* blockmax;
    int grid;
    hipStream_t stream;
    const unsigned short* perm;
    const double* res_x;
    u64 res_stride;
    int nchunks;
};
struct ModelOps {
    int dim_state = 0, dim_obs = 0;
    int k1_items = 1;
    virtual ~ModelOps() {}
    virtual void propagate(const PropagateArgs& a) const = 0;
};
template <class Model>
struct ModelOpsT : ModelOps {
    Model model;
    explicit ModelOpsT(const Model& m) : model(m) {
        dim_state = Model::DIM_STATE;
        dim_obs = Model::DIM_OBS;
        k1_items = ::k1_items<Model>();
        static_assert(Model::DIM_STATE <= MP_MAX_STATE && Model::DIM_OBS <= MP_MAX_OBS, "model too wide for mp_obs / mp_state0");
    }
    void propagate(const PropagateArgs& a) const override {
        hipLaunchKernelGGL(k_propagate<Model>, dim3(a.grid), dim3(K1_THREADS), 0, a.stream, model, a.n, a.slot_offset, a.k0, a.k1, a.t,
                           a.x_in, a.x_out, a.logw, a.obs, a.s0, a.overwrite, a.blockmax, a.perm, a.res_x, a.res_stride, a.nchunks);
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
    int nb = 0;  // K1 grid == entries of blockmax
    int nt = 0;  // scan tiles
    int k3_grid = 0;
    // device buffers
    double* x[2] = {nullptr, nullptr};
    int cur = 0;
    double* logw = nullptr;
    mp_cx* cx = nullptr;
    unsigned short* guide = nullptr;
    uint32_t* parent = nullptr;
    double* blockmax = nullptr;
    u64* tilesum = nullptr;
    u64* tilesum2 = nullptr;
    mp_dev_scalars* scal = nullptr;
    double* aos = nullptr;  // staging for read_state
    mp_dev_scalars* h_scal = nullptr;  // pinned
    // binned resampling scratch: segments [bin][chunk][1024]
    u64* seg_lt = nullptr;              // [8 * nchunks * 1024]: tile-local target of every binned draw
    uint32_t* seg_row = nullptr;        // table row where the forward scan of every binned draw starts
    unsigned short* perm = nullptr;     // [n]: (bin << 10 | position) of every slot's draw
    unsigned short* seg_cnt = nullptr;
    double* res_x = nullptr;            // [d][8 * nchunks * 1024]: resampled states in segment order
    uint32_t* res_parent = nullptr;     // [8 * nchunks * 1024]
    u64 res_stride = 0;
    bool permuted = false;              // the current states / parents / (zero) log-weights live in res_* (lazy slot order)
    int nchunks = 0;
    int use_binned = 1;  // MP_BINNED_RESAMPLE=0 selects the single-kernel path (A/B measurements)
    // sharded-resample scratch (allocated on first use)
    unsigned char* sh_dest = nullptr;
    u64* sh_lt = nullptr;
    uint32_t* sh_req_slot = nullptr;
    uint32_t* sh_blockcount = nullptr;
    uint32_t* sh_blockoff = nullptr;
    long long* sh_counts = nullptr;
    long long* h_counts = nullptr;  // pinned
    int sh_world = 0;
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
    double fam_ms[MP_K_COUNT] = {0, 0, 0};
    uint64_t fam_launches[MP_K_COUNT] = {0, 0, 0};
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
    if (!h->permuted) return MP_OK;
    hipLaunchKernelGGL(k_unpermute, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, h->stream, h->n, h->ops->dim_state, h->nchunks, h->perm,
                       h->res_x, h->res_stride, h->res_parent, h->x[h->cur], h->parent, h->logw);
    h->permuted = false;
    return check_launch("k_unpermute");
}

static int32_t launch_propagate(mp_pf* h, const double* args0, const double* obs, bool overwrite) {
    PropagateArgs a;
    a.n = h->n; a.slot_offset = h->slot_offset;
    a.k0 = (uint32_t)h->seed; a.k1 = (uint32_t)(h->seed >> 32);
    a.t = h->t;
    a.x_in = h->x[h->cur]; a.x_out = h->x[h->cur];
    a.logw = h->logw;
    for (int j = 0; j < MP_MAX_OBS; ++j) a.obs.v[j] = (j < h->ops->dim_obs) ? obs[j] : 0.;
    for (int j = 0; j < MP_MAX_STATE; ++j) a.s0.v[j] = (args0 && j < h->ops->dim_state) ? args0[j] : 0.;
    a.overwrite = overwrite ? 1 : (h->permuted ? 2 : 0);
    a.perm = h->permuted ? h->perm : nullptr;
    a.res_x = h->res_x;
    a.res_stride = h->res_stride;
    a.nchunks = h->nchunks;
    a.blockmax = h->blockmax;
    a.grid = h->nb;
    a.stream = h->stream;
    {
        LaunchTimer lt(h, MP_K_PROPAGATE);
        h->ops->propagate(a);
    }
    h->t += 1;
    h->permuted = false;  // k_propagate wrote x[cur] and logw in slot order
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

static int32_t launch_normalize(mp_pf* h) {
    int32_t rcm = materialize(h);
    if (rcm != MP_OK) return rcm;
    LaunchTimer lt(h, MP_K_NORMALIZE_SCAN);
    hipLaunchKernelGGL(k_normalize_scan, dim3(h->nt), dim3(SCAN_THREADS), 0, h->stream, h->logw, h->x[h->cur], h->n, h->blockmax, h->nb, h->S,
                       h->cx, h->guide, h->tilesum, h->tilesum2, h->scal);
    return check_launch("k_normalize_scan");
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
    h->nb = (int)((n + (u64)K1_THREADS * h->ops->k1_items - 1) / ((u64)K1_THREADS * h->ops->k1_items));
    if (h->nb > K1_MAX_BLOCKS) h->nb = K1_MAX_BLOCKS;
    h->nt = (int)((n + TILE - 1) / TILE);
    if (h->nt > MAX_TILES) return mp_fail(MP_ERR_UNSUPPORTED, "n_particles per handle is limited to 2^24 in this build");
    h->k3_grid = (int)((n + K3_THREADS * K3_ITEMS - 1) / (K3_THREADS * K3_ITEMS));
    if (h->k3_grid > K3_MAX_BLOCKS) h->k3_grid = K3_MAX_BLOCKS;
    HIPCK(hipMalloc(&h->x[0], sizeof(double) * n * d));
    HIPCK(hipMalloc(&h->x[1], sizeof(double) * n * d));
    HIPCK(hipMalloc(&h->logw, sizeof(double) * n));
    HIPCK(hipMalloc(&h->cx, sizeof(mp_cx) * n));
    HIPCK(hipMalloc(&h->guide, sizeof(unsigned short) * (size_t)h->nt * GUIDE_N));
    HIPCK(hipMalloc(&h->parent, sizeof(uint32_t) * n));
    HIPCK(hipMalloc(&h->blockmax, sizeof(double) * K1_MAX_BLOCKS));
    HIPCK(hipMalloc(&h->tilesum, sizeof(u64) * h->nt));
    HIPCK(hipMalloc(&h->tilesum2, sizeof(u64) * h->nt));
    HIPCK(hipMalloc(&h->scal, sizeof(mp_dev_scalars)));
    HIPCK(hipMalloc(&h->aos, sizeof(double) * n * d));
    h->nchunks = (int)((n + BIN_CHUNK - 1) / BIN_CHUNK);
    {
        const char* env = getenv("MP_BINNED_RESAMPLE");
        if (env && env[0] == '0') h->use_binned = 0;
    }
    HIPCK(hipMalloc(&h->seg_lt, sizeof(u64) * 8 * (size_t)h->nchunks * BIN_CHUNK));
    HIPCK(hipMalloc(&h->seg_row, sizeof(uint32_t) * 8 * (size_t)h->nchunks * BIN_CHUNK));
    h->res_stride = 8ull * (u64)h->nchunks * BIN_CHUNK;
    HIPCK(hipMalloc(&h->perm, sizeof(unsigned short) * (size_t)h->nchunks * BIN_CHUNK));
    HIPCK(hipMalloc(&h->res_x, sizeof(double) * h->res_stride * (size_t)d));
    HIPCK(hipMalloc(&h->res_parent, sizeof(uint32_t) * h->res_stride));
    HIPCK(hipMalloc(&h->seg_cnt, sizeof(unsigned short) * 8 * (size_t)h->nchunks));
    HIPCK(hipHostMalloc(&h->h_scal, sizeof(mp_dev_scalars)));
    // ParticleSystem::new: log_weights = 0, parents = 0, log_ml_estimate = 0 (particle_filter.rs:44-57)
    HIPCK(hipMemsetAsync(h->x[0], 0, sizeof(double) * n * d, h->stream));
    HIPCK(hipMemsetAsync(h->x[1], 0, sizeof(double) * n * d, h->stream));
    HIPCK(hipMemsetAsync(h->logw, 0, sizeof(double) * n, h->stream));
    HIPCK(hipMemsetAsync(h->parent, 0, sizeof(uint32_t) * n, h->stream));
    HIPCK(hipMemsetAsync(h->blockmax, 0, sizeof(double) * K1_MAX_BLOCKS, h->stream));
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
    if (scheme != MP_RESAMPLE_MULTINOMIAL && scheme != MP_RESAMPLE_SYSTEMATIC) return mp_fail(MP_ERR_INVALID_ARG, "unknown resampling scheme");
    if (h->sharded) return mp_fail(MP_ERR_STATE, "sharded handle: resample runs through the mp_pf_shard_* phases");
    HIPCK(hipSetDevice(h->device));
    int32_t rc = launch_normalize(h);
    if (rc != MP_OK) return rc;
    const int d = h->ops->dim_state;
    const size_t lds = sizeof(u64) * ((size_t)h->nt + K3_THREADS / 64);
    bool binned = false;
    {
        LaunchTimer lt(h, MP_K_RESAMPLE_GATHER);
        if (scheme == MP_RESAMPLE_MULTINOMIAL && h->use_binned) {
            const size_t lds_a = sizeof(u64) * ((size_t)h->nt + BIN_THREADS / 64) + sizeof(uint32_t) * (2 * BIN_ITEMS * (BIN_THREADS / 64) * 8 + 8);
            hipLaunchKernelGGL(k_bin_draws, dim3(h->nchunks), dim3(BIN_THREADS), lds_a, h->stream, h->n, h->n_global, h->slot_offset, (uint32_t)h->seed,
                               (uint32_t)(h->seed >> 32), h->resample_count, h->S, h->nchunks, h->tilesum, h->tilesum2, h->nt, h->guide, h->seg_lt,
                               h->seg_row, h->perm, h->seg_cnt, h->blockmax, h->nb, h->scal);
            const int ngroups = (h->nchunks + BIN_GROUP - 1) / BIN_GROUP;
            hipLaunchKernelGGL(k_resolve_bins<0>, dim3(ngroups * 8), dim3(K3_THREADS), 0, h->stream, h->n, d, h->nchunks, h->seg_lt, h->seg_row,
                               h->seg_cnt, h->cx, h->x[h->cur], h->res_x, h->res_stride, h->res_parent);
            binned = true;
        } else if (scheme == MP_RESAMPLE_SYSTEMATIC)
            hipLaunchKernelGGL((k_resample_gather<0, true>), dim3(h->k3_grid), dim3(K3_THREADS), lds, h->stream, h->n, h->n, h->n_global, h->slot_offset,
                               (uint32_t)MP_DOM_RESAMPLE, (uint32_t)h->seed, (uint32_t)(h->seed >> 32), h->resample_count, h->S, d, h->cx, h->guide,
                               h->tilesum, h->tilesum2, h->nt, h->x[h->cur], h->x[h->cur ^ 1], h->parent, h->logw, h->blockmax, h->nb, h->scal);
        else
            hipLaunchKernelGGL((k_resample_gather<0, false>), dim3(h->k3_grid), dim3(K3_THREADS), lds, h->stream, h->n, h->n, h->n_global, h->slot_offset,
                               (uint32_t)MP_DOM_RESAMPLE, (uint32_t)h->seed, (uint32_t)(h->seed >> 32), h->resample_count, h->S, d, h->cx, h->guide,
                               h->tilesum, h->tilesum2, h->nt, h->x[h->cur], h->x[h->cur ^ 1], h->parent, h->logw, h->blockmax, h->nb, h->scal);
    }
    rc = check_launch("k_resample_gather");
    if (rc != MP_OK) return rc;
    if (binned) h->permuted = true;   // results stay in segment order; x[cur] is the (stale) pre-resample state
    else h->cur ^= 1;
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
    int32_t rc = launch_normalize(h);
    if (rc != MP_OK) return rc;
    hipLaunchKernelGGL(k_lse_finalize, dim3(1), dim3(K3_THREADS), 0, h->stream, h->tilesum, h->tilesum2, h->nt, h->S, h->n_global, h->scal);
    rc = check_launch("k_lse_finalize");
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
    const int d = h->ops->dim_state;
    hipLaunchKernelGGL(k_soa_to_aos, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, h->stream, h->x[h->cur], h->n, d, h->aos);
    int32_t rc = check_launch("k_soa_to_aos");
    if (rc != MP_OK) return rc;
    HIPCK(hipMemcpyAsync(x_out, h->aos, sizeof(double) * h->n * d, hipMemcpyDeviceToHost, h->stream));
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
int32_t mp_pf_shard_local_max(mp_pf* h, double* d_out) {
    if (!h || !d_out) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    if (!h->initialised) return mp_fail(MP_ERR_STATE, "resample before init_step");
    HIPCK(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_reduce_max, dim3(1), dim3(SH_THREADS), 0, h->stream, h->blockmax, h->nb, d_out);
    return check_launch("k_reduce_max");
}

int32_t mp_pf_shard_normalize(mp_pf* h, const double* d_global_max, uint64_t* d_totals_out) {
    if (!h || !d_global_max || !d_totals_out) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    if (!h->initialised) return mp_fail(MP_ERR_STATE, "resample before init_step");
    HIPCK(hipSetDevice(h->device));
    {
        LaunchTimer lt(h, MP_K_NORMALIZE_SCAN);
        hipLaunchKernelGGL(k_normalize_scan, dim3(h->nt), dim3(SCAN_THREADS), 0, h->stream, h->logw, h->x[h->cur], h->n, d_global_max, 1, h->S,
                           h->cx, h->guide, h->tilesum, h->tilesum2, h->scal);
    }
    int32_t rc = check_launch("k_normalize_scan");
    if (rc != MP_OK) return rc;
    hipLaunchKernelGGL(k_sum_tiles, dim3(1), dim3(SH_THREADS), 0, h->stream, h->tilesum, h->tilesum2, h->nt, (u64*)d_totals_out);
    return check_launch("k_sum_tiles");
}

int32_t mp_pf_shard_route(mp_pf* h, int32_t scheme, const uint64_t* d_totals_all, int32_t world, int32_t rank, uint64_t* d_req_out,
                          int64_t* send_counts) {
    if (!h || !d_totals_all || !d_req_out || !send_counts) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    if (scheme != MP_RESAMPLE_MULTINOMIAL && scheme != MP_RESAMPLE_SYSTEMATIC) return mp_fail(MP_ERR_INVALID_ARG, "unknown resampling scheme");
    if (world < 1 || world > SH_MAX_WORLD || rank < 0 || rank >= world) return mp_fail(MP_ERR_INVALID_ARG, "1 <= world <= 64, 0 <= rank < world");
    HIPCK(hipSetDevice(h->device));
    const int nblk = (int)((h->n + SH_THREADS - 1) / SH_THREADS);
    if (!h->sh_dest || h->sh_world < world) {
        (void)hipFree(h->sh_dest); (void)hipFree(h->sh_lt); (void)hipFree(h->sh_req_slot); (void)hipFree(h->sh_blockcount);
        (void)hipFree(h->sh_blockoff); (void)hipFree(h->sh_counts);
        if (h->h_counts) (void)hipHostFree(h->h_counts);
        HIPCK(hipMalloc(&h->sh_dest, h->n));
        HIPCK(hipMalloc(&h->sh_lt, sizeof(u64) * h->n));
        HIPCK(hipMalloc(&h->sh_req_slot, sizeof(uint32_t) * h->n));
        HIPCK(hipMalloc(&h->sh_blockcount, sizeof(uint32_t) * (size_t)nblk * world));
        HIPCK(hipMalloc(&h->sh_blockoff, sizeof(uint32_t) * (size_t)nblk * world));
        HIPCK(hipMalloc(&h->sh_counts, sizeof(long long) * SH_MAX_WORLD));
        HIPCK(hipHostMalloc(&h->h_counts, sizeof(long long) * SH_MAX_WORLD));
        h->sh_world = world;
    }
    {
        LaunchTimer lt(h, MP_K_RESAMPLE_GATHER);
        hipLaunchKernelGGL(k_shard_targets, dim3(nblk), dim3(SH_THREADS), 0, h->stream, h->n, h->n_global, h->slot_offset, (uint32_t)h->seed,
                           (uint32_t)(h->seed >> 32), h->resample_count, scheme == MP_RESAMPLE_SYSTEMATIC ? 1 : 0, (const u64*)d_totals_all, world,
                           h->sh_dest, h->sh_lt, h->sh_blockcount);
        hipLaunchKernelGGL(k_shard_offsets, dim3(1), dim3(SH_THREADS), 0, h->stream, h->sh_blockcount, nblk, world, h->sh_blockoff, h->sh_counts);
        hipLaunchKernelGGL(k_shard_pack, dim3(nblk), dim3(SH_THREADS), 0, h->stream, h->n, h->sh_dest, h->sh_lt, h->sh_blockoff, h->sh_counts, world,
                           (u64*)d_req_out, h->sh_req_slot);
    }
    int32_t rc = check_launch("k_shard_targets/offsets/pack");
    if (rc != MP_OK) return rc;
    // the finalisation of this normalisation (L, ESS, log-ML) only needs the gathered totals
    hipLaunchKernelGGL(k_shard_finalize, dim3(1), dim3(64), 0, h->stream, (const u64*)d_totals_all, world, h->S, h->n_global, 0, h->scal);
    rc = check_launch("k_shard_finalize");
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
    const size_t lds = sizeof(u64) * ((size_t)h->nt + K3_THREADS / 64);
    {
        LaunchTimer lt(h, MP_K_RESAMPLE_GATHER);
        hipLaunchKernelGGL(k_shard_resolve, dim3(grid), dim3(K3_THREADS), lds, h->stream, h->n, (u64)n_req, h->slot_offset, h->ops->dim_state,
                           (const u64*)d_req_in, h->cx, h->guide, h->tilesum, h->nt, h->x[h->cur], d_rows_out);
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
                           h->ops->dim_state, d_rows_in, h->sh_req_slot, h->x[h->cur ^ 1], h->parent, h->logw, h->blockmax, h->nb);
    }
    int32_t rc = check_launch("k_shard_scatter");
    if (rc != MP_OK) return rc;
    h->cur ^= 1;
    h->resample_count += 1;
    if (log_total_weight) {
        rc = fetch_scalars(h);
        if (rc != MP_OK) return rc;
        *log_total_weight = h->h_scal->L;
    }
    return MP_OK;
}

int32_t mp_pf_shard_query(mp_pf* h, const uint64_t* d_totals_all, int32_t world, double* log_ml, double* ess) {
    if (!h || !d_totals_all) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    HIPCK(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_shard_finalize, dim3(1), dim3(64), 0, h->stream, (const u64*)d_totals_all, world, h->S, h->n_global, 1, h->scal);
    int32_t rc = check_launch("k_shard_finalize");
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
            for (int k = 0; k < d; ++k)
                HIPCK(hipMemcpy(out + (size_t)t * d + k, (const double*)ev.buf + (size_t)k * h->n + a, sizeof(double), hipMemcpyDeviceToHost));
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
    (void)hipFree(h->x[0]);
    (void)hipFree(h->x[1]);
    (void)hipFree(h->logw);
    (void)hipFree(h->cx);
    (void)hipFree(h->guide);
    (void)hipFree(h->parent);
    (void)hipFree(h->blockmax);
    (void)hipFree(h->tilesum);
    (void)hipFree(h->tilesum2);
    (void)hipFree(h->scal);
    (void)hipFree(h->aos);
    (void)hipFree(h->seg_lt); (void)hipFree(h->seg_row); (void)hipFree(h->perm); (void)hipFree(h->seg_cnt); (void)hipFree(h->res_x); (void)hipFree(h->res_parent);
    (void)hipFree(h->sh_dest); (void)hipFree(h->sh_lt); (void)hipFree(h->sh_req_slot); (void)hipFree(h->sh_blockcount);
    (void)hipFree(h->sh_blockoff); (void)hipFree(h->sh_counts);
    if (h->h_counts) (void)hipHostFree(h->h_counts);
    (void)hipHostFree(h->h_scal);
    if (h->own_stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return MP_OK;
}

// out[i] = a[i] - *b  (log_normalized_weights = w_i - log_total_weight, importance.rs:23-25)
__global__ void k_sub_scalar(const double* __restrict__ a, const double* __restrict__ b, u64 n, double* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] - *b;
}
// IS query: log total weight and log_ml = L - ln N from the tile totals (importance.rs:21-22)
__global__ __launch_bounds__(K3_THREADS) void k_is_finalize(const u64* __restrict__ tilesum, int nt, int S, u64 n, mp_dev_scalars* scal) {
    __shared__ u64 s_a[K3_THREADS / 64];
    u64 q = 0;
    for (int j = threadIdx.x; j < nt; j += K3_THREADS) q += tilesum[j];
    q = wave_sum_u64(q);
    if ((threadIdx.x & 63) == 0) s_a[threadIdx.x >> 6] = q;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 Q = 0;
        for (int k = 0; k < K3_THREADS / 64; ++k) Q += s_a[k];
        const double inv = mp_u2f((u64)(1023 - S) << 52);
        const double L = scal->m + mp_log((double)Q * inv);
        scal->L = L;
        scal->lml_fresh = L - mp_log((double)n);
    }
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
    rc = launch_normalize(h);
    if (rc != MP_OK) return rc;
    hipLaunchKernelGGL(k_is_finalize, dim3(1), dim3(K3_THREADS), 0, h->stream, h->tilesum, h->nt, h->S, h->n, h->scal);
    rc = check_launch("k_is_finalize");
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
        const size_t lds = sizeof(u64) * ((size_t)h->nt + K3_THREADS / 64);
        hipLaunchKernelGGL((k_resample_gather<0, false>), dim3(grid), dim3(K3_THREADS), lds, h->stream, h->n, (u64)num_ret_samples, h->n_global, (u64)0,
                           (uint32_t)MP_DOM_IS, (uint32_t)h->seed, (uint32_t)(h->seed >> 32), 0u, h->S, h->ops->dim_state, h->cx, h->guide,
                           h->tilesum, h->tilesum2, h->nt, (const double*)nullptr, (double*)nullptr, d_idx, (double*)nullptr,
                           (double*)nullptr, 0, (mp_dev_scalars*)nullptr);
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
