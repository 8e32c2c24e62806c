// mp_pf_shard_kernels.h — device code of the sharded filter's resample phases (included by mp_pf.hip only, after
// mp_pf_kernels.h): the owner-keeps exchange (k_shard_table / k_shard_table_mw, k_shard_own_draw, k_shard_own_plan, k_shard_own_place), the
// slot-order exchange (variable-size three-pass route, fixed-capacity single-pass route, owner-side resolve), adoption.
// Protocol: include/modppl_hip.h "sharded filter", DESIGN.md §8.
#pragma once
#include "mp_binomial.h"
// ---------------------------------------------------------------------------------------------
// sharded filter phases (include/modppl_hip.h "sharded filter").  Shards are tile-aligned, so a shard's tiles are
// tiles of the job; every rank gathers all tiles' (m, W, W2) and builds the same table.
// ---------------------------------------------------------------------------------------------
constexpr int SH_THREADS = 256;
constexpr uint32_t MP_INV_LOCAL = 0x80000000u;   // inv[] entry of a kept offspring with a wide state: | the parent's local row
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
            target = mp_target(mp_resample_k52(slot_offset + i, rc, (uint32_t)MP_DOM_RESAMPLE, k0, k1), Q);
        }
        uint32_t b, gs;
        u64 lt;
        mp_locate<true>(s_incl, s_W, (uint32_t)nt_all, target, nt_over_Q, &b, &lt, &gs);
        const int s = (int)(b / (uint32_t)nt_local);
        dest[i] = (unsigned char)s;
        lt_out[i] = lt;
        tile_out[i] = b % (uint32_t)nt_local;
        atomicAdd(&s_cnt[s], 1u);
    }
    __syncthreads();
    if ((int)threadIdx.x < world) blockcount[(u64)blockIdx.x * world + threadIdx.x] = s_cnt[threadIdx.x];
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
        uint32_t j = mp_guide_row(guide[b * GUIDE_N + g]);
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
    unsigned long long counts[SH_MAX_WORLD];   // "owner keeps" form: offspring per rank of the last resample
    unsigned long long verdict;                // owner-keeps resample number k: (k << 8) | degenerate << 1 | overflow, ONE 8-byte store the host polls
};
constexpr int SHF_ITEMS = 8;   // draws per thread: 4 / 8 / 16 measured 22.7 / 19.9 / 25.6 us at world 1 and 35.0 / 27.6 / 32.6 us at world 8
constexpr int SH_BINS = 8;
constexpr int SH_MAX_KEYS = SH_MAX_WORLD * SH_BINS;
__global__ __launch_bounds__(SH_THREADS) void k_shard_route_fused(u64 n, u64 n_global, u64 slot_offset, uint32_t k0, uint32_t k1, uint32_t rc,
                                                                  int systematic, const u64* __restrict__ incl_all,
                                                                  const u64* __restrict__ tW_all, const double* __restrict__ ratio_all, int nt_all, int nt_local, int world, u64 capb,
                                                                  unsigned long long* __restrict__ counts, u64* __restrict__ req_out,
                                                                  uint32_t* __restrict__ inv, unsigned int* done, int* overflow) {
    __shared__ u64 s_base[SH_MAX_KEYS];        // [keys] start inside the sub-segment
    __shared__ uint32_t s_cnt[SH_MAX_KEYS];    // [keys] draws of this workgroup per sub-segment
    const int keys = world * SH_BINS;
    for (int k = threadIdx.x; k < keys; k += SH_THREADS) s_cnt[k] = 0;
    // The job's tile table was built once by k_shard_table and is probed where it lies (L2-resident: 16 B per tile of the
    // whole job).  It covers world x nt_local tiles: rebuilding it, or even copying it into LDS, in every workgroup costs
    // more than the draws themselves from 4 ranks up, and the LDS footprint would halve the occupancy.
    const u64* s_incl = incl_all;
    const u64* s_W = tW_all;
    __syncthreads();
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
                target = mp_target(mp_resample_k52(slot_offset + i, rc, (uint32_t)MP_DOM_RESAMPLE, k0, k1), Q);
            }
            uint32_t b, gs;
            mp_locate_r<true>(s_incl, s_W, ratio_all, (uint32_t)nt_all, target, nt_over_Q, &b, &lt[k], &gs);
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
    // The last workgroup to get here writes the sub-segment headers {count, "some sub-segment of mine overflowed"}.  Only
    // atomically updated words are read across workgroups (the counters, each workgroup's additions complete before its
    // ticket), so no fence is needed; the headers themselves are read by the next kernel / collective.
    __shared__ unsigned int s_ticket;
    if (threadIdx.x == 0) s_ticket = atomicAdd(done, 1u);
    __syncthreads();
    if (s_ticket == gridDim.x - 1) {
        int mine = 0;
        for (int k = threadIdx.x; k < keys; k += SH_THREADS) mine |= (atomicAdd(&counts[k], 0ull) > capb) ? 1 : 0;
        const int any = __syncthreads_or(mine);
        for (int k = threadIdx.x; k < keys; k += SH_THREADS) {
            const u64 c = atomicAdd(&counts[k], 0ull);
            u64* sub = req_out + (u64)k * (capb + 1) * 2;
            sub[0] = c < capb ? c : capb;
            sub[1] = (u64)any;
        }
        if (threadIdx.x == 0) {
            if (any) atomicOr(overflow, 1);
            atomicExch(done, 0u);
        }
    }
}
// Before the route, one workgroup: unpack the gathered tiles, build the job's tile table ONCE (inclusive prefix of T_b to
// global memory), zero the request counters, and fold this normalisation into the filter scalars (L, ESS, log-ML),
// keeping a copy for the case that the fixed-capacity exchange overflows.
// (mp_own_range — this rank's own draws of a resample — is declared in mp_pf_kernels.h: k_propagate's SHD form reads it too)
constexpr int MP_SCHEME_SPLIT = 3;           // MP_RESAMPLE_MULTINOMIAL_SPLIT (include/modppl_hip.h): sharded resamples only
// Offspring per rank of the split multinomial resample (mp_binomial.h): binary splitting of the N draws over the ranks, one binomial
// variate per tree node.  s_bound[r] = the job's inclusive prefix of T_b at the end of rank r's tiles; s_n = [2 * 64] heap of draw
// counts (node k: children 2k, 2k + 1; leaf of rank r: P + r).  Called by EVERY thread of the workgroup (barriers), after a barrier
// that made s_bound visible; the counts are s_n[P + r] afterwards, P = 1 << mp_split_levels(world).
// A variate is what mp_binomial_ratio returns — the FIRST accepted attempt of the sampler — but one lane walking a level of the tree
// is a chain of ~600 dependent double-precision instructions at ~4 ns each (2.3 us per level measured, cold or warm alike:
// profiles/r05/table_stamps.txt), so the work is laid out across lanes: MP_SPLIT_LANES adjacent lanes per node take one attempt
// each (mp_btrs_fast: the sampler's own operations); the uniforms (mp_split_uniforms: before the table is even there) and the node's
// masses — everything that does not depend on the node's draw count — are ready before the first level; the expensive acceptance
// test (mp_btrs_slow: v <= the sum of seven terms) runs only for an attempt in front of the first one the squeeze accepted, its eight
// pieces in the eight lanes of the group, added up in the sampler's order.  Tiny n p (the inversion branch) and "nothing accepted among
// the side-by-side attempts" fall back to the sequential sampler itself.
constexpr int MP_SPLIT_LANES = 8;
// a cold path as a real call: inlined, its constants (logarithms, the inversion loop) are hoisted over the level loop and spilled
// (measured both ways on one box: 20.6 us inlined against 19.8 for the whole count phase)
__device__ __attribute__((noinline)) u64 mp_binomial_ratio_cold(u64 n, u64 a, u64 b, uint32_t node, uint32_t rc, uint32_t k0, uint32_t k1) {
    return mp_binomial_ratio(n, a, b, node, rc, k0, k1);
}
// thread tid's attempt: the pair (U, V) of attempt tid % 8 at tree node tid / 8 (a pure function of the seed and the resample number)
__device__ __forceinline__ void mp_split_uniforms(uint32_t rc, uint32_t k0, uint32_t k1, double* U, double* V) {
    const mp_u64x2 blk = mp_philox4x32_10((uint32_t)threadIdx.x / MP_SPLIT_LANES, rc, ((uint32_t)MP_DOM_RESAMPLE << 16) | MP_SITE_SPLIT_COUNTS,
                                          (uint32_t)threadIdx.x % MP_SPLIT_LANES, k0, k1);
    *U = mp_u01(blk.a);
    *V = mp_u01(blk.b);
}
__device__ __forceinline__ double mp_shfl_f64(double x, int src_lane) {
    const u64 b = __builtin_bit_cast(u64, x);
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)b, src_lane), hi = (uint32_t)__shfl((int)(uint32_t)(b >> 32), src_lane);
    return __builtin_bit_cast(double, ((u64)hi << 32) | lo);
}
// what a lane needs at its node's level, parked in LDS meanwhile: nothing of it is live in registers across the levels of the tree
// (and across the call of the cold path, whose caller-saved registers were spilled to scratch — a first touch of scratch memory in
// front of a barrier, 1.5 us)
struct mp_split_lds {
    double U[SH_MAX_WORLD * MP_SPLIT_LANES], V[SH_MAX_WORLD * MP_SPLIT_LANES];   // [node * 8 + attempt]
    double p[SH_MAX_WORLD];                                                      // [node] the smaller share of the node's mass
    u64 ma[SH_MAX_WORLD], mb[SH_MAX_WORLD];                                      // [node] left mass, whole mass
};
__device__ __forceinline__ void mp_split_counts(const u64* s_bound, int world, u64 n_global, uint32_t rc, uint32_t k0, uint32_t k1, u64* s_n, double U0,
                                                double V0, mp_split_lds* sl) {
    constexpr int A = MP_SPLIT_LANES;
    static_assert(A == 8, "the acceptance test is laid out over eight lanes");
    const int tid = threadIdx.x;
    const int L = mp_split_levels(world), P = 1 << L;
    {
        const int node = tid / A;
        if (node >= 1 && node < P) {   // (P <= 64: the first 512 threads at most)
            sl->U[tid] = U0;
            sl->V[tid] = V0;
            if (tid % A == 0) {
                const int lvl = 31 - __clz(node);
                const int width = P >> lvl;   // leaves under this node: [a, a + width), split at a + width / 2
                const int a = (node - (1 << lvl)) * width, mid = a + width / 2, b = a + width;
                // mass of the leaves [0, r): ranks beyond `world` are empty
                const u64 pa = a <= 0 ? 0ull : s_bound[(a < world ? a : world) - 1];
                const u64 pm = s_bound[(mid < world ? mid : world) - 1];   // (mid >= 1)
                const u64 pb = s_bound[(b < world ? b : world) - 1];
                const u64 ma = pm - pa, mb = pb - pa, other = mb - ma;
                sl->ma[node] = ma;
                sl->mb[node] = mb;
                sl->p[node] = (double)(ma > other ? other : ma) / (double)mb;   // mp_binomial_ratio samples the smaller of the two masses
            }
        }
    }
    if (tid == 0) s_n[1] = n_global;
    __syncthreads();
    MP_STAMP(2, 9, 1);
    for (int l = 0; l < L; ++l) {
        // the nodes of level l: 1 << l of them, A lanes each — threads [A << l, A << (l + 1))
        if ((tid >> l) / A == 1) {   // (whole groups of A lanes; everything below that is not per attempt is the same in the A lanes)
            const int node = tid / A, att = tid % A;
            const int sh = (tid & 63) & ~(A - 1);      // first lane of this node's group inside its wave
            const double U = sl->U[tid], V = sl->V[tid], p = sl->p[node];
            const u64 ma = sl->ma[node], mb = sl->mb[node];
            const bool flipped = ma > mb - ma;
            const u64 nk = s_n[node];
            const bool trivial = nk == 0ull || ma == 0ull || ma >= mb;
            int st = 0;             // this lane's attempt: 0 rejected, 1 accepted, 2 the squeeze did not decide
            double k = 0.;
            mp_btrs T = {};
            bool generic = false;   // the sequential sampler decides
            if (!trivial) {
                const double n = (double)nk;
                if (n * p < 10.) generic = true;
                else {
                    mp_btrs_setup(T, n, p);
                    st = mp_btrs_fast(T, U, V, &k);
                }
            }
            uint32_t acc = (uint32_t)(__ballot(st == 1) >> sh) & 0xFFu;
            // the sampler takes the FIRST accepted attempt: an undecided one matters only in front of the first the squeeze accepted
            uint32_t und = (uint32_t)(__ballot(st == 2) >> sh) & 0xFFu & (acc ? ((acc & (0u - acc)) - 1u) : 0xFFu);
            while (und) {   // (uniform in the group) the first undecided attempt, its test by the eight lanes
                const int cand = __ffs((int)und) - 1;
                const double Uc = mp_shfl_f64(U, sh + cand), Vc = mp_shfl_f64(V, sh + cand), kc = mp_shfl_f64(k, sh + cand);
                const double piece = mp_btrs_piece(T, att, Uc, Vc, kc);   // (lanes 0 .. 3 a logarithm each, lanes 4 .. 7 a Stirling tail each)
                const bool ok = mp_btrs_accept(mp_shfl_f64(piece, sh), mp_shfl_f64(piece, sh + 1), mp_shfl_f64(piece, sh + 2), mp_shfl_f64(piece, sh + 3),
                                               mp_shfl_f64(piece, sh + 4), mp_shfl_f64(piece, sh + 5), mp_shfl_f64(piece, sh + 6), mp_shfl_f64(piece, sh + 7));
                if (ok) {   // accepted, and in front of every attempt the squeeze accepted
                    acc |= 1u << cand;
                    break;
                }
                und &= und - 1u;
            }
            if (!trivial && !generic && acc == 0u) generic = true;
            u64 left = 0ull;
            bool writer = att == 0;
            if (trivial) {
                left = (nk == 0ull || ma == 0ull) ? 0ull : nk;
            } else if (generic) {
                if (writer) left = mp_binomial_ratio_cold(nk, ma, mb, (uint32_t)node, rc, k0, k1);
            } else {
                writer = att == __ffs((int)acc) - 1;
                const u64 kk = (u64)k;
                left = flipped ? nk - kk : kk;
            }
            if (writer) {
                s_n[2 * node] = left;
                s_n[2 * node + 1] = nk - left;
            }
        }
        __syncthreads();
        MP_STAMP(2, 10 + (l < 6 ? l : 5), 1);
    }
}
// Owner-keeps exchange under a lattice scheme: targets are non-decreasing in g, so the draws that land in rank r's rows are the
// range [G_{r-1}, G_r) with G_r = #{g : target(g) <= B_r}, B_r = the end of r's tiles — the same target function as the draw kernels,
// so the ranges are exactly the draws those kernels will call their own.  target(g) = ((g 2^32 + k32_g) Q >> 32) / N + 1 crosses B_r
// within a draw of g* = B_r N / Q: 8 adjacent lanes per rank evaluate g* - 4 .. g* + 3 and count the targets <= B_r; a pattern that
// is not "true .. true false .. false" (it always is) would be settled by the bisection this replaces (33 serial probes, 3.3 - 5.5 us).
// Called by every thread; s_G[r] = G_r afterwards (a barrier has passed).
__device__ __forceinline__ void mp_lattice_ranges(const u64* s_bound, u64* s_G, int world, int scheme, u64 Q, u64 n_global, uint32_t rc, uint32_t k0,
                                                  uint32_t k1) {
    constexpr int W = 8;
    const int tid = threadIdx.x, r = tid / W, j = tid % W;
    if (r < world) {
        const int sh = (tid & 63) & ~(W - 1);
        const uint32_t k32 = scheme == 1 ? mp_systematic_k32(rc, k0, k1) : 0u;
        const u64 B = s_bound[r];
        const double est = (double)B * (double)n_global / (double)Q;   // (Q == 0: NaN or inf)
        const u64 gc = !(est >= 0.) ? 0ull : (est >= (double)n_global ? n_global : (u64)est);
        const long long g = (long long)gc - W / 2 + j;
        bool le;
        if (g < 0) le = true;
        else if ((u64)g >= n_global) le = false;
        else le = mp_target_lattice(scheme, (u64)g, k32, rc, k0, k1, Q, n_global) <= B;
        const uint32_t pat = (uint32_t)(__ballot(le) >> sh) & 0xFFu;
        if (j == 0) {
            u64 G;
            if ((pat & 1u) && !(pat & 0x80u) && (pat & (pat + 1u)) == 0u) {
                G = (u64)((long long)gc - W / 2 + (long long)__popc(pat));
            } else {
                u64 lo = 0, hi = n_global;
                while (lo < hi) {   // at most 33 probes
                    const u64 mid = lo + ((hi - lo) >> 1);
                    if (mp_target_lattice(scheme, mid, k32, rc, k0, k1, Q, n_global) <= B) lo = mid + 1;
                    else hi = mid;
                }
                G = lo;
            }
            s_G[r] = G;
        }
    }
    __syncthreads();
}
// the exchange plan: unit u of the surplus (donors in rank order) fills unit u of the deficit (receivers in rank order)
struct mp_owned_plan {
    u64 S[SH_MAX_WORLD], D[SH_MAX_WORLD], PS[SH_MAX_WORLD], PD[SH_MAX_WORLD];
};
// stores the host will read while the stream is still running: write-through to system scope (sc0 sc1), no cache write-back
__device__ __forceinline__ void mp_st_sys(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void mp_st_sys(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void mp_st_sys(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
struct mp_own_plan_args {
    u64 n, n_global, cap;
    int world, nsc, lattice, S;
    uint32_t* sccnt;                 // [nsc] own draws per super-chunk (in), read with agent-scope loads
    unsigned long long* c_all;       // [world] offspring per rank: k_shard_table (lattice: closed form; multinomial: zeroed) + the draw kernel's atomics
    mp_dev_scalars* scal;
    mp_dev_scalars* undo;
    const mp_tab_head* head;         // world of one: the single filter's table (ensure_table) is the job's table; k_shard_own_draw folds it
    uint32_t* base;                  // [nsc] out: first offspring position of every super-chunk
    mp_owned_plan* plan_out;
    mp_shard_pub* pub;               // host-mapped
    unsigned long long seq;
    const mp_own_range* range;       // lattice: only the super-chunks of the own range were written (and are scanned)
    u64 Wd;
};
template <int THREADS>
__device__ __forceinline__ void mp_own_plan(const mp_own_plan_args& a);
constexpr int SHT_THREADS = 1024;
constexpr int SHT_PER = MAX_TILES / SHT_THREADS;   // tiles per thread, held in registers (8)
// owner side: blockIdx.x & 7 = eighth of this shard's tiles (workgroups are dealt round-robin to the 8 XCDs, gridDim.x is
// a multiple of 8), blockIdx.y = asking rank.  Per request: guide cell -> first row -> short forward walk, all inside
// the eighth.  Rows go back in the order the requests came.
constexpr int SHR_ITEMS = 1;   // requests per thread and round
__global__ __launch_bounds__(K3_THREADS) void k_shard_resolve_binned(u64 n, u64 capb, u64 slot_offset, int D, const u64* __restrict__ req,
                                                                     const mp_cx* __restrict__ cx, const unsigned short* __restrict__ guide,
                                                                     const u64* __restrict__ tile_W, const double* __restrict__ x,
                                                                     double* __restrict__ rows, int* overflow) {
    const int bin = blockIdx.x & (SH_BINS - 1), grp = blockIdx.x >> 3, ngrp = gridDim.x >> 3;
    const u64 key = (u64)blockIdx.y * SH_BINS + bin;
    const ulonglong2* sub = reinterpret_cast<const ulonglong2*>(req) + key * (capb + 1);
    const ulonglong2 head = sub[0];
    const u64 cnt = head.x < capb ? head.x : capb;
    if (grp == 0 && threadIdx.x == 0 && head.y) atomicOr(overflow, 1);
    double* out_sub = rows + key * capb * (u64)(D + 1);
    for (u64 q0 = (u64)grp * (K3_THREADS * SHR_ITEMS); q0 < cnt; q0 += (u64)ngrp * (K3_THREADS * SHR_ITEMS)) {
        u64 lt[SHR_ITEMS], tbase[SHR_ITEMS], last[SHR_ITEMS];
        uint32_t gi[SHR_ITEMS];
        bool live[SHR_ITEMS];
#pragma unroll
        for (int k = 0; k < SHR_ITEMS; ++k) {   // hop 0: the requests (coalesced)
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
        u64 p[SHR_ITEMS];
#pragma unroll
        for (int k = 0; k < SHR_ITEMS; ++k) {   // hop 1: guide cells
            u64 j = tbase[k] + mp_guide_row(guide[gi[k]]);
            p[k] = j < last[k] ? j : last[k];
        }
        mp_cx r0[SHR_ITEMS], r1[SHR_ITEMS];
#pragma unroll
        for (int k = 0; k < SHR_ITEMS; ++k) {   // hop 2: the row and its successor
            r0[k] = cx[p[k]];
            r1[k] = cx[p[k] + (p[k] < last[k] ? 1 : 0)];
        }
        // lanes of a row group when states are wider than one double: G lanes copy one parent's row together, so that a
        // row is one coalesced request instead of D scattered 8-byte ones
        const int G = D <= 2 ? 2 : (D <= 4 ? 4 : (D <= 8 ? 8 : 16));
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int k = 0; k < SHR_ITEMS; ++k) {
            mp_cx cur = r0[k];
            u64 pp = p[k];
            if (live[k] && cur.cum < lt[k] && pp < last[k]) {
                cur = r1[k];
                ++pp;
                while (cur.cum < lt[k] && pp < last[k]) {
                    ++pp;
                    cur = cx[pp];
                }
            }
            const u64 q = q0 + (u64)k * K3_THREADS + threadIdx.x;
            if (D == 1) {
                if (live[k]) *reinterpret_cast<double2*>(out_sub + q * 2) = make_double2(cur.x0, (double)(slot_offset + pp));
            } else {
                if (live[k]) out_sub[q * (u64)(D + 1) + D] = (double)(slot_offset + pp);
                // the wave's 64 requests of this round, 64 / G at a time (wave-uniform trip count; D <= 16)
                const uint32_t pp_lo = (uint32_t)pp, pp_hi = (uint32_t)(pp >> 32);
                const uint32_t ql = (uint32_t)(q - q0);   // < K3_THREADS * SHR_ITEMS
                const int comp = lane % G;
                for (int base = 0; base < 64; base += 64 / G) {
                    const int src = base + lane / G;
                    const int lv = __shfl((int)live[k], src, 64);
                    const u64 spp = ((u64)(uint32_t)__shfl((int)pp_hi, src, 64) << 32) | (u64)(uint32_t)__shfl((int)pp_lo, src, 64);
                    const u64 sq = q0 + (u64)(uint32_t)__shfl((int)ql, src, 64);
                    if (lv) {
                        for (int d = comp; d < D; d += G) out_sub[sq * (u64)(D + 1) + d] = x[spp * D + d];
                    }
                }
            }
        }
    }
}
// ---------------------------------------------------------------------------------------------
// "Owner keeps" form of the sharded resample.  Every rank enumerates the N draws of the job (the draws of the single
// filter: same Philox counters, same targets, hence the same parent for every draw g) and keeps those that land in its own
// rows; offspring then stay on the rank that owns their parent, in the order of their draws, and only the surplus over n
// slots travels (to the ranks that drew fewer than n): the xGMI traffic of a resample drops from ~40 B per particle to a
// few thousand rows.  WHERE an offspring sits depends on the number of ranks (a world of one is the single filter).
//
// Round 3: a kept offspring is a {tile-local target, start row} pair in the slot-order arrays the NEXT k_propagate consumes,
// exactly as after an unsharded resample (the parent is local by construction); only the surplus is looked up early.
//   k_shard_table[_mw]  the job's tile table (one workgroup, or one per rank beyond 2048 tiles); the rank boundaries as
//                       thresholds on the uniforms (multinomial); for the lattice schemes every rank's own range of draws
//                       [G_{r-1}, G_r) from the monotone target function (mp_lattice_ranges): offspring counts in closed form,
//                       and a rank only looks at its own draws — O(n_local), whatever the world size.
//   k_shard_own_draw    super-chunk = R x 1024 consecutive draws, R = min(world, 4) (multinomial: resident workgroups take
//                       turns; a WAVE enumerates 128 R of them and compacts its own ones in LDS without a workgroup barrier;
//                       offspring of every rank counted with ballots against the boundaries, added to the job's counts with
//                       `world` atomics per workgroup) or 1 (lattice); then for the own draws: target -> tile -> guide ->
//                       start row, written in draw order into the super-chunk's window.
//   k_shard_own_plan    (one workgroup) first offspring position of every super-chunk (scan), the exchange plan, the
//                       verdict "some pair needs more than cap rows", what the host reads.
//   k_shard_own_place   offspring p = base + rank: p < n -> the pair goes to slot p; p >= n -> looked up here, its row goes
//                       to the send buffer where the plan says; deficit slots get MP_DRAW_RECV | the arriving row's index.
// No counting of offspring per row, no scattered 4-byte read-modify-writes (a first version spent 36 of 73 us there).
// ---------------------------------------------------------------------------------------------
constexpr int OWN_THREADS = 512;             // a thread owns two ADJACENT draws of a round (one Philox block)
constexpr int OWN_ROUND = 2 * OWN_THREADS;   // draws per round
constexpr int OWN_RMAX = 4;                  // rounds per workgroup at most (LDS: 8 B per draw of the super-chunk)
constexpr int OWN_NW = OWN_THREADS / 64;
__host__ __device__ inline int mp_own_rounds(int world) { return world < OWN_RMAX ? world : OWN_RMAX; }

// lattice schemes: first and last super-chunk (of Wd draws) that hold a draw of this rank (last < first: none)
__device__ __forceinline__ void mp_own_span(const mp_own_range* range, u64 Wd, int* sc_first, int* sc_last) {
    const u64 glo = range->g_lo, ghi = range->g_hi;
    *sc_first = (int)(glo / Wd);
    *sc_last = ghi > glo ? (int)((ghi - 1) / Wd) : *sc_first - 1;
}
// tile of a target inside ONE rank's tiles.  `incl` = the rank's entries of the job's inclusive prefix, either as they
// are (tabs = target, excl0 = prefix below the rank's first tile) or rebased to the rank's share (tabs = trel, excl0 = 0)
__device__ __forceinline__ void mp_locate_own(const u64* incl, const u64* W_, const double* ratio, uint32_t nt, u64 tabs, u64 trel, u64 excl0,
                                              double nt_over_span, uint32_t* tile, u64* lt, uint32_t* gslot) {
    int b = (int)((double)trel * nt_over_span);   // only a starting guess for the walk: no effect on the result
    if (b > (int)nt - 1) b = (int)nt - 1;
    if (b < 0) b = 0;
    int budget = 4;                               // (a walk that uses its budget up — collapsed weights — is settled by bisection)
    while (b > 0 && incl[b - 1] >= tabs && budget > 0) { --b; --budget; }
    while (b < (int)nt - 1 && incl[b] < tabs && budget > 0) { ++b; --budget; }
    if (budget == 0) {
        uint32_t lo = 0u, hi = nt - 1u;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (incl[mid] >= tabs) hi = mid;
            else lo = mid + 1u;
        }
        b = (int)lo;
    }
    const u64 excl = b ? incl[b - 1] : excl0;
    const u64 W = W_[b];
    const u64 x = mp_local_target_r(tabs - excl, W, ratio[b]);
    uint32_t g = (uint32_t)(x >> mp_guide_shift(W));
    if (g > GUIDE_N - 1) g = GUIDE_N - 1;
    *tile = (uint32_t)b; *lt = x; *gslot = (uint32_t)b * (uint32_t)GUIDE_N + g;
}

// by ONE workgroup of THREADS threads
template <int THREADS>
__device__ __forceinline__ void mp_own_plan(const mp_own_plan_args& a) {
    __shared__ mp_owned_plan pl;
    __shared__ u64 s_ptot[THREADS / 64];
    __shared__ u64 s_c[SH_MAX_WORLD];
    __shared__ int s_over;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int world = a.world;
    int nsc = a.nsc;
    const uint32_t* sccnt = a.sccnt;
    uint32_t* base = a.base;
    if (a.lattice) {
        int f, l;
        mp_own_span(a.range, a.Wd, &f, &l);
        sccnt += f; base += f; nsc = l - f + 1;
    }
    // every load this workgroup depends on goes out now (one round trip instead of three in a kernel that is all latency)
    const int per = (nsc + THREADS - 1) / THREADS;   // scan below: thread t owns entries [t * per, (t + 1) * per)
    const int b0 = tid * per;
    u64 run = 0;
    for (int j = 0; j < per; ++j)
        if (b0 + j < nsc) run += sccnt[b0 + j];
    const double scal_L = a.scal->L;
    const int scal_deg = a.scal->degenerate;
    // offspring per rank: k_shard_table found them in closed form (lattice), or the draw kernel's workgroups added them up
    if (tid < world) s_c[tid] = mp_ld_agent(reinterpret_cast<const u64*>(a.c_all) + tid);
    if (tid == 0) s_over = 0;
    __syncthreads();
    if (tid == 0) {
        u64 ps = 0, pd = 0;
        for (int r = 0; r < world; ++r) {
            const u64 c = s_c[r];
            pl.S[r] = c > a.n ? c - a.n : 0ull;
            pl.D[r] = c < a.n ? a.n - c : 0ull;
            pl.PS[r] = ps; pl.PD[r] = pd;
            ps += pl.S[r]; pd += pl.D[r];
            mp_st_sys(&a.pub->counts[r], (unsigned long long)c);
        }
        mp_st_sys(&a.pub->L, scal_L);
        mp_st_sys(&a.pub->degenerate, scal_deg);
    }
    // exclusive scan of the super-chunks' own counts: thread t owns entries [t * per, (t + 1) * per)
    const u64 incl = wave_incl_scan_u64(run, lane);
    if (lane == 63) s_ptot[wave] = incl;
    __syncthreads();
    u64 woff = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w)
        if (w < wave) woff += s_ptot[w];
    u64 off = woff + (incl - run);
    for (int j = 0; j < per; ++j) {
        if (b0 + j < nsc) {
            base[b0 + j] = (uint32_t)off;
            off += sccnt[b0 + j];
        }
    }
    // every rank must reach the same verdict on "some pair needs more than cap rows" (the collective that follows is symmetric),
    // so every rank looks at every pair of the plan
    int over = 0;
    if (a.cap) {
        for (int pq = tid; pq < world * world; pq += THREADS) {
            const int r = pq / world, s2 = pq - r * world;
            const u64 a0 = pl.PS[r] > pl.PD[s2] ? pl.PS[r] : pl.PD[s2];
            const u64 e0 = pl.PS[r] + pl.S[r], e1 = pl.PD[s2] + pl.D[s2];
            const u64 a1 = e0 < e1 ? e0 : e1;
            if (a1 > a0 && a1 - a0 > a.cap) over = 1;
        }
    }
    if (over) atomicOr(&s_over, 1);
    for (int r = tid; r < world; r += THREADS) {
        a.plan_out->S[r] = pl.S[r]; a.plan_out->D[r] = pl.D[r]; a.plan_out->PS[r] = pl.PS[r]; a.plan_out->PD[r] = pl.PD[r];
    }
    __syncthreads();
    if (tid == 0) {
        mp_st_sys(&a.pub->overflow, s_over);
        // what the host polls between resamples: one word, so it needs no ordering against the fields above (those are read
        // after the stream has drained: log total weight of a synchronous resample, counts of the exact-size repeat)
        mp_st_sys(&a.pub->verdict, (a.seq << 8) | (unsigned long long)((scal_deg ? 2 : 0) | (s_over ? 1 : 0)));
    }
}
constexpr int SHP_THREADS = 1024;
__global__ __launch_bounds__(SHP_THREADS) void k_shard_own_plan(mp_own_plan_args a) { mp_own_plan<SHP_THREADS>(a); }

template <int TABMODE, bool LATTICE>   // TABMODE 1: this rank's part of the tile table copied to LDS (rebased); 2: probed where it lies (more than 1024 tiles)
__global__ __launch_bounds__(OWN_THREADS) __attribute__((amdgpu_num_sgpr(80))) void k_shard_own_draw(
    u64 n, u64 n_global, uint32_t k0, uint32_t k1, uint32_t rc, int scheme, int R, const u64* __restrict__ incl_all, const u64* __restrict__ tW_all,
    const double* __restrict__ ratio_all, int nt_local, int world, int rank, const unsigned short* __restrict__ guide,
    const mp_own_range* __restrict__ range, u64* __restrict__ win_lt, uint32_t* __restrict__ win_row, mp_own_plan_args pa,
    const u64* __restrict__ kthr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nt_lds = TABMODE == 1 ? nt_local : 0;
    u64* s_tgt = reinterpret_cast<u64*>(smem);                          // [R * 1024] own targets of the super-chunk, draw order
    u64* s_incl_lds = s_tgt + (size_t)R * OWN_ROUND;                    // [nt_local]
    u64* s_W_lds = s_incl_lds + nt_lds;
    double* s_ratio_lds = reinterpret_cast<double*>(s_W_lds + nt_lds);
    __shared__ u64 s_bound[SH_MAX_WORLD];                               // inclusive prefix of T_b at the end of every rank's tiles
    __shared__ uint32_t s_above[OWN_NW][SH_MAX_WORLD];                  // per wave: draws whose target lies above s_bound[r]
    __shared__ uint32_t s_wown[OWN_NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 Wd = (u64)R * OWN_ROUND;                                  // draws (and window entries) per super-chunk
    const u64* my_incl = incl_all + (u64)rank * nt_local;
    const u64* my_W = tW_all + (u64)rank * nt_local;
    const double* my_ratio = ratio_all + (u64)rank * nt_local;
    const u64 lo = rank ? incl_all[(u64)rank * nt_local - 1] : 0ull;    // uniform loads
    const u64 hi = my_incl[nt_local - 1];
    const u64 Q = incl_all[(u64)world * nt_local - 1];
    if (pa.head && blockIdx.x == 0 && tid == 0) {   // fold this normalisation into the filter scalars (k_shard_table's part otherwise)
        *pa.undo = *pa.scal;
        fold_scalars(pa.scal, pa.head->Q, pa.head->Q2, pa.S, pa.head->m, n_global, 0);
    }
    // multinomial: workgroup = super-chunk; lattice: the super-chunks that hold this rank's own range of draws, dealt round-robin
    // (LATTICE is a template parameter so that the multinomial form stays straight-line code: as a run-time loop it cost 15
    // more registers and 2 us)
    // multinomial: every super-chunk of the job, dealt round-robin over the (resident) workgroups
    int sc_first = 0, sc_last = pa.nsc - 1;
    if constexpr (LATTICE) mp_own_span(range, Wd, &sc_first, &sc_last);
    if constexpr (TABMODE == 1) {
        for (int b = tid; b < nt_local; b += OWN_THREADS) {
            s_incl_lds[b] = my_incl[b] - lo;
            s_W_lds[b] = my_W[b];
            s_ratio_lds[b] = my_ratio[b];
        }
    }
    // multinomial in a world of several ranks: s_bound holds the boundaries as thresholds on the uniforms (k_shard_table's kthr)
    const bool by_k = !LATTICE && kthr != nullptr;
    for (int r = tid; r < world; r += OWN_THREADS) s_bound[r] = by_k ? kthr[r] : incl_all[(u64)(r + 1) * nt_local - 1];
    const u64 klo = (by_k && rank) ? kthr[rank - 1] : 0ull, khi = by_k ? kthr[rank] : (1ull << 52);
    const bool count_all = scheme == 0 && world > 1;
    const uint32_t k32 = scheme == 1 ? mp_systematic_k32(rc, k0, k1) : 0u;
    uint32_t above_acc = 0u;   // lane r: draws of this wave above boundary r (r < world - 1), over all super-chunks of this workgroup
    u64 live_tot = 0;          // draws this workgroup enumerated (uniform)
    for (int sc = sc_first + (int)blockIdx.x; sc <= sc_last; sc += (int)gridDim.x) {   // workgroup-uniform
    const u64 g_base = (u64)sc * Wd;
    __syncthreads();   // s_bound and the LDS table are there (first trip); the previous super-chunk's LDS state is free
    {
        const u64 rest = n_global - g_base;
        live_tot += rest < Wd ? rest : Wd;
    }
    uint32_t run = 0u;         // own draws of the rounds so far (lattice: of the workgroup; multinomial: of this wave)
    uint32_t woff = 0u;        // multinomial: own draws of the waves before this one
    uint32_t own_sc = 0u;
    if constexpr (LATTICE) {
#pragma unroll 1
    for (int rr = 0; rr < R; ++rr) {
        const u64 i0 = g_base + (u64)rr * OWN_ROUND + 2u * (u64)tid;   // even
        const bool live0 = i0 < n_global, live1 = i0 + 1 < n_global;
        const u64 t0 = mp_target_lattice(scheme, i0, k32, rc, k0, k1, Q, n_global);
        const u64 t1 = mp_target_lattice(scheme, i0 + 1, k32, rc, k0, k1, Q, n_global);
        const bool mine0 = live0 && t0 > lo && t0 <= hi;
        const bool mine1 = live1 && t1 > lo && t1 <= hi;
        // compaction in draw order: (round, wave, lane, slot of the pair) ascending IS g ascending
        const u64 m0 = __ballot(mine0), m1 = __ballot(mine1);
        const uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1,
                             __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u))));
        if (lane == 0) s_wown[wave] = (uint32_t)(__popcll(m0) + __popcll(m1));
        __syncthreads();
        uint32_t wo = 0u, tot = 0u;
#pragma unroll
        for (int w = 0; w < OWN_NW; ++w) {
            const uint32_t c = s_wown[w];
            wo += (w < wave) ? c : 0u;
            tot += c;
        }
        if (mine0) s_tgt[run + wo + pre] = t0;
        if (mine1) s_tgt[run + wo + pre + (mine0 ? 1u : 0u)] = t1;
        run += tot;
        __syncthreads();
    }
    } else {
    // multinomial: a WAVE takes R * 128 consecutive draws of the super-chunk and compacts its own ones — the 52-bit uniforms; their
    // targets are worked out afterwards, for this rank's own draws only — into its segment of s_tgt: no barrier inside the
    // enumeration, which is all a rank does for the (world - 1) / world of the draws that are not its own
    u64* seg = s_tgt + (size_t)wave * ((size_t)R * 128u);
    const u64 w_base = g_base + (u64)wave * ((u64)R * 128u);
#pragma unroll 1
    for (int rr = 0; rr < R; ++rr) {
        const u64 i0 = w_base + (u64)rr * 128u + 2u * (u64)lane;   // even
        const bool live0 = i0 < n_global, live1 = i0 + 1 < n_global;
        const mp_u64x2 blk = mp_resample_block(i0 >> 1, rc, (uint32_t)MP_DOM_RESAMPLE, k0, k1);
        const u64 t0 = mp_u52(blk.a), t1 = mp_u52(blk.b);
        const bool mine0 = live0 && t0 >= klo && t0 < khi;
        const bool mine1 = live1 && t1 >= klo && t1 < khi;
        if (count_all) {
            for (int r = 0; r + 1 < world; ++r) {
                const u64 Kr = s_bound[r];
                const uint32_t c = (uint32_t)__popcll(__ballot(live0 && t0 >= Kr)) + (uint32_t)__popcll(__ballot(live1 && t1 >= Kr));
                above_acc += (lane == r) ? c : 0u;
            }
        }
        const u64 m0 = __ballot(mine0), m1 = __ballot(mine1);
        const uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1,
                             __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u))));
        if (mine0) seg[run + pre] = t0;
        if (mine1) seg[run + pre + (mine0 ? 1u : 0u)] = t1;
        run += (uint32_t)(__popcll(m0) + __popcll(m1));
    }
    if (lane == 0) s_wown[wave] = run;
    __syncthreads();
    uint32_t tot = 0u;
#pragma unroll
    for (int w = 0; w < OWN_NW; ++w) {
        const uint32_t c = s_wown[w];
        woff += (w < wave) ? c : 0u;
        tot += c;
    }
    own_sc = tot;
    }
    const uint32_t own = LATTICE ? run : own_sc;   // the super-chunk's count (`run`: the workgroup's under a lattice scheme, this wave's otherwise)
    if (tid == 0) pa.sccnt[sc] = own;   // this super-chunk's counts, for k_shard_own_plan
    // ---- the compacted own draws, in draw order: target -> tile -> guide cell -> {tile-local target, start row}, written at the
    // draw's rank inside the super-chunk (k_shard_own_place, or in a world of one the next k_propagate itself, takes it from
    // there: the row lookups run under that kernel's arithmetic, as in the single filter) ----
    const u64 span = hi - lo;
    auto tgt_of = [&](u64 v) { return LATTICE ? v : mp_target(v, Q); };   // compacted value -> target
    const u64* t_incl = TABMODE == 1 ? s_incl_lds : my_incl;
    const u64* t_W = TABMODE == 1 ? s_W_lds : my_W;
    const double* t_ratio = TABMODE == 1 ? s_ratio_lds : my_ratio;
    const double nt_over_span = (double)nt_local / (double)span;   // own > 0 implies span > 0
    // lattice: the workgroup's entries dealt over all threads; multinomial: a wave's own entries over its lanes, placed behind
    // those of the waves before it (draw order: the waves' sub-ranges ascend)
    const u64 wbase = (u64)sc * Wd + (LATTICE ? 0u : woff);
    const u64* ent = LATTICE ? s_tgt : s_tgt + (size_t)wave * ((size_t)R * 128u);
    const uint32_t n_ent = LATTICE ? own : run;
    const uint32_t stride = LATTICE ? (uint32_t)OWN_ROUND : 128u;
    const uint32_t me = LATTICE ? (uint32_t)tid : (uint32_t)lane;
    for (uint32_t jb = 0; jb < n_ent; jb += stride) {   // trip count uniform over the workgroup (lattice) / the wave (multinomial)
        uint32_t j[2], gslot[2], tile_of[2], j0[2];
        u64 lt[2];
        bool act[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            j[q] = jb + 2u * me + (uint32_t)q;
            act[q] = j[q] < n_ent;
            const u64 t = act[q] ? tgt_of(ent[j[q]]) : lo + 1ull;
            const u64 trel = t - lo;
            mp_locate_own(t_incl, t_W, t_ratio, (uint32_t)nt_local, TABMODE == 1 ? trel : t, trel, TABMODE == 1 ? 0ull : lo, nt_over_span,
                          &tile_of[q], &lt[q], &gslot[q]);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) j0[q] = mp_guide_row(guide[gslot[q]]);
        uint32_t srow[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const u64 tbase = (u64)tile_of[q] * TILE;
            const uint32_t tlen = (uint32_t)((n - tbase) < (u64)TILE ? (n - tbase) : (u64)TILE);
            const uint32_t jj = j0[q] > tlen - 1 ? tlen - 1 : j0[q];
            srow[q] = (uint32_t)tbase + jj;          // row where the forward scan starts
        }
        // a thread's two entries are adjacent: one 16-byte and one 8-byte store where the pair is aligned (lattice: wbase and j[0]
        // are even; multinomial: the waves before may have left an odd count)
        if (act[1] && (((wbase + j[0]) & 1ull) == 0ull)) {
            mp_u64v2 v2; v2.x = lt[0]; v2.y = lt[1];
            *reinterpret_cast<mp_u64v2*>(win_lt + wbase + j[0]) = v2;
            *reinterpret_cast<uint2*>(win_row + wbase + j[0]) = make_uint2(srow[0], srow[1]);
        } else {
            if (act[0]) { win_lt[wbase + j[0]] = lt[0]; win_row[wbase + j[0]] = srow[0]; }
            if (act[1]) { win_lt[wbase + j[1]] = lt[1]; win_row[wbase + j[1]] = srow[1]; }
        }
    }
    __syncthreads();   // the next super-chunk of this workgroup reuses the LDS state
    }
    // multinomial in a world of several ranks: offspring per rank over this workgroup's super-chunks -> the job's counts
    // (k_shard_table zeroed them; k_shard_own_plan reads them): world atomics per workgroup instead of a [super-chunk][rank]
    // table for the plan to sum
    if (count_all) {
        if (lane < world) s_above[wave][lane] = above_acc;
        __syncthreads();
        if (tid < world) {
            u64 a_prev = 0, a_me = 0;
#pragma unroll
            for (int w = 0; w < OWN_NW; ++w) {
                a_prev += tid ? s_above[w][tid - 1] : 0u;
                a_me += (tid + 1 < world) ? s_above[w][tid] : 0u;
            }
            if (tid == 0) a_prev = live_tot;
            const u64 c = a_prev - a_me;
            if (c) __hip_atomic_fetch_add(pa.c_all + tid, (unsigned long long)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}


// After the plan (worlds of more than one rank): the own draws of every super-chunk go where their offspring's place says —
// offspring p = base[super-chunk] + rank in draw order:
//   p < n   kept: {target, start row} into the draws' slot-order arrays at p — the next k_propagate (or k_resolve_slots) looks
//           the parent up, exactly as after an unsharded resample;
//   p >= n  surplus: looked up HERE (O(sqrt N) of them) and its row {state, parent's global id} written into the send buffer
//           where the plan says;
// and the slots this rank could not fill itself, [c_me, n), get MP_DRAW_RECV | the index of the row that will arrive for them.
template <bool LATTICE>
__global__ __launch_bounds__(256) void k_shard_own_place(u64 n, u64 slot_offset, int D, int world, int rank, int R, int nsc, u64 cap,
                                                         const u64* __restrict__ win_lt, const uint32_t* __restrict__ win_row,
                                                         const uint32_t* __restrict__ sccnt, const uint32_t* __restrict__ base, const mp_cx* __restrict__ cx,
                                                         const double* __restrict__ x, const mp_owned_plan* __restrict__ plan,
                                                         const unsigned long long* __restrict__ c_all, double* __restrict__ send,
                                                         u64* __restrict__ dfr_lt, uint32_t* __restrict__ dfr_row, const mp_own_range* __restrict__ range) {
    const int tid = threadIdx.x;
    const u64 Wd = (u64)R * OWN_ROUND;
    int sc_first = 0, sc_last = nsc - 1;
    if constexpr (LATTICE) mp_own_span(range, Wd, &sc_first, &sc_last);   // only the own range's super-chunks hold entries
    const u64 PS_me = plan->PS[rank];
    for (int sc = sc_first + (int)blockIdx.x; sc <= sc_last; sc += (int)gridDim.x) {   // workgroup-uniform
        const uint32_t cnt = sccnt[sc];
        const u64 bk = base[sc];
        const u64 wbase = (u64)sc * Wd;
        for (uint32_t j = (uint32_t)tid; j < cnt; j += 256u) {
            const u64 p = bk + j;
            const u64 lt = win_lt[wbase + j];
            const uint32_t row0 = win_row[wbase + j];
            if (p < n) {
                dfr_lt[p] = lt;
                dfr_row[p] = row0;
                continue;
            }
            uint32_t par;
            double x0;
            mp_resolve_draw<true>(cx, n, lt, row0, &par, &x0);
            const u64 u = PS_me + (p - n);
            int s2 = 0;
            while (s2 + 1 < world && !(plan->D[s2] && u < plan->PD[s2] + plan->D[s2])) ++s2;
            double* dst = nullptr;
            if (cap) {
                const u64 first = PS_me > plan->PD[s2] ? PS_me : plan->PD[s2];
                const u64 jj = u - first;
                // jj >= cap: k_shard_own_plan has flagged it, nothing of this attempt is committed
                if (jj < cap) dst = send + ((u64)s2 * cap + jj) * (u64)(D + 1);
            } else {
                dst = send + (u - PS_me) * (u64)(D + 1);
            }
            if (dst) {
                if (D == 1) {
                    dst[0] = x0;
                } else {
                    for (int d = 0; d < D; ++d) dst[d] = x[(u64)par * D + d];
                }
                dst[D] = (double)(slot_offset + par);
            }
        }
    }
    // slots this rank could not fill itself: where in the receive buffer their rows will arrive
    const u64 c_me = c_all[rank];
    const u64 PD_me = plan->PD[rank], D_me = plan->D[rank];
    for (u64 k = (u64)blockIdx.x * 256u + tid; k < D_me; k += (u64)gridDim.x * 256u) {
        u64 idx = k;
        if (cap) {
            const u64 u = PD_me + k;
            int r = 0;
            while (r + 1 < world && !(plan->S[r] && u < plan->PS[r] + plan->S[r])) ++r;
            const u64 first = plan->PS[r] > PD_me ? plan->PS[r] : PD_me;
            const u64 jj = u - first;
            idx = jj >= cap ? 0ull : (u64)r * cap + jj;
        }
        dfr_lt[c_me + k] = 0ull;
        dfr_row[c_me + k] = MP_DRAW_RECV | (uint32_t)idx;
    }
}
// ---------------------------------------------------------------------------------------------
// Self-drawn owner-keeps resample (lattice schemes; split multinomial).  Offspring p of this rank — p = 0 .. c_me - 1, in the order
// of its draws — has its target in CLOSED FORM:
//   lattice             draw g = g_lo + p of the job's lattice (k_shard_table found the rank's range [g_lo, g_hi): mp_lattice_ranges)
//   split multinomial   uniform p of the rank's own stream (Philox block p >> 1, word 3 = the rank) scaled to the rank's own share
//                       (lo, hi] of the job's fixed-point mass (mp_binomial.h; k_shard_table drew the counts c_r)
// so nothing is enumerated, compacted or scanned: kept offspring p < min(c_me, n) IS slot p.  The kept draws are made by the next
// k_propagate itself (its SHD form, mp_pf_kernels.h) or, for models whose propagate kernel cannot draw and for readers that come
// before the next step, by k_shard_self_draw below; k_shard_self_place looks up the O(sqrt N) surplus offspring p >= n for the send
// buffer and flags the deficit slots [c_me, n) with the rows that will arrive for them.  Same offspring, same slots, same rows as
// k_shard_own_draw / _plan / _place produce for the lattice schemes (tests/test_gpu_owned.py runs both).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ mp_u64x2 mp_split_block(u64 j_pair, uint32_t rc, int rank, uint32_t k0, uint32_t k1) {
    return mp_philox4x32_10((uint32_t)j_pair, rc, (uint32_t)MP_DOM_RESAMPLE << 16, (uint32_t)rank, k0, k1);
}
// target of own offspring p relative to the rank's share: in [1, hi - lo]
template <bool LATTICE>
__device__ __forceinline__ u64 mp_self_trel(int scheme, u64 p, u64 g_lo, u64 lo, u64 span, u64 Q, u64 n_global, uint32_t k32, int rank, uint32_t rc,
                                            uint32_t k0, uint32_t k1) {
    if constexpr (LATTICE) {
        return mp_target_lattice(scheme, g_lo + p, k32, rc, k0, k1, Q, n_global) - lo;
    } else {
        const mp_u64x2 blk = mp_split_block(p >> 1, rc, rank, k0, k1);
        return mp_target(mp_u52((p & 1ull) ? blk.b : blk.a), span);
    }
}
constexpr int SELF_THREADS = 512;   // a thread owns two ADJACENT slots of a trip
template <int TABMODE, bool LATTICE>   // TABMODE 1: the rank's slice of the tile table copied to LDS (rebased); 2: probed where it lies
__global__ __launch_bounds__(SELF_THREADS) void k_shard_self_draw(u64 n, u64 n_global, uint32_t k0, uint32_t k1, uint32_t rc, int scheme,
                                                                  const u64* __restrict__ incl_sl, const u64* __restrict__ W_sl,
                                                                  const double* __restrict__ ratio_sl, int nt_local, int world, int rank,
                                                                  const unsigned short* __restrict__ guide, const mp_own_range* __restrict__ range,
                                                                  u64* __restrict__ dfr_lt, uint32_t* __restrict__ dfr_row) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* s_incl_lds = reinterpret_cast<u64*>(smem);
    u64* s_W_lds = s_incl_lds + (TABMODE == 1 ? nt_local : 0);
    double* s_ratio_lds = reinterpret_cast<double*>(s_W_lds + (TABMODE == 1 ? nt_local : 0));
    const int tid = threadIdx.x;
    const u64 lo = rank ? incl_sl[-1] : 0ull;   // uniform loads
    const u64 hi = incl_sl[nt_local - 1];
    const u64 Q = incl_sl[(u64)(world - rank) * nt_local - 1];
    const u64 g_lo = range->g_lo, c_me = range->g_hi - range->g_lo;
    const u64 keep = c_me < n ? c_me : n;
    if constexpr (TABMODE == 1) {
        for (int b = tid; b < nt_local; b += SELF_THREADS) {
            s_incl_lds[b] = incl_sl[b] - lo;
            s_W_lds[b] = W_sl[b];
            s_ratio_lds[b] = ratio_sl[b];
        }
        __syncthreads();
    }
    const u64* t_incl = TABMODE == 1 ? s_incl_lds : incl_sl;
    const u64* t_W = TABMODE == 1 ? s_W_lds : W_sl;
    const double* t_ratio = TABMODE == 1 ? s_ratio_lds : ratio_sl;
    const u64 span = hi - lo;
    const double nt_over_span = (double)nt_local / (double)span;   // keep > 0 implies span > 0
    const uint32_t k32 = (LATTICE && scheme == 1) ? mp_systematic_k32(rc, k0, k1) : 0u;
    for (u64 p0 = ((u64)blockIdx.x * SELF_THREADS + tid) * 2ull; p0 < keep; p0 += (u64)gridDim.x * SELF_THREADS * 2ull) {
        uint32_t gslot[2], tile_of[2], j0[2], srow[2];
        u64 lt[2];
        bool act[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            act[q] = p0 + q < keep;
            const u64 trel = act[q] ? mp_self_trel<LATTICE>(scheme, p0 + q, g_lo, lo, span, Q, n_global, k32, rank, rc, k0, k1) : 1ull;
            mp_locate_own(t_incl, t_W, t_ratio, (uint32_t)nt_local, TABMODE == 1 ? trel : trel + lo, trel, TABMODE == 1 ? 0ull : lo, nt_over_span,
                          &tile_of[q], &lt[q], &gslot[q]);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) j0[q] = mp_guide_row(guide[gslot[q]]);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const u64 tbase = (u64)tile_of[q] * TILE;
            const uint32_t tlen = (uint32_t)((n - tbase) < (u64)TILE ? (n - tbase) : (u64)TILE);
            srow[q] = (uint32_t)tbase + (j0[q] > tlen - 1 ? tlen - 1 : j0[q]);
        }
        if (act[1]) {   // (p0 is even: aligned pair)
            mp_u64v2 v2; v2.x = lt[0]; v2.y = lt[1];
            *reinterpret_cast<mp_u64v2*>(dfr_lt + p0) = v2;
            *reinterpret_cast<uint2*>(dfr_row + p0) = make_uint2(srow[0], srow[1]);
        } else {
            dfr_lt[p0] = lt[0]; dfr_row[p0] = srow[0];
        }
    }
}
// the surplus offspring p = n .. c_me - 1 (looked up here: row {state, parent's global id} into the send buffer where the plan says)
// and the deficit slots [c_me, n) (MP_DRAW_RECV | the index of the row that will arrive); threads t0, t0 + stride, ... of whoever runs it
struct mp_self_place_args {
    u64 n, slot_offset, cap;
    int D;
    const u64* incl_sl;            // this rank's slice of the job's table (k_shard_table's arrays)
    const u64* W_sl;
    const double* ratio_sl;
    const unsigned short* guide;
    const mp_cx* cx;
    const double* x;
    double* send;
    u64* dfr_lt;
    uint32_t* dfr_row;
};
template <bool LATTICE>
__device__ __forceinline__ void mp_self_place(const mp_self_place_args& a, const mp_owned_plan* plan, int scheme, int world, int rank, int nt_local,
                                              u64 n_global, uint32_t k0, uint32_t k1, uint32_t rc, u64 lo, u64 hi, u64 Q, u64 g_lo, u64 c_me, u64 t0,
                                              u64 stride) {
    const u64 n = a.n;
    const int D = a.D;
    const u64 span = hi - lo;
    const double nt_over_span = (double)nt_local / (double)span;
    const uint32_t k32 = (LATTICE && scheme == 1) ? mp_systematic_k32(rc, k0, k1) : 0u;
    const u64 PS_me = plan->PS[rank];
    for (u64 p = n + t0; p < c_me; p += stride) {
        const u64 trel = mp_self_trel<LATTICE>(scheme, p, g_lo, lo, span, Q, n_global, k32, rank, rc, k0, k1);
        uint32_t tile, gslot;
        u64 lt;
        mp_locate_own(a.incl_sl, a.W_sl, a.ratio_sl, (uint32_t)nt_local, trel + lo, trel, lo, nt_over_span, &tile, &lt, &gslot);
        const uint32_t j0 = mp_guide_row(a.guide[gslot]);
        const u64 tbase = (u64)tile * TILE;
        const uint32_t tlen = (uint32_t)((n - tbase) < (u64)TILE ? (n - tbase) : (u64)TILE);
        const uint32_t row0 = (uint32_t)tbase + (j0 > tlen - 1 ? tlen - 1 : j0);
        uint32_t par;
        double x0;
        mp_resolve_draw<true>(a.cx, n, lt, row0, &par, &x0);
        const u64 u = PS_me + (p - n);
        int s2 = 0;
        while (s2 + 1 < world && !(plan->D[s2] && u < plan->PD[s2] + plan->D[s2])) ++s2;
        double* dst = nullptr;
        if (a.cap) {
            const u64 first = PS_me > plan->PD[s2] ? PS_me : plan->PD[s2];
            const u64 jj = u - first;
            if (jj < a.cap) dst = a.send + ((u64)s2 * a.cap + jj) * (u64)(D + 1);   // (jj >= cap: the plan has flagged it, nothing of this attempt is committed)
        } else {
            dst = a.send + (u - PS_me) * (u64)(D + 1);
        }
        if (dst) {
            if (D == 1) dst[0] = x0;
            else
                for (int d = 0; d < D; ++d) dst[d] = a.x[(u64)par * D + d];
            dst[D] = (double)(a.slot_offset + par);
        }
    }
    const u64 PD_me = plan->PD[rank], D_me = plan->D[rank];
    for (u64 k = t0; k < D_me; k += stride) {
        u64 idx = k;
        if (a.cap) {
            const u64 u = PD_me + k;
            int r = 0;
            while (r + 1 < world && !(plan->S[r] && u < plan->PS[r] + plan->S[r])) ++r;
            const u64 first = plan->PS[r] > PD_me ? plan->PS[r] : PD_me;
            const u64 jj = u - first;
            idx = jj >= a.cap ? 0ull : (u64)r * a.cap + jj;
        }
        a.dfr_lt[c_me + k] = 0ull;
        a.dfr_row[c_me + k] = MP_DRAW_RECV | (uint32_t)idx;
    }
}
// a launch of its own (exact sizes: the host sized the send buffer from the counts first; models whose kept draws are a launch too)
template <bool LATTICE>
__global__ __launch_bounds__(256) void k_shard_self_place(mp_self_place_args a, u64 n_global, uint32_t k0, uint32_t k1, uint32_t rc, int scheme, int world,
                                                          int rank, int nt_local, const mp_own_range* __restrict__ range,
                                                          const mp_owned_plan* __restrict__ plan) {
    MP_STAMP(3, 0, 1);
    const u64 lo = rank ? a.incl_sl[-1] : 0ull;
    const u64 hi = a.incl_sl[nt_local - 1];
    const u64 Q = a.incl_sl[(u64)(world - rank) * nt_local - 1];
    mp_self_place<LATTICE>(a, plan, scheme, world, rank, nt_local, n_global, k0, k1, rc, lo, hi, Q, range->g_lo, range->g_hi - range->g_lo,
                           (u64)blockIdx.x * blockDim.x + threadIdx.x, (u64)gridDim.x * blockDim.x);
    MP_STAMP(3, 2, 1);
}
// ---------------------------------------------------------------------------------------------
// The job's tile table from the gathered tiles, and — by the table's LEADING workgroup, on its way out — everything of a resample
// that needs nothing but the table: the filter scalars of this normalisation (L, ESS, log-ML; a copy kept for the case that a
// fixed-capacity exchange overflows), the rank boundaries (multinomial: thresholds on the uniforms; lattice: every rank's range of
// draws; split multinomial: the offspring per rank), and for a self-drawn resample the exchange plan, the host's verdict word and
// the placement (surplus lookups into the send buffer, deficit slots flagged) — one launch where there were three.
// ---------------------------------------------------------------------------------------------
// the exchange plan of a self-drawn resample, by the workgroup that has just made the counts (s_c, LDS; *s_deg: degenerate weights):
// no round trip through memory and nothing serial over the ranks (pub->L follows from the lane that folds the scalars).  The plan stays in LDS (pl) for the placement
// that follows in the same launch, goes to memory for k_shard_self_place / the commit, and the host's part to its mapped page.
template <int THREADS>
__device__ __forceinline__ void mp_self_plan(const mp_own_plan_args& a, const u64* s_c, const int* s_deg, mp_owned_plan* pl, int* s_over) {
    const int tid = threadIdx.x, world = a.world;
    if (tid < 64) {   // (world <= 64: one wave, the WHOLE of it scans: lanes beyond `world` with zeros)
        const bool have = tid < world;
        const u64 c = have ? s_c[tid] : a.n;
        const u64 S = c > a.n ? c - a.n : 0ull, D = c < a.n ? a.n - c : 0ull;
        const u64 PS = wave_incl_scan_u64(S, tid) - S, PD = wave_incl_scan_u64(D, tid) - D;
        if (have) {
            pl->S[tid] = S; pl->D[tid] = D; pl->PS[tid] = PS; pl->PD[tid] = PD;
            a.plan_out->S[tid] = S; a.plan_out->D[tid] = D; a.plan_out->PS[tid] = PS; a.plan_out->PD[tid] = PD;
            mp_st_sys(&a.pub->counts[tid], (unsigned long long)c);
        }
        if (tid == 0) *s_over = 0;
    }
    __syncthreads();
    // every rank must reach the same verdict on "some pair needs more than cap rows" (the collective that follows is symmetric),
    // so every rank looks at every pair of the plan
    int over = 0;
    if (a.cap) {
        for (int pq = tid; pq < world * world; pq += THREADS) {
            const int r = pq / world, s2 = pq - r * world;
            const u64 a0 = pl->PS[r] > pl->PD[s2] ? pl->PS[r] : pl->PD[s2];
            const u64 e0 = pl->PS[r] + pl->S[r], e1 = pl->PD[s2] + pl->D[s2];
            const u64 a1 = e0 < e1 ? e0 : e1;
            if (a1 > a0 && a1 - a0 > a.cap) over = 1;
        }
    }
    if (over) atomicOr(s_over, 1);
    __syncthreads();
    if (tid == 0) {
        const int deg = *s_deg, ov = *s_over;
        mp_st_sys(&a.pub->degenerate, deg);
        mp_st_sys(&a.pub->overflow, ov);
        // what the host polls between resamples: one word, so it needs no ordering against the fields above (those are read
        // after the stream has drained: log total weight of a synchronous resample, counts of the exact-size repeat)
        mp_st_sys(&a.pub->verdict, (a.seq << 8) | (unsigned long long)((deg ? 2 : 0) | (ov ? 1 : 0)));
    }
}
struct mp_table_tail {
    unsigned long long* c_all;   // [world] offspring per rank (null: the slot-order route — table and scalars only)
    int scheme, rank;
    uint32_t k0, k1, rc;
    mp_own_range* range;
    u64* kthr;                   // owner-keeps multinomial (the window form): [world] thresholds on the 52-bit uniforms
    mp_own_plan_args plan;       // self-drawn resample: the plan from the counts alone
    int do_plan;
    mp_self_place_args place;    // and, with an equal-split capacity, the placement
    int do_place;
};
// by the leading workgroup, every thread; s_bound (if c_all) is written and a barrier has passed.  lo / hi: the leader's own
// share (lo, hi] of the job's mass — the leader quantised `rank`'s tiles itself, so the placement reads nothing but its own stores.
// su, sv: mp_split_uniforms (split multinomial), made while the table was still on its way.
template <int THREADS>
__device__ __forceinline__ void mp_table_leader(const mp_table_tail& t, const u64* s_bound, u64* s_G, u64 Q, u64 Q2, double m, int world, int nt_local,
                                                int S, u64 n_global, mp_dev_scalars* scal, mp_dev_scalars* undo, u64 lo, u64 hi, double su, double sv, int prev_deg) {
    __shared__ u64 s_heap[2 * SH_MAX_WORLD];
    __shared__ u64 s_c[SH_MAX_WORLD];
    __shared__ mp_owned_plan s_pl;
    __shared__ int s_deg, s_over;
    const int tid = threadIdx.x;
    MP_STAMP(2, 4, 1);
    // The fold of this normalisation into the filter scalars (a logarithm, a dozen loads and stores: 1.1 us by one lane) is nobody's
    // input in this launch except for the sticky "degenerate" flag, which is a comparison: one lane of a wave with nothing else to do
    // (prev_deg: its early load of the flag) folds AFTER the last barrier, while the others place.
    const bool folder = tid == THREADS - 64;
    if (folder) s_deg = (prev_deg || !(m > MP_NEG_INF) || !(m < MP_INF) || Q == 0ull) ? 1 : 0;
    if (!t.c_all || !t.do_plan) {
        if (folder) {
            *undo = *scal;   // a fixed-capacity exchange that overflows puts these back
            fold_scalars(scal, Q, Q2, S, m, n_global, 0);
        }
    }
    if (!t.c_all) return;   // (workgroup-uniform, like every condition below)
    // Owner-keeps exchange, multinomial: the rank boundaries as thresholds on the 52-bit uniforms themselves.  target(k) =
    // max(1, ceil(k Q / 2^52)) is non-decreasing in k, so target(k) > B  <=>  k >= K(B) = the smallest such k (= floor(B 2^52 / Q)
    // + 1, found from a double estimate and corrected with the target function itself): the draw kernel then tells whose a
    // draw is with 64-bit compares and pays the 128-bit target only for its own.
    if (t.kthr && t.scheme == 0) {
        if (tid < world) {
            const u64 B = s_bound[tid];
            const u64 top = 1ull << 52;
            double est = (double)B * 4503599627370496.0 / (double)Q;   // Q == 0: NaN / inf -> clamped below
            u64 K = (est >= 4503599627370496.0) ? top : ((est >= 8.) ? (u64)est - 8ull : 0ull);
            if (!(est == est)) K = 0ull;
            while (K > 0ull && mp_target(K - 1ull, Q) > B) --K;          // (a few steps at most: the estimate is within a few units)
            while (K < top && mp_target(K, Q) <= B) ++K;
            t.kthr[tid] = K;
            if (world > 1) t.c_all[tid] = 0ull;   // the draw kernel's workgroups add the offspring per rank up
        }
    }
    u64 g_lo = 0ull;
    if (t.scheme == MP_SCHEME_SPLIT) {
        __shared__ mp_split_lds s_split;
        mp_split_counts(s_bound, world, n_global, t.rc, t.k0, t.k1, s_heap, su, sv, &s_split);
        if (tid < world) {
            const u64 c = s_heap[(1 << mp_split_levels(world)) + tid];
            s_c[tid] = c;
            t.c_all[tid] = c;
            if (tid == t.rank) { t.range->g_lo = 0ull; t.range->g_hi = c; }
        }
    }
    if (t.scheme == 1 || t.scheme == 2) {
        mp_lattice_ranges(s_bound, s_G, world, t.scheme, Q, n_global, t.rc, t.k0, t.k1);
        if (tid < world) {
            const u64 g0 = tid ? s_G[tid - 1] : 0ull;
            s_c[tid] = s_G[tid] - g0;
            t.c_all[tid] = s_G[tid] - g0;
            if (tid == t.rank) { t.range->g_lo = g0; t.range->g_hi = s_G[tid]; }
        }
        g_lo = t.rank ? s_G[t.rank - 1] : 0ull;
    }
    MP_STAMP(2, 5, 1);
    if (!t.do_plan) return;
    __syncthreads();   // s_c, s_deg
    mp_self_plan<THREADS>(t.plan, s_c, &s_deg, &s_pl, &s_over);
    MP_STAMP(2, 6, 1);
    if (folder) {
        *undo = *scal;
        fold_scalars(scal, Q, Q2, S, m, n_global, 0);
        mp_st_sys(&t.plan.pub->L, scal->L);   // (read by a synchronous commit, after the stream has drained)
        return;
    }
    if (!t.do_place || s_deg) return;   // (degenerate weights: the commit refuses, nothing to place)
    const u64 c_me = s_c[t.rank];
    // (the folding lane has left: the other THREADS - 1 share the placement out among themselves)
    const u64 t0 = (u64)(tid < THREADS - 64 ? tid : tid - 1), stride = (u64)(THREADS - 1);
    if (t.scheme == MP_SCHEME_SPLIT)
        mp_self_place<false>(t.place, &s_pl, t.scheme, world, t.rank, nt_local, n_global, t.k0, t.k1, t.rc, lo, hi, Q, g_lo, c_me, t0, stride);
    else
        mp_self_place<true>(t.place, &s_pl, t.scheme, world, t.rank, nt_local, n_global, t.k0, t.k1, t.rc, lo, hi, Q, g_lo, c_me, t0, stride);
    MP_STAMP(2, 7, 1);
}
// One workgroup builds the whole table (jobs of up to 2048 tiles, and a world of one whose table no level-0 launch left)
template <int PER_MAX>   // tiles per thread at most (registers): 2 covers the jobs this kernel is chosen for, SHT_PER every handle
__global__ __launch_bounds__(SHT_THREADS) void k_shard_table(const u64* __restrict__ packed, int world, int nt_local, int S, u64 n_global,
                                                             double* __restrict__ tm, u64* __restrict__ tW, u64* __restrict__ tW2,
                                                             u64* __restrict__ incl_all, double* __restrict__ ratio_all,
                                                             long long* __restrict__ zero_counts, mp_dev_scalars* scal, mp_dev_scalars* undo,
                                                             mp_table_tail tail) {
    __shared__ double s_red[SHT_THREADS / 64];
    __shared__ u64 s_wtot[SHT_THREADS / 64];
    __shared__ u64 s_wtot2[SHT_THREADS / 64];
    __shared__ u64 s_bound[SH_MAX_WORLD];
    __shared__ u64 s_G[SH_MAX_WORLD];
    const int nt = world * nt_local;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < SH_MAX_KEYS) zero_counts[tid] = 0;
    const int prev_deg = tid == SHT_THREADS - 64 ? scal->degenerate : 0;   // (the lane that folds: mp_table_leader)
    double su = 0., sv = 0.;
    if (tail.c_all && tail.scheme == MP_SCHEME_SPLIT) mp_split_uniforms(tail.rc, tail.k0, tail.k1, &su, &sv);
    // thread t owns tiles t * per .. t * per + per - 1 (consecutive, so that a thread-local running sum is a prefix); each
    // tile is read once, kept in registers, and its unpacked copy written for the kernels that want plain arrays
    const int per = (nt + SHT_THREADS - 1) / SHT_THREADS;   // <= PER_MAX (the host picks the instance)
    const int b0 = tid * per;
    double mb[PER_MAX];
    u64 Wb[PER_MAX], W2b[PER_MAX];
    double m = MP_NEG_INF;
#pragma unroll
    for (int j = 0; j < PER_MAX; ++j) {
        const int i = b0 + j;
        mb[j] = MP_NEG_INF; Wb[j] = 0; W2b[j] = 0;
        if (j < per && i < nt) {
            const int r = i / nt_local, b = i - r * nt_local;
            const u64* base = packed + (u64)r * 3 * nt_local;
            mb[j] = mp_u2f(base[b]);
            Wb[j] = base[nt_local + b];
            W2b[j] = base[2 * nt_local + b];
            tm[i] = mb[j]; tW[i] = Wb[j]; tW2[i] = W2b[j];
            m = fmax(m, mb[j]);
        }
    }
    m = wave_max(m);
    if (lane == 0) s_red[wave] = m;
    __syncthreads();
    m = s_red[0];
#pragma unroll
    for (int w = 1; w < SHT_THREADS / 64; ++w) m = fmax(m, s_red[w]);
    // level 1 exactly as block_tile_table / block_sum_T2 state it: T_b = rint(W_b exp(m_b - m) 2^(S-51)), T2_b likewise with
    // exp(2 (m_b - m)); integer sums, so the order of the reduction is immaterial
    const bool ok = (m > MP_NEG_INF) && (m < MP_INF);
    const double sc = mp_u2f((u64)(1023 + S - FIX_BITS) << 52);
    u64 pre[PER_MAX];
    double ratio[PER_MAX];
    u64 run = 0, run2 = 0;
#pragma unroll
    for (int j = 0; j < PER_MAX; ++j) {
        const int i = b0 + j;
        pre[j] = 0;
        if (j < per && i < nt) {
            const double f = ok ? mp_exp(mb[j] - m) : 0.;
            const double f2 = ok ? mp_exp(2. * (mb[j] - m)) : 0.;
            const u64 T = mp_quantize((double)Wb[j] * f * sc, 1.0);
            run += T;
            run2 += mp_quantize((double)W2b[j] * f2 * sc, 1.0);
            pre[j] = run;
            ratio[j] = (double)Wb[j] / (double)T;   // of mp_local_target; never used for a tile with T = 0 (no target lands in it)
        }
    }
    const u64 incl = wave_incl_scan_u64(run, lane);
    const u64 tot2 = wave_sum_u64(run2);
    if (lane == 63) s_wtot[wave] = incl;
    if (lane == 0) s_wtot2[wave] = tot2;
    __syncthreads();
    u64 woff = 0, Q = 0, Q2 = 0;
#pragma unroll
    for (int w = 0; w < SHT_THREADS / 64; ++w) {
        if (w < wave) woff += s_wtot[w];
        Q += s_wtot[w];
        Q2 += s_wtot2[w];
    }
    const u64 off = woff + (incl - run);
#pragma unroll
    for (int j = 0; j < PER_MAX; ++j) {
        const int i = b0 + j;
        if (j < per && i < nt) { incl_all[i] = off + pre[j]; ratio_all[i] = ratio[j]; }
    }
    if (tail.c_all) {
#pragma unroll
        for (int j = 0; j < PER_MAX; ++j) {
            const int i = b0 + j;
            if (j < per && i < nt && (i + 1) % nt_local == 0) s_bound[i / nt_local] = off + pre[j];
        }
        __syncthreads();   // (also: this workgroup's stores of the table are out — the placement reads them)
    }
    const u64 lo = (tail.c_all && tail.rank) ? s_bound[tail.rank - 1] : 0ull;
    const u64 hi = tail.c_all ? s_bound[tail.rank] : 0ull;
    mp_table_leader<SHT_THREADS>(tail, s_bound, s_G, Q, Q2, m, world, nt_local, S, n_global, scal, undo, lo, hi, su, sv, prev_deg);
}
// The same table by `world` workgroups (worlds of more than one rank): workgroup r takes rank r's nt_local tiles, so the
// serial part no longer grows with the job (16.9 us for 8 x 512 tiles by one workgroup, profiles/r03/route_scale.txt).
// Every workgroup finds the job's maximum itself (world x nt_local loads), quantises its own tiles, publishes its two sums
// {sum T, sum T2} and a ticket (agent scope), waits until all `world` tickets of this launch are there — the workgroups are
// co-resident: world <= 64 of them on 256 CUs — and adds the sums of the ranks before it to its local prefix.  The workgroup of
// THIS rank's tiles is the leader: what it goes on to read (its slice of the table, for the placement) it wrote itself.
// `ticket` counts up by `world` per launch (never reset): this launch waits for `ticket_target`.
struct mp_tab_part {
    u64 Q, Q2;
};
template <int PER_MAX>   // tiles of a rank per thread at most: 1 up to 1024 tiles per rank (2^21 particles), SHT_PER beyond
__global__ __launch_bounds__(SHT_THREADS) void k_shard_table_mw(const u64* __restrict__ packed, int world, int nt_local, int S, u64 n_global,
                                                                double* __restrict__ tm, u64* __restrict__ tW, u64* __restrict__ tW2,
                                                                u64* __restrict__ incl_all, double* __restrict__ ratio_all,
                                                                long long* __restrict__ zero_counts, mp_dev_scalars* scal, mp_dev_scalars* undo,
                                                                mp_tab_part* __restrict__ part, unsigned int* __restrict__ ticket,
                                                                unsigned int ticket_target, mp_table_tail tail) {
    __shared__ double s_red[SHT_THREADS / 64];
    __shared__ u64 s_wtot[SHT_THREADS / 64];
    __shared__ u64 s_wtot2[SHT_THREADS / 64];
    __shared__ u64 s_bound[SH_MAX_WORLD];
    __shared__ u64 s_G[SH_MAX_WORLD];
    __shared__ u64 s_off, s_Q, s_Q2;
    const int nt = world * nt_local;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int me = (int)blockIdx.x;   // the rank whose tiles this workgroup quantises
    const bool leader = me == tail.rank;
    MP_STAMP(2, 0, 1);
    if (leader && tid < SH_MAX_KEYS) zero_counts[tid] = 0;
    const int prev_deg = (leader && tid == SHT_THREADS - 64) ? scal->degenerate : 0;   // (the lane that folds: mp_table_leader)
    // the job's maximum: every tile maximum of the gathered buffer
    double m = MP_NEG_INF;
    for (int i = tid; i < nt; i += SHT_THREADS) {
        const int r = i / nt_local, b = i - r * nt_local;
        m = fmax(m, mp_u2f(packed[(u64)r * 3 * nt_local + b]));
    }
    // this rank's tiles: thread t owns tiles t * per .. t * per + per - 1 of them
    const int per = (nt_local + SHT_THREADS - 1) / SHT_THREADS;   // <= PER_MAX (the host picks the instance)
    const int b0 = tid * per;
    const u64* base = packed + (u64)me * 3 * nt_local;
    double mb[PER_MAX];
    u64 Wb[PER_MAX], W2b[PER_MAX];
#pragma unroll
    for (int j = 0; j < PER_MAX; ++j) {
        const int b = b0 + j;
        mb[j] = MP_NEG_INF; Wb[j] = 0; W2b[j] = 0;
        if (j < per && b < nt_local) {
            mb[j] = mp_u2f(base[b]);
            Wb[j] = base[nt_local + b];
            W2b[j] = base[2 * nt_local + b];
            const int i = me * nt_local + b;
            tm[i] = mb[j]; tW[i] = Wb[j]; tW2[i] = W2b[j];
        }
    }
    m = wave_max(m);
    if (lane == 0) s_red[wave] = m;
    __syncthreads();
    MP_STAMP(2, 1, 1);
    m = s_red[0];
#pragma unroll
    for (int w = 1; w < SHT_THREADS / 64; ++w) m = fmax(m, s_red[w]);
    const bool ok = (m > MP_NEG_INF) && (m < MP_INF);
    const double sc = mp_u2f((u64)(1023 + S - FIX_BITS) << 52);
    u64 pre[PER_MAX];
    double ratio[PER_MAX];
    u64 run = 0, run2 = 0;
#pragma unroll
    for (int j = 0; j < PER_MAX; ++j) {
        const int b = b0 + j;
        pre[j] = 0;
        if (j < per && b < nt_local) {
            const double f = ok ? mp_exp(mb[j] - m) : 0.;
            const double f2 = ok ? mp_exp(2. * (mb[j] - m)) : 0.;
            const u64 T = mp_quantize((double)Wb[j] * f * sc, 1.0);
            run += T;
            run2 += mp_quantize((double)W2b[j] * f2 * sc, 1.0);
            pre[j] = run;
            ratio[j] = (double)Wb[j] / (double)T;
        }
    }
    const u64 incl = wave_incl_scan_u64(run, lane);
    const u64 tot2 = wave_sum_u64(run2);
    if (lane == 63) s_wtot[wave] = incl;
    if (lane == 0) s_wtot2[wave] = tot2;
    __syncthreads();
    u64 woff = 0, Qr = 0, Q2r = 0;
#pragma unroll
    for (int w = 0; w < SHT_THREADS / 64; ++w) {
        if (w < wave) woff += s_wtot[w];
        Qr += s_wtot[w];
        Q2r += s_wtot2[w];
    }
    MP_STAMP(2, 2, 1);
    // (the leader's lanes make the uniforms of the split counts while the ticket goes round)
    double su = 0., sv = 0.;
    if (leader && tail.c_all && tail.scheme == MP_SCHEME_SPLIT && tid != 0) mp_split_uniforms(tail.rc, tail.k0, tail.k1, &su, &sv);
    // publish this rank's sums, then wait for everybody's
    if (tid == 0) {
        mp_st_agent(&part[me].Q, Qr);
        mp_st_agent(&part[me].Q2, Q2r);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the sums are out before the ticket says so
        (void)__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while ((int)(__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - ticket_target) < 0) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    MP_STAMP(2, 3, 1);
    if (tid < 64) {
        // inclusive prefix over the ranks (world <= 64: one wave).  The WHOLE of wave 0 takes part, lanes beyond `world` with
        // zeros: wave_sum_u64 reads lane 63 of the scan, which an inactive lane would never have written
        const bool have = tid < world;
        const u64 q = have ? mp_ld_agent(&part[tid].Q) : 0ull, q2 = have ? mp_ld_agent(&part[tid].Q2) : 0ull;
        const u64 inc = wave_incl_scan_u64(q, lane);
        const u64 t2 = wave_sum_u64(q2);
        if (have) s_bound[tid] = inc;
        if (tid == me) s_off = inc - q;
        if (tid == world - 1) { s_Q = inc; s_Q2 = t2; }
    }
    __syncthreads();
    const u64 off = s_off + woff + (incl - run);
    const u64 Q = s_Q, Q2 = s_Q2;
#pragma unroll
    for (int j = 0; j < PER_MAX; ++j) {
        const int b = b0 + j;
        if (j < per && b < nt_local) { incl_all[me * nt_local + b] = off + pre[j]; ratio_all[me * nt_local + b] = ratio[j]; }
    }
    if (!leader) return;
    if (tail.do_place) __syncthreads();   // this workgroup's slice of the table is out: the placement reads it
    mp_table_leader<SHT_THREADS>(tail, s_bound, s_G, Q, Q2, m, world, nt_local, S, n_global, scal, undo, s_off, s_bound[me], su, sv, prev_deg);
}
// a world of one: the job's table is the one the last level-0 launch left; nobody has folded this normalisation into the filter
// scalars yet (the next k_propagate does on the way, unless the host wants the resample's value first)
__global__ void k_shard_solo_fold(const mp_tab_head* __restrict__ head, mp_dev_scalars* scal, int S, u64 n_global) {
    fold_scalars(scal, head->Q, head->Q2, S, head->m, n_global, 0);
}
// After the resolve: "somebody overflowed" (flags are only ever OR-ed atomically) and the scalars of this normalisation,
// where the host reads them after waiting for ev_resolved (host-mapped memory: no copy command, no stream sync).  A kernel of
// its own: a ticket per resolve workgroup (1300 same-address atomics) measured 7 us, this launch 4.
__global__ void k_shard_publish(int* overflow, const mp_dev_scalars* scal, mp_shard_pub* pub) {
    pub->L = scal->L;
    pub->degenerate = scal->degenerate;
    pub->overflow = atomicOr(overflow, 0);
}
// requester side, only when something other than the next propagate needs slot order: x[i], parent[i] from row inv[i].
// Owner-keeps exchange with states wider than one double: a kept offspring has no row — inv[i] = MP_INV_LOCAL | parent's
// local row, its state is the parent's in the pre-resample buffer x_old.
__global__ __launch_bounds__(SH_THREADS) void k_shard_adopt_rows(u64 n, int D, const double* __restrict__ rows, const uint32_t* __restrict__ inv,
                                                                 const double* __restrict__ x_old, u64 slot_offset, double* __restrict__ x_new,
                                                                 uint32_t* __restrict__ parent) {
    const u64 i = (u64)blockIdx.x * SH_THREADS + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = inv[i];
    if (D > 1 && (v & MP_INV_LOCAL)) {
        const u64 pr = v & ~MP_INV_LOCAL;
        for (int d = 0; d < D; ++d) x_new[i * D + d] = x_old[pr * D + d];
        parent[i] = (uint32_t)(slot_offset + pr);
        return;
    }
    const double* in = rows + (u64)v * (u64)(D + 1);
    for (int d = 0; d < D; ++d) x_new[i * D + d] = in[d];
    parent[i] = (uint32_t)in[D];
}

// parents alone (a step has already consumed the states of those rows)
__global__ __launch_bounds__(SH_THREADS) void k_shard_adopt_parents(u64 n, int D, const double* __restrict__ rows, const uint32_t* __restrict__ inv,
                                                                    u64 slot_offset, uint32_t* __restrict__ parent) {
    const u64 i = (u64)blockIdx.x * SH_THREADS + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = inv[i];
    parent[i] = (D > 1 && (v & MP_INV_LOCAL)) ? (uint32_t)(slot_offset + (v & ~MP_INV_LOCAL)) : (uint32_t)rows[(u64)v * (u64)(D + 1) + D];
}

// out[i] = a[i] - *b  (log_normalized_weights = w_i - log_total_weight, importance.rs:23-25)
__global__ void k_sub_scalar(const double* __restrict__ a, const double* __restrict__ b, u64 n, double* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] - *b;
}

