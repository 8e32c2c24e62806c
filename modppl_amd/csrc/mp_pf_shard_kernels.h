// mp_pf_shard_kernels.h — device code of the sharded filter's resample phases (included by mp_pf.hip only, after
// mp_pf_kernels.h): variable-size three-pass route, fixed-capacity single-pass route, owner-side resolve, adoption.
// Protocol: include/modppl_hip.h "sharded filter", DESIGN.md §8.
#pragma once
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
            target = mp_target(mp_resample_k52(slot_offset + i, rc, (uint32_t)MP_DOM_RESAMPLE, k0, k1), Q);
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
    unsigned long long counts[SH_MAX_WORLD];   // "owner keeps" form: offspring per rank of the last resample
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
            mp_locate_r(s_incl, s_W, ratio_all, (uint32_t)nt_all, target, nt_over_Q, &b, &lt[k], &gs);
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
constexpr int SHT_THREADS = 1024;
constexpr int SHT_PER = MAX_TILES / SHT_THREADS;   // tiles per thread, held in registers (8)
__global__ __launch_bounds__(SHT_THREADS) void k_shard_table(const u64* __restrict__ packed, int world, int nt_local, int S, u64 n_global,
                                                             double* __restrict__ tm, u64* __restrict__ tW, u64* __restrict__ tW2,
                                                             u64* __restrict__ incl_all, double* __restrict__ ratio_all,
                                                             long long* __restrict__ zero_counts, mp_dev_scalars* scal, mp_dev_scalars* undo,
                                                             unsigned long long* __restrict__ zero_call = nullptr) {
    __shared__ double s_red[SHT_THREADS / 64];
    __shared__ u64 s_wtot[SHT_THREADS / 64];
    __shared__ u64 s_wtot2[SHT_THREADS / 64];
    const int nt = world * nt_local;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < SH_MAX_KEYS) zero_counts[tid] = 0;
    if (zero_call && tid < SH_MAX_WORLD) zero_call[tid] = 0ull;
    // thread t owns tiles t * per .. t * per + per - 1 (consecutive, so that a thread-local running sum is a prefix); each
    // tile is read once, kept in registers, and its unpacked copy written for the kernels that want plain arrays
    const int per = (nt + SHT_THREADS - 1) / SHT_THREADS;   // <= SHT_PER since nt <= MAX_TILES
    const int b0 = tid * per;
    double mb[SHT_PER];
    u64 Wb[SHT_PER], W2b[SHT_PER];
    double m = MP_NEG_INF;
#pragma unroll
    for (int j = 0; j < SHT_PER; ++j) {
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
    u64 pre[SHT_PER];
    double ratio[SHT_PER];
    u64 run = 0, run2 = 0;
#pragma unroll
    for (int j = 0; j < SHT_PER; ++j) {
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
    for (int j = 0; j < SHT_PER; ++j) {
        const int i = b0 + j;
        if (j < per && i < nt) { incl_all[i] = off + pre[j]; ratio_all[i] = ratio[j]; }
    }
    if (tid == 0) {
        *undo = *scal;   // a fixed-capacity exchange that overflows puts these back
        fold_scalars(scal, Q, Q2, S, m, n_global, 0);
    }
}
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
            u64 j = tbase[k] + guide[gi[k]];
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
// "Owner keeps" form of the sharded resample.  Every rank enumerates ALL N draws of the job (the draws of the single
// filter: same Philox counters, same targets, hence the same parent for every draw g) and keeps those that land in its own
// rows; offspring then stay on the rank that owns their parent, in the order of their draws, and only the surplus over n
// slots travels (to the ranks that drew fewer than n): the xGMI traffic of a resample drops from ~40 B per particle to a
// few thousand rows.  WHERE an offspring sits depends on the number of ranks (a world of one is the single filter).
//   k_shard_own_draws : draws -> (mine?) -> this workgroup's own targets, compacted in draw order; offspring per rank (ballots)
//   k_shard_own_plan  : one workgroup: first offspring position of every k_shard_own_draws workgroup (scan), the exchange
//                       plan, the verdict "some pair needs more than cap rows", what the host reads
//   k_shard_own_place : own target -> tile, guide, walk -> row {x, parent id} written at the offspring's position: lane q
//                       of a workgroup handles its q-th own draw, so positions are consecutive and the writes coalesced.
// No counting, no scattered writes: a first version counted offspring per row with one atomic per draw and expanded the
// counts per tile (parent order); its 2^20 scattered 4-byte read-modify-writes alone cost 36 us of a 73 us count phase.
// ---------------------------------------------------------------------------------------------
constexpr int SHO_ITEMS = 8;
constexpr int SHO_CHUNK = SH_THREADS * SHO_ITEMS;   // draws per k_shard_own_draws workgroup
__global__ __launch_bounds__(SH_THREADS) void k_shard_own_draws(u64 n_global, uint32_t k0, uint32_t k1, uint32_t rc, int scheme,
                                                                const u64* __restrict__ incl_all, int nt_local, int world, int rank,
                                                                u64* __restrict__ gq, uint32_t* __restrict__ wgcnt,
                                                                unsigned long long* __restrict__ c_all) {
    __shared__ u64 s_bound[SH_MAX_WORLD];       // inclusive prefix of T_b at the end of every rank's tiles
    __shared__ uint32_t s_above[SH_MAX_WORLD];  // draws of this workgroup whose target lies above s_bound[r]
    __shared__ uint32_t s_wc[SHO_ITEMS][SH_THREADS / 64];   // own draws of (round k, wave w)
    __shared__ uint32_t s_live;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int r = tid; r < world; r += SH_THREADS) {
        s_bound[r] = incl_all[(u64)(r + 1) * nt_local - 1];
        s_above[r] = 0u;
    }
    if (tid == 0) s_live = 0u;
    __syncthreads();
    const u64 Q = s_bound[world - 1];
    const u64 lo = rank ? s_bound[rank - 1] : 0ull, hi = s_bound[rank];
    const uint32_t k32 = scheme == 1 ? mp_systematic_k32(rc, k0, k1) : 0u;
    const u64 g0 = (u64)blockIdx.x * SHO_CHUNK + tid;
    u64 target[SHO_ITEMS];
    uint32_t before[SHO_ITEMS];   // own draws of the same round in lower lanes of this wave
    bool mine[SHO_ITEMS];
    uint32_t nlive = 0;
#pragma unroll
    for (int k = 0; k < SHO_ITEMS; ++k) {
        const u64 g = g0 + (u64)k * SH_THREADS;
        target[k] = 0ull;   // below every boundary, and never "mine" (targets are >= 1)
        if (g < n_global) {
            if (scheme) {
                target[k] = mp_target_lattice(scheme, g, k32, rc, k0, k1, Q, n_global);
            } else {
                target[k] = mp_target(mp_resample_k52(g, rc, (uint32_t)MP_DOM_RESAMPLE, k0, k1), Q);
            }
            ++nlive;
        }
        mine[k] = target[k] > lo && target[k] <= hi;
        const u64 bal = __ballot(mine[k]);
        before[k] = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (lane == 0) s_wc[k][wave] = (uint32_t)__popcll(bal);
    }
    // offspring per rank: (# above the previous boundary) - (# above this one), counted per wave with ballots
    for (int r = 0; r + 1 < world; ++r) {
        const u64 B = s_bound[r];
        uint32_t a = 0;
#pragma unroll
        for (int k = 0; k < SHO_ITEMS; ++k) a += (uint32_t)__popcll(__ballot(target[k] > B));
        if (lane == 0 && a) atomicAdd(&s_above[r], a);
    }
    {
        uint32_t v = nlive;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0 && v) atomicAdd(&s_live, v);
    }
    __syncthreads();
    for (int r = tid; r < world; r += SH_THREADS) {
        const uint32_t above_prev = r ? s_above[r - 1] : s_live;
        const uint32_t above = (r + 1 < world) ? s_above[r] : 0u;
        if (above_prev > above) atomicAdd(&c_all[r], (unsigned long long)(above_prev - above));
    }
    // compaction in draw order: draw g = chunk + k * 256 + tid, so (round k, wave, lane) ascending IS g ascending
    u64* q_out = gq + (u64)blockIdx.x * SHO_CHUNK;
    uint32_t run = 0;
#pragma unroll
    for (int k = 0; k < SHO_ITEMS; ++k) {
        uint32_t off = run;
#pragma unroll
        for (int w = 0; w < SH_THREADS / 64; ++w) {
            const uint32_t c = s_wc[k][w];
            if (w < wave) off += c;
            run += c;
        }
        if (mine[k]) q_out[off + before[k]] = target[k];
    }
    if (tid == 0) wgcnt[blockIdx.x] = run;
}

// the exchange plan: unit u of the surplus (donors in rank order) fills unit u of the deficit (receivers in rank order)
struct mp_owned_plan {
    u64 S[SH_MAX_WORLD], D[SH_MAX_WORLD], PS[SH_MAX_WORLD], PD[SH_MAX_WORLD];
};
constexpr int SHP_THREADS = 1024;
__global__ __launch_bounds__(SHP_THREADS) void k_shard_own_plan(u64 n, int world, u64 cap, int nblk, const uint32_t* __restrict__ wgcnt,
                                                                const unsigned long long* __restrict__ c_all, const mp_dev_scalars* scal,
                                                                uint32_t* __restrict__ base, mp_owned_plan* __restrict__ plan_out, mp_shard_pub* pub) {
    __shared__ mp_owned_plan pl;
    __shared__ u64 s_wtot[SHP_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        u64 ps = 0, pd = 0;
        for (int r = 0; r < world; ++r) {
            const u64 c = c_all[r];
            pl.S[r] = c > n ? c - n : 0ull;
            pl.D[r] = c < n ? n - c : 0ull;
            pl.PS[r] = ps; pl.PD[r] = pd;
            ps += pl.S[r]; pd += pl.D[r];
            pub->counts[r] = c;
        }
        pub->L = scal->L;
        pub->degenerate = scal->degenerate;
    }
    // exclusive scan of the workgroup counts: thread t owns entries [t * per, (t + 1) * per)
    const int per = (nblk + SHP_THREADS - 1) / SHP_THREADS;
    const int b0 = tid * per;
    u64 run = 0;
    for (int j = 0; j < per; ++j)
        if (b0 + j < nblk) run += wgcnt[b0 + j];
    const u64 incl = wave_incl_scan_u64(run, lane);
    if (lane == 63) s_wtot[wave] = incl;
    __syncthreads();
    u64 woff = 0;
#pragma unroll
    for (int w = 0; w < SHP_THREADS / 64; ++w)
        if (w < wave) woff += s_wtot[w];
    u64 off = woff + (incl - run);
    for (int j = 0; j < per; ++j) {
        if (b0 + j < nblk) {
            base[b0 + j] = (uint32_t)off;
            off += wgcnt[b0 + j];
        }
    }
    // every rank must reach the same verdict on "some pair needs more than cap rows" (the collective that follows is symmetric),
    // so every rank looks at every pair of the plan
    int over = 0;
    if (cap) {
        for (int pq = tid; pq < world * world; pq += SHP_THREADS) {
            const int r = pq / world, s2 = pq - r * world;
            const u64 a0 = pl.PS[r] > pl.PD[s2] ? pl.PS[r] : pl.PD[s2];
            const u64 e0 = pl.PS[r] + pl.S[r], e1 = pl.PD[s2] + pl.D[s2];
            const u64 a1 = e0 < e1 ? e0 : e1;
            if (a1 > a0 && a1 - a0 > cap) over = 1;
        }
    }
    over = __syncthreads_or(over);
    if (tid == 0) pub->overflow = over;
    for (int r = tid; r < world; r += SHP_THREADS) {
        plan_out->S[r] = pl.S[r]; plan_out->D[r] = pl.D[r]; plan_out->PS[r] = pl.PS[r]; plan_out->PD[r] = pl.PD[r];
    }
}

__global__ __launch_bounds__(SH_THREADS) void k_shard_own_place(u64 n, u64 slot_offset, int D, int world, int rank, int nt_all, int nt_local, u64 cap,
                                                                u64 recv_rows, const u64* __restrict__ incl_all, const u64* __restrict__ tW_all,
                                                                const double* __restrict__ ratio_all, const mp_cx* __restrict__ cx,
                                                                const unsigned short* __restrict__ guide, const double* __restrict__ x,
                                                                const u64* __restrict__ gq, const uint32_t* __restrict__ wgcnt,
                                                                const uint32_t* __restrict__ base, const mp_owned_plan* __restrict__ plan,
                                                                const unsigned long long* __restrict__ c_all, double* __restrict__ rows,
                                                                double* __restrict__ send, uint32_t* __restrict__ inv) {
    const int tid = threadIdx.x;
    const uint32_t qn = wgcnt[blockIdx.x];
    const u64 p0 = base[blockIdx.x];
    const u64 Q = incl_all[nt_all - 1];
    const double nt_over_Q = (double)nt_all / (double)Q;
    const u64 PS_me = plan->PS[rank];
    const u64* q_in = gq + (u64)blockIdx.x * SHO_CHUNK;
    const int lane = tid & 63;
    // lanes of a row group when states are wider than one double (as k_shard_resolve_binned): G lanes copy one parent's row
    // together, a row is then one coalesced request instead of D scattered 8-byte ones
    const int G = D <= 2 ? 2 : (D <= 4 ? 4 : (D <= 8 ? 8 : 16));
    for (uint32_t q0 = 0; q0 < qn; q0 += SH_THREADS) {   // wave-uniform trip count
        const uint32_t q = q0 + tid;
        u64 i = 0;
        double* dst = nullptr;
        mp_cx cur;
        cur.cum = 0; cur.x0 = 0.;
        if (q < qn) {
            uint32_t b, gs;
            u64 lt;
            mp_locate_r(incl_all, tW_all, ratio_all, (uint32_t)nt_all, q_in[q], nt_over_Q, &b, &lt, &gs);
            const uint32_t tl = b - (uint32_t)rank * (uint32_t)nt_local;
            const u64 tbase = (u64)tl * TILE;
            const u64 tend = tbase + TILE;
            const u64 last = (tend < n ? tend : n) - 1;
            const u64 j = tbase + guide[(u64)tl * GUIDE_N + (gs - b * (uint32_t)GUIDE_N)];
            i = j < last ? j : last;
            cur = cx[i];
            while (cur.cum < lt && i < last) {
                ++i;
                cur = cx[i];
            }
            const u64 p = p0 + q;
            if (p < n) {
                dst = rows + (recv_rows + p) * (u64)(D + 1);
                inv[p] = (uint32_t)(recv_rows + p);
            } else {
                const u64 u = PS_me + (p - n);
                int s = 0;
                while (s + 1 < world && !(plan->D[s] && u < plan->PD[s] + plan->D[s])) ++s;
                if (cap) {
                    const u64 first = PS_me > plan->PD[s] ? PS_me : plan->PD[s];
                    const u64 jj = u - first;
                    // jj >= cap: k_shard_own_plan has flagged it, nothing of this attempt is committed
                    if (jj < cap) dst = send + ((u64)s * cap + jj) * (u64)(D + 1);
                } else {
                    dst = send + (u - PS_me) * (u64)(D + 1);
                }
            }
        }
        if (D == 1) {
            if (dst) *reinterpret_cast<double2*>(dst) = make_double2(cur.x0, (double)(slot_offset + i));
        } else {
            if (dst) dst[D] = (double)(slot_offset + i);
            const u64 dbits = (u64)(uintptr_t)dst;
            const uint32_t d_lo = (uint32_t)dbits, d_hi = (uint32_t)(dbits >> 32);
            const uint32_t i_lo = (uint32_t)i, i_hi = (uint32_t)(i >> 32);
            const int comp = lane % G;
            for (int base_l = 0; base_l < 64; base_l += 64 / G) {
                const int src = base_l + lane / G;
                const u64 sd = ((u64)(uint32_t)__shfl((int)d_hi, src, 64) << 32) | (u64)(uint32_t)__shfl((int)d_lo, src, 64);
                const u64 si = ((u64)(uint32_t)__shfl((int)i_hi, src, 64) << 32) | (u64)(uint32_t)__shfl((int)i_lo, src, 64);
                if (sd) {
                    double* out = reinterpret_cast<double*>((uintptr_t)sd);
                    for (int d = comp; d < D; d += G) out[d] = x[si * D + d];
                }
            }
        }
    }
    // slots this rank could not fill itself: where in the receive buffer their rows will arrive
    const u64 c_me = c_all[rank];
    const u64 PD_me = plan->PD[rank], D_me = plan->D[rank];
    for (u64 k = (u64)blockIdx.x * SH_THREADS + tid; k < D_me; k += (u64)gridDim.x * SH_THREADS) {
        u64 idx = k;
        if (cap) {
            const u64 u = PD_me + k;
            int r = 0;
            while (r + 1 < world && !(plan->S[r] && u < plan->PS[r] + plan->S[r])) ++r;
            const u64 first = plan->PS[r] > PD_me ? plan->PS[r] : PD_me;
            const u64 jj = u - first;
            idx = jj >= cap ? 0ull : (u64)r * cap + jj;
        }
        inv[c_me + k] = (uint32_t)idx;
    }
}
// After the resolve: "somebody overflowed" (flags are only ever OR-ed atomically) and the scalars of this normalisation,
// where the host reads them after waiting for ev_resolved (host-mapped memory: no copy command, no stream sync).  A kernel of
// its own: a ticket per resolve workgroup (1300 same-address atomics) measured 7 us, this launch 4.
__global__ void k_shard_publish(int* overflow, const mp_dev_scalars* scal, mp_shard_pub* pub) {
    pub->L = scal->L;
    pub->degenerate = scal->degenerate;
    pub->overflow = atomicOr(overflow, 0);
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

// parents alone (a step has already consumed the states of those rows)
__global__ __launch_bounds__(SH_THREADS) void k_shard_adopt_parents(u64 n, int D, const double* __restrict__ rows, const uint32_t* __restrict__ inv,
                                                                    uint32_t* __restrict__ parent) {
    const u64 i = (u64)blockIdx.x * SH_THREADS + threadIdx.x;
    if (i >= n) return;
    parent[i] = (uint32_t)rows[(u64)inv[i] * (u64)(D + 1) + D];
}

// out[i] = a[i] - *b  (log_normalized_weights = w_i - log_total_weight, importance.rs:23-25)
__global__ void k_sub_scalar(const double* __restrict__ a, const double* __restrict__ b, u64 n, double* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] - *b;
}

