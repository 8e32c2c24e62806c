// mp_pf_k1mt.h — K1 as ONE workgroup per CU that walks SEVERAL tiles (included by mp_pf.hip after mp_pf_kernels.h).
//
// k_propagate (mp_pf_kernels.h) is one workgroup per tile: at 2^20 particles two workgroups share a CU, both build the job's
// tile table, and both reach every phase — table, targets, row lookups, deviates, normalisation — at about the same time, so the
// CU's vector-memory path and its VALUs are busy one AFTER the other (profiles/r03: 19.5 us of instruction issue + ~18 us of
// row lookups in a 38.3 us kernel).  Here a workgroup of 1024 threads (16 waves, up to 128 VGPRs: one workgroup per CU) owns TWO
// tiles, A = blockIdx.x and B = blockIdx.x + gridDim.x (gridDim.x = half the tiles, rounded up), and software-pipelines them in
// program order (straight-line code: a loop over tiles made the compiler carry both tiles' state through scratch memory):
//
//     table (ONCE per workgroup) | draws(A) | draws(B) | rows(A) asked | deviates(A), deviates(B)  [rows(A) in flight] |
//     parents(A), rows(B) asked | model + normalise(A)  [rows(B) in flight] | parents(B) | model + normalise(B)
//
// so that a tile's row gathers (the fabric's time) run under the arithmetic of the same lane: both tiles' deviates under A's,
// A's normalisation under B's.  Same draws (Philox counters, targets, walks), same deviates, same normalisation:
// bit-identical parents / states / weights (tests/test_gpu_deferred.py runs both kernels against the checker and each other).
// Reference work: `multinomial_resampling` + the clone loop of `resample` (particle_filter.rs:37-41, 109-114), then `step`
// (:73-96) and `normalize_weights` (:27-35) for every tile of the workgroup.
#pragma once

template <int N>
struct mp_obs_n {           // the observation of one time step, only as wide as the model's (mp_obs is 16 doubles whatever the model)
    double v[N];
};

struct mp_k1mt {            // by value (kernel arguments)
    u64 n, slot_offset, n_global;
    long long t;
    uint32_t k0, k1, rc;
    int S;
    int flags;              // MP_MT_SKIP_LOGW: the log-weights are not stored, MP_MT_SKIP_PARENT: parent[] is not stored (both produced on demand by a
                            // launch with MP_MT_REPLAY: `resample` zeroes the one and replaces the other, particle_filter.rs:109-114, so in a
                            // step / resample loop nobody ever reads them)
    double* logw;
    const mp_cx* cx_old;                 // the generation that was resampled: its rows and guide (read) ...
    const unsigned short* guide_old;
    mp_cx* cx_new;                       // ... and the one this launch writes
    unsigned short* guide_new;
    double* tm_new;
    u64* tW_new;
    u64* tW2_new;
    uint32_t* parent;
    mp_dev_scalars* scal;
    // MP_MT_PEEK (the caller's loop is synchronous: `L = resample()` follows, particle_filter.rs:103-116): the workgroup that finishes last
    // folds level 1 of the tile scalars THIS launch wrote into the host-mapped mirror — what k_peek_level1 does as a launch of its own,
    // without the launch, its dependency gap and its start on an idle queue
    unsigned int* peek_ticket;
    mp_host_mirror* mirror;
    unsigned long long peek_seq;
};
constexpr int MP_MT_SKIP_LOGW = 1, MP_MT_SKIP_PARENT = 2;
// MP_MT_REPLAY: the launch repeats an earlier one of the same arguments ONLY to produce what that one skipped — the log-weights and the
// parents (mp_pf.hip ensure_lazy): nothing else is stored, nothing folded, no normalisation
constexpr int MP_MT_REPLAY = 4;
constexpr int MP_MT_PEEK = 8;

// what a lane carries for one of its tiles between the stages of the pipeline
struct mp_mt_tile {
    u64 base;               // first of the lane's two slots (local to the handle)
    u64 plt[2];             // tile-local targets of its two draws
    uint32_t tile_of[2];    // the draws' tiles, then (mt_rows) their start rows
    uint32_t g[2];          // guide cells (in flight after mt_draw)
    uint32_t sp[2];         // the 32nd of its guide cell each target lies in (mp_guide_sub): with the cell's position bits it says which rows to ask for
    mp_u64v2 a[2], b2[2];   // start row and successor (in flight after mt_rows)
};

// stage 1: the two draws of the lane's slots of tile `tile`: targets, tile walk in the LDS table, guide cells requested
template <bool WALKB>
__device__ __forceinline__ void mt_draw(mp_mt_tile& T, u64 tile, const mp_u64x2& blk, const mp_k1mt& a, int nt,
                                        const u64* s_incl, const u64* s_W, const double* s_ratio, u64 Q, double nt_over_Q) {
    T.base = tile * TILE + (u64)threadIdx.x * 2;
#if MP_MT_PROBE & 8
    for (int q = 0; q < 2; ++q) {
        T.tile_of[q] = (uint32_t)tile;
        T.plt[q] = 1ull;
        T.g[q] = (uint32_t)(threadIdx.x * 2 + q);
        T.sp[q] = 0u;
    }
    return;
#endif
    uint32_t gslot[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const u64 tg = mp_target(mp_u52(q ? blk.b : blk.a), Q);
        mp_locate_r<WALKB>(s_incl, s_W, s_ratio, (uint32_t)nt, tg, nt_over_Q, &T.tile_of[q], &T.plt[q], &gslot[q], &T.sp[q]);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) T.g[q] = a.guide_old[gslot[q]];
}
// stage 2: start rows from the guide cells; the rows requested.  MP_MT_Q5 (default): the cell's position bits against the target's own
// 32nd of the cell (mp_guide_q5 / mp_guide_sub, mp_pf_kernels.h) say which rows — r0 ALONE when the target lies below cum[r0] (39 % of
// the draws: the parent is r0, only its state is wanted), r0 + 1 and r0 + 2 when it lies above (r0's row is not fetched; the pair ends all
// but 1.2 % of the walks), r0 and r0 + 1 when the two fall into the same 32nd (3 %).  Before: r0 and its successor for every draw, and a
// third hop (three more rows) for the quarter of the lanes whose walk went past the pair — which every wave of the workgroup waited for.
#ifndef MP_MT_Q5
#define MP_MT_Q5 1
#endif
// Timing probes (tools/build_variant.py builds of their own, NEVER the product: results are not the filter's): what the kernel takes
// without one of its two big resources.  Bit 0: the standard deviates cost nothing (a hash of the slot instead of Philox, the polar
// loop, a logarithm, a root and a division per deviate: same spread, so the weights stay healthy); bit 1: no row gathers (the parent
// IS the guide cell's start row, its state a hash of the slot: no random 16-byte reads of the 16 MB table, the rest unchanged).
// Bit 2: no normalisation arithmetic (every row weighs 2^40: no exponentials, scans, barriers or guide walk; the table stays valid);
// bit 3: no draws (no Philox block, 128-bit target, tile walk or guide gather: a slot's parent is its own row; with bit 1).
#ifndef MP_MT_PROBE
#define MP_MT_PROBE 0
#endif
__device__ __forceinline__ double mt_probe_deviate(u64 slot, long long t, int salt) {
    uint32_t h = (uint32_t)slot * 2654435761u ^ (uint32_t)t * 40503u ^ (uint32_t)salt * 97u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    return (double)(h & 0xFFFFFu) * (3.4641016 / 1048576.) - 1.7320508;   // uniform, variance 1
}
__device__ __forceinline__ void mt_rows(mp_mt_tile& T, const mp_k1mt& a) {
    bool hbv[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const u64 tbase = (u64)T.tile_of[q] * TILE;
        const uint32_t tlen = (uint32_t)((a.n - tbase) < (u64)TILE ? (a.n - tbase) : (u64)TILE);
        const uint32_t j0 = mp_guide_row(T.g[q]);
        uint32_t r0 = (uint32_t)tbase + (j0 > tlen - 1 ? tlen - 1 : j0);   // row where the forward scan starts
        const uint32_t last = (uint32_t)tbase + tlen - 1;
        if constexpr (MP_MT_Q5 != 0) {
            const uint32_t q5 = mp_guide_q5(T.g[q]);
            const bool below = T.sp[q] < q5, above = T.sp[q] > q5;
            if (above && r0 < last) ++r0;                            // cum[r0] < target: the scan may start one row on
            hbv[q] = !below && r0 < last;
        } else {
            hbv[q] = r0 < last && (MP_PAIR_SAME_LINE ? (r0 & 3u) != 3u : true);
        }
        T.tile_of[q] = r0;
    }
#if MP_MT_PROBE & 2
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        T.a[q].x = ~0ull;
        T.a[q].y = __builtin_bit_cast(u64, 0.8 * mt_probe_deviate(T.base + q, a.t, 7));
        T.b2[q] = T.a[q];
        T.sp[q] = 0u;
    }
    return;
#endif
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const uint32_t r0 = T.tile_of[q];
        T.a[q] = mp_ld_row(a.cx_old + r0);
        // (a lane that wants no second row reads row 0 of the table with every other such lane of the chip — or, without the position bits,
        // its own row again, as before)
        T.b2[q] = mp_ld_row(a.cx_old + (hbv[q] ? (u64)r0 + 1 : (MP_MT_Q5 != 0 ? 0ull : (u64)r0)));
        T.sp[q] = hbv[q] ? 1u : 0u;                                  // (from here on: whether b2 is the successor)
    }
}
// stage 3: the parents (mp_resolve_draws' walk on the rows that have arrived) and their states — in two halves, because a CU's
// vector-memory operations are served IN ORDER: whatever a wave asks for behind another tile's 4096 row gathers waits for all
// of them.  So the first walk loads of this tile (mt_resolve_first: the next MP_MT_WALK_ROWS rows at once, which ends nearly
// every walk) go out BEFORE the next tile's row pairs, and only the rare deeper walk (mt_resolve_rest) waits behind them.
// (The first form of this kernel asked for B's rows before A's walks: 45 us against 39.)
#ifndef MP_MT_WALK_ROWS
#define MP_MT_WALK_ROWS 3
#endif
struct mp_mt_walk {
    uint32_t p[2];                       // row reached so far
    mp_u64v2 cur[2];                     // its contents
    mp_u64v2 nx[2][MP_MT_WALK_ROWS];     // (more) the rows p + 1 ..: requested, not yet looked at
    bool more[2];
};
__device__ __forceinline__ void mt_resolve_first(mp_mt_tile& T, const mp_k1mt& a, mp_mt_walk& W) {
#pragma unroll
    for (int q = 0; q < 2; ++q) mp_pin_rows(T.a[q], T.b2[q]);   // the rows are looked at here, not earlier
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const uint32_t r0 = T.tile_of[q];
        const u64 last = mp_tile_last(r0, a.n);
        const bool hb = T.sp[q] != 0u;                          // (mt_rows: the successor was asked for)
        const bool step1 = T.a[q].x < T.plt[q] && hb;
        W.p[q] = r0 + (step1 ? 1u : 0u);
        W.cur[q] = step1 ? T.b2[q] : T.a[q];
        W.more[q] = W.cur[q].x < T.plt[q] && (u64)W.p[q] < last;
        // more than one row past the guide's start (a successor in the next 64-byte line, mostly): the next rows, asked for by EVERY
        // lane — the lanes that need none read row 0, one line for the whole wave — so that this is straight-line code: behind a
        // branch the compiler no longer knows how many loads are in flight and waits for all of them, the next tile's included
#pragma unroll
        for (int k = 0; k < MP_MT_WALK_ROWS; ++k) {
#if MP_MT_PROBE & 2
            W.nx[q][k] = W.cur[q];
#else
            const u64 r = (u64)W.p[q] + 1 + k;
            W.nx[q][k] = mp_ld_row(a.cx_old + (W.more[q] ? (r < last ? r : last) : 0ull));
#endif
        }
    }
}
template <bool WALKB>
__device__ __forceinline__ void mt_resolve_rest(const mp_mt_tile& T, const mp_k1mt& a, mp_mt_walk& W, uint32_t* parent, double* x0) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const u64 last = mp_tile_last(T.tile_of[q], a.n);
        u64 p = W.p[q];
        mp_u64v2 cur = W.cur[q];
        if (W.more[q]) {
#pragma unroll
            for (int k = 0; k < MP_MT_WALK_ROWS; ++k) {
                if (cur.x < T.plt[q] && p < last) { ++p; cur = W.nx[q][k]; }
            }
            if (cur.x < T.plt[q] && p < last) {   // rare: the walk is longer than the rows asked for in advance — the rest by bisection (the first
                u64 lo = p + 1, hi = last;        // row of (p, last] whose cumulative weight reaches the target, or `last`: mp_resolve_draws' BISECT)
                if constexpr (WALKB) {
                    // (a few more rows one at a time first: a bisection is 11 dependent loads, a 5 - 8 row walk fewer)
                    int steps = 0;
                    while (cur.x < T.plt[q] && p < last && steps < MP_WALK_LINEAR - MP_MT_WALK_ROWS) {
                        ++p; ++steps;
                        cur = mp_ld_row(a.cx_old + p);
                    }
                    if (!(cur.x < T.plt[q] && p < last)) { lo = p; hi = p; }
                    else lo = p + 1;
                    while (lo < hi) {
                        const u64 mid = lo + ((hi - lo) >> 1);
                        if (mp_ld_row(a.cx_old + mid).x >= T.plt[q]) hi = mid;
                        else lo = mid + 1;
                    }
                    p = lo;
                    cur = mp_ld_row(a.cx_old + p);
                } else {
                    while (cur.x < T.plt[q] && p < last) {
                        ++p;
                        cur = mp_ld_row(a.cx_old + p);
                    }
                }
            }
        }
        parent[q] = (uint32_t)p;
        x0[q] = __builtin_bit_cast(double, (u64)cur.y);
    }
}

// The standard deviates of a lane's two particles (k_propagate's wave-cooperative form, same attempts in the same order):
// attempt 0 of every deviate straight-line, the rejected ones retried by the wave.  s_it: this wave's 64 words of LDS.
template <class Model>
__device__ __forceinline__ void mt_deviates(const Model& model, int ns, const mp_k1mt& a, u64 base, u64 tile, uint32_t* s_it, double* z) {
    constexpr int NS = Model::MAX_NORMALS;
    constexpr int M = 2 * NS;
#if MP_MT_PROBE & 1
    for (int q = 0; q < M; ++q) z[q] = mt_probe_deviate(a.slot_offset + base + q / NS, a.t, q % NS);
    return;
#endif
    const int lane_ = threadIdx.x & 63, wave_ = threadIdx.x >> 6;
    double pu[M], pr[M];
    uint32_t pend = 0u;
    uint32_t att[M];
#pragma unroll
    for (int q = 0; q < M; ++q) {
        const int p = q / NS, sidx = q % NS;
        pu[q] = 0.; pr[q] = 1.; att[q] = 1u;
        if (sidx < ns && base + p < a.n) {
            const mp_u64x2 b = mp_philox4x32_10((uint32_t)(a.slot_offset + base + p), (uint32_t)a.t,
                                                ((uint32_t)MP_DOM_MODEL << 16) | model.normal_site(sidx), 0u, a.k0, a.k1);
            if (!mp_polar_attempt(b, &pu[q], &pr[q])) pend |= 1u << q;
        }
    }
    const u64 wave_slot0 = a.slot_offset + tile * TILE + (u64)wave_ * 128;   // first slot of this wave's lanes
    for (uint32_t guard = 0; guard < MP_MAX_ATTEMPTS; ++guard) {
        uint32_t idx[M];
        uint32_t R = 0;   // wave-uniform
#pragma unroll
        for (int q = 0; q < M; ++q) {
            const u64 bal = __ballot((pend >> q) & 1u);
            idx[q] = R + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            R += (uint32_t)__popcll(bal);
        }
        if (R == 0u) break;
        const uint32_t R1 = R < 64u ? R : 64u;
        const int lg = R1 <= 1u ? 0 : 32 - __builtin_clz(R1 - 1u);
        const uint32_t R2 = 1u << lg;
#pragma unroll
        for (int q = 0; q < M; ++q)
            if (((pend >> q) & 1u) && idx[q] < 64u) s_it[idx[q]] = (uint32_t)lane_ | ((uint32_t)q << 8) | (att[q] << 16);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t w = (uint32_t)lane_ & (R2 - 1u), k = (uint32_t)lane_ >> lg;
        const bool work = w < R1;
        const uint32_t ent = s_it[work ? w : 0u];
        const uint32_t owner = ent & 63u, oq = (ent >> 8) & 0xFFu, oatt = ent >> 16;
        const mp_u64x2 b = mp_philox4x32_10((uint32_t)(wave_slot0 + owner * 2 + oq / NS), (uint32_t)a.t,
                                            ((uint32_t)MP_DOM_MODEL << 16) | model.normal_site((int)(oq % NS)), oatt + k, a.k0, a.k1);
        double u, r;
        const bool acc = mp_polar_attempt(b, &u, &r) && work;
        const u64 A = __ballot(acc);
        __builtin_amdgcn_wave_barrier();
        const u64 pat = lg == 0 ? ~0ull : lg == 1 ? 0x5555555555555555ull : lg == 2 ? 0x1111111111111111ull : lg == 3 ? 0x0101010101010101ull
                      : lg == 4 ? 0x0001000100010001ull : lg == 5 ? 0x0000000100000001ull : 1ull;
#pragma unroll
        for (int q = 0; q < M; ++q) {
            const bool mine = ((pend >> q) & 1u) && idx[q] < 64u;
            const u64 hits = mine ? ((A >> idx[q]) & pat) : 0ull;
            const int src = mine && hits ? (int)idx[q] + (__ffsll((long long)hits) - 1) : lane_;
            const double gu = __shfl(u, src, 64), gr = __shfl(r, src, 64);
            if (hits) { pu[q] = gu; pr[q] = gr; pend &= ~(1u << q); }
            else if (mine) att[q] += 64u >> lg;
        }
    }
#pragma unroll
    for (int q = 0; q < M; ++q) z[q] = mp_std_normal_from_pair(pu[q], pr[q]);
}

#ifndef MP_MT_WALK_BARRIER
#define MP_MT_WALK_BARRIER 1
#endif
#ifndef MP_MT_ORDER
#define MP_MT_ORDER 0   // (A/B builds) 0: the deviates of A under the guide gathers, those of B under A's row gathers;  1: both under A's row gathers;
                        // 2, 3: as 1, 0 with B's draws behind A's row requests
#endif

// `step` for the lane's two slots of one tile: the model functor in Generate mode on the parents' states.  Nothing is stored here
// (mt_store_tile): between a tile's row gathers and the next tile's, every vector-memory operation would queue behind 4096 gathers.
template <class Model>
__device__ __forceinline__ void mt_model(const Model& model, const mp_k1mt& a, const mp_obs_n<Model::DIM_OBS>& obs, u64 base, const double* px0, const double* z,
                                         double (&lw)[2], double (&xv)[2]) {
    constexpr int NS = Model::MAX_NORMALS;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        lw[p] = MP_NEG_INF; xv[p] = 0.;
        const u64 i = base + (u64)p;
        if (i < a.n) {
            double prev[1] = {px0[p]}, next[1];
            mp_stream rng;
            rng.k0 = a.k0; rng.k1 = a.k1; rng.slot = (uint32_t)(a.slot_offset + i); rng.step = (uint32_t)a.t;
            mp_generate_handler<Model> g(rng, obs.v, &z[p * NS]);
            model(g, a.t, prev, next);
            lw[p] = 0. + g.weight;   // log_weights.fill(0.) of the resample (particle_filter.rs:114), then += (:81)
            xv[p] = next[0];
        }
    }
}

// Level 0 of the normalisation of one tile (normalize_tile's arithmetic, operation for operation) WITHOUT its stores: the lane's
// two row values come back in cum[], the tile's guide is left in s_guide (LDS, this tile's own 4 KB), its scalars in (m, W, W2)
// — W2 valid in thread 0.  Two barriers; the guide is complete for other waves only after a further barrier of the caller's.
struct mp_mt_lds {
    double s_red[16];
    u64 s_wsum[16];
    u64 s_wsum2[16];
};
// A barrier for data exchanged through LDS only: __syncthreads() is also a workgroup-scope fence for global memory, i.e. an
// s_waitcnt vmcnt(0) — with a tile's row gathers in flight every barrier of the other tile's normalisation would wait for them.
__device__ __forceinline__ void mt_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void mt_norm_compute(const double (&lw)[2], u64 n, u64 tile, mp_mt_lds& L, unsigned short* s_guide, u64 (&cum)[2], double& m_out,
                                                u64& W_out, u64& W2_out) {
    constexpr int THREADS = 1024;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 base = tile * TILE + (u64)tid * 2;
#if MP_MT_PROBE & 4
    {
        const u64 q0 = 1ull << 40;
        cum[0] = (u64)(tid * 2 + 1) * q0; cum[1] = (u64)(tid * 2 + 2) * q0;
        m_out = 0.; W_out = (u64)TILE * q0; W2_out = (u64)TILE * q0;
        reinterpret_cast<uint32_t*>(s_guide)[tid] = (uint32_t)((tid * 2) | (31 << MP_GUIDE_Q_SHIFT)) | ((uint32_t)((tid * 2 + 1) | (31 << MP_GUIDE_Q_SHIFT)) << 16);
        (void)lane; (void)wave; (void)base; (void)lw; (void)n; (void)L;
        return;
    }
#endif
    double m = MP_NEG_INF;
#pragma unroll
    for (int j = 0; j < 2; ++j)
        if (base + j < n) m = fmax(m, lw[j]);
    m = wave_max(m);
    if (lane == 0) L.s_red[wave] = m;
    reinterpret_cast<uint32_t*>(s_guide)[tid] = 0u;   // 1024 threads x 4 B = the whole guide
    mt_lds_barrier();
    m = L.s_red[0];
#pragma unroll
    for (int w = 1; w < THREADS / 64; ++w) m = fmax(m, L.s_red[w]);
    const bool ok = (m > MP_NEG_INF) && (m < MP_INF);
    u64 c[2];
    u64 run = 0, run2 = 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const bool live = ok && (base + j < n);
        const double e = live ? mp_exp_nonpos(lw[j] - m) : 0.;
        run += mp_quantize51(e);
        run2 += mp_quantize51(e * e);
        c[j] = run;
    }
    const u64 incl = wave_incl_scan_u64(run, lane);
    const u64 wtot2 = wave_sum_u64(run2);
    if (lane == 63) L.s_wsum[wave] = incl;
    if (lane == 0) L.s_wsum2[wave] = wtot2;
    mt_lds_barrier();
    u64 woff = 0, W = 0;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
    for (int k = 0; k < THREADS / 64; ++k) {
        const u64 vv = L.s_wsum[k];
        const u64 v = ((u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(vv >> 32)) << 32) | (u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)vv);
        if (k < wave_s) woff += v;
        W += v;
    }
    const u64 off = woff + (incl - run);
    u64 t2 = 0;
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < THREADS / 64; ++k) t2 += L.s_wsum2[k];
    }
    m_out = m; W_out = W; W2_out = t2;
    cum[0] = off + c[0]; cum[1] = off + c[1];
    // ---- guide table of this tile (normalize_tile's, two rows per thread) ----
    const int shift = mp_guide_shift(W);
    u64 prev = off;
    int long_lo = 0, long_hi = -1, long_j = 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const u64 cur = off + c[j];
        if (cur > prev) {
            const int g_lo = prev ? (int)(prev >> shift) + 1 : 0;
            const int g_hi = (int)(cur >> shift);
            const unsigned short idx = (unsigned short)((tid * 2 + j) | (31 << MP_GUIDE_Q_SHIFT));                            // (mp_guide_q5, mp_pf_kernels.h)
            const unsigned short idx_hi = (unsigned short)((tid * 2 + j) | (mp_guide_sub(cur, shift) << MP_GUIDE_Q_SHIFT));
            if (g_hi - g_lo < GUIDE_DIRECT || long_hi >= long_lo) {
                for (int g = g_lo; g < g_hi; ++g) s_guide[g] = idx;
                if (g_hi >= g_lo) s_guide[g_hi] = idx_hi;
            } else {   // (the run's last cell now, the cells in front of it by the wave below)
                s_guide[g_hi] = idx_hi;
                long_lo = g_lo; long_hi = g_hi - 1; long_j = idx;
            }
        }
        prev = cur;
    }
    u64 pending = __ballot(long_hi >= long_lo);
    while (pending) {
        const int leader = __ffsll((long long)pending) - 1;
        const int lo = __shfl(long_lo, leader, 64), hi = __shfl(long_hi, leader, 64), jj = __shfl(long_j, leader, 64);
        for (int g = lo + lane; g <= hi; g += 64) s_guide[g] = (unsigned short)jj;
        pending &= pending - 1;
    }
}
// what a tile leaves with a lane until its stores go out
struct mp_mt_out {
    u64 cum[2];
    double xv[2], lw[2];
    uint32_t par[2];
    double m;
    u64 W, W2;
};
// ... and the stores: parents, log-weights, rows, the tile's scalars (everything but the guide, which needs a barrier after its build)
__device__ __forceinline__ void mt_store_tile(const mp_k1mt& a, u64 tile, const mp_mt_out& o) {
    const u64 base = tile * TILE + (u64)threadIdx.x * 2;
    if (!(a.flags & MP_MT_SKIP_PARENT)) {
        if (base + 1 < a.n) *reinterpret_cast<uint2*>(a.parent + base) = make_uint2(o.par[0], o.par[1]);
        else if (base < a.n) a.parent[base] = o.par[0];
    }
    if (!(a.flags & MP_MT_SKIP_LOGW)) {
        if (base + 1 < a.n) *reinterpret_cast<double2*>(a.logw + base) = make_double2(o.lw[0], o.lw[1]);
        else if (base < a.n) a.logw[base] = o.lw[0];
    }
    mp_cx* cx = mp_as_global(a.cx_new);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (base + j < a.n) {
            mp_u64v2_ row;
            row.x = o.cum[j];
            row.y = mp_f2u(o.xv[j]);
            mp_st_stream16<4>(cx + base + j, row);
        }
    }
    if (threadIdx.x == 0) {
        if (a.flags & MP_MT_PEEK) {   // (read by another workgroup of THIS launch: agent-scope stores, mt_peek_tail)
            mp_st_agent(mp_as_global(a.tm_new) + tile, o.m);
            mp_st_agent(mp_as_global(a.tW_new) + tile, o.W);
            mp_st_agent(mp_as_global(a.tW2_new) + tile, o.W2);
        } else {
            mp_as_global(a.tm_new)[tile] = o.m;
            mp_as_global(a.tW_new)[tile] = o.W;
            mp_as_global(a.tW2_new)[tile] = o.W2;
        }
    }
}
// MP_MT_PEEK: every workgroup takes a ticket once its tile scalars are out; the one whose ticket comes last computes level 1 of the NEW
// tile scalars (k_peek_level1's arithmetic entry by entry: the same integer sums, the same L and ESS bits) and hands them to the host
// through the mirror, sequence word last.  Nothing is folded into the filter's scalars: the resample's draws — and with them the fold
// into log_ml — are still made by the next step's launch.
__device__ __forceinline__ void mt_peek_tail(const mp_k1mt& a, int nt) {
    __shared__ int s_last;
    __shared__ double s_red[16];
    __shared__ u64 s_q[16], s_q2[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this workgroup's tile scalars are out before its ticket says so
        const unsigned int t = __hip_atomic_fetch_add(a.peek_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = t == gridDim.x - 1u;
    }
    __syncthreads();
    if (!s_last) return;   // workgroup-uniform
    const double* tm = mp_as_global(a.tm_new);
    const u64 *tW = mp_as_global(a.tW_new), *tW2 = mp_as_global(a.tW2_new);
    // nt <= 1024 (the launch condition of this kernel): one tile per thread, its three scalars asked for together — one round trip
    // through the L2, not two (the maximum first, the rest afterwards)
    const bool have = tid < nt;
    const double mb = have ? mp_ld_agent(tm + tid) : MP_NEG_INF;
    const u64 Wb = have ? mp_ld_agent(tW + tid) : 0ull, W2b = have ? mp_ld_agent(tW2 + tid) : 0ull;
    double m = wave_max(mb);
    if (lane == 0) s_red[wave] = m;
    __syncthreads();
    m = s_red[0];
#pragma unroll 1
    for (int w = 1; w < 16; ++w) m = fmax(m, s_red[w]);
    const bool ok = (m > MP_NEG_INF) && (m < MP_INF);
    const double sc = mp_u2f((u64)(1023 + a.S - FIX_BITS) << 52);  // 2^(S-51)
    u64 q = 0, q2 = 0;
    if (have) {
        const double d = mb - m;
        q = mp_quantize((double)Wb * (ok ? mp_exp(d) : 0.) * sc, 1.0);
        q2 = mp_quantize((double)W2b * (ok ? mp_exp(2. * d) : 0.) * sc, 1.0);
    }
    q = wave_sum_u64(q);
    q2 = wave_sum_u64(q2);
    if (lane == 0) { s_q[wave] = q; s_q2[wave] = q2; }
    __syncthreads();
    if (tid == 0) {
        u64 Q = 0, Q2 = 0;
        for (int w = 0; w < 16; ++w) { Q += s_q[w]; Q2 += s_q2[w]; }
        double L, ess;
        finalize_scalars(Q, Q2, a.S, &L, &ess, m);
        const int degenerate = (!ok || Q == 0) ? 1 : 0;
        if (degenerate) mp_flag_degenerate(mp_as_global(a.scal));
        mp_host_mirror* hm = a.mirror;
        __hip_atomic_store(reinterpret_cast<u64*>(&hm->peek_L), mp_f2u(L), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(reinterpret_cast<u64*>(&hm->peek_ess), mp_f2u(ess), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&hm->peek_degenerate, degenerate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        mp_st_sys_seq(&hm->peek_seq, a.peek_seq);
        __hip_atomic_store(a.peek_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
    }
}
// (MP_MT_REPLAY) only what the first launch of these arguments left out
__device__ __forceinline__ void mt_store_replay(const mp_k1mt& a, u64 tile, const mp_mt_out& o) {
    const u64 base = tile * TILE + (u64)threadIdx.x * 2;
    if (base + 1 < a.n) {
        *reinterpret_cast<uint2*>(a.parent + base) = make_uint2(o.par[0], o.par[1]);
        *reinterpret_cast<double2*>(a.logw + base) = make_double2(o.lw[0], o.lw[1]);
    } else if (base < a.n) {
        a.parent[base] = o.par[0];
        a.logw[base] = o.lw[0];
    }
}
__device__ __forceinline__ void mt_store_guide(const mp_k1mt& a, u64 tile, const unsigned short* s_guide) {
    reinterpret_cast<uint32_t*>(mp_as_global(a.guide_new) + tile * GUIDE_N)[threadIdx.x] = reinterpret_cast<const uint32_t*>(s_guide)[threadIdx.x];
}

// WALKB: walks beyond the rows asked for in advance finish by bisection (collapsed weights: the host picks, mp_pf.hip launch_propagate)
template <class Model, bool WALKB = false>
__global__ __launch_bounds__(1024) void k_propagate_mt(const double* __restrict__ pre_tm, const u64* __restrict__ pre_tW, const u64* __restrict__ pre_tW2,
                                                       int pre_nt, int /*drw: multinomial draws only*/, Model model, mp_k1mt a, mp_obs_n<Model::DIM_OBS> obs) {
    constexpr int THREADS = 1024, NWV = THREADS / 64;
    constexpr int NS = Model::MAX_NORMALS;
    static_assert(Model::DIM_STATE == 1 && 2 * NS <= 4, "k_propagate_mt: two-slot lanes of a one-dimensional state");
    static_assert(GUIDE_N * 2 == THREADS * 4, "k_propagate_mt: one 4-byte word of the guide per thread");
    extern __shared__ __attribute__((aligned(16))) unsigned char k1_dyn[];
    u64* s_incl = reinterpret_cast<u64*>(k1_dyn);              // [nt]
    u64* s_W = s_incl + pre_nt;                                 // [nt]
    double* s_ratio = reinterpret_cast<double*>(s_W + pre_nt);  // [nt]
    __shared__ double s_l1_red[NWV];
    __shared__ u64 s_l1_tot[NWV], s_l1_tot2[NWV];
    __shared__ uint32_t s_it[NWV][64];
    __shared__ mp_mt_lds s_norm;
    __shared__ __attribute__((aligned(16))) unsigned short s_guideA[GUIDE_N], s_guideB[GUIDE_N];
    const int tb = (int)threadIdx.x, lane1 = tb & 63, wave1 = tb >> 6;
    const int ns = model.n_normals(a.t);  // wave-uniform
    a.guide_old = mp_as_global(a.guide_old); a.cx_old = mp_as_global(a.cx_old); a.logw = mp_as_global(a.logw); a.parent = mp_as_global(a.parent);
    MP_STAMP_L_DECL;
    MP_STAMP_L(0, 0); MP_STAMP_L(1, 1); MP_STAMP_L(6, 2);
    const u64 tileA = blockIdx.x, tileB = (u64)blockIdx.x + gridDim.x;
    const bool hasB = tileB < (u64)pre_nt;   // workgroup-uniform

    // ---- level 1 of the normalisation that was resampled: the job's tile table, ONCE per workgroup, in LDS (k_propagate's
    // arithmetic entry by entry); tile A's Philox block under the loads ----
    const bool have_tb = tb < pre_nt;
    const bool wave_has = wave1 * 64 < pre_nt;   // wave-uniform
    double mb = MP_NEG_INF;
    u64 Wb = 0ull, W2b = 0ull;
    if (wave_has) {
        mb = have_tb ? pre_tm[tb] : MP_NEG_INF;
        Wb = have_tb ? pre_tW[tb] : 0ull;
        W2b = (have_tb && blockIdx.x == 0) ? pre_tW2[tb] : 0ull;
    }
    mp_u64x2 blk;
    blk.a = 0ull; blk.b = 0ull;
    blk = mp_resample_block((a.slot_offset + tileA * TILE + (u64)tb * 2) >> 1, a.rc, (uint32_t)MP_DOM_RESAMPLE, a.k0, a.k1);
    asm volatile("" : "+v"(blk.a), "+v"(blk.b));
    if (wave_has) {
        const double mw = wave_max(mb);
        if (lane1 == 0) s_l1_red[wave1] = mw;
    } else if (lane1 == 0) {
        s_l1_red[wave1] = MP_NEG_INF;
        s_l1_tot[wave1] = 0ull;
        s_l1_tot2[wave1] = 0ull;
    }
    __syncthreads();
    double m = s_l1_red[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) m = fmax(m, s_l1_red[w]);
    u64 incl = 0ull, Tq = 0ull;
    if (wave_has) {
        const bool ok = (m > MP_NEG_INF) && (m < MP_INF);
        const double sc = mp_u2f((u64)(1023 + a.S - FIX_BITS) << 52);  // 2^(S-51)
        const double dm = mb - m;
        Tq = have_tb ? mp_quantize((double)Wb * (ok ? mp_exp_nonpos(dm) : 0.) * sc, 1.0) : 0ull;
        incl = wave_incl_scan_u64(Tq, lane1);
        if (lane1 == 63) s_l1_tot[wave1] = incl;
        if (blockIdx.x == 0) {   // (workgroup-uniform) the scalars of this normalisation: Q2 as well
            const u64 T2 = have_tb ? mp_quantize((double)W2b * (ok ? mp_exp_nonpos(2. * dm) : 0.) * sc, 1.0) : 0ull;
            const u64 tot2 = wave_sum_u64(T2);
            if (lane1 == 0) s_l1_tot2[wave1] = tot2;
        }
    }
    __syncthreads();
    {
        u64 Qall = 0;
        if (wave_has || (blockIdx.x == 0 && threadIdx.x == 0)) {
            u64 woff = 0;
#pragma unroll
            for (int k = 0; k < NWV; ++k) {
                const u64 tk = s_l1_tot[k];
                if (k < wave1) woff += tk;
                Qall += tk;
            }
            if (have_tb) {
                s_incl[tb] = woff + incl;
                s_W[tb] = Wb;
                s_ratio[tb] = (double)Wb / (double)Tq;
            }
        }
        if (blockIdx.x == 0 && threadIdx.x == 0 && !(a.flags & MP_MT_REPLAY)) {
            u64 Q2all = 0;
            for (int k = 0; k < NWV; ++k) Q2all += s_l1_tot2[k];
            fold_scalars(mp_as_global(a.scal), Qall, Q2all, a.S, m, a.n_global, 0);
        }
    }
    __syncthreads();
    MP_STAMP_L(17, 0);
    const u64 Q = s_incl[pre_nt - 1];
    const double nt_over_Q = (double)pre_nt / (double)Q;   // only a starting guess for the tile walk: no effect on results

    // ---- the two tiles, pipelined ----
    mp_mt_tile A, B;
    mp_mt_walk WK;
    mp_mt_out oA, oB;
    double zA[2 * NS], zB[2 * NS], px0[2];
    // (a workgroup without a second tile — the last one of an odd number of tiles — draws and gathers tile A's slots twice and
    // drops the copy: every memory operation up to B's parents is then straight-line code, whose waits the compiler counts exactly)
    const u64 tileBe = hasB ? tileB : tileA;
    mt_draw<WALKB>(A, tileA, blk, a, pre_nt, s_incl, s_W, s_ratio, Q, nt_over_Q);
    if constexpr (MP_MT_ORDER <= 1) {
        blk = mp_resample_block((a.slot_offset + tileBe * TILE + (u64)tb * 2) >> 1, a.rc, (uint32_t)MP_DOM_RESAMPLE, a.k0, a.k1);
        mt_draw<WALKB>(B, tileBe, blk, a, pre_nt, s_incl, s_W, s_ratio, Q, nt_over_Q);
    }
    MP_STAMP_L(18, 0);
    if constexpr (MP_MT_ORDER == 0 || MP_MT_ORDER == 3) mt_deviates<Model>(model, ns, a, A.base, tileA, s_it[wave1], zA);   // under the guide gathers
    MP_STAMP_L(19, 0);
    asm volatile("" : "+v"(A.g[0]), "+v"(A.g[1]));
    mt_rows(A, a);
    MP_STAMP_L(2, 0);
    if constexpr (MP_MT_ORDER >= 2) {   // B's draws (and its guide gathers) only now: A's row gathers are asked for as early as they can be
        blk = mp_resample_block((a.slot_offset + tileBe * TILE + (u64)tb * 2) >> 1, a.rc, (uint32_t)MP_DOM_RESAMPLE, a.k0, a.k1);
        mt_draw<WALKB>(B, tileBe, blk, a, pre_nt, s_incl, s_W, s_ratio, Q, nt_over_Q);
    }
    if constexpr (MP_MT_ORDER == 1 || MP_MT_ORDER == 2) mt_deviates<Model>(model, ns, a, A.base, tileA, s_it[wave1], zA);
    mt_deviates<Model>(model, ns, a, B.base, tileBe, s_it[wave1], zB);                                  // under A's row gathers
    MP_STAMP_L(22, 0);
    mt_resolve_first(A, a, WK);           // A's rows have landed: parents, first walk loads
#if MP_MT_WALK_BARRIER
    // every wave's walks of A are over before any wave asks for B's rows: the CU serves vector-memory operations in order, ACROSS
    // waves, so a late wave's walk loads would otherwise wait behind the early waves' 64-lane gathers for B (and A's normalisation,
    // behind its first barrier, for that wave)
    mt_resolve_rest<WALKB>(A, a, WK, oA.par, px0);
    MP_STAMP_L(23, 0);
    mt_lds_barrier();
    asm volatile("" : "+v"(B.g[0]), "+v"(B.g[1]));
    mt_rows(B, a);
#else
    asm volatile("" : "+v"(B.g[0]), "+v"(B.g[1]));   // (B's start rows are computed here, not hoisted to where its guide cells were asked for)
    mt_rows(B, a);                        // B's row pairs go out behind them
    MP_STAMP_L(23, 0);
    mt_resolve_rest<WALKB>(A, a, WK, oA.par, px0);
#endif
    MP_STAMP_L(25, 0);
    mt_model<Model>(model, a, obs, A.base, px0, zA, oA.lw, oA.xv);
    MP_STAMP_L(26, 0);
    const bool replay = (a.flags & MP_MT_REPLAY) != 0;   // workgroup-uniform
    if (!replay) mt_norm_compute(oA.lw, a.n, tileA, s_norm, s_guideA, oA.cum, oA.m, oA.W, oA.W2);   // LDS and registers only: under B's row gathers
    MP_STAMP_L(20, 0);
    mt_resolve_first(B, a, WK);
    mt_resolve_rest<WALKB>(B, a, WK, oB.par, px0);
    if (replay) {
        mt_store_replay(a, tileA, oA);
        if (hasB) {
            mt_model<Model>(model, a, obs, B.base, px0, zB, oB.lw, oB.xv);
            mt_store_replay(a, tileB, oB);
        }
    } else if (hasB) {
        MP_STAMP_L(27, 0);
        mt_store_tile(a, tileA, oA);      // the vector-memory path is free again: A's stores under B's arithmetic
        mt_model<Model>(model, a, obs, B.base, px0, zB, oB.lw, oB.xv);
        MP_STAMP_L(28, 0);
        mt_norm_compute(oB.lw, a.n, tileB, s_norm, s_guideB, oB.cum, oB.m, oB.W, oB.W2);
        mt_store_tile(a, tileB, oB);
        __syncthreads();                  // both guides are complete
        mt_store_guide(a, tileA, s_guideA);
        mt_store_guide(a, tileB, s_guideB);
    } else {
        mt_store_tile(a, tileA, oA);
        __syncthreads();
        mt_store_guide(a, tileA, s_guideA);
    }
    if ((a.flags & MP_MT_PEEK) && !replay) mt_peek_tail(a, pre_nt);   // (workgroup-uniform)
    MP_STAMP_L(4, 0); MP_STAMP_L(5, 1);
    MP_STAMP_L_FLUSH(0);
}
