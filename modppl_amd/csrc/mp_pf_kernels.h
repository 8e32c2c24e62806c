// mp_pf_kernels.h — device code of the single-GPU particle-filter step (included by mp_pf.hip only): level 0 / level 1
// of the normalisation, K1 k_propagate, the single-kernel resample K3, the XCD-binned resample K3a / K3b.
// Spec: DESIGN.md §4; CPU restatement: oracle/src/inference.hpp.
#pragma once
// ---------------------------------------------------------------------------------------------
// diagnostic stamps (tools/stamp_probe.py; compiled only with -DMP_STAMPS into libmodppl_hip_stamps.so, never into the
// product library): wave 0 of every workgroup records shader-clock / 100 MHz real-time stamps into a buffer of its own.
// ---------------------------------------------------------------------------------------------
#ifdef MP_STAMPS
constexpr int MP_STAMP_MAX_WG = 16384, MP_STAMP_SLOTS = 32, MP_STAMP_KERNELS = 4;
__device__ unsigned long long* g_mp_stamp_buf = nullptr;
__device__ __forceinline__ void mp_stamp(int kernel, int slot, int what /*0 shader clock, 1 real time, 2 hw id*/) {
    if (threadIdx.x == 0 && g_mp_stamp_buf && blockIdx.x < MP_STAMP_MAX_WG) {
        unsigned long long t = 0;
        if (what == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        else if (what == 1) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        else {
            unsigned int a, b;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(a), "=s"(b));
            t = ((unsigned long long)b << 32) | a;
        }
        g_mp_stamp_buf[((size_t)kernel * MP_STAMP_MAX_WG + blockIdx.x) * MP_STAMP_SLOTS + slot] = t;
    }
}
#define MP_STAMP(k, s, w) mp_stamp(k, s, w)
// The same stamps parked in LDS and written out at the kernel's end (k_propagate_mt): a stamp that is a global store queues
// behind whatever the CU's vector-memory path is busy with — 4096 row gathers, in that kernel — and measures that queue, not the phase.
__device__ __forceinline__ void mp_stamp_lds(unsigned long long* s, int slot, int what) {
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        if (what == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        else if (what == 1) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        else {
            unsigned int a, b;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(a), "=s"(b));
            t = ((unsigned long long)b << 32) | a;
        }
        s[slot] = t;
    }
}
__device__ __forceinline__ void mp_stamp_flush(int kernel, const unsigned long long* s) {
    if (threadIdx.x == 0 && g_mp_stamp_buf && blockIdx.x < MP_STAMP_MAX_WG)
        for (int i = 0; i < MP_STAMP_SLOTS; ++i) g_mp_stamp_buf[((size_t)kernel * MP_STAMP_MAX_WG + blockIdx.x) * MP_STAMP_SLOTS + i] = s[i];
}
#define MP_STAMP_L(s, w) mp_stamp_lds(s_stamps, s, w)
#define MP_STAMP_L_DECL __shared__ unsigned long long s_stamps[MP_STAMP_SLOTS]; if (threadIdx.x < MP_STAMP_SLOTS) s_stamps[threadIdx.x] = 0ull
#define MP_STAMP_L_FLUSH(k) mp_stamp_flush(k, s_stamps)
#else
#define MP_STAMP(k, s, w) do { } while (0)
#define MP_STAMP_L(s, w) do { } while (0)
#define MP_STAMP_L_DECL do { } while (0)
#define MP_STAMP_L_FLUSH(k) do { } while (0)
#endif

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
    int* host_flag;    // host-mapped mirror of `degenerate` (set with a write-through store by whoever sets it): mp_pf_synchronize
                       // then needs no copy of this struct to report it
    struct mp_host_mirror* mirror;   // host-mapped: what a synchronous caller asks for after every resample (below); null = none
    u64 folds;         // resamples folded into these scalars so far (mode-0 folds)
};
// What `resample() -> f64` and `effective_sample_size()` (particle_filter.rs:98-116) hand back to the host, in host-mapped memory
// written with system-scope stores by the thread that computes it: the host polls a sequence word instead of enqueueing a copy
// of mp_dev_scalars and waiting for the stream to drain (16 us per call, and a resample whose draws the next step would have
// made had to be drawn by a launch of its own first).  Data first, sequence word last (release).
struct mp_host_mirror {
    unsigned long long fold_seq;    // == mp_dev_scalars::folds once the values below belong to that resample
    double L, ess_stale, log_ml;
    unsigned long long peek_seq;    // k_peek_level1: the number the host passed, once peek_* are valid
    double peek_L, peek_ess;
    int peek_degenerate, pad;
};
// `degenerate` goes up, on the device and in the host's mirror
// The sequence word of a host-mapped record, AFTER the record's fields.  Those are system-scope atomic stores — write-through, counted by
// vmcnt until they are visible to the host — so waiting for them IS the release the host's acquire needs; a release fence at system
// scope would also write the XCD's whole L2 back (buffer_wbl2 sc0 sc1: up to 4 MB of freshly stored rows) — 1.2 us in the LAST workgroup
// of a synchronous step (mt_peek_tail, k_peek_level1), which the host is waiting for.  Only for records written with system-scope
// stores by the calling thread.
__device__ __forceinline__ void mp_st_sys_seq(unsigned long long* seq, unsigned long long v) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(seq, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void mp_flag_degenerate(mp_dev_scalars* scal) {
    scal->degenerate = 1;
    if (scal->host_flag) __hip_atomic_store(scal->host_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Address spaces.  A pointer that the kernel READS FROM MEMORY (a field of mp_k1_tail / mp_k1_draw) is a generic pointer to
// the compiler: every access through it becomes a flat_* instruction, which ticks BOTH vmcnt and lgkmcnt, completes out of
// order and must be waited for with vmcnt(0) lgkmcnt(0) before the next LDS operation — the table copy of a drawing
// k_propagate was four serialised round trips that way.  mp_as_global says what is true anyway (these are hipMalloc'ed
// buffers): the round trip through address space 1 lets the address-space inference turn the accesses into global_* (vmcnt
// only, in order).  mp_ld_const reads a launch-constant struct at a wave-uniform address through the scalar cache (s_load)
// instead of one vector load per lane.
template <class T>
__device__ __forceinline__ T* mp_as_global(T* p) {
    return (T*)(__attribute__((address_space(1))) T*)p;
}
template <class T>
__device__ __forceinline__ T mp_ld_const(const T* p) {
    T v;
    __builtin_memcpy(&v, (const __attribute__((address_space(4))) T*)p, sizeof(T));
    return v;
}

// Streaming outputs of a step (states, log-weights, table rows, guide, draws) as WRITE-THROUGH stores (sc1): the bytes leave
// the XCD's L2 while the kernel still computes, instead of sitting there dirty until the end-of-kernel release writes all of
// them back at once (the next launch cannot start before that: MI355X_MICROARCH.md, "dependent kernel boundary ... + B / 6 TB/s
// when the predecessor leaves B bytes dirty").  MP_WT_STORES=0 builds the plain stores (A/B).
#ifndef MP_WT_STORES
#define MP_WT_STORES 0   // bit mask (A/B builds): 1 log-weights, 2 states, 4 table rows, 8 guide.  Measured with all of them (and the
                         // draws' stores, since removed): 44.4 -> 46.5 us per step — the write traffic competes with the gathers
#endif
#ifndef MP_NT_STORES
#define MP_NT_STORES 0   // same bit mask: non-temporal stores (streamed through the L2, first in line for eviction)
#endif
template <int BIT>
__device__ __forceinline__ void mp_st_stream(double* p, double v) {
    if constexpr ((MP_WT_STORES & BIT) != 0) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if constexpr ((MP_NT_STORES & BIT) != 0) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <int BIT>
__device__ __forceinline__ void mp_st_stream(uint32_t* p, uint32_t v) {
    if constexpr ((MP_WT_STORES & BIT) != 0) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if constexpr ((MP_NT_STORES & BIT) != 0) __builtin_nontemporal_store(v, p);
    else *p = v;
}
typedef u64 mp_u64v2_ __attribute__((ext_vector_type(2)));
template <int BIT>
__device__ __forceinline__ void mp_st_stream16(void* p, mp_u64v2_ v) {
    // (s_nop: a store of more than 64 bits must not be followed at once by a write of its data registers — a hazard the
    // compiler pads for its own stores and cannot see inside inline assembly)
    if constexpr ((MP_WT_STORES & BIT) != 0) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 0" ::"v"(p), "v"(v) : "memory");
    else if constexpr ((MP_NT_STORES & BIT) != 0) __builtin_nontemporal_store(v, reinterpret_cast<mp_u64v2_*>(p));
    else *reinterpret_cast<mp_u64v2_*>(p) = v;
}

constexpr int TILE_THREADS = 512;
constexpr int TILE_ITEMS = 4;
constexpr int TILE = TILE_THREADS * TILE_ITEMS;  // 2048 rows per tile: a constant of the normalisation spec
#ifndef MP_GUIDE_BITS
#define MP_GUIDE_BITS 11
#endif
constexpr int GUIDE_BITS = MP_GUIDE_BITS;        // 11: one guide bucket per table row (GUIDE_N == TILE), 2 B per particle
constexpr int GUIDE_N = 1 << GUIDE_BITS;
static_assert(GUIDE_N <= TILE && GUIDE_N >= 1024, "normalize_tile zeroes / stores the guide with one word of >= 2 bytes per thread");
// (measured at 2^20 particles: 1024 cells per tile 40.2 us per step against 39.1 with 2048 — the guide's smaller footprint in the
// L2s does not pay for the longer walks)
constexpr int GUIDE_DIRECT = 8;                  // bucket runs longer than this are filled by the whole wave
constexpr int FIX_BITS = 51;                     // level-0 fixed point: q = rint(exp(lw - m_tile) * 2^51), 2048 * 2^51 < 2^63
constexpr int DRAW_CHUNK = 1024;                 // output slots per workgroup of k_draw_slots
constexpr int DRAW_THREADS = 512;                // x 2 draws per thread (256 x 4: 20.4 us, 512 x 2: 18.4 us, 1024 x 1: 23.9 us at 2^20, round 1)
constexpr int DRAW_ITEMS = DRAW_CHUNK / DRAW_THREADS;   // = 2: a thread owns two ADJACENT output slots (one Philox block)
static_assert(DRAW_ITEMS == 2, "k_draw_slots: two adjacent draws per thread");
constexpr int K3_THREADS = 256;
constexpr int K3_ITEMS = 4;
constexpr int K3_MAX_BLOCKS = 4096;
constexpr int MAX_TILES = 8192;                  // LDS tile table: 16 B per tile
constexpr int K1_TABLE_LDS_MAX_TILES = 2048;     // k_draw_slots copies the prebuilt tile table (24 B per tile) into LDS up to here, probes it in L2 beyond

// particles per predraw round in k_propagate: 4 pre-drawn (u, r) pairs per lane whatever the model's number of normal sites
template <class Model>
constexpr int k1_items() { return Model::MAX_NORMALS >= 4 ? 1 : 4 / Model::MAX_NORMALS; }

// Wave-wide reductions and scans on the DPP path (row_shr within the 16-lane rows, then row_bcast:15 / row_bcast:31 across
// rows): register-to-register, where __shfl_up / __shfl_xor go through ds_bpermute — an LDS round trip per step of a serial
// chain.  Integer sums and fmax: any order gives the same bits.
template <int CTRL, int ROW_MASK, bool ZERO>
__device__ __forceinline__ u64 mp_dpp_u64(u64 old, u64 v) {
    const int lo = __builtin_amdgcn_update_dpp((int)(uint32_t)old, (int)(uint32_t)v, CTRL, ROW_MASK, 0xF, ZERO);
    const int hi = __builtin_amdgcn_update_dpp((int)(uint32_t)(old >> 32), (int)(uint32_t)(v >> 32), CTRL, ROW_MASK, 0xF, ZERO);
    return ((u64)(uint32_t)hi << 32) | (u64)(uint32_t)lo;
}
__device__ __forceinline__ u64 mp_readlane_u64(u64 v, int l) {
    return ((u64)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32) | (u64)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
}
// inclusive scan across the 64 lanes of a wave (lanes without a source add 0)
__device__ __forceinline__ u64 wave_incl_scan_u64(u64 v, int /*lane*/) {
    v += mp_dpp_u64<0x111, 0xF, true>(0ull, v);   // row_shr:1
    v += mp_dpp_u64<0x112, 0xF, true>(0ull, v);   // row_shr:2
    v += mp_dpp_u64<0x114, 0xF, true>(0ull, v);   // row_shr:4
    v += mp_dpp_u64<0x118, 0xF, true>(0ull, v);   // row_shr:8
    v += mp_dpp_u64<0x142, 0xA, false>(0ull, v);  // row_bcast:15 into rows 1 and 3
    v += mp_dpp_u64<0x143, 0xC, false>(0ull, v);  // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ u64 wave_sum_u64(u64 v) { return mp_readlane_u64(wave_incl_scan_u64(v, 0), 63); }
__device__ __forceinline__ double wave_max(double x) {   // lanes without a source keep their own value
    u64 v = mp_f2u(x);
#define MP_MAX_STEP(CTRL, MASK) v = mp_f2u(fmax(mp_u2f(v), mp_u2f(mp_dpp_u64<CTRL, MASK, false>(v, v))))
    MP_MAX_STEP(0x111, 0xF);
    MP_MAX_STEP(0x112, 0xF);
    MP_MAX_STEP(0x114, 0xF);
    MP_MAX_STEP(0x118, 0xF);
    MP_MAX_STEP(0x142, 0xA);
    MP_MAX_STEP(0x143, 0xC);
#undef MP_MAX_STEP
    return mp_u2f(mp_readlane_u64(v, 63));
}
__device__ __forceinline__ u64 mp_quantize(double e, double scale) {
    const double r = rint(e * scale);
    return (r >= 0.) ? (u64)r : 0ull;  // NaN -> 0
}
// rint(e * 2^51) as an integer for e in [0, 1] (level 0: e = exp(lw - max) <= 1), without rint / f64 -> u64 conversion
// sequences: x + 2^52 rounds x to the nearest-even integer (ulp = 1 in [2^52, 2^53)) and leaves it in the mantissa.
__device__ __forceinline__ u64 mp_quantize51(double e) {
    const double x = e * mp_u2f((u64)(1023 + FIX_BITS) << 52);
    const u64 q = mp_f2u(x + 4503599627370496.0) & 0x000FFFFFFFFFFFFFull;
    return (x >= 0.) ? q : 0ull;  // NaN -> 0
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
// A guide cell is 16 bits and a row index needs 11: bits 11..15 carry WHERE in the cell the start row's cumulative weight lies, in
// 32nds of the cell (round 5).  Cell g covers targets [g << shift, (g + 1) << shift); r0 = its start row; q5 = mp_guide_sub(cum[r0])
// when cum[r0] lies inside the cell and 31 when it lies beyond.  A draw with target t in that cell compares its own 32nd,
// sp = mp_guide_sub(t), with q5:  sp < q5 -> t < cum[r0]: the parent IS r0 (its row is fetched for the state only);  sp > q5 ->
// t > cum[r0]: the walk starts at r0 + 1, r0's row is not fetched at all;  sp == q5 (one draw in ~ 30) -> undecided: r0 and its successor
// as before.  Nothing changes in WHICH row a draw finds — the first row at or after r0 whose cumulative weight reaches the target —,
// only in which rows are asked for on the way.  Readers that do not use the bits take mp_guide_row() of the cell.
constexpr int MP_GUIDE_Q_SHIFT = 11;
static_assert((TILE - 1) < (1 << MP_GUIDE_Q_SHIFT), "a tile-local row index fits below the position bits of a guide cell");
__device__ __forceinline__ uint32_t mp_guide_row(uint32_t cell) { return cell & (uint32_t)(TILE - 1); }
__device__ __forceinline__ uint32_t mp_guide_q5(uint32_t cell) { return cell >> MP_GUIDE_Q_SHIFT; }
__device__ __forceinline__ uint32_t mp_guide_sub(u64 v, int shift) {   // the 32nd of its cell that tile-local value v lies in (cells narrower than 32: the offset itself)
    return shift >= 5 ? (uint32_t)(v >> (shift - 5)) & 31u : (uint32_t)v & ((1u << shift) - 1u);
}

// ---------------------------------------------------------------------------------------------
// level 1 built ONCE per normalisation, by the last workgroup of the level-0 launch to finish (atomic ticket): the
// job's tile table as global arrays, for the draw kernels of the resample that follows (no launch in between).
//   incl[b] = inclusive prefix of T_b;  ratio[b] = (double)W_b / (double)T_b;  head = {m, Q, Q2}.
// Cross-workgroup hand-off inside the launch (cdna_hip_programming.md G16, counter form): every workgroup stores its
// (m_b, W_b, W2_b) write-through (agent-scope atomic stores = sc1), drains them, then takes a ticket with an agent-scope
// atomic add; the workgroup whose add came last reads all of them with agent-scope (sc1) loads after a workgroup barrier.
// Nothing depends on which workgroup that is.  The ticket word is reset by that workgroup for the next launch.
// ---------------------------------------------------------------------------------------------
struct mp_tab_head {
    double m;
    u64 Q, Q2;
    int S, nt;
};
struct mp_tab {               // null ticket = no table is built (sharded handles: the job's tiles live on other ranks)
    unsigned int* ticket;
    u64* incl;
    double* ratio;
    mp_tab_head* head;
    int S;
    u64* W;                   // copy of tile_W as of the build (a k_propagate that DRAWS reads the table while the next generation's tile_W is being written)
};
__device__ __forceinline__ double mp_ld_agent(const double* p) {
    return __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const u64*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ u64 mp_ld_agent(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void mp_st_agent(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<u64*>(p), __builtin_bit_cast(u64, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void mp_st_agent(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t mp_ld_agent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void mp_st_agent(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// by ONE workgroup of THREADS threads; thread t owns the `per` consecutive tiles t*per ..  (three short passes with O(1)
// registers: this code sits in the tail of k_propagate and must not raise its register count)
template <int THREADS>
__device__ __forceinline__ void build_tile_table_global(const double* tile_m, const u64* tile_W, const u64* tile_W2, int nt, const mp_tab& tab) {
    __shared__ double s_tred[THREADS / 64];
    __shared__ u64 s_ttot[THREADS / 64];
    __shared__ u64 s_ttot2[THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (nt + THREADS - 1) / THREADS;
    const int b0 = tid * per;
    if (per == 1) {
        // At most one tile per thread (jobs of up to THREADS x 2048 particles: the bench's 2^20): ONE round of loads, everything
        // in registers.  This workgroup runs after every other one has left, so its duration is added to the kernel's:
        // the general form below (five dependent memory round trips) measured 4.8 us of a 27 us k_propagate.
        const bool have = tid < nt;
        const double mb = have ? mp_ld_agent(tile_m + tid) : MP_NEG_INF;
        const u64 Wb = have ? mp_ld_agent(tile_W + tid) : 0ull;
        const u64 W2b = have ? mp_ld_agent(tile_W2 + tid) : 0ull;
        double m = wave_max(mb);
        if (lane == 0) s_tred[wave] = m;
        __syncthreads();
        m = s_tred[0];
#pragma unroll 1
        for (int w = 1; w < THREADS / 64; ++w) m = fmax(m, s_tred[w]);
        const bool ok = (m > MP_NEG_INF) && (m < MP_INF);
        const double sc = mp_u2f((u64)(1023 + tab.S - FIX_BITS) << 52);  // 2^(S-51)
        const double d = mb - m;
        u64 T = 0, T2 = 0;
#pragma unroll 1
        for (int j = 0; j < 2; ++j) {   // one mp_exp body for T_b and T2_b: the tail of k_propagate stays within its registers
            const double f = ok ? mp_exp(j ? 2. * d : d) : 0.;
            const u64 v = have ? mp_quantize((double)(j ? W2b : Wb) * f * sc, 1.0) : 0ull;
            if (j) T2 = v;
            else T = v;
        }
        const u64 incl = wave_incl_scan_u64(T, lane);
        const u64 tot2 = wave_sum_u64(T2);
        if (lane == 63) s_ttot[wave] = incl;
        if (lane == 0) s_ttot2[wave] = tot2;
        __syncthreads();
        u64 woff = 0, Q = 0, Q2 = 0;
#pragma unroll 1
        for (int k = 0; k < THREADS / 64; ++k) {
            const u64 tk = s_ttot[k];
            if (k < wave) woff += tk;
            Q += tk;
            Q2 += s_ttot2[k];
        }
        if (have) {
            tab.incl[tid] = woff + incl;
            tab.ratio[tid] = (double)Wb / (double)T;
            tab.W[tid] = Wb;
        }
        if (tid == 0) {
            mp_tab_head h;
            h.m = m; h.Q = Q; h.Q2 = Q2; h.S = tab.S; h.nt = nt;
            *tab.head = h;
        }
        return;
    }
    double m = MP_NEG_INF;
#pragma unroll 1
    for (int j = 0; j < per; ++j)
        if (b0 + j < nt) m = fmax(m, mp_ld_agent(tile_m + b0 + j));
    m = wave_max(m);
    if (lane == 0) s_tred[wave] = m;
    __syncthreads();
    m = s_tred[0];
#pragma unroll 1
    for (int w = 1; w < THREADS / 64; ++w) m = fmax(m, s_tred[w]);
    const bool ok = (m > MP_NEG_INF) && (m < MP_INF);
    const double sc = mp_u2f((u64)(1023 + tab.S - FIX_BITS) << 52);  // 2^(S-51)
    u64 run = 0, run2 = 0;
#pragma unroll 1
    for (int j = 0; j < 2 * per; ++j) {   // even j: T_b, odd j: T2_b — one mp_exp body, so the tail of k_propagate stays within its registers
        const int b = b0 + (j >> 1);
        if (b < nt) {
            const bool second = (j & 1) != 0;
            const double d = mp_ld_agent(tile_m + b) - m;
            const double f = ok ? mp_exp(second ? 2. * d : d) : 0.;
            const u64 T = mp_quantize((double)mp_ld_agent((second ? tile_W2 : tile_W) + b) * f * sc, 1.0);
            if (second) {
                run2 += T;
            } else {
                tab.incl[b] = T;   // parked; this thread turns it into the prefix below
                run += T;
            }
        }
    }
    const u64 incl = wave_incl_scan_u64(run, lane);
    const u64 tot2 = wave_sum_u64(run2);
    if (lane == 63) s_ttot[wave] = incl;
    if (lane == 0) s_ttot2[wave] = tot2;
    __syncthreads();
    u64 woff = 0, Q = 0, Q2 = 0;
#pragma unroll 1
    for (int k = 0; k < THREADS / 64; ++k) {
        if (k < wave) woff += s_ttot[k];
        Q += s_ttot[k];
        Q2 += s_ttot2[k];
    }
    u64 cum = woff + (incl - run);
#pragma unroll 1
    for (int j = 0; j < per; ++j) {
        if (b0 + j < nt) {
            const u64 T = tab.incl[b0 + j];
            cum += T;
            tab.incl[b0 + j] = cum;
            const u64 Wb = mp_ld_agent(tile_W + b0 + j);
            tab.ratio[b0 + j] = (double)Wb / (double)T;
            tab.W[b0 + j] = Wb;
        }
    }
    if (tid == 0) {
        mp_tab_head h;
        h.m = m; h.Q = Q; h.Q2 = Q2; h.S = tab.S; h.nt = nt;
        *tab.head = h;
    }
}

// ---------------------------------------------------------------------------------------------
// level 0 of the normalisation for ONE tile, by the workgroup (THREADS threads) that owns it.
// Thread t holds rows tile*2048 + ITEMS*t ..: log-weights lw[] and first state components xv[].
//   m_b = max lw;  a = mp_exp(lw - m_b);  q = rint(a * 2^51);  rows = tile-local inclusive prefix;  W_b, W2_b;  guide.
// ---------------------------------------------------------------------------------------------
// mp_exp for arguments <= 0 (level 0: lw - max): the overflow arms of mp_exp cannot be taken; same bits for every x <= 0 and NaN
__device__ __forceinline__ double mp_exp_nonpos(double x) {
    if (x != x) return x;
    if (x < -745.1332191019412) return 0.0;
    const double INV_LN2 = 1.4426950408889634;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    const double kf = rint(x * INV_LN2);
    double r = fma(-kf, LN2_HI, x);
    r = fma(-kf, LN2_LO, r);
    double p = 1.6059043836821613e-10;
    p = fma(p, r, 2.08767569878681e-09);
    p = fma(p, r, 2.505210838544172e-08);
    p = fma(p, r, 2.755731922398589e-07);
    p = fma(p, r, 2.7557319223985893e-06);
    p = fma(p, r, 2.48015873015873e-05);
    p = fma(p, r, 1.984126984126984e-04);
    p = fma(p, r, 1.388888888888889e-03);
    p = fma(p, r, 8.333333333333333e-03);
    p = fma(p, r, 4.1666666666666664e-02);
    p = fma(p, r, 1.6666666666666666e-01);
    p = fma(p, r, 0.5);
    const double y = 1.0 + fma(r * r, p, r);
    const int k = (int)kf;   // <= 0
    if (k < -1022) return y * mp_u2f((uint64_t)(k + 54 + 1023) << 52) * 5.551115123125783e-17;  // 2^-54
    return y * mp_u2f((uint64_t)(k + 1023) << 52);
}

// A barrier for data exchanged through LDS only: __syncthreads() is also a workgroup-scope fence for global memory, i.e. an
// s_waitcnt vmcnt(0) in front of the s_barrier — every barrier of level 0 then waits for the log-weight / state / parent stores
// the workgroup has just issued (and for whatever gathers are still in flight) before a single LDS word changes hands.
#ifndef MP_NORM_LDS_BARRIER
#define MP_NORM_LDS_BARRIER 0
#endif
__device__ __forceinline__ void mp_lds_barrier() {
    if constexpr (MP_NORM_LDS_BARRIER) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else __syncthreads();
}
template <int THREADS>
__device__ __forceinline__ void normalize_tile(const double (&lw)[TILE / THREADS], const double (&xv)[TILE / THREADS], u64 n, u64 tile,
                                               mp_cx* __restrict__ cx, unsigned short* __restrict__ guide,
                                               double* tile_m, u64* tile_W, u64* tile_W2, const mp_tab& tab) {
    constexpr int ITEMS_ = TILE / THREADS;   // 512 x 4 or 1024 x 2: a tile is always 2048 consecutive slots
    __shared__ double s_red[THREADS / 64];
    __shared__ u64 s_wsum[THREADS / 64];
    __shared__ u64 s_wsum2[THREADS / 64];
    __shared__ int s_last;
    __shared__ __attribute__((aligned(16))) unsigned short s_guide[GUIDE_N];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 base = tile * TILE + (u64)tid * ITEMS_;

    double m = MP_NEG_INF;
#pragma unroll
    for (int j = 0; j < ITEMS_; ++j)
        if (base + j < n) m = fmax(m, lw[j]);
    m = wave_max(m);
    MP_STAMP(0, 8, 0);
    if (lane == 0) s_red[wave] = m;
    // THREADS x (16 | 8 | 4) B = the whole guide
    // THREADS words of GUIDE_N * 2 / THREADS bytes each = the whole guide
    constexpr int GW = GUIDE_N * 2 / THREADS;   // bytes per thread: 16 | 8 | 4 | 2
    if constexpr (GW == 16) reinterpret_cast<uint4*>(s_guide)[tid] = make_uint4(0u, 0u, 0u, 0u);
    else if constexpr (GW == 8) reinterpret_cast<u64*>(s_guide)[tid] = 0ull;
    else if constexpr (GW == 4) reinterpret_cast<uint32_t*>(s_guide)[tid] = 0u;
    else s_guide[tid] = 0;
    mp_lds_barrier();
    MP_STAMP(0, 9, 0);
    m = s_red[0];
#pragma unroll
    for (int w = 1; w < THREADS / 64; ++w) m = fmax(m, s_red[w]);
    const bool ok = (m > MP_NEG_INF) && (m < MP_INF);

    u64 c[ITEMS_];
    u64 run = 0, run2 = 0;
#pragma unroll
    for (int j = 0; j < ITEMS_; ++j) {
        const bool live = ok && (base + j < n);
        const double a = live ? mp_exp_nonpos(lw[j] - m) : 0.;
        run += mp_quantize51(a);
        run2 += mp_quantize51(a * a);
        c[j] = run;
    }
    MP_STAMP(0, 10, 0);
    const u64 incl = wave_incl_scan_u64(run, lane);
    MP_STAMP(0, 11, 0);
    const u64 wtot2 = wave_sum_u64(run2);   // (an LDS atomic per lane instead costs ~10 cycles per LANE: measured 12 k cycles for this tile)
    if (lane == 63) s_wsum[wave] = incl;
    if (lane == 0) s_wsum2[wave] = wtot2;
    mp_lds_barrier();
    MP_STAMP(0, 12, 0);
    // cross-wave offsets on the scalar unit: the per-wave totals and `wave` are wave-uniform, so the sums need no VALU issue
    u64 woff = 0, W = 0;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
    for (int k = 0; k < THREADS / 64; ++k) {
        const u64 vv = s_wsum[k];
        const u64 v = ((u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(vv >> 32)) << 32) | (u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)vv);
        if (k < wave_s) woff += v;
        W += v;
    }
    const u64 off = woff + (incl - run);
    MP_STAMP(0, 7, 0);
    // the tile's scalars go out (and the ticket is taken) now: its round trip is covered by the row stores and the guide
    unsigned int my_ticket = 0;
    if (tid == 0) {
        u64 t2 = 0;
#pragma unroll
        for (int k = 0; k < THREADS / 64; ++k) t2 += s_wsum2[k];
        if (tab.ticket) {
            mp_st_agent(tile_m + tile, m);
            mp_st_agent(tile_W + tile, W);
            mp_st_agent(tile_W2 + tile, t2);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores are out before the ticket says so
            my_ticket = __hip_atomic_fetch_add(tab.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            tile_m[tile] = m;
            tile_W[tile] = W;
            tile_W2[tile] = t2;
        }
    }
#pragma unroll
    for (int j = 0; j < ITEMS_; ++j) {
        if (base + j < n) {
            mp_u64v2_ row;
            row.x = off + c[j];
            row.y = mp_f2u(xv[j]);
            mp_st_stream16<4>(cx + base + j, row);
        }
    }

    MP_STAMP(0, 13, 0);
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
            // (mp_guide_q5: the row ends inside cell g_hi, at 32nd mp_guide_sub(cur), and beyond every earlier cell it covers)
            const unsigned short idx = (unsigned short)((tid * ITEMS_ + j) | (31 << MP_GUIDE_Q_SHIFT));
            const unsigned short idx_hi = (unsigned short)((tid * ITEMS_ + j) | (mp_guide_sub(cur, shift) << MP_GUIDE_Q_SHIFT));
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
    // wave-cooperative fill of long runs (a particle holding a large share of the tile's weight)
    u64 pending = __ballot(long_hi >= long_lo);
    while (pending) {
        const int leader = __ffsll((long long)pending) - 1;
        const int lo = __shfl(long_lo, leader, 64), hi = __shfl(long_hi, leader, 64), jj = __shfl(long_j, leader, 64);
        for (int g = lo + lane; g <= hi; g += 64) s_guide[g] = (unsigned short)jj;
        pending &= pending - 1;
    }
    MP_STAMP(0, 14, 0);
    if (tid == 0) s_last = (tab.ticket != nullptr && my_ticket == gridDim.x - 1u) ? 1 : 0;
    mp_lds_barrier();
    MP_STAMP(0, 15, 0);
    if constexpr (GW == 16) reinterpret_cast<uint4*>(guide + tile * GUIDE_N)[tid] = reinterpret_cast<const uint4*>(s_guide)[tid];
    else if constexpr (GW == 8) reinterpret_cast<u64*>(guide + tile * GUIDE_N)[tid] = reinterpret_cast<const u64*>(s_guide)[tid];
    else if constexpr (GW == 4) mp_st_stream<8>(reinterpret_cast<uint32_t*>(guide + tile * GUIDE_N) + tid, reinterpret_cast<const uint32_t*>(s_guide)[tid]);
    else guide[tile * GUIDE_N + tid] = s_guide[tid];
    if (s_last) {   // workgroup-uniform: every other workgroup's scalars are out (their tickets precede ours)
#ifndef MP_TEST_NOTABLE
        build_tile_table_global<THREADS>(tile_m, tile_W, tile_W2, (int)gridDim.x, tab);
#endif
        if (tid == 0) __hip_atomic_store(tab.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// standalone form: used when the log-weights changed without a propagate (after a resample, before a query or a
// second resample)
__global__ __launch_bounds__(TILE_THREADS) void k_normalize_tiles(const double* __restrict__ logw, const double* __restrict__ x0, int D, u64 n,
                                                                  mp_cx* __restrict__ cx, unsigned short* __restrict__ guide,
                                                                  double* tile_m, u64* tile_W, u64* tile_W2, mp_tab tab) {
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
    normalize_tile<TILE_THREADS>(lw, xv, n, blockIdx.x, cx, guide, tile_m, tile_W, tile_W2, tab);
}

// Slot-order states of a filter with dim_state 1 out of its row table (k_propagate does not store them separately)
__global__ void k_rows_to_x(u64 n, const mp_cx* __restrict__ cx, double* __restrict__ x) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = cx[i].x0;
}

// The job's tile table as a launch of its own: for handles whose level-0 launches build none (drawing k_propagates build it
// per workgroup in LDS) when something else wants it — k_draw_slots of a synchronous or lattice resample.
__global__ __launch_bounds__(1024) void k_build_table(const double* tile_m, const u64* tile_W, const u64* tile_W2, int nt, mp_tab tab) {
    build_tile_table_global<1024>(tile_m, tile_W, tile_W2, nt, tab);
}

// GenFn::simulate over an Unfold model (dynunfold.rs:22-39): one lane = one trace of n_steps kernel calls, every site
// sampled.  states[i][t][d], obs[i][t][dim_obs] (particle-major, like everything that crosses the ABI).
template <class Model>
__global__ __launch_bounds__(256) void k_simulate(Model model, u64 n, uint32_t k0, uint32_t k1, int n_steps, mp_state0 s0,
                                                  double* __restrict__ states, double* __restrict__ obs) {
    constexpr int D = Model::DIM_STATE, DO = Model::DIM_OBS;
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double prev[D], next[D], y[DO];
    for (int d = 0; d < D; ++d) prev[d] = s0.v[d];
    mp_stream rng;
    rng.k0 = k0; rng.k1 = k1; rng.slot = (uint32_t)i;
    for (int t = 0; t < n_steps; ++t) {
        rng.step = (uint32_t)t;
        for (int j = 0; j < DO; ++j) y[j] = 0.;
        mp_simulate_handler<Model> g(rng, y);
        model(g, (long long)t, prev, next);
        double* xs = states + (i * (u64)n_steps + (u64)t) * D;
        double* ys = obs + (i * (u64)n_steps + (u64)t) * DO;
        for (int d = 0; d < D; ++d) { xs[d] = next[d]; prev[d] = next[d]; }
        for (int j = 0; j < DO; ++j) ys[j] = y[j];
    }
}

constexpr uint32_t MP_MAX_ATTEMPTS = 1u << 16;   // exit condition of every retry loop (acceptance pi/4: never reached)
// whether the rejected attempts of a model's deviates are retried by the wave (few sites per particle) or by the lane-local
// queue of k_propagate (many sites per particle: the queue is long enough to even out the attempts)
template <class Model>
constexpr bool mp_coop_model() { return Model::MAX_NORMALS <= 4; }

// ---------------------------------------------------------------------------------------------
// K1: propagate + weight + level 0 of the normalisation, one workgroup per tile
// ---------------------------------------------------------------------------------------------
// A lane owns LANE_ITEMS consecutive particles of its tile.  Standard deviates of the free normal sites come
//   (a) few sites per particle: attempt 0 of every deviate in straight-line code, the rejected ones retried by the wave, or
//   (b) many sites: from a lane-local work queue over the lane's (particle, site) items (~1.6 blocks per item at 16).
// Then the model kernel runs per particle in Generate mode on them.
struct mp_k1_aux {
    mp_tab tab;
    int x_rows;   // (dim_state 1, a plain step) x_in is the current row table: slot i's state is the x0 of row i
};
// A draw of the last resample that has not been looked up yet (k_draw_slots left {tile-local target, start row} per output
// slot): its parent is the first row of the start row's tile, at or after it, whose cumulative weight reaches the target
// (categorical.rs:23-37 restricted to the tile).  Start row and successor go out together, whole 16-byte rows.
typedef u64 mp_u64v2 __attribute__((ext_vector_type(2)));   // one table row as it is loaded: {cum, bits of x0}
__device__ __forceinline__ mp_u64v2 mp_ld_row(const mp_cx* p) { return *reinterpret_cast<const mp_u64v2*>(p); }
__device__ __forceinline__ void mp_pin_rows(mp_u64v2& a, mp_u64v2& b2) { asm volatile("" : "+v"(a), "+v"(b2)::"memory"); }
__device__ __forceinline__ void mp_pin3(int& a, u64& b2, uint32_t& c) { asm volatile("" : "+v"(a), "+v"(b2), "+v"(c)::"memory"); }
__device__ __forceinline__ u64 mp_tile_last(uint32_t row, u64 n) {
    const u64 tend = (((u64)row / TILE) + 1) * TILE;
    return (tend < n ? tend : n) - 1;      // last row of the tile
}
// A slot of a SHARDED filter whose offspring arrives from another rank has no draw to look up: its entry in the start-row array
// is MP_DRAW_RECV | the index of its row {state[dim], parent's global slot id} in the exchange buffer (`rows`, rows of `rw`
// doubles); the "parent" handed on keeps the flag, so that the consumer knows where the state is.
constexpr uint32_t MP_DRAW_RECV = 0x80000000u;
// (round 5) a draw whose guide cell has already decided it (mp_guide_q5 against the target's own 32nd of the cell, mp_start_row below):
// the start row IS the parent — its row is fetched only by consumers that want the state out of it (dim_state 1), not to be compared.
// Handles hold at most 2^24 rows (MAX_TILES), so bit 30 of a start row is free.
constexpr uint32_t MP_DRAW_SURE0 = 0x40000000u;
// what a draw kernel leaves as a draw's start row, from its guide cell: r0 | MP_DRAW_SURE0 when the target lies below cum[r0] (the
// parent is r0), r0 + 1 when it lies above (cum[r0] < target: the forward scan may start one row on), r0 when the two fall into the
// same 32nd of the cell.  The scan `first row at or after the start row whose cumulative weight reaches the target` finds the same
// parent from either start.
__device__ __forceinline__ uint32_t mp_start_row(uint32_t cell, uint32_t sub, uint32_t tbase, uint32_t tlen) {
    const uint32_t j0 = mp_guide_row(cell), q5 = mp_guide_q5(cell);
    const uint32_t j = j0 > tlen - 1 ? tlen - 1 : j0;
    if (sub < q5) return (tbase + j) | MP_DRAW_SURE0;
    return tbase + ((sub > q5 && j + 1 < tlen) ? j + 1 : j);
}
#ifndef MP_PAIR_SAME_LINE
#define MP_PAIR_SAME_LINE 1
#endif
// N draws at a time: every load goes out before the first is used.
// BISECT: a walk that is still going after MP_WALK_LINEAR rows finishes by bisection over the rest of the tile (the rows' cumulative
// weights ascend: "first row at or after the start row whose cumulative weight reaches the target" is a lower bound, the same row
// either way).  Walks are 0 - 2 rows long when a tile's weights are of one order of magnitude; when a few particles carry a tile
// (d = 16: an ESS of a few hundred out of 2^21) the guide cell that holds the light rows in front of a heavy one holds hundreds of
// them, 1 / 1024 of the tile's draws start there, and under a lattice scheme those draws are CONSECUTIVE slots — the four slots of
// one lane walk a thousand dependent loads each, one after the other, and their workgroup ends 100 - 180 us after everybody else
// (C5 under systematic resampling: 565 us per step against 410 with flat weights -> 435 with the bisection).
// The bisection is not free where walks are short (same box: C3 + 5 %, C5 under multinomial draws + 1 %: registers), so the
// propagate kernels carry it in an instantiation of their own (WALKB) that the host launches when the draws being looked up are a
// lattice's and the model is a wide one; the 1024-thread kernels have the same two instantiations and always look deferred draws up
// with the bisecting one (mp_pf.hip launch_propagate says why).
#ifndef MP_WALK_LINEAR
#define MP_WALK_LINEAR 12   // (a bisection is 11 dependent loads: switching to it after 3 rows made every 5 - 8 row walk — one lane in a few
                            //  waves even with healthy weights — cost more than walking on, and the waves of a workgroup wait for it: + 7 - 11 %)
#endif

// NEEDX0: the caller wants the parent's first state component out of its row (dim_state 1); wider states are gathered from x_in by
// the parent's index, and a draw flagged MP_DRAW_SURE0 then needs no row at all
template <int N, bool BISECT = false, bool NEEDX0 = true>
__device__ __forceinline__ void mp_resolve_draws(const mp_cx* __restrict__ cx, u64 n, const u64* lt, const uint32_t* row, uint32_t* parent, double* x0,
                                                 const double* __restrict__ rows = nullptr, int rw = 0) {
    mp_u64v2 a[N], b2[N];
    u64 last[N];
    uint32_t r0[N];
    bool hb[N], sure[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const bool recv = rows && (row[k] & MP_DRAW_RECV);
        sure[k] = !recv && (row[k] & MP_DRAW_SURE0);
        r0[k] = recv ? 0u : (row[k] & ~MP_DRAW_SURE0);
        last[k] = mp_tile_last(r0[k], n);
        if (!NEEDX0 && sure[k]) {   // decided by the guide cell, and nobody wants the row's state: nothing to fetch
            a[k].x = ~0ull; a[k].y = 0ull; b2[k] = a[k];
            hb[k] = false;
            continue;
        }
        a[k] = mp_ld_row(cx + r0[k]);
        if (sure[k]) { a[k].x = ~0ull; b2[k] = a[k]; hb[k] = false; continue; }   // (its cumulative weight reaches the target: mp_start_row)
        // the successor goes out with the start row only where it lies in the SAME 64-byte line (four rows): a successor in the
        // next line is needed by 37 % of the draws only, and fetching it for all of them was a quarter-line-miss per draw for
        // nothing — the lookups are bound by the fabric's miss rate (tools/gather_probe.hip); the few that need it walk on below
        hb[k] = (u64)r0[k] < last[k] && (MP_PAIR_SAME_LINE ? (r0[k] & 3u) != 3u : true);
        b2[k] = mp_ld_row(cx + (u64)r0[k] + (hb[k] ? 1 : 0));
    }
#pragma unroll
    for (int k = 0; k < N; ++k) mp_pin_rows(a[k], b2[k]);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (rows && (row[k] & MP_DRAW_RECV)) {   // (sharded filters only)
            x0[k] = rows[(u64)(row[k] & ~MP_DRAW_RECV) * (u64)rw];
            parent[k] = row[k];
            continue;
        }
        const bool step1 = a[k].x < lt[k] && hb[k];
        u64 p = (u64)r0[k] + (step1 ? 1 : 0);
        mp_u64v2 cur = step1 ? b2[k] : a[k];
        if constexpr (BISECT) {
            int steps = 0;
            while (cur.x < lt[k] && p < last[k]) {   // rare: more than one row past the guide's start
                if (++steps > MP_WALK_LINEAR) {      // very rare: the first row of (p, last] whose cumulative weight reaches the target, or `last`
                    u64 lo = p + 1, hi = last[k];
                    while (lo < hi) {
                        const u64 mid = lo + ((hi - lo) >> 1);
                        if (mp_ld_row(cx + mid).x >= lt[k]) hi = mid;
                        else lo = mid + 1;
                    }
                    p = lo;
                    cur = mp_ld_row(cx + p);
                    break;
                }
                ++p;
                cur = mp_ld_row(cx + p);
            }
        } else {
            while (cur.x < lt[k] && p < last[k]) {   // rare: more than one row past the guide's start
                ++p;
                cur = mp_ld_row(cx + p);
            }
        }
        parent[k] = (uint32_t)p;   // (may alias row[])
        x0[k] = __builtin_bit_cast(double, (u64)cur.y);
    }
}
template <bool BISECT = false, bool NEEDX0 = true>
__device__ __forceinline__ void mp_resolve_draw(const mp_cx* __restrict__ cx, u64 n, u64 lt, uint32_t row, uint32_t* parent, double* x0,
                                                const double* __restrict__ rows = nullptr, int rw = 0) {
    mp_resolve_draws<1, BISECT, NEEDX0>(cx, n, &lt, &row, parent, x0, rows, rw);
}
// What k_propagate needs only in its LAST phase (level 0 / level 1 of the normalisation), constant per handle: kept in
// device memory and read there — as kernel arguments these 11 pointers sat in SGPRs through the whole VALU-bound part of the
// kernel, which ran out of them (78 at 8 waves per SIMD) and spilled to VGPR lanes.
struct mp_k1_tail {
    mp_cx* cx;
    unsigned short* guide;
    double* tile_m;
    u64* tile_W;
    u64* tile_W2;
    mp_tab tab;
};
// What a k_propagate that makes the previous resample's DRAWS itself reads (kernels whose lanes own two adjacent slots = one
// Philox block: k_draw_slots' work, without its launch, its lockstep round and the round trip of its output through memory).
// Everything here belongs to the generation that was resampled and is not written by this launch before its last workgroup
// runs (the table, tab.W) or at all (guide_old: the launch writes the other guide buffer).
struct mp_own_range {   // a sharded filter's own draws of a resample — lattice schemes: g_lo <= g < g_hi of the job's lattice; split multinomial: its c_r draws, [0, c_r)
    u64 g_lo, g_hi;
};
struct mp_k1_draw {
    // the tile scalars of the generation that was resampled: no workgroup of the level-0 launch that wrote them built the job's
    // tile table (mp_tab::ticket was null), every drawing workgroup builds it in LDS from these (buffers this launch does not
    // write: its own tile scalars go to the other set)
    const double* tile_m_old;
    const u64* tile_W_old;
    const u64* tile_W2_old;
    const unsigned short* guide_old;
    mp_dev_scalars* scal;
    uint32_t* parent;         // the parents are written out once the lookups have resolved them (particle_filter.rs:20 keeps `parents`): 4 B per
                              // slot at the end of the lookups, where {target, start row} written during the draws were 12 B per slot of
                              // store traffic in the middle of the phase that is bound by the memory fabric
    u64 n_global;
    int nt, S;
    // SHD launches (a sharded filter's self-drawn resample, mp_pf_shard_kernels.h): this rank's own offspring are [0, g_hi - g_lo) of
    // `shd_range` (device memory: k_shard_table wrote it); the kernel's first three arguments are then this rank's SLICE of the job's
    // tile table {ratio, W, inclusive prefix} instead of tile scalars; shd_head: a world of one folds the scalars here (nobody else did)
    const mp_own_range* shd_range;
    const mp_tab_head* shd_head;
    int shd_rank, shd_world;
};
__device__ __forceinline__ void fold_scalars(mp_dev_scalars* scal, u64 Q, u64 Q2, int S, double m, u64 n_global, int mode);
__device__ __forceinline__ u64 mp_target(u64 k52, u64 Q);
__device__ __forceinline__ uint32_t mp_systematic_k32(uint32_t rc, uint32_t k0, uint32_t k1);
__device__ __forceinline__ u64 mp_target_lattice(int scheme, u64 g, uint32_t shared_k32, uint32_t rc, uint32_t k0, uint32_t k1, u64 Q, u64 n_global);
__device__ __forceinline__ mp_u64x2 mp_resample_block(u64 g_pair, uint32_t rc, uint32_t domain, uint32_t k0, uint32_t k1);
template <bool BS = false>
__device__ __forceinline__ void mp_locate_r(const u64* s_incl, const u64* s_W, const double* s_ratio, uint32_t nt, u64 target, double nt_over_Q,
                                            uint32_t* tile, u64* lt, uint32_t* gslot, uint32_t* sub = nullptr);
// The model kernel in Generate mode for ONE particle (slot i): previous state from wherever the last resample left it, the
// functor with a Generate handler over the deviates zp[0..NS), new state and log-weight out.
template <class Model>
__device__ __forceinline__ void mp_run_particle(const Model& model, u64 n, u64 slot_offset, uint32_t k0, uint32_t k1, long long t,
                                                const double* __restrict__ x_in, double* __restrict__ x_out, double* logw, const double* obs_v,
                                                const double* s0_v, int overwrite, bool deferred, const double* __restrict__ inv_rows,
                                                bool via_inv, uint32_t pmv, const double* x0p, u64 i, const double* zp, double* lw_out, double* x0_out,
                                                bool x_rows = false) {
    constexpr int D = Model::DIM_STATE;
    if (i >= n) return;
    double prev[D], next[D];
    if (via_inv) {
        // the last (sharded) resample left the parents' states where the all-to-all put them: slot i's row {x[0..D), parent
        // id} is row inv[i] (= pmv) of the exchange buffer (inv_rows here); owner-keeps exchange, D > 1: a kept offspring has no row
        // — its entry names the parent's local row of the pre-resample buffer x_in (mp_pf_shard_kernels.h MP_INV_LOCAL)
        const bool local_parent = D > 1 && (pmv & 0x80000000u);
        const double* row = local_parent ? x_in + (u64)(pmv & 0x7FFFFFFFu) * D : inv_rows + (u64)pmv * (u64)(D + 1);
#pragma unroll
        for (int d = 0; d < D; ++d) prev[d] = row[d];
    } else if (deferred) {
        // the last resample only DREW (k_draw_slots) and the caller has looked this lane's draws up (mp_resolve_draws): pmv = the
        // parent, *x0p = its first state component — the clone loop of `resample` (particle_filter.rs:109-114), fused into the
        // step that consumes it
        if constexpr (D == 1) {
            prev[0] = *x0p;                    // the row carries the first state component
        } else {
            // wider states: the parent's (particle-major) row of the pre-resample buffer — or, for a slot of a sharded filter whose
            // offspring came from another rank, its row of the exchange buffer
            const double* src = (inv_rows && (pmv & MP_DRAW_RECV)) ? inv_rows + (u64)(pmv & ~MP_DRAW_RECV) * (u64)(D + 1) : x_in + (u64)pmv * D;
#pragma unroll
            for (int d = 0; d < D; ++d) prev[d] = src[d];
        }
    } else if (t == 0) {   // (two branches, not a select between two loads: that form kept the by-value kernel argument in scratch)
#pragma unroll
        for (int d = 0; d < D; ++d) prev[d] = s0_v[d];
    } else {
        // (d = 1: x_in may be the current row table itself — {cum, x0} pairs: the states of such filters live there, below)
        if constexpr (D == 1) prev[0] = x_rows ? x_in[2 * i + 1] : x_in[i];
        else {
#pragma unroll
            for (int d = 0; d < D; ++d) prev[d] = x_in[i * D + d];
        }
    }
    mp_stream rng;
    rng.k0 = k0; rng.k1 = k1; rng.slot = (uint32_t)(slot_offset + i); rng.step = (uint32_t)t;
    mp_generate_handler<Model> g(rng, obs_v, zp);
    model(g, t, prev, next);
    // d = 1: the state is not stored on its own — it is the x0 of this particle's row of the table that the level-0 pass below
    // writes anyway; who wants slot-order states afterwards copies them out of the rows (mp_pf.hip ensure_x).  8 MB less written
    // per step at 2^20 particles, in a kernel whose lookups are bound by what the L2s can hold and the fabric can move.
    if constexpr (D > 1) {
#pragma unroll
        for (int d = 0; d < D; ++d) x_out[i * D + d] = next[d];
    }
    // particle_filter.rs:68 (init: overwrite) / :81 (accumulate); overwrite == 2: the log-weights are known to be all zero
    // after a resample (log_weights.fill(0.), :114) and are not re-read
    const double w = overwrite == 1 ? g.weight : (overwrite == 2 ? 0. + g.weight : logw[i] + g.weight);
    mp_st_stream<1>(logw + i, w);
    *lw_out = w;
    *x0_out = next[0];
}
// TAB2: the drawing launch's job has more tiles than the workgroup has threads (two table entries per thread): an instantiation
// of its own, so that the headline kernel — 62 of its 64 registers in use — does not carry a second form of the table build
// LAT: the pending draws are a lattice's (systematic / stratified, scheme = drw >> 1): targets from mp_target_lattice instead of a
// Philox block per lane — again an instantiation of its own (its 64-bit divisions would cost the multinomial form registers)
// SHD: the pending draws are a SHARDED filter's self-drawn ones (lattice range or split multinomial, mp_pf_shard_kernels.h): the table
// is this rank's slice of the job's, own offspring p < c_me have their target in closed form, the slots beyond read the row that
// arrived for them (dfr_row: MP_DRAW_RECV | index)
// WALKB: long walks finish by bisection (mp_resolve_draws' BISECT): wide-state kernels looking up a lattice's draws
template <class Model, int THREADS, bool TAB2 = false, bool LAT = false, bool SHD = false, bool WALKB = false>
#ifndef MP_K1_512_WAVES
#define MP_K1_512_WAVES 4   // waves per SIMD the 512-thread kernels are compiled for (A/B builds: 2 = 256 VGPRs, no scratch, one workgroup per CU)
#endif
__global__ __launch_bounds__(THREADS, (THREADS == 1024 ? 8 : (THREADS == 512 ? MP_K1_512_WAVES : 1))) void k_propagate(const double* __restrict__ pre_tm, const u64* __restrict__ pre_tW, const u64* __restrict__ pre_tW2, int pre_nt, int drw,
                                                            Model model, u64 n, u64 slot_offset, uint32_t k0, uint32_t k1,
                                                            long long t, const double* x_in, double* x_out, double* logw,
                                                            mp_obs obs, mp_state0 s0, int overwrite,
                                                            const uint32_t* __restrict__ dfr_row, const double* __restrict__ inv_rows,
                                                            const mp_cx* __restrict__ cx_old, const mp_k1_tail* tail,
                                                            const uint32_t* __restrict__ inv, const u64* __restrict__ dfr_lt, mp_k1_aux aux,
                                                            mp_k1_draw drw_v, uint32_t rc) {
    // (the first five arguments are what a drawing launch needs to get its tile-scalar loads out — preloading the Philox inputs as
    // well, 15 dwords, measured no better —: scalars and pointers at the head of
    // the argument list are preloaded into SGPRs at wave launch (-amdgpu-kernarg-preload-count), so those loads do not wait for
    // the kernel-argument fetch — ~1 us from device memory — that everything else starts with)
    constexpr int D = Model::DIM_STATE;
    constexpr int NS = Model::MAX_NORMALS;
    constexpr int LANE_ITEMS = TILE / THREADS;
    const int ns = model.n_normals(t);  // wave-uniform
    MP_STAMP(0, 0, 0); MP_STAMP(0, 1, 1); MP_STAMP(0, 6, 2);
    const u64 base = (u64)blockIdx.x * TILE + (u64)threadIdx.x * LANE_ITEMS;
    double lw[LANE_ITEMS], xv[LANE_ITEMS];
    // models with few normal sites draw every deviate of the lane first (z[]), then run the model; the others go particle by
    // particle (QUEUE: 64 deviates per lane held across the phases lived in scratch memory — 640 B per thread at d = 16)
    constexpr bool QUEUE = !(mp_coop_model<Model>() && LANE_ITEMS * Model::MAX_NORMALS <= 4);
    double z[QUEUE ? 1 : LANE_ITEMS * NS];
#pragma unroll
    for (int j = 0; j < LANE_ITEMS; ++j) { lw[j] = MP_NEG_INF; xv[j] = 0.; }
    // the first hop of the state fetch goes out before anything else: its latency runs under phase 1
    // (a deferred draw = {start row, tile-local target}; kernels of wider states have no registers to carry the targets
    // through the deviates and read them where they look the draw up)
    constexpr bool LT_LATE = D > 1;
    // a lane of two adjacent slots owns exactly the two draws of one Philox block (the host passes `drw` to these kernels only: jobs of
    // at most 2 * THREADS tiles, whose table — 24 B per tile — fits the launch's dynamic LDS)
    constexpr bool CAN_DRAW = LANE_ITEMS == 2 && THREADS == 1024;
    uint32_t pm[LANE_ITEMS];
    u64 plt[LANE_ITEMS];
    bool drew = false;
    if constexpr (CAN_DRAW) {
        if (drw) {
            // ---- phase 0: the draws of the previous resample for this lane's two adjacent slots (k_draw_slots<1, 0>'s body:
            // same Philox block, same target, same tile walk, same guide cell -> bit-identical {target, start row}) ----
            drew = true;
            // (the struct travels BY VALUE in the kernel arguments: read from device memory it was a second, dependent scalar-load
            // miss — 1.2 k cycles — in front of everything this launch does)
            mp_k1_draw dw = drw_v;
            dw.guide_old = mp_as_global(dw.guide_old); dw.scal = mp_as_global(dw.scal);
            extern __shared__ __attribute__((aligned(16))) unsigned char k1_dyn[];
            u64* s_incl = reinterpret_cast<u64*>(k1_dyn);              // [nt]
            u64* s_W = s_incl + dw.nt;                                  // [nt]
            double* s_ratio = reinterpret_cast<double*>(s_W + dw.nt);  // [nt]
            // (the host lets a launch draw only for jobs of at most THREADS tiles: one table entry per thread, its loads in flight
            // together and the Philox block computed under them)
            const int tb = (int)threadIdx.x;
            const bool have_tb = tb < pre_nt;
            mp_u64x2 blk;
            uint32_t lat_k32 = 0u;
            __shared__ double s_l1_red[THREADS / 64];
            __shared__ u64 s_l1_tot[THREADS / 64], s_l1_tot2[THREADS / 64];
            u64 shd_lo = 0ull, shd_glo = 0ull, shd_cme = 0ull, shd_Q = 0ull;
            if constexpr (SHD) {
                static_assert(!TAB2, "SHD launches: at most one table entry per thread");
                // this rank's slice of the job's table as k_shard_table left it (pre_tm = ratio, pre_tW = W, pre_tW2 = inclusive prefix),
                // rebased to the rank's own share (lo, hi] of the fixed-point mass
                const int nt_sl = dw.nt;
                const mp_own_range* rgp = mp_as_global(dw.shd_range);
                shd_glo = rgp->g_lo;
                shd_cme = rgp->g_hi - shd_glo;
                shd_lo = dw.shd_rank ? pre_tW2[-1] : 0ull;
                shd_Q = pre_tW2[(u64)(dw.shd_world - dw.shd_rank) * (u64)nt_sl - 1];
                if (have_tb) {
                    s_incl[tb] = pre_tW2[tb] - shd_lo;
                    s_W[tb] = pre_tW[tb];
                    s_ratio[tb] = pre_tm[tb];
                }
                if constexpr (LAT) {
                    lat_k32 = (drw >> 1) == 1 ? mp_systematic_k32(rc, k0, k1) : 0u;
                } else {   // the rank's own stream: block (p >> 1), word 3 = the rank (rank 0: the single filter's)
                    blk = mp_philox4x32_10((uint32_t)(base >> 1), rc, (uint32_t)MP_DOM_RESAMPLE << 16, (uint32_t)dw.shd_rank, k0, k1);
                }
                if (dw.shd_head && blockIdx.x == 0 && threadIdx.x == 0) {
                    const mp_tab_head* hd = mp_as_global(dw.shd_head);
                    fold_scalars(dw.scal, hd->Q, hd->Q2, dw.S, hd->m, dw.n_global, 0);
                }
            } else if constexpr (!TAB2) {
                // Level 1 of the normalisation by THIS workgroup, in LDS (build_tile_table_global's arithmetic, entry by entry):
                // no workgroup of the previous launch stayed behind to build the job's table after everybody else had left — that
                // serial tail (a round of remote loads, two barriers, the stores) was 3 - 4 us of every step's kernel.  Here the
                // same work costs each workgroup one mp_exp and one division per THREAD, under the start-up latencies.
                const int lane1 = tb & 63, wave1 = tb >> 6;
                // (only the waves that own table entries do the arithmetic — at 512 tiles half of the workgroup's; the others
                // go straight to the barriers: everything in front of the first gather is on the step's critical path, twice,
                // because the CU's other workgroup is doing the same)
                const bool wave_has = wave1 * 64 < pre_nt;   // wave-uniform
                MP_STAMP(0, 25, 0);
                double mb = MP_NEG_INF;
                u64 Wb = 0ull, W2b = 0ull;
                if (wave_has) {
                    mb = have_tb ? pre_tm[tb] : MP_NEG_INF;
                    Wb = have_tb ? pre_tW[tb] : 0ull;
                    W2b = (have_tb && blockIdx.x == 0) ? pre_tW2[tb] : 0ull;
                }
                MP_STAMP(0, 26, 0);
                if constexpr (LAT) {   // the resample's one shared uniform (systematic); stratified draws one per slot below
                    lat_k32 = (drw >> 1) == 1 ? mp_systematic_k32(rc, k0, k1) : 0u;
                } else {
                    blk = mp_resample_block((slot_offset + base) >> 1, rc, (uint32_t)MP_DOM_RESAMPLE, k0, k1);   // base is even
                    asm volatile("" : "+v"(blk.a), "+v"(blk.b));
                }
                MP_STAMP(0, 27, 0);
                if (wave_has) {
                    const double mw = wave_max(mb);
                    MP_STAMP(0, 28, 0);
                    if (lane1 == 0) s_l1_red[wave1] = mw;
                } else if (lane1 == 0) {
                    s_l1_red[wave1] = MP_NEG_INF;
                    s_l1_tot[wave1] = 0ull;
                    s_l1_tot2[wave1] = 0ull;
                }
                __syncthreads();
                MP_STAMP(0, 29, 0);
                double m = s_l1_red[0];
#pragma unroll
                for (int w = 1; w < THREADS / 64; ++w) m = fmax(m, s_l1_red[w]);
                u64 incl = 0ull;
                u64 T = 0ull;
                if (wave_has) {
                    const bool ok = (m > MP_NEG_INF) && (m < MP_INF);
                    const double sc = mp_u2f((u64)(1023 + dw.S - FIX_BITS) << 52);  // 2^(S-51)
                    const double dm = mb - m;
                    T = have_tb ? mp_quantize((double)Wb * (ok ? mp_exp_nonpos(dm) : 0.) * sc, 1.0) : 0ull;   // (dm <= 0: mp_exp's bits)
                    incl = wave_incl_scan_u64(T, lane1);
                    if (lane1 == 63) s_l1_tot[wave1] = incl;
                    if (blockIdx.x == 0) {   // (workgroup-uniform) the scalars of this normalisation: Q2 as well
                        const u64 T2 = have_tb ? mp_quantize((double)W2b * (ok ? mp_exp_nonpos(2. * dm) : 0.) * sc, 1.0) : 0ull;
                        const u64 tot2 = wave_sum_u64(T2);
                        if (lane1 == 0) s_l1_tot2[wave1] = tot2;
                    }
                }
                __syncthreads();
                u64 Qall = 0;
                if (wave_has || (blockIdx.x == 0 && threadIdx.x == 0)) {
                    u64 woff = 0;
#pragma unroll
                    for (int k = 0; k < THREADS / 64; ++k) {
                        const u64 tk = s_l1_tot[k];
                        if (k < wave1) woff += tk;
                        Qall += tk;
                    }
                    if (have_tb) {
                        s_incl[tb] = woff + incl;
                        s_W[tb] = Wb;
                        s_ratio[tb] = (double)Wb / (double)T;
                    }
                }
                if (blockIdx.x == 0 && threadIdx.x == 0) {
                    u64 Q2all = 0;
                    for (int k = 0; k < THREADS / 64; ++k) Q2all += s_l1_tot2[k];
                    fold_scalars(dw.scal, Qall, Q2all, dw.S, m, dw.n_global, 0);
                }
            } else {
                // THREADS < tiles <= 2 THREADS (2^22 particles): thread t owns the ADJACENT entries 2t, 2t + 1 — same arithmetic entry by
                // entry, a thread-local running sum under the wave scan
                const int lane1 = tb & 63, wave1 = tb >> 6;
                const int e0 = 2 * tb, e1 = 2 * tb + 1;
                const bool h0 = e0 < pre_nt, h1 = e1 < pre_nt;
                const double mb0 = h0 ? pre_tm[e0] : MP_NEG_INF, mb1 = h1 ? pre_tm[e1] : MP_NEG_INF;
                const u64 Wb0 = h0 ? pre_tW[e0] : 0ull, Wb1 = h1 ? pre_tW[e1] : 0ull;
                u64 W2b0 = 0ull, W2b1 = 0ull;
                if (blockIdx.x == 0) { W2b0 = h0 ? pre_tW2[e0] : 0ull; W2b1 = h1 ? pre_tW2[e1] : 0ull; }
                if constexpr (LAT) {   // the resample's one shared uniform (systematic); stratified draws one per slot below
                    lat_k32 = (drw >> 1) == 1 ? mp_systematic_k32(rc, k0, k1) : 0u;
                } else {
                    blk = mp_resample_block((slot_offset + base) >> 1, rc, (uint32_t)MP_DOM_RESAMPLE, k0, k1);   // base is even
                    asm volatile("" : "+v"(blk.a), "+v"(blk.b));
                }
                const double mw = wave_max(fmax(mb0, mb1));
                if (lane1 == 0) s_l1_red[wave1] = mw;
                __syncthreads();
                double m = s_l1_red[0];
#pragma unroll
                for (int w = 1; w < THREADS / 64; ++w) m = fmax(m, s_l1_red[w]);
                const bool ok = (m > MP_NEG_INF) && (m < MP_INF);
                const double sc = mp_u2f((u64)(1023 + dw.S - FIX_BITS) << 52);  // 2^(S-51)
                const u64 T0 = h0 ? mp_quantize((double)Wb0 * (ok ? mp_exp_nonpos(mb0 - m) : 0.) * sc, 1.0) : 0ull;
                const u64 T1 = h1 ? mp_quantize((double)Wb1 * (ok ? mp_exp_nonpos(mb1 - m) : 0.) * sc, 1.0) : 0ull;
                const u64 run = T0 + T1;
                const u64 incl = wave_incl_scan_u64(run, lane1);
                if (lane1 == 63) s_l1_tot[wave1] = incl;
                if (blockIdx.x == 0) {   // (workgroup-uniform) the scalars of this normalisation: Q2 as well
                    const u64 T20 = h0 ? mp_quantize((double)W2b0 * (ok ? mp_exp_nonpos(2. * (mb0 - m)) : 0.) * sc, 1.0) : 0ull;
                    const u64 T21 = h1 ? mp_quantize((double)W2b1 * (ok ? mp_exp_nonpos(2. * (mb1 - m)) : 0.) * sc, 1.0) : 0ull;
                    const u64 tot2 = wave_sum_u64(T20 + T21);
                    if (lane1 == 0) s_l1_tot2[wave1] = tot2;
                }
                __syncthreads();
                u64 woff = 0, Qall = 0;
#pragma unroll
                for (int k = 0; k < THREADS / 64; ++k) {
                    const u64 tk = s_l1_tot[k];
                    if (k < wave1) woff += tk;
                    Qall += tk;
                }
                const u64 before = woff + (incl - run);
                if (h0) { s_incl[e0] = before + T0; s_W[e0] = Wb0; s_ratio[e0] = (double)Wb0 / (double)T0; }
                if (h1) { s_incl[e1] = before + run; s_W[e1] = Wb1; s_ratio[e1] = (double)Wb1 / (double)T1; }
                if (blockIdx.x == 0 && threadIdx.x == 0) {
                    u64 Q2all = 0;
                    for (int k = 0; k < THREADS / 64; ++k) Q2all += s_l1_tot2[k];
                    fold_scalars(dw.scal, Qall, Q2all, dw.S, m, dw.n_global, 0);
                }
            }
            MP_STAMP(0, 16, 0);
            __syncthreads();
            MP_STAMP(0, 17, 0);
            const u64 Q = s_incl[dw.nt - 1];   // (SHD: the rank's own share, hi - lo)
            const double nt_over_Q = (double)dw.nt / (double)Q;   // only a starting guess for the tile walk: no effect on results
            uint32_t gslot[2], tile_of[2], sp[2];
#pragma unroll
            for (int q = 0; q < 2; ++q)
            {
                u64 tg;
                if constexpr (SHD) {
                    const bool own = base + q < shd_cme;   // (a slot beyond the rank's own offspring: any in-range target; overwritten below)
                    if constexpr (LAT) tg = mp_target_lattice(drw >> 1, shd_glo + (own ? base + q : 0ull), lat_k32, rc, k0, k1, shd_Q, dw.n_global) - shd_lo;
                    else tg = mp_target(mp_u52(q ? blk.b : blk.a), Q);
                    if (!own) tg = 1ull;
                } else if constexpr (LAT) tg = mp_target_lattice(drw >> 1, slot_offset + (base + q < n ? base + q : 0ull), lat_k32, rc, k0, k1, Q, dw.n_global);
                else tg = mp_target(mp_u52(q ? blk.b : blk.a), Q);
                mp_locate_r<WALKB>(s_incl, s_W, s_ratio, (uint32_t)dw.nt, tg, nt_over_Q, &tile_of[q], &plt[q], &gslot[q], &sp[q]);
            }
            uint32_t j0[2];
            MP_STAMP(0, 18, 0);
#pragma unroll
            for (int q = 0; q < 2; ++q) j0[q] = dw.guide_old[gslot[q]];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const u64 tbase = (u64)tile_of[q] * TILE;
                const uint32_t tlen = (uint32_t)((n - tbase) < (u64)TILE ? (n - tbase) : (u64)TILE);
                pm[q] = mp_start_row(j0[q], sp[q], (uint32_t)tbase, tlen);   // row where the forward scan starts (| MP_DRAW_SURE0: where it ends)
            }
            if constexpr (SHD) {   // the slots this rank could not fill itself: the row that arrived for them (k_shard_self_place flagged them)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (base + q >= shd_cme) {
                        pm[q] = base + q < n ? dfr_row[base + q] : 0u;
                        plt[q] = 0ull;
                    }
                }
            }
            MP_STAMP(0, 19, 0);
        }
    }
    if (!drew) {
#pragma unroll
        for (int p = 0; p < LANE_ITEMS; ++p) {
            pm[p] = base + p < n ? (inv ? inv[base + p] : (dfr_row ? dfr_row[base + p] : 0u)) : 0u;
            if constexpr (!LT_LATE) plt[p] = (dfr_lt && base + p < n) ? dfr_lt[base + p] : 0ull;
        }
    }
    // phase 2 = mp_run_particle (above): the model kernel in Generate mode for one particle
#define MP_RUN_PARTICLE(P, ZP)                                                                                                             \
    mp_run_particle<Model>(model, n, slot_offset, k0, k1, t, x_in, x_out, logw, obs.v, s0.v, overwrite, cx_old != nullptr, inv_rows, inv != nullptr, \
                           pm[P], &px0[P], base + (u64)(P), ZP, &lw[P], &xv[P], aux.x_rows != 0)
    // what a drawing launch stores as a slot's parent: the local row — or, for a sharded filter's self-drawn resample, the GLOBAL slot
    // id: slot_offset + the row, or what the exchange row says for a slot whose offspring came from another rank
    auto parent_id = [&](uint32_t v) -> uint32_t {
        if constexpr (SHD) return (v & MP_DRAW_RECV) ? (uint32_t)inv_rows[(u64)(v & ~MP_DRAW_RECV) * (u64)(D + 1) + D] : (uint32_t)slot_offset + v;
        else return v;
    };
#define MP_PARENT_ID(V) parent_id(V)
    double px0[LANE_ITEMS];   // (written and read only with deferred draws)
    // Where a lane's deferred draws are looked up (every form gives the same parents; what differs is when a CU's 4096 row
    // gathers hit its vector-memory path, and bursts are what this kernel pays for):
    //   two particles per lane, d = 1 (the headline kernel): the first particle's draw BEFORE the deviates, the second's after
    //     (37.1 us; both after 37.6, both before 39.3, half of the waves before and half after 40.2 .. 40.6)
    //   many sites per particle (lane-local queue below): each round's particles just before the model runs on them (all of
    //     a lane's draws in one batch up front: the bearings tracker 314 -> 327 us per step, the banded d = 16 model 440 -> 434 / 442)
    //   otherwise: after the deviates, one draw at a time (four rows in flight on top of a wider model's registers spill)
    constexpr bool SPLIT2 = !QUEUE && D == 1 && LANE_ITEMS == 2;
    if constexpr (SPLIT2) {
        if (cx_old) mp_resolve_draw<WALKB>(cx_old, n, plt[0], pm[0], &pm[0], &px0[0], inv ? nullptr : inv_rows, D + 1);   // pm[] = the parent from here on
    }
    MP_STAMP(0, 20, 0);
    // ---- phase 1: standard deviates of every (particle, free normal site) of this lane ----
    if constexpr (QUEUE) {
        // lane-local queue, rounds of ITEMS particles (ITEMS * NS accepted pairs in registers at a time); the model runs on a
        // round's particles as soon as their deviates are there
        constexpr int ITEMS = k1_items<Model>() < LANE_ITEMS ? k1_items<Model>() : LANE_ITEMS;
        constexpr int ROUNDS = LANE_ITEMS / ITEMS;
        constexpr int M = ITEMS * NS;
        // the queue holds G accepted pairs at a time (an accepted pair is filed with a chain of 2 G selects, and G pairs are
        // 4 G registers): a particle with more than 8 sites is drawn in groups of 8
        constexpr int G = (ITEMS == 1 && NS > 8 && NS % 8 == 0) ? 8 : M;
        constexpr int GROUPS = M / G;
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            const u64 i0 = base + (u64)rd * ITEMS;
            double zr[M];
#pragma unroll
            for (int gr = 0; gr < GROUPS; ++gr) {
                if constexpr (GROUPS > 1) {   // (no gain for 4-site models, measured on the bearings tracker)
                    // many sites: accepted pairs are filed in LDS (slot-major, so that the lanes of a wave never share a bank
                    // whatever slot each is at) instead of by a chain of selects over 4 G registers
                    __shared__ double2 s_pair[G][THREADS];
                    int q = 0;
                    uint32_t att = 0;
                    while (q < G && att < MP_MAX_ATTEMPTS) {
                        const int slot = gr * G + q;
                        const int p = slot / NS, sidx = slot % NS;
                        const u64 i = i0 + (u64)p;
                        if (sidx >= ns || i >= n) { s_pair[q][threadIdx.x] = make_double2(0., 1.); ++q; continue; }
                        const mp_u64x2 b = mp_philox4x32_10((uint32_t)(slot_offset + i), (uint32_t)t,
                                                            ((uint32_t)MP_DOM_MODEL << 16) | model.normal_site(sidx), att, k0, k1);
                        double u, r;
                        if (!mp_polar_attempt(b, &u, &r)) {  // normal.rs:22
                            ++att;
                        } else {
                            s_pair[q][threadIdx.x] = make_double2(u, r);
                            att = 0;
                            ++q;
                        }
                    }
#pragma unroll
                    for (int qq = 0; qq < G; ++qq) {   // (a lane reads only what it wrote: no barrier)
                        const double2 pr2 = s_pair[qq][threadIdx.x];
                        zr[gr * G + qq] = mp_std_normal_from_pair(pr2.x, pr2.y);
                    }
                    continue;
                }
                double pu[G], pr[G];
#pragma unroll
                for (int q = 0; q < G; ++q) { pu[q] = 0.; pr[q] = 1.; }
                int q = 0;            // next slot of the group: slot gr * G + q = (particle of the round) * NS + site
                uint32_t att = 0;
                while (q < G && att < MP_MAX_ATTEMPTS) {
                    const int slot = gr * G + q;
                    const int p = slot / NS, sidx = slot % NS;
                    if (sidx >= ns) { ++q; continue; }   // a site this time step does not draw (uniform)
                    const u64 i = i0 + (u64)p;
                    if (i >= n) break;
                    const mp_u64x2 b = mp_philox4x32_10((uint32_t)(slot_offset + i), (uint32_t)t,
                                                        ((uint32_t)MP_DOM_MODEL << 16) | model.normal_site(sidx), att, k0, k1);
                    double u, r;
                    if (!mp_polar_attempt(b, &u, &r)) {  // normal.rs:22
                        ++att;
                    } else {
#pragma unroll
                        for (int qq = 0; qq < G; ++qq) {
                            pu[qq] = (qq == q) ? u : pu[qq];
                            pr[qq] = (qq == q) ? r : pr[qq];
                        }
                        att = 0;
                        ++q;
                    }
                }
#pragma unroll
                for (int qq = 0; qq < G; ++qq) zr[gr * G + qq] = mp_std_normal_from_pair(pu[qq], pr[qq]);
            }
            if (cx_old) {
                u64 ltr[ITEMS];
#pragma unroll
                for (int pp = 0; pp < ITEMS; ++pp) ltr[pp] = (LT_LATE && !drew) ? (i0 + pp < n ? dfr_lt[i0 + pp] : 0ull) : plt[rd * ITEMS + pp];
                mp_resolve_draws<ITEMS, WALKB, D == 1>(cx_old, n, ltr, &pm[rd * ITEMS], &pm[rd * ITEMS], &px0[rd * ITEMS], inv ? nullptr : inv_rows, D + 1);
            }
#pragma unroll
            for (int pp = 0; pp < ITEMS; ++pp) MP_RUN_PARTICLE(rd * ITEMS + pp, &zr[pp * NS]);
        }
        if constexpr (CAN_DRAW) {
            if (drew) {   // this lane's two parents, slot order (particle_filter.rs:20 keeps `parents`): one 8-byte store
                uint32_t* pp2 = mp_as_global(drw_v.parent);
                if (base + 1 < n) *reinterpret_cast<uint2*>(pp2 + base) = make_uint2(MP_PARENT_ID(pm[0]), MP_PARENT_ID(pm[1]));
                else if (base < n) pp2[base] = MP_PARENT_ID(pm[0]);
            }
        }
    } else {
        // few sites per particle: attempt 0 of every deviate in straight-line code (independent Philox chains, no
        // divergence; the state fetch's first hop lands meanwhile), then the rejected ones (21.5 %) are retried by the WAVE:
        // its 64 lanes share out the pending deviates and try several further attempts of each at once (attempts are
        // independent counters; the lowest accepted one wins, which is what the reference's recursion returns).  Two rounds
        // nearly always, for every wave alike — a lane-local loop makes the tile's first barrier wait for the unluckiest of
        // 1024 lanes (measured: 10 k cycles of a 43 k-cycle workgroup), and a workgroup-wide queue pays five barriers.
        constexpr int M = LANE_ITEMS * NS;
        __shared__ uint32_t s_it[THREADS / 64][64];   // wave-private: identities of the pending deviates of a round
        const int lane_ = threadIdx.x & 63, wave_ = threadIdx.x >> 6;
        double pu[M], pr[M];
        uint32_t pend = 0u;        // bit q: deviate q of this lane is still to be drawn
        uint32_t att[M];           // its next attempt
#pragma unroll
        for (int q = 0; q < M; ++q) {
            const int p = q / NS, sidx = q % NS;
            pu[q] = 0.; pr[q] = 1.; att[q] = 1u;
            if (sidx < ns && base + p < n) {
                const mp_u64x2 b = mp_philox4x32_10((uint32_t)(slot_offset + base + p), (uint32_t)t,
                                                    ((uint32_t)MP_DOM_MODEL << 16) | model.normal_site(sidx), 0u, k0, k1);
                if (!mp_polar_attempt(b, &pu[q], &pr[q])) pend |= 1u << q;
            }
        }
        const u64 wave_slot0 = slot_offset + (u64)blockIdx.x * TILE + (u64)wave_ * 64 * LANE_ITEMS;   // first slot of this wave's lanes
        MP_STAMP(0, 21, 0);
        for (uint32_t guard = 0; guard < MP_MAX_ATTEMPTS; ++guard) {
            // number the pending deviates of the wave: (q-major, lane) order
            uint32_t idx[M];
            uint32_t R = 0;   // wave-uniform
#pragma unroll
            for (int q = 0; q < M; ++q) {
                const u64 bal = __ballot((pend >> q) & 1u);
                idx[q] = R + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                R += (uint32_t)__popcll(bal);
            }
            if (R == 0u) break;
            const uint32_t R1 = R < 64u ? R : 64u;                           // deviates taken up in this round
            const int lg = R1 <= 1u ? 0 : 32 - __builtin_clz(R1 - 1u);       // R2 = 2^lg >= R1 lanes apart: 64 / R2 attempts of each at once
            const uint32_t R2 = 1u << lg;
#pragma unroll
            for (int q = 0; q < M; ++q)
                if (((pend >> q) & 1u) && idx[q] < 64u) s_it[wave_][idx[q]] = (uint32_t)lane_ | ((uint32_t)q << 8) | (att[q] << 16);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t w = (uint32_t)lane_ & (R2 - 1u), k = (uint32_t)lane_ >> lg;
            const bool work = w < R1;
            const uint32_t ent = s_it[wave_][work ? w : 0u];
            const uint32_t owner = ent & 63u, oq = (ent >> 8) & 0xFFu, oatt = ent >> 16;
            const mp_u64x2 b = mp_philox4x32_10((uint32_t)(wave_slot0 + owner * LANE_ITEMS + oq / NS), (uint32_t)t,
                                                ((uint32_t)MP_DOM_MODEL << 16) | model.normal_site((int)(oq % NS)), oatt + k, k0, k1);
            double u, r;
            const bool acc = mp_polar_attempt(b, &u, &r) && work;
            const u64 A = __ballot(acc);
            __builtin_amdgcn_wave_barrier();   // every lane has read s_it before the next round overwrites it
            // lanes w, w + R2, w + 2 R2 ... tried attempts att, att + 1, ... of deviate w: the lowest accepted one is the draw
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
        MP_STAMP(0, 22, 0);
#pragma unroll
        for (int q = 0; q < M; ++q) z[q] = mp_std_normal_from_pair(pu[q], pr[q]);
    }
    MP_STAMP(0, 2, 0);
    if constexpr (!QUEUE) {
        if (cx_old) {   // pm[] = the parents from here on
            if constexpr (SPLIT2) {
                // (an empty statement that makes the target "depend" on the last deviate: the compiler would otherwise run this
                // lookup ahead of the deviates as well)
                asm volatile("" : "+v"(plt[1]) : "v"(z[NS + NS - 1]));
                mp_resolve_draw<WALKB>(cx_old, n, plt[1], pm[1], &pm[1], &px0[1], inv ? nullptr : inv_rows, D + 1);
                if constexpr (CAN_DRAW) {
                    if (drew) {   // this lane's two parents, slot order: one 8-byte store
                        uint32_t* pp = mp_as_global(drw_v.parent);
                        if (base + 1 < n) *reinterpret_cast<uint2*>(pp + base) = make_uint2(MP_PARENT_ID(pm[0]), MP_PARENT_ID(pm[1]));
                        else if (base < n) pp[base] = MP_PARENT_ID(pm[0]);
                    }
                }
                MP_STAMP(0, 24, 0);
            } else if constexpr (D == 1) {
                mp_resolve_draws<LANE_ITEMS, WALKB>(cx_old, n, plt, pm, pm, px0, inv ? nullptr : inv_rows, D + 1);
            } else {
#pragma unroll
                for (int p = 0; p < LANE_ITEMS; ++p)
                    mp_resolve_draw<WALKB, false>(cx_old, n, drew ? plt[p] : (base + p < n ? dfr_lt[base + p] : 0ull), pm[p], &pm[p], &px0[p], inv ? nullptr : inv_rows, D + 1);
                if constexpr (CAN_DRAW) {
                    if (drew) {
                        uint32_t* pp2 = mp_as_global(drw_v.parent);
                        if (base + 1 < n) *reinterpret_cast<uint2*>(pp2 + base) = make_uint2(MP_PARENT_ID(pm[0]), MP_PARENT_ID(pm[1]));
                        else if (base < n) pp2[base] = MP_PARENT_ID(pm[0]);
                    }
                }
            }
        }
#pragma unroll
        for (int p = 0; p < LANE_ITEMS; ++p) MP_RUN_PARTICLE(p, &z[p * NS]);
    }
#undef MP_RUN_PARTICLE
#undef MP_PARENT_ID
    MP_STAMP(0, 3, 0);

    // ---- level 0 of normalize_weights for this tile, while everything is still in registers ----
    const mp_k1_tail* tp = tail;
    asm volatile("" : "+s"(tp)::"memory");   // the loads of *tp stay here (hoisted to the kernel's entry they would be SGPR pressure again)
    mp_k1_tail tl = mp_ld_const(tp);
    tl.tab.ticket = mp_as_global(tl.tab.ticket); tl.tab.incl = mp_as_global(tl.tab.incl); tl.tab.ratio = mp_as_global(tl.tab.ratio);
    tl.tab.head = mp_as_global(tl.tab.head); tl.tab.W = mp_as_global(tl.tab.W);
    normalize_tile<THREADS>(lw, xv, n, blockIdx.x, mp_as_global(tl.cx), mp_as_global(tl.guide), mp_as_global(tl.tile_m), mp_as_global(tl.tile_W),
                            mp_as_global(tl.tile_W2), tl.tab);
    MP_STAMP(0, 4, 0); MP_STAMP(0, 5, 1);
}

// ---------------------------------------------------------------------------------------------
// K1 for the dense-transition model mp_lgssm_dense<16>: the four 16 x 16 products of a particle-step — A x, T z,
// (y - x)^T R^-1 and its dot with (y - x) — on the matrix cores (v_mfma_f64_16x16x4_f64).  Same results as the scalar
// interpretation of the model functor, bit for bit: every product there is the k-ascending fma chain the MFMA evaluates.
//
// Workgroup = one tile (2048 particles), 512 threads = 8 waves; a wave takes its 256 particles in 4 rounds of 64.  In a
// round, lane = particle while the 16 normals of the "x" site are drawn (one sequential stream per particle: a lane-local
// queue of ~20 Philox blocks), then the 64 particles go through the MFMAs in 4 groups of 16, TRANSPOSED: D[j][i] with the
// state index j on the rows (lane >> 4, register) and the particle i on the columns (lane & 15).  In that orientation
// every product's result is already laid out as the next product's operand: no cross-lane movement between them.
//   A-operand lane l, k-step s = M[l & 15][(l >> 4) + 4 s]   (the constant matrix)
//   B-operand lane l, k-step s = v[(l >> 4) + 4 s] of particle l & 15
//   C / D     lane l, register r = [(l >> 4) + 4 r][l & 15]
// ---------------------------------------------------------------------------------------------
typedef double mp_f64x4 __attribute__((ext_vector_type(4)));
constexpr int DENSE_THREADS = 512;
template <bool WALKB>
__global__ __launch_bounds__(DENSE_THREADS) void k_propagate_dense16(mp_lgssm_dense<16> model, u64 n, u64 slot_offset, uint32_t k0, uint32_t k1, long long t,
                                                                     const double* __restrict__ x_in, double* __restrict__ x_out, double* logw, mp_obs obs,
                                                                     int overwrite, const uint32_t* __restrict__ dfr_row,
                                                                     const u64* __restrict__ dfr_lt, const mp_cx* __restrict__ cx_old, mp_cx* __restrict__ cx,
                                                                     unsigned short* __restrict__ guide, double* tile_m, u64* tile_W, u64* tile_W2,
                                                                     mp_k1_aux aux, const uint32_t* __restrict__ inv, const double* __restrict__ rows) {
    constexpr int D = 16;
    // z of the round's 64 particles per wave, column j of lane l at [l][j ^ (l & 15)]: the swizzle spreads both the per-lane
    // writes and the transposed MFMA-operand reads over the banks without a padding column — with the two staging rows below
    // the kernel needs 78 KB of LDS, so that TWO workgroups fit a CU (16 waves to hide the Philox chains instead of 8)
    __shared__ double s_z[DENSE_THREADS / 64][64][D];
    __shared__ double s_wst[DENSE_THREADS / 64][64], s_xst[DENSE_THREADS / 64][64];   // a round's log-weights / x0, by owning lane
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double lw[TILE / DENSE_THREADS], xv[TILE / DENSE_THREADS];
    const int li = lane & 15, lg = lane >> 4;
    const u64 tile0 = (u64)blockIdx.x * TILE;
    // the constant operands of this lane
    double amat[4], tmat[4], rmat[4], yv[4];
    const double* T = t == 0 ? model.T0() : model.TQ();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        amat[s] = model.A()[li * D + lg + 4 * s];
        tmat[s] = T[li * D + lg + 4 * s];
        rmat[s] = model.Rinv()[(lg + 4 * s) * D + li];   // M[j][k] = Rinv[k][j]: r_j = sum_k c_k Rinv[k][j]
        yv[s] = obs.v[lg + 4 * s];
    }
    const double lp_const = (double)D * MP_LN_2PI_CANON + model.ln_det_R;
#pragma unroll 1
    for (int rd = 0; rd < 4; ++rd) {
        const int pl = tid * 4 + rd;                   // this lane's particle of the round, tile-local: thread t owns rows 4 t .. 4 t + 3,
                                                       // the layout normalize_tile wants its inputs in (registers, no LDS re-map)
        const u64 p = tile0 + (u64)pl;
        const bool live = p < n;
        // ---- the 16 normals of site "x": one sequential stream per particle (mvnormal.rs:35) ----
        {
            mp_stream rng;
            rng.k0 = k0; rng.k1 = k1; rng.slot = (uint32_t)(slot_offset + p); rng.step = (uint32_t)t;
            mp_site st(rng, MP_DOM_MODEL, (uint32_t)mp_lgssm_dense<16>::X);
            int j = live ? 0 : D;
            while (j < D && st.blk < MP_MAX_ATTEMPTS) {
                const mp_u64x2 b = st.next_block();
                double u, r;
                if (mp_polar_attempt(b, &u, &r)) {
                    s_z[wave][lane][j ^ (lane & 15)] = mp_normal_from_pair(u, r, 0., 1.);   // normal.random(rng, (0., 1.))
                    ++j;
                }
            }
            if (!live) {
#pragma unroll
                for (int q = 0; q < D; ++q) s_z[wave][lane][q] = 0.;   // (every column: the swizzle only permutes them)
            }
        }
        // where this particle's previous state lives (slot order, or the parent's row after a binned resample)
        // (as an address: after a sharded resample it is a row of the exchange buffer, or — kept offspring of the owner-keeps
        // exchange — the parent's row of the pre-resample buffer x_in)
        const double* myrow = x_in + p * D;
        if (inv && live) {
            const uint32_t v = inv[p];
            myrow = (v & 0x80000000u) ? x_in + (u64)(v & 0x7FFFFFFFu) * D : rows + (u64)v * (u64)(D + 1);
        } else if (dfr_row && live) {   // a draw of the last resample, looked up here (mp_resolve_draw)
            uint32_t par;
            double x0;
            mp_resolve_draw<WALKB, false>(cx_old, n, dfr_lt[p], dfr_row[p], &par, &x0, rows, D + 1);
            myrow = (rows && (par & MP_DRAW_RECV)) ? rows + (u64)(par & ~MP_DRAW_RECV) * (u64)(D + 1) : x_in + (u64)par * D;
        }
        const u64 myaddr = (u64)(uintptr_t)myrow;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 1
        for (int g = 0; g < 4; ++g) {
            const int src = 16 * g + li;                       // the lane that owns this column's particle
            const double* xrow = reinterpret_cast<const double*>((uintptr_t)(((u64)(uint32_t)__shfl((int)(myaddr >> 32), src, 64) << 32) |
                                                                              (u64)(uint32_t)__shfl((int)myaddr, src, 64)));
            const bool live_i = __shfl((int)live, src, 64) != 0;
            const int pl_i = (wave * 64 + src) * 4 + rd;
            const u64 p_i = tile0 + (u64)pl_i;
            mp_f64x4 mean = {0., 0., 0., 0.};
            if (t != 0) {   // uniform
                double xk[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) xk[s] = live_i ? xrow[lg + 4 * s] : 0.;
#pragma unroll
                for (int s = 0; s < 4; ++s) mean = __builtin_amdgcn_mfma_f64_16x16x4f64(amat[s], xk[s], mean, 0, 0, 0);
            }
            mp_f64x4 tz = {0., 0., 0., 0.};
#pragma unroll
            for (int s = 0; s < 4; ++s) tz = __builtin_amdgcn_mfma_f64_16x16x4f64(tmat[s], s_z[wave][src][(lg + 4 * s) ^ li], tz, 0, 0, 0);   // src & 15 == li
            mp_f64x4 xn, c;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                xn[r] = tz[r] + mean[r];        // transform * z + mu
                c[r] = yv[r] - xn[r];           // centered_x of the observation site
            }
            mp_f64x4 rj = {0., 0., 0., 0.};
#pragma unroll
            for (int s = 0; s < 4; ++s) rj = __builtin_amdgcn_mfma_f64_16x16x4f64(rmat[s], c[s], rj, 0, 0, 0);
            mp_f64x4 pq = {0., 0., 0., 0.};
#pragma unroll
            for (int s = 0; s < 4; ++s) pq = __builtin_amdgcn_mfma_f64_16x16x4f64(rj[s], c[s], pq, 0, 0, 0);
            // diagonal of pq = the Mahalanobis terms: element [i][i] sits in lane (i & 3) * 16 + i, register i >> 2
            if (live_i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) x_out[p_i * D + lg + 4 * r] = xn[r];
            }
            if (lg == (li & 3)) {
                const int rr = li >> 2;
                const double maha = rr == 0 ? pq[0] : rr == 1 ? pq[1] : rr == 2 ? pq[2] : pq[3];
                const double gw = 0. + (-(lp_const + maha) / 2.);
                double w = MP_NEG_INF;
                if (live_i) {
                    w = overwrite == 1 ? gw : (overwrite == 2 ? 0. + gw : logw[p_i] + gw);
                    logw[p_i] = w;
                }
                s_wst[wave][src] = w;
            }
            if (lg == 0) s_xst[wave][src] = live_i ? xn[0] : 0.;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();   // the staged results are this wave's own; the next round overwrites its z
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const double w_mine = s_wst[wave][lane], x_mine = s_xst[wave][lane];
#pragma unroll
        for (int j = 0; j < 4; ++j) {   // (static indices: lw / xv stay in registers)
            lw[j] = j == rd ? w_mine : lw[j];
            xv[j] = j == rd ? x_mine : xv[j];
        }
        __builtin_amdgcn_wave_barrier();
    }
    normalize_tile<DENSE_THREADS>(lw, xv, n, blockIdx.x, cx, guide, tile_m, tile_W, tile_W2, aux.tab);
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
// (the ratio is a per-tile constant: a table that is built once may carry it — mp_local_target_r — instead of dividing per draw)
__device__ __forceinline__ u64 mp_local_target_r(u64 r, u64 W, double ratio) {
    const double v = ceil((double)r * ratio);
    u64 x = (v >= 1.) ? (u64)v : 1ull;
    if (x > W) x = W;
    if (x < 1ull) x = 1ull;
    return x;
}
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
    if (!(m > MP_NEG_INF) || !(m < MP_INF) || Q == 0) mp_flag_degenerate(scal);
    if (mode == 0) {  // resample (particle_filter.rs:104-105)
        scal->L = L;
        scal->ess_stale = ess;
        scal->Q = Q;
        scal->Q2 = Q2;
        const double lml = scal->log_ml + (L - mp_log((double)n_global));
        scal->log_ml = lml;
        const u64 folds = scal->folds + 1;
        scal->folds = folds;
        if (scal->mirror) {
            mp_host_mirror* hm = scal->mirror;
            __hip_atomic_store(reinterpret_cast<u64*>(&hm->L), mp_f2u(L), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(reinterpret_cast<u64*>(&hm->ess_stale), mp_f2u(ess), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(reinterpret_cast<u64*>(&hm->log_ml), mp_f2u(lml), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            // (a release fence here — buffer_wbl2 and all — is at the START of the launch, in front of nobody's last instruction: kept)
            __hip_atomic_store(&hm->fold_seq, folds, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
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
    if ((n_global & (n_global - 1ull)) == 0ull) {   // power-of-two populations (the usual case): a shift instead of two 64-bit divisions
        const int sh = __ffsll((long long)n_global) - 1;   // 0 .. 31
        return (sh == 0 ? a_lo : ((a_lo >> sh) | (a_hi << (64 - sh)))) + 1ull;
    }
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
// Stratified resampling (extension): the same lattice with one 32-bit uniform PER output slot, u_g = (g + k32_g / 2^32) / N:
// word (g & 3) of Philox block g >> 2 at site 2 (oracle/src/rng.hpp resample_k32); parents still come out sorted.
__device__ __forceinline__ uint32_t mp_stratified_k32(u64 g, uint32_t rc, uint32_t k0, uint32_t k1) {
    const mp_u64x2 r = mp_philox4x32_10((uint32_t)(g >> 2), rc, ((uint32_t)MP_DOM_RESAMPLE << 16) | 2u, 0u, k0, k1);
    const u64 w = (g & 2) ? r.b : r.a;
    return (g & 1) ? (uint32_t)(w >> 32) : (uint32_t)w;
}
// The categorical draws of one resample (domain RESAMPLE) or of importance_resampling (domain IS) are ONE sequential
// uniform stream, like the reference's loop over one rng (particle_filter.rs:38-40): draw g is half (g & 1) of Philox
// block g >> 1, the block index in the counter's slot field (oracle/src/rng.hpp resample_rng).  A lane that owns the
// adjacent draws 2j, 2j + 1 pays one block for both.
__device__ __forceinline__ mp_u64x2 mp_resample_block(u64 g_pair, uint32_t rc, uint32_t domain, uint32_t k0, uint32_t k1) {
    return mp_philox4x32_10((uint32_t)g_pair, rc, domain << 16, 0u, k0, k1);
}
__device__ __forceinline__ u64 mp_resample_k52(u64 g, uint32_t rc, uint32_t domain, uint32_t k0, uint32_t k1) {
    const mp_u64x2 r = mp_resample_block(g >> 1, rc, domain, k0, k1);
    return mp_u52((g & 1) ? r.b : r.a);
}
// target of global output slot g under scheme 1 (systematic, shared k32) or 2 (stratified)
__device__ __forceinline__ u64 mp_target_lattice(int scheme, u64 g, uint32_t shared_k32, uint32_t rc, uint32_t k0, uint32_t k1, u64 Q, u64 n_global) {
    return mp_target_systematic(g, scheme == 2 ? mp_stratified_k32(g, rc, k0, k1) : shared_k32, Q, n_global);
}

// Tile of a global target: tile totals are nearly equal (each sums 2048 weights), so target * nt / Q lands within a
// tile or two of the answer; walk from there.  Same result as a lower_bound over s_incl, fewer LDS reads.
// BS: a plain lower bound by bisection instead — when the weights have collapsed most tiles carry no mass, the guess lands anywhere and
// the walk crosses hundreds of flat entries (the WALKB kernels: same tile either way)
template <bool BS = false>
__device__ __forceinline__ uint32_t tile_of_target(const u64* s_incl, uint32_t nt, u64 target, double nt_over_Q) {
    if constexpr (BS) {
        // the plain walk with a budget of steps; a walk that uses it up is settled by a bisection of the whole table (first b with
        // incl[b] >= target, or nt - 1: what the walk finds).  Measured against the plain walk and against "three steps, then bisect what
        // is left" on healthy weights: no difference (37.4 - 37.9 us, four alternations on one box).
        int b = (int)((double)target * nt_over_Q);
        if (b > (int)nt - 1) b = (int)nt - 1;
        if (b < 0) b = 0;
        int budget = 4;
        while (b > 0 && s_incl[b - 1] >= target && budget > 0) { --b; --budget; }
        while (b < (int)nt - 1 && s_incl[b] < target && budget > 0) { ++b; --budget; }
        if (budget == 0) {
            uint32_t lo = 0u, hi = nt - 1u;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (s_incl[mid] >= target) hi = mid;
                else lo = mid + 1u;
            }
            b = (int)lo;
        }
        return (uint32_t)b;
    }
    int b = (int)((double)target * nt_over_Q);
    if (b > (int)nt - 1) b = (int)nt - 1;
    if (b < 0) b = 0;
    while (b > 0 && s_incl[b - 1] >= target) --b;          // first b with incl[b] >= target ...
    while (b < (int)nt - 1 && s_incl[b] < target) ++b;     // ... from either side
    return (uint32_t)b;
}
// the same with the per-tile ratios (double)W_b / (double)T_b precomputed (k_shard_table)
// (sub: the 32nd of its guide cell the tile-local target lies in, mp_guide_sub — for the readers that use the cell's position bits)
template <bool BS>
__device__ __forceinline__ void mp_locate_r(const u64* s_incl, const u64* s_W, const double* s_ratio, uint32_t nt, u64 target, double nt_over_Q,
                                            uint32_t* tile, u64* lt, uint32_t* gslot, uint32_t* sub) {
    const uint32_t b = tile_of_target<BS>(s_incl, nt, target, nt_over_Q);
    const u64 excl = b ? s_incl[b - 1] : 0ull;
    const u64 W = s_W[b];
    const u64 x = mp_local_target_r(target - excl, W, s_ratio[b]);
    const int shift = mp_guide_shift(W);
    uint32_t g = (uint32_t)(x >> shift);
    if (g > GUIDE_N - 1) g = GUIDE_N - 1;
    *tile = b; *lt = x; *gslot = b * (uint32_t)GUIDE_N + g;
    if (sub) *sub = mp_guide_sub(x, shift);
}
// global target -> (tile, tile-local target, guide slot)
template <bool BS = false>
__device__ __forceinline__ void mp_locate(const u64* s_incl, const u64* s_W, uint32_t nt, u64 target, double nt_over_Q, uint32_t* tile, u64* lt,
                                          uint32_t* gslot) {
    const uint32_t b = tile_of_target<BS>(s_incl, nt, target, nt_over_Q);
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
constexpr int KG_THREADS = 256;   // the single-kernel resample (lattice schemes, importance_resampling's M draws)
constexpr int KG_ITEMS = 4;   // 256 x 4 beats 512 x 2 here (23.3 vs 25.5 us: one tile table per workgroup, sorted parents)
template <int SCHEME>
__global__ __launch_bounds__(KG_THREADS) void k_resample_gather(u64 n, u64 n_out, u64 n_global, u64 slot_offset, uint32_t domain,
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
    u64* s_wtot = reinterpret_cast<u64*>(s_red + KG_THREADS / 64);
    const double m = block_tile_table<KG_THREADS>(tile_m, tile_W, nt, S, s_incl, s_W, s_red, s_wtot);
    const u64 Q = s_incl[nt - 1];
    if (blockIdx.x == 0 && scal != nullptr) {  // workgroup-uniform: fold this normalisation into the filter scalars
        const u64 Q2 = block_sum_T2<KG_THREADS>(tile_m, tile_W2, nt, S, m, s_wtot);
        if (threadIdx.x == 0) fold_scalars(scal, Q, Q2, S, m, n_global, 0);
    }
    const uint32_t sys_k32 = SCHEME == 1 ? mp_systematic_k32(rc, k0, k1) : 0u;
    const double nt_over_Q = (double)nt / (double)Q;  // only a starting guess for the tile walk: no effect on results
    for (u64 i0 = (u64)blockIdx.x * (KG_THREADS * KG_ITEMS) + threadIdx.x; i0 < n_out; i0 += (u64)gridDim.x * (KG_THREADS * KG_ITEMS)) {
        u64 lt[KG_ITEMS], tbase[KG_ITEMS];
        uint32_t tlen[KG_ITEMS], j[KG_ITEMS], gslot[KG_ITEMS];
#pragma unroll
        for (int k = 0; k < KG_ITEMS; ++k) {
            const u64 i = i0 + (u64)k * KG_THREADS;
            u64 target;
            if (SCHEME != 0) {
                target = mp_target_lattice(SCHEME, slot_offset + (i < n_out ? i : 0), sys_k32, rc, k0, k1, Q, n_global);
            } else {
                target = mp_target(mp_resample_k52(slot_offset + i, rc, domain, k0, k1), Q);
            }
            uint32_t b;
            mp_locate<true>(s_incl, s_W, (uint32_t)nt, target, nt_over_Q, &b, &lt[k], &gslot[k]);   // (the budgeted walk: collapsed weights)
            tbase[k] = (u64)b * TILE;
            tlen[k] = (uint32_t)((n - tbase[k]) < (u64)TILE ? (n - tbase[k]) : (u64)TILE);
        }
#pragma unroll
        for (int k = 0; k < KG_ITEMS; ++k) j[k] = mp_guide_row(guide[gslot[k]]);
        mp_cx r0[KG_ITEMS], r1[KG_ITEMS];
#pragma unroll
        for (int k = 0; k < KG_ITEMS; ++k) {
            if (j[k] > tlen[k] - 1) j[k] = tlen[k] - 1;
            const uint32_t j1 = (j[k] + 1 < tlen[k]) ? j[k] + 1 : j[k];
            r0[k] = load_row_nt(cx + tbase[k] + j[k]);
            r1[k] = load_row_nt(cx + tbase[k] + j1);
        }
#pragma unroll
        for (int k = 0; k < KG_ITEMS; ++k) {
            const u64 i = i0 + (u64)k * KG_THREADS;
            mp_cx row = r0[k];
            uint32_t jj = j[k];
            if (row.cum < lt[k] && jj + 1 < tlen[k]) {    // first row with cum >= lt
                row = r1[k];
                ++jj;
                int steps = 0;
                while (row.cum < lt[k] && jj + 1 < tlen[k]) {
                    if (++steps > MP_WALK_LINEAR) {   // a long walk (collapsed weights) finishes by bisection: mp_resolve_draws' BISECT
                        uint32_t lo = jj + 1, hi = tlen[k] - 1;
                        while (lo < hi) {
                            const uint32_t mid = lo + ((hi - lo) >> 1);
                            if (load_row_nt(cx + tbase[k] + mid).cum >= lt[k]) hi = mid;
                            else lo = mid + 1;
                        }
                        jj = lo;
                        row = load_row_nt(cx + tbase[k] + jj);
                        break;
                    }
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
// Multinomial resampling in two halves (same parents per slot as k_resample_gather, bit for bit)
// ---------------------------------------------------------------------------------------------
//   k_draw_slots    every chunk of 1024 output slots: tile table, Philox (one block per two adjacent slots), target, tile +
//                   guide lookup (the 2 MB guide is L2-resident everywhere) -> {tile-local target, start row} per slot.
//   the row lookups (start row, successor, rarely more: mp_resolve_draws) are left to the kernel that consumes the parents:
//                   the next step's k_propagate, which looks its own slots' draws up under its arithmetic, or
//                   k_resolve_slots when the host asks for slot-order states / parents first.
// The row table (16 B x N) does not fit one XCD's 4 MB L2, so the random row reads pull whole lines through the fabric
// whoever issues them.  Round 1 / early round 2 binned the draws by the top 3 bits of their uniform and resolved bin b on
// XCD b in a kernel of its own (L2 hit rate 0.58 -> 0.86, 14 us at 2^20); fused into k_propagate the unbinned lookups cost
// that kernel 10 us, the binning's scan, barriers and segment traffic disappear from this one (12.6 -> 9.9 us), and there
// is no third kernel: 53.7 -> 47.5 us per step.
// Tile table (TABMODE): 0 = every workgroup builds it in LDS from the tile scalars (handles without a k_propagate-built
// table); 1 = built once by the last workgroup of the level-0 launch (build_tile_table_global) and copied to LDS here;
// 2 = the same, probed where it lies in L2 (more tiles than fit LDS).
// SCHEME: 0 multinomial (one Philox block per two adjacent slots), 1 systematic, 2 stratified (mp_target_lattice: the draws
// are sorted, so the lookups the consumer makes are nearly sequential).
template <int TABMODE, int SCHEME>
__global__ __launch_bounds__(DRAW_THREADS) __attribute__((amdgpu_num_sgpr(80))) void k_draw_slots(u64 n, u64 n_global, u64 slot_offset, uint32_t k0, uint32_t k1, uint32_t rc, int S,
                                                           const double* __restrict__ tile_m, const u64* __restrict__ tile_W,
                                                           const u64* __restrict__ tile_W2, int nt,
                                                           const unsigned short* __restrict__ guide,
                                                           u64* __restrict__ dfr_lt, uint32_t* __restrict__ dfr_row,
                                                           mp_dev_scalars* scal, const u64* __restrict__ incl_pre,
                                                           const double* __restrict__ ratio_pre, const mp_tab_head* __restrict__ head, int n_chunks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NW = DRAW_THREADS / 64;
    const int nt_lds = TABMODE == 2 ? 0 : nt;
    u64* s_incl_lds = reinterpret_cast<u64*>(smem);                    // [nt]
    u64* s_W_lds = s_incl_lds + nt_lds;                                // [nt]
    double* s_ratio_lds = reinterpret_cast<double*>(s_W_lds + nt_lds); // [nt] (TABMODE 1)
    double* s_red = s_ratio_lds + (TABMODE == 1 ? nt_lds : 0);         // [NW]
    u64* s_wtot = reinterpret_cast<u64*>(s_red + NW);                  // [NW]
    const u64* s_incl = TABMODE == 2 ? incl_pre : s_incl_lds;
    const u64* s_W = TABMODE == 2 ? tile_W : s_W_lds;
    const double* s_ratio = TABMODE == 2 ? ratio_pre : s_ratio_lds;
    // A workgroup takes the chunks blockIdx.x, blockIdx.x + gridDim.x, ... (n_chunks of them in all): its LDS copy of the tile
    // table serves every one of them — one workgroup per chunk copied 24 B x tiles from L2 for every 1024 draws: 200 MB per
    // launch at 2^22 particles (2048 tiles, 4096 chunks), most of that kernel's 55 us
    int c = blockIdx.x;
    const int tid = threadIdx.x;
    MP_STAMP(1, 0, 0); MP_STAMP(1, 1, 1); MP_STAMP(1, 6, 2);
    if constexpr (TABMODE == 0) {
        const double m = block_tile_table<DRAW_THREADS>(tile_m, tile_W, nt, S, s_incl_lds, s_W_lds, s_red, s_wtot);
        if (blockIdx.x == 0) {  // fold this normalisation into the filter scalars
            const u64 Q2 = block_sum_T2<DRAW_THREADS>(tile_m, tile_W2, nt, S, m, s_wtot);
            if (threadIdx.x == 0) fold_scalars(scal, s_incl_lds[nt - 1], Q2, S, m, n_global, 0);
        }
    }
    // this thread's two adjacent output slots share one Philox block (it depends on no table: in TABMODE 1 it is computed
    // while the table's loads are in flight)
    u64 i0 = (u64)c * DRAW_CHUNK + 2u * (u64)tid;
    mp_u64x2 blk;
    blk.a = 0ull; blk.b = 0ull;
    if constexpr (TABMODE == 1) {
        constexpr int TPT = (K1_TABLE_LDS_MAX_TILES + DRAW_THREADS - 1) / DRAW_THREADS;   // table entries per thread
        u64 tI[TPT], tW[TPT];
        double tR[TPT];
#pragma unroll
        for (int k = 0; k < TPT; ++k) {
            const int b = tid + k * DRAW_THREADS;
            tI[k] = b < nt ? incl_pre[b] : 0ull;
            tW[k] = b < nt ? tile_W[b] : 0ull;
            tR[k] = b < nt ? ratio_pre[b] : 0.;
        }
        if constexpr (SCHEME == 0) blk = mp_resample_block((slot_offset + i0) >> 1, rc, (uint32_t)MP_DOM_RESAMPLE, k0, k1);
#pragma unroll
        for (int k = 0; k < TPT; ++k) {
            const int b = tid + k * DRAW_THREADS;
            if (b < nt) { s_incl_lds[b] = tI[k]; s_W_lds[b] = tW[k]; s_ratio_lds[b] = tR[k]; }
        }
    } else {
        if constexpr (SCHEME == 0) blk = mp_resample_block((slot_offset + i0) >> 1, rc, (uint32_t)MP_DOM_RESAMPLE, k0, k1);
    }
    if constexpr (TABMODE != 0) {
        if (blockIdx.x == 0 && tid == 0) fold_scalars(scal, head->Q, head->Q2, S, head->m, n_global, 0);
        __syncthreads();
    }
    const u64 Q = s_incl[nt - 1];
    MP_STAMP(1, 2, 0);
    const double nt_over_Q = (double)nt / (double)Q;  // only a starting guess for the tile walk: no effect on results
    const uint32_t sys_k32 = SCHEME == 1 ? mp_systematic_k32(rc, k0, k1) : 0u;
    for (;;) {
    u64 lt[2];
    uint32_t gslot[2], tile_of[2], j0[2];
    bool live[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        live[q] = i0 + q < n;
        u64 target;
        if constexpr (SCHEME == 0) target = mp_target(mp_u52(q ? blk.b : blk.a), Q);
        else target = mp_target_lattice(SCHEME, slot_offset + (live[q] ? i0 + q : 0), sys_k32, rc, k0, k1, Q, n_global);
        // (the budgeted tile walk: with collapsed weights the guess lands anywhere in a flat table — this kernel 11.9 -> 37.7 us)
        if constexpr (TABMODE == 0) mp_locate<true>(s_incl, s_W, (uint32_t)nt, target, nt_over_Q, &tile_of[q], &lt[q], &gslot[q]);
        else mp_locate_r<true>(s_incl, s_W, s_ratio, (uint32_t)nt, target, nt_over_Q, &tile_of[q], &lt[q], &gslot[q]);
    }
    // the guide lookups (the guide is L2-resident on every XCD)
#pragma unroll
    for (int q = 0; q < 2; ++q) j0[q] = guide[gslot[q]];
    // {target, start row} of the two draws in SLOT order: one 16-byte and one 8-byte store per lane
    uint32_t srow[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const u64 tbase = (u64)tile_of[q] * TILE;
        const uint32_t tlen = (uint32_t)((n - tbase) < (u64)TILE ? (n - tbase) : (u64)TILE);
        // row where the forward scan starts — or, flagged MP_DRAW_SURE0, where the guide cell says it ends (mp_start_row)
        srow[q] = mp_start_row(j0[q], mp_guide_sub(lt[q], mp_guide_shift(s_W[tile_of[q]])), (uint32_t)tbase, tlen);
    }
    if (live[1]) {
        mp_u64v2 v2; v2.x = lt[0]; v2.y = lt[1];
        *reinterpret_cast<mp_u64v2*>(dfr_lt + i0) = v2;                                      // i0 is even: aligned
        *reinterpret_cast<uint2*>(dfr_row + i0) = make_uint2(srow[0], srow[1]);
    } else if (live[0]) {
        dfr_lt[i0] = lt[0];
        dfr_row[i0] = srow[0];
    }
    c += (int)gridDim.x;
    if (c >= n_chunks) break;
    i0 = (u64)c * DRAW_CHUNK + 2u * (u64)tid;
    if constexpr (SCHEME == 0) blk = mp_resample_block((slot_offset + i0) >> 1, rc, (uint32_t)MP_DOM_RESAMPLE, k0, k1);
    }
    MP_STAMP(1, 3, 0); MP_STAMP(1, 4, 0); MP_STAMP(1, 5, 1);
}

// The lookups of a resample that only drew (k_draw_slots), for whoever needs slot-order results before — or instead of — the
// next k_propagate: traces[i] = traces[parents[i]].clone(); log_weights.fill(0.) (particle_filter.rs:109-114).
// STATES = false: parents only (a step has already consumed the draws and moved the states on).
// (sharded filters: `parent` holds GLOBAL slot ids — slot_offset + the local row, or what the exchange row says for a slot
// whose offspring came from another rank: `rows`, MP_DRAW_RECV)
template <bool STATES>
__global__ void k_resolve_slots(u64 n, int D, const u64* __restrict__ dfr_lt, const uint32_t* __restrict__ dfr_row, const mp_cx* __restrict__ cx,
                                const double* __restrict__ x_old, double* __restrict__ x_new, uint32_t* __restrict__ parent, double* __restrict__ logw,
                                const double* __restrict__ rows, u64 slot_offset) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t p;
    double x0;
    mp_resolve_draw<true>(cx, n, dfr_lt[i], dfr_row[i], &p, &x0, rows, D + 1);
    const bool recv = rows && (p & MP_DRAW_RECV);
    const double* rrow = recv ? rows + (u64)(p & ~MP_DRAW_RECV) * (u64)(D + 1) : nullptr;
    parent[i] = recv ? (uint32_t)rrow[D] : (uint32_t)(slot_offset + p);
    if constexpr (STATES) {
        if (D == 1) {
            x_new[i] = x0;
        } else {
            for (int d = 0; d < D; ++d) x_new[i * D + d] = recv ? rrow[d] : x_old[(u64)p * D + d];
        }
        logw[i] = 0.;
    }
}

// `traces[i].retv` for a range of particles at once (particle_filter.rs:13; tests/smc.rs:67 walks every particle): thread =
// particle, walking the event log backwards — a resample maps slot -> parent slot (traces[i] = traces[parents[i]].clone(),
// :109-113), a step contributes the state of the current ancestor slot (retv.push, dynunfold.rs:58,92).
struct mp_hist_event {
    const void* buf;   // kind 0: [n][d] f64 states after an Unfold step; kind 1: [n] u32 parents of a resample
    int kind;
    int pad;
};
__global__ void k_trajectories(u64 first, u64 count, int D, int T, int n_events, const mp_hist_event* __restrict__ ev, double* __restrict__ out) {
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    u64 a = first + j;
    int t = T;
    for (int e = n_events - 1; e >= 0; --e) {
        if (ev[e].kind == 1) {
            a = static_cast<const uint32_t*>(ev[e].buf)[a];
        } else {
            --t;
            const double* src = static_cast<const double*>(ev[e].buf) + a * (u64)D;
            double* dst = out + (j * (u64)T + (u64)t) * (u64)D;
            for (int d = 0; d < D; ++d) dst[d] = src[d];
        }
    }
}

// Level 1 of the CURRENT tile scalars for a synchronous `resample() -> f64` (particle_filter.rs:103-105, 116): the log total
// weight (and the ESS of these weights) straight into host-mapped memory, nothing folded — the resample's draws, and with them
// the fold into log_ml, are still left to the next step's k_propagate exactly as after an asynchronous resample.  Same
// arithmetic as block_tile_table / block_sum_T2 (T_b, T2_b entry by entry, integer sums), one workgroup, no table.
__global__ __launch_bounds__(1024) void k_peek_level1(const double* __restrict__ tile_m, const u64* __restrict__ tile_W, const u64* __restrict__ tile_W2,
                                                      int nt, int S, mp_dev_scalars* scal, mp_host_mirror* hm, unsigned long long seq) {
    __shared__ double s_red[16];
    __shared__ u64 s_q[16], s_q2[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double m = MP_NEG_INF;
    for (int b = tid; b < nt; b += 1024) m = fmax(m, tile_m[b]);
    m = wave_max(m);
    if (lane == 0) s_red[wave] = m;
    __syncthreads();
    m = s_red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) m = fmax(m, s_red[w]);
    const bool ok = (m > MP_NEG_INF) && (m < MP_INF);
    const double sc = mp_u2f((u64)(1023 + S - FIX_BITS) << 52);  // 2^(S-51)
    u64 q = 0, q2 = 0;
    for (int b = tid; b < nt; b += 1024) {
        const double d = tile_m[b] - m;
        q += mp_quantize((double)tile_W[b] * (ok ? mp_exp(d) : 0.) * sc, 1.0);
        q2 += mp_quantize((double)tile_W2[b] * (ok ? mp_exp(2. * d) : 0.) * sc, 1.0);
    }
    q = wave_sum_u64(q);
    q2 = wave_sum_u64(q2);
    if (lane == 0) { s_q[wave] = q; s_q2[wave] = q2; }
    __syncthreads();
    if (tid == 0) {
        u64 Q = 0, Q2 = 0;
        for (int w = 0; w < 16; ++w) { Q += s_q[w]; Q2 += s_q2[w]; }
        double L, ess;
        finalize_scalars(Q, Q2, S, &L, &ess, m);
        const int degenerate = (!ok || Q == 0) ? 1 : 0;
        if (degenerate) mp_flag_degenerate(scal);
        __hip_atomic_store(reinterpret_cast<u64*>(&hm->peek_L), mp_f2u(L), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(reinterpret_cast<u64*>(&hm->peek_ess), mp_f2u(ess), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&hm->peek_degenerate, degenerate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        mp_st_sys_seq(&hm->peek_seq, seq);
    }
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
            if (!(m > MP_NEG_INF) || !(m < MP_INF) || Q == 0) mp_flag_degenerate(scal);
            scal->L = L;
            scal->lml_fresh = L - mp_log((double)n_global);
        } else {
            fold_scalars(scal, Q, Q2, S, m, n_global, mode);
        }
    }
}

