// What a CU's vector-memory path charges for the random small reads of a gather-based resample, and what the memory-side
// counters report for them (calibration of FETCH_SIZE for k_propagate's row lookups: run under
// `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and divide by the requests printed here).
// Every lane makes ROUNDS x 4 independent reads of W bytes at pseudo-random rows of a table of `rows` rows (the grid is
// k_propagate's: 512 workgroups x 1024 threads, all resident); reported: ns and shader cycles per request per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/gather_probe.hip -o tools/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
constexpr int ROUNDS = 16;

enum { V_PLAIN16, V_NT16, V_SC1_16, V_SC01_16, V_PAIR16, V_U16, V_B64, V_PAIR16_ALIGNED };

template <int V>
__device__ __forceinline__ u64 rd(const unsigned char* t, uint32_t row) {
    if constexpr (V == V_PLAIN16) { const u64x2 v = *reinterpret_cast<const u64x2*>(t + (size_t)row * 16); return v.x ^ v.y; }
    if constexpr (V == V_NT16) { const u64x2 v = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(t + (size_t)row * 16)); return v.x ^ v.y; }
    if constexpr (V == V_PAIR16) {   // a row and its successor: two requests, one 64-byte line three times out of four
        const u64x2 a = *reinterpret_cast<const u64x2*>(t + (size_t)row * 16), b = *reinterpret_cast<const u64x2*>(t + (size_t)row * 16 + 16);
        return a.x ^ a.y ^ b.x ^ b.y;
    }
    if constexpr (V == V_PAIR16_ALIGNED) {   // two rows of one 32-byte cell: always one line
        const size_t o = ((size_t)row * 16) & ~(size_t)31;
        const u64x2 a = *reinterpret_cast<const u64x2*>(t + o), b = *reinterpret_cast<const u64x2*>(t + o + 16);
        return a.x ^ a.y ^ b.x ^ b.y;
    }
    if constexpr (V == V_U16) return *reinterpret_cast<const unsigned short*>(t + (size_t)row * 2);
    if constexpr (V == V_B64) return *reinterpret_cast<const u64*>(t + (size_t)row * 8);
    return 0;
}

template <int V>
__global__ __launch_bounds__(1024) void k_gather(const unsigned char* __restrict__ t, uint32_t row_mask, u64* out, u64* stamps) {
    uint32_t s = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
    u64 acc = 0;
    u64 t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
    for (int r = 0; r < ROUNDS; ++r) {
        uint32_t row[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { s = s * 1664525u + 1013904223u; row[k] = (s >> 7) & row_mask; }
        u64 v[4];
        if constexpr (V == V_SC1_16 || V == V_SC01_16) {   // cache-policy bits need the instruction spelt out: four loads, then one wait
            u64x2 w0, w1, w2, w3;
            const unsigned char *p0 = t + (size_t)row[0] * 16, *p1 = t + (size_t)row[1] * 16, *p2 = t + (size_t)row[2] * 16, *p3 = t + (size_t)row[3] * 16;
            if constexpr (V == V_SC1_16)
                asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off sc1\n\t"
                             "global_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
            else
                asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\tglobal_load_dwordx4 %1, %5, off sc0 sc1\n\tglobal_load_dwordx4 %2, %6, off sc0 sc1\n\t"
                             "global_load_dwordx4 %3, %7, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
            v[0] = w0.x ^ w0.y; v[1] = w1.x ^ w1.y; v[2] = w2.x ^ w2.y; v[3] = w3.x ^ w3.y;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = rd<V>(t, row[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc ^= v[k];
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (acc == 0x123456789ull) out[0] = acc;
    if ((threadIdx.x & 63) == 0) stamps[(size_t)blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int V>
static void run(const char* name, const unsigned char* t, size_t table_bytes, int width, int reqs_per_read, u64* out, u64* stamps) {
    const uint32_t rows = (uint32_t)(table_bytes / (size_t)width);
    const uint32_t mask = rows - 1u - (reqs_per_read > 1 ? 1u : 0u) * 0u;   // (the pair forms read one row past: the table has a spare row)
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k_gather<V>, dim3(512), dim3(1024), 0, 0, t, mask, out, stamps);
    (void)hipDeviceSynchronize();
    const int reps = 5;
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_gather<V>, dim3(512), dim3(1024), 0, 0, t, mask, out, stamps);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    std::vector<u64> h(512 * 16);
    (void)hipMemcpy(h.data(), stamps, sizeof(u64) * h.size(), hipMemcpyDeviceToHost);
    double mean = 0;
    for (u64 x : h) mean += (double)x;
    mean /= (double)h.size();
    const double reads = 512.0 * 1024 * ROUNDS * 4, reqs = reads * reqs_per_read;
    const double us = ms * 1e3 / reps;
    std::printf("%-34s table %7.1f MB  %8.1f us/launch  %6.2f ns/request/CU  %6.2f wave-lifetime cycles/request/CU  (%.0f requests per launch)\n", name,
                table_bytes / 1048576.0, us, us * 1e3 / (reqs / 256.0), mean / (reqs / 256.0), reqs);
}

int main() {
    const size_t big = (size_t)1 << 30;
    unsigned char* t;
    (void)hipMalloc(&t, big + 64);
    (void)hipMemset(t, 1, big + 64);
    u64 *out, *stamps;
    (void)hipMalloc(&out, 64);
    (void)hipMalloc(&stamps, sizeof(u64) * 512 * 16);
    std::printf("# random reads by 512 x 1024 threads (8 waves per SIMD), %d x 4 reads per lane\n", ROUNDS);
    run<V_PLAIN16>("16 B rows, plain", t, (size_t)16 << 20, 16, 1, out, stamps);
    run<V_NT16>("16 B rows, nt", t, (size_t)16 << 20, 16, 1, out, stamps);
    run<V_SC1_16>("16 B rows, sc1", t, (size_t)16 << 20, 16, 1, out, stamps);
    run<V_SC01_16>("16 B rows, sc0 sc1", t, (size_t)16 << 20, 16, 1, out, stamps);
    run<V_PAIR16>("16 B row + successor, plain", t, (size_t)16 << 20, 16, 2, out, stamps);
    run<V_PAIR16_ALIGNED>("2 rows of a 32 B cell, plain", t, (size_t)16 << 20, 16, 2, out, stamps);
    run<V_U16>("2 B cells (the guide), plain", t, (size_t)2 << 20, 2, 1, out, stamps);
    run<V_B64>("8 B rows, plain", t, (size_t)8 << 20, 8, 1, out, stamps);
    run<V_PLAIN16>("16 B rows, plain", t, (size_t)2 << 20, 16, 1, out, stamps);
    run<V_PLAIN16>("16 B rows, plain", t, (size_t)64 << 20, 16, 1, out, stamps);
    run<V_PLAIN16>("16 B rows, plain", t, (size_t)1 << 30, 16, 1, out, stamps);
    run<V_NT16>("16 B rows, nt", t, (size_t)1 << 30, 16, 1, out, stamps);
    return 0;
}
