// Is the cost of K1's random row gathers a per-CU limit or a chip-wide one?  The gather kernel of tools/gather_probe.hip (every lane
// makes ROUNDS x 4 independent random 16-byte reads) with 32 .. 512 workgroups of 1024 threads: if a CU's time per request does
// not change with the number of CUs gathering, the limit is the CU's own (outstanding misses x latency); if it grows, the fabric's.
//   hipcc --offload-arch=gfx950 -O3 tools/gather_scale.hip -o tools/gather_scale
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
constexpr int ROUNDS = 16;
template <int PER>
__global__ __launch_bounds__(1024) void k_gather(const unsigned char* __restrict__ t, uint32_t row_mask, u64* out) {
    uint32_t s = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
    u64 acc = 0;
#pragma unroll 1
    for (int r = 0; r < ROUNDS * 4 / PER; ++r) {
        uint32_t row[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) { s = s * 1664525u + 1013904223u; row[k] = (s >> 7) & row_mask; }
        u64x2 v[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) v[k] = *reinterpret_cast<const u64x2*>(t + (size_t)row[k] * 16);
#pragma unroll
        for (int k = 0; k < PER; ++k) acc ^= v[k].x ^ v[k].y;
    }
    if (acc == 0x123456789ull) out[0] = acc;
}
template <int PER>
static void run(const unsigned char* t, size_t table_bytes, int grid, u64* out) {
    const uint32_t mask = (uint32_t)(table_bytes / 16) - 1u;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k_gather<PER>, dim3(grid), dim3(1024), 0, 0, t, mask, out);
    (void)hipDeviceSynchronize();
    const int reps = 5;
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_gather<PER>, dim3(grid), dim3(1024), 0, 0, t, mask, out);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    const double reqs = (double)grid * 1024 * ROUNDS * 4, us = ms * 1e3 / reps;
    const int cus = grid < 256 ? grid : 256;
    std::printf("table %6.1f MB  %d loads in flight per lane  grid %4d (%3d CUs)  %8.1f us/launch  %6.2f ns/request/active CU  %6.1f G requests/s chip-wide\n",
                table_bytes / 1048576.0, PER, grid, cus, us, us * 1e3 / (reqs / cus), reqs / us / 1e3);
}
int main() {
    const size_t big = (size_t)64 << 20;
    unsigned char* t;
    (void)hipMalloc(&t, big + 64);
    (void)hipMemset(t, 1, big + 64);
    u64* out;
    (void)hipMalloc(&out, 64);
    for (size_t mb : {16, 32}) {
        for (int grid : {32, 64, 128, 256, 512}) run<4>(t, mb << 20, grid, out);
        for (int grid : {64, 256}) run<1>(t, mb << 20, grid, out);
        for (int grid : {64, 256}) run<2>(t, mb << 20, grid, out);
    }
    return 0;
}
