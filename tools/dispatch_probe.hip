// When do the workgroups of a 512 x 1024 launch START, XCD by XCD — right after an idle queue, right behind a kernel that
// left nothing dirty, and right behind one that wrote 48 MB (what a k_propagate leaves in the L2s)?  Every workgroup stamps the
// 100 MHz real-time counter at its first and last instruction.
//   hipcc --offload-arch=gfx950 -O3 tools/dispatch_probe.hip -o tools/dispatch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

typedef unsigned long long u64;

__global__ __launch_bounds__(1024) void k_writer(double* out, int per_thread, double v) {
    // per_thread doubles per thread, coalesced: 512 x 1024 x per_thread x 8 bytes dirty in the L2s at the end
    const size_t i0 = (size_t)blockIdx.x * 1024 * per_thread + threadIdx.x;
    for (int k = 0; k < per_thread; ++k) out[i0 + (size_t)k * 1024] = v + k;
}
__global__ __launch_bounds__(1024) void k_stamped(u64* stamps, int spin, double* sink, int lds_words) {
    extern __shared__ unsigned int lds[];
    u64 t0, t1;
    unsigned int xcc;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    double acc = threadIdx.x;
    for (int i = 0; i < spin; ++i) acc = acc * 1.0000001 + 1e-9;
    if (lds_words > 0) lds[threadIdx.x % lds_words] = (unsigned int)acc;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) {
        stamps[3 * blockIdx.x] = t0;
        stamps[3 * blockIdx.x + 1] = t1;
        stamps[3 * blockIdx.x + 2] = xcc & 0xF;
    }
    if (acc == 12345.678) sink[0] = acc;
}

static void report(const char* name, const std::vector<u64>& h, int n) {
    u64 first = ~0ull, last_end = 0;
    for (int b = 0; b < n; ++b) { first = std::min(first, h[3 * b]); last_end = std::max(last_end, h[3 * b + 1]); }
    double xs[8] = {0}, xe[8] = {0};
    int cnt[8] = {0};
    double smax = 0;
    for (int b = 0; b < n; ++b) {
        const int x = (int)h[3 * b + 2];
        const double s = (double)(h[3 * b] - first) / 100.0, e = (double)(h[3 * b + 1] - first) / 100.0;
        xs[x] += s; xe[x] += e; cnt[x]++;
        smax = std::max(smax, s);
    }
    std::printf("%-46s span %6.2f us, last start %5.2f us; mean start by XCD:", name, (double)(last_end - first) / 100.0, smax);
    for (int x = 0; x < 8; ++x) std::printf(" %4.2f", cnt[x] ? xs[x] / cnt[x] : -1.);
    std::printf("\n");
}

int main() {
    const int n = 512;
    u64* d;
    double *big, *sink;
    (void)hipMalloc(&d, sizeof(u64) * 3 * n);
    (void)hipMalloc(&big, (size_t)512 * 1024 * 16 * 8);
    (void)hipMalloc(&sink, 64);
    std::vector<u64> h(3 * n);
    for (int spin : {0, 20000}) {
        for (int lds : {0, 2304}) {
            for (int pred = 0; pred < 4; ++pred) {
                // pred 0: idle queue; 1: behind a stamped kernel (nothing dirty); 2: behind a writer of 8 MB; 3: behind a writer of 48 MB
                for (int rep = 0; rep < 3; ++rep) {
                    (void)hipDeviceSynchronize();
                    if (pred == 1) hipLaunchKernelGGL(k_stamped, dim3(n), dim3(1024), lds * 4, 0, d, spin, sink, lds);
                    if (pred == 2) hipLaunchKernelGGL(k_writer, dim3(n), dim3(1024), 0, 0, big, 2, 1.0 + rep);
                    if (pred == 3) hipLaunchKernelGGL(k_writer, dim3(n), dim3(1024), 0, 0, big, 12, 1.0 + rep);
                    hipLaunchKernelGGL(k_stamped, dim3(n), dim3(1024), lds * 4, 0, d, spin, sink, lds);
                    (void)hipDeviceSynchronize();
                }
                (void)hipMemcpy(h.data(), d, sizeof(u64) * 3 * n, hipMemcpyDeviceToHost);
                char name[128];
                std::snprintf(name, sizeof name, "spin %5d, LDS %5d B, predecessor %s", spin, lds * 4,
                              pred == 0 ? "none (idle)" : pred == 1 ? "same kernel" : pred == 2 ? "writer 8 MB" : "writer 48 MB");
                report(name, h, n);
            }
        }
    }
    return 0;
}
