// Launch-shape probe: how long does an (almost) empty kernel of a given shape take on MI355X?  The floor that any kernel of that
// shape inherits from workgroup dispatch and drain.   hipcc --offload-arch=gfx950 -O3 tools/launch_probe.hip -o tools/launch_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int LDS_BYTES>
__global__ void k_probe(double* out, int work) {
    __shared__ unsigned char lds[LDS_BYTES > 0 ? LDS_BYTES : 1];
    double acc = threadIdx.x;
    for (int i = 0; i < work; ++i) acc = acc * 1.0000001 + 1e-9;   // `work` dependent fp64 FMAs per lane
    if (LDS_BYTES > 0) { lds[threadIdx.x % (LDS_BYTES > 0 ? LDS_BYTES : 1)] = (unsigned char)acc; __syncthreads(); acc += lds[0]; }
    if (acc == 12345.678) out[0] = acc;   // never true: keeps the loop alive
}

template <int LDS>
static void run(const char* name, int grid, int block, int work, double* d) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_probe<LDS>, dim3(grid), dim3(block), 0, 0, d, work);
    hipDeviceSynchronize();
    const int reps = 50;
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_probe<LDS>, dim3(grid), dim3(block), 0, 0, d, work);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    std::printf("%-34s grid %5d x %4d  work %5d : %7.2f us per launch (back to back)\n", name, grid, block, work, ms * 1e3 / reps);
}

int main() {
    double* d;
    hipMalloc(&d, 64);
    for (int work : {0, 1000, 4000}) {
        run<0>("512 x 1024 (K1 light models)", 512, 1024, work, d);
        run<4288>("512 x 1024 + 4.3 KB LDS", 512, 1024, work, d);
        run<0>("512 x 512", 512, 512, work, d);
        run<0>("1024 x 512 (K3a)", 1024, 512, work, d);
        run<0>("2048 x 256", 2048, 256, work, d);
        run<0>("4096 x 256 (K3b)", 4096, 256, work, d);
        run<0>("8192 x 128", 8192, 128, work, d);
        run<0>("256 x 1024", 256, 1024, work, d);
    }
    return 0;
}
