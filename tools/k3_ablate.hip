// tools/k3_ablate.hip — timing-only ablation of the resample kernels (not part of the product).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -o /tmp/k3_ablate tools/k3_ablate.hip
#include "../modppl_amd/csrc/mp_pf.hip"
#include <cstdio>
#include <cmath>
#include <random>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int ABL>
float run_k3(mp_pf* h, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const size_t lds = sizeof(u64) * ((size_t)h->nt + K3_THREADS / 64);
    hipEventRecord(a, h->stream);
    for (int it = 0; it < iters; ++it)
        hipLaunchKernelGGL((k_resample_gather<ABL, false>), dim3(h->k3_grid), dim3(K3_THREADS), lds, h->stream, h->n, h->n, h->n_global, h->slot_offset,
                           (uint32_t)MP_DOM_RESAMPLE, 1u, 2u, (uint32_t)it, h->S, 1, h->cx, h->guide, h->tilesum, h->tilesum2, h->nt, h->x[0], h->x[1], h->parent,
                           h->aos /* scratch instead of logw */, h->blockmax + 1024, 0, h->scal);
    hipEventRecord(b, h->stream);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / iters * 1e3f;
}

int main(int argc, char** argv) {
    const u64 n = argc > 1 ? strtoull(argv[1], 0, 10) : (1ull << 20);
    double params[5] = {0, 1, 0.9, 0.5, 1.0};
    mp_model_desc d{MP_MODEL_LGSSM1, 1, 1, 5, params};
    mp_pf* h;
    if (mp_pf_create(&d, n, 7, nullptr, 0, 0, nullptr, &h)) { printf("create: %s\n", mp_last_error()); return 1; }
    double y0 = 0.3, y1 = 0.7;
    mp_pf_init_step(h, nullptr, &y0, 1);
    mp_pf_resample(h, 0, nullptr);
    mp_pf_step(h, &y1, 1);
    launch_normalize(h);
    CK(hipStreamSynchronize(h->stream));
    // K2 timing
    {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, h->stream);
        for (int it = 0; it < 20; ++it) launch_normalize(h);
        hipEventRecord(b, h->stream); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("K2 normalize_scan           : %8.2f us\n", ms / 20 * 1e3);
    }
    run_k3<0>(h, 3);
    printf("K3 full                     : %8.2f us\n", run_k3<0>(h, 20));
    printf("K3 no global reads (ALU+LDS): %8.2f us\n", run_k3<1>(h, 20));
    printf("K3 guide only               : %8.2f us\n", run_k3<2>(h, 20));
    printf("K3 floor - Philox           : %8.2f us\n", run_k3<3>(h, 20));
    printf("K3 floor - LDS search       : %8.2f us\n", run_k3<4>(h, 20));
    printf("K3 floor - tile scan        : %8.2f us\n", run_k3<5>(h, 20));
    printf("K3 full, draws in XCD eighth: %8.2f us\n", run_k3<6>(h, 20));
    printf("K3 full again               : %8.2f us\n", run_k3<0>(h, 20));
    {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        const size_t lds = sizeof(u64) * ((size_t)h->nt + K3_THREADS / 64);
        hipEventRecord(a, h->stream);
        for (int it = 0; it < 20; ++it)
            hipLaunchKernelGGL((k_resample_gather<0, true>), dim3(h->k3_grid), dim3(K3_THREADS), lds, h->stream, h->n, h->n, h->n_global, h->slot_offset,
                               (uint32_t)MP_DOM_RESAMPLE, 1u, 2u, (uint32_t)it, h->S, 1, h->cx, h->guide, h->tilesum, h->tilesum2, h->nt, h->x[0], h->x[1],
                               h->parent, h->aos, h->blockmax + 1024, 0, h->scal);
        hipEventRecord(b, h->stream); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("K3 systematic (coalesced)   : %8.2f us\n", ms / 20 * 1e3);
    }
    // binned path pieces
    {
        auto timeit = [&](const char* name, auto fn) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            fn(); hipEventRecord(a, h->stream);
            for (int it = 0; it < 20; ++it) fn();
            hipEventRecord(b, h->stream); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("%-28s: %8.2f us\n", name, ms / 20 * 1e3);
        };
        const size_t lds = sizeof(u64) * ((size_t)h->nt + K3_THREADS / 64);
        const int ngroups = (h->nchunks + BIN_GROUP - 1) / BIN_GROUP;
        const size_t lds_a = sizeof(u64) * ((size_t)h->nt + BIN_THREADS / 64) + sizeof(uint32_t) * (2 * BIN_ITEMS * (BIN_THREADS / 64) * 8 + 8);
        auto k3a = [&] { hipLaunchKernelGGL(k_bin_draws, dim3(h->nchunks), dim3(BIN_THREADS), lds_a, h->stream, h->n, h->n_global, h->slot_offset, 1u, 2u, 3u, h->S, h->nchunks, h->tilesum, h->tilesum2, h->nt, h->guide, h->seg_lt, h->seg_row, h->perm, h->seg_cnt, h->blockmax + 1024, 0, h->scal); };
#define K3B(V) [&] { hipLaunchKernelGGL(k_resolve_bins<V>, dim3(ngroups * 8), dim3(K3_THREADS), 0, h->stream, h->n, 1, h->nchunks, h->seg_lt, h->seg_row, h->seg_cnt, h->cx, h->x[0], h->res_x, h->res_stride, h->res_parent); }
        timeit("K3a bin draws", k3a);
        timeit("K3b resolve bins", K3B(0));
        timeit("K3b no result stores", K3B(1));
        timeit("K3b no guide/row loads", K3B(4));
        h->permuted = true;
        timeit("K1 propagate (perm input)", [&] { h->t = 5; h->permuted = true; launch_propagate(h, nullptr, &y1, false); });
    }
    // K1 timing
    {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, h->stream);
        for (int it = 0; it < 20; ++it) { h->t = 5; launch_propagate(h, nullptr, &y1, false); }
        hipEventRecord(b, h->stream); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("K1 propagate                : %8.2f us\n", ms / 20 * 1e3);
    }
    mp_pf_destroy(h);
    return 0;
}
