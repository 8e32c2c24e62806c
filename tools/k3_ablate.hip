// tools/k3_ablate.hip — timing-only microbenchmark of the step kernels (not part of the product).
// HISTORICAL: written against the kernel signatures of mid round 1 (before k_bin_draws<PREBUILT> and the particle-major
// k_resolve_bins); it produced profiles/r01/k3_ablation_n2e20.txt and no longer compiles against the current kernels.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -o tools/k3_ablate tools/k3_ablate.hip
#include "../modppl_amd/csrc/mp_pf.hip"
#include <cstdio>

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
    (void)hipStreamSynchronize(h->stream);
    auto timeit = [&](const char* name, auto fn) {
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        fn(); (void)hipEventRecord(a, h->stream);
        for (int it = 0; it < 20; ++it) fn();
        (void)hipEventRecord(b, h->stream); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        printf("%-34s: %8.2f us\n", name, ms / 20 * 1e3);
    };
    const int dd = 1;
    const size_t lds_a = table_lds(h->nt, BIN_THREADS) + sizeof(uint32_t) * 2 * BIN_ITEMS * (BIN_THREADS / 64) * 8;
    const int ngroups = (h->nchunks + BIN_GROUP - 1) / BIN_GROUP;
    timeit("K1 propagate+normalize (slot in)", [&] { h->t = 5; h->permuted = false; launch_propagate(h, nullptr, &y1, false); });
    timeit("K1 propagate+normalize (perm in)", [&] { h->t = 5; h->permuted = true; launch_propagate(h, nullptr, &y1, false); });
    timeit("k_normalize_tiles (standalone)", [&] { hipLaunchKernelGGL(k_normalize_tiles, dim3(h->nt), dim3(TILE_THREADS), 0, h->stream, h->logw, h->x[h->cur], h->n, h->cx, h->guide, h->tile_m, h->tile_W, h->tile_W2); });
    timeit("K3a bin draws", [&] { hipLaunchKernelGGL(k_bin_draws, dim3(h->nchunks), dim3(BIN_THREADS), lds_a, h->stream, h->n, h->n_global, h->slot_offset, 1u, 2u, 3u, h->S, h->nchunks, h->tile_m, h->tile_W, h->tile_W2, h->nt, h->guide, h->seg_lt, h->seg_row, h->perm, h->seg_cnt, h->scal); });
    timeit("K3b resolve bins", [&] { hipLaunchKernelGGL(k_resolve_bins, dim3(ngroups * 8), dim3(K3_THREADS), 0, h->stream, h->n, dd, h->nchunks, h->seg_lt, h->seg_row, h->seg_cnt, h->cx, h->x[h->cur], h->res_x, h->res_stride, h->res_parent); });
    timeit("K3 single kernel (multinomial)", [&] { hipLaunchKernelGGL(k_resample_gather<false>, dim3(h->k3_grid), dim3(K3_THREADS), table_lds(h->nt, K3_THREADS), h->stream, h->n, h->n, h->n_global, h->slot_offset, (uint32_t)MP_DOM_RESAMPLE, 1u, 2u, 3u, h->S, dd, h->cx, h->guide, h->tile_m, h->tile_W, h->tile_W2, h->nt, h->x[h->cur], h->x[h->cur ^ 1], h->parent, h->aos, (mp_dev_scalars*)nullptr); });
    timeit("K3 single kernel (systematic)", [&] { hipLaunchKernelGGL(k_resample_gather<true>, dim3(h->k3_grid), dim3(K3_THREADS), table_lds(h->nt, K3_THREADS), h->stream, h->n, h->n, h->n_global, h->slot_offset, (uint32_t)MP_DOM_RESAMPLE, 1u, 2u, 3u, h->S, dd, h->cx, h->guide, h->tile_m, h->tile_W, h->tile_W2, h->nt, h->x[h->cur], h->x[h->cur ^ 1], h->parent, h->aos, (mp_dev_scalars*)nullptr); });
    mp_pf_destroy(h);
    return 0;
}
