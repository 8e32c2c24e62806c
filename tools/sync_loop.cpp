// The reference-shaped loop — step; effective_sample_size() -> f64; resample() -> f64, every call synchronous
// (modppl/src/inference/particle_filter.rs:73-116, tests/smc.rs:64-90) — driven from COMPILED host code through the C++ wrapper
// (modppl_amd/cpp/modppl.hpp over the C ABI), as a Rust host would drive it: what bench.py's `reference_shaped_loop` measures from
// Python minus the interpreter's 8 - 10 us per iteration, which sit on the critical path of a loop whose every call waits.
//   g++ -std=c++17 -O2 tools/sync_loop.cpp -o /tmp/sync_loop -Lmodppl_amd/csrc -lmodppl_hip -Wl,-rpath,$PWD/modppl_amd/csrc && /tmp/sync_loop [particles] [steps]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../modppl_amd/cpp/modppl.hpp"

int main(int argc, char** argv) {
    const uint64_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : (1ull << 20);
    const int K = argc > 2 ? std::atoi(argv[2]) : 200, W = 20;
    // observations simulated from the model (mu0 0, sig0 1, a 0.9, sig_x 0.5, sig_y 1: bench.py's LGSSM_PARAMS)
    std::mt19937_64 g(20241008);
    std::normal_distribution<double> z(0., 1.);
    std::vector<double> ys(1 + W + K);
    double x = z(g);
    for (size_t t = 0; t < ys.size(); ++t) {
        if (t) x = 0.9 * x + 0.5 * z(g);
        ys[t] = x + z(g);
    }
    try {
        modppl::ParticleSystem pf(modppl::UnfoldModel::lgssm(), n, 20241008);
        pf.init_step({}, {ys[0]});
        double sumL = 0., sumE = 0.;
        for (int rep = 0; rep < 3; ++rep) {
            for (int t = 1; t <= W; ++t) { pf.step({ys[t]}); sumE += pf.effective_sample_size(); sumL += pf.resample(); }
            const auto t0 = std::chrono::steady_clock::now();
            for (int t = 1 + W; t <= W + K; ++t) {
                pf.step({ys[t]});
                sumE += pf.effective_sample_size();
                sumL += pf.resample();
            }
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / K;
            std::printf("reference-shaped loop from C++: %.1f us per step (%llu particles, %d steps, every call synchronous)\n", us, (unsigned long long)n, K);
        }
        std::printf("checksums: sum L %.6f, mean ESS %.1f\n", sumL, sumE / (3.0 * (W + K)));
    } catch (const modppl::Panic& p) {
        std::fprintf(stderr, "panic %d: %s\n", p.code, p.what());
        return 2;
    }
    return 0;
}
