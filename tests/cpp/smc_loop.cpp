// The reference's SMC loop (modppl/tests/smc.rs:64-90) through the C++ wrapper (modppl_amd/cpp/modppl.hpp):
//   new -> init_step -> resample -> (step -> resample)*, then the log-ML estimate and a state checksum.
// Prints one line "lml=<%.17g> x0=<%.17g> parents0=<u>" that tests/test_cpp_wrapper.py compares with the Python mirror
// (both sit on the same C ABI, so the values must be identical).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../modppl_amd/cpp/modppl.hpp"

int main(int argc, char** argv) {
    const uint64_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 4096;
    const uint64_t seed = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 7;
    const std::vector<double> ys = {0.31, -0.12, 0.58, 1.02, 0.44, -0.27};
    try {
        modppl::ParticleSystem filter(modppl::UnfoldModel::lgssm(), n, seed);
        filter.init_step({}, {ys[0]});
        filter.resample();
        for (size_t t = 1; t < ys.size(); ++t) {
            filter.step({ys[t]});
            filter.resample();
        }
        const double lml = filter.log_marginal_likelihood_estimate();
        const std::vector<double> x = filter.states();
        const std::vector<uint32_t> par = filter.parents();
        std::printf("lml=%.17g x0=%.17g parents0=%u\n", lml, x[0], par[0]);
        modppl::HierarchicalChains chains({-1., 0., 1.}, {0.2, 0.9, 2.2}, 256, seed);
        const uint64_t a1 = chains.mh_add_or_remove(2);
        const uint64_t a2 = chains.mh(0.1, 3);
        std::printf("accepted=%llu,%llu\n", (unsigned long long)a1, (unsigned long long)a2);
        // a registered functor model (kind 101 = the hierarchical model again) through the generic entry points: same moves, same counts
        modppl::FunctionChains fchains(MP_MH_MODEL_HIERARCHICAL_FN, {-1., 0., 1.}, {{MP_SITE_Y0, 0.2}, {MP_SITE_Y0 + 1, 0.9}, {MP_SITE_Y0 + 2, 2.2}}, 256, seed);
        const uint64_t f1 = fchains.mh(MP_MH_PROPOSAL_HIERARCHICAL_ADD_OR_REMOVE, {}, 2);
        const uint64_t f2 = fchains.mh(MP_MH_PROPOSAL_HIERARCHICAL_DRIFT, {0.1}, 3);
        std::printf("accepted_fn=%llu,%llu sites=%d\n", (unsigned long long)f1, (unsigned long long)f2, fchains.num_sites());
        // (round 5) generate / simulate on a handle, and importance sampling over a registered function with DECLARED data sites (kind 105:
        // the same model, its observations as data) — importance.rs:12-50 for `impl GenFn`
        {
            const std::vector<std::pair<int32_t, double>> obs = {{MP_SITE_Y0, 0.2}, {MP_SITE_Y0 + 1, 0.9}, {MP_SITE_Y0 + 2, 2.2}};
            const std::vector<double> w = fchains.generate(obs, 40);
            const std::vector<double> lj = fchains.simulate(41);
            modppl::FnImportance is = modppl::fn_importance_resampling(MP_MH_MODEL_HIERARCHICAL_DATA_FN, {-1., 0., 1.}, obs, 2048, 8, seed);
            std::printf("fn w0=%.17g lj0=%.17g lml=%.17g idx0=%llu idx7=%llu lnw0=%.17g sites=%d\n", w[0], lj[0], is.log_ml_estimate,
                        (unsigned long long)is.resampled_indices[0], (unsigned long long)is.resampled_indices[7], is.log_normalized_weights[0],
                        is.traces->num_sites());
        }
        // the sharded filter in a world of one with both RCCL collectives forced: one library call per resample
        if (argc > 3 && std::atoi(argv[3])) {
            modppl::ShardedParticleSystem sf(modppl::UnfoldModel::lgssm(), n, seed, 1, 0, nullptr, nullptr, 0, true);
            sf.init_step({}, {ys[0]});
            sf.resample();
            for (size_t t = 1; t < ys.size(); ++t) {
                sf.step({ys[t]});
                sf.resample();
            }
            std::printf("sharded lml=%.17g\n", sf.log_marginal_likelihood_estimate());
        }
    } catch (const modppl::Panic& p) {
        std::fprintf(stderr, "panic %d: %s\n", p.code, p.what());
        return 2;
    }
    return 0;
}
