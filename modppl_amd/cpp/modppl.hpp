// modppl.hpp — header-only C++ host wrapper over the C ABI (include/modppl_hip.h), mirroring the reference's names:
//   modppl::ParticleSystem   ::new / init_step / step / effective_sample_size / resample /
//                            log_marginal_likelihood_estimate      (modppl/src/inference/particle_filter.rs:44-121)
//   modppl::importance_resampling                                   (modppl/src/inference/importance.rs:37-50)
//   modppl::HierarchicalChains::mh / regen_mh                       (modppl/src/inference/mh.rs:9-75)
// A reference `panic!` becomes a modppl::Panic exception carrying the status code.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <memory>
#include <vector>

#include "../../include/modppl_hip.h"

namespace modppl {

struct Panic : std::runtime_error {
    int32_t code;
    Panic(int32_t c, const std::string& m) : std::runtime_error(m), code(c) {}
};
inline void check(int32_t rc) {
    if (rc != MP_OK) throw Panic(rc, mp_last_error());
}

// Counterpart of DynUnfold<State>: a model descriptor {kind, dims, params}.
struct UnfoldModel {
    int32_t kind, dim_state, dim_obs;
    std::vector<double> params;
    mp_model_desc desc() const { return mp_model_desc{kind, dim_state, dim_obs, (int32_t)params.size(), params.data()}; }
    static UnfoldModel lgssm(double mu0 = 0., double sig0 = 1., double a = 0.9, double sig_x = 0.5, double sig_y = 1.) {
        return {MP_MODEL_LGSSM1, 1, 1, {mu0, sig0, a, sig_x, sig_y}};
    }
    static UnfoldModel spiral() { return {MP_MODEL_SPIRAL, 2, 2, {}}; }
    static UnfoldModel bearings(double p0x = 1., double p0y = 1., double sig_p0 = 1., double sig_v0 = .1, double sig_a = .05, double sig_th = .02) {
        return {MP_MODEL_BEARINGS, 4, 1, {p0x, p0y, sig_p0, sig_v0, sig_a, sig_th}};
    }
    static UnfoldModel lgssm_band(int D = 16, double a = 0.9, double band = 0.05, double sig0 = 1., double sig_x = 0.5, double sig_y = 1.) {
        return {MP_MODEL_LGSSM_BAND, D, D, {(double)D, a, band, sig0, sig_x, sig_y}};
    }
};

class ParticleSystem {
    mp_pf* h_ = nullptr;
    UnfoldModel model_;
    uint64_t n_;

public:
    // ParticleSystem::new(model, num_particles, rng): `seed` replaces the unseedable ThreadRng
    ParticleSystem(UnfoldModel model, uint64_t num_particles, uint64_t seed, uint32_t flags = 0, int device = 0, void* stream = nullptr)
        : model_(std::move(model)), n_(num_particles) {
        const mp_model_desc d = model_.desc();
        check(mp_pf_create(&d, num_particles, seed, nullptr, flags, device, stream, &h_));
    }
    ParticleSystem(const ParticleSystem&) = delete;
    ParticleSystem& operator=(const ParticleSystem&) = delete;
    ~ParticleSystem() { mp_pf_destroy(h_); }

    void init_step(const std::vector<double>& args, const std::vector<double>& constraints) {
        check(mp_pf_init_step(h_, args.empty() ? nullptr : args.data(), constraints.data(), (int32_t)(constraints.size() / model_.dim_obs)));
    }
    ParticleSystem& step(const std::vector<double>& constraints) {
        check(mp_pf_step(h_, constraints.data(), (int32_t)(constraints.size() / model_.dim_obs)));
        return *this;
    }
    double effective_sample_size(bool fresh = false) {
        double v;
        check(mp_pf_effective_sample_size(h_, fresh ? MP_ESS_FRESH : MP_ESS_REFERENCE, &v));
        return v;
    }
    double resample(int32_t scheme = MP_RESAMPLE_MULTINOMIAL) {
        double v;
        check(mp_pf_resample(h_, scheme, &v));
        return v;
    }
    void resample_async(int32_t scheme = MP_RESAMPLE_MULTINOMIAL) { check(mp_pf_resample(h_, scheme, nullptr)); }
    double log_marginal_likelihood_estimate() {
        double v;
        check(mp_pf_log_marginal_likelihood_estimate(h_, &v));
        return v;
    }
    // traces[i].retv.last() for all i, row-major [n][dim_state]
    std::vector<double> states() {
        std::vector<double> x(n_ * (size_t)model_.dim_state);
        check(mp_pf_read_state(h_, x.data()));
        return x;
    }
    std::vector<double> log_weights() {
        std::vector<double> w(n_);
        check(mp_pf_read_log_weights(h_, w.data()));
        return w;
    }
    std::vector<uint32_t> parents() {
        std::vector<uint32_t> p(n_);
        check(mp_pf_read_parents(h_, p.data()));
        return p;
    }
    void synchronize() { check(mp_pf_synchronize(h_)); }
    mp_pf* raw() { return h_; }
};

struct ImportanceResult {
    std::vector<double> final_states, log_normalized_weights;
    std::vector<uint64_t> resampled_indices;
    double log_ml_estimate;
};
inline ImportanceResult importance_resampling(const UnfoldModel& model, const std::vector<double>& model_args, const std::vector<double>& constraints,
                                              uint64_t num_samples, uint64_t num_ret_samples, uint64_t seed, int device = 0) {
    ImportanceResult r;
    r.final_states.resize(num_samples * (size_t)model.dim_state);
    r.log_normalized_weights.resize(num_samples);
    r.resampled_indices.resize(num_ret_samples);
    const mp_model_desc d = model.desc();
    check(mp_importance_resampling(&d, model_args.empty() ? nullptr : model_args.data(), constraints.data(),
                                   (int32_t)(constraints.size() / model.dim_obs), num_samples, num_ret_samples, seed, device, &r.log_ml_estimate,
                                   r.log_normalized_weights.data(), num_ret_samples ? r.resampled_indices.data() : nullptr, r.final_states.data()));
    return r;
}

class HierarchicalChains {
    mp_mh* h_ = nullptr;
    uint64_t n_;

public:
    HierarchicalChains(const std::vector<double>& xs, const std::vector<double>& ys, uint64_t num_chains, uint64_t seed, int constrain_is_linear = -1,
                       int device = 0, void* stream = nullptr)
        : n_(num_chains) {
        check(mp_mh_create(MP_MH_MODEL_HIERARCHICAL, xs.data(), ys.data(), (int32_t)xs.size(), constrain_is_linear, num_chains, seed, device, stream, &h_));
    }
    HierarchicalChains(const HierarchicalChains&) = delete;
    ~HierarchicalChains() { mp_mh_destroy(h_); }
    uint64_t mh(double drift_std, int32_t n_iters = 1) {
        uint64_t acc;
        check(mp_mh_step(h_, MP_MH_PROPOSAL_HIERARCHICAL_DRIFT, &drift_std, 1, n_iters, &acc));
        return acc;
    }
    // mh(&hierarchical_model, trace, &add_or_remove_param_proposal, ()) — tests/mh.rs:94
    uint64_t mh_add_or_remove(int32_t n_iters = 1) {
        uint64_t acc;
        check(mp_mh_step(h_, MP_MH_PROPOSAL_HIERARCHICAL_ADD_OR_REMOVE, nullptr, 0, n_iters, &acc));
        return acc;
    }
    uint64_t regen_mh(const std::vector<int32_t>& mask_sites, int32_t n_iters = 1, bool cycle = false) {
        uint64_t acc;
        check(mp_regen_mh_step(h_, mask_sites.data(), (int32_t)mask_sites.size(), cycle ? 1 : 0, n_iters, &acc));
        return acc;
    }
    std::vector<double> states() {
        std::vector<double> s(n_ * 4);
        check(mp_mh_read_state(h_, s.data()));
        return s;
    }
};

// mh / regen_mh over a REGISTERED generative function (csrc/mp_mh_models.h: MP_REGISTER_MH_MODEL / _PROPOSAL) — what calling the
// reference's mh(&model, trace, &proposal, args) / regen_mh(&model, trace, &mask) with functions of one's own becomes.
class FunctionChains {
    mp_mh* h_ = nullptr;
    uint64_t n_;
    int32_t ns_ = 0;

public:
    // constraints = (site id, value) pairs: creation runs model.generate(params, constraints) per chain
    FunctionChains(int32_t model_kind, const std::vector<double>& params, const std::vector<std::pair<int32_t, double>>& constraints, uint64_t num_chains,
                   uint64_t seed, int device = 0, void* stream = nullptr)
        : n_(num_chains) {
        std::vector<int32_t> sites;
        std::vector<double> vals;
        for (const auto& c : constraints) { sites.push_back(c.first); vals.push_back(c.second); }
        check(mp_mh_create_fn(model_kind, params.empty() ? nullptr : params.data(), (int32_t)params.size(), sites.empty() ? nullptr : sites.data(),
                              vals.empty() ? nullptr : vals.data(), (int32_t)sites.size(), num_chains, seed, device, stream, &h_));
        if (mp_mh_n_sites(h_, &ns_) != MP_OK) {   // (the destructor does not run for a half-constructed object)
            const std::string why = mp_last_error();
            mp_mh_destroy(h_);
            throw std::runtime_error(why);
        }
    }
    // a handle the library returned (the traces of fn_importance_*): sample i = chain i
    FunctionChains(mp_mh* adopted, uint64_t num_chains) : h_(adopted), n_(num_chains) {
        if (mp_mh_n_sites(h_, &ns_) != MP_OK) {
            const std::string why = mp_last_error();
            mp_mh_destroy(h_);
            throw std::runtime_error(why);
        }
    }
    FunctionChains(const FunctionChains&) = delete;
    ~FunctionChains() { mp_mh_destroy(h_); }
    int32_t num_sites() const { return ns_; }
    // (trace, weight) = model.generate(args, constraints) on every chain, shared constraints (gfi.rs:53-55); the traces are replaced
    std::vector<double> generate(const std::vector<std::pair<int32_t, double>>& constraints, uint32_t rng_step = 0) {
        std::vector<int32_t> sites;
        std::vector<double> vals, w(n_);
        for (const auto& c : constraints) { sites.push_back(c.first); vals.push_back(c.second); }
        check(mp_fn_generate(h_, rng_step, sites.empty() ? nullptr : sites.data(), vals.empty() ? nullptr : vals.data(), (int32_t)sites.size(), nullptr, nullptr,
                             w.data()));
        return w;
    }
    // trace = model.simulate(args) on every chain (gfi.rs:51) -> each new trace's logjp
    std::vector<double> simulate(uint32_t rng_step = 0) {
        std::vector<double> lj(n_);
        check(mp_fn_simulate(h_, rng_step, lj.data()));
        return lj;
    }
    uint64_t mh(int32_t proposal_kind, const std::vector<double>& args = {}, int32_t n_iters = 1) {
        uint64_t acc;
        check(mp_mh_step(h_, proposal_kind, args.empty() ? nullptr : args.data(), (int32_t)args.size(), n_iters, &acc));
        return acc;
    }
    uint64_t regen_mh(const std::vector<int32_t>& mask_sites, int32_t n_iters = 1, bool cycle = false) {
        uint64_t acc;
        check(mp_regen_mh_step(h_, mask_sites.empty() ? nullptr : mask_sites.data(), (int32_t)mask_sites.size(), cycle ? 1 : 0, n_iters, &acc));
        return acc;
    }
    // values[chain][site] (0 where absent), present[chain][words()] (32-bit words, bit k of the chain's words = site k is in the trace;
    // ONE word per chain for models of up to 32 sites)
    size_t words() const { return ((size_t)ns_ + 31) / 32; }
    void trace(std::vector<double>& values, std::vector<uint32_t>& present) {
        values.resize(n_ * (size_t)ns_);
        present.resize(n_ * words());
        check(mp_mh_read_trace(h_, values.data(), present.data()));
    }
    // ---- GenFn::update / regenerate / assess / propose one at a time, every chain per call (gfi.rs:57-90; mp_fn_* of the C ABI) ----
    // A per-chain table of choices: values[chain][site], present[chain][words()] — what propose() and update()'s discard return.
    struct Choices {
        std::vector<double> values;
        std::vector<uint32_t> present;
    };
    // shared constraints {(site, value)}; the chains' traces are replaced; -> weights (the discard into *discard when asked for)
    std::vector<double> update(const std::vector<std::pair<int32_t, double>>& constraints, int32_t argdiff = MP_ARGDIFF_NOCHANGE, uint32_t rng_step = 0,
                               Choices* discard = nullptr) {
        std::vector<int32_t> sites;
        std::vector<double> vals, w(n_);
        for (const auto& c : constraints) { sites.push_back(c.first); vals.push_back(c.second); }
        if (discard) { discard->values.resize(n_ * (size_t)ns_); discard->present.resize(n_ * words()); }
        check(mp_fn_update(h_, argdiff, rng_step, sites.empty() ? nullptr : sites.data(), vals.empty() ? nullptr : vals.data(), (int32_t)sites.size(), nullptr,
                           nullptr, w.data(), discard ? discard->values.data() : nullptr, discard ? discard->present.data() : nullptr));
        return w;
    }
    // per-chain constraints (e.g. the choices of propose())
    std::vector<double> update(const Choices& constraints, int32_t argdiff = MP_ARGDIFF_NOCHANGE, uint32_t rng_step = 0, Choices* discard = nullptr) {
        std::vector<double> w(n_);
        if (discard) { discard->values.resize(n_ * (size_t)ns_); discard->present.resize(n_ * words()); }
        check(mp_fn_update(h_, argdiff, rng_step, nullptr, nullptr, 0, constraints.values.data(), constraints.present.data(), w.data(),
                           discard ? discard->values.data() : nullptr, discard ? discard->present.data() : nullptr));
        return w;
    }
    std::vector<double> regenerate(const std::vector<int32_t>& mask_sites, int32_t argdiff = MP_ARGDIFF_NOCHANGE, uint32_t rng_step = 0) {
        std::vector<double> w(n_);
        check(mp_fn_regenerate(h_, argdiff, rng_step, mask_sites.empty() ? nullptr : mask_sites.data(), (int32_t)mask_sites.size(), w.data()));
        return w;
    }
    // proposal_kind < 0: the model's assess; otherwise the registered proposal's on each chain's current trace (mh.rs:25-27)
    std::vector<double> assess(const Choices& constraints, int32_t proposal_kind = -1, const std::vector<double>& args = {}, uint32_t rng_step = 0) {
        std::vector<double> w(n_);
        check(mp_fn_assess(h_, proposal_kind, args.empty() ? nullptr : args.data(), (int32_t)args.size(), rng_step, nullptr, nullptr, 0, constraints.values.data(),
                           constraints.present.data(), w.data()));
        return w;
    }
    std::vector<double> propose(int32_t proposal_kind, const std::vector<double>& args, Choices& choices, uint32_t rng_step = 0) {
        std::vector<double> w(n_);
        choices.values.resize(n_ * (size_t)ns_);
        choices.present.resize(n_ * words());
        check(mp_fn_propose(h_, proposal_kind, args.empty() ? nullptr : args.data(), (int32_t)args.size(), rng_step, choices.values.data(), choices.present.data(),
                            w.data()));
        return w;
    }
};

// One rank of a filter sharded over `world` GPUs (one process per GPU): same calls as ParticleSystem, and resample() is ONE
// library call that issues its RCCL collectives itself (mp_pf_shard_resample_rccl).  `comm`: the host's ncclComm_t, or null to
// let the library make a communicator of its own from `id128` (mp_rccl_unique_id on rank 0, handed to the others by the host).
// importance_sampling / importance_resampling over a registered generative function (importance.rs:12-50; mp_fn_importance_* of the C ABI)
struct FnImportance {
    std::unique_ptr<FunctionChains> traces;       // sample i = chain i
    std::vector<double> log_normalized_weights;   // [num_samples]
    std::vector<uint64_t> resampled_indices;      // [num_ret_samples]
    double log_ml_estimate = 0.;
};
inline FnImportance fn_importance_resampling(int32_t model_kind, const std::vector<double>& params, const std::vector<std::pair<int32_t, double>>& constraints,
                                             uint64_t num_samples, uint64_t num_ret_samples, uint64_t seed, int device = 0) {
    std::vector<int32_t> sites;
    std::vector<double> vals;
    for (const auto& c : constraints) { sites.push_back(c.first); vals.push_back(c.second); }
    FnImportance out;
    out.log_normalized_weights.resize(num_samples);
    out.resampled_indices.resize(num_ret_samples);
    mp_mh* h = nullptr;
    check(mp_fn_importance_resampling(model_kind, params.empty() ? nullptr : params.data(), (int32_t)params.size(), sites.empty() ? nullptr : sites.data(),
                                      vals.empty() ? nullptr : vals.data(), (int32_t)sites.size(), num_samples, num_ret_samples, seed, device, &out.log_ml_estimate,
                                      out.log_normalized_weights.data(), num_ret_samples ? out.resampled_indices.data() : nullptr, &h));
    out.traces.reset(new FunctionChains(h, num_samples));
    return out;
}
inline FnImportance fn_importance_sampling(int32_t model_kind, const std::vector<double>& params, const std::vector<std::pair<int32_t, double>>& constraints,
                                           uint64_t num_samples, uint64_t seed, int device = 0) {
    return fn_importance_resampling(model_kind, params, constraints, num_samples, 0, seed, device);
}

class ShardedParticleSystem {
    mp_pf* h_ = nullptr;
    UnfoldModel model_;
    uint64_t n_;
    int32_t world_, rank_;
    void* comm_ = nullptr;
    bool own_comm_ = false;
    bool force_;

public:
    ShardedParticleSystem(UnfoldModel model, uint64_t particles_per_rank, uint64_t seed, int32_t world, int32_t rank, void* comm, const void* id128 = nullptr,
                          int device = 0, bool force_collectives = false)
        : model_(std::move(model)), n_(particles_per_rank), world_(world), rank_(rank), comm_(comm), force_(force_collectives) {
        const mp_model_desc d = model_.desc();
        const mp_shard sh{particles_per_rank * (uint64_t)world, particles_per_rank * (uint64_t)rank};
        check(mp_pf_create(&d, particles_per_rank, seed, &sh, 0, device, nullptr, &h_));
        if (!comm_ && (world > 1 || force_collectives)) {
            unsigned char id[128];
            if (!id128 && world > 1) {
                // (every rank would make an id of its own and wait in ncclCommInitRank for peers that never come)
                mp_pf_destroy(h_);
                throw std::invalid_argument("ShardedParticleSystem: a world of several ranks needs the host's communicator or the id128 rank 0 made (mp_rccl_unique_id)");
            }
            if (!id128) { check(mp_rccl_unique_id(id)); id128 = id; }   // (a world of one: its own id)
            check(mp_rccl_comm_create(world, rank, id128, device, &comm_));
            own_comm_ = true;
        }
    }
    ShardedParticleSystem(const ShardedParticleSystem&) = delete;
    ~ShardedParticleSystem() {
        mp_pf_destroy(h_);
        if (own_comm_) mp_rccl_comm_destroy(comm_);
    }
    void init_step(const std::vector<double>& args, const std::vector<double>& constraints) {
        check(mp_pf_init_step(h_, args.empty() ? nullptr : args.data(), constraints.data(), (int32_t)(constraints.size() / model_.dim_obs)));
    }
    ShardedParticleSystem& step(const std::vector<double>& constraints) {
        check(mp_pf_step(h_, constraints.data(), (int32_t)(constraints.size() / model_.dim_obs)));
        return *this;
    }
    // log total weight of the whole job
    double resample(int32_t scheme = MP_RESAMPLE_MULTINOMIAL) {
        double v;
        check(mp_pf_shard_resample_rccl(h_, comm_, world_, rank_, scheme, force_ ? 1 : 0, &v));
        return v;
    }
    void resample_async(int32_t scheme = MP_RESAMPLE_MULTINOMIAL) { check(mp_pf_shard_resample_rccl(h_, comm_, world_, rank_, scheme, force_ ? 1 : 0, nullptr)); }
    double log_marginal_likelihood_estimate() {
        mp_transport t{};
        double lml, ess;
        if (comm_) check(mp_transport_rccl(comm_, &t));
        check(mp_pf_shard_query_native(h_, comm_ ? &t : nullptr, world_, force_ ? 1 : 0, &lml, &ess));
        return lml;
    }
    std::vector<double> states() {   // this rank's slots
        std::vector<double> x(n_ * (size_t)model_.dim_state);
        check(mp_pf_read_state(h_, x.data()));
        return x;
    }
    mp_pf* raw() { return h_; }
};

}  // namespace modppl
