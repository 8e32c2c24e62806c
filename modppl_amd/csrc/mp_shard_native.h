// mp_shard_native.h — the sharded resample as ONE library call (included by mp_pf.hip inside its extern "C" block).
//
// ParticleSystem::resample over G devices (particle_filter.rs:103-116): the owner-keeps exchange of DESIGN.md section 8.1 —
// all-gather of the packed tiles, count, expand, all-to-all of the surplus rows, commit — with the collectives issued by the
// library itself on the filter's stream, through a transport the caller names once:
//   * RCCL (mp_transport_rccl): ncclAllGather + grouped ncclSend / ncclRecv, resolved from librccl.so.1 at first use (dlopen by
//     SONAME: inside a PyTorch process that is the copy torch has already mapped; the single-GPU path never loads it);
//   * any other {all_gather, all_to_all} pair of C function pointers (tests: host-staged gloo through ctypes callbacks).
// A C++, Rust or Python host runs a sharded filter with this one entry point; modppl_amd/distributed.py keeps no protocol.
#pragma once
// (mp_pf.hip includes <dlfcn.h> and <rccl/rccl.h> — types and prototypes only: nothing links against RCCL — at file scope)

struct mp_shard_native_state {
    int world = 0;
    bool fixed = true, cap_forced = false;
    uint64_t cap = 0, fixed_max_bytes = 4ull << 20;
    uint64_t* tiles = nullptr;       // [3][nt] packed level-0 tiles of this shard
    uint64_t* tiles_all = nullptr;   // [world][3][nt]
    double* send = nullptr;     // equal splits: [world][cap][d + 1]
    double* rows[2] = {nullptr, nullptr};   // [world * cap + n][d + 1], used alternately (the previous one is still read by the step in flight)
    int flip = 0;
    double* xsend[2] = {nullptr, nullptr};  // exact sizes: grow-only, two sets used alternately
    double* xrows[2] = {nullptr, nullptr};
    uint64_t xsend_rows[2] = {0, 0}, xrows_rows[2] = {0, 0};
    int xflip = 0;
    uint64_t fallbacks = 0, last_exchange_rows = 0;
    uint64_t last_counts[SH_MAX_WORLD] = {0};
    bool have_counts = false;
};

static void shard_native_free(mp_shard_native_state* s) {
    if (!s) return;
    (void)hipFree(s->tiles); (void)hipFree(s->tiles_all); (void)hipFree(s->send);
    for (int k = 0; k < 2; ++k) { (void)hipFree(s->rows[k]); (void)hipFree(s->xsend[k]); (void)hipFree(s->xrows[k]); }
    delete s;
}

static int32_t shard_native_alloc_fixed(mp_pf* h, mp_shard_native_state* s, uint64_t cap) {
    const int d = h->ops->dim_state;
    (void)hipFree(s->send); (void)hipFree(s->rows[0]); (void)hipFree(s->rows[1]);
    s->send = nullptr; s->rows[0] = s->rows[1] = nullptr;
    s->cap = cap;
    const size_t send_b = sizeof(double) * (size_t)s->world * cap * (d + 1), rows_b = sizeof(double) * ((size_t)s->world * cap + h->n) * (d + 1);
    HIPCK(hipMalloc(&s->send, send_b));
    HIPCK(hipMemsetAsync(s->send, 0, send_b, h->stream));
    for (int k = 0; k < 2; ++k) {
        HIPCK(hipMalloc(&s->rows[k], rows_b));
        HIPCK(hipMemsetAsync(s->rows[k], 0, rows_b, h->stream));
    }
    s->flip = 0;
    return MP_OK;
}

static int32_t shard_native_init(mp_pf* h, int world) {
    auto* s = new mp_shard_native_state();
    h->native = s;
    s->world = world;
    const int d = h->ops->dim_state;
    HIPCK(hipMalloc(&s->tiles, sizeof(uint64_t) * 3 * h->nt));
    HIPCK(hipMalloc(&s->tiles_all, sizeof(uint64_t) * 3 * h->nt * (size_t)world));
    int32_t rc = mp_pf_shard_bind_tiles(h, s->tiles);   // the filter keeps its tiles where the all-gather reads them
    if (rc != MP_OK) return rc;
    // surplus rows per pair of ranks in the equal-split exchange: the surplus of a rank is the spread of a Binomial(N, ~1/world)
    // count plus the imbalance of the shard masses, both O(sqrt n): sd ~ sqrt(2 n) = 1400 rows at 2^20 particles per rank, so
    // 8192 rows per pair is ~6 sigma; a pair that needs more falls back once (exact sizes) and the capacity doubles.
    uint64_t cap = std::max<uint64_t>(4096, h->n / 128);
    if (const char* e = mp_diag_env("MP_SHARD_OWNED_CAP")) { cap = strtoull(e, nullptr, 10); s->cap_forced = true; }   // tests: force the overflow path
    cap = std::min<uint64_t>(cap, h->n);
    if (const char* e = mp_diag_env("MP_SHARD_OWNED_FIXED_MAX_BYTES")) s->fixed_max_bytes = strtoull(e, nullptr, 10);
    if (const char* e = mp_diag_env("MP_SHARD_FIXED")) s->fixed = e[0] != '0';
    // The equal-split all-to-all moves `cap` rows to every peer whatever the surplus is: fine for 16-byte rows, not for wide
    // states whose shard masses differ by percents; beyond this many padded bytes per rank the exchange uses exact sizes
    // (one host round trip per resample, a few percent of such a step).
    if (s->fixed && !s->cap_forced && (uint64_t)world * cap * (d + 1) * 8 > s->fixed_max_bytes) s->fixed = false;
    if (s->fixed) {
        rc = shard_native_alloc_fixed(h, s, cap);
        if (rc != MP_OK) return rc;
    } else {
        s->cap = cap;
    }
    return MP_OK;
}

static int32_t shard_native_grow(double** buf, uint64_t* have_rows, uint64_t want_rows, int width) {
    if (*buf && *have_rows >= want_rows) return MP_OK;
    (void)hipFree(*buf);
    *buf = nullptr;
    const uint64_t rows = want_rows + want_rows / 4 + 64;
    HIPCK(hipMalloc(buf, sizeof(double) * rows * (size_t)width));
    *have_rows = rows;
    return MP_OK;
}

// amount[r][s]: rows rank r sends to rank s when unit u of the job's surplus (donors in rank order) fills unit u of the job's
// deficit (receivers in rank order) — the plan every rank derives from the same offspring counts
static void shard_owned_plan(const uint64_t* counts, int world, uint64_t n, std::vector<uint64_t>& amount) {
    std::vector<uint64_t> S(world), D(world), PS(world), PD(world);
    uint64_t ps = 0, pd = 0;
    for (int r = 0; r < world; ++r) {
        S[r] = counts[r] > n ? counts[r] - n : 0;
        D[r] = counts[r] < n ? n - counts[r] : 0;
        PS[r] = ps; PD[r] = pd;
        ps += S[r]; pd += D[r];
    }
    amount.assign((size_t)world * world, 0);
    for (int r = 0; r < world; ++r)
        for (int q = 0; q < world; ++q) {
            const uint64_t lo = std::max(PS[r], PD[q]), hi = std::min(PS[r] + S[r], PD[q] + D[q]);
            amount[(size_t)r * world + q] = hi > lo ? hi - lo : 0;
        }
}

int32_t mp_pf_shard_resample(mp_pf* h, const mp_transport* t, int32_t world, int32_t rank, int32_t scheme, int32_t force_collectives,
                             double* log_total_weight) {
    if (!h) return mp_fail(MP_ERR_INVALID_ARG, "null handle");
    if (world < 1 || world > SH_MAX_WORLD || rank < 0 || rank >= world) return mp_fail(MP_ERR_INVALID_ARG, "1 <= world <= 64, 0 <= rank < world");
    const bool solo = world == 1 && !force_collectives;
    if (!solo && (!t || !t->all_gather || !t->all_to_all)) return mp_fail(MP_ERR_INVALID_ARG, "a world of more than one needs a transport");
    HIPCK(hipSetDevice(h->device));
    if (!h->native || h->native->world != world) {
        shard_native_free(h->native);
        h->native = nullptr;
        int32_t rci = shard_native_init(h, world);
        if (rci != MP_OK) return rci;
    }
    mp_shard_native_state* s = h->native;
    const int d = h->ops->dim_state;
    const size_t row_b = sizeof(double) * (size_t)(d + 1);
    int32_t rc = mp_pf_shard_tiles_packed(h, s->tiles);
    if (rc != MP_OK) return rc;
    const uint64_t* tiles_all = s->tiles;
    if (!solo) {
        rc = t->all_gather(t->ctx, s->tiles, s->tiles_all, sizeof(uint64_t) * 3 * (uint64_t)h->nt, (void*)h->stream);
        if (rc != MP_OK) return mp_fail(rc, "sharded resample: all-gather of the tiles failed");
        tiles_all = s->tiles_all;
    }
    std::vector<uint64_t> so(world), sb(world), ro(world), rb(world);
    uint64_t counts[SH_MAX_WORLD];
    bool need_exact = !s->fixed;
    if (s->fixed) {
        // 2 collectives, 3 phases, one host wait (for the plan's verdict word, while the rows are written and travel)
        double* rows = s->rows[s->flip];
        s->flip ^= 1;
        rc = mp_pf_shard_owned_count_expand(h, scheme, tiles_all, world, rank, s->cap, s->send, rows, (uint64_t)world * s->cap);
        if (rc != MP_OK) return rc;
        if (!solo) {
            for (int q = 0; q < world; ++q) { so[q] = ro[q] = (uint64_t)q * s->cap * row_b; sb[q] = rb[q] = s->cap * row_b; }
            rc = t->all_to_all(t->ctx, s->send, so.data(), sb.data(), rows, ro.data(), rb.data(), world, (void*)h->stream);
            if (rc != MP_OK) return mp_fail(rc, "sharded resample: all-to-all of the surplus rows failed");
        }
        rc = mp_pf_shard_owned_commit(h, rows, log_total_weight, log_total_weight ? counts : nullptr);   // (a synchronous resample waits for the stream anyway)
        if (rc == MP_OK) {
            // (an asynchronous resample never reads the counts back: the statistics then say "none" rather than repeat an
            //  earlier resample's)
            s->have_counts = log_total_weight != nullptr;
            if (log_total_weight) {
                s->last_exchange_rows = 0;
                for (int r = 0; r < world; ++r) { s->last_counts[r] = counts[r]; s->last_exchange_rows += counts[r] > h->n ? counts[r] - h->n : 0; }
            }
            return MP_OK;
        }
        if (rc != MP_ERR_CAPACITY) return rc;
        s->fallbacks += 1;   // some pair of ranks exchanges more than cap rows: exact sizes this time (every rank reaches this verdict)
        rc = mp_pf_shard_owned_commit(h, rows, nullptr, counts);   // same verdict, now with the counts
        if (rc != MP_ERR_CAPACITY) return rc == MP_OK ? mp_fail(MP_ERR_STATE, "owner-keeps exchange: the capacity verdict changed between two reads") : rc;
        need_exact = true;
    } else {
        rc = mp_pf_shard_owned_count(h, scheme, tiles_all, world, rank, 0, counts);
        if (rc != MP_OK) return rc;
    }
    (void)need_exact;
    s->have_counts = true;
    s->last_exchange_rows = 0;
    for (int r = 0; r < world; ++r) { s->last_counts[r] = counts[r]; s->last_exchange_rows += counts[r] > h->n ? counts[r] - h->n : 0; }
    std::vector<uint64_t> amount;
    shard_owned_plan(counts, world, h->n, amount);
    uint64_t n_send = 0, n_recv = 0, total = 0;
    for (int q = 0; q < world; ++q) {
        so[q] = n_send * row_b; sb[q] = amount[(size_t)rank * world + q] * row_b; n_send += amount[(size_t)rank * world + q];
        ro[q] = n_recv * row_b; rb[q] = amount[(size_t)q * world + rank] * row_b; n_recv += amount[(size_t)q * world + rank];
    }
    for (uint64_t a : amount) total += a;
    // grow-only buffers, two sets used alternately; with states wider than one double the library copies no kept offspring,
    // so only the received rows need room
    const uint64_t keep_rows = d == 1 ? h->n : 0;
    const int k = s->xflip;
    s->xflip ^= 1;
    rc = shard_native_grow(&s->xsend[k], &s->xsend_rows[k], std::max<uint64_t>(n_send, 1), d + 1);
    if (rc != MP_OK) return rc;
    rc = shard_native_grow(&s->xrows[k], &s->xrows_rows[k], std::max<uint64_t>(n_recv + keep_rows, 1), d + 1);
    if (rc != MP_OK) return rc;
    rc = mp_pf_shard_owned_expand(h, world, rank, 0, s->xsend[k], s->xrows[k], n_recv);
    if (rc != MP_OK) return rc;
    if (!solo && total > 0) {   // (nobody has a surplus: every rank sees the same counts, so every rank skips the collective)
        rc = t->all_to_all(t->ctx, s->xsend[k], so.data(), sb.data(), s->xrows[k], ro.data(), rb.data(), world, (void*)h->stream);
        if (rc != MP_OK) return mp_fail(rc, "sharded resample: exact-size all-to-all failed");
    } else if (n_recv) {
        HIPCK(hipMemcpyAsync(s->xrows[k], s->xsend[k], n_recv * row_b, hipMemcpyDeviceToDevice, h->stream));   // a world of one with forced exact sizes
    }
    rc = mp_pf_shard_owned_commit(h, s->xrows[k], log_total_weight, nullptr);
    if (rc != MP_OK) return rc;
    if (s->fixed && !s->cap_forced && s->cap < h->n) {
        // the next resample can use equal splits again, with room for what this one needed (every rank grows alike)
        uint64_t worst = 0;
        for (uint64_t a : amount) worst = std::max(worst, a);
        const uint64_t grown = std::min<uint64_t>(h->n, std::max<uint64_t>(2 * s->cap, 2 * worst));
        if ((uint64_t)world * grown * (d + 1) * 8 > s->fixed_max_bytes) {
            s->fixed = false;
        } else {
            HIPCK(stream_wait(h->stream));   // nothing refers to the old equal-split buffers any more
            rc = shard_native_alloc_fixed(h, s, grown);
            if (rc != MP_OK) return rc;
        }
    }
    return MP_OK;
}

int32_t mp_pf_shard_resample_stats(mp_pf* h, uint64_t* fallbacks, uint64_t* exchange_rows, uint64_t* counts_out, uint64_t* capacity) {
    if (!h || !h->native) return mp_fail(MP_ERR_STATE, "no sharded resample has run on this handle");
    if (fallbacks) *fallbacks = h->native->fallbacks;
    if (exchange_rows) *exchange_rows = h->native->have_counts ? h->native->last_exchange_rows : ~0ull;
    if (counts_out && h->native->have_counts)
        for (int r = 0; r < h->native->world; ++r) counts_out[r] = h->native->last_counts[r];
    if (capacity) *capacity = h->native->fixed ? h->native->cap : 0;
    return MP_OK;
}

// job-wide log-ML / fresh ESS of a sharded filter: level 0 of this shard, all-gather, level 1 (particle_filter.rs:119-121, :98-100)
int32_t mp_pf_shard_query_native(mp_pf* h, const mp_transport* t, int32_t world, int32_t force_collectives, double* log_ml, double* ess) {
    if (!h) return mp_fail(MP_ERR_INVALID_ARG, "null handle");
    const bool solo = world == 1 && !force_collectives;
    if (!solo && (!t || !t->all_gather)) return mp_fail(MP_ERR_INVALID_ARG, "a world of more than one needs a transport");
    HIPCK(hipSetDevice(h->device));
    if (!h->native || h->native->world != world) {
        shard_native_free(h->native);
        h->native = nullptr;
        int32_t rci = shard_native_init(h, world);
        if (rci != MP_OK) return rci;
    }
    mp_shard_native_state* s = h->native;
    int32_t rc = mp_pf_shard_tiles_packed(h, s->tiles);
    if (rc != MP_OK) return rc;
    const uint64_t* tiles_all = s->tiles;
    if (!solo) {
        rc = t->all_gather(t->ctx, s->tiles, s->tiles_all, sizeof(uint64_t) * 3 * (uint64_t)h->nt, (void*)h->stream);
        if (rc != MP_OK) return mp_fail(rc, "sharded query: all-gather of the tiles failed");
        tiles_all = s->tiles_all;
    }
    return mp_pf_shard_query_packed(h, tiles_all, world, log_ml, ess);
}

// ---- the RCCL transport ------------------------------------------------------------------------------------------
struct mp_rccl_api {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
static bool rccl_resolve(mp_rccl_api& api) {
    // by SONAME first: a process that already has an RCCL mapped (PyTorch's bundled one) gets THAT copy, not a second one
    for (const char* name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
        api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (api.lib) break;
    }
    if (!api.lib) return false;
#define MP_RCCL_SYM(F) api.F = reinterpret_cast<decltype(api.F)>(dlsym(api.lib, "nccl" #F)); if (!api.F) { api.lib = nullptr; return false; }
    MP_RCCL_SYM(GetUniqueId) MP_RCCL_SYM(CommInitRank) MP_RCCL_SYM(CommDestroy) MP_RCCL_SYM(AllGather) MP_RCCL_SYM(GroupStart)
    MP_RCCL_SYM(GroupEnd) MP_RCCL_SYM(Send) MP_RCCL_SYM(Recv) MP_RCCL_SYM(GetErrorString)
#undef MP_RCCL_SYM
    return true;
}
static mp_rccl_api* rccl_api() {
    // resolved once, whichever host thread asks first (initialisation of a function-local static is synchronised)
    static mp_rccl_api api;
    static const bool ok = rccl_resolve(api);
    return ok ? &api : nullptr;
}
#define RCCLCK(call)                                                                                                         \
    do {                                                                                                                     \
        ncclResult_t r_ = (call);                                                                                            \
        if (r_ != ncclSuccess) return mp_fail(MP_ERR_HIP, std::string(#call " failed: ") + rccl_api()->GetErrorString(r_));  \
    } while (0)

static int32_t rccl_all_gather(void* ctx, const void* d_send, void* d_recv, uint64_t bytes_per_rank, void* stream) {
    mp_rccl_api* a = rccl_api();
    RCCLCK(a->AllGather(d_send, d_recv, bytes_per_rank, ncclUint8, (ncclComm_t)ctx, (hipStream_t)stream));
    return MP_OK;
}
// rows to / from every peer as ONE group of point-to-point operations (xGMI is point-to-point: no staging through a root)
static int32_t rccl_all_to_all(void* ctx, const void* d_send, const uint64_t* send_off, const uint64_t* send_bytes, void* d_recv,
                               const uint64_t* recv_off, const uint64_t* recv_bytes, int32_t world, void* stream) {
    mp_rccl_api* a = rccl_api();
    RCCLCK(a->GroupStart());
    // an error inside the group still closes it (an open group would swallow every later RCCL call of this thread); the FIRST
    // error is the one reported
    ncclResult_t first = ncclSuccess;
    const char* where = "";
    for (int q = 0; q < world && first == ncclSuccess; ++q) {
        if (send_bytes[q]) {
            first = a->Send(static_cast<const char*>(d_send) + send_off[q], send_bytes[q], ncclUint8, q, (ncclComm_t)ctx, (hipStream_t)stream);
            if (first != ncclSuccess) { where = "ncclSend"; break; }
        }
        if (recv_bytes[q]) {
            first = a->Recv(static_cast<char*>(d_recv) + recv_off[q], recv_bytes[q], ncclUint8, q, (ncclComm_t)ctx, (hipStream_t)stream);
            if (first != ncclSuccess) where = "ncclRecv";
        }
    }
    const ncclResult_t end = a->GroupEnd();
    if (first != ncclSuccess) return mp_fail(MP_ERR_HIP, std::string(where) + " failed: " + a->GetErrorString(first));
    if (end != ncclSuccess) return mp_fail(MP_ERR_HIP, std::string("ncclGroupEnd failed: ") + a->GetErrorString(end));
    return MP_OK;
}
// local, non-collective: can this process resolve an RCCL at all?  (hosts agree on the answer BEFORE any rendezvous)
int32_t mp_rccl_available(void) { return rccl_api() ? 1 : 0; }
int32_t mp_rccl_unique_id(void* out128) {
    mp_rccl_api* a = rccl_api();
    if (!a) return mp_fail(MP_ERR_UNSUPPORTED, "librccl.so.1 could not be loaded");
    if (!out128) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId");
    RCCLCK(a->GetUniqueId(static_cast<ncclUniqueId*>(out128)));
    return MP_OK;
}
// one communicator per process / device (rank 0 makes the id with mp_rccl_unique_id and hands it to the others by whatever
// channel the host has: MPI, a file, torch.distributed's store)
int32_t mp_rccl_comm_create(int32_t world, int32_t rank, const void* id128, int32_t device, void** comm_out) {
    mp_rccl_api* a = rccl_api();
    if (!a) return mp_fail(MP_ERR_UNSUPPORTED, "librccl.so.1 could not be loaded");
    if (!id128 || !comm_out) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    HIPCK(hipSetDevice(device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t c = nullptr;
    RCCLCK(a->CommInitRank(&c, world, id, rank));
    *comm_out = c;
    return MP_OK;
}
int32_t mp_rccl_comm_destroy(void* comm) {
    mp_rccl_api* a = rccl_api();
    if (!a || !comm) return MP_OK;
    RCCLCK(a->CommDestroy((ncclComm_t)comm));
    return MP_OK;
}
int32_t mp_transport_rccl(void* nccl_comm, mp_transport* out) {
    if (!nccl_comm || !out) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    if (!rccl_api()) return mp_fail(MP_ERR_UNSUPPORTED, "librccl.so.1 could not be loaded");
    out->ctx = nccl_comm;
    out->all_gather = rccl_all_gather;
    out->all_to_all = rccl_all_to_all;
    return MP_OK;
}
int32_t mp_pf_shard_resample_rccl(mp_pf* h, void* nccl_comm, int32_t world, int32_t rank, int32_t scheme, int32_t force_collectives,
                                  double* log_total_weight) {
    mp_transport t;
    if (world > 1 || force_collectives) {
        int32_t rc = mp_transport_rccl(nccl_comm, &t);
        if (rc != MP_OK) return rc;
    }
    return mp_pf_shard_resample(h, (world > 1 || force_collectives) ? &t : nullptr, world, rank, scheme, force_collectives, log_total_weight);
}
// plain device <-> host copies on the filter's stream, for transports that stage through the host (tests)
int32_t mp_pf_stream_copy(mp_pf* h, void* dst, const void* src, uint64_t bytes, int32_t to_host) {
    if (!h || !dst || !src) return mp_fail(MP_ERR_INVALID_ARG, "null argument");
    HIPCK(hipSetDevice(h->device));
    HIPCK(hipMemcpyAsync(dst, src, bytes, to_host ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice, h->stream));
    HIPCK(stream_wait(h->stream));
    return MP_OK;
}
