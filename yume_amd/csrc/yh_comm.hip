// yh_comm.hip — multi-GPU: RCCL (opened with dlopen on the first yh_comm_init) and the gathered draw lists.
#define YH_K_GATHER
#include "yh_internal.hpp"

// -------------------------------------------------------------------------------------------- RCCL
// RCCL is loaded on demand so that single-GPU users never depend on it.  Types are kept opaque; the
// enum values below are RCCL's (rccl.h: ncclUint32 = 3, ncclSuccess = 0).
typedef struct {
    char internal[YH_UNIQUE_ID_BYTES];
} yh_nccl_id;
typedef int (*fn_GetUniqueId)(yh_nccl_id*);
typedef int (*fn_CommInitRank)(void**, int, yh_nccl_id, int);
typedef int (*fn_CommDestroy)(void*);
typedef int (*fn_AllGather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*fn_Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_Group)(void);
typedef const char* (*fn_GetErrorString)(int);

static struct {
    void* lib;
    fn_GetUniqueId GetUniqueId;
    fn_CommInitRank CommInitRank;
    fn_CommDestroy CommDestroy;
    fn_AllGather AllGather;
    fn_Broadcast Broadcast;
    fn_Group GroupStart, GroupEnd;
    fn_GetErrorString GetErrorString;
    int version;     // ncclGetVersion: major * 10000 + minor * 100 + patch
    char path[512];  // the file the loader resolved "librccl.so.1" to (dladdr of one of its symbols)
} g_rccl;

static int rccl_load(yh_ctx* ctx) {
    if (g_rccl.lib) return YH_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    // By SONAME: a process that already holds an RCCL (torch bundles one with the same SONAME) gets THAT copy again, so
    // there is one RCCL in the process.  RTLD_LOCAL: when this library is first and torch loads its own copy later, the
    // two must not see each other's symbols (with RTLD_GLOBAL such a process died in a double free at exit).
    void* lib = nullptr;
    for (const char* n : names)
        if ((lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!lib) return fail(ctx, YH_ERR_RCCL, "RCCL not found: %s", dlerror());
#define SYM(field, name)                                                              \
    g_rccl.field = (decltype(g_rccl.field))dlsym(lib, name);                          \
    if (!g_rccl.field) return fail(ctx, YH_ERR_RCCL, "RCCL symbol %s missing", name);
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllGather, "ncclAllGather")
    SYM(Broadcast, "ncclBroadcast")
    SYM(GroupStart, "ncclGroupStart")
    SYM(GroupEnd, "ncclGroupEnd")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    // Which RCCL this is.  A process that has torch in it already holds torch's bundled librccl (same SONAME): the loader
    // then hands that one out again and the process runs ONE RCCL; a second copy would show here as another path.
    Dl_info where;
    memset(&where, 0, sizeof where);
    if (dladdr((void*)g_rccl.GetUniqueId, &where) && where.dli_fname) snprintf(g_rccl.path, sizeof g_rccl.path, "%s", where.dli_fname);
    typedef int (*fn_GetVersion)(int*);
    if (fn_GetVersion get_version = (fn_GetVersion)dlsym(lib, "ncclGetVersion")) (void)get_version(&g_rccl.version);
    g_rccl.lib = lib;
    return YH_OK;
}

extern "C" int yh_comm_describe(char* buf, uint64_t cap) {
    if (!buf || !cap) return fail(nullptr, YH_ERR_BAD_ARG, "yh_comm_describe: no buffer");
    if (int rc = rccl_load(nullptr)) return rc;
    snprintf(buf, (size_t)cap, "rccl %d.%d.%d from %s", g_rccl.version / 10000, g_rccl.version / 100 % 100, g_rccl.version % 100,
             g_rccl.path[0] ? g_rccl.path : "(path unknown)");
    return YH_OK;
}

#define RCCL_TRY(ctx, call)                                                                                   \
    do {                                                                                                      \
        int r_ = (call);                                                                                      \
        if (r_ != 0) return fail((ctx), YH_ERR_RCCL, "%s failed: %s", #call, g_rccl.GetErrorString(r_));      \
    } while (0)

extern "C" int yh_comm_unique_id(void* out_id) {
    if (!out_id) return fail(nullptr, YH_ERR_BAD_ARG, "yh_comm_unique_id: out_id is NULL");
    if (int rc = rccl_load(nullptr)) return rc;
    yh_nccl_id id;
    memset(&id, 0, sizeof id);
    RCCL_TRY(nullptr, g_rccl.GetUniqueId(&id));
    memcpy(out_id, &id, sizeof id);
    return YH_OK;
}

extern "C" int yh_comm_init(yh_ctx* ctx, int n_ranks, int rank, const void* unique_id) {
    if (!ctx || !unique_id) return fail(ctx, YH_ERR_BAD_ARG, "yh_comm_init: NULL argument");
    ARG_TRY(ctx, n_ranks >= 1 && rank >= 0 && rank < n_ranks, "yh_comm_init: bad rank / n_ranks");
    if (ctx->comm) return fail(ctx, YH_ERR_STATE, "yh_comm_init: communicator already initialised");
    if (int rc = rccl_load(ctx)) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    yh_nccl_id id;
    memcpy(&id, unique_id, sizeof id);
    if (getenv("YUME_HIP_COMM_LOG")) {  // one line per rank on stderr: which RCCL, which device
        char what[640];
        if (yh_comm_describe(what, sizeof what) == YH_OK)
            fprintf(stderr, "yume_hip: rank %d of %d on device %d: %s\n", rank, n_ranks, ctx->device, what);
    }
    RCCL_TRY(ctx, g_rccl.CommInitRank(&ctx->comm, n_ranks, id, rank));
    ctx->n_ranks = n_ranks;
    ctx->rank = rank;
    return YH_OK;
}

extern "C" void yh_comm_destroy(yh_ctx* ctx) {
    if (ctx && ctx->comm && g_rccl.CommDestroy) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->stream_c) (void)hipStreamSynchronize(ctx->stream_c);
        g_rccl.CommDestroy(ctx->comm);
        ctx->comm = nullptr;
        ctx->n_ranks = 1;
        ctx->rank = 0;
    }
}

static int ensure_all_counts(yh_scene* s, int R) {
    yh_ctx* ctx = s->ctx;
    if (s->d_all_counts && s->counts_cap >= R) return YH_OK;
    if (int rc = sync_scene(s)) return rc;
    if (s->d_all_counts) HIP_TRY(ctx, hipFree(s->d_all_counts));
    if (s->h_all_counts) HIP_TRY(ctx, hipHostFree(s->h_all_counts));
    s->d_all_counts = nullptr;
    s->h_all_counts = nullptr;
    s->counts_cap = 0;
    HIP_TRY(ctx, hipMalloc((void**)&s->d_all_counts, (size_t)R * YH_MAX_FRUSTA * 4));
    HIP_TRY(ctx, hipHostMalloc((void**)&s->h_all_counts, (size_t)R * YH_MAX_FRUSTA * 4, hipHostMallocMapped));
    HIP_TRY(ctx, hipHostGetDevicePointer((void**)&s->h_all_counts_dev, s->h_all_counts, 0));
    s->counts_cap = R;
    return YH_OK;
}

extern "C" int yh_allgather_draw_lists(yh_scene* s, uint32_t* all_counts) {
    if (!s) return fail(nullptr, YH_ERR_BAD_ARG, "yh_allgather_draw_lists: scene is NULL");
    yh_ctx* ctx = s->ctx;
    if (!ctx->comm) return fail(ctx, YH_ERR_STATE, "yh_allgather_draw_lists: call yh_comm_init first");
    if (s->frames == 0) return fail(ctx, YH_ERR_STATE, "yh_allgather_draw_lists: no frame has run");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int R = ctx->n_ranks;
    const uint32_t F = s->last_n_frusta;
    if (int rc = ensure_all_counts(s, R)) return rc;
    hipStream_t st = ctx->stream;
    if (int rc = sync_scene(s)) return rc;  // a no-sync gather may still be in flight on the other stream
    s->gathered_async = false;
    s->counts_ranks = R;
    // (1) per-rank visible counts: 8 x u32 per rank
    RCCL_TRY(ctx, g_rccl.AllGather(s->counts, s->d_all_counts, YH_MAX_FRUSTA, /*ncclUint32*/ 3, ctx->comm, st));
    HIP_TRY(ctx, hipMemcpyAsync(s->h_all_counts, s->d_all_counts, (size_t)R * YH_MAX_FRUSTA * 4, hipMemcpyDeviceToHost, st));
    if (int rc = sync_scene(s)) return rc;
    // (2) exact-size lists at prefix-sum offsets
    for (uint32_t f = 0; f < F; ++f) {
        uint64_t total = 0;
        for (int r = 0; r < R; ++r) total += s->h_all_counts[r * YH_MAX_FRUSTA + f];
        if (total > s->gathered_cap[f]) {
            if (s->gathered[f]) HIP_TRY(ctx, hipFree(s->gathered[f]));
            s->gathered[f] = nullptr;
            s->gathered_cap[f] = 0;
            const uint64_t cap = total + total / 2 + 1024;
            HIP_TRY(ctx, hipMalloc((void**)&s->gathered[f], cap * 4));
            s->gathered_cap[f] = cap;
        }
        s->gathered_len[f] = total;
    }
    for (int r = 0; r < R; ++r)
        for (uint32_t f = 0; f < F; ++f)
            if (s->h_all_counts[r * YH_MAX_FRUSTA + f] > s->draw_cap)
                return fail(ctx, YH_ERR_CAPACITY, "yh_allgather_draw_lists: rank %d frustum %u overflowed its draw list", r, f);
    RCCL_TRY(ctx, g_rccl.GroupStart());
    for (uint32_t f = 0; f < F; ++f) {
        uint64_t off = 0;
        for (int r = 0; r < R; ++r) {
            const uint64_t c = s->h_all_counts[r * YH_MAX_FRUSTA + f];
            if (c) {
                int rc = g_rccl.Broadcast(s->draw[f], s->gathered[f] + off, c, /*ncclUint32*/ 3, r, ctx->comm, st);
                if (rc != 0) {
                    g_rccl.GroupEnd();
                    return fail(ctx, YH_ERR_RCCL, "ncclBroadcast failed: %s", g_rccl.GetErrorString(rc));
                }
            }
            off += c;
        }
    }
    RCCL_TRY(ctx, g_rccl.GroupEnd());
    if (int rc = sync_scene(s)) return rc;
    if (all_counts)
        for (int r = 0; r < R; ++r)
            for (uint32_t f = 0; f < F; ++f) all_counts[r * F + f] = s->h_all_counts[r * YH_MAX_FRUSTA + f];
    return YH_OK;
}

// The same result without a host round trip, for frames that are enqueued back to back.  xGMI is point to point
// (one ~50 GB/s link per peer), so the bytes on the wire decide how this scales: the VISIBILITY WORDS travel (n/8
// bytes per frustum, a fixed size: one ncclAllGather, no counts needed first) and every rank expands all R masks into
// the gathered lists itself (k_gather_count / k_gather_expand) — 32x fewer bytes than lists padded to capacity, 20x
// fewer than exact lists at 64 % visible.  Everything runs on the context's collective stream behind this frame's
// launches and beside the next frames'; frames alternate between two sets of words, so only the frame after next
// waits (for the all-gather's read of the set it rewrites).  One-time set-up per scene (host sync): the ranks
// exchange their word counts and id bases.
static int gather_setup(yh_scene* s) {
    yh_ctx* ctx = s->ctx;
    const int R = ctx->n_ranks;
    if (s->g_ready && s->g_ranks == R) return YH_OK;
    if (s->g_ready) return fail(ctx, YH_ERR_STATE, "yh_allgather_draw_lists_async: the communicator changed size under this scene");
    if (int rc = sync_scene(s)) return rc;
    const uint32_t F = s->desc.n_frusta;
    // a set-up that failed half way (a rank with other frusta, out of memory) starts over
    for (void** p : {(void**)&s->d_peer, (void**)&s->gmask, (void**)&s->send_pad, (void**)&s->gtile}) {
        if (*p) HIP_TRY(ctx, hipFree(*p));
        *p = nullptr;
    }
    if (s->ev_frame) HIP_TRY(ctx, hipEventDestroy(s->ev_frame));
    s->ev_frame = nullptr;
    for (hipEvent_t& e : s->ev_mask) {
        if (e) HIP_TRY(ctx, hipEventDestroy(e));
        e = nullptr;
    }
    HIP_TRY(ctx, hipMalloc((void**)&s->d_peer, ((size_t)R + 1) * 16));
    const uint32_t mine[4] = {s->n_chunks, s->desc.id_base, F, 0u};
    HIP_TRY(ctx, hipMemcpy(s->d_peer + (size_t)R * 4, mine, 16, hipMemcpyHostToDevice));
    RCCL_TRY(ctx, g_rccl.AllGather(s->d_peer + (size_t)R * 4, s->d_peer, 4, /*ncclUint32*/ 3, ctx->comm, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    s->h_peer.assign((size_t)R * 4, 0u);
    HIP_TRY(ctx, hipMemcpy(s->h_peer.data(), s->d_peer, (size_t)R * 16, hipMemcpyDeviceToHost));
    uint32_t words = 0;
    for (int r = 0; r < R; ++r) {
        if (s->h_peer[r * 4 + 2] != F)
            return fail(ctx, YH_ERR_BAD_ARG, "yh_allgather_draw_lists_async: rank %d has %u frusta, this rank %u", r, s->h_peer[r * 4 + 2], F);
        words = std::max(words, s->h_peer[r * 4 + 0]);
    }
    s->g_words = words;
    s->g_tile_words = words <= 32u * 4096u ? 64u : (uint32_t)YH_EXPAND_WORDS_MAX;
    s->g_tiles = (words + s->g_tile_words - 1) / s->g_tile_words;
    s->g_frusta = F;
    HIP_TRY(ctx, hipMalloc((void**)&s->gmask, (uint64_t)R * F * words * 8));
    HIP_TRY(ctx, hipMalloc((void**)&s->gtile, (uint64_t)F * R * s->g_tiles * 4));
    if (s->n_chunks != words) {
        HIP_TRY(ctx, hipMalloc((void**)&s->send_pad, (uint64_t)F * words * 8));
        HIP_TRY(ctx, hipMemsetAsync(s->send_pad, 0, (uint64_t)F * words * 8, ctx->stream));
    }
    // (fills go on the context's stream and are waited for: the null stream is not ordered with our non-blocking streams)
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (uint32_t f = 0; f < F; ++f) {
        const uint64_t cap = (uint64_t)R * words * 64;
        if (s->gathered_cap[f] < cap) {
            if (s->gathered[f]) HIP_TRY(ctx, hipFree(s->gathered[f]));
            s->gathered[f] = nullptr;
            s->gathered_cap[f] = 0;
            HIP_TRY(ctx, hipMalloc((void**)&s->gathered[f], cap * 4));
            s->gathered_cap[f] = cap;
        }
    }
    if (int rc = ensure_all_counts(s, R)) return rc;
    HIP_TRY(ctx, hipEventCreateWithFlags(&s->ev_frame, hipEventDisableTiming));
    for (hipEvent_t& e : s->ev_mask) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    s->g_ranks = R;
    s->g_ready = true;
    return YH_OK;
}

// Expansion of the received words on stream sc (shared by the gather and its single-GPU self-test).
static int gather_expand(yh_scene* s, const uint64_t* gmask, uint32_t F, hipStream_t sc) {
    yh_ctx* ctx = s->ctx;
    GatherArgs g;
    memset(&g, 0, sizeof g);
    g.gmask = gmask;
    g.peer = s->d_peer;
    g.gtile = s->gtile;
    g.cum = s->h_all_counts_dev;  // host-mapped: the totals need no copy
    for (uint32_t f = 0; f < YH_MAX_FRUSTA; ++f) g.out[f] = s->gathered[f];
    g.words = s->g_words;
    g.n_tiles = s->g_tiles;
    g.tile_words = s->g_tile_words;
    g.n_ranks = (uint32_t)s->g_ranks;
    g.n_frusta = F;
    const dim3 grid(s->g_tiles, F, (uint32_t)s->g_ranks);
    hipLaunchKernelGGL(k_gather_count, grid, dim3(YH_BLOCK), 0, sc, g);
    hipLaunchKernelGGL(k_gather_expand, grid, dim3(YH_BLOCK), 0, sc, g);
    HIP_TRY(ctx, hipGetLastError());
    s->counts_ranks = s->g_ranks;
    return YH_OK;
}

extern "C" int yh_allgather_draw_lists_async(yh_scene* s) {
    if (!s) return fail(nullptr, YH_ERR_BAD_ARG, "yh_allgather_draw_lists_async: scene is NULL");
    yh_ctx* ctx = s->ctx;
    if (!ctx->comm) return fail(ctx, YH_ERR_STATE, "yh_allgather_draw_lists_async: call yh_comm_init first");
    if (s->frames == 0 || !s->last_mask || s->last_n_frusta == 0)
        return fail(ctx, YH_ERR_STATE, "yh_allgather_draw_lists_async: no culling frame has run");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = gather_setup(s)) return rc;
    const uint32_t F = s->last_n_frusta;
    const uint32_t W = s->g_words;
    hipStream_t sc = ctx->stream_c;
    HIP_TRY(ctx, hipEventRecord(s->ev_frame, ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(sc, s->ev_frame, 0));
    const uint64_t* send = s->last_mask;  // [F][n_chunks], rows back to back
    if (s->send_pad) {                    // shorter shard: rows padded with zero words to the longest one's length
        HIP_TRY(ctx, hipMemcpy2DAsync(s->send_pad, (size_t)W * 8, s->last_mask, (size_t)s->n_chunks * 8, (size_t)s->n_chunks * 8, F,
                                      hipMemcpyDeviceToDevice, sc));
        send = s->send_pad;
    }
    const uint32_t p = s->last_mparity;  // the set of words the frame wrote
    RCCL_TRY(ctx, g_rccl.AllGather(send, s->gmask, (size_t)F * W, /*ncclUint64*/ 5, ctx->comm, sc));
    HIP_TRY(ctx, hipEventRecord(s->ev_mask[p], sc));
    s->mask_busy[p] = true;
    s->mask_frame[p] = s->frames;
    // expansion behind it on the same stream (a third stream for it, so that the next all-gather need not queue
    // behind this expansion, measured slower: 49-60 us per cfg2 frame against 46-48)
    if (int rc = gather_expand(s, s->gmask, F, sc)) return rc;
    s->gathered_async = true;
    s->frame_pending = true;
    return YH_OK;
}

// Testing aid (one GPU, no communicator): the gather's expansion fed with what R ranks would have all-gathered.
// Row r of the receive buffer is shard r's visibility words of its last frame, zero-padded to the longest shard's
// row length exactly as send_pad pads them on a real rank (plus `extra_words` more: a longer shard that is not here),
// peer r = that shard's word count and id base.  The expansion runs on out's buffers; results are read from `out` with
// yh_get_gathered_counts / yh_get_gathered_draw_list as after a real gather.
struct FakeRank {
    const uint64_t* words;
    uint32_t n_chunks, id_base;
};
static int gather_selftest_run(yh_scene* out, const std::vector<FakeRank>& ranks, uint32_t F, uint32_t extra_words) {
    yh_ctx* ctx = out->ctx;
    const uint32_t R = (uint32_t)ranks.size();
    uint32_t W = 0;
    for (const FakeRank& r : ranks) W = std::max(W, r.n_chunks);
    W += extra_words;
    if (int rc = ensure_all_counts(out, (int)R)) return rc;
    out->g_ranks = (int)R;
    out->g_words = W;
    out->g_tile_words = W <= 32u * 4096u ? 64u : (uint32_t)YH_EXPAND_WORDS_MAX;
    out->g_tiles = (W + out->g_tile_words - 1) / out->g_tile_words;
    uint64_t* gmask = nullptr;
    uint32_t *gtile = nullptr, *peer = nullptr;
    std::vector<uint32_t> hp((size_t)R * 4);
    for (uint32_t r = 0; r < R; ++r) {
        hp[r * 4 + 0] = ranks[r].n_chunks;
        hp[r * 4 + 1] = ranks[r].id_base;
        hp[r * 4 + 2] = F;
    }
    hipError_t e = hipMalloc((void**)&gmask, (uint64_t)R * F * W * 8);
    if (e == hipSuccess) e = hipMemsetAsync(gmask, 0, (uint64_t)R * F * W * 8, ctx->stream);
    if (e == hipSuccess) e = hipMalloc((void**)&gtile, (uint64_t)F * R * out->g_tiles * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&peer, (size_t)R * 16);
    if (e == hipSuccess) e = hipMemcpyAsync(peer, hp.data(), (size_t)R * 16, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // hp is a host temporary
    for (uint32_t r = 0; r < R && e == hipSuccess; ++r)  // the copy yh_allgather_draw_lists_async makes into send_pad
        e = hipMemcpy2DAsync(gmask + (uint64_t)r * F * W, (size_t)W * 8, ranks[r].words, (size_t)ranks[r].n_chunks * 8,
                             (size_t)ranks[r].n_chunks * 8, F, hipMemcpyDeviceToDevice, ctx->stream);
    for (uint32_t f = 0; f < F && e == hipSuccess; ++f) {
        const uint64_t cap = (uint64_t)R * W * 64;
        if (out->gathered_cap[f] < cap) {
            if (out->gathered[f]) (void)hipFree(out->gathered[f]);
            out->gathered[f] = nullptr;
            out->gathered_cap[f] = 0;
            e = hipMalloc((void**)&out->gathered[f], cap * 4);
            if (e == hipSuccess) out->gathered_cap[f] = cap;
        }
    }
    int rc = YH_OK;
    if (e == hipSuccess) {
        // lend the buffers to the scene's gather fields for one expansion
        out->gmask = gmask;
        out->gtile = gtile;
        out->d_peer = peer;
        rc = gather_expand(out, gmask, F, ctx->stream);
        if (rc == YH_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ctx, YH_ERR_HIP, "yh_gather_selftest: expansion failed");
        out->gmask = nullptr;
        out->gtile = nullptr;
        out->d_peer = nullptr;
        out->gathered_async = rc == YH_OK;
    }
    out->g_ranks = 0;
    out->g_words = out->g_tiles = 0;
    for (void* p : {(void*)gmask, (void*)gtile, (void*)peer})
        if (p) (void)hipFree(p);
    if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? YH_ERR_OOM : YH_ERR_HIP, "yh_gather_selftest: %s", hipGetErrorString(e));
    return rc;
}

// fake_ranks copies of this scene's own words, rank r pretending to own ids id_base + r * n_entities (what R identical
// shards would have sent), rows three words longer than the scene's own (the zero padding).
extern "C" int yh_gather_selftest(yh_scene* s, uint32_t fake_ranks) {
    if (!s) return fail(nullptr, YH_ERR_BAD_ARG, "yh_gather_selftest: scene is NULL");
    yh_ctx* ctx = s->ctx;
    ARG_TRY(ctx, fake_ranks >= 1 && fake_ranks <= 64, "yh_gather_selftest: 1..64 ranks");
    if (ctx->comm || s->g_ready) return fail(ctx, YH_ERR_STATE, "yh_gather_selftest: not on a scene or context that gathers for real");
    if (s->frames == 0 || s->last_n_frusta == 0 || !s->last_mask) return fail(ctx, YH_ERR_STATE, "yh_gather_selftest: no culling frame has run");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = sync_scene(s)) return rc;
    std::vector<FakeRank> ranks;
    for (uint32_t r = 0; r < fake_ranks; ++r) ranks.push_back({s->last_mask, s->n_chunks, s->desc.id_base + r * (uint32_t)s->n});
    return gather_selftest_run(s, ranks, s->last_n_frusta, 3);
}

// The same with R DIFFERENT shards: the scenes of one context (one per part of yh_partition_subtrees, each framed
// with the same frusta) stand for the ranks, in rank order; every row keeps its own length and id base, the shorter
// ones are padded.  The gathered lists land in shards[0] — what rank 0 would hold after the exchange.
extern "C" int yh_gather_selftest_shards(yh_scene* const* shards, uint32_t n_shards) {
    if (!shards || !n_shards || !shards[0]) return fail(nullptr, YH_ERR_BAD_ARG, "yh_gather_selftest_shards: NULL argument");
    yh_scene* out = shards[0];
    yh_ctx* ctx = out->ctx;
    ARG_TRY(ctx, n_shards <= 64, "yh_gather_selftest_shards: 1..64 shards");
    if (ctx->comm || out->g_ready) return fail(ctx, YH_ERR_STATE, "yh_gather_selftest_shards: not on a scene or context that gathers for real");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<FakeRank> ranks;
    for (uint32_t r = 0; r < n_shards; ++r) {
        yh_scene* s = shards[r];
        ARG_TRY(ctx, s && s->ctx == ctx, "yh_gather_selftest_shards: every shard must be a scene of the same context");
        if (s->frames == 0 || s->last_n_frusta == 0 || !s->last_mask)
            return fail(ctx, YH_ERR_STATE, "yh_gather_selftest_shards: shard %u has run no culling frame", r);
        ARG_TRY(ctx, s->last_n_frusta == out->last_n_frusta, "yh_gather_selftest_shards: the shards culled different numbers of frusta");
        if (int rc = sync_scene(s)) return rc;
        ranks.push_back({s->last_mask, s->n_chunks, s->desc.id_base});
    }
    return gather_selftest_run(out, ranks, out->last_n_frusta, 0);
}

// After a no-sync gather the host buffer holds running totals (ranks 0..r); turn them into per-rank counts once.
static int finish_async_counts(yh_scene* s) {
    if (!s->gathered_async) return YH_OK;
    if (int rc = sync_scene(s)) return rc;
    const int R = s->counts_ranks;
    for (uint32_t f = 0; f < s->last_n_frusta; ++f) {
        s->gathered_len[f] = s->h_all_counts[(R - 1) * YH_MAX_FRUSTA + f];
        for (int r = R - 1; r > 0; --r) s->h_all_counts[r * YH_MAX_FRUSTA + f] -= s->h_all_counts[(r - 1) * YH_MAX_FRUSTA + f];
    }
    s->gathered_async = false;
    return YH_OK;
}

extern "C" int yh_get_gathered_counts(yh_scene* s, uint32_t* all_counts) {
    if (!s || !all_counts) return fail(s ? s->ctx : nullptr, YH_ERR_BAD_ARG, "yh_get_gathered_counts: NULL argument");
    if (!s->h_all_counts) return fail(s->ctx, YH_ERR_STATE, "yh_get_gathered_counts: no gather has run");
    if (int rc = sync_scene(s)) return rc;
    if (int rc = finish_async_counts(s)) return rc;
    const uint32_t F = s->last_n_frusta;
    const int R = s->counts_ranks;
    for (int r = 0; r < R; ++r)
        for (uint32_t f = 0; f < F; ++f) all_counts[r * F + f] = s->h_all_counts[r * YH_MAX_FRUSTA + f];
    return YH_OK;
}

extern "C" int yh_get_gathered_draw_list(yh_scene* s, uint32_t frustum, uint32_t* out, uint64_t cap, uint64_t* written) {
    if (!s) return fail(nullptr, YH_ERR_BAD_ARG, "yh_get_gathered_draw_list: scene is NULL");
    ARG_TRY(s->ctx, frustum < s->last_n_frusta, "yh_get_gathered_draw_list: frustum index out of range");
    if (int rc = finish_async_counts(s)) return rc;  // lengths become known on the host only now
    const uint64_t c = s->gathered_len[frustum];
    if (written) *written = c;
    if (c > cap || (c && !out))
        return fail(s->ctx, YH_ERR_CAPACITY, "yh_get_gathered_draw_list: buffer of %llu entries is too small, %llu required",
                    (unsigned long long)cap, (unsigned long long)c);
    HIP_TRY(s->ctx, hipSetDevice(s->ctx->device));
    if (c) HIP_TRY(s->ctx, hipMemcpy(out, s->gathered[frustum], c * 4, hipMemcpyDeviceToHost));
    return YH_OK;
}

