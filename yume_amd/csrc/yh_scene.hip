// yh_scene.hip — context and scene: device buffers, hierarchy and launch plan, uploads / downloads (planes and flecs'
// 68-byte rows), sparse rows and change lists, stage 1 alone, batched decompose, synthetic scenes, the stream probe.
#define YH_K_SCENE
#include "yh_internal.hpp"

static thread_local std::string g_err;

int fail(yh_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    if (ctx) ctx->err = buf;
    return code;
}

// ---------------------------------------------------------------------------------------------- context
extern "C" uint32_t yh_abi_version(void) { return YH_ABI_VERSION; }

extern "C" const char* yh_last_error(const yh_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

extern "C" int yh_init(int device_id, yh_ctx** out) {
    if (!out) return fail(nullptr, YH_ERR_BAD_ARG, "yh_init: out is NULL");
    *out = nullptr;
    int n = 0;
    HIP_TRY(nullptr, hipGetDeviceCount(&n));
    if (device_id < 0 || device_id >= n)
        return fail(nullptr, YH_ERR_BAD_ARG, "yh_init: device %d out of range (%d devices)", device_id, n);
    yh_ctx* c = new (std::nothrow) yh_ctx();
    if (!c) return fail(nullptr, YH_ERR_OOM, "yh_init: out of host memory");
    c->device = device_id;
    hipError_t e = hipSetDevice(device_id);
    if (e == hipSuccess) e = hipGetDeviceProperties(&c->prop, device_id);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    // The collective stream of the no-sync gather is created and first used here, right after the frames' stream: the
    // runtime multiplexes streams onto four hardware queues in the order they are first used, and a wait packet in a
    // shared queue holds back everything behind it (a gather stream created after RCCL's own landed on the frames'
    // queue and serialised the exchange with the frames: 50 -> 69 us per cfg2 frame).
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream_c, hipStreamNonBlocking);
    for (hipStream_t st : {c->stream, c->stream_c}) {
        hipEvent_t touch = nullptr;
        if (e == hipSuccess) e = hipEventCreateWithFlags(&touch, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(touch, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (touch) (void)hipEventDestroy(touch);
    }
    if (e != hipSuccess) {
        for (hipStream_t st : {c->stream_c, c->stream})
            if (st) (void)hipStreamDestroy(st);
        delete c;
        return fail(nullptr, YH_ERR_HIP, "yh_init: %s", hipGetErrorString(e));
    }
    if (strncmp(c->prop.gcnArchName, "gfx950", 6) != 0) {
        std::string arch = c->prop.gcnArchName;
        for (hipStream_t st : {c->stream_c, c->stream}) (void)hipStreamDestroy(st);
        delete c;
        return fail(nullptr, YH_ERR_UNSUPPORTED, "yh_init: device is %s; this library is built for gfx950 only",
                    arch.c_str());
    }
    // the code object exists for gfx950 only: the symbol is looked up after the architecture is known to be that
    if (resolve_static_scratch(&c->scratch_ptr) != YH_OK) {  // device `device_id` is current
        for (hipStream_t st : {c->stream_c, c->stream}) (void)hipStreamDestroy(st);
        delete c;
        return YH_ERR_HIP;  // (the message is resolve_static_scratch's)
    }
    // The process' first host <-> device copy of some size costs the runtime ~8 ms (its copy queues come up then, 1.5 ms
    // for the same copy afterwards): paid here rather than inside somebody's first frame - both directions, page-locked
    // and ordinary memory, small and large, on the null stream and on the context's own.  Failure is not an error.
    {
        void *d = nullptr, *h = nullptr;
        const size_t bytes = 1u << 20;
        void* pageable = malloc(bytes);
        if (hipMalloc(&d, bytes) == hipSuccess && hipHostMalloc(&h, bytes, hipHostMallocDefault) == hipSuccess && pageable) {
            memset(h, 0, bytes);
            memset(pageable, 0, bytes);
            (void)hipMemcpy(d, h, bytes, hipMemcpyHostToDevice);
            (void)hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost);
            for (void* host : {h, pageable})
                for (size_t sz : {(size_t)256, (size_t)27 * 1024, bytes}) {
                    (void)hipMemcpyAsync(d, host, sz, hipMemcpyHostToDevice, c->stream);
                    (void)hipMemcpyAsync(host, d, sz, hipMemcpyDeviceToHost, c->stream);
                    (void)hipStreamSynchronize(c->stream);
                }
        }
        free(pageable);
        if (h) (void)hipHostFree(h);
        if (d) (void)hipFree(d);
        (void)hipGetLastError();
    }
    *out = c;
    return YH_OK;
}

extern "C" void yh_comm_destroy(yh_ctx* ctx);

extern "C" void yh_shutdown(yh_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    yh_comm_destroy(ctx);
    for (hipStream_t st : {ctx->stream_c, ctx->stream}) {
        if (!st) continue;
        (void)hipStreamSynchronize(st);
        (void)hipStreamDestroy(st);
    }
    release_static(ctx);
    delete ctx;
}

extern "C" int yh_device_info_get(yh_ctx* ctx, yh_device_info* out) {
    if (!ctx || !out) return fail(ctx, YH_ERR_BAD_ARG, "yh_device_info_get: NULL argument");
    memset(out, 0, sizeof *out);
    out->device_id = ctx->device;
    out->n_cus = ctx->prop.multiProcessorCount;
    out->wavefront_size = ctx->prop.warpSize;
    out->hbm_bytes = ctx->prop.totalGlobalMem;
    snprintf(out->arch, sizeof out->arch, "%s", ctx->prop.gcnArchName);
    if (char* colon = strchr(out->arch, ':')) *colon = 0;
    snprintf(out->name, sizeof out->name, "%s", ctx->prop.name);
    return YH_OK;
}

extern "C" int yh_wait(yh_ctx* ctx) {
    if (!ctx) return fail(nullptr, YH_ERR_BAD_ARG, "yh_wait: ctx is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (yh_scene* s : ctx->scenes)
        if (int rc = flush_expand(s)) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->stream_c) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_c));
    release_static(ctx);
    return YH_OK;
}

extern "C" void* yh_stream(yh_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

// Page-locked host memory for the caller's staging columns (rows cross PCIe at link speed from it, and asynchronously).
extern "C" void* yh_host_alloc(yh_ctx* ctx, uint64_t bytes) {
    if (!ctx || !bytes) return nullptr;
    void* p = nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        fail(ctx, YH_ERR_OOM, "yh_host_alloc: %llu bytes of page-locked memory not available", (unsigned long long)bytes);
        return nullptr;
    }
    return p;
}
extern "C" void yh_host_free(yh_ctx* ctx, void* p) {
    if (!p) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    (void)hipHostFree(p);
}

// ------------------------------------------------------------------------------------------------ scene
static void scene_free(yh_scene* s) {
    if (!s) return;
    auto& live = s->ctx->scenes;
    live.erase(std::remove(live.begin(), live.end(), s), live.end());
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    if (s->ctx->stream_c) (void)hipStreamSynchronize(s->ctx->stream_c);
    void* ptrs[] = {s->trs, s->lmat, s->world, s->parent, s->bounds, s->mesh_id, s->ldirty, s->wdirty, s->counts, s->bsets,
                    s->staging, s->d_all_counts, s->chg, s->chg_list, s->chg_list_l, s->alive};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (s->vcount) (void)hipFree(s->vcount);
    if (s->bscratch) (void)hipFree(s->bscratch);
    for (uint32_t f = 0; f < YH_MAX_FRUSTA; ++f) {
        if (s->blist[f]) (void)hipFree(s->blist[f]);
        if (s->bkt[f]) (void)hipFree(s->bkt[f]);
        if (s->obj[f]) (void)hipFree(s->obj[f]);
        if (s->cmd[f]) (void)hipFree(s->cmd[f]);
        if (s->draw[f]) (void)hipFree(s->draw[f]);
        if (s->gathered[f]) (void)hipFree(s->gathered[f]);
    }
    if (s->h_counts) (void)hipHostFree(s->h_counts);
    if (s->h_chg) (void)hipHostFree(s->h_chg);
    if (s->h_all_counts) (void)hipHostFree(s->h_all_counts);
    if (s->ev_frame) (void)hipEventDestroy(s->ev_frame);
    for (hipEvent_t e : s->piece_ev)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : s->ev_mask)
        if (e) (void)hipEventDestroy(e);
    for (void* p : {(void*)s->d_peer, (void*)s->gmask, (void*)s->send_pad, (void*)s->gtile, (void*)s->mask_alt, (void*)s->viscnt_alt})
        if (p) (void)hipFree(p);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    for (hipEvent_t e : s->kev)
        if (e) (void)hipEventDestroy(e);
    delete s;
}

extern "C" void yh_scene_destroy(yh_scene* s) { scene_free(s); }

extern "C" int yh_scene_create(yh_ctx* ctx, const yh_scene_desc* d, yh_scene** out) {
    if (!ctx || !d || !out) return fail(ctx, YH_ERR_BAD_ARG, "yh_scene_create: NULL argument");
    *out = nullptr;
    ARG_TRY(ctx, d->n_entities >= 1 && d->n_entities <= (1ull << 30), "yh_scene_create: n_entities must be in [1, 2^30]");
    ARG_TRY(ctx, d->n_levels >= 1 && d->n_levels <= YH_MAX_LEVELS, "yh_scene_create: n_levels must be in [1, 64]");
    ARG_TRY(ctx, d->n_frusta <= YH_MAX_FRUSTA, "yh_scene_create: n_frusta exceeds YH_MAX_FRUSTA");
    ARG_TRY(ctx, d->local_kind <= YH_LOCAL_MAT4, "yh_scene_create: bad local_kind");
    ARG_TRY(ctx, d->bounds_kind <= YH_BOUNDS_MESH_TABLE, "yh_scene_create: bad bounds_kind");
    ARG_TRY(ctx, d->bounds_kind != YH_BOUNDS_MESH_TABLE || d->max_meshes >= 1, "yh_scene_create: max_meshes must be >= 1");
    ARG_TRY(ctx, d->bounds_kind != YH_BOUNDS_NONE || d->n_frusta == 0, "yh_scene_create: frusta without bounds");
    HIP_TRY(ctx, hipSetDevice(ctx->device));

    yh_scene* s = new (std::nothrow) yh_scene();
    if (!s) return fail(ctx, YH_ERR_OOM, "yh_scene_create: out of host memory");
    s->ctx = ctx;
    s->desc = *d;
    s->n = d->n_entities;
    s->stride = (s->n + 63) & ~63ull;
    s->draw_cap = d->draw_capacity ? d->draw_capacity : s->n;
    for (int i = 0; i < 16; ++i) s->root[i] = (i % 5 == 0) ? 1.0f : 0.0f;

    const uint64_t S = s->stride;
    hipError_t e = hipSuccess;
    auto alloc = [&](void** p, uint64_t bytes) {
        if (e == hipSuccess) e = hipMalloc(p, bytes);
    };
    if (d->local_kind == YH_LOCAL_TRS) alloc((void**)&s->trs, 10 * S * 4);
    if (d->local_kind == YH_LOCAL_MAT4) alloc((void**)&s->lmat, 16 * S * 4);
    alloc((void**)&s->world, 16 * S * 4);
    alloc((void**)&s->parent, S * 4);
    alloc((void**)&s->ldirty, S);
    alloc((void**)&s->wdirty, S);
    if (d->bounds_kind == YH_BOUNDS_PER_ENTITY) alloc((void**)&s->bounds, 6 * S * 4);
    if (d->bounds_kind == YH_BOUNDS_MESH_TABLE) {
        alloc((void**)&s->bounds, (uint64_t)d->max_meshes * 6 * 4);
        alloc((void**)&s->mesh_id, S * 4);
    }
    for (uint32_t f = 0; f < d->n_frusta; ++f) alloc((void**)&s->draw[f], s->draw_cap * 4);
    // stage-3 buffers: one allocation = 32 total words | tile counts | visibility words | their popcounts; zeroed once
    // here, never per frame (tile counts and visibility words are fully rewritten, boundary counters alternate)
    s->n_chunks = (uint32_t)(S / YH_CHUNK);
    // expand tiles: 32 words (2048 entities) while that still fits the self-counting path (1M entities: 39.8 vs 40.6 us
    // per frame), 64 words keep mid-size scenes parallel, 256 words beyond 8M entities (all measured)
    s->tile_words = s->n_chunks <= 32u * 512u ? 32u
                    : s->n_chunks <= 32u * 4096u                       ? 64u
                                                                       : (uint32_t)YH_EXPAND_WORDS_MAX;
    if (const char* v = getenv("YH_TILE_WORDS")) {  // A/B of the expand tile size (tools/ab_tile_words.sh): 32, 64 or 256
        const uint32_t w = (uint32_t)strtoul(v, nullptr, 10);
        if (w == 32u || w == 64u || w == 256u) s->tile_words = w;
    }
    s->n_tiles = (s->n_chunks + s->tile_words - 1) / s->tile_words;
    const uint64_t F_alloc = d->n_frusta ? d->n_frusta : 1;
    s->n_segs = (s->n_chunks + (1u << YH_SEG_SHIFT) - 1) >> YH_SEG_SHIFT;
    const uint64_t tile_words = (F_alloc * (s->n_tiles + s->n_segs) + 1) & ~1ull;
    s->cnt_stride = (s->n_chunks + 15u) & ~15u;
    s->vis_bytes = (4 * YH_MAX_FRUSTA + tile_words) * 4 + F_alloc * s->n_chunks * 8 + F_alloc * s->cnt_stride + 16;
    alloc((void**)&s->counts, s->vis_bytes);
    alloc((void**)&s->bsets, (uint64_t)YH_BSETS * YH_MAX_FRUSTA * 4);
    if (e == hipSuccess) {
        s->tilecnt = s->counts + 4 * YH_MAX_FRUSTA;  // [0,8) totals ([8,32): unused since the boundary counters moved to s->bsets)
        s->vismask = (uint64_t*)(s->tilecnt + tile_words);
        // byte popcounts of the words, rows of cnt_stride bytes read 16 at a time: 16-byte aligned start
        s->viscnt = (uint8_t*)(((uintptr_t)(s->vismask + F_alloc * s->n_chunks) + 15) & ~(uintptr_t)15);
    }
    if (d->n_frusta) {  // the other set(s) of visibility words and of their byte counts (frames alternate)
        const uint64_t wb = (uint64_t)(YH_MASK_SETS - 1) * F_alloc * s->n_chunks * 8, cb = (uint64_t)(YH_MASK_SETS - 1) * F_alloc * s->cnt_stride;
        alloc((void**)&s->mask_alt, wb);
        alloc((void**)&s->viscnt_alt, cb);
        if (e == hipSuccess) e = hipMemsetAsync(s->mask_alt, 0, wb, ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(s->viscnt_alt, 0, cb, ctx->stream);
    }
    if (e == hipSuccess) e = hipHostMalloc((void**)&s->h_counts, 2 * YH_MAX_FRUSTA * 4, hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&s->h_counts_dev, s->h_counts, 0);
    if (e == hipSuccess) e = hipEventCreate(&s->ev0);
    if (e == hipSuccess) e = hipEventCreate(&s->ev1);
    // world = IDENTITY, dirty: LocalTransform.onAdd (transform.zig:38-44)
    if (e == hipSuccess) e = hipMemsetAsync(s->world, 0, 16 * S * 4, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(s->ldirty, 1, S, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(s->wdirty, 1, S, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(s->counts, 0, s->vis_bytes, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(s->bsets, 0, (uint64_t)YH_BSETS * YH_MAX_FRUSTA * 4, ctx->stream);
    // every entity starts with mesh 0: a frame or a consumer that runs before yh_upload_mesh_ids covered every entity
    // indexes the table, the vertex counts and the bucket histogram in bounds (hipMalloc does not zero)
    if (e == hipSuccess && s->mesh_id) e = hipMemsetAsync(s->mesh_id, 0, S * 4, ctx->stream);
    if (e == hipSuccess) {
        memset(s->h_counts, 0, 2 * YH_MAX_FRUSTA * 4);
        const float one = 1.0f;
        uint32_t bits;
        memcpy(&bits, &one, 4);
        (void)bits;
        for (int p : {0, 5, 10, 15}) {
            hipLaunchKernelGGL(k_fill_plane, dim3(cdiv(s->n, YH_BLOCK)), dim3(YH_BLOCK), 0, ctx->stream, s->world, 16u,
                               (uint32_t)p, one, s->n);
            if (e == hipSuccess) e = hipGetLastError();
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        int code = fail(ctx, e == hipErrorOutOfMemory ? YH_ERR_OOM : YH_ERR_HIP, "yh_scene_create: %s", hipGetErrorString(e));
        scene_free(s);
        return code;
    }
    if (const char* v = getenv("YH_NO_CULL_PLAN")) s->use_cull_plan = !(v[0] == '1');
    if (const char* v = getenv("YH_LDS_PARENTS")) s->lds_parents = v[0] == '1';
    if (const char* v = getenv("YH_NO_SPEC")) s->spec_clean = !(v[0] == '1');
    if (const char* v = getenv("YH_NO_DIRECT_ROWS")) s->direct_rows = !(v[0] == '1');
    if (const char* v = getenv("YH_BIG_TREE")) s->big_tree = strtoull(v, nullptr, 10);
    if (const char* v = getenv("YH_BIG_FLAT")) s->big_flat = strtoull(v, nullptr, 10);
    if (const char* v = getenv("YH_DEFER_FIRST_MAX")) s->defer_first_max = (uint32_t)strtoul(v, nullptr, 10);
    ctx->scenes.push_back(s);
    *out = s;
    return YH_OK;
}

int check_range(yh_scene* s, const char* who, uint64_t first, uint64_t count) {
    if (!s) return fail(nullptr, YH_ERR_BAD_ARG, "%s: scene is NULL", who);
    if (first > s->n || count > s->n - first)
        return fail(s->ctx, YH_ERR_BAD_ARG, "%s: range [%llu, +%llu) exceeds %llu entities", who,
                    (unsigned long long)first, (unsigned long long)count, (unsigned long long)s->n);
    return YH_OK;
}

// Cuts the levels of a TRS scene into launches (Mat4 scenes run one level per launch: the dirty rule needs the
// parent's flag of this frame).  Measured on MI355X with tools/plan_sweep.py over all 128 cuts of the 8-level
// 2^k trees of 1M and 12.5M entities (profiles/r01/plan_sweep_*.txt): every dependent launch costs 2-6 us, a chain
// group costs its deepest entity's dependent walk, so
//   - levels whose sum fits one residency of the chip (~320k lanes) share one launch, up to 8 levels (the first
//     launch of a small scene that needs a second one: ~128k, see below);
//   - larger levels go three to a launch (one or two ancestors re-derived per entity beats a launch boundary
//     and the 64-byte parent gather even at 6M entities per level), the short group first.
// An explicit plan or grouping (yh_scene_set_plan / yh_scene_set_grouping) replaces this.
static int plan_groups(yh_scene* s) {
    s->groups.clear();
    const uint32_t L = s->desc.n_levels;
    if (!s->explicit_plan.empty()) {
        s->groups = s->explicit_plan;
        return YH_OK;
    }
    auto span = [&](uint32_t lo, uint32_t hi) -> uint64_t { return (uint64_t)s->level_offsets[hi + 1] - s->level_offsets[lo]; };
    // Scenes whose expansion rides on the next frame's first launch (small TRS scenes, see yh_frame) do better with a
    // lighter first launch once they need a second one anyway: 1M entities 34.5 us for 0..4 | 5..7 against 35.1 for
    // 0..5 | 6..7, 500k 22.2 us for 0..5 | 6..7 against 23.4 for 0..6 | 7; a 250k scene stays one launch (13.9 us).
    const bool rides = s->desc.local_kind == YH_LOCAL_TRS && s->desc.bounds_kind != YH_BOUNDS_NONE &&
                       (s->n_tiles <= (uint32_t)YH_SELF_COUNT_TILES && s->tile_words <= 64u) && s->n > 320 * 1024;
    // Large TRS trees (re-measured in round 2 over all 128 cuts at 4M, 8M, 12.5M and 25M entities,
    // profiles/r02/plan_sweep_*.txt): from the LAST level backwards, a level of 5M entities or more is a launch of its
    // own (the one-level kernel: no ancestors re-read), smaller levels go up to three to a launch while the launch stays
    // under ~6.1M entities, and once the levels above sum to 1M or less they share one chain launch.  12.5M: 0..3 | 4..6
    // | 7, 331.8 us (the forward triples 0..1 | 2..4 | 5..7: 334.8; best of the 128: 329.4); 4M: 0..4 | 5..7, 107.7
    // (before 0..3 | 4 | 5..7: >= 109.3; best 106.3); 8M: 0..2 | 3..5 | 6..7 and 1M: 0..4 | 5..7 as before, both the best.
    if (s->desc.local_kind == YH_LOCAL_TRS && !s->max_group_levels && s->n > (2u << 20) && L > 1) {
        std::vector<LevelGroup> back;
        uint32_t hi = L - 1;
        for (;;) {
            if (span(0, hi) <= (1u << 20) && hi + 1 <= (uint32_t)YH_MAX_GROUP) {  // the small levels at the top: one chain launch
                LevelGroup g;
                g.lo = 0;
                g.hi = hi;
                back.push_back(g);
                break;
            }
            uint32_t lo = hi;
            if (span(hi, hi) < 5000000ull)
                while (lo > 0 && hi - lo + 1 < 3 && span(lo - 1, hi) <= 6100000ull) --lo;
            LevelGroup g;
            g.lo = lo;
            g.hi = hi;
            back.push_back(g);
            if (lo == 0) break;
            hi = lo - 1;
        }
        s->groups.assign(back.rbegin(), back.rend());
        return YH_OK;
    }
    uint32_t lo = 0;
    while (lo < L) {
        const uint64_t resident = (lo == 0 && rides) ? 128 * 1024 : 320 * 1024;
        uint32_t hi = lo;
        if (s->max_group_levels == 1) {
            // one level per launch
        } else if (s->max_group_levels) {
            hi = std::min<uint32_t>(L - 1, lo + std::min<uint32_t>(s->max_group_levels, YH_MAX_GROUP) - 1);
        } else {
            while (hi + 1 < L && hi + 1 - lo < YH_MAX_GROUP && span(lo, hi + 1) <= resident) ++hi;
            if (hi == lo && span(lo, lo) > resident / 4 && s->desc.local_kind == YH_LOCAL_TRS) {  // a large level: groups of three, the remainder first
                // (Mat4 scenes keep large levels one per launch: replaying ancestors' flags and rows costs more than the
                // launch boundary there, measured 314 vs 218 us on a clean 12.5M-entity frame)
                uint32_t big_end = lo;  // the run of levels that no longer fit the residency rule
                while (big_end + 1 < L) ++big_end;
                const uint32_t run = big_end - lo + 1;
                const uint32_t take = run % 3 ? run % 3 : 3;
                hi = lo + take - 1;
            }
        }
        LevelGroup g;
        g.lo = lo;
        g.hi = hi;
        s->groups.push_back(g);
        lo = hi + 1;
    }
    return YH_OK;
}

extern "C" int yh_scene_set_grouping(yh_scene* s, uint32_t max_group_levels) {
    if (!s) return fail(nullptr, YH_ERR_BAD_ARG, "yh_scene_set_grouping: scene is NULL");
    ARG_TRY(s->ctx, max_group_levels <= YH_MAX_GROUP, "yh_scene_set_grouping: at most 8 levels per group");
    s->max_group_levels = max_group_levels;
    s->explicit_plan.clear();
    if (s->have_hierarchy) return plan_groups(s);
    return YH_OK;
}

extern "C" int yh_scene_set_lds_parents(yh_scene* s, int enable) {
    if (!s) return fail(nullptr, YH_ERR_BAD_ARG, "yh_scene_set_lds_parents: scene is NULL");
    s->lds_parents = enable != 0;
    return YH_OK;
}

extern "C" int yh_scene_set_plan(yh_scene* s, uint32_t n_groups, const uint32_t* last_level) {
    if (!s) return fail(nullptr, YH_ERR_BAD_ARG, "yh_scene_set_plan: scene is NULL");
    yh_ctx* ctx = s->ctx;
    std::vector<LevelGroup> plan;
    if (n_groups) {
        ARG_TRY(ctx, last_level, "yh_scene_set_plan: last_level is NULL");
        uint32_t lo = 0;
        for (uint32_t i = 0; i < n_groups; ++i) {
            LevelGroup g;
            g.lo = lo;
            g.hi = last_level[i];
            ARG_TRY(ctx, g.hi >= g.lo && g.hi < s->desc.n_levels && g.hi - g.lo < YH_MAX_GROUP,
                    "yh_scene_set_plan: groups must be consecutive, non-empty and at most 8 levels each");
            plan.push_back(g);
            lo = g.hi + 1;
        }
        ARG_TRY(ctx, lo == s->desc.n_levels, "yh_scene_set_plan: the groups must cover every level");
    }
    s->explicit_plan = plan;
    if (s->have_hierarchy) return plan_groups(s);
    return YH_OK;
}

extern "C" int yh_upload_hierarchy(yh_scene* s, const uint32_t* parent_index, const uint32_t* level_offsets) {
    if (!s) return fail(nullptr, YH_ERR_BAD_ARG, "yh_upload_hierarchy: scene is NULL");
    yh_ctx* ctx = s->ctx;
    ARG_TRY(ctx, parent_index && level_offsets, "yh_upload_hierarchy: NULL argument");
    const uint32_t L = s->desc.n_levels;
    ARG_TRY(ctx, level_offsets[0] == 0 && level_offsets[L] == s->n, "yh_upload_hierarchy: level_offsets must span [0, n_entities]");
    for (uint32_t k = 0; k < L; ++k)
        ARG_TRY(ctx, level_offsets[k] <= level_offsets[k + 1], "yh_upload_hierarchy: level_offsets must be non-decreasing");
    // every parent must live in the previous level: this is what makes the kernels' gathers in-bounds
    for (uint32_t k = 1; k < L; ++k) {
        const uint32_t plo = level_offsets[k - 1], phi = level_offsets[k];
        for (uint64_t e = level_offsets[k]; e < level_offsets[k + 1]; ++e) {
            const uint32_t p = parent_index[e];
            if (p < plo || p >= phi)
                return fail(ctx, YH_ERR_BAD_ARG, "yh_upload_hierarchy: entity %llu (level %u) has parent %u outside level %u",
                            (unsigned long long)e, k, p, k - 1);
        }
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = sync_scene(s)) return rc;  // frames in flight still read the old hierarchy
    HIP_TRY(ctx, hipMemcpy(s->parent, parent_index, s->n * 4, hipMemcpyHostToDevice));
    s->level_offsets.assign(level_offsets, level_offsets + L + 1);
    s->have_hierarchy = true;
    return plan_groups(s);
}

// Parent indices of slots [first, first + count), all of ONE level; the level layout and the launch plan stay.
extern "C" int yh_upload_parent_range(yh_scene* s, const uint32_t* parent_index, uint64_t first, uint64_t count) {
    if (int rc = check_range(s, "yh_upload_parent_range", first, count)) return rc;
    yh_ctx* ctx = s->ctx;
    ARG_TRY(ctx, parent_index, "yh_upload_parent_range: parent_index is NULL");
    if (!s->have_hierarchy) return fail(ctx, YH_ERR_STATE, "yh_upload_parent_range: call yh_upload_hierarchy first");
    if (count == 0) return YH_OK;
    const uint32_t L = s->desc.n_levels;
    uint32_t k = 0;
    while (k + 1 < L && s->level_offsets[k + 1] <= first) ++k;
    if (first + count > s->level_offsets[k + 1])
        return fail(ctx, YH_ERR_BAD_ARG, "yh_upload_parent_range: slots [%llu, %llu) leave level %u", (unsigned long long)first,
                    (unsigned long long)(first + count), k);
    for (uint64_t i = 0; i < count; ++i) {
        const uint32_t p = parent_index[i];
        const bool ok = k == 0 ? true : (p >= s->level_offsets[k - 1] && p < s->level_offsets[k]);  // level 0: value unused
        if (!ok)
            return fail(ctx, YH_ERR_BAD_ARG, "yh_upload_parent_range: slot %llu (level %u) has parent %u outside level %u",
                        (unsigned long long)(first + i), k, p, k - 1);
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = sync_scene(s)) return rc;  // frames in flight still read the old indices
    HIP_TRY(ctx, hipMemcpy(s->parent + first, parent_index, count * 4, hipMemcpyHostToDevice));
    return YH_OK;
}

// Slots that hold no entity: never listed as visible, never counted as boundary cases, never reported as changed.
extern "C" int yh_scene_set_alive_range(yh_scene* s, uint64_t first, uint64_t count, int alive) {
    if (int rc = check_range(s, "yh_scene_set_alive_range", first, count)) return rc;
    yh_ctx* ctx = s->ctx;
    if (s->desc.local_kind != YH_LOCAL_MAT4)
        return fail(ctx, YH_ERR_STATE, "yh_scene_set_alive_range: TRS scenes have no spare slots (every slot is an instance)");
    if (count == 0) return YH_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = sync_scene(s)) return rc;
    if (!s->alive) {
        s->h_alive.assign(s->n_chunks, ~0ull);
        if (s->n & 63u) s->h_alive.back() = (1ull << (s->n & 63u)) - 1ull;
        HIP_TRY(ctx, hipMalloc((void**)&s->alive, (uint64_t)s->n_chunks * 8));
        HIP_TRY(ctx, hipMemcpy(s->alive, s->h_alive.data(), (uint64_t)s->n_chunks * 8, hipMemcpyHostToDevice));
    }
    const uint64_t c0 = first >> 6, c1 = (first + count - 1) >> 6;
    for (uint64_t c = c0; c <= c1; ++c) {
        const uint64_t lo = c == c0 ? (first & 63u) : 0u, hi = c == c1 ? ((first + count - 1) & 63u) : 63u;
        const uint64_t bits = (hi == 63u ? ~0ull : ((1ull << (hi + 1)) - 1ull)) & ~((1ull << lo) - 1ull);
        if (alive) s->h_alive[c] |= bits;
        else s->h_alive[c] &= ~bits;
    }
    HIP_TRY(ctx, hipMemcpy(s->alive + c0, s->h_alive.data() + c0, (c1 - c0 + 1) * 8, hipMemcpyHostToDevice));
    return YH_OK;
}

extern "C" int yh_set_root(yh_scene* s, const float root_world[16]) {
    if (!s || !root_world) return fail(s ? s->ctx : nullptr, YH_ERR_BAD_ARG, "yh_set_root: NULL argument");
    memcpy(s->root, root_world, sizeof s->root);
    s->root_dirty = 1;  // WorldTransform.onSet (transform.zig:92-97)
    return YH_OK;
}

extern "C" int yh_upload_root_aos68(yh_scene* s, const void* row) {
    if (!s || !row) return fail(s ? s->ctx : nullptr, YH_ERR_BAD_ARG, "yh_upload_root_aos68: NULL argument");
    memcpy(s->root, row, sizeof s->root);
    s->root_dirty = static_cast<const uint8_t*>(row)[64] != 0;  // the row's own flag: no onSet implied
    return YH_OK;
}


// elements per staging pass of plane uploads / downloads
#define YH_PLANE_CHUNK (16ull << 20)

// Host planes -> chunked device array `dev` (np_total planes per chunk), planes [q0, q0+np).
static int copy_planes_h2d(yh_scene* s, float* dev, uint32_t np_total, uint32_t q0, const float* const* planes, int np,
                           uint64_t first, uint64_t count) {
    HIP_TRY(s->ctx, hipSetDevice(s->ctx->device));
    if (count == 0) return YH_OK;
    const uint64_t chunk = std::min<uint64_t>(count, YH_PLANE_CHUNK);
    if (int rc = ensure_staging(s, chunk * 4)) return rc;
    hipStream_t st = s->ctx->stream;
    for (int p = 0; p < np; ++p) {
        ARG_TRY(s->ctx, planes[p], "plane pointer is NULL");
        for (uint64_t done = 0; done < count; done += chunk) {
            const uint64_t c = std::min<uint64_t>(chunk, count - done);
            HIP_TRY(s->ctx, hipMemcpyAsync(s->staging, planes[p] + done, c * 4, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_scatter_plane, dim3(cdiv(c, YH_BLOCK)), dim3(YH_BLOCK), 0, st, (const float*)s->staging, dev,
                               np_total, q0 + (uint32_t)p, first + done, c);
            HIP_TRY(s->ctx, hipGetLastError());
            HIP_TRY(s->ctx, hipStreamSynchronize(st));
        }
    }
    return YH_OK;
}

static int copy_planes_d2h(yh_scene* s, const float* dev, uint32_t np_total, uint32_t q0, float* const* planes, int np,
                           uint64_t first, uint64_t count) {
    HIP_TRY(s->ctx, hipSetDevice(s->ctx->device));
    if (int rc = sync_scene(s)) return rc;
    if (count == 0) return YH_OK;
    const uint64_t chunk = std::min<uint64_t>(count, YH_PLANE_CHUNK);
    if (int rc = ensure_staging(s, chunk * 4)) return rc;
    hipStream_t st = s->ctx->stream;
    for (int p = 0; p < np; ++p) {
        ARG_TRY(s->ctx, planes[p], "plane pointer is NULL");
        for (uint64_t done = 0; done < count; done += chunk) {
            const uint64_t c = std::min<uint64_t>(chunk, count - done);
            hipLaunchKernelGGL(k_gather_plane, dim3(cdiv(c, YH_BLOCK)), dim3(YH_BLOCK), 0, st, dev, (float*)s->staging,
                               np_total, q0 + (uint32_t)p, first + done, c);
            HIP_TRY(s->ctx, hipGetLastError());
            HIP_TRY(s->ctx, hipMemcpyAsync(planes[p] + done, s->staging, c * 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(s->ctx, hipStreamSynchronize(st));
        }
    }
    return YH_OK;
}

extern "C" int yh_upload_trs(yh_scene* s, const float* const planes[10], uint64_t first, uint64_t count) {
    if (int rc = check_range(s, "yh_upload_trs", first, count)) return rc;
    ARG_TRY(s->ctx, planes, "yh_upload_trs: planes is NULL");
    if (s->desc.local_kind != YH_LOCAL_TRS) return fail(s->ctx, YH_ERR_STATE, "yh_upload_trs: scene holds Mat4 locals");
    return copy_planes_h2d(s, s->trs, 10, 0, planes, 10, first, count);
}

extern "C" int yh_download_trs(yh_scene* s, float* const planes[10], uint64_t first, uint64_t count) {
    if (int rc = check_range(s, "yh_download_trs", first, count)) return rc;
    ARG_TRY(s->ctx, planes, "yh_download_trs: planes is NULL");
    if (s->desc.local_kind != YH_LOCAL_TRS) return fail(s->ctx, YH_ERR_STATE, "yh_download_trs: scene holds Mat4 locals");
    return copy_planes_d2h(s, s->trs, 10, 0, planes, 10, first, count);
}

extern "C" int yh_upload_aabb(yh_scene* s, const float* const min_planes[3], const float* const max_planes[3],
                              uint64_t first, uint64_t count) {
    if (int rc = check_range(s, "yh_upload_aabb", first, count)) return rc;
    ARG_TRY(s->ctx, min_planes && max_planes, "yh_upload_aabb: NULL argument");
    if (s->desc.bounds_kind != YH_BOUNDS_PER_ENTITY) return fail(s->ctx, YH_ERR_STATE, "yh_upload_aabb: scene has no per-entity bounds");
    if (int rc = copy_planes_h2d(s, s->bounds, 6, 0, min_planes, 3, first, count)) return rc;
    return copy_planes_h2d(s, s->bounds, 6, 3, max_planes, 3, first, count);
}

extern "C" int yh_download_aabb(yh_scene* s, float* const min_planes[3], float* const max_planes[3], uint64_t first,
                                uint64_t count) {
    if (int rc = check_range(s, "yh_download_aabb", first, count)) return rc;
    ARG_TRY(s->ctx, min_planes && max_planes, "yh_download_aabb: NULL argument");
    if (s->desc.bounds_kind != YH_BOUNDS_PER_ENTITY) return fail(s->ctx, YH_ERR_STATE, "yh_download_aabb: scene has no per-entity bounds");
    if (int rc = copy_planes_d2h(s, s->bounds, 6, 0, min_planes, 3, first, count)) return rc;
    return copy_planes_d2h(s, s->bounds, 6, 3, max_planes, 3, first, count);
}

extern "C" int yh_upload_mesh_aabbs(yh_scene* s, const float* table, uint32_t n_meshes) {
    if (!s || !table) return fail(s ? s->ctx : nullptr, YH_ERR_BAD_ARG, "yh_upload_mesh_aabbs: NULL argument");
    if (s->desc.bounds_kind != YH_BOUNDS_MESH_TABLE) return fail(s->ctx, YH_ERR_STATE, "yh_upload_mesh_aabbs: scene has no mesh table");
    ARG_TRY(s->ctx, n_meshes >= 1 && n_meshes <= s->desc.max_meshes, "yh_upload_mesh_aabbs: n_meshes out of range");
    if (n_meshes <= s->max_mesh_id)  // ids validated against the old table would index past the new one
        return fail(s->ctx, YH_ERR_STATE, "yh_upload_mesh_aabbs: %u meshes, but mesh id %u is in use; upload smaller ids first",
                    n_meshes, s->max_mesh_id);
    HIP_TRY(s->ctx, hipSetDevice(s->ctx->device));
    if (int rc = sync_scene(s)) return rc;  // frames in flight still read the old table (their stream is non-blocking)
    HIP_TRY(s->ctx, hipMemcpyAsync(s->bounds, table, (uint64_t)n_meshes * 24, hipMemcpyHostToDevice, s->ctx->stream));
    HIP_TRY(s->ctx, hipStreamSynchronize(s->ctx->stream));
    s->n_meshes = n_meshes;
    return YH_OK;
}

extern "C" int yh_upload_mesh_ids(yh_scene* s, const uint32_t* mesh_id, uint64_t first, uint64_t count) {
    if (int rc = check_range(s, "yh_upload_mesh_ids", first, count)) return rc;
    ARG_TRY(s->ctx, mesh_id, "yh_upload_mesh_ids: mesh_id is NULL");
    if (s->desc.bounds_kind != YH_BOUNDS_MESH_TABLE) return fail(s->ctx, YH_ERR_STATE, "yh_upload_mesh_ids: scene has no mesh table");
    ARG_TRY(s->ctx, s->n_meshes >= 1, "yh_upload_mesh_ids: upload the mesh table first");
    uint32_t top = 0;
    for (uint64_t i = 0; i < count; ++i) {
        if (mesh_id[i] >= s->n_meshes)
            return fail(s->ctx, YH_ERR_BAD_ARG, "yh_upload_mesh_ids: id %u at %llu >= n_meshes %u", mesh_id[i],
                        (unsigned long long)(first + i), s->n_meshes);
        top = std::max(top, mesh_id[i]);
    }
    HIP_TRY(s->ctx, hipSetDevice(s->ctx->device));
    if (int rc = sync_scene(s)) return rc;
    if (count) {
        HIP_TRY(s->ctx, hipMemcpyAsync(s->mesh_id + first, mesh_id, count * 4, hipMemcpyHostToDevice, s->ctx->stream));
        HIP_TRY(s->ctx, hipStreamSynchronize(s->ctx->stream));
    }
    // (a full re-upload is the way to lower the mark again: it then describes every entity)
    s->max_mesh_id = (first == 0 && count == s->n) ? top : std::max(s->max_mesh_id, top);
    return YH_OK;
}

// ---- the same updates for MANY short runs at once (a structural patch of the drop-in touches thousands of tables: one
// blocking call per table and array was ~0.15 ms a table) -------------------------------------------------------------
extern "C" int yh_scene_set_alive_runs(yh_scene* s, const uint32_t* first, const uint32_t* count, const uint8_t* alive, uint32_t n_runs) {
    if (!s) return fail(nullptr, YH_ERR_BAD_ARG, "yh_scene_set_alive_runs: scene is NULL");
    yh_ctx* ctx = s->ctx;
    if (n_runs == 0) return YH_OK;
    ARG_TRY(ctx, first && count && alive, "yh_scene_set_alive_runs: NULL argument");
    if (s->desc.local_kind != YH_LOCAL_MAT4)
        return fail(ctx, YH_ERR_STATE, "yh_scene_set_alive_runs: TRS scenes have no spare slots (every slot is an instance)");
    for (uint32_t r = 0; r < n_runs; ++r)
        if ((uint64_t)first[r] + count[r] > s->n)
            return fail(ctx, YH_ERR_BAD_ARG, "yh_scene_set_alive_runs: run %u [%u, +%u) leaves the scene's %llu slots", r, first[r], count[r],
                        (unsigned long long)s->n);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = sync_scene(s)) return rc;
    if (!s->alive) {
        s->h_alive.assign(s->n_chunks, ~0ull);
        if (s->n & 63u) s->h_alive.back() = (1ull << (s->n & 63u)) - 1ull;
        HIP_TRY(ctx, hipMalloc((void**)&s->alive, (uint64_t)s->n_chunks * 8));
        HIP_TRY(ctx, hipMemcpy(s->alive, s->h_alive.data(), (uint64_t)s->n_chunks * 8, hipMemcpyHostToDevice));
    }
    uint64_t lo_c = ~0ull, hi_c = 0;
    for (uint32_t r = 0; r < n_runs; ++r) {  // in the order given: a later run overrides an earlier one where they overlap
        if (!count[r]) continue;
        const uint64_t f = first[r], c0 = f >> 6, c1 = (f + count[r] - 1) >> 6;
        for (uint64_t c = c0; c <= c1; ++c) {
            const uint64_t lo = c == c0 ? (f & 63u) : 0u, hi = c == c1 ? ((f + count[r] - 1) & 63u) : 63u;
            const uint64_t bits = (hi == 63u ? ~0ull : ((1ull << (hi + 1)) - 1ull)) & ~((1ull << lo) - 1ull);
            if (alive[r]) s->h_alive[c] |= bits;
            else s->h_alive[c] &= ~bits;
        }
        lo_c = std::min(lo_c, c0);
        hi_c = std::max(hi_c, c1);
    }
    if (lo_c <= hi_c) HIP_TRY(ctx, hipMemcpy(s->alive + lo_c, s->h_alive.data() + lo_c, (hi_c - lo_c + 1) * 8, hipMemcpyHostToDevice));
    return YH_OK;
}

// slots + values to the staging area, one scatter launch; returns with the copy done (the host arrays are the caller's again)
static int scatter_words(yh_scene* s, uint32_t* dst, const uint32_t* slots, const uint32_t* vals, uint64_t count) {
    yh_ctx* ctx = s->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = sync_scene(s)) return rc;  // frames in flight still read the old values
    if (int rc = ensure_staging(s, count * 8)) return rc;
    uint32_t* d_slots = (uint32_t*)s->staging;
    uint32_t* d_vals = d_slots + count;
    HIP_TRY(ctx, hipMemcpyAsync(d_slots, slots, count * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_vals, vals, count * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_scatter_u32, dim3((uint32_t)cdiv(count, YH_BLOCK)), dim3(YH_BLOCK), 0, ctx->stream, dst, d_slots, d_vals, count);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return YH_OK;
}

// Parent indices of single slots (any levels): slot slots[i] gets parent parent_index[i], which must lie in the level
// before the slot's own (level-0 slots: value unused).  Should a slot be named twice, either value may win.
extern "C" int yh_upload_parent_rows(yh_scene* s, const uint32_t* slots, const uint32_t* parent_index, uint64_t count) {
    if (!s) return fail(nullptr, YH_ERR_BAD_ARG, "yh_upload_parent_rows: scene is NULL");
    yh_ctx* ctx = s->ctx;
    if (count == 0) return YH_OK;
    ARG_TRY(ctx, slots && parent_index, "yh_upload_parent_rows: NULL argument");
    if (!s->have_hierarchy) return fail(ctx, YH_ERR_STATE, "yh_upload_parent_rows: call yh_upload_hierarchy first");
    const uint32_t L = s->desc.n_levels;
    uint32_t k = 0;
    for (uint64_t i = 0; i < count; ++i) {
        const uint32_t e = slots[i];
        if (e >= s->n) return fail(ctx, YH_ERR_BAD_ARG, "yh_upload_parent_rows: slot %u at %llu >= n_entities", e, (unsigned long long)i);
        if (!(e >= s->level_offsets[k] && e < s->level_offsets[k + 1])) {  // (runs of slots of one level: the last level found is tried first)
            k = 0;
            while (k + 1 < L && s->level_offsets[k + 1] <= e) ++k;
        }
        const uint32_t p = parent_index[i];
        if (k > 0 && !(p >= s->level_offsets[k - 1] && p < s->level_offsets[k]))
            return fail(ctx, YH_ERR_BAD_ARG, "yh_upload_parent_rows: slot %u (level %u) has parent %u outside level %u", e, k, p, k - 1);
    }
    return scatter_words(s, s->parent, slots, parent_index, count);
}

extern "C" int yh_upload_mesh_id_rows(yh_scene* s, const uint32_t* slots, const uint32_t* mesh_id, uint64_t count) {
    if (!s) return fail(nullptr, YH_ERR_BAD_ARG, "yh_upload_mesh_id_rows: scene is NULL");
    yh_ctx* ctx = s->ctx;
    if (count == 0) return YH_OK;
    ARG_TRY(ctx, slots && mesh_id, "yh_upload_mesh_id_rows: NULL argument");
    if (s->desc.bounds_kind != YH_BOUNDS_MESH_TABLE) return fail(ctx, YH_ERR_STATE, "yh_upload_mesh_id_rows: scene has no mesh table");
    ARG_TRY(ctx, s->n_meshes >= 1, "yh_upload_mesh_id_rows: upload the mesh table first");
    uint32_t top = 0;
    for (uint64_t i = 0; i < count; ++i) {
        if (slots[i] >= s->n) return fail(ctx, YH_ERR_BAD_ARG, "yh_upload_mesh_id_rows: slot %u at %llu >= n_entities", slots[i], (unsigned long long)i);
        if (mesh_id[i] >= s->n_meshes)
            return fail(ctx, YH_ERR_BAD_ARG, "yh_upload_mesh_id_rows: id %u at %llu >= n_meshes %u", mesh_id[i], (unsigned long long)i, s->n_meshes);
        top = std::max(top, mesh_id[i]);
    }
    if (int rc = scatter_words(s, s->mesh_id, slots, mesh_id, count)) return rc;
    s->max_mesh_id = std::max(s->max_mesh_id, top);
    return YH_OK;
}

// Per-entity boxes of single slots: min_max holds count x {min.x, min.y, min.z, max.x, max.y, max.z}.
extern "C" int yh_upload_aabb_rows(yh_scene* s, const uint32_t* slots, const float* min_max, uint64_t count) {
    if (!s) return fail(nullptr, YH_ERR_BAD_ARG, "yh_upload_aabb_rows: scene is NULL");
    yh_ctx* ctx = s->ctx;
    if (count == 0) return YH_OK;
    ARG_TRY(ctx, slots && min_max, "yh_upload_aabb_rows: NULL argument");
    if (s->desc.bounds_kind != YH_BOUNDS_PER_ENTITY) return fail(ctx, YH_ERR_STATE, "yh_upload_aabb_rows: scene has no per-entity bounds");
    for (uint64_t i = 0; i < count; ++i)
        if (slots[i] >= s->n) return fail(ctx, YH_ERR_BAD_ARG, "yh_upload_aabb_rows: slot %u at %llu >= n_entities", slots[i], (unsigned long long)i);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = sync_scene(s)) return rc;
    if (int rc = ensure_staging(s, count * 28)) return rc;
    float* d_mm = (float*)s->staging;
    uint32_t* d_slots = (uint32_t*)(d_mm + count * 6);
    HIP_TRY(ctx, hipMemcpyAsync(d_mm, min_max, count * 24, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_slots, slots, count * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_scatter_aabb, dim3((uint32_t)cdiv(count, YH_BLOCK)), dim3(YH_BLOCK), 0, ctx->stream, s->bounds, d_slots, d_mm, count);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return YH_OK;
}

int ensure_staging(yh_scene* s, uint64_t bytes) {
    if (s->staging_bytes >= bytes) return YH_OK;
    if (s->staging) HIP_TRY(s->ctx, hipFree(s->staging));
    s->staging = nullptr;
    s->staging_bytes = 0;
    HIP_TRY(s->ctx, hipMalloc(&s->staging, bytes));
    s->staging_bytes = bytes;
    return YH_OK;
}

// rows per staging chunk of the AoS68 import/export (68 MB of staging)
#define YH_AOS_CHUNK (1u << 20)

// A few thousand rows between page-locked host memory and the scene: a copy engine's transfer into device staging and the
// kernel behind it are two queues handing over to each other, ~0.15 ms of hand-over for 20 us of data.  The import /
// export kernels read and write their 68-byte rows as coalesced dwords, which is as good across PCIe: when the caller's
// buffers are page-locked and mapped (yh_host_alloc), rows up to this many go through the kernel alone.
#define YH_DIRECT_ROWS_MAX 32768u
static bool device_visible_host(const void* p, const void** dev) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // ordinary memory: not an error, the staged path takes it
        return false;
    }
    if (a.type != hipMemoryTypeHost || !a.devicePointer) return false;
    *dev = a.devicePointer;
    return true;
}

// host_slots (optional): row r belongs to entity host_slots[r] (sparse upload) instead of first + r
static int upload_aos68(yh_scene* s, const void* column, uint64_t first, uint64_t count, float* planes, uint8_t* dirty,
                        const uint32_t* host_slots = nullptr) {
    HIP_TRY(s->ctx, hipSetDevice(s->ctx->device));
    if (int rc = sync_scene(s)) return rc;
    if (host_slots && count && count <= YH_DIRECT_ROWS_MAX && s->direct_rows) {
        const void *drows = nullptr, *dslots = nullptr;
        if (device_visible_host(column, &drows) && device_visible_host(host_slots, &dslots)) {
            hipLaunchKernelGGL(k_aos68_to_soa, dim3(cdiv(count, YH_BLOCK)), dim3(YH_BLOCK), 0, s->ctx->stream, (const uint32_t*)drows,
                               count, planes, dirty, first, (const uint32_t*)dslots);
            HIP_TRY(s->ctx, hipGetLastError());
            HIP_TRY(s->ctx, hipStreamSynchronize(s->ctx->stream));
            return YH_OK;
        }
    }
    const uint64_t chunk_rows = std::min<uint64_t>(count ? count : 1, YH_AOS_CHUNK);
    if (int rc = ensure_staging(s, chunk_rows * 72)) return rc;  // rows, then (sparse) their slots
    uint32_t* d_slots = host_slots ? (uint32_t*)((char*)s->staging + chunk_rows * 68) : nullptr;
    const char* src = (const char*)column;
    for (uint64_t done = 0; done < count; done += YH_AOS_CHUNK) {
        const uint64_t c = std::min<uint64_t>(YH_AOS_CHUNK, count - done);
        HIP_TRY(s->ctx, hipMemcpyAsync(s->staging, src + done * 68, c * 68, hipMemcpyHostToDevice, s->ctx->stream));
        if (host_slots) HIP_TRY(s->ctx, hipMemcpyAsync(d_slots, host_slots + done, c * 4, hipMemcpyHostToDevice, s->ctx->stream));
        hipLaunchKernelGGL(k_aos68_to_soa, dim3(cdiv(c, YH_BLOCK)), dim3(YH_BLOCK), 0, s->ctx->stream,
                           (const uint32_t*)s->staging, c, planes, dirty, first + done, (const uint32_t*)d_slots);
        HIP_TRY(s->ctx, hipGetLastError());
        HIP_TRY(s->ctx, hipStreamSynchronize(s->ctx->stream));
    }
    return YH_OK;
}

// dev_slots (optional): row first + r comes from entity dev_slots[first + r] (a device-resident slot list)
static int download_aos68(yh_scene* s, void* column, uint64_t first, uint64_t count, const float* planes, const uint8_t* dirty,
                          const uint32_t* dev_slots = nullptr) {
    HIP_TRY(s->ctx, hipSetDevice(s->ctx->device));
    if (int rc = sync_scene(s)) return rc;
    if (int rc = ensure_staging(s, (uint64_t)std::min<uint64_t>(count ? count : 1, YH_AOS_CHUNK) * 68)) return rc;
    char* dst = (char*)column;
    for (uint64_t done = 0; done < count; done += YH_AOS_CHUNK) {
        const uint64_t c = std::min<uint64_t>(YH_AOS_CHUNK, count - done);
        hipLaunchKernelGGL(k_soa_to_aos68, dim3(cdiv(c, YH_BLOCK)), dim3(YH_BLOCK), 0, s->ctx->stream, planes, dirty,
                           first + done, c, (uint32_t*)s->staging, dev_slots);
        HIP_TRY(s->ctx, hipGetLastError());
        HIP_TRY(s->ctx, hipMemcpyAsync(dst + done * 68, s->staging, c * 68, hipMemcpyDeviceToHost, s->ctx->stream));
        HIP_TRY(s->ctx, hipStreamSynchronize(s->ctx->stream));
    }
    return YH_OK;
}

extern "C" int yh_upload_local_aos68(yh_scene* s, const void* column, uint64_t first, uint64_t count) {
    if (int rc = check_range(s, "yh_upload_local_aos68", first, count)) return rc;
    ARG_TRY(s->ctx, column, "yh_upload_local_aos68: column is NULL");
    if (s->desc.local_kind != YH_LOCAL_MAT4) return fail(s->ctx, YH_ERR_STATE, "yh_upload_local_aos68: scene holds TRS locals");
    return upload_aos68(s, column, first, count, s->lmat, s->ldirty);
}

extern "C" int yh_upload_world_aos68(yh_scene* s, const void* column, uint64_t first, uint64_t count) {
    if (int rc = check_range(s, "yh_upload_world_aos68", first, count)) return rc;
    ARG_TRY(s->ctx, column, "yh_upload_world_aos68: column is NULL");
    return upload_aos68(s, column, first, count, s->world, s->wdirty);
}

extern "C" int yh_download_world_aos68(yh_scene* s, void* column, uint64_t first, uint64_t count) {
    if (int rc = check_range(s, "yh_download_world_aos68", first, count)) return rc;
    ARG_TRY(s->ctx, column, "yh_download_world_aos68: column is NULL");
    return download_aos68(s, column, first, count, s->world, s->wdirty);
}

extern "C" int yh_download_local_aos68(yh_scene* s, void* column, uint64_t first, uint64_t count) {
    if (int rc = check_range(s, "yh_download_local_aos68", first, count)) return rc;
    ARG_TRY(s->ctx, column, "yh_download_local_aos68: column is NULL");
    if (s->desc.local_kind != YH_LOCAL_MAT4) return fail(s->ctx, YH_ERR_STATE, "yh_download_local_aos68: scene holds TRS locals");
    return download_aos68(s, column, first, count, s->lmat, s->ldirty);
}

// ---- sparse rows and change tracking (only dirty rows cross PCIe) ---------------------------------
static int upload_rows(yh_scene* s, const char* who, const uint32_t* slots, const void* rows, uint64_t count, float* planes,
                       uint8_t* dirty) {
    if (!s) return fail(nullptr, YH_ERR_BAD_ARG, "%s: scene is NULL", who);
    if (count == 0) return YH_OK;
    if (!slots || !rows) return fail(s->ctx, YH_ERR_BAD_ARG, "%s: NULL argument", who);
    if (!planes) return fail(s->ctx, YH_ERR_STATE, "%s: scene holds TRS locals", who);
    for (uint64_t i = 0; i < count; ++i)
        if (slots[i] >= s->n) return fail(s->ctx, YH_ERR_BAD_ARG, "%s: slot %u at %llu >= n_entities", who, slots[i], (unsigned long long)i);
    return upload_aos68(s, rows, 0, count, planes, dirty, slots);
}

extern "C" int yh_upload_local_rows_aos68(yh_scene* s, const uint32_t* slots, const void* rows, uint64_t count) {
    return upload_rows(s, "yh_upload_local_rows_aos68", slots, rows, count, s ? s->lmat : nullptr, s ? s->ldirty : nullptr);
}

extern "C" int yh_upload_world_rows_aos68(yh_scene* s, const uint32_t* slots, const void* rows, uint64_t count) {
    return upload_rows(s, "yh_upload_world_rows_aos68", slots, rows, count, s ? s->world : nullptr, s ? s->wdirty : nullptr);
}

// Rows that change slots ON THE DEVICE (a host whose own copies are older than the scene's: frames ran without readback
// and now its tables are re-ordered): WorldTransform and LocalTransform rows, flags included, of src_slots[i] become
// the rows of dst_slots[i].  Every source row is read before any destination row is written, so the two lists may overlap.
extern "C" int yh_move_rows(yh_scene* s, const uint32_t* src_slots, const uint32_t* dst_slots, uint64_t count) {
    if (!s) return fail(nullptr, YH_ERR_BAD_ARG, "yh_move_rows: scene is NULL");
    yh_ctx* ctx = s->ctx;
    if (s->desc.local_kind != YH_LOCAL_MAT4) return fail(ctx, YH_ERR_STATE, "yh_move_rows: scene holds TRS locals");
    if (count == 0) return YH_OK;
    ARG_TRY(ctx, src_slots && dst_slots, "yh_move_rows: NULL argument");
    ARG_TRY(ctx, count <= s->n, "yh_move_rows: more rows than the scene has slots");
    for (uint64_t i = 0; i < count; ++i)
        if (src_slots[i] >= s->n || dst_slots[i] >= s->n)
            return fail(ctx, YH_ERR_BAD_ARG, "yh_move_rows: pair %llu (%u -> %u) names a slot >= n_entities", (unsigned long long)i,
                        src_slots[i], dst_slots[i]);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = sync_scene(s)) return rc;
    // staging: every row of one component (68 B), then the two slot lists
    if (int rc = ensure_staging(s, count * 76)) return rc;
    uint32_t* d_src = (uint32_t*)((char*)s->staging + count * 68);
    uint32_t* d_dst = d_src + count;
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(d_src, src_slots, count * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d_dst, dst_slots, count * 4, hipMemcpyHostToDevice, st));
    const dim3 grid(cdiv(count, YH_BLOCK));
    for (int k = 0; k < 2; ++k) {
        float* planes = k ? s->lmat : s->world;
        uint8_t* dirty = k ? s->ldirty : s->wdirty;
        hipLaunchKernelGGL(k_soa_to_aos68, grid, dim3(YH_BLOCK), 0, st, (const float*)planes, (const uint8_t*)dirty, (uint64_t)0, count,
                           (uint32_t*)s->staging, (const uint32_t*)d_src);
        hipLaunchKernelGGL(k_aos68_to_soa, grid, dim3(YH_BLOCK), 0, st, (const uint32_t*)s->staging, count, planes, dirty, (uint64_t)0,
                           (const uint32_t*)d_dst);
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, hipStreamSynchronize(st));  // the lists are the caller's
    return YH_OK;
}

extern "C" int yh_scene_track_changes(yh_scene* s, int enable) {
    if (!s) return fail(nullptr, YH_ERR_BAD_ARG, "yh_scene_track_changes: scene is NULL");
    yh_ctx* ctx = s->ctx;
    if (s->desc.local_kind != YH_LOCAL_MAT4)
        return fail(ctx, YH_ERR_STATE, "yh_scene_track_changes: TRS scenes recompute every row each frame");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = sync_scene(s)) return rc;
    if (enable && !s->chg) {
        const uint64_t tile_words = ((uint64_t)s->n_tiles + s->n_segs + 1) & ~1ull;
        // two masks (k_sweep_dirty), laid out as two frusta of a visibility set: [mask][n_tiles | n_chunks | cnt_stride]
        const uint64_t bytes = (32 + 2 * tile_words) * 4 + 2 * (uint64_t)s->n_chunks * 8 + 2 * (uint64_t)s->cnt_stride + 16;
        HIP_TRY(ctx, hipMalloc((void**)&s->chg, bytes));
        HIP_TRY(ctx, hipMemsetAsync(s->chg, 0, bytes, ctx->stream));  // ordered with the frames: not the null stream
        s->chg_tilecnt = s->chg + 32;
        s->chg_words = (uint64_t*)(s->chg_tilecnt + 2 * tile_words);
        s->chg_cnt = (uint8_t*)(((uintptr_t)(s->chg_words + 2 * (uint64_t)s->n_chunks) + 15) & ~(uintptr_t)15);
        HIP_TRY(ctx, hipMalloc((void**)&s->chg_list, s->stride * 4));
        HIP_TRY(ctx, hipMalloc((void**)&s->chg_list_l, s->stride * 4));
        HIP_TRY(ctx, hipHostMalloc((void**)&s->h_chg, 2 * YH_MAX_FRUSTA * 4, hipHostMallocMapped));
        HIP_TRY(ctx, hipHostGetDevicePointer((void**)&s->h_chg_dev, s->h_chg, 0));
        memset(s->h_chg, 0, 2 * YH_MAX_FRUSTA * 4);
    }
    if (enable < 0 || enable > YH_TRACK_REWRITTEN_LOCALS) return fail(ctx, YH_ERR_BAD_ARG, "yh_scene_track_changes: enable must be 0, 1 or 2");
    s->track = enable != 0;
    s->track_rewritten_only = enable == YH_TRACK_REWRITTEN_LOCALS;
    s->chg_valid = false;
    return YH_OK;
}

extern "C" int yh_get_changed_count(yh_scene* s, uint64_t* n_changed) {
    if (!s || !n_changed) return fail(s ? s->ctx : nullptr, YH_ERR_BAD_ARG, "yh_get_changed_count: NULL argument");
    if (!s->track || !s->chg_valid) return fail(s->ctx, YH_ERR_STATE, "yh_get_changed_count: no tracked frame has run");
    if (s->frame_pending)
        if (int rc = sync_scene(s)) return rc;
    *n_changed = s->h_chg[0];
    return YH_OK;
}

extern "C" int yh_download_changed_aos68(yh_scene* s, uint32_t* slots, void* world_rows, void* local_rows, uint64_t cap,
                                         uint64_t* written) {
    if (!s || !written) return fail(s ? s->ctx : nullptr, YH_ERR_BAD_ARG, "yh_download_changed_aos68: NULL argument");
    uint64_t n = 0;
    if (int rc = yh_get_changed_count(s, &n)) return rc;
    *written = n;
    if (n > cap) return fail(s->ctx, YH_ERR_CAPACITY, "yh_download_changed_aos68: %llu rows changed, room for %llu",
                             (unsigned long long)n, (unsigned long long)cap);
    if (n == 0) return YH_OK;
    HIP_TRY(s->ctx, hipSetDevice(s->ctx->device));
    if (slots) HIP_TRY(s->ctx, hipMemcpy(slots, s->chg_list, n * 4, hipMemcpyDeviceToHost));
    // flags were cleared by the frame's sweep: rows come back with dirty = 0, as PostTransformSystem leaves them
    if (world_rows)
        if (int rc = download_aos68(s, world_rows, 0, n, s->world, s->wdirty, s->chg_list)) return rc;
    if (local_rows)
        if (int rc = download_aos68(s, local_rows, 0, n, s->lmat, s->ldirty, s->chg_list)) return rc;
    return YH_OK;
}

extern "C" int yh_get_changed_counts(yh_scene* s, uint64_t* n_world, uint64_t* n_local) {
    if (!s || !n_world || !n_local) return fail(s ? s->ctx : nullptr, YH_ERR_BAD_ARG, "yh_get_changed_counts: NULL argument");
    if (!s->track || !s->chg_valid) return fail(s->ctx, YH_ERR_STATE, "yh_get_changed_counts: no tracked frame has run");
    if (s->frame_pending)
        if (int rc = sync_scene(s)) return rc;
    *n_world = s->h_chg[0];
    *n_local = s->h_chg[1];
    return YH_OK;
}

// One column of the changed rows, STREAMED: one kernel gathers every listed row into 68-byte rows on the device, the
// rows then cross PCIe as YH_DL_PIECES-or-fewer back-to-back copies (nothing between them on the stream: the link
// never idles), and `ready` hears of each piece as its event completes - the caller copies piece i into its tables
// while piece i + 1 is in flight.
extern "C" int yh_download_changed_rows_aos68(yh_scene* s, int which, uint32_t* slots, void* rows, uint64_t cap,
                                              uint64_t* written, yh_rows_ready_fn ready, void* user) {
    if (!s || !written) return fail(s ? s->ctx : nullptr, YH_ERR_BAD_ARG, "yh_download_changed_rows_aos68: NULL argument");
    yh_ctx* ctx = s->ctx;
    ARG_TRY(ctx, which == 0 || which == 1, "yh_download_changed_rows_aos68: which must be 0 (WorldTransform) or 1 (LocalTransform)");
    uint64_t nw = 0, nl = 0;
    if (int rc = yh_get_changed_counts(s, &nw, &nl)) return rc;
    const uint64_t n = which ? nl : nw;
    *written = n;
    if (n > cap) return fail(ctx, YH_ERR_CAPACITY, "yh_download_changed_rows_aos68: %llu rows changed, room for %llu",
                             (unsigned long long)n, (unsigned long long)cap);
    if (n == 0) return YH_OK;
    ARG_TRY(ctx, rows, "yh_download_changed_rows_aos68: rows is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = sync_scene(s)) return rc;
    if (!s->piece_ev_made) {
        for (hipEvent_t& e : s->piece_ev) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        s->piece_ev_made = true;
    }
    const uint32_t* list = which ? s->chg_list_l : s->chg_list;
    const float* planes = which ? s->lmat : s->world;
    const uint8_t* dirty = which ? s->ldirty : s->wdirty;  // cleared by the sweep: rows come back with dirty = 0
    hipStream_t st = ctx->stream;
    if (n <= YH_DIRECT_ROWS_MAX && s->direct_rows) {  // a few rows into page-locked memory: the export kernel writes them there itself
        const void *drows = nullptr, *dslots = nullptr;
        if (device_visible_host(rows, &drows) && (!slots || device_visible_host(slots, &dslots))) {
            if (slots) hipLaunchKernelGGL(k_copy_u32, dim3(cdiv(n, YH_BLOCK)), dim3(YH_BLOCK), 0, st, list, (uint32_t*)dslots, n);
            hipLaunchKernelGGL(k_soa_to_aos68, dim3(cdiv(n, YH_BLOCK)), dim3(YH_BLOCK), 0, st, planes, dirty, (uint64_t)0, n,
                               (uint32_t*)drows, list);
            HIP_TRY(ctx, hipGetLastError());
            HIP_TRY(ctx, hipStreamSynchronize(st));
            if (ready) ready(user, n);
            return YH_OK;
        }
    }
    if (slots) HIP_TRY(ctx, hipMemcpyAsync(slots, list, n * 4, hipMemcpyDeviceToHost, st));
    // device staging for at most YH_DL_ROWS_MAX rows at a time (a 256M-row column does not get a 17 GB twin)
    if (int rc = ensure_staging(s, std::min<uint64_t>(n, YH_DL_ROWS_MAX) * 68)) return rc;
    char* dst = (char*)rows;
    for (uint64_t base = 0; base < n; base += YH_DL_ROWS_MAX) {
        const uint64_t c = std::min<uint64_t>(YH_DL_ROWS_MAX, n - base);
        hipLaunchKernelGGL(k_soa_to_aos68, dim3(cdiv(c, YH_BLOCK)), dim3(YH_BLOCK), 0, st, planes, dirty, base, c,
                           (uint32_t*)s->staging, list);
        HIP_TRY(ctx, hipGetLastError());
        const uint64_t piece = std::max<uint64_t>(YH_DL_PIECE_ROWS_MIN, cdiv(c, YH_DL_PIECES));
        uint32_t pieces = 0;
        for (uint64_t r = 0; r < c; r += piece, ++pieces) {
            const uint64_t pc = std::min<uint64_t>(piece, c - r);
            HIP_TRY(ctx, hipMemcpyAsync(dst + (base + r) * 68, (const char*)s->staging + r * 68, pc * 68, hipMemcpyDeviceToHost, st));
            HIP_TRY(ctx, hipEventRecord(s->piece_ev[pieces], st));
        }
        for (uint32_t i = 0; i < pieces; ++i) {
            HIP_TRY(ctx, hipEventSynchronize(s->piece_ev[i]));
            if (ready) ready(user, std::min<uint64_t>(n, base + (uint64_t)(i + 1) * piece));
        }
    }
    return YH_OK;
}

extern "C" int yh_download_world_soa(yh_scene* s, float* const planes[16], uint64_t first, uint64_t count) {
    if (int rc = check_range(s, "yh_download_world_soa", first, count)) return rc;
    ARG_TRY(s->ctx, planes, "yh_download_world_soa: planes is NULL");
    return copy_planes_d2h(s, s->world, 16, 0, planes, 16, first, count);
}

extern "C" int yh_mark_dirty(yh_scene* s, int local, int world, uint64_t first, uint64_t count) {
    if (int rc = check_range(s, "yh_mark_dirty", first, count)) return rc;
    HIP_TRY(s->ctx, hipSetDevice(s->ctx->device));
    if (count == 0) return YH_OK;
    if (local) hipLaunchKernelGGL(k_fill_u8, dim3(cdiv(count, YH_BLOCK)), dim3(YH_BLOCK), 0, s->ctx->stream, s->ldirty, (uint8_t)1, first, count);
    if (world) hipLaunchKernelGGL(k_fill_u8, dim3(cdiv(count, YH_BLOCK)), dim3(YH_BLOCK), 0, s->ctx->stream, s->wdirty, (uint8_t)1, first, count);
    HIP_TRY(s->ctx, hipGetLastError());
    return YH_OK;
}

extern "C" int yh_compose(yh_scene* s, float* const planes[16], uint64_t first, uint64_t count) {
    if (int rc = check_range(s, "yh_compose", first, count)) return rc;
    ARG_TRY(s->ctx, planes, "yh_compose: planes is NULL");
    if (s->desc.local_kind != YH_LOCAL_TRS) return fail(s->ctx, YH_ERR_STATE, "yh_compose: scene holds Mat4 locals");
    if (count == 0) return YH_OK;
    HIP_TRY(s->ctx, hipSetDevice(s->ctx->device));
    if (int rc = sync_scene(s)) return rc;
    const uint64_t chunk = std::min<uint64_t>(count, YH_AOS_CHUNK);
    if (int rc = ensure_staging(s, chunk * 64)) return rc;
    for (uint64_t done = 0; done < count; done += chunk) {
        const uint64_t c = std::min<uint64_t>(chunk, count - done);
        hipLaunchKernelGGL(k_compose, dim3(cdiv(c, YH_BLOCK)), dim3(YH_BLOCK), 0, s->ctx->stream, s->trs, (float*)s->staging,
                           chunk, first + done, c);
        HIP_TRY(s->ctx, hipGetLastError());
        HIP_TRY(s->ctx, hipStreamSynchronize(s->ctx->stream));
        for (int p = 0; p < 16; ++p) {
            ARG_TRY(s->ctx, planes[p], "yh_compose: plane pointer is NULL");
            HIP_TRY(s->ctx, hipMemcpy(planes[p] + done, (float*)s->staging + (uint64_t)p * chunk, c * 4, hipMemcpyDeviceToHost));
        }
    }
    return YH_OK;
}

// batched Mat4.decompose of the world (which = 0) or local (which = 1, Mat4 scenes) matrices.
extern "C" int yh_decompose(yh_scene* s, int which, float* const planes[10], uint8_t* valid, uint64_t first, uint64_t count) {
    if (int rc = check_range(s, "yh_decompose", first, count)) return rc;
    ARG_TRY(s->ctx, planes, "yh_decompose: planes is NULL");
    ARG_TRY(s->ctx, which == 0 || which == 1, "yh_decompose: which must be 0 (world) or 1 (local)");
    const float* arr = which == 0 ? s->world : s->lmat;
    if (!arr) return fail(s->ctx, YH_ERR_STATE, "yh_decompose: scene holds TRS locals");
    for (int p = 0; p < 10; ++p) ARG_TRY(s->ctx, planes[p], "yh_decompose: plane pointer is NULL");
    if (count == 0) return YH_OK;
    HIP_TRY(s->ctx, hipSetDevice(s->ctx->device));
    if (int rc = sync_scene(s)) return rc;
    const uint64_t chunk = std::min<uint64_t>(count, YH_AOS_CHUNK);
    if (int rc = ensure_staging(s, chunk * 41)) return rc;  // 10 f32 planes + the validity bytes
    uint8_t* d_valid = (uint8_t*)s->staging + chunk * 40;
    for (uint64_t done = 0; done < count; done += chunk) {
        const uint64_t c = std::min<uint64_t>(chunk, count - done);
        hipLaunchKernelGGL(k_decompose, dim3(cdiv(c, YH_BLOCK)), dim3(YH_BLOCK), 0, s->ctx->stream, arr, (float*)s->staging,
                           d_valid, chunk, first + done, c);
        HIP_TRY(s->ctx, hipGetLastError());
        HIP_TRY(s->ctx, hipStreamSynchronize(s->ctx->stream));
        for (int p = 0; p < 10; ++p)
            HIP_TRY(s->ctx, hipMemcpy(planes[p] + done, (float*)s->staging + (uint64_t)p * chunk, c * 4, hipMemcpyDeviceToHost));
        if (valid) HIP_TRY(s->ctx, hipMemcpy(valid + done, d_valid, c, hipMemcpyDeviceToHost));
    }
    return YH_OK;
}

// Measurement harness: a plain read + write stream over a working set of the given size (bench.py's cache ceiling).
extern "C" int yh_stream_probe(yh_ctx* ctx, uint64_t read_bytes, uint64_t write_bytes, uint32_t launches, float* mean_ms) {
    if (!ctx || !mean_ms || !launches) return fail(ctx, YH_ERR_BAD_ARG, "yh_stream_probe: NULL argument or no launches");
    ARG_TRY(ctx, read_bytes + write_bytes >= 16 && read_bytes <= (64ull << 30) && write_bytes <= (64ull << 30),
            "yh_stream_probe: sizes must be 16 bytes .. 64 GiB");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint64_t nr = read_bytes / 16, nw = write_bytes / 16, top = nr > nw ? nr : nw;
    float4 *src = nullptr, *dst = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipSuccess;
    if (nr) e = hipMalloc((void**)&src, nr * 16);
    if (e == hipSuccess && nw) e = hipMalloc((void**)&dst, nw * 16);
    if (e == hipSuccess && nr) e = hipMemsetAsync(src, 0, nr * 16, ctx->stream);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    // every thread takes at most 4 steps: enough workgroups to fill the chip many times over at any size
    const uint32_t grid = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(cdiv(top, YH_BLOCK * 4ull), 1), 1u << 20);
    for (uint32_t i = 0; e == hipSuccess && i < 3 + launches; ++i) {
        if (i == 3) e = hipEventRecord(e0, ctx->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_stream_probe, dim3(grid), dim3(YH_BLOCK), 0, ctx->stream, src, dst, nr, nw);
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.0f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (src) (void)hipFree(src);
    if (dst) (void)hipFree(dst);
    if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? YH_ERR_OOM : YH_ERR_HIP, "yh_stream_probe: %s", hipGetErrorString(e));
    *mean_ms = ms / (float)launches;
    return YH_OK;
}

// ------------------------------------------------------------------------------------ synthetic scenes
static int fill_synthetic(yh_scene* s, const yh_synth_desc* d, const uint32_t* global_of_local);

extern "C" int yh_scene_fill_synthetic(yh_scene* s, const yh_synth_desc* d) { return fill_synthetic(s, d, nullptr); }

extern "C" int yh_scene_fill_synthetic_shard(yh_scene* s, const yh_synth_desc* d, const uint32_t* global_of_local) {
    if (!global_of_local) return fail(s ? s->ctx : nullptr, YH_ERR_BAD_ARG, "yh_scene_fill_synthetic_shard: global_of_local is NULL");
    return fill_synthetic(s, d, global_of_local);
}

static int fill_synthetic(yh_scene* s, const yh_synth_desc* d, const uint32_t* global_of_local) {
    if (!s || !d) return fail(s ? s->ctx : nullptr, YH_ERR_BAD_ARG, "yh_scene_fill_synthetic: NULL argument");
    if (!s->have_hierarchy) return fail(s->ctx, YH_ERR_STATE, "yh_scene_fill_synthetic: upload the hierarchy first");
    if (s->desc.local_kind != YH_LOCAL_TRS) return fail(s->ctx, YH_ERR_STATE, "yh_scene_fill_synthetic: TRS scenes only");
    HIP_TRY(s->ctx, hipSetDevice(s->ctx->device));
    const bool table = s->desc.bounds_kind == YH_BOUNDS_MESH_TABLE;
    if (table) {
        const uint32_t nm = s->desc.max_meshes;
        std::vector<float> t((size_t)nm * 6);
        for (uint32_t m = 0; m < nm; ++m)
            for (int c = 0; c < 3; ++c) {
                const float h = yh_synth_uniform(d->seed, 16u + (uint32_t)c, m, 0.25f, 2.0f);
                t[m * 6 + c] = -h;
                t[m * 6 + 3 + c] = h;
            }
        if (int rc = yh_upload_mesh_aabbs(s, t.data(), nm)) return rc;
    }
    uint32_t* d_map = nullptr;
    if (global_of_local) {  // the shard's global slots, through the staging buffer
        if (int rc = sync_scene(s)) return rc;
        if (int rc = ensure_staging(s, s->n * 4)) return rc;
        d_map = (uint32_t*)s->staging;
        HIP_TRY(s->ctx, hipMemcpyAsync(d_map, global_of_local, s->n * 4, hipMemcpyHostToDevice, s->ctx->stream));
    }
    hipLaunchKernelGGL(k_synth, dim3(cdiv(s->n, YH_BLOCK)), dim3(YH_BLOCK), 0, s->ctx->stream, s->trs,
                       s->desc.bounds_kind == YH_BOUNDS_PER_ENTITY ? s->bounds : (float*)nullptr,
                       table ? s->mesh_id : (uint32_t*)nullptr, table ? s->n_meshes : 1u, s->n,
                       (uint64_t)s->level_offsets[1], d->seed, d->index_base, (const uint32_t*)d_map);
    HIP_TRY(s->ctx, hipGetLastError());
    if (table) s->max_mesh_id = s->n_meshes - 1;  // k_synth draws ids from the whole table
    return sync_scene(s);
}

