// yume_hip.hip — C ABI (include/yume_hip.h) over the gfx950 kernels in yh_kernels.hpp.
//
// Host side of the drop-in boundary: owns the device buffers of a scene, cuts the depth-sorted hierarchy
// into level groups, enqueues one fused kernel per group on the context's stream, and exposes results.
// One process per GPU; RCCL (dlopen'ed, only when yh_comm_init is called) gathers draw lists across ranks.
#include "../../include/yume_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <new>
#include <string>
#include <vector>

#include "yh_kernels.hpp"

static_assert(YH_MAX_FRUSTA == YH_K_MAX_FRUSTA, "header / kernel frustum limits differ");

// yh_download_changed_rows_aos68: rows gathered per kernel, copies per kernel, smallest copy worth an event
#define YH_DL_ROWS_MAX (4ull << 20)
// Boundary counters: frame k adds to set k % YH_BSETS; its k_expand reports the set and zeroes set k + 2 (the frame in
// between may already be adding to set k + 1 when a deferred expansion runs).  Frames that leave their lists unexpanded
// (YH_FRAME_LISTS_ON_DEMAND) zero YH_BZERO_AHEAD sets ahead at a time instead, one small launch per that many frames.
#define YH_BSETS 64u
#define YH_BZERO_AHEAD 32u
#define YH_DL_PIECES 16
#define YH_DL_PIECE_ROWS_MIN 32768ull

// ------------------------------------------------------------------------------------------------ state
struct yh_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    std::vector<struct yh_scene*> scenes;  // live scenes of this context (yh_wait flushes their deferred expansions)
    hipStream_t stream_c = nullptr;  // collectives of the no-host-sync gather (created by yh_comm_init)
    hipDeviceProp_t prop{};
    unsigned char* scratch_ptr = nullptr;  // this device's copy of the file-scope `scratch` array (yh_frame_static)
    std::string err;
    // RCCL (optional)
    void* rccl_lib = nullptr;
    void* comm = nullptr;
    int n_ranks = 1, rank = 0;
};

struct LevelGroup {
    uint32_t lo, hi;  // levels [lo, hi] of one launch: k_frame_trs if one level, else k_frame_chain
};

#define YH_TIMING_RING 2048
/* Entities from which a frame's inputs no longer stay in the 256 MB Infinity Cache from one frame to the next: larger
   scenes stream their read-once inputs (measured crossovers: trees between 3M and 4M, flat scenes between 4M and 6M). */
#define YH_BIG_TREE (7u << 19)
#define YH_BIG_FLAT (5u << 20)
#define YH_XCD_RUN 8u /* consecutive tiles one XCD takes in a tree launch */
#define YH_MASK_SETS 2u /* sets of visibility words frames rotate through while a gather is in flight */
#define YH_MASK_HALF 1u /* the frames' stream checks on the gather every YH_MASK_HALF-th frame (8 / 4 measured the same) */

struct yh_scene {
    yh_ctx* ctx = nullptr;
    yh_scene_desc desc{};
    uint64_t n = 0, stride = 0, draw_cap = 0;
    float* trs = nullptr;
    float* lmat = nullptr;
    float* world = nullptr;
    uint32_t* parent = nullptr;
    float* bounds = nullptr;  // 6 planes, or mesh table
    uint32_t* mesh_id = nullptr;
    uint8_t* ldirty = nullptr;
    uint8_t* wdirty = nullptr;
    uint32_t* draw[YH_MAX_FRUSTA] = {};
    uint32_t* counts = nullptr;       // device: 8 visible + 2 x 8 boundary totals, then tile counts, then visibility words
    uint32_t* tilecnt = nullptr;      // inside the counts allocation: [n_frusta][n_tiles]
    uint64_t* vismask = nullptr;      // inside the counts allocation: [n_frusta][n_chunks]
    uint8_t* viscnt = nullptr;        // inside the counts allocation: [n_frusta][cnt_stride] popcounts of the words
    uint32_t cnt_stride = 0;          // n_chunks rounded up to 16
    uint8_t* chg_cnt = nullptr;       // the same for the change mask
    uint64_t vis_bytes = 0;           // bytes of the counts allocation (zeroed once at creation)
    uint32_t n_chunks = 0, n_tiles = 0, tile_words = 64;
    uint32_t n_segs = 0;              // k_count segments (4096 words each); tile-count rows hold n_tiles + n_segs words
    uint32_t* h_counts = nullptr;     // host-mapped: written by k_expand ([f] totals, [8+f] boundary counts)
    uint32_t* h_counts_dev = nullptr; // device address of h_counts
    uint32_t bset = 0;                // which of the YH_BSETS boundary-counter sets this frame uses
    uint32_t* bsets = nullptr;        // [YH_BSETS][YH_MAX_FRUSTA]
    uint32_t bzero_ahead = 0;         // upcoming frames whose set "two ahead" a bulk k_zero_sets has already zeroed
    uint8_t* viscnt_alt = nullptr;    // byte counts of the other sets of visibility words: [YH_MASK_SETS-1][n_frusta][cnt_stride]
    // Deferred expansion (TRS scenes of at most YH_SELF_COUNT_TILES tiles, YH_FRAME_ASYNC frames): the frame's
    // k_expand is not launched; the first launch of the next frame carries it (k_frame_*_x), or the next
    // synchronising / list-reading call launches it on its own.
    bool exp_pending = false;
    bool pend_demand = false;  // the pending expansion is wanted only if somebody reads the lists (YH_FRAME_LISTS_ON_DEMAND)
    bool pend_count = false;   // ... and its tile bases (k_count) have not been computed either
    ExpandArgs pend;
    uint32_t pend_frusta = 0;
    // change tracking (Mat4 scenes, yh_scene_track_changes): mask of rows a frame touched -> ascending slot list
    bool track = false;
    uint32_t* chg = nullptr;          // device: 32 count words | tile counts | mask words (one allocation)
    uint32_t* chg_tilecnt = nullptr;
    uint64_t* chg_words = nullptr;
    uint64_t* alive = nullptr;        // device [n_chunks]: bit = the slot holds an entity (yh_scene_set_alive_range); NULL: all do
    std::vector<uint64_t> h_alive;    // its host copy
    uint32_t* chg_list = nullptr;     // device: changed slots of the last frame (WorldTransform rows: mask 0)
    uint32_t* chg_list_l = nullptr;   // device: slots whose LocalTransform row changed (mask 1)
    hipEvent_t piece_ev[YH_DL_PIECES] = {};  // yh_download_changed_rows_aos68: one per piece in flight
    bool piece_ev_made = false;
    uint32_t* h_chg = nullptr;        // host-mapped totals written by k_expand
    uint32_t* h_chg_dev = nullptr;
    bool chg_valid = false;           // a tracked frame has run since tracking was enabled
    // draw-list consumer (N2): per frustum, model matrices + draw records of the visible entities, list order
    uint32_t* obj[YH_MAX_FRUSTA] = {};
    uint4* cmd[YH_MAX_FRUSTA] = {};
    uint32_t* vcount = nullptr;       // vertices per mesh (mesh-table scenes) or per entity
    // per-mesh buckets (yh_build_mesh_buckets): bucketed entity ids + bucket table per frustum, shared scratch
    uint32_t* blist[YH_MAX_FRUSTA] = {};
    uint32_t* bkt[YH_MAX_FRUSTA] = {};
    uint32_t* bscratch = nullptr;     // hist [n_meshes][slices] | totals | starts
    uint64_t bscratch_words = 0;
    bool bucketed[YH_MAX_FRUSTA] = {};
    void* staging = nullptr;          // device staging for AoS68 import/export
    uint64_t staging_bytes = 0;
    std::vector<uint32_t> level_offsets;
    std::vector<LevelGroup> groups;
    std::vector<std::string> group_kernel;  // kernel each group's launch of the last frame ran (rocprofv3's name for it)
    uint32_t max_group_levels = 0;  // 0 = auto
    std::vector<LevelGroup> explicit_plan;  // yh_scene_set_plan; empty = planner decides
    bool have_hierarchy = false;
    bool frame_pending = false;
    uint32_t last_n_frusta = 0;
    uint32_t n_meshes = 0;
    uint64_t big_tree = YH_BIG_TREE, big_flat = YH_BIG_FLAT;  // entities from which read-once inputs are streamed (A/B: env of the same names)
    bool lds_parents = false;  // one-level launches stage their parents in LDS (yh_scene_set_lds_parents / YH_LDS_PARENTS=1)
    bool use_cull_plan = true;  // group planes by normal when some share one (YH_NO_CULL_PLAN=1 in the environment: A/B runs)
    uint32_t defer_first_max = 1u << 20;  // entities in a first launch that still carries the previous frame's expansion
    uint32_t max_mesh_id = 0;  // largest id ever uploaded (ids start out as 0): the table may not shrink below it
    float root[16];
    uint32_t root_dirty = 1;  // world.set(scene_root, WorldTransform{IDENTITY}) -> dirty (GameApp.zig:95)
    // timing
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    uint64_t frames = 0;
    uint32_t last_launches = 0;
    bool kernel_timing = false;
    int timing_only = -1;  // >= 0: only this launch index gets events (and no frame events); -2: the transform launches
    uint32_t timing_period = 1;  // events on every timing_period-th frame only
    bool frame_timed = false;
    std::vector<hipEvent_t> kev;  // YH_TIMING_RING pairs, lazily created
    std::vector<uint32_t> kev_group;
    uint32_t kev_used = 0;
    // gathered lists (multi-GPU)
    uint32_t* d_all_counts = nullptr;
    uint32_t* h_all_counts = nullptr;
    uint32_t* gathered[YH_MAX_FRUSTA] = {};
    uint64_t gathered_cap[YH_MAX_FRUSTA] = {};
    uint64_t gathered_len[YH_MAX_FRUSTA] = {};
    // no-host-sync gather (yh_allgather_draw_lists_async): the visibility words travel, every rank expands them
    bool g_ready = false;
    int g_ranks = 0;                  // ranks the buffers below were sized for
    int counts_cap = 0;               // ranks d_all_counts / h_all_counts have room for
    int counts_ranks = 0;             // ranks the host counts currently describe
    uint32_t g_words = 0, g_tiles = 0, g_frusta = 0, g_tile_words = 64;  // longest shard's words per frustum row, its tiles
    uint32_t* d_peer = nullptr;       // [R][4] n_chunks, id_base, frusta, - of every rank (+ 4 words: this rank's own)
    std::vector<uint32_t> h_peer;
    uint64_t* gmask = nullptr;        // [R][F][g_words] received words
    uint32_t* h_all_counts_dev = nullptr;  // device address of the host-mapped totals
    uint64_t* send_pad = nullptr;     // [F][g_words] zero-padded copy of this rank's words (only if n_chunks < g_words)
    uint32_t* gtile = nullptr;        // [F][R][g_tiles]
    // Ring of YH_MASK_SETS sets of visibility words (set 0 = vismask, the others in mask_alt): consecutive culling
    // frames write consecutive sets, so a frame's words stay intact while the collective stream sends them; before a
    // set is rewritten the frames' stream waits for the all-gather that last sent it.
    uint64_t* mask_alt = nullptr;
    uint32_t mparity = 0;             // set the next culling frame writes
    uint64_t mask_frame[YH_MASK_SETS] = {};  // frame number of the gather that last sent set p
    uint64_t* last_mask = nullptr;    // the words the last culling frame wrote
    uint32_t last_mparity = 0;
    hipEvent_t ev_frame = nullptr, ev_mask[YH_MASK_SETS] = {};
    bool mask_busy[YH_MASK_SETS] = {};  // ev_mask[p] must have passed before a frame writes word set p again
    bool gathered_async = false;  // h_all_counts holds running totals to be turned into counts after a sync
};

static int flush_expand(yh_scene* s);

// Stage-3 scratch of yh_frame_static (the entry point that allocates nothing): boundary counters, tile counts,
// visibility words and their byte counts of ONE frame.  Every GPU the library runs on holds its own copy of the array
// at its own address until the process ends; `scratch` itself is only a host-side stand-in for the symbol, so host
// code never passes it anywhere: yh_init looks the device address up once per context (hipGetSymbolAddress, with
// that device current) and everything afterwards uses yh_ctx::scratch_ptr.
static __device__ __align__(256) unsigned char scratch[YH_STATIC_SCRATCH_BYTES];

// Who is using a GPU's copy of `scratch`: the context whose yh_frame_static has not been waited for yet (NULL: nobody).
// Two contexts of one GPU share the array; a second one that would write it while the first one's frame may still be
// running is turned away (YH_ERR_STATE) instead of corrupting both frames' lists.  Claimed by yh_frame_static, released
// when the owner's stream has been synchronised through the library (yh_wait, any yh_get_ / yh_download_ call, a frame
// without YH_FRAME_ASYNC) or the context is shut down.  Work the caller replays itself (a captured graph launched on
// yh_stream) is beyond the library's sight: its ordering stays the caller's business, as the header says.
#define YH_MAX_DEVICES 64
static std::atomic<yh_ctx*> g_static_owner[YH_MAX_DEVICES];
static void release_static(yh_ctx* ctx) {
    yh_ctx* me = ctx;
    if (ctx && ctx->device >= 0) g_static_owner[ctx->device % YH_MAX_DEVICES].compare_exchange_strong(me, nullptr);
}

static thread_local std::string g_err;

static int fail(yh_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    if (ctx) ctx->err = buf;
    return code;
}

#define HIP_TRY(ctx, call)                                                                                    \
    do {                                                                                                      \
        hipError_t e_ = (call);                                                                               \
        if (e_ != hipSuccess)                                                                                 \
            return fail((ctx), e_ == hipErrorOutOfMemory ? YH_ERR_OOM : YH_ERR_HIP, "%s failed: %s (%s:%d)", \
                        #call, hipGetErrorString(e_), __FILE__, __LINE__);                                    \
    } while (0)

#define ARG_TRY(ctx, cond, msg)                                   \
    do {                                                          \
        if (!(cond)) return fail((ctx), YH_ERR_BAD_ARG, "%s", msg); \
    } while (0)

static inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

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
    e = hipGetSymbolAddress((void**)&c->scratch_ptr, HIP_SYMBOL(scratch));  // device `device_id` is current
    if (e == hipSuccess && !c->scratch_ptr) e = hipErrorInvalidSymbol;
    if (e != hipSuccess) {
        for (hipStream_t st : {c->stream_c, c->stream}) (void)hipStreamDestroy(st);
        delete c;
        return fail(nullptr, YH_ERR_HIP, "yh_init: the static scratch array could not be resolved: %s", hipGetErrorString(e));
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
    if (const char* v = getenv("YH_BIG_TREE")) s->big_tree = strtoull(v, nullptr, 10);
    if (const char* v = getenv("YH_BIG_FLAT")) s->big_flat = strtoull(v, nullptr, 10);
    if (const char* v = getenv("YH_DEFER_FIRST_MAX")) s->defer_first_max = (uint32_t)strtoul(v, nullptr, 10);
    ctx->scenes.push_back(s);
    *out = s;
    return YH_OK;
}

static int check_range(yh_scene* s, const char* who, uint64_t first, uint64_t count) {
    if (!s) return fail(nullptr, YH_ERR_BAD_ARG, "%s: scene is NULL", who);
    if (first > s->n || count > s->n - first)
        return fail(s->ctx, YH_ERR_BAD_ARG, "%s: range [%llu, +%llu) exceeds %llu entities", who,
                    (unsigned long long)first, (unsigned long long)count, (unsigned long long)s->n);
    return YH_OK;
}

// Launches the expansion a YH_FRAME_ASYNC frame left for the next frame's first launch to carry (stream order keeps
// it behind that frame's kernels).  Called by everything that reads lists or totals, or synchronises.
static int flush_expand(yh_scene* s) {
    if (!s->exp_pending) return YH_OK;
    s->exp_pending = false;
    s->pend_demand = false;
    if (s->pend_count) {
        s->pend_count = false;
        hipLaunchKernelGGL(k_count, dim3(s->n_segs, s->pend_frusta), dim3(YH_BLOCK), 0, s->ctx->stream, s->pend);
    }
    hipLaunchKernelGGL(k_expand, dim3(s->pend.n_tiles, s->pend_frusta), dim3(YH_BLOCK), 0, s->ctx->stream, s->pend);
    HIP_TRY(s->ctx, hipGetLastError());
    return YH_OK;
}

static int sync_scene(yh_scene* s) {
    if (int rc = flush_expand(s)) return rc;
    HIP_TRY(s->ctx, hipStreamSynchronize(s->ctx->stream));
    if (s->ctx->stream_c) HIP_TRY(s->ctx, hipStreamSynchronize(s->ctx->stream_c));
    s->frame_pending = false;
    release_static(s->ctx);
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

static int ensure_staging(yh_scene* s, uint64_t bytes);

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

static int ensure_staging(yh_scene* s, uint64_t bytes) {
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

// host_slots (optional): row r belongs to entity host_slots[r] (sparse upload) instead of first + r
static int upload_aos68(yh_scene* s, const void* column, uint64_t first, uint64_t count, float* planes, uint8_t* dirty,
                        const uint32_t* host_slots = nullptr) {
    HIP_TRY(s->ctx, hipSetDevice(s->ctx->device));
    if (int rc = sync_scene(s)) return rc;
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
    s->track = enable != 0;
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

// ------------------------------------------------------------------------------------------------ frame
// A pair of HIP events handed to the launch itself (hipExtLaunchKernelGGL) receives the dispatch's own start and end
// timestamps — the ones rocprofv3's kernel trace reports — and puts no extra packet into the stream; events recorded
// around a launch read 3-5 us more and stall the stream by as much.
struct LaunchEvents {
    hipEvent_t start = nullptr, stop = nullptr;
};
#define YH_LAUNCH(kernel, grid, st, ev, ...)                                                                   \
    do {                                                                                                       \
        if ((ev).start) hipExtLaunchKernelGGL(kernel, grid, dim3(YH_BLOCK), 0, st, (ev).start, (ev).stop, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, grid, dim3(YH_BLOCK), 0, st, __VA_ARGS__);                             \
    } while (0)

// Every launch helper also reports the kernel it picked, spelled as rocprofv3's kernel trace spells it: the
// measurement harness looks a launch's PMC traffic up by that name (yh_scene_get_group_kernel_names) instead of
// re-deriving these dispatch rules.
static std::string kname(const char* fmt, ...) {
    char buf[160];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return buf;
}
static inline const char* tf(bool b) { return b ? "true" : "false"; }
static inline int chain_g(int G) { return G <= 2 ? 2 : G <= 3 ? 3 : G <= 4 ? 4 : G <= 6 ? 6 : 8; }

template <int BOUNDS>
static std::string launch_trs_lds(dim3 grid, hipStream_t st, LaunchEvents ev, const FrameArgs& a, bool big) {
    if (big) YH_LAUNCH((k_frame_trs_lds<BOUNDS, true>), grid, st, ev, a);
    else YH_LAUNCH((k_frame_trs_lds<BOUNDS, false>), grid, st, ev, a);
    return kname("void k_frame_trs_lds<%d, %s>(FrameArgs)", BOUNDS, tf(big));
}

template <int BOUNDS>
static std::string launch_trs(bool has_base, dim3 grid, hipStream_t st, LaunchEvents ev, const FrameArgs& a, bool big) {
    if (big && has_base) YH_LAUNCH((k_frame_trs_big<BOUNDS, true>), grid, st, ev, a);
    else if (big) YH_LAUNCH((k_frame_trs_big<BOUNDS, false>), grid, st, ev, a);
    else if (has_base) YH_LAUNCH((k_frame_trs<BOUNDS, true>), grid, st, ev, a);
    else YH_LAUNCH((k_frame_trs<BOUNDS, false>), grid, st, ev, a);
    return kname("void k_frame_trs%s<%d, %s>(FrameArgs)", big ? "_big" : "", BOUNDS, tf(has_base));
}

template <int BOUNDS, bool HAS_BASE>
static std::string launch_chain_g(int G, dim3 grid, hipStream_t st, LaunchEvents ev, const FrameArgs& a, bool big) {
    if (big && G <= 2) YH_LAUNCH((k_frame_chain_big<2, BOUNDS, HAS_BASE>), grid, st, ev, a);
    else if (big && G <= 3) YH_LAUNCH((k_frame_chain_big<3, BOUNDS, HAS_BASE>), grid, st, ev, a);
    else if (G <= 2) YH_LAUNCH((k_frame_chain<2, BOUNDS, HAS_BASE>), grid, st, ev, a);
    else if (G <= 3) YH_LAUNCH((k_frame_chain<3, BOUNDS, HAS_BASE>), grid, st, ev, a);
    else if (G <= 4) YH_LAUNCH((k_frame_chain<4, BOUNDS, HAS_BASE>), grid, st, ev, a);
    else if (G <= 6) YH_LAUNCH((k_frame_chain<6, BOUNDS, HAS_BASE>), grid, st, ev, a);
    else YH_LAUNCH((k_frame_chain<8, BOUNDS, HAS_BASE>), grid, st, ev, a);
    return kname("void k_frame_chain%s<%d, %d, %s>(FrameArgs)", big && G <= 3 ? "_big" : "", chain_g(G), BOUNDS, tf(HAS_BASE));
}
template <int BOUNDS>
static std::string launch_chain(int G, bool has_base, dim3 grid, hipStream_t st, LaunchEvents ev, const FrameArgs& a, bool big) {
    if (has_base) return launch_chain_g<BOUNDS, true>(G, grid, st, ev, a, big);
    return launch_chain_g<BOUNDS, false>(G, grid, st, ev, a, big);
}

// First launch of a frame carrying the previous frame's expansion (levels from 0: no stored base parent).
template <int BOUNDS>
static std::string launch_first_x(int G, dim3 grid, hipStream_t st, LaunchEvents ev, const FrameArgs& a, const ExpandArgs& x) {
    if (G <= 1) {
        YH_LAUNCH((k_frame_trs_x<BOUNDS>), grid, st, ev, a, x);
        return kname("void k_frame_trs_x<%d>(FrameArgs, ExpandArgs)", BOUNDS);
    }
    if (G <= 2) YH_LAUNCH((k_frame_chain_x<2, BOUNDS>), grid, st, ev, a, x);
    else if (G <= 3) YH_LAUNCH((k_frame_chain_x<3, BOUNDS>), grid, st, ev, a, x);
    else if (G <= 4) YH_LAUNCH((k_frame_chain_x<4, BOUNDS>), grid, st, ev, a, x);
    else if (G <= 6) YH_LAUNCH((k_frame_chain_x<6, BOUNDS>), grid, st, ev, a, x);
    else YH_LAUNCH((k_frame_chain_x<8, BOUNDS>), grid, st, ev, a, x);
    return kname("void k_frame_chain_x<%d, %d>(FrameArgs, ExpandArgs)", chain_g(G), BOUNDS);
}

template <int BOUNDS, bool HAS_BASE>
static std::string launch_chain_mat4_g(int G, dim3 grid, hipStream_t st, const FrameArgs& a, uint32_t root_dirty) {
    if (G <= 2) hipLaunchKernelGGL((k_chain_mat4<2, BOUNDS, HAS_BASE>), grid, dim3(YH_BLOCK), 0, st, a, root_dirty);
    else if (G <= 3) hipLaunchKernelGGL((k_chain_mat4<3, BOUNDS, HAS_BASE>), grid, dim3(YH_BLOCK), 0, st, a, root_dirty);
    else if (G <= 4) hipLaunchKernelGGL((k_chain_mat4<4, BOUNDS, HAS_BASE>), grid, dim3(YH_BLOCK), 0, st, a, root_dirty);
    else if (G <= 6) hipLaunchKernelGGL((k_chain_mat4<6, BOUNDS, HAS_BASE>), grid, dim3(YH_BLOCK), 0, st, a, root_dirty);
    else hipLaunchKernelGGL((k_chain_mat4<8, BOUNDS, HAS_BASE>), grid, dim3(YH_BLOCK), 0, st, a, root_dirty);
    return kname("void k_chain_mat4<%d, %d, %s>(FrameArgs, unsigned int)", chain_g(G), BOUNDS, tf(HAS_BASE));
}
template <int BOUNDS>
static std::string launch_chain_mat4(int G, bool has_base, dim3 grid, hipStream_t st, const FrameArgs& a, uint32_t root_dirty) {
    if (has_base) return launch_chain_mat4_g<BOUNDS, true>(G, grid, st, a, root_dirty);
    return launch_chain_mat4_g<BOUNDS, false>(G, grid, st, a, root_dirty);
}

template <int BOUNDS>
static std::string launch_mat4(bool has_base, bool big, dim3 grid, hipStream_t st, const FrameArgs& a, uint32_t root_dirty) {
    if (has_base && big) hipLaunchKernelGGL((k_level_mat4<BOUNDS, true, true>), grid, dim3(YH_BLOCK), 0, st, a, root_dirty);
    else if (has_base) hipLaunchKernelGGL((k_level_mat4<BOUNDS, true, false>), grid, dim3(YH_BLOCK), 0, st, a, root_dirty);
    else if (big) hipLaunchKernelGGL((k_level_mat4<BOUNDS, false, true>), grid, dim3(YH_BLOCK), 0, st, a, root_dirty);
    else hipLaunchKernelGGL((k_level_mat4<BOUNDS, false, false>), grid, dim3(YH_BLOCK), 0, st, a, root_dirty);
    return kname("void k_level_mat4<%d, %s, %s>(FrameArgs, unsigned int)", BOUNDS, tf(has_base), tf(big));
}

static int ensure_kev(yh_scene* s) {
    if (!s->kev.empty()) return YH_OK;
    s->kev.assign(2 * YH_TIMING_RING, nullptr);
    s->kev_group.assign(YH_TIMING_RING, 0);
    for (auto& e : s->kev) HIP_TRY(s->ctx, hipEventCreate(&e));
    return YH_OK;
}

// Groups a frame's planes by normal for cull_planned (yh_math.hpp): two planes share a group when their normals are
// equal component by component, or exactly opposite (== on floats: +0 and -0 are the same normal, a NaN component
// makes a plane a group of its own).  Left alone (n_normals = 0, the kernels test plane by plane) when no two planes
// share a normal: the plan saves nothing then.
static void build_cull_plan(const float* planes, uint32_t n_frusta, CullPlan& plan) {
    memset(&plan, 0, sizeof plan);
    const uint32_t P = n_frusta * 6;
    if (P > YH_PLAN_MAX_PLANES) return;
    int group_of[YH_PLAN_MAX_PLANES];
    float sgn_of[YH_PLAN_MAX_PLANES];
    uint32_t rep[YH_PLAN_MAX_PLANES], n_groups = 0;
    for (uint32_t i = 0; i < P; ++i) {
        const float* n = planes + 4 * i;
        group_of[i] = -1;
        for (uint32_t g = 0; g < n_groups && group_of[i] < 0; ++g) {
            const float* m = planes + 4 * rep[g];
            if (n[0] == m[0] && n[1] == m[1] && n[2] == m[2]) {
                group_of[i] = (int)g;
                sgn_of[i] = 1.0f;
            } else if (n[0] == -m[0] && n[1] == -m[1] && n[2] == -m[2]) {
                group_of[i] = (int)g;
                sgn_of[i] = -1.0f;
            }
        }
        if (group_of[i] < 0) {
            rep[n_groups] = i;
            group_of[i] = (int)n_groups++;
            sgn_of[i] = 1.0f;
        }
    }
    if (n_groups == P) return;
    uint32_t k = 0;
    for (uint32_t g = 0; g < n_groups; ++g) {
        const float* m = planes + 4 * rep[g];
        plan.normal[g].a = m[0];
        plan.normal[g].b = m[1];
        plan.normal[g].c = m[2];
        uint32_t mid = 0;
        for (float sgn : {1.0f, -1.0f}) {  // the planes on the normal itself first, then those on its negation
            for (uint32_t i = 0; i < P; ++i)
                if (group_of[i] == (int)g && sgn_of[i] == sgn) {
                    plan.plane[k].d = planes[4 * i + 3];
                    plan.plane[k].frustum = i / 6;
                    ++k;
                }
            if (sgn > 0) mid = k;
        }
        plan.normal[g].mid_end = mid | (k << 16);
    }
    plan.n_normals = n_groups;
}


// Where the stage-3 scratch of one frame lies inside the static array (yh_frame_static): 32 counter words (only
// [8, 24) are used: the boundary counters and the set k_expand zeroes), tile counts, visibility words, byte counts.
struct StaticLayout {
    uint64_t tilecnt_off, words_off, cnt_off, bytes;
};
static StaticLayout static_layout(const yh_scene* s, uint32_t n_frusta) {
    const uint64_t F = n_frusta ? n_frusta : 1;
    StaticLayout l;
    l.tilecnt_off = 4 * YH_MAX_FRUSTA * 4;
    l.words_off = l.tilecnt_off + ((F * (s->n_tiles + s->n_segs) + 1) & ~1ull) * 4;
    l.cnt_off = (l.words_off + F * s->n_chunks * 8 + 15) & ~15ull;
    l.bytes = l.cnt_off + F * s->cnt_stride;
    return l;
}

// One frame.  use_static: the stage-3 scratch lies in the context's copy of the static array instead of the scene's
// own buffers, nothing is deferred to a later call, and no event or buffer is created (yh_frame_static).
static int frame_impl(yh_scene* s, const float* planes, uint32_t n_frusta, uint32_t flags, const bool use_static, const char* who) {
    if (!s) return fail(nullptr, YH_ERR_BAD_ARG, "%s: scene is NULL", who);
    yh_ctx* ctx = s->ctx;
    if (!s->have_hierarchy) return fail(ctx, YH_ERR_STATE, "%s: call yh_upload_hierarchy first", who);
    const bool cull = !(flags & YH_FRAME_NO_CULL) && s->desc.bounds_kind != YH_BOUNDS_NONE && n_frusta > 0;
    if (cull) {
        if (!planes) return fail(ctx, YH_ERR_BAD_ARG, "%s: planes is NULL", who);
        if (n_frusta > s->desc.n_frusta) return fail(ctx, YH_ERR_BAD_ARG, "%s: n_frusta exceeds the scene's draw lists", who);
        if (s->desc.bounds_kind == YH_BOUNDS_MESH_TABLE && s->n_meshes == 0)
            return fail(ctx, YH_ERR_STATE, "%s: mesh table not uploaded", who);
    }
    if (s->desc.local_kind == YH_LOCAL_TRS && !(flags & YH_FRAME_ALL_DIRTY))
        return fail(ctx, YH_ERR_UNSUPPORTED, "%s: TRS scenes recompute every entity; pass YH_FRAME_ALL_DIRTY", who);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (s->exp_pending && s->pend_demand)  // nobody read the lists of a YH_FRAME_LISTS_ON_DEMAND frame: dropped unexpanded
        s->exp_pending = s->pend_demand = s->pend_count = false;
    StaticLayout sl{};
    unsigned char* const sp = ctx->scratch_ptr;  // the device address of this GPU's copy, never the array's name
    if (use_static) {
        if (!sp) return fail(ctx, YH_ERR_STATE, "%s: the static scratch array was not resolved at yh_init", who);
        sl = static_layout(s, cull ? n_frusta : 0);
        if (cull && sl.bytes > (uint64_t)YH_STATIC_SCRATCH_BYTES)
            return fail(ctx, YH_ERR_CAPACITY,
                        "%s: %llu entities x %u frusta need %llu bytes of stage-3 scratch, the static array holds %llu "
                        "(about 119M entity-frusta); use yh_frame",
                        who, (unsigned long long)s->n, n_frusta, (unsigned long long)sl.bytes, (unsigned long long)YH_STATIC_SCRATCH_BYTES);
        if (cull) {  // this GPU's array is one: another context's static frame that nobody has waited for still owns it
            yh_ctx* owner = nullptr;
            if (!g_static_owner[ctx->device % YH_MAX_DEVICES].compare_exchange_strong(owner, ctx) && owner != ctx)
                return fail(ctx, YH_ERR_STATE,
                            "%s: another context on device %d has a static frame in flight (its stage-3 scratch is this GPU's one "
                            "static array); wait for that context (yh_wait) first, or use yh_frame", who, ctx->device);
        }
        if (int rc = flush_expand(s)) return rc;  // an expansion a deferred frame left behind: launch it now (no allocation)
        // the array keeps the previous call's contents: the boundary counters must start at zero, everything else
        // (tile counts, words, byte counts) is fully rewritten by this frame's launches before it is read
        if (cull) HIP_TRY(ctx, hipMemsetAsync(sp, 0, 4 * YH_MAX_FRUSTA * 4, st));
    }
    const bool ktiming = s->kernel_timing && !use_static;  // per-launch events are created lazily: not here

    FrameArgs a;
    memset(&a, 0, sizeof a);
    a.trs = s->trs;
    a.lmat = s->lmat;
    a.parent = s->parent;
    a.bounds = s->bounds;
    a.mesh_id = s->mesh_id;
    a.world = s->world;
    a.ldirty = s->ldirty;
    a.wdirty = s->wdirty;
    a.alive = s->alive;
    a.bcount = s->bsets + YH_MAX_FRUSTA * s->bset;
    // visibility words and their byte counts: frames alternate between two sets (the previous frame's may still be
    // waiting for its deferred expansion, or be on their way through the collective stream)
    uint64_t* words = s->vismask;
    uint8_t* byte_counts = s->viscnt;
    if (use_static) {
        a.bcount = (uint32_t*)sp + YH_MAX_FRUSTA;
        words = (uint64_t*)(sp + sl.words_off);
        byte_counts = sp + sl.cnt_off;
    } else if (cull && s->mask_alt) {
        const uint32_t p = s->mparity;
        if (p % YH_MASK_HALF == 0) {  // about to rewrite sets p .. p+HALF-1: the latest gather that sent one of them
            int latest = -1;          // must be done (the collective stream runs in order, so the earlier ones are too)
            for (uint32_t q = p; q < p + YH_MASK_HALF; ++q)
                if (s->mask_busy[q]) {
                    if (latest < 0 || s->mask_frame[q] > s->mask_frame[latest]) latest = (int)q;
                    s->mask_busy[q] = false;
                }
            if (latest >= 0) HIP_TRY(ctx, hipStreamWaitEvent(st, s->ev_mask[latest], 0));
        }
        if (p) {
            words = s->mask_alt + (uint64_t)(p - 1) * s->desc.n_frusta * s->n_chunks;
            byte_counts = s->viscnt_alt + (uint64_t)(p - 1) * s->desc.n_frusta * s->cnt_stride;
        }
    }
    a.vismask = words;
    a.viscnt = byte_counts;
    a.cnt_stride = s->cnt_stride;
    a.n_chunks = s->n_chunks;
    a.n_frusta = cull ? n_frusta : 0;
    a.id_base = s->desc.id_base;
    a.n_meshes = s->n_meshes;
    memcpy(a.root, s->root, sizeof a.root);
    if (cull) memcpy(a.planes, planes, (size_t)n_frusta * 24 * sizeof(float));
    if (cull && s->use_cull_plan) build_cull_plan(planes, n_frusta, a.plan);
    const int bounds = cull ? (int)s->desc.bounds_kind : 0;
    // XCD-aware tile order (xcd_tile), as measured: the launch of a flat scene gives every XCD one contiguous run of
    // tiles (256M instances 4.98 -> 4.77 ms; runs of 64-512 tiles do the same, runs of 8 are slower than no mapping);
    // tree launches give every XCD runs of 8 consecutive tiles (2048 sorted siblings share their parents' lines in
    // one L2 while all XCDs still work in the same region: cfg2 36.0 -> 34.4 us, cfg4 0.372 -> 0.368 ms; runs of 2-4
    // or 16-64 gain less, one contiguous run per XCD LOSES 2-4.5 %; the Mat4 level kernels gain 1.5-2.5 % on trees).
    auto grid_for = [&](FrameArgs& fa, bool flat_scene = false) -> dim3 {
        fa.n_tiles = (uint32_t)cdiv((uint64_t)fa.end - (fa.begin & ~63u), YH_BLOCK);  // chunk-aligned 256-entity tiles
        fa.xcd_per = fa.xcd_k = 0;
        if (fa.n_tiles < 64 || (s->desc.local_kind != YH_LOCAL_TRS && s->desc.n_levels == 1)) return dim3(fa.n_tiles);
        if (flat_scene) {
            fa.xcd_per = (fa.n_tiles + 7u) / 8u;
            return dim3(fa.xcd_per * 8u);
        }
        fa.xcd_k = YH_XCD_RUN;
        return dim3((fa.n_tiles + 8u * YH_XCD_RUN - 1u) / (8u * YH_XCD_RUN) * (8u * YH_XCD_RUN));
    };

    const bool frame_events = ktiming && s->timing_only == -1 && (s->frames % s->timing_period == 0);
    if (frame_events) HIP_TRY(ctx, hipEventRecord(s->ev0, st));
    uint32_t launches = 0;
    if (ktiming)
        if (int rc = ensure_kev(s)) return rc;

    bool timing_this = false;
    auto time_begin = [&](uint32_t group) -> int {
        timing_this = ktiming && s->kev_used < YH_TIMING_RING && (s->timing_only == -1 || s->timing_only == (int)group) &&
                      (s->frames % s->timing_period == 0);
        if (timing_this) {
            s->kev_group[s->kev_used] = group;
            HIP_TRY(ctx, hipEventRecord(s->kev[2 * s->kev_used], st));
        }
        return YH_OK;
    };
    auto time_end = [&]() -> int {
        if (timing_this) {
            HIP_TRY(ctx, hipEventRecord(s->kev[2 * s->kev_used + 1], st));
            ++s->kev_used;
        }
        return YH_OK;
    };
    // single-kernel groups: the event pair rides on the launch instead
    auto time_launch = [&](uint32_t group) -> LaunchEvents {
        LaunchEvents ev;
        if (ktiming && s->kev_used < YH_TIMING_RING && (s->timing_only < 0 || (uint32_t)s->timing_only == group) &&
            (s->frames % s->timing_period == 0)) {
            s->kev_group[s->kev_used] = group;
            ev.start = s->kev[2 * s->kev_used];
            ev.stop = s->kev[2 * s->kev_used + 1];
            ++s->kev_used;
        }
        return ev;
    };

    s->group_kernel.assign(s->groups.size(), std::string());
    std::string kn;
    if (s->desc.local_kind == YH_LOCAL_TRS) {
        if (s->exp_pending && (s->groups.empty() || s->groups[0].lo != 0 || s->level_offsets[s->groups[0].hi + 1] == s->level_offsets[0]))
            if (int rc = flush_expand(s)) return rc;
        for (uint32_t gi = 0; gi < s->groups.size(); ++gi) {
            const LevelGroup g = s->groups[gi];
            a.begin = s->level_offsets[g.lo];
            a.end = s->level_offsets[g.hi + 1];
            if (a.end == a.begin) continue;
            const bool has_base = g.lo > 0;
            const LaunchEvents ev = time_launch(gi);
            if (s->exp_pending && !has_base) {  // this launch carries the previous frame's expansion
                ExpandArgs xf = s->pend;
                xf.fuse_blocks = xf.n_tiles * s->pend_frusta;
                s->exp_pending = false;
                for (uint32_t j = 0; j < YH_MAX_GROUP; ++j)
                    a.lvl_begin[j] = (g.lo + 1 + j <= g.hi) ? s->level_offsets[g.lo + 1 + j] : 0xFFFFFFFFu;
                dim3 grid = grid_for(a, s->desc.n_levels == 1);
                grid.x += xf.fuse_blocks;
                const int G = (int)(g.hi - g.lo + 1);
                if (bounds == 0) kn = launch_first_x<0>(G, grid, st, ev, a, xf);
                else if (bounds == 1) kn = launch_first_x<1>(G, grid, st, ev, a, xf);
                else kn = launch_first_x<2>(G, grid, st, ev, a, xf);
            } else if (g.hi > g.lo) {
                for (uint32_t j = 0; j < YH_MAX_GROUP; ++j)
                    a.lvl_begin[j] = (g.lo + 1 + j <= g.hi) ? s->level_offsets[g.lo + 1 + j] : 0xFFFFFFFFu;
                const dim3 grid = grid_for(a);
                const int G = (int)(g.hi - g.lo + 1);
                const bool big = s->n >= s->big_tree;
                if (bounds == 0) kn = launch_chain<0>(G, has_base, grid, st, ev, a, big);
                else if (bounds == 1) kn = launch_chain<1>(G, has_base, grid, st, ev, a, big);
                else kn = launch_chain<2>(G, has_base, grid, st, ev, a, big);
            } else {
                const dim3 grid = grid_for(a, s->desc.n_levels == 1);
                const bool big = s->n >= (s->desc.n_levels == 1 ? s->big_flat : s->big_tree);
                if (has_base && s->lds_parents) {  // parents staged in LDS (A/B variant, yh_scene_set_lds_parents)
                    if (bounds == 0) kn = launch_trs_lds<0>(grid, st, ev, a, big);
                    else if (bounds == 1) kn = launch_trs_lds<1>(grid, st, ev, a, big);
                    else kn = launch_trs_lds<2>(grid, st, ev, a, big);
                } else if (bounds == 0) kn = launch_trs<0>(has_base, grid, st, ev, a, big);
                else if (bounds == 1) kn = launch_trs<1>(has_base, grid, st, ev, a, big);
                else kn = launch_trs<2>(has_base, grid, st, ev, a, big);
            }
            HIP_TRY(ctx, hipGetLastError());
            s->group_kernel[gi] = kn;
            ++launches;
        }
    } else {
        if (flags & YH_FRAME_ALL_DIRTY) HIP_TRY(ctx, hipMemsetAsync(s->ldirty, 1, s->stride, st));
        for (uint32_t gi = 0; gi < s->groups.size(); ++gi) {
            const LevelGroup g = s->groups[gi];
            a.begin = s->level_offsets[g.lo];
            a.end = s->level_offsets[g.hi + 1];
            if (a.end == a.begin) continue;
            const bool has_base = g.lo > 0;
            const dim3 grid = grid_for(a);
            if (int rc = time_begin(gi)) return rc;
            if (g.hi > g.lo) {
                for (uint32_t j = 0; j < YH_MAX_GROUP; ++j)
                    a.lvl_begin[j] = (g.lo + 1 + j <= g.hi) ? s->level_offsets[g.lo + 1 + j] : 0xFFFFFFFFu;
                const int G = (int)(g.hi - g.lo + 1);
                if (bounds == 0) kn = launch_chain_mat4<0>(G, has_base, grid, st, a, s->root_dirty);
                else if (bounds == 1) kn = launch_chain_mat4<1>(G, has_base, grid, st, a, s->root_dirty);
                else kn = launch_chain_mat4<2>(G, has_base, grid, st, a, s->root_dirty);
            } else {
                const bool big = s->n >= s->big_tree;  // 12.5M columns, all dirty: 452 -> 417 us with streamed locals and bounds
                if (bounds == 0) kn = launch_mat4<0>(has_base, big, grid, st, a, s->root_dirty);
                else if (bounds == 1) kn = launch_mat4<1>(has_base, big, grid, st, a, s->root_dirty);
                else kn = launch_mat4<2>(has_base, big, grid, st, a, s->root_dirty);
            }
            HIP_TRY(ctx, hipGetLastError());
            s->group_kernel[gi] = kn;
            if (int rc = time_end()) return rc;
            ++launches;
        }
        // PostWorldTransform / PostLocalTransform (PostTransformSystem.zig:12-37)
        if (s->track) {
            hipLaunchKernelGGL(k_sweep_dirty, dim3(cdiv(s->stride, YH_BLOCK * 16ull)), dim3(YH_BLOCK), 0, st, s->ldirty, s->wdirty,
                               (uint32_t)s->n, s->chg_words, s->chg_cnt, s->n_chunks, s->cnt_stride, s->alive);
            HIP_TRY(ctx, hipGetLastError());
            ExpandArgs c;
            memset(&c, 0, sizeof c);
            c.vismask = s->chg_words;
            c.viscnt = s->chg_cnt;
            c.cnt_stride = s->cnt_stride;
            c.tilecnt = s->chg_tilecnt;
            c.tc_stride = (uint32_t)(((uint64_t)s->n_tiles + s->n_segs + 1) & ~1ull);
            c.counts = s->chg;
            c.bcount_cur = s->chg + YH_MAX_FRUSTA;
            c.bcount_next = s->chg + 2 * YH_MAX_FRUSTA;
            c.host_counts = s->h_chg_dev;
            c.draw[0] = s->chg_list;
            c.draw[1] = s->chg_list_l;
            c.draw_cap = s->stride;
            c.n_chunks = s->n_chunks;
            c.n_tiles = s->n_tiles;
            c.words = s->tile_words;
            c.self_count = (s->n_tiles <= (uint32_t)YH_SELF_COUNT_TILES && s->tile_words <= 64u);
            if (!c.self_count) hipLaunchKernelGGL(k_count, dim3(s->n_segs, 2), dim3(YH_BLOCK), 0, st, c);
            hipLaunchKernelGGL(k_expand, dim3(s->n_tiles, 2), dim3(YH_BLOCK), 0, st, c);
            HIP_TRY(ctx, hipGetLastError());
            s->chg_valid = true;
            launches += c.self_count ? 2 : 3;
        } else {
            hipLaunchKernelGGL(k_clear_dirty, dim3(cdiv(s->stride / 4, YH_BLOCK)), dim3(YH_BLOCK), 0, st, (uint32_t*)s->ldirty,
                               (uint32_t*)s->wdirty, s->stride / 4);
            HIP_TRY(ctx, hipGetLastError());
            ++launches;
        }
    }
    s->root_dirty = 0;  // the world sweep clears the root's flag too
    if (cull) {
        // stage 3, part 2: tile prefix sums + expansion of the visibility words into ordered draw lists
        ExpandArgs x;
        memset(&x, 0, sizeof x);
        x.vismask = words;
        x.viscnt = byte_counts;
        x.cnt_stride = s->cnt_stride;
        x.counts = s->counts;
        x.host_counts = s->h_counts_dev;
        if (use_static) {
            s->last_mask = nullptr;  // these words are gone with the next call on this GPU: nothing may send them later
            x.tilecnt = (uint32_t*)(sp + sl.tilecnt_off);
            x.bcount_cur = (uint32_t*)sp + YH_MAX_FRUSTA;
            x.bcount_next = (uint32_t*)sp + 2 * YH_MAX_FRUSTA;
        } else {
            s->last_mask = words;
            s->last_mparity = s->mparity;
            if (s->mask_alt) s->mparity = (s->mparity + 1u) % YH_MASK_SETS;
            x.tilecnt = s->tilecnt;
            // boundary counters (a ring of YH_BSETS sets): this frame's, the next frame's (its launches may already be
            // adding to it when a deferred expansion runs) and the one after, which this expansion zeroes
            x.bcount_cur = s->bsets + YH_MAX_FRUSTA * s->bset;
            x.bcount_next = s->bsets + YH_MAX_FRUSTA * ((s->bset + 2u) % YH_BSETS);
            s->bset = (s->bset + 1u) % YH_BSETS;
            if (s->bzero_ahead) --s->bzero_ahead;  // (the window of bulk-zeroed sets ends one frame nearer)
        }
        for (uint32_t f = 0; f < YH_MAX_FRUSTA; ++f) x.draw[f] = s->draw[f];
        x.draw_cap = s->draw_cap;
        x.n_chunks = s->n_chunks;
        x.n_tiles = s->n_tiles;
        x.tc_stride = s->n_tiles + s->n_segs;
        x.words = s->tile_words;
        x.id_base = s->desc.id_base;
        x.self_count = (s->n_tiles <= (uint32_t)YH_SELF_COUNT_TILES && s->tile_words <= 64u);
        // Lists on demand (frames whose words go to the gather, which expands this GPU's part along with the others'):
        // neither tile bases nor lists unless somebody asks before the next frame; what k_expand does on the side - the
        // boundary counters of the frame after next start at zero - is done by a one-wave kernel.
        const bool on_demand = !use_static && (flags & YH_FRAME_ASYNC) && (flags & YH_FRAME_LISTS_ON_DEMAND) &&
                               !(ktiming && s->timing_only == -1);
        if (on_demand) {
            if (s->bzero_ahead == 0) {  // sets k + 2 .. k + 33 (s->bset is already k + 1): last used 31..62 frames ago
                hipLaunchKernelGGL(k_zero_sets, dim3(1), dim3(YH_BLOCK), 0, st, s->bsets, (s->bset + 1u) % YH_BSETS,
                                   (uint32_t)YH_BZERO_AHEAD, (uint32_t)YH_BSETS);
                HIP_TRY(ctx, hipGetLastError());
                ++launches;
                s->bzero_ahead = YH_BZERO_AHEAD;
            }
            s->pend = x;
            s->pend_frusta = n_frusta;
            s->exp_pending = true;
            s->pend_demand = true;
            s->pend_count = !x.self_count;
            goto cull_done;
        }
        if (int rc = time_begin(YH_MAX_LEVELS - 1)) return rc;
        if (!x.self_count) {
            hipLaunchKernelGGL(k_count, dim3(s->n_segs, n_frusta), dim3(YH_BLOCK), 0, st, x);
            HIP_TRY(ctx, hipGetLastError());
            ++launches;
        }
        // TRS scenes, frames enqueued without a host sync: leave the expansion to the next frame's first launch (or to
        // whatever reads the lists first).  For small scenes its own dependent launch is a tenth of the frame; larger
        // trees still gain 1-2 % when their first launch is one of up to 1M entities, which is what the planner gives
        // the top of a large tree (12.5M entities, first launch 740K: 331.7 -> 328.8 us; profiles/r02/ab_defer_first.txt).
        // Flat large scenes keep k_expand: riding on their one big launch costs that kernel more than the boundary.
        const bool small_first = !s->groups.empty() && s->groups.size() > 1 &&
                                 s->level_offsets[s->groups[0].hi + 1] - s->level_offsets[0] <= s->defer_first_max;
        const bool defer = !use_static && (flags & YH_FRAME_ASYNC) && s->desc.local_kind == YH_LOCAL_TRS &&
                           (x.self_count || small_first) && !(ktiming && s->timing_only == -1);
        if (defer) {
            s->pend = x;
            s->pend_frusta = n_frusta;
            s->exp_pending = true;
        } else {
            hipLaunchKernelGGL(k_expand, dim3(s->n_tiles, n_frusta), dim3(YH_BLOCK), 0, st, x);
            HIP_TRY(ctx, hipGetLastError());
            ++launches;
        }
        if (int rc = time_end()) return rc;
    }
cull_done:
    if (frame_events) HIP_TRY(ctx, hipEventRecord(s->ev1, st));
    s->frame_timed = frame_events;
    s->frame_pending = true;
    s->last_n_frusta = a.n_frusta;
    s->last_launches = launches;
    ++s->frames;
    if (!(flags & YH_FRAME_ASYNC)) return sync_scene(s);
    return YH_OK;
}

extern "C" int yh_frame(yh_scene* s, const float* planes, uint32_t n_frusta, uint32_t flags) {
    return frame_impl(s, planes, n_frusta, flags, false, "yh_frame");
}

// yh_frame for callers that cannot allocate (HIP graph capture, serving loops, harnesses that forbid it): the same
// launches, the stage-3 scratch in this GPU's copy of the file-scope `scratch` array.  No hipMalloc / hipFree /
// event creation on this path; hipMemsetAsync and the launches take the device address kept in the context.
extern "C" int yh_frame_static(yh_scene* s, const float* planes, uint32_t n_frusta, uint32_t flags) {
    return frame_impl(s, planes, n_frusta, flags, true, "yh_frame_static");
}

extern "C" int yh_get_counts(yh_scene* s, uint32_t* counts, uint32_t* boundary_counts) {
    if (!s) return fail(nullptr, YH_ERR_BAD_ARG, "yh_get_counts: scene is NULL");
    if (s->frame_pending)
        if (int rc = sync_scene(s)) return rc;
    for (uint32_t f = 0; f < s->last_n_frusta; ++f) {
        if (counts) counts[f] = s->h_counts[f];
        if (boundary_counts) boundary_counts[f] = s->h_counts[YH_MAX_FRUSTA + f];
    }
    return YH_OK;
}

extern "C" int yh_get_draw_list(yh_scene* s, uint32_t frustum, uint32_t* out, uint64_t cap, uint64_t* written) {
    if (!s) return fail(nullptr, YH_ERR_BAD_ARG, "yh_get_draw_list: scene is NULL");
    ARG_TRY(s->ctx, frustum < s->last_n_frusta, "yh_get_draw_list: frustum index out of range");
    if (s->frame_pending)
        if (int rc = sync_scene(s)) return rc;
    const uint64_t c = s->h_counts[frustum];
    if (written) *written = c;
    if (c > s->draw_cap)
        return fail(s->ctx, YH_ERR_CAPACITY, "yh_get_draw_list: %llu visible entities exceed the scene's draw capacity %llu",
                    (unsigned long long)c, (unsigned long long)s->draw_cap);
    if (c > cap || (c && !out))
        return fail(s->ctx, YH_ERR_CAPACITY, "yh_get_draw_list: buffer of %llu entries is too small, %llu required",
                    (unsigned long long)cap, (unsigned long long)c);
    HIP_TRY(s->ctx, hipSetDevice(s->ctx->device));
    if (c) HIP_TRY(s->ctx, hipMemcpy(out, s->draw[frustum], c * 4, hipMemcpyDeviceToHost));
    return YH_OK;
}

// ---- draw-list consumer -----------------------------------------------------------------------
extern "C" int yh_upload_vertex_counts(yh_scene* s, const uint32_t* counts, uint64_t n) {
    if (!s || !counts) return fail(s ? s->ctx : nullptr, YH_ERR_BAD_ARG, "yh_upload_vertex_counts: NULL argument");
    yh_ctx* ctx = s->ctx;
    const bool table = s->desc.bounds_kind == YH_BOUNDS_MESH_TABLE;
    const uint64_t want = table ? s->desc.max_meshes : s->n;
    if (table ? (n == 0 || n > want) : (n != want))
        return fail(ctx, YH_ERR_BAD_ARG, "yh_upload_vertex_counts: %llu counts, scene wants %s%llu", (unsigned long long)n,
                    table ? "at most " : "", (unsigned long long)want);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = sync_scene(s)) return rc;
    if (!s->vcount) {
        HIP_TRY(ctx, hipMalloc((void**)&s->vcount, want * 4));
        HIP_TRY(ctx, hipMemset(s->vcount, 0, want * 4));
    }
    HIP_TRY(ctx, hipMemcpy(s->vcount, counts, n * 4, hipMemcpyHostToDevice));
    return YH_OK;
}

extern "C" int yh_build_object_data(yh_scene* s, uint32_t frustum, uint32_t batch_offset) {
    if (!s) return fail(nullptr, YH_ERR_BAD_ARG, "yh_build_object_data: scene is NULL");
    yh_ctx* ctx = s->ctx;
    if (frustum >= s->last_n_frusta) return fail(ctx, YH_ERR_STATE, "yh_build_object_data: frustum %u was not culled by the last frame", frustum);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = flush_expand(s)) return rc;
    if (!s->obj[frustum]) {
        HIP_TRY(ctx, hipMalloc((void**)&s->obj[frustum], s->draw_cap * 64));
        HIP_TRY(ctx, hipMalloc((void**)&s->cmd[frustum], s->draw_cap * 16));
    }
    hipLaunchKernelGGL(k_build_objects, dim3(cdiv(std::min<uint64_t>(s->n, s->draw_cap), YH_BLOCK)), dim3(YH_BLOCK), 0, ctx->stream,
                       s->world, s->draw[frustum], s->counts + frustum, s->draw_cap, s->desc.id_base, s->vcount,
                       s->desc.bounds_kind == YH_BOUNDS_MESH_TABLE ? s->mesh_id : nullptr, batch_offset, s->obj[frustum],
                       s->cmd[frustum]);
    HIP_TRY(ctx, hipGetLastError());
    s->frame_pending = true;
    s->bucketed[frustum] = false;
    return YH_OK;
}

// Per-mesh bucketing: the same object data and draw records, grouped by mesh id.
extern "C" int yh_build_mesh_buckets(yh_scene* s, uint32_t frustum, uint32_t batch_offset) {
    if (!s) return fail(nullptr, YH_ERR_BAD_ARG, "yh_build_mesh_buckets: scene is NULL");
    yh_ctx* ctx = s->ctx;
    if (s->desc.bounds_kind != YH_BOUNDS_MESH_TABLE)
        return fail(ctx, YH_ERR_STATE, "yh_build_mesh_buckets: the scene has no mesh table (entities carry no mesh id)");
    if (frustum >= s->last_n_frusta) return fail(ctx, YH_ERR_STATE, "yh_build_mesh_buckets: frustum %u was not culled by the last frame", frustum);
    if (s->n_meshes < 1 || s->n_meshes > YH_BUCKET_MAX_KEYS)
        return fail(ctx, YH_ERR_UNSUPPORTED, "yh_build_mesh_buckets: %u meshes, 1..%u supported", s->n_meshes, YH_BUCKET_MAX_KEYS);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = flush_expand(s)) return rc;
    if (!s->obj[frustum]) {
        HIP_TRY(ctx, hipMalloc((void**)&s->obj[frustum], s->draw_cap * 64));
        HIP_TRY(ctx, hipMalloc((void**)&s->cmd[frustum], s->draw_cap * 16));
    }
    // Up to 256 keys: 4096 workgroup slices, tiles sorted inside LDS.  More (one key per (material, mesh) pair of a large
    // scene): one slice per WAVE, counters in global memory only, keys x slices kept within 16M counters.
    const uint32_t M = std::min<uint32_t>(std::max<uint32_t>(s->desc.max_meshes, 1u), YH_BUCKET_MAX_KEYS);
    const bool wide = s->n_meshes > YH_BUCKET_MAX_MESHES || getenv("YH_BUCKET_WIDE") != nullptr;  // (the knob: A/B runs of the two paths)
    // Slices the list is cut into.  Few keys: up to 32768 workgroup slices of at least four 256-entry tiles each, keys x
    // slices within 4M counters (64M instances / 16 keys: 4096 slices 1.945 ms, 32768 1.898, one slice per tile - 250k -
    // 2.448: the scans over the slices grow; profiles/r03/bucket_slices.txt).  Many keys: one slice per wave, <= 16384.
    uint32_t slices;
    if (wide) {
        slices = 16384u;
        while (slices > 256u && (uint64_t)slices * s->n_meshes > (16ull << 20)) slices >>= 1;
    } else {
        const uint64_t tiles = (std::min<uint64_t>(s->draw_cap, s->n) + YH_BLOCK - 1) / YH_BLOCK;
        const uint64_t room = (4ull << 20) / s->n_meshes;
        slices = (uint32_t)std::max<uint64_t>(std::min<uint64_t>(std::min<uint64_t>(tiles / 4, 32768), room), 256);
        if (const char* v = getenv("YH_BUCKET_SLICES")) slices = (uint32_t)strtoul(v, nullptr, 10);  // (A/B runs)
        slices = (slices + 255u) & ~255u;
    }
    if (!s->blist[frustum]) {
        HIP_TRY(ctx, hipMalloc((void**)&s->blist[frustum], s->draw_cap * 4));
        HIP_TRY(ctx, hipMalloc((void**)&s->bkt[frustum], (uint64_t)M * 12));
    }
    const uint64_t scratch_words = (uint64_t)s->n_meshes * slices + 2ull * M;
    if (scratch_words > s->bscratch_words) {  // (the mesh table grew, or went from few keys to many)
        if (s->bscratch) {
            if (int rc = sync_scene(s)) return rc;
            HIP_TRY(ctx, hipFree(s->bscratch));
            s->bscratch = nullptr;
            s->bscratch_words = 0;
        }
        const uint64_t words = scratch_words + scratch_words / 4;
        HIP_TRY(ctx, hipMalloc((void**)&s->bscratch, words * 4));
        s->bscratch_words = words;
    }
    BucketArgs a{};
    a.list = s->draw[frustum];
    a.count_ptr = s->counts + frustum;
    a.cap = s->draw_cap;
    a.id_base = s->desc.id_base;
    a.mesh_id = s->mesh_id;
    a.n_meshes = s->n_meshes;
    a.hist = s->bscratch;
    a.totals = s->bscratch + (uint64_t)s->n_meshes * slices;
    a.starts = a.totals + M;
    a.buckets = s->bkt[frustum];
    a.world = s->world;
    a.vcount = s->vcount;
    a.batch_offset = batch_offset;
    a.out_list = s->blist[frustum];
    a.obj = s->obj[frustum];
    a.cmds = s->cmd[frustum];
    a.n_slices = slices;
    if (wide) {
        HIP_TRY(ctx, hipMemsetAsync(a.hist, 0, (uint64_t)s->n_meshes * slices * 4, ctx->stream));
        hipLaunchKernelGGL(k_bucketw_hist, dim3(slices / 4), dim3(YH_BLOCK), 0, ctx->stream, a);
    } else {
        hipLaunchKernelGGL(k_bucket_hist, dim3(slices), dim3(YH_BLOCK), 0, ctx->stream, a);
    }
    hipLaunchKernelGGL(k_bucket_rows, dim3(s->n_meshes), dim3(YH_BLOCK), 0, ctx->stream, a);
    hipLaunchKernelGGL(k_bucket_starts, dim3(1), dim3(YH_BLOCK), 0, ctx->stream, a);
    if (wide) hipLaunchKernelGGL(k_bucketw_scatter, dim3(slices / 4), dim3(YH_BLOCK), 0, ctx->stream, a);
    else hipLaunchKernelGGL(k_bucket_scatter, dim3(slices), dim3(YH_BLOCK), 0, ctx->stream, a);
    HIP_TRY(ctx, hipGetLastError());
    s->frame_pending = true;
    s->bucketed[frustum] = true;
    return YH_OK;
}

extern "C" int yh_get_mesh_buckets(yh_scene* s, uint32_t frustum, yh_bucket* buckets, uint32_t cap_buckets,
                                   uint32_t* entities, uint64_t cap, uint64_t* written) {
    if (!s || !written) return fail(s ? s->ctx : nullptr, YH_ERR_BAD_ARG, "yh_get_mesh_buckets: NULL argument");
    yh_ctx* ctx = s->ctx;
    if (frustum >= YH_MAX_FRUSTA || !s->bucketed[frustum]) return fail(ctx, YH_ERR_STATE, "yh_get_mesh_buckets: call yh_build_mesh_buckets first");
    if (int rc = sync_scene(s)) return rc;
    const uint64_t n = std::min<uint64_t>(s->h_counts[frustum], s->draw_cap);
    *written = n;
    if (buckets && cap_buckets < s->n_meshes)
        return fail(ctx, YH_ERR_CAPACITY, "yh_get_mesh_buckets: %u buckets, room for %u", s->n_meshes, cap_buckets);
    if (entities && n > cap) return fail(ctx, YH_ERR_CAPACITY, "yh_get_mesh_buckets: %llu entities, room for %llu", (unsigned long long)n, (unsigned long long)cap);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (buckets) HIP_TRY(ctx, hipMemcpy(buckets, s->bkt[frustum], (uint64_t)s->n_meshes * 12, hipMemcpyDeviceToHost));
    if (entities && n) HIP_TRY(ctx, hipMemcpy(entities, s->blist[frustum], n * 4, hipMemcpyDeviceToHost));
    return YH_OK;
}

extern "C" int yh_get_object_data(yh_scene* s, uint32_t frustum, void* model_matrices, yh_draw_cmd* cmds, uint64_t cap,
                                  uint64_t* written) {
    if (!s || !written) return fail(s ? s->ctx : nullptr, YH_ERR_BAD_ARG, "yh_get_object_data: NULL argument");
    yh_ctx* ctx = s->ctx;
    if (frustum >= YH_MAX_FRUSTA || !s->obj[frustum]) return fail(ctx, YH_ERR_STATE, "yh_get_object_data: call yh_build_object_data first");
    if (int rc = sync_scene(s)) return rc;
    const uint64_t n = std::min<uint64_t>(s->h_counts[frustum], s->draw_cap);
    *written = n;
    if (n > cap) return fail(ctx, YH_ERR_CAPACITY, "yh_get_object_data: %llu objects, room for %llu", (unsigned long long)n, (unsigned long long)cap);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (model_matrices && n) HIP_TRY(ctx, hipMemcpy(model_matrices, s->obj[frustum], n * 64, hipMemcpyDeviceToHost));
    if (cmds && n) HIP_TRY(ctx, hipMemcpy(cmds, s->cmd[frustum], n * 16, hipMemcpyDeviceToHost));
    return YH_OK;
}

extern "C" void* yh_object_data_device(yh_scene* s, uint32_t frustum, void** cmds_device) {
    if (!s || frustum >= YH_MAX_FRUSTA) return nullptr;
    if (cmds_device) *cmds_device = s->cmd[frustum];
    return s->obj[frustum];
}

extern "C" int yh_timings(yh_scene* s, yh_timing* out) {
    if (!s || !out) return fail(s ? s->ctx : nullptr, YH_ERR_BAD_ARG, "yh_timings: NULL argument");
    memset(out, 0, sizeof *out);
    out->frames = s->frames;
    out->n_launches = s->last_launches;
    out->n_groups = (uint32_t)s->groups.size();
    if (s->frames == 0) return YH_OK;
    if (int rc = sync_scene(s)) return rc;
    if (s->frame_timed) HIP_TRY(s->ctx, hipEventElapsedTime(&out->last_frame_ms, s->ev0, s->ev1));
    // per-launch events (when enabled): mean duration of the slowest group since the last call
    if (s->kernel_timing && s->kev_used) {
        double sum[YH_MAX_LEVELS] = {0};
        uint32_t cnt[YH_MAX_LEVELS] = {0};
        for (uint32_t i = 0; i < s->kev_used; ++i) {
            float ms = 0;
            HIP_TRY(s->ctx, hipEventElapsedTime(&ms, s->kev[2 * i], s->kev[2 * i + 1]));
            const uint32_t g = s->kev_group[i] % YH_MAX_LEVELS;
            sum[g] += ms;
            cnt[g]++;
        }
        double best = 0;
        for (uint32_t g = 0; g < YH_MAX_LEVELS; ++g)
            if (cnt[g] && sum[g] / cnt[g] > best) best = sum[g] / cnt[g];
        out->last_kernel_ms = (float)best;
        s->kev_used = 0;
    }
    return YH_OK;
}

// Not part of the reference-facing boundary: per-launch HIP-event timing for the measurement harness.
extern "C" int yh_scene_set_kernel_timing_period(yh_scene* s, uint32_t period) {
    if (!s) return fail(nullptr, YH_ERR_BAD_ARG, "yh_scene_set_kernel_timing_period: scene is NULL");
    s->timing_period = period ? period : 1;
    return YH_OK;
}

extern "C" int yh_scene_set_kernel_timing(yh_scene* s, int enable) {
    if (!s) return fail(nullptr, YH_ERR_BAD_ARG, "yh_scene_set_kernel_timing: scene is NULL");
    s->kernel_timing = enable != 0;
    s->timing_only = enable >= 2 ? enable - 2 : enable < 0 ? -2 : -1;
    s->kev_used = 0;
    return YH_OK;
}

// Level groups of the current plan: writes up to cap (lo, hi, first_entity, end_entity) quadruples.
extern "C" int yh_scene_get_groups(yh_scene* s, uint32_t* quads, uint32_t cap, uint32_t* n_groups) {
    if (!s || !n_groups) return fail(s ? s->ctx : nullptr, YH_ERR_BAD_ARG, "yh_scene_get_groups: NULL argument");
    if (!s->have_hierarchy) return fail(s->ctx, YH_ERR_STATE, "yh_scene_get_groups: no hierarchy");
    *n_groups = (uint32_t)s->groups.size();
    for (uint32_t i = 0; i < s->groups.size() && i < cap && quads; ++i) {
        quads[4 * i] = s->groups[i].lo;
        quads[4 * i + 1] = s->groups[i].hi;
        quads[4 * i + 2] = s->level_offsets[s->groups[i].lo];
        quads[4 * i + 3] = s->level_offsets[s->groups[i].hi + 1];
    }
    return YH_OK;
}

// Kernel each level group's launch of the LAST frame ran, spelled as rocprofv3's kernel trace spells it, one per
// line in group order (an empty line for a group without entities).
extern "C" int yh_scene_get_group_kernel_names(yh_scene* s, char* buf, uint64_t cap, uint32_t* n_groups) {
    if (!s || !buf || !n_groups) return fail(s ? s->ctx : nullptr, YH_ERR_BAD_ARG, "yh_scene_get_group_kernel_names: NULL argument");
    if (s->frames == 0) return fail(s->ctx, YH_ERR_STATE, "yh_scene_get_group_kernel_names: no frame has run");
    std::string all;
    for (const std::string& k : s->group_kernel) all += k + "\n";
    *n_groups = (uint32_t)s->group_kernel.size();
    if (all.size() + 1 > cap)
        return fail(s->ctx, YH_ERR_CAPACITY, "yh_scene_get_group_kernel_names: %llu bytes needed, room for %llu",
                    (unsigned long long)all.size() + 1, (unsigned long long)cap);
    memcpy(buf, all.c_str(), all.size() + 1);
    return YH_OK;
}

// Mean device time per level group from the per-launch events recorded since the last call.
extern "C" int yh_scene_get_group_timings(yh_scene* s, float* mean_ms, uint32_t* samples, uint32_t cap) {
    if (!s || !mean_ms) return fail(s ? s->ctx : nullptr, YH_ERR_BAD_ARG, "yh_scene_get_group_timings: NULL argument");
    if (int rc = sync_scene(s)) return rc;
    std::vector<double> sum(cap, 0.0);
    std::vector<uint32_t> cnt(cap, 0);
    for (uint32_t i = 0; i < s->kev_used; ++i) {
        float ms = 0;
        HIP_TRY(s->ctx, hipEventElapsedTime(&ms, s->kev[2 * i], s->kev[2 * i + 1]));
        const uint32_t g = s->kev_group[i];
        if (g < cap) {
            sum[g] += ms;
            cnt[g]++;
        }
    }
    for (uint32_t g = 0; g < cap; ++g) {
        mean_ms[g] = cnt[g] ? (float)(sum[g] / cnt[g]) : 0.0f;
        if (samples) samples[g] = cnt[g];
    }
    s->kev_used = 0;
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

// ------------------------------------------------------------------------------------ frustum helpers
extern "C" void yh_mat4_mul(const float a[16], const float b[16], float out[16]) {
    M4 x, y;
    memcpy(x.v, a, 64);
    memcpy(y.v, b, 64);
    const M4 r = m4_mul(x, y);
    memcpy(out, r.v, 64);
}

// engine/math3d.zig:649-659
extern "C" void yh_perspective(float fovy_rad, float aspect, float near_z, float far_z, float out[16]) {
    const float f = 1.0f / tanf(fovy_rad / 2.0f);
    memset(out, 0, 64);
    out[0] = f / aspect;
    out[5] = f;
    out[10] = far_z / (near_z - far_z);
    out[11] = -1.0f;
    out[14] = -(far_z * near_z) / (far_z - near_z);
}

// engine/math3d.zig:661-683
extern "C" void yh_orthographic(float l, float r, float b, float t, float n, float f, float out[16]) {
    memset(out, 0, 64);
    out[0] = 2.0f / (r - l);
    out[5] = 2.0f / (b - t);
    out[10] = 1.0f / (n - f);
    out[12] = -(r + l) / (r - l);
    out[13] = -(b + t) / (b - t);
    out[14] = n / (n - f);
    out[15] = 1.0f;
}

static void h_normalized(const float a[3], float o[3]) {  // math3d.zig:116-123
    const float l = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    if (l > 0.0f) {
        o[0] = a[0] / l;
        o[1] = a[1] / l;
        o[2] = a[2] / l;
    } else {
        o[0] = o[1] = o[2] = 0.0f;
    }
}
static float h_dot(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void h_cross(const float a[3], const float b[3], float o[3]) {
    const float x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x;
    o[1] = y;
    o[2] = z;
}

// engine/math3d.zig:991-1021
extern "C" void yh_look_at(const float eye[3], const float at[3], const float up[3], float out[16]) {
    float x[3], y[3], z[3], tmp[3] = {eye[0] - at[0], eye[1] - at[1], eye[2] - at[2]};
    h_normalized(tmp, z);
    h_normalized(up, y);
    h_cross(y, z, tmp);
    h_normalized(tmp, x);
    h_cross(z, x, tmp);
    h_normalized(tmp, y);
    const float m[16] = {x[0], y[0], z[0], 0.0f, x[1], y[1], z[1], 0.0f, x[2], y[2], z[2], 0.0f,
                         -h_dot(x, eye), -h_dot(y, eye), -h_dot(z, eye), 1.0f};
    memcpy(out, m, 64);
}

// engine/math3d.zig:687-710 (host libm cos / sin; Zig uses its own routines, so the trig values are not pinned)
static void h_rotation(const float axis_in[3], float angle_rad, float out[16]) {
    float axis[3] = {axis_in[0], axis_in[1], axis_in[2]};
    const float c = cosf(angle_rad), s = sinf(angle_rad);
    const float t = 1.0f - c;
    const float sqr_norm = axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2];
    if (sqr_norm == 0.0f) {
        for (int i = 0; i < 16; ++i) out[i] = (i % 5 == 0) ? 1.0f : 0.0f;
        return;
    }
    if (fabsf(sqr_norm - 1.0f) > 0.0001f) {
        const float norm = sqrtf(sqr_norm);
        for (int i = 0; i < 3; ++i) axis[i] = axis[i] / norm;
    }
    const float x = axis[0], y = axis[1], z = axis[2];
    const float m[16] = {x * x * t + c, y * x * t + z * s, z * x * t - y * s, 0.0f,
                         x * y * t - z * s, y * y * t + c, z * y * t + x * s, 0.0f,
                         x * z * t + y * s, y * z * t - x * s, z * z * t + c, 0.0f,
                         0.0f, 0.0f, 0.0f, 1.0f};
    memcpy(out, m, 64);
}

// engine/components/camera.zig:161-166, 253-302: Camera.updateMatrices — view, projection and view_projection from the
// camera's position and Euler angles (updatePerspective* / updateOrthographic*, updateViewProjection).
extern "C" void yh_camera_update_matrices(int orthographic, const float pos[3], const float rot[3], float aspect,
                                          float fovy_rad, float near_z, float far_z, float view[16], float projection[16],
                                          float view_projection[16]) {
    const float rad_per_deg = 0.017453292519943295769f;  // std.math.degreesToRadians (camera.zig:261-263)
    const float ax[3] = {1.0f, 0.0f, 0.0f}, ay[3] = {0.0f, 1.0f, 0.0f}, az[3] = {0.0f, 0.0f, 1.0f};
    float rx[16], ry[16], rz[16], tr[16], a[16], b[16], v[16], proj[16];
    h_rotation(ax, orthographic ? rot[0] : rot[0] * rad_per_deg, rx);  // camera.zig:283-285 takes the angles as they come
    h_rotation(ay, orthographic ? rot[1] : rot[1] * rad_per_deg, ry);
    h_rotation(az, orthographic ? rot[2] : rot[2] * rad_per_deg, rz);
    for (int i = 0; i < 16; ++i) tr[i] = (i % 5 == 0) ? 1.0f : 0.0f;  // Mat4.translation(-pos), math3d.zig:628-635
    tr[12] = -pos[0];
    tr[13] = -pos[1];
    tr[14] = -pos[2];
    yh_mat4_mul(rx, ry, a);
    yh_mat4_mul(a, rz, b);
    yh_mat4_mul(b, tr, v);
    if (orthographic) {
        yh_orthographic(-10.0f, 10.0f, -10.0f, 10.0f, 0.0f, 200.0f, proj);  // camera.zig:296
    } else {
        yh_perspective(fovy_rad, aspect, near_z, far_z, proj);
        proj[5] *= -1.0f;  // camera.zig:272
    }
    if (view) memcpy(view, v, 64);
    if (projection) memcpy(projection, proj, 64);
    yh_mat4_mul(proj, v, view_projection);  // camera.zig:300-302
}

extern "C" void yh_camera_view_projection(int orthographic, const float pos[3], const float rot[3], float aspect,
                                          float fovy_rad, float near_z, float far_z, float out[16]) {
    yh_camera_update_matrices(orthographic, pos, rot, aspect, fovy_rad, near_z, far_z, nullptr, nullptr, out);
}

// components/transform.zig:68-70: `self.matrix.decompose() catch Mat4.Decomposed.IDENTITY` on one matrix, on the host
// (m4_decompose is the kernels' own restatement of math3d.zig:828-981).  Returns 0 where IDENTITY was substituted.
extern "C" int yh_mat4_decompose(const float m[16], float t[3], float q[4], float s[3]) {
    M4 x;
    memcpy(x.v, m, 64);
    float o[10];
    const bool ok = m4_decompose(x, o);
    if (!ok) {  // math3d.zig:535-539
        o[0] = o[1] = o[2] = 0.0f;
        o[3] = o[4] = o[5] = 0.0f;
        o[6] = 1.0f;
        o[7] = o[8] = o[9] = 1.0f;
    }
    memcpy(t, o, 12);
    memcpy(q, o + 3, 16);
    memcpy(s, o + 7, 12);
    return ok ? 1 : 0;
}

// engine/math3d.zig:418-442 — Quat.toEuler: roll / pitch / yaw in DEGREES.  atan2 / asin are libm's here; Zig 0.13 uses
// its own std.math routines (ports of musl's atan2f / asinf), absent from this image: their last bit is "parity unpinned".
extern "C" void yh_quat_to_euler(const float q[4], float euler_deg[3]) {
    const float half_pi = 1.57079632679489661923f, pi = 3.14159265358979323846f;
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    const float sinr_cosp = 2.0f * (w * x + y * z);
    const float cosr_cosp = 1.0f - 2.0f * (x * x + y * y);
    const float ex = atan2f(sinr_cosp, cosr_cosp);
    const float sinp = 2.0f * (w * y - z * x);
    const float ey = fabsf(sinp) >= 1.0f ? copysignf(half_pi, sinp) : asinf(sinp);
    const float siny_cosp = 2.0f * (w * z + x * y);
    const float cosy_cosp = 1.0f - 2.0f * (y * y + z * z);
    const float ez = atan2f(siny_cosp, cosy_cosp);
    euler_deg[0] = ex * 180.0f / pi;  // euler.mulf(180.0).divf(std.math.pi)
    euler_deg[1] = ey * 180.0f / pi;
    euler_deg[2] = ez * 180.0f / pi;
}

// editor/windows/GameWindow.zig:160-161 — what the engine does per Camera entity before drawing with it:
//   const decomposed = transform.decompose(); camera.updateMatrices(decomposed.translation, decomposed.rotation.toEuler(), aspect);
extern "C" int yh_camera_update_from_world(int orthographic, const float world[16], float aspect, float fovy_rad,
                                           float near_z, float far_z, float view[16], float projection[16],
                                           float view_projection[16]) {
    float t[3], q[4], s[3], euler[3];
    const int ok = yh_mat4_decompose(world, t, q, s);
    yh_quat_to_euler(q, euler);
    yh_camera_update_matrices(orthographic, t, euler, aspect, fovy_rad, near_z, far_z, view, projection, view_projection);
    return ok;
}

// Gribb-Hartmann, depth 0..1: L,R,B,T,N,F, each divided by |n|.
extern "C" void yh_extract_planes(const float vp[16], float planes_out[24]) {
    float p[6][4];
    for (int c = 0; c < 4; ++c) {
        const float r0 = vp[c * 4 + 0], r1 = vp[c * 4 + 1], r2 = vp[c * 4 + 2], r3 = vp[c * 4 + 3];
        p[0][c] = r3 + r0;
        p[1][c] = r3 - r0;
        p[2][c] = r3 + r1;
        p[3][c] = r3 - r1;
        p[4][c] = r2;
        p[5][c] = r3 - r2;
    }
    for (int i = 0; i < 6; ++i) {
        const float len = sqrtf(p[i][0] * p[i][0] + p[i][1] * p[i][1] + p[i][2] * p[i][2]);
        for (int c = 0; c < 4; ++c) planes_out[i * 4 + c] = p[i][c] / len;
    }
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

// ------------------------------------------------------------------------------------------- sharding
extern "C" int yh_partition_subtrees(const uint32_t* parent, const uint32_t* lo, uint32_t n_levels, uint32_t n_parts,
                                     uint32_t* part_of, uint64_t* part_sizes) {
    if (!parent || !lo || !part_of || !part_sizes) return fail(nullptr, YH_ERR_BAD_ARG, "yh_partition_subtrees: NULL argument");
    if (n_levels < 1 || n_levels > YH_MAX_LEVELS || n_parts < 1) return fail(nullptr, YH_ERR_BAD_ARG, "yh_partition_subtrees: bad n_levels / n_parts");
    const uint32_t n = lo[n_levels], n0 = lo[1];
    // root_of[e] = level-0 ancestor; parents precede children, so one forward sweep is enough
    std::vector<uint32_t> root_of(n);
    std::vector<uint64_t> subtree(n0, 0);
    for (uint32_t e = 0; e < n0; ++e) root_of[e] = e;
    for (uint32_t k = 1; k < n_levels; ++k)
        for (uint32_t e = lo[k]; e < lo[k + 1]; ++e) {
            const uint32_t p = parent[e];
            if (p < lo[k - 1] || p >= lo[k]) return fail(nullptr, YH_ERR_BAD_ARG, "yh_partition_subtrees: entity %u has a parent outside the previous level", e);
            root_of[e] = root_of[p];
        }
    for (uint32_t e = 0; e < n; ++e) subtree[root_of[e]]++;
    // greedy largest-first bin packing (ties: lower slot first, so the result is deterministic)
    std::vector<uint32_t> order(n0);
    for (uint32_t i = 0; i < n0; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return subtree[a] > subtree[b]; });
    std::vector<uint64_t> load(n_parts, 0);
    std::vector<uint32_t> part_of_root(n0);
    for (uint32_t r : order) {
        uint32_t best = 0;
        for (uint32_t p = 1; p < n_parts; ++p)
            if (load[p] < load[best]) best = p;
        part_of_root[r] = best;
        load[best] += subtree[r];
    }
    for (uint32_t e = 0; e < n; ++e) part_of[e] = part_of_root[root_of[e]];
    for (uint32_t p = 0; p < n_parts; ++p) part_sizes[p] = load[p];
    return YH_OK;
}

extern "C" int yh_shard_extract(const uint32_t* parent, const uint32_t* lo, uint32_t n_levels, const uint32_t* part_of,
                                uint32_t part, uint32_t* local_parent, uint32_t* local_lo, uint32_t* global_of_local) {
    if (!parent || !lo || !part_of || !local_parent || !local_lo || !global_of_local)
        return fail(nullptr, YH_ERR_BAD_ARG, "yh_shard_extract: NULL argument");
    if (n_levels < 1 || n_levels > YH_MAX_LEVELS) return fail(nullptr, YH_ERR_BAD_ARG, "yh_shard_extract: bad n_levels");
    const uint32_t n = lo[n_levels];
    std::vector<uint32_t> local_of(n, YH_NO_PARENT);
    uint32_t next = 0;
    local_lo[0] = 0;
    for (uint32_t k = 0; k < n_levels; ++k) {
        for (uint32_t e = lo[k]; e < lo[k + 1]; ++e) {
            if (part_of[e] != part) continue;
            local_of[e] = next;
            global_of_local[next] = e;
            if (k == 0) {
                local_parent[next] = YH_NO_PARENT;
            } else {
                const uint32_t lp = local_of[parent[e]];
                if (lp == YH_NO_PARENT) return fail(nullptr, YH_ERR_BAD_ARG, "yh_shard_extract: entity %u and its parent are in different parts", e);
                local_parent[next] = lp;  // global sibling order is kept, so siblings stay sorted by parent
            }
            ++next;
        }
        local_lo[k + 1] = next;
    }
    return YH_OK;
}

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
