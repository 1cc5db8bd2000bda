// yh_frame.hip — one frame: the launch plan's kernels on the context's stream (yh_frame, yh_frame_static), the deferred
// expansion, results (counts, draw lists), per-launch timings.
#define YH_K_FRAME
#include "yh_internal.hpp"

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
void release_static(yh_ctx* ctx) {
    yh_ctx* me = ctx;
    if (ctx && ctx->device >= 0) g_static_owner[ctx->device % YH_MAX_DEVICES].compare_exchange_strong(me, nullptr);
}

int resolve_static_scratch(unsigned char** p) {
    *p = nullptr;
    hipError_t e = hipGetSymbolAddress((void**)p, HIP_SYMBOL(scratch));  // the caller made its device current
    if (e == hipSuccess && !*p) e = hipErrorInvalidSymbol;
    if (e != hipSuccess) return fail(nullptr, YH_ERR_HIP, "yh_init: the static scratch array could not be resolved: %s", hipGetErrorString(e));
    return YH_OK;
}

// Launches the expansion a YH_FRAME_ASYNC frame left for the next frame's first launch to carry (stream order keeps
// it behind that frame's kernels).  Called by everything that reads lists or totals, or synchronises.
int flush_expand(yh_scene* s) {
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

int sync_scene(yh_scene* s) {
    if (int rc = flush_expand(s)) return rc;
    HIP_TRY(s->ctx, hipStreamSynchronize(s->ctx->stream));
    if (s->ctx->stream_c) HIP_TRY(s->ctx, hipStreamSynchronize(s->ctx->stream_c));
    s->frame_pending = false;
    release_static(s->ctx);
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

template <int BOUNDS, bool SPEC>
static std::string launch_mat4_s(bool has_base, bool big, dim3 grid, hipStream_t st, const FrameArgs& a, uint32_t root_dirty) {
    if (has_base && big) hipLaunchKernelGGL((k_level_mat4<BOUNDS, true, true, SPEC>), grid, dim3(YH_BLOCK), 0, st, a, root_dirty);
    else if (has_base) hipLaunchKernelGGL((k_level_mat4<BOUNDS, true, false, SPEC>), grid, dim3(YH_BLOCK), 0, st, a, root_dirty);
    else if (big) hipLaunchKernelGGL((k_level_mat4<BOUNDS, false, true, SPEC>), grid, dim3(YH_BLOCK), 0, st, a, root_dirty);
    else hipLaunchKernelGGL((k_level_mat4<BOUNDS, false, false, SPEC>), grid, dim3(YH_BLOCK), 0, st, a, root_dirty);
    return kname("void k_level_mat4<%d, %s, %s, %s>(FrameArgs, unsigned int)", BOUNDS, tf(has_base), tf(big), tf(SPEC));
}
// spec: most rows are expected to be clean (k_level_mat4's SPEC)
template <int BOUNDS>
static std::string launch_mat4(bool has_base, bool big, bool spec, dim3 grid, hipStream_t st, const FrameArgs& a, uint32_t root_dirty) {
    if (spec) return launch_mat4_s<BOUNDS, true>(has_base, big, grid, st, a, root_dirty);
    return launch_mat4_s<BOUNDS, false>(has_base, big, grid, st, a, root_dirty);
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
    const bool cull_only = (flags & YH_FRAME_CULL_ONLY) != 0;
    if (cull_only) {
        if (flags & (YH_FRAME_ALL_DIRTY | YH_FRAME_LISTS_ON_DEMAND))
            return fail(ctx, YH_ERR_BAD_ARG, "%s: YH_FRAME_CULL_ONLY goes with neither YH_FRAME_ALL_DIRTY nor YH_FRAME_LISTS_ON_DEMAND", who);
        if (s->frames == 0) return fail(ctx, YH_ERR_STATE, "%s: YH_FRAME_CULL_ONLY needs the world matrices of an earlier frame", who);
    }
    if (s->desc.local_kind == YH_LOCAL_TRS && !(flags & YH_FRAME_ALL_DIRTY) && !cull_only)
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
    if (cull_only) {
        if (s->exp_pending)  // an expansion a deferred frame left behind: no first launch of a rule here to carry it
            if (int rc = flush_expand(s)) return rc;
        if (cull) {  // (without frusta there is nothing to do at all)
            a.begin = 0;
            a.end = (uint32_t)s->n;
            const dim3 grid((uint32_t)cdiv(s->n, YH_BLOCK));
            uint32_t* const chg = s->track ? s->h_chg_dev : nullptr;  // no row changed: both change lists are empty
            if (bounds == 1) hipLaunchKernelGGL((k_cull_only<1>), grid, dim3(YH_BLOCK), 0, st, a, chg);
            else hipLaunchKernelGGL((k_cull_only<2>), grid, dim3(YH_BLOCK), 0, st, a, chg);
            HIP_TRY(ctx, hipGetLastError());
            if (!s->group_kernel.empty()) s->group_kernel[0] = bounds == 1 ? "k_cull_only<1>" : "k_cull_only<2>";
            ++launches;
        }
        if (s->track) {
            if (!cull) HIP_TRY(ctx, hipMemsetAsync(s->h_chg_dev, 0, 2 * sizeof(uint32_t), st));
            s->chg_valid = true;
        }
    } else if (s->desc.local_kind == YH_LOCAL_TRS) {
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
                // a frame that is not flagged all-dirty and whose root did not move is expected to find most rows clean
                const bool spec = s->spec_clean && !(flags & YH_FRAME_ALL_DIRTY) && !s->root_dirty;
                if (bounds == 0) kn = launch_mat4<0>(has_base, big, spec, grid, st, a, s->root_dirty);
                else if (bounds == 1) kn = launch_mat4<1>(has_base, big, spec, grid, st, a, s->root_dirty);
                else kn = launch_mat4<2>(has_base, big, spec, grid, st, a, s->root_dirty);
            }
            HIP_TRY(ctx, hipGetLastError());
            s->group_kernel[gi] = kn;
            if (int rc = time_end()) return rc;
            ++launches;
        }
        // PostWorldTransform / PostLocalTransform (PostTransformSystem.zig:12-37)
        if (s->track) {
            hipLaunchKernelGGL(k_sweep_dirty, dim3(cdiv(s->stride, YH_BLOCK * 16ull)), dim3(YH_BLOCK), 0, st, s->ldirty, s->wdirty,
                               (uint32_t)s->n, s->chg_words, s->chg_cnt, s->n_chunks, s->cnt_stride, s->alive, s->track_rewritten_only ? 1u : 0u);
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
    if (!cull_only) s->root_dirty = 0;  // the world sweep clears the root's flag too
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

