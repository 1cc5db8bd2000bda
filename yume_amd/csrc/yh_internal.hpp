// yh_internal.hpp — what the translation units of libyume_hip.so share: the state behind the two opaque handles of
// include/yume_hip.h, the error helpers, and the few functions one unit calls in another.
//
//   yh_scene.hip      context, scenes, hierarchy and launch plan, uploads / downloads, sparse rows and change lists,
//                     stage 1 alone, batched decompose, synthetic scenes, the stream probe
//   yh_frame.hip      yh_frame / yh_frame_static (the launches of a frame), deferred expansion, results, timings
//   yh_consumers.hip  draw-list consumers: object data, draw records, buckets
//   yh_comm.hip       RCCL (dlopen'ed on the first yh_comm_init) and the gathered draw lists
//   yh_hostmath.hip   host-only helpers: planes, projections, the camera pipeline, sharding
// Nothing here is part of the ABI: the library exports yh_* only (csrc/exports.map).
#pragma once
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

#include "yh_kernels.hpp"  // argument structs and (per unit: YH_K_* before this header) the kernels

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
    bool track_rewritten_only = false;  // YH_TRACK_REWRITTEN_LOCALS: the LocalTransform list holds the reverse-branch rows only
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
    bool direct_rows = true;  // sparse rows to / from page-locked host memory go through the import / export kernel alone (YH_NO_DIRECT_ROWS=1: off)
    bool spec_clean = true;     // Mat4 level launches of frames expected to be mostly clean load the row's own data ahead of the flags (YH_NO_SPEC=1: A/B runs)
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

// ---- shared helpers ------------------------------------------------------------------------------------------------
int fail(yh_ctx* ctx, int code, const char* fmt, ...) __attribute__((format(printf, 3, 4)));  // records the message, returns code

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

// yh_frame.hip
int flush_expand(yh_scene* s);                  // launches a deferred expansion, if one is pending
int sync_scene(yh_scene* s);                    // flush_expand + both streams drained
void release_static(yh_ctx* ctx);               // the context no longer owns its GPU's static scratch (its stream was drained)
int resolve_static_scratch(unsigned char** p);  // device address of this GPU's copy of the static scratch (yh_init)
// yh_scene.hip
int check_range(yh_scene* s, const char* who, uint64_t first, uint64_t count);
int ensure_staging(yh_scene* s, uint64_t bytes);
