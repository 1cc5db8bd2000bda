/*
 * yume_hip.h — C ABI of the MI355X (gfx950) transform-and-visibility path for the Yume engine.
 *
 * This is the drop-in boundary: everything the engine-side FFI module (`engine/hip/hip.zig`,
 * `extern fn` declarations, one per symbol below) binds.  Plain C: opaque handles, plain pointers
 * and sizes, `int` status codes, no exceptions, no C++/torch types.
 *
 * What each entry point replaces in the reference (paths under the reference tree):
 *   - the per-frame system  engine/systems/TransformSyncSystem.zig:53-74  (rule A3)
 *     + engine/systems/PostTransformSystem.zig:27-37 (dirty reset, A7)         -> yh_frame
 *   - Mat4.compose / fromQuat  engine/math3d.zig:816-822, 727-749 (A5)          -> yh_frame (fused), yh_compose
 *   - Mat4.mul  engine/math3d.zig:583-610 (A4), Mat4.inverse :763-814 (A8)      -> inside yh_frame
 *   - column access  engine/ecs.zig:695-705 (`ecs.field`, A9) and the 68-byte
 *     {Mat4, bool dirty} component layout engine/components/transform.zig:8-12,62-66 (A1)
 *                                                   -> yh_upload_*_aos68 / yh_download_*_aos68
 *   - hierarchy source of truth (ChildOf pairs) engine/ecs.zig:328-343,470-477 (A10)
 *                                                   -> yh_upload_hierarchy
 *   - culling has no counterpart in the reference: the OBB-vs-6-plane test is
 *     defined by this build; bounds come from engine/meshes.zig:72-95 (BoundingBox{mins,maxs}),
 *     frusta from engine/components/camera.zig:253-302 (view_projection).
 *
 * Conventions: Mat4 is 16 x f32 column-major, values[c*4+r] (engine/math3d.zig:529-549).
 * All arithmetic is IEEE f32, unfused, left-to-right, exactly as the reference evaluates it.
 *
 * Ownership: the library owns all device memory and streams.  Host pointers are borrowed for
 * the duration of the call only.  Threading: one host thread per yh_ctx at a time.
 * Errors: 0 = ok, negative yh_status otherwise; yh_last_error() gives a message.  Nothing
 * aborts or throws across this boundary.
 */
#ifndef YUME_HIP_H
#define YUME_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YH_ABI_VERSION 1u
#define YH_MAX_FRUSTA 8u   /* frusta per frame (main camera + shadow cascades) */
#define YH_MAX_LEVELS 64u  /* hierarchy depth limit */
#define YH_NO_PARENT 0xFFFFFFFFu

typedef enum yh_status {
    YH_OK = 0,
    YH_ERR_BAD_ARG = -1,   /* null pointer, range out of bounds, malformed hierarchy */
    YH_ERR_OOM = -2,       /* host or device allocation failed */
    YH_ERR_HIP = -3,       /* a HIP runtime call failed */
    YH_ERR_RCCL = -4,      /* RCCL missing or a collective failed */
    YH_ERR_CAPACITY = -5,  /* a draw list or caller buffer is too small (required size reported) */
    YH_ERR_STATE = -6,     /* call order violated (e.g. frame before hierarchy upload) */
    YH_ERR_UNSUPPORTED = -7
} yh_status;

typedef struct yh_ctx yh_ctx;     /* one GPU: device, stream, events, optional RCCL communicator */
typedef struct yh_scene yh_scene; /* device-resident entity arrays of one (shard of a) scene   */

/* How local transforms are held on the device. */
typedef enum yh_local_kind {
    YH_LOCAL_TRS = 0, /* 10 f32 SoA planes tx,ty,tz,qx,qy,qz,qw,sx,sy,sz; stage 1 = batched Mat4.compose */
    YH_LOCAL_MAT4 = 1 /* 16 f32 SoA planes (imported flecs LocalTransform column); carries dirty flags     */
} yh_local_kind;

/* Where local-space bounds come from. */
typedef enum yh_bounds_kind {
    YH_BOUNDS_NONE = 0,       /* no culling inputs: transforms only                               */
    YH_BOUNDS_PER_ENTITY = 1, /* 6 f32 SoA planes min.xyz, max.xyz                                */
    YH_BOUNDS_MESH_TABLE = 2  /* u32 mesh id per entity + table of n_meshes x {min.xyz, max.xyz}  */
} yh_bounds_kind;

typedef struct yh_scene_desc {
    uint64_t n_entities;      /* 1 .. 2^30 (kernels address planes with 32-bit byte offsets) */
    uint32_t n_levels;        /* hierarchy depth, 1 = flat; <= YH_MAX_LEVELS */
    uint32_t n_frusta;        /* draw lists kept per frame; <= YH_MAX_FRUSTA */
    uint32_t local_kind;      /* yh_local_kind  */
    uint32_t bounds_kind;     /* yh_bounds_kind */
    uint32_t max_meshes;      /* YH_BOUNDS_MESH_TABLE only */
    uint32_t id_base;         /* added to every emitted draw index (shard_base of this GPU) */
    uint64_t draw_capacity;   /* entries per draw list; 0 = n_entities */
} yh_scene_desc;

/* yh_frame flags */
#define YH_FRAME_ALL_DIRTY   0x1u /* every LocalTransform is dirty: forward branch everywhere (benchmark mode;
                                     the only mode of YH_LOCAL_TRS scenes)                                   */
#define YH_FRAME_NO_CULL     0x2u /* skip stage 3 */
#define YH_FRAME_ASYNC       0x4u /* enqueue only; yh_wait() or any yh_get_ / yh_download_ call synchronises.  Small TRS
                                     scenes (<= ~1M entities) leave the frame's list expansion to the next frame's first
                                     launch, or to the first call that reads the lists: results are the same */
#define YH_FRAME_ORDERED     0x8u /* accepted, no effect: draw lists are ALWAYS in ascending entity order        */
#define YH_FRAME_LISTS_ON_DEMAND 0x10u /* with YH_FRAME_ASYNC: this frame's own draw lists are expanded only if somebody
                                     reads them (counts, a list, a consumer, yh_wait) before the next frame starts; the
                                     next frame otherwise drops them unexpanded.  For frames whose visibility words go
                                     to yh_allgather_draw_lists_async, whose gathered lists already hold this GPU's part */
#define YH_FRAME_CULL_ONLY 0x20u /* cull what the device holds: one launch tests the world matrices the last frame left
                                     (and the bounds as they are now) against these frusta and writes the lists; no
                                     rule, no sweep.  For frames in which nothing moved but a camera did (GameWindow.zig:
                                     154-178) or a mesh changed, and for further frusta over the same frame's matrices.
                                     Rows uploaded or marked dirty since the last frame, a dirty root row included, stay
                                     pending for the next full frame; with change tracking on this frame reports no changed row.  Not
                                     before the first frame (YH_ERR_STATE), not with YH_FRAME_ALL_DIRTY /
                                     _LISTS_ON_DEMAND (YH_ERR_BAD_ARG); yh_frame_static takes it too              */

typedef struct yh_timing {
    float last_frame_ms;    /* device time of the most recent frame (hipEvent; 0 unless kernel timing is on) */
    float last_kernel_ms;   /* device time of its longest kernel launch                             */
    uint32_t n_launches;    /* kernel launches in that frame                                        */
    uint32_t n_groups;      /* level groups the hierarchy was cut into                              */
    uint64_t frames;        /* frames run on this scene                                             */
} yh_timing;

typedef struct yh_device_info {
    int device_id;
    int n_cus;
    int wavefront_size;
    uint64_t hbm_bytes;
    char arch[32];  /* "gfx950" */
    char name[128];
} yh_device_info;

/* ---- context ------------------------------------------------------------------------------------------- */
uint32_t yh_abi_version(void);
int yh_init(int device_id, yh_ctx** out);                 /* GameApp.zig:50-66 analogue: once at start-up */
void yh_shutdown(yh_ctx* ctx);
const char* yh_last_error(const yh_ctx* ctx);             /* ctx may be NULL: last error of this thread   */
int yh_device_info_get(yh_ctx* ctx, yh_device_info* out);
int yh_wait(yh_ctx* ctx);                                 /* drain the context's stream                    */
void* yh_stream(yh_ctx* ctx);                             /* the hipStream_t all work is enqueued on       */
/* Page-locked host memory for staging columns (the engine-side system copies flecs table rows into one depth-sorted
 * column before yh_upload_*_aos68, ecs.zig:695-705 hands out per-table pointers only): copies to and from it run at
 * link speed.  NULL when it cannot be had (yh_last_error); ordinary memory works everywhere, slower.            */
void* yh_host_alloc(yh_ctx* ctx, uint64_t bytes);
void yh_host_free(yh_ctx* ctx, void* p);

/* ---- scene --------------------------------------------------------------------------------------------- */
int yh_scene_create(yh_ctx* ctx, const yh_scene_desc* desc, yh_scene** out);
void yh_scene_destroy(yh_scene* scene);

/* Depth-sorted hierarchy (A10): entities are stored level-major, siblings contiguous and sorted by parent.
 * parent_index[e] is the slot of e's parent (an entity of the previous level) or YH_NO_PARENT for level 0,
 * whose parent is the scene root (GameApp.zig:77,95).  level_offsets has n_levels+1 entries.           */
int yh_upload_hierarchy(yh_scene* s, const uint32_t* parent_index, const uint32_t* level_offsets);
/* A hierarchy that changes a little (entities created, deleted or re-parented between frames: editor/Editor.zig:590-636,
 * GameApp.zig:204-215) without a new yh_upload_hierarchy: the host gives every level spare slots, marks the slots
 * that hold no entity, and moves the rows of a table that changed to free slots of its level.
 * yh_upload_parent_range: parent indices of slots [first, first + count), all of ONE level, every parent inside the
 * level before it (level 0: values unused); level layout and launch plan stay as uploaded.
 * yh_scene_set_alive_range (Mat4 scenes): a slot that is not alive is never listed as visible, never counted as a
 * boundary case, never reported as changed; its rows are computed like any other, so it needs a valid parent index.
 * All slots are alive until the first call.                                                                   */
int yh_upload_parent_range(yh_scene* s, const uint32_t* parent_index, uint64_t first, uint64_t count);
int yh_scene_set_alive_range(yh_scene* s, uint64_t first, uint64_t count, int alive);
/* The same for many short runs or single slots in ONE call each (a burst of editor traffic touches thousands of flecs
 * tables, Editor.zig:590-636; one blocking call per table and array cost ~0.15 ms a table).  Runs are applied in the order
 * given.  yh_upload_parent_rows: slot slots[i] gets parent parent_index[i] (any levels; each parent inside the level before
 * its slot's).  No counterpart in the reference: flecs keeps (ChildOf, parent) in the archetype.                        */
int yh_scene_set_alive_runs(yh_scene* s, const uint32_t* first, const uint32_t* count, const uint8_t* alive, uint32_t n_runs);
int yh_upload_parent_rows(yh_scene* s, const uint32_t* slots, const uint32_t* parent_index, uint64_t count);
/* WorldTransform of the scene root (GameApp.zig:95); identity until set. */
int yh_set_root(yh_scene* s, const float root_world[16]);
/* The same from the root's 68-byte WorldTransform row as the `Up` term hands it to level-0 tables
 * (TransformSyncSystem.zig:34-36, 47): matrix AND the row's own dirty byte, nothing implied.             */
int yh_upload_root_aos68(yh_scene* s, const void* row);

/* Stage-1 inputs: SoA TRS planes, order tx,ty,tz,qx,qy,qz,qw,sx,sy,sz (Mat4.compose arguments,
 * math3d.zig:816).  Entities [first, first+count).                                                       */
int yh_upload_trs(yh_scene* s, const float* const planes[10], uint64_t first, uint64_t count);
/* flecs table columns, 68-byte rows {float m[16]; uint8 dirty; pad[3]} (transform.zig:8-12, 62-66),
 * as returned by ecs.field (ecs.zig:695-705).  Scene must be YH_LOCAL_MAT4.                             */
int yh_upload_local_aos68(yh_scene* s, const void* column, uint64_t first, uint64_t count);
int yh_upload_world_aos68(yh_scene* s, const void* column, uint64_t first, uint64_t count);
/* The onSet hooks (transform.zig:46-51, 92-97): mark a range dirty without re-uploading matrices.       */
int yh_mark_dirty(yh_scene* s, int local, int world, uint64_t first, uint64_t count);

/* Ordering of uploads: every yh_upload_ call is ordered behind the frames already enqueued on this context
 * (YH_FRAME_ASYNC) — it copies on the context's stream, or waits for that stream first — and returns when its copy
 * is done: a frame never sees a half-written input and an upload never overtakes an earlier frame.  yh_set_root /
 * yh_upload_root_aos68 only change what the NEXT frame is handed by value.
 *
 * Bounds (meshes.zig:72-95).  Mesh-table scenes: every entity refers to mesh 0 until yh_upload_mesh_ids says
 * otherwise; the table may be re-uploaded with more or fewer meshes, but never fewer than the largest id in use
 * (YH_ERR_STATE) — upload the ids of the whole scene again first.                                          */
int yh_upload_aabb(yh_scene* s, const float* const min_planes[3], const float* const max_planes[3],
                   uint64_t first, uint64_t count);
int yh_upload_mesh_aabbs(yh_scene* s, const float* table /* n_meshes x 6: min.xyz,max.xyz */, uint32_t n_meshes);
int yh_upload_mesh_ids(yh_scene* s, const uint32_t* mesh_id, uint64_t first, uint64_t count);
/* Single slots: slot slots[i] gets mesh_id[i] / the box min_max[6 i .. 6 i + 5] = {min.xyz, max.xyz} (meshes.zig:72-95).   */
int yh_upload_mesh_id_rows(yh_scene* s, const uint32_t* slots, const uint32_t* mesh_id, uint64_t count);
int yh_upload_aabb_rows(yh_scene* s, const uint32_t* slots, const float* min_max, uint64_t count);

/* ---- per frame ----------------------------------------------------------------------------------------- */
/* One pass of the hot path over the whole scene: compose -> level-ordered propagate -> cull + compact,
 * then dirty reset.  planes: n_frusta x 6 x 4 floats (nx,ny,nz,d), normalised, order L,R,B,T,N,F
 * (yh_extract_planes).  Replaces one `world.progress` worth of TransformSyncSystem + PostTransformSystem
 * callbacks (Editor.zig:301-306).                                                                        */
int yh_frame(yh_scene* s, const float* planes, uint32_t n_frusta, uint32_t flags);

/* yh_frame for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists for hosts that may not call
 * hipMalloc around the frame — code under HIP graph capture, tight serving loops, harnesses that forbid allocation.
 * Same arguments, same launches and the same results as yh_frame (world matrices, draw lists, counts and boundary
 * counts land in the scene's buffers and are read with the same yh_get_ / yh_download_ calls); the frame's stage-3
 * scratch (boundary counters, tile counts, visibility words, their byte counts) lies in a file-scope device array of
 * YH_STATIC_SCRATCH_BYTES that every GPU the library runs on holds until the process ends, so the call makes no
 * hipMalloc / hipFree / event creation (kernel-timing events are not recorded on this path, nothing is deferred to a
 * later call).  Limit: n_frusta x (9 bytes per 64 entities + 4 per expand tile and per 262144 entities) + 144 <= YH_STATIC_SCRATCH_BYTES,
 * about 119M entity-frusta (16M entities x 7 frusta); beyond it YH_ERR_CAPACITY and nothing is launched.  All calls on
 * one GPU share that GPU's array (it keeps the last call's contents; the call clears what must start at zero): calls
 * on ONE context are ordered by its stream, calls on two contexts of the same GPU that overlap RACE.  The words of
 * such a frame cannot be sent by yh_allgather_draw_lists_async afterwards (YH_ERR_STATE); the synchronous
 * yh_allgather_draw_lists works.                                                                            */
#define YH_STATIC_SCRATCH_BYTES (16u << 20)
int yh_frame_static(yh_scene* s, const float* planes, uint32_t n_frusta, uint32_t flags);

/* Results.  counts/boundary: n_frusta entries.  A "boundary" entity has |s + r| < 1e-5 for some plane. */
int yh_get_counts(yh_scene* s, uint32_t* counts, uint32_t* boundary_counts);
/* Copies draw list `frustum` to `out`; *written = entries of the list.  If cap is too small returns
 * YH_ERR_CAPACITY with *written = required size.                                                         */
int yh_get_draw_list(yh_scene* s, uint32_t frustum, uint32_t* out, uint64_t cap, uint64_t* written);
/* World matrices: 16 SoA planes (plane p = values[p]) or flecs 68-byte rows (dirty bytes included).     */
int yh_download_world_soa(yh_scene* s, float* const planes[16], uint64_t first, uint64_t count);
int yh_download_world_aos68(yh_scene* s, void* column, uint64_t first, uint64_t count);
int yh_download_local_aos68(yh_scene* s, void* column, uint64_t first, uint64_t count);
/* ---- sparse rows and change tracking (Mat4 scenes) -------------------------------------------
 * The onSet hooks (transform.zig:46-51, 92-97) are the only writers of `dirty`, so the rows a host changed since
 * the last frame are known to it: yh_upload_*_rows_aos68 sends just those (row r belongs to entity slots[r]).
 * With yh_scene_track_changes on, every yh_frame also records which rows it recomputed or found edited (dirty
 * before the PostTransformSystem sweep); yh_download_changed_aos68 returns their slots (ascending) and both rows,
 * dirty bytes cleared — what the CPU systems would have left in those rows; all other rows are untouched by the
 * frame.  cap too small: YH_ERR_CAPACITY with *written = rows needed.  Any of the three outputs may be NULL.  */
int yh_upload_local_rows_aos68(yh_scene* s, const uint32_t* slots, const void* rows, uint64_t count);
int yh_upload_world_rows_aos68(yh_scene* s, const uint32_t* slots, const void* rows, uint64_t count);
/* Rows that change slots on the device: both rows (matrices and dirty bytes) of src_slots[i] become the rows of
 * dst_slots[i]; every source is read before any destination is written, so the lists may overlap.  For a host whose own
 * copies are older than the scene's (frames ran with the results left on the device, §8f N2) and whose tables are now
 * re-ordered - the GPU twin of flecs moving a row between tables on re-parenting (ecs.zig:470-477).               */
int yh_move_rows(yh_scene* s, const uint32_t* src_slots, const uint32_t* dst_slots, uint64_t count);
/* enable: 0 off; YH_TRACK_ROWS: the two lists as described below; YH_TRACK_REWRITTEN_LOCALS: the LocalTransform list
 * (which = 1) holds only the rows whose MATRIX the frame rewrote (reverse branch) - for a host that clears the dirty byte
 * of every LocalTransform row at the moment it sends it (the frame would leave exactly that row: same matrix, flag
 * cleared), so that frames in which gameplay sets local transforms bring no LocalTransform row back at all. */
#define YH_TRACK_ROWS 1
#define YH_TRACK_REWRITTEN_LOCALS 2
int yh_scene_track_changes(yh_scene* s, int enable);
int yh_get_changed_count(yh_scene* s, uint64_t* n_changed);
int yh_download_changed_aos68(yh_scene* s, uint32_t* slots, void* world_rows, void* local_rows, uint64_t cap,
                              uint64_t* written);
/* The same rows as two lists, one per component, each fetched on its own and STREAMED (what a host that copies the
 * rows into many small tables wants: GpuTransformCullSystem.c).  A tracked frame files every row it touched under
 *   which = 0: WorldTransform changed (recomputed, or its dirty flag was set) - the list of yh_download_changed_aos68;
 *   which = 1: LocalTransform changed (the reverse branch re-derived it, TransformSyncSystem.zig:69-72, or its dirty
 *              flag was set).  A parent that moves changes every WorldTransform below it and no LocalTransform: half
 *              the bytes of yh_download_changed_aos68 stay on the device.
 * yh_download_changed_rows_aos68 copies list `which` (slots ascending, may be NULL) and its 68-byte rows, dirty bytes
 * cleared.  The rows arrive in up to 16 pieces per 4M rows, back to back on the link; after each piece
 * `ready(user, rows_ready)` is called ON THE CALLING THREAD (NULL: no calls): rows [0, rows_ready) and all slots are
 * in place, so helper threads of the caller can place them while the next piece is in flight.  Page-locked `rows`
 * (yh_host_alloc) keep the copies asynchronous; ordinary memory works, each piece then completes inside its copy.
 * cap too small: YH_ERR_CAPACITY with *written = rows needed, nothing copied.                                    */
typedef void (*yh_rows_ready_fn)(void* user, uint64_t rows_ready);
int yh_get_changed_counts(yh_scene* s, uint64_t* n_world, uint64_t* n_local);
int yh_download_changed_rows_aos68(yh_scene* s, int which, uint32_t* slots, void* rows, uint64_t cap, uint64_t* written,
                                   yh_rows_ready_fn ready, void* user);
/* ---- draw-list consumer ------------------------------------------------------------------------
 * What the reference's draw loop does on the CPU for every object it is handed, here on the device for the visible
 * entities of one frustum, in list order i = 0 .. count-1:
 *   object_data[i] = GPUObjectData{ model_matrix = WorldTransform.matrix }   (GAL/vulkan/RenderApi.zig:97-99, 1397-1404)
 *   cmd[i] = { vertices_count, 1, 0, batch_offset + i }                      (vkCmdDraw arguments, :1441)
 * yh_upload_vertex_counts: mesh.vertices_count per mesh (mesh-table scenes, n <= max_meshes) or per entity
 * (n = n_entities); without it vertex_count is 0.  yh_build_object_data enqueues after yh_frame (no host sync);
 * yh_get_object_data copies to the host; yh_object_data_device hands out the device buffers for an importing
 * renderer (valid until the scene is destroyed; contents until the next build).                            */
typedef struct yh_draw_cmd {
    uint32_t vertex_count, instance_count, first_vertex, first_instance; /* VkDrawIndirectCommand */
} yh_draw_cmd;
int yh_upload_vertex_counts(yh_scene* s, const uint32_t* counts, uint64_t n);
int yh_build_object_data(yh_scene* s, uint32_t frustum, uint32_t batch_offset);
int yh_get_object_data(yh_scene* s, uint32_t frustum, void* model_matrices, yh_draw_cmd* cmds, uint64_t cap,
                       uint64_t* written);
void* yh_object_data_device(yh_scene* s, uint32_t frustum, void** cmds_device);
/* Per-mesh bucketing (mesh-table scenes, n_meshes <= 65536: up to 256 keys the tiles of the list are
 * sorted inside LDS, beyond that every wave owns a slice of the list and the counters live in device memory).  The reference's draw loop rebinds the
 * vertex buffer whenever the mesh differs from the previous object's (RenderApi.zig:1434; the pipeline likewise when
 * the material does, :1411): yh_build_mesh_buckets orders the frustum's visible entities by mesh id — ascending
 * mesh, ascending entity inside a mesh (a stable counting sort of the draw list, on the device) — so that loop binds
 * each mesh once.  Object data and draw records (yh_get_object_data / yh_object_data_device) are then in bucket
 * order, record i still {vertices_count, 1, 0, batch_offset + i} (simple.vert:29 indexes objects by gl_BaseInstance,
 * so one record per object stays), and bucket m = {m, first, count} says which records [first, first + count) draw
 * mesh m: one vkCmdDrawIndirect(count, stride 16) per bound mesh.  Material buckets (the pipeline rebinding rule,
 * RenderApi.zig:1411): a key is a mesh-table ENTRY, so a host gives each (material, mesh) pair its own entry, ordered
 * by material first - GpuTransformCullSystem_setMaterialComponent does exactly that from the ECS.  yh_get_mesh_buckets: buckets (n_meshes entries) and the
 * bucket-ordered global entity ids; either may be NULL.                                                            */
typedef struct yh_bucket {
    uint32_t mesh, first, count;
} yh_bucket;
int yh_build_mesh_buckets(yh_scene* s, uint32_t frustum, uint32_t batch_offset);
int yh_get_mesh_buckets(yh_scene* s, uint32_t frustum, yh_bucket* buckets, uint32_t cap_buckets, uint32_t* entities,
                        uint64_t cap, uint64_t* written);
/* Batched Mat4.decompose (math3d.zig:828-981) of the world (which = 0) or local (which = 1, Mat4
 * scenes) matrices: what the position() / rotation() / scale() accessors compute one row at a time
 * (components/transform.zig:14-28, 68-82).  planes: t.xyz, q.xyzw, s.xyz on the host; valid (optional): 0 where the
 * reference returns error.InvalidMatrix and the accessors substitute Decomposed.IDENTITY, which is what is written. */
int yh_decompose(yh_scene* s, int which, float* const planes[10], uint8_t* valid, uint64_t first, uint64_t count);
/* Stage 1 alone: Mat4.compose over a range, result as 16 SoA planes on the host (testing aid).          */
int yh_compose(yh_scene* s, float* const planes[16], uint64_t first, uint64_t count);

int yh_timings(yh_scene* s, yh_timing* out);
/* Measurement harness (no reference counterpart): HIP events around every launch of yh_frame, on the
 * context's stream.  yh_scene_get_groups writes up to cap (first_level, last_level, first_entity,
 * end_entity) quadruples, one per launch; yh_scene_get_group_timings the mean device time per launch
 * since the previous call (at most 2048 launches are recorded between calls).  enable: 0 = off, 1 = every
 * launch plus the frame, 2 + i = only launch i of the frame, -1 = every transform launch of a TRS scene and nothing
 * else.  A transform launch of a TRS scene carries its event pair itself (hipExtLaunchKernelGGL: the dispatch's
 * own start/end timestamps, nothing added to the stream); everything else is bracketed by recorded events.   */
int yh_scene_set_kernel_timing(yh_scene* s, int enable);
/* Sample: record the events on every period-th frame only (an event record stalls the stream for ~3 us). */
int yh_scene_set_kernel_timing_period(yh_scene* s, uint32_t period);
int yh_scene_get_groups(yh_scene* s, uint32_t* quads, uint32_t cap, uint32_t* n_groups);
int yh_scene_get_group_timings(yh_scene* s, float* mean_ms, uint32_t* samples, uint32_t cap);
/* The kernel each group's launch of the LAST frame ran, spelled as rocprofv3's kernel trace spells it (one per
 * line, group order): the harness looks a launch's counters up by the name the library really dispatched.   */
int yh_scene_get_group_kernel_names(yh_scene* s, char* buf, uint64_t cap, uint32_t* n_groups);
/* Measurement harness (no reference counterpart): mean device time of `launches` back-to-back launches of a plain
 * stream kernel that reads read_bytes and writes write_bytes (16-byte accesses, ordinary loads and stores) of buffers
 * that stay the same from launch to launch - so a working set that fits the 256 MB Infinity Cache is served from it.
 * bench.py divides a cache-resident workload's rate by what this reaches on a working set of the same size and mix
 * (`roofline.cache_ceiling`): the HBM peak is no yardstick there.  Allocates and frees its buffers; synchronises. */
int yh_stream_probe(yh_ctx* ctx, uint64_t read_bytes, uint64_t write_bytes, uint32_t launches, float* mean_ms);
/* Hierarchy launches (TRS scenes): levels are cut into groups of consecutive levels, one kernel launch per
 * group; inside a group an entity re-derives its ancestors' matrices (a walk up the parent chain) instead of
 * waiting for a launch boundary.  Results do not depend on the plan.  yh_scene_set_grouping: exactly
 * max_group_levels (<= 8) levels per launch, 1 = one launch per level, 0 = the planner decides.
 * yh_scene_set_plan: explicit plan, group i = levels (last_level[i-1], last_level[i]], at most 8 each;
 * n_groups = 0 returns to the planner.                                                                   */
int yh_scene_set_grouping(yh_scene* s, uint32_t max_group_levels);
int yh_scene_set_plan(yh_scene* s, uint32_t n_groups, const uint32_t* last_level);
/* One-level launches of a TRS scene (a group of exactly one level > 0) stage the workgroup's span of parent rows in
 * LDS instead of gathering them through L1/L2.  Same results; off by default (measured slower or equal on the
 * benchmark trees, profiles/r02/ab_lds_parents.txt) — kept selectable for hierarchies with many children per parent. */
int yh_scene_set_lds_parents(yh_scene* s, int enable);

/* ---- frustum helpers (host, scalar; same arithmetic as the kernels' inputs expect) --------------------- */
/* Gribb-Hartmann planes from a column-major view-projection matrix (camera.zig:300-302; depth 0..1 as
 * math3d.zig:649-659).  Writes 6 x 4 floats, normalised.                                                 */
void yh_extract_planes(const float view_projection[16], float planes_out[24]);
void yh_mat4_mul(const float a[16], const float b[16], float out[16]);             /* math3d.zig:583-610 */
void yh_perspective(float fovy_rad, float aspect, float near_z, float far_z, float out[16]); /* :649-659 */
void yh_orthographic(float l, float r, float b, float t, float n, float f, float out[16]);   /* :661-683 */
void yh_look_at(const float eye[3], const float at[3], const float up[3], float out[16]);    /* :991-1021 */
/* Camera.view_projection as the engine updates it (components/camera.zig:253-302): view =
 * Rx*Ry*Rz*T(-pos) (Mat4.rotation, math3d.zig:687-710), perspective: rot in degrees, projection Y-flipped;
 * orthographic: rot in radians, the engine's fixed +-10 box, depth 0..200.  Feed the result to yh_extract_planes. */
void yh_camera_view_projection(int orthographic, const float pos[3], const float rot[3], float aspect, float fovy_rad,
                               float near_z, float far_z, float view_projection_out[16]);
/* The whole of Camera.updateMatrices (camera.zig:161-166): view and projection too (either may be NULL). */
void yh_camera_update_matrices(int orthographic, const float pos[3], const float rot[3], float aspect, float fovy_rad,
                               float near_z, float far_z, float view_out[16], float projection_out[16],
                               float view_projection_out[16]);
/* The per-camera update the engine runs before drawing with a Camera entity (editor/windows/GameWindow.zig:160-161;
 * SURVEY 8f N4 "camera matrix pipeline"): world.decompose() -> rotation.toEuler() -> updateMatrices.
 * yh_mat4_decompose: `matrix.decompose() catch Decomposed.IDENTITY` of one matrix (components/transform.zig:68-70,
 * math3d.zig:828-981), t.xyz / q.xyzw / s.xyz; returns 0 where IDENTITY was substituted, else 1.
 * yh_quat_to_euler: Quat.toEuler (math3d.zig:418-442), degrees; atan2 / asin are the host libm's ("parity unpinned":
 * Zig's own routines are not in this image).  yh_camera_update_from_world chains the three; returns what
 * yh_mat4_decompose does.                                                                                       */
int yh_mat4_decompose(const float m[16], float t[3], float q[4], float s[3]);
void yh_quat_to_euler(const float q[4], float euler_deg[3]);
int yh_camera_update_from_world(int orthographic, const float world[16], float aspect, float fovy_rad, float near_z,
                                float far_z, float view_out[16], float projection_out[16], float view_projection_out[16]);

/* ---- synthetic scenes (measurement harness) ---------------------------------------------- */
typedef struct yh_synth_desc {
    uint64_t seed;         /* 0x59554D45 in the benchmarks */
    uint64_t index_base;   /* global index of local entity 0 (so shards of one scene stay consistent) */
} yh_synth_desc;
/* Fills TRS (and bounds) planes of the scene on the device from the counter-based generator; the
 * hierarchy must have been uploaded (level 0 and deeper levels draw from different ranges).              */
int yh_scene_fill_synthetic(yh_scene* s, const yh_synth_desc* desc);
/* The same for one shard of a partitioned scene: local entity e draws the values of global entity
 * index_base + global_of_local[e] (yh_shard_extract's map), so the shards together hold exactly the unsharded scene. */
int yh_scene_fill_synthetic_shard(yh_scene* s, const yh_synth_desc* desc, const uint32_t* global_of_local);
int yh_download_trs(yh_scene* s, float* const planes[10], uint64_t first, uint64_t count);
int yh_download_aabb(yh_scene* s, float* const min_planes[3], float* const max_planes[3], uint64_t first,
                     uint64_t count);

/* ---- sharding (host only) ------------------------------------------------------------------ */
/* Subtree-closed partition of a depth-sorted hierarchy over n_parts GPUs: every level-0 entity's whole subtree
 * goes to one part (greedy largest-first bin packing on subtree sizes), so no parent is ever read across GPUs.
 * part_of_entity: n entries; part_sizes: n_parts entries.  Replaces nothing in the reference (single process).  */
int yh_partition_subtrees(const uint32_t* parent_index, const uint32_t* level_offsets, uint32_t n_levels,
                          uint32_t n_parts, uint32_t* part_of_entity, uint64_t* part_sizes);
/* Builds shard `part`: its own depth-sorted arrays (level-major, sibling order kept, parents re-indexed) and the
 * map back to global slots.  local_parent / global_of_local: part_sizes[part] entries; local_level_offsets:
 * n_levels + 1.                                                                                          */
int yh_shard_extract(const uint32_t* parent_index, const uint32_t* level_offsets, uint32_t n_levels,
                     const uint32_t* part_of_entity, uint32_t part, uint32_t* local_parent,
                     uint32_t* local_level_offsets, uint32_t* global_of_local);

/* ---- multi-GPU (one process per GPU; RCCL over xGMI) --------------------------------------------------- */
#define YH_UNIQUE_ID_BYTES 128
int yh_comm_unique_id(void* out_id /* YH_UNIQUE_ID_BYTES */);
int yh_comm_init(yh_ctx* ctx, int n_ranks, int rank, const void* unique_id);
void yh_comm_destroy(yh_ctx* ctx);
/* Which RCCL the library resolved ("rccl 2.x.y from /path/librccl.so.1"): RCCL is dlopen'ed by SONAME, so inside a
 * process that already holds one (torch bundles its own) the loader hands the same copy out again; two different
 * copies in one process would show here.  YUME_HIP_COMM_LOG=1 makes yh_comm_init print the same on stderr.  Loads
 * RCCL if it is not loaded yet (YH_ERR_RCCL when there is none).                                                 */
int yh_comm_describe(char* buf, uint64_t cap);
/* All-gather of per-rank visible counts, then of the compacted draw lists (exact sizes, written at
 * prefix-sum offsets into a library-owned gathered list per frustum).  all_counts: n_ranks x n_frusta.   */
int yh_allgather_draw_lists(yh_scene* s, uint32_t* all_counts /* may be NULL */);
/* The same result without a host round trip: everything is enqueued on the context's collective stream behind
 * this frame and runs beside the next frames' launches.  What crosses xGMI is the frame's visibility words (one bit
 * per entity and frustum: a fixed size, one ncclAllGather, 20-32x fewer bytes than the lists); every rank expands
 * all ranks' words into the gathered lists itself, rank order, each rank's id_base added.  The gathered lists hold
 * every visible entity even where a rank's own draw_capacity was too small for its list.  The first call on a scene
 * synchronises once (ranks exchange word counts and id bases; every rank must have created the scene with the same
 * n_frusta).  Counts and lists are read afterwards with yh_get_gathered_counts / yh_get_gathered_draw_list (they
 * synchronise).                                                                                                  */
int yh_allgather_draw_lists_async(yh_scene* s);
/* Testing aid (one GPU, no communicator): runs the gather's expansion on fake_ranks copies of this scene's own
 * words, rank r owning ids id_base + r * n_entities; read the result like a real gather's.                      */
int yh_gather_selftest(yh_scene* s, uint32_t fake_ranks);
/* The same with n_shards DIFFERENT shards on one GPU: the scenes of one context stand for the ranks (rank order),
 * each with its own row length and id_base — e.g. one scene per part of yh_partition_subtrees, all framed with the
 * same frusta.  The gathered lists land in shards[0], as on rank 0 after a real exchange.                    */
int yh_gather_selftest_shards(yh_scene* const* shards, uint32_t n_shards);
int yh_get_gathered_counts(yh_scene* s, uint32_t* all_counts /* n_ranks x n_frusta */);
int yh_get_gathered_draw_list(yh_scene* s, uint32_t frustum, uint32_t* out, uint64_t cap, uint64_t* written);

#ifdef __cplusplus
}
#endif
#endif /* YUME_HIP_H */
