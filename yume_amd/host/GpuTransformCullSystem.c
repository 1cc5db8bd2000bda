/*
 * GpuTransformCullSystem.c — see GpuTransformCullSystem.h.  Plain C against the flecs C API and the yume_hip C ABI.
 *
 * Per frame (one call of the `run` callback, SystemDesc.run — ecs.zig:1209):
 *   1. walk every matched table once (ecs_iter_next) and note its columns, row count and the source entity of the
 *      `Up` term (ONE parent per table: ChildOf is part of the archetype);
 *   2. if the walk differs from the last one (entities created / deleted / re-parented), rebuild the depth-sorted
 *      slot order: a table's level is its parent's level + 1, tables whose parent is not itself matched (the scene
 *      root, GameApp.zig:77,95) are level 0; tables of one level are ordered by their parent's slot, which makes
 *      siblings contiguous and sorted by parent, as yh_upload_hierarchy wants (A10);
 *   3. plain mode: copy both 68-byte columns of every table into slot order, yh_upload_*_aos68, yh_frame,
 *      yh_download_*_aos68, copy back.  Incremental mode (default): after a rebuild the same upload once; otherwise
 *      only the rows whose OnSet fired since the last frame go up (yh_upload_*_rows_aos68: two observers collect them,
 *      the same events that run the components' onSet hooks, the only writers of `dirty`), and only the rows the
 *      frame recomputed or found dirty come back (yh_download_changed_aos68) — a frame in which nothing moved
 *      exchanges no rows at all.  The frame applies the three-branch rule (TransformSyncSystem.zig:53-74) level by
 *      level and then clears every dirty flag (PostTransformSystem.zig:27-37), the root's included.
 *   4. structural changes (observers, not polling): while no id was added to or removed from a table that holds one of
 *      the components, steps 1-2 are skipped altogether; a few tables that gained or lost rows are patched (patch():
 *      their rows move to spare slots of their level, no walk, no rebuild); anything else walks and rebuilds.
 *
 * Difference from the CPU systems worth knowing: the reference visits tables in flecs' storage order and
 * is parent-first only when entities were created parent-first; a child table stored before its parent's table reads
 * last frame's parent matrix there.  This system always runs parents first.
 */
#include "GpuTransformCullSystem.h"

#include <pthread.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#define ROW 68u
#define MAX_THREADS 32
#define MESH_TABLE_MAX 65536u /* distinct meshes (+ entities with bounds of their own) a mesh-table scene can hold */
#define SPARE_MIN 1024u /* spare slots every level gets at least (a pasted subtree, a burst of spawns) */
#define SPARE_LEVELS 2u /* empty levels below the deepest one: a subtree dragged under a leaf is patched, not rebuilt */
#define ROWS_PER_THREAD_MIN 2048u /* below this many changed rows per helper a thread made for the call costs more than it saves */
#define ROWS_PER_WORKER_MIN 128u  /* the same for the workers that outlive the frame (GpuTransformCullSystem_pool.inc) */

typedef struct Batch {
    void* world_col;
    void* local_col;
    void* up_row; /* the shared parent row the Up term resolved to (NULL: not fetched yet) */
    ecs_table_t* table;
    int32_t col_w, col_l; /* column indices of the two components in the table */
    ecs_entity_t src; /* entity the Up term was matched on */
    uint32_t count;
    uint32_t walk_begin; /* first row in walk order */
    uint32_t slot_begin; /* first row in slot order */
    uint32_t parent_slot; /* slot of src (YH_NO_PARENT at level 0): every row of the batch has this parent */
    int32_t level;
} Batch;

typedef struct MapEntry {
    ecs_entity_t key; /* 0 = empty */
    uint32_t value;
} MapEntry;

typedef struct Map {
    MapEntry* e;
    uint64_t cap;   /* power of two */
    uint64_t pmask; /* entries per partition - 1 */
    uint32_t pbits; /* 2^pbits partitions (0: one): the top bits of the hash pick the partition, probing wraps inside it */
    int overflow;   /* a partition was full and a key was dropped (cap >= 2 n and 4096+ entries per partition: never seen) */
    size_t mapped;  /* bytes, when the table came from mmap (large tables: huge pages where the kernel gives them) */
} Map;

typedef struct SlotRun {
    uint32_t first, count;
} SlotRun;

typedef struct BoundsEntry {
    ecs_entity_t key;
    float mm[6];
} BoundsEntry;

#include "GpuTransformCullSystem_pool.inc"

struct GpuTransformCullSystem {
    ecs_world_t* world;
    ecs_entity_t local_id, world_id, system;
    yh_ctx* ctx;
    yh_scene* scene;
    uint64_t scene_entities;
    uint32_t scene_levels;

    Batch* batches;
    uint32_t n_batches, cap_batches;
    ecs_entity_t* walk_entities;   /* this walk, walk order */
    ecs_entity_t* cached_entities; /* last walk */
    ecs_entity_t* cached_src;      /* last walk, per batch */
    uint32_t* cached_counts;
    int32_t* cached_levels;
    ecs_table_t** cached_tables; /* last walk, per batch: for the walk-free frame */
    int32_t* cached_cols;        /* 2 per batch: world, local column index */
    int32_t cached_match_count;
    int walk_is_whole_tables;    /* every batch of the last walk was a whole table (offset 0, count = table count) */
    uint64_t walks, walks_skipped;
    uint32_t cached_batches;
    uint64_t n, cap_n, cached_n, cap_cached; /* cap_n / cap_cached: room in walk_entities / cached_entities */

    ecs_entity_t* entity_of_slot;
    uint32_t* order; /* batch indices in slot order */
    uint8_t* stage_w;
    uint8_t* stage_l;
    uint32_t n_levels;

    BoundsEntry* bounds;
    uint64_t bounds_cap, bounds_n;
    int bounds_stale;
    ecs_entity_t* mesh_dirty; /* entities whose Mesh / Material was set since the last frame (mesh-table scenes) */
    uint64_t n_mesh_dirty, cap_mesh_dirty;

    float planes[YH_MAX_FRUSTA * 24];
    float frame_planes[YH_MAX_FRUSTA * 24]; /* the inputs of the last frame that ran on the device (idle frames: run()) */
    uint32_t frame_n_frusta;
    uint8_t frame_root[64];
    int frame_has_root, frame_valid, idle_skip;
    uint64_t frames_idle, frames_cull_only;
    uint32_t n_frusta;
    uint32_t n_explicit;                      /* frusta handed over with setFrusta: they override the cameras */
    /* inputs read from the ECS (optional): Camera.view_projection and the AABB of the mesh an entity's Mesh refers to */
    ecs_entity_t camera_id, mesh_id;
    uint32_t camera_vp_offset;
    ecs_query_t* camera_query;
    ecs_entity_t frustum_camera[YH_MAX_FRUSTA];
    int camera_pipeline;                      /* frusta from the cameras' world matrices of THIS frame (setCameraPipeline) */
    GpuCameraLayout camera_layout;
    float camera_aspect;
    uint32_t mesh_bounds_offset;
    int32_t mesh_vertices_offset;
    uint32_t scene_bounds_kind;               /* bounds kind the device scene was created with */
    uint32_t n_meshes;                        /* entries of the mesh table of the last upload (entry 0: the empty box) */
    float* mesh_table;                        /* host copy of the table: MESH_TABLE_MAX x {mins, maxs} */
    uint32_t* mesh_vcount;
    Map mesh_index;                           /* (material, mesh) pointer pair -> table entry (pair_key) */
    ecs_entity_t material_id;                 /* Material{ ref: *shading.Material } (components/material.zig), 0: meshes only */
    const void** entry_material;              /* per table entry: the pair it stands for (NULL, NULL for entry 0 and overrides) */
    const void** entry_mesh;
    uint32_t mesh_count;                      /* entries of the host copy (patch() appends; n_meshes follows at the next upload) */

    /* incremental mode: only rows whose onSet fired go up, only rows the frame touched come back */
    int incremental;
    int skip_walks;          /* reuse the last walk of the query while nothing structural changed (default on) */
    /* Structural-change tracking: observers on every id added to / removed from a table that holds one of the two
     * components, and on such tables being created / deleted, raise structure_dirty.  While it is clear the batches of
     * the last walk ARE the world's tables, row for row (nothing can enter, leave or move inside a table without one of
     * those events), so a frame neither iterates the query nor validates any table: its cost is O(rows that changed). */
    int structure_dirty;
    int batches_live;        /* batches[] describe the tables as of the last structural event */
    int trust_observers;     /* default on; off: every frame re-validates every cached table (round 1's behaviour) */
    uint64_t frames_clean;   /* frames that touched no table at all */
    ecs_entity_t root;       /* holder of the root WorldTransform (the Up source of level 0), from the last walk */
    uint32_t* batch_of;      /* per slot: index into batches[] (O(1) instead of a binary search per changed run) */
    int pinned;              /* stage_w / stage_l / slot_buf are page-locked (yh_host_alloc) */
    int readback;            /* 1 (default): rows a frame changed come back into the flecs columns; 0: they stay on the device */
    int host_stale;          /* frames ran without readback: the next frame with readback brings whole columns back */
    int threads;             /* helpers for copying changed rows back into the tables (1 = the calling thread only) */
    Pool pool;               /* threads - 1 workers for the per-frame jobs, started at the first such job */
    int frame_rebuilt;       /* this frame sent whole columns (stage_l / stage_w hold them by slot) */
    int host_clears_local;   /* the scene tracks YH_TRACK_REWRITTEN_LOCALS: every LocalTransform row sent has its dirty byte cleared here */
    Map slot_of;             /* entity -> slot, valid between rebuilds */
    ecs_entity_t* dirty[2];  /* [0] LocalTransform, [1] WorldTransform: entities set since the last frame */
    uint64_t n_dirty[2], cap_dirty[2];
    uint32_t* stamp[2];      /* per slot: frame in which the row was last queued (duplicates in the dirty lists) */
    uint32_t epoch;
    uint32_t* slot_buf;      /* n entries: slots of a sparse upload / of the changed rows */
    int slot_buf_pinned;
    uint64_t slot_buf_cap;   /* entries it has room for (>= n_slots) */
    ecs_entity_t observers[8];
    uint64_t last_up_rows, last_down_rows;
    double last_ms[6];       /* the last frame's wall time: tables + inputs, rows up, device frame, rows down and placed; of the first: query walk, slot-order rebuild */

    /* Small structural changes without a rebuild (patch(), below).  Every level's range of slots ends in spare slots,
     * dead on the device (yh_scene_set_alive_range); the observers list the tables rows entered or left; the rows of
     * such a table move to spare slots of its level, its old slots die. */
    int patching;                 /* default on; needs incremental mode and trusted observers */
    int refresh;                  /* default on: a rebuild takes the tables from the observers' lists, not from a walk (refresh_batches) */
    uint64_t n_slots;             /* slots of the device scene: rows + spare */
    uint32_t level_begin[YH_MAX_LEVELS + 1];
    uint32_t level_fill[YH_MAX_LEVELS]; /* first free slot of every level's range */
    struct SlotRun* free_runs[YH_MAX_LEVELS]; /* per level: runs of slots that died (their table's rows moved away or left), reused before the tail */
    uint32_t n_free[YH_MAX_LEVELS];
    ecs_table_t** touched;        /* tables with one of the components that rows entered or left since the last frame */
    uint32_t n_touched, cap_touched;
    ecs_table_t** deleted;        /* tables with one of the components deleted since the last frame: their batches die */
    uint32_t n_deleted, cap_deleted;
    int need_full;                /* an event patch() cannot follow: a table deleted, too many tables touched */
    int slots_untrusted;          /* a patch failed half-way: entity_of_slot / slot_of / the device rows disagree until the next rebuild */
    int walk_cache_valid;         /* cached_* describe the batches (false after a patch: the next walk rebuilds) */
    uint32_t* cached_slot;        /* per batch: slot_begin and parent_slot as the last rebuild assigned them */
    uint64_t slot_entries;        /* entries put into slot_of (stale ones stay until the next rebuild) */
    uint64_t patches, last_patch_tables;
    Map table_batch;              /* table pointer -> its live batch (rebuilt with the slots, kept up to date by patch()) */
    uint64_t table_batch_entries;
    int patch_index_valid;        /* table_batch / slot_has_kids describe the batches (rebuild() clears it) */
    pthread_t index_thread;       /* builds them in the background after a rebuilt frame (index_begin / index_join) */
    int index_thread_live;
    uint8_t* slot_has_kids;       /* per slot: some batch names it as its parent slot (set, never cleared between rebuilds) */

    int status;
    uint64_t rebuilds;
};

/* fn(job p) for p = 0 .. parts-1, jobs of job_size bytes each in one array: part 0 on the calling thread, the others on
 * the workers (the calling thread takes what no worker took).  Only ever called from the thread that runs the system. */
static void run_parts(GpuTransformCullSystem* s, void* (*fn)(void*), void* jobs, size_t job_size, uint32_t parts) {
    void* args[MAX_THREADS];
    if (parts > MAX_THREADS) parts = MAX_THREADS;
    for (uint32_t p = 0; p < parts; ++p) args[p] = (char*)jobs + (size_t)p * job_size;
    if (parts < 2) {
        if (parts) fn(args[0]);
        return;
    }
    pool_start(&s->pool, s->threads);
    pool_begin(&s->pool, fn, args + 1, parts - 1);
    fn(args[0]);
    pool_finish(&s->pool);
}

/* The parts, in dependency order; one translation unit (everything but the public functions at the end is static). */
#include "GpuTransformCullSystem_map.inc"
#include "GpuTransformCullSystem_walk.inc"
#include "GpuTransformCullSystem_rebuild.inc"
#include "GpuTransformCullSystem_inputs.inc"
#include "GpuTransformCullSystem_observers.inc"
#include "GpuTransformCullSystem_rows.inc"
#include "GpuTransformCullSystem_patch.inc"

/* The look-up tables of patch() cost 7 ms at 300k tables.  The rebuild does not build them (the first frame is long
 * enough), the first patch used to; now a thread builds them once the rebuilt frame is over, while the engine goes on:
 * by the time something structural happens they are there (a first re-parented leaf: 7.2 -> 0.35 ms).  The thread reads
 * the batches' tables, counts and parent slots, which nothing writes while frames are clean; whatever is about to change
 * them joins it first. */
static void* index_thread_main(void* arg) {
    patch_index_build((GpuTransformCullSystem*)arg);
    return NULL;
}
static void index_join(GpuTransformCullSystem* s) {
    if (!s->index_thread_live) return;
    pthread_join(s->index_thread, NULL);
    s->index_thread_live = 0;
}
static void index_begin(GpuTransformCullSystem* s) {
    if (s->index_thread_live || s->patch_index_valid || !s->patching || !s->incremental || getenv("YUME_GPU_SYSTEM_LAZY_INDEX")) return;
    s->index_thread_live = pthread_create(&s->index_thread, NULL, index_thread_main, s) == 0;
}

/* YUME_GPU_SYSTEM_TRACE=1: the steps of a frame with their wall time on stderr (read once) */
static int trace_on(void) {
    static int on = -1;
    if (on < 0) on = getenv("YUME_GPU_SYSTEM_TRACE") != NULL;
    return on;
}

static double now_ms(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

/* ---- the camera matrix pipeline ------------------------------------------------------------- */
/* WorldTransform.matrix of `e` as THIS frame will leave it, before the frame has run: the three-branch rule
 * (TransformSyncSystem.zig:60-73) replayed from the topmost holder of a WorldTransform down e's chain of `Up` sources,
 * on the rows the frame is about to read - flecs' own (onSet hooks have set their dirty bytes), or the device's when
 * frames ran without readback (the uploads of this frame are done by now).  The same products in the same order as the
 * kernels (yh_mat4_mul = m4_mul, parent * local), so the result IS the row the frame writes.  Returns 1 and the matrix,
 * 0 if e holds no WorldTransform, a negative status on a failed row fetch. */
static int camera_world_this_frame(GpuTransformCullSystem* s, ecs_entity_t e, float out[16]) {
    ecs_entity_t chain[YH_MAX_LEVELS + SPARE_LEVELS];
    uint32_t chain_slot[YH_MAX_LEVELS + SPARE_LEVELS];
    uint32_t depth = 0, slot;
    ecs_entity_t top = e;
    while (slot_lookup(s, top, &slot)) { /* rows of this system: up through the Up sources of their tables */
        if (depth == YH_MAX_LEVELS + SPARE_LEVELS) return YH_ERR_STATE;
        chain[depth] = top;
        chain_slot[depth++] = slot;
        top = batch_of_slot(s, slot)->src;
    }
    /* `top` holds a WorldTransform and no slot: the scene root (GameApp.zig:77,95), or e itself without a LocalTransform */
    const uint8_t* row = top ? (const uint8_t*)ecs_get_id(s->world, top, s->world_id) : NULL;
    if (!row) return 0;
    float pm[16];
    memcpy(pm, row, 64);
    int pdirty = row[64] != 0;
    for (uint32_t i = depth; i-- > 0;) {
        uint8_t wrow[ROW], lrow[ROW];
        if (s->host_stale) {
            int rc;
            if ((rc = yh_download_world_aos68(s->scene, wrow, chain_slot[i], 1)) != YH_OK ||
                (rc = yh_download_local_aos68(s->scene, lrow, chain_slot[i], 1)) != YH_OK)
                return rc;
        } else {
            const void* w = ecs_get_id(s->world, chain[i], s->world_id);
            const void* l = ecs_get_id(s->world, chain[i], s->local_id);
            if (!w || !l) return 0;
            memcpy(wrow, w, ROW);
            memcpy(lrow, l, ROW);
        }
        int ld = lrow[64] != 0;
        const int wd = wrow[64] != 0;
        /* a LocalTransform row that went up this frame had its flag cleared in the table at that moment (gather_range,
         * columns_thread): the flag the frame will read is the one that was sent */
        if (!s->host_stale && s->host_clears_local)
            ld = ld || (s->frame_rebuilt ? s->stage_l[(size_t)chain_slot[i] * ROW + 64] != 0 : s->stamp[0][chain_slot[i]] == s->epoch);
        if (!pdirty && !ld && !wd) { /* :62-64: the row stays as it is */
            memcpy(pm, wrow, 64);
        } else if (ld || pdirty) { /* :67-68 */
            float l[16], w[16];
            memcpy(l, lrow, 64);
            yh_mat4_mul(pm, l, w);
            memcpy(pm, w, 64);
            pdirty = 1;
        } else { /* :69-72: the world matrix was edited; the local one is re-derived from it, the world row stays */
            memcpy(pm, wrow, 64);
            pdirty = 1;
        }
    }
    memcpy(out, pm, 64);
    return 1;
}

static void run(ecs_iter_t* it) {
    GpuTransformCullSystem* s = (GpuTransformCullSystem*)it->ctx;
    int rc = YH_OK;
    uint64_t patched_rows_up = 0;
    const double t_begin = now_ms();
    memset(s->last_ms, 0, sizeof s->last_ms);
    int rebuilt = 0;
    /* Nothing structural happened since the batches were last brought up to date (no observer fired, the query's cache
     * matched or unmatched no table): they still are the world's tables, row for row.  No iteration, no validation. */
    int clean = s->skip_walks && s->trust_observers && !s->structure_dirty && s->scene && s->cached_n && s->batches_live &&
                it->query && ecs_query_match_count(it->query) == s->cached_match_count;
    s->last_up_rows = 0;
    s->last_patch_tables = 0;
    if (!clean) index_join(s); /* the batches are about to change */
    if (!clean && s->patching && s->incremental && s->skip_walks && s->trust_observers &&
        !s->need_full && s->scene && s->cached_n && s->batches_live && it->query) {
        /* rows entered or left a few tables: move just those (no walk of the query, no rebuild) */
        const double t_patch = now_ms();
        rc = patch(s);
        s->last_ms[5] = now_ms() - t_patch;
        if (rc == YH_OK) {
            s->structure_dirty = 0;
            s->n_touched = s->n_deleted = 0;
            s->cached_match_count = ecs_query_match_count(it->query);
            clean = 1;
            patched_rows_up = s->last_up_rows;
        } else if (rc < 0) { /* half-done: nothing of the cached state is trusted any more */
            s->slots_untrusted = 1; /* (nor the slot -> entity table: see the rebuild below) */
            s->need_full = 1;
            s->structure_dirty = 1;
            s->walk_cache_valid = 0;
            s->batches_live = 0;
            ecs_iter_fini(it);
            s->status = rc;
            return;
        }
        rc = YH_OK;
    }
    uint8_t* root_row = NULL;
    if (clean) {
        ecs_iter_fini(it); /* the iterator is not consumed */
        s->walks_skipped += 1;
        s->frames_clean += 1;
    } else {
        /* the batches were the world's tables a frame ago and the observers' lists of what changed since are complete */
        const int lists_complete = s->batches_live && !s->need_full && s->skip_walks && s->trust_observers && s->incremental &&
                                   s->scene && it->query && s->refresh && (s->n_touched || s->n_deleted);
        s->batches_live = 0;
        s->structure_dirty = 0; /* (cleared first: what the walk below sees is what the flag stands for) */
        s->need_full = 0;
        if (s->skip_walks && restore_walk(s, it)) {
            ecs_iter_fini(it);
            s->n_touched = s->n_deleted = 0;
        } else if (lists_complete && (rc = refresh_batches(s)) == YH_OK) {
            ecs_iter_fini(it); /* the batches were brought up to date from the observers' lists: no walk */
            s->walks_skipped += 1;
            s->n_touched = s->n_deleted = 0;
            rebuilt = 1;
            s->cached_match_count = ecs_query_match_count(it->query);
        } else if (rc < 0) {
            ecs_iter_fini(it);
            s->status = rc;
            return;
        } else {
            s->n_touched = s->n_deleted = 0;
            rc = walk(s, it);
            s->last_ms[4] = now_ms() - t_begin;
            if (rc != YH_OK || s->n == 0) {
                s->status = rc;
                return;
            }
            rebuilt = !same_as_last_walk(s);
            s->cached_match_count = it->query ? ecs_query_match_count(it->query) : 0;
        }
        if (!rebuilt) restore_slots(s);
        else {
            if (s->host_stale && s->scene && s->slots_untrusted) {
                /* A patch failed half-way (a device error) while flecs' columns were behind the device: slot -> entity is
                 * partly rewritten, so bringing rows home through it could hand an entity another one's matrices.  The rows
                 * computed on the device since readback was switched off are given up; flecs' own copies go up again. */
                s->host_stale = 0;
            } else if (s->host_stale && s->scene && (rc = catch_up_by_entity(s)) != YH_OK) {
                s->status = rc;
                return;
            }
            s->slots_untrusted = 0;
            const double t_rebuild = now_ms();
            rc = rebuild(s);
            s->last_ms[5] = now_ms() - t_rebuild;
            if (rc != YH_OK) {
                s->status = rc;
                return;
            }
        }
        /* the scene root: every level-0 table must name the same holder of the parent matrix */
        ecs_entity_t root = 0;
        for (uint32_t i = 0; i < s->n_batches; ++i) {
            const Batch* b = &s->batches[i];
            if (b->level != 0) continue;
            if (!root) {
                root = b->src;
            } else if (root != b->src) {
                s->status = YH_ERR_UNSUPPORTED; /* more than one root transform: not a reference scene (GameApp.zig:77) */
                return;
            }
        }
        s->root = root;
        s->batches_live = 1;
    }
    const int inputs_changed = s->bounds_stale || s->n_mesh_dirty; /* boxes or keys go up below */
    if (s->bounds_stale) s->n_mesh_dirty = 0; /* (every id is read afresh) */
    if ((s->bounds_stale && (rc = upload_bounds(s)) != YH_OK) || (s->n_mesh_dirty && (rc = mesh_ids_refresh(s)) != YH_OK)) {
        s->status = rc;
        return;
    }
    /* the root's row is looked up every frame: its own table may have moved it */
    if (s->root) root_row = (uint8_t*)ecs_get_id(s->world, s->root, s->world_id);
    /* matrix and dirty byte exactly as level-0 rows read them through the Up term */
    if (root_row && (rc = yh_upload_root_aos68(s->scene, root_row)) != YH_OK) {
        s->status = rc;
        return;
    }
    const double t_inputs = now_ms();
    s->frame_rebuilt = rebuilt;
    if (!s->incremental || rebuilt) {
        /* whole columns up; in plain mode whole columns down again */
        copy_columns(s, 1);
        if ((rc = yh_upload_world_aos68(s->scene, s->stage_w, 0, s->n_slots)) != YH_OK ||
            (rc = yh_upload_local_aos68(s->scene, s->stage_l, 0, s->n_slots)) != YH_OK) {
            s->status = rc;
            return;
        }
        s->last_up_rows = 2 * s->n_slots;
    } else {
        /* the rows whose onSet hook fired since the last frame (transform.zig:46-51, 92-97: the only writers of dirty) */
        s->epoch += 1;
        s->last_up_rows = patched_rows_up;
        for (int k = 0; k < 2; ++k) {
            uint8_t* stage = k == 0 ? s->stage_l : s->stage_w;
            const double tg0 = now_ms();
            const uint64_t m = gather_dirty(s, k, stage);
            const double tg1 = now_ms();
            rc = k == 0 ? yh_upload_local_rows_aos68(s->scene, s->slot_buf, stage, m)
                        : yh_upload_world_rows_aos68(s->scene, s->slot_buf, stage, m);
            if (m && trace_on())
                fprintf(stderr, "  rows up: %llu %s rows gathered %.3f ms, sent %.3f ms\n", (unsigned long long)m, k ? "WorldTransform" : "LocalTransform",
                        tg1 - tg0, now_ms() - tg1);
            if (rc != YH_OK) {
                s->status = rc;
                return;
            }
            s->last_up_rows += m;
        }
    }
    /* frusta: the ones handed over with setFrusta, else one per Camera entity - from this frame's world matrix of the
     * camera (setCameraPipeline: GameWindow.zig:160-161 done before the frame instead of after it), or from
     * Camera.view_projection as the engine last updated it (components/camera.zig:300-302).  After the uploads: the
     * device rows the replay may have to read (readback off) are this frame's inputs by now. */
    if (!s->n_explicit && s->camera_query) {
        uint32_t nf = 0;
        ecs_iter_t cit = ecs_query_iter(s->world, s->camera_query);
        while (ecs_query_next(&cit)) {
            const int32_t col = cit.table ? ecs_table_get_column_index(s->world, cit.table, s->camera_id) : -1;
            if (col < 0) continue;
            const size_t size = ecs_table_get_column_size(cit.table, col);
            uint8_t* cams = (uint8_t*)ecs_table_get_column(cit.table, col, cit.offset);
            for (int32_t r = 0; r < cit.count && nf < YH_MAX_FRUSTA; ++r) {
                uint8_t* cam = cams + (size_t)r * size;
                float vp[16], wm[16];
                if (s->camera_pipeline && (rc = camera_world_this_frame(s, cit.entities[r], wm)) == 1) {
                    const GpuCameraLayout* L = &s->camera_layout;
                    float opt[3], view[16], proj[16];
                    memcpy(opt, cam + L->options_offset, sizeof opt); /* PerspectiveOptions{fovy_rad, near, far} */
                    yh_camera_update_from_world(cam[L->kind_offset] == 1, wm, s->camera_aspect, opt[0], opt[1], opt[2], view, proj, vp);
                    memcpy(cam + L->view_offset, view, 64); /* camera.zig:266, 271-272, 301 */
                    memcpy(cam + L->projection_offset, proj, 64);
                    memcpy(cam + L->view_projection_offset, vp, 64);
                } else if (s->camera_pipeline && rc < 0) {
                    while (ecs_query_next(&cit)) {} /* the iterator owns resources: run it to its end */
                    s->status = rc;
                    return;
                } else {
                    memcpy(vp, cam + s->camera_vp_offset, 64);
                }
                yh_extract_planes(vp, s->planes + 24 * nf);
                s->frustum_camera[nf++] = cit.entities[r];
            }
        }
        s->n_frusta = nf;
    }
    const uint32_t cull_flags = s->n_frusta ? 0u : YH_FRAME_NO_CULL;
    const double t_up = now_ms();
    /* A frame that has nothing to do: no structural change, no row set since the last frame, the root's row clean and the
     * same, the same bounds and the same frusta as the frame before.  The device would recompute nothing and list the same
     * entities: no launch, no copy - the lists and rows of the last frame stand.  (The reference's CPU systems visit every
     * row every frame to find the same out: TransformSyncSystem.zig:60.) */
    /* nothing moved: no structural change, no row set since the last frame, the root's row clean and the same */
    const int nothing_moved = s->idle_skip && s->incremental && clean && !rebuilt && s->last_up_rows == 0 && s->last_patch_tables == 0 &&
                              s->frame_valid && !(s->readback && s->host_stale) &&
                              (root_row ? (s->frame_has_root && root_row[64] == 0 && memcmp(root_row, s->frame_root, 64) == 0) : !s->frame_has_root);
    if (nothing_moved && !inputs_changed && s->n_frusta == s->frame_n_frusta &&
        memcmp(s->planes, s->frame_planes, sizeof(float) * 24 * s->n_frusta) == 0) {
        s->n_dirty[0] = s->n_dirty[1] = 0;
        s->last_down_rows = 0;
        s->frames_idle += 1;
        s->last_ms[0] = t_inputs - t_begin;
        s->last_ms[1] = t_up - t_inputs;
        s->status = YH_OK;
        return;
    }
    s->frame_valid = 0;
    /* Only the frusta (a camera moved: the common frame) or the bounds changed: the device culls the world matrices it
     * holds, in one launch - no rule, no sweep, no changed rows. */
    const uint32_t only_cull = nothing_moved && s->n_frusta ? YH_FRAME_CULL_ONLY : 0u;
    s->frames_cull_only += only_cull != 0;
    rc = yh_frame(s->scene, s->planes, s->n_frusta, cull_flags | only_cull); /* returns when the device is done */
    const double t_frame = now_ms();
    if (rc != YH_OK) {
        s->n_dirty[0] = s->n_dirty[1] = 0;
        s->status = rc;
        return;
    }
    if (s->incremental && !s->readback) {
        /* A device consumer reads the results where they are (yh_object_data_device, draw lists): nothing comes back.
         * The rows whose onSet fired had their dirty byte set by the hook; the frame cleared it on the device, so it is
         * cleared here too — their matrices, and every row the frame recomputed, are now newer on the device. */
        for (int k = 0; k < 2; ++k)
            for (uint64_t i = 0; i < s->n_dirty[k]; ++i) {
                uint32_t slot;
                if (!slot_lookup(s, s->dirty[k][i], &slot)) continue;
                const Batch* b = batch_of_slot(s, slot);
                ((uint8_t*)(k == 0 ? b->local_col : b->world_col))[(size_t)(slot - b->slot_begin) * ROW + 64] = 0;
            }
        s->host_stale = 1;
        s->last_down_rows = 0;
    } else if (!s->incremental || s->host_stale) {
        if ((rc = yh_download_world_aos68(s->scene, s->stage_w, 0, s->n_slots)) != YH_OK ||
            (rc = yh_download_local_aos68(s->scene, s->stage_l, 0, s->n_slots)) != YH_OK) {
            s->status = rc;
            return;
        }
        copy_columns(s, 0);
        s->last_down_rows = 2 * s->n_slots;
        s->host_stale = 0;
    } else {
        /* Every row the frame recomputed or found dirty, flags cleared; all other rows are as flecs holds them.  The two
         * components come back separately: a parent that moved rewrote every WorldTransform below it and no
         * LocalTransform, so only the world rows cross PCIe. */
        uint64_t mw = 0, ml = 0;
        const int trace = trace_on();
        const double t0 = trace ? now_ms() : 0;
        rc = yh_get_changed_counts(s->scene, &mw, &ml);
        const double t1 = trace ? now_ms() : 0;
        if (rc == YH_OK && mw) rc = fetch_changed(s, 1, mw);
        const double t2 = trace ? now_ms() : 0;
        if (rc == YH_OK && ml) rc = fetch_changed(s, 0, ml);
        if (trace)
            fprintf(stderr, "  rows down: counts %.3f ms, %llu WorldTransform rows %.3f ms, %llu LocalTransform rows %.3f ms\n", t1 - t0,
                    (unsigned long long)mw, t2 - t1, (unsigned long long)ml, now_ms() - t2);
        if (rc != YH_OK) {
            s->status = rc;
            return;
        }
        s->last_down_rows = mw + ml;
    }
    s->n_dirty[0] = s->n_dirty[1] = 0;
    if (rebuilt && s->incremental && s->n_slots) {
        /* Some time after a scene's set-up the HIP runtime spends 5-6 ms inside one hipMemcpyAsync, once (seen in the first
         * frame with a few edited rows when cameras and meshes come from the ECS: 27 KB of slots took 5.6 ms to enqueue;
         * warming every kind of copy up in yh_init does not prevent it, timing the calls makes it go away).  A one-row
         * download here, at the end of a frame that cost a rebuild anyway, takes it off that frame in the default modes;
         * with setPatching(0) it still shows in the first frame that brings every row back. */
        uint8_t row[ROW];
        (void)yh_download_world_aos68(s->scene, s->pinned ? s->stage_w : row, 0, 1);
    }
    if (root_row) root_row[64] = 0; /* PostWorldTransform reaches the root too (single-term query) */
    s->frame_n_frusta = s->n_frusta; /* what this frame was computed from: the next one may find it has nothing to do */
    memcpy(s->frame_planes, s->planes, sizeof s->frame_planes);
    s->frame_has_root = root_row != NULL;
    if (root_row) memcpy(s->frame_root, root_row, 64);
    s->frame_valid = 1;
    s->last_ms[0] = t_inputs - t_begin;
    s->last_ms[1] = t_up - t_inputs;
    s->last_ms[2] = t_frame - t_up;
    s->last_ms[3] = now_ms() - t_frame;
    s->status = YH_OK;
    if (rebuilt) index_begin(s);
}

/* ---- public ------------------------------------------------------------------------------------------------ */
ecs_entity_t GpuTransformCullSystem_registerTo(ecs_world_t* world, ecs_entity_t local_transform,
                                               ecs_entity_t world_transform, yh_ctx* ctx,
                                               GpuTransformCullSystem** out_state) {
    if (!world || !ctx || !out_state || !local_transform || !world_transform) return 0;
    GpuTransformCullSystem* s = (GpuTransformCullSystem*)calloc(1, sizeof *s);
    if (!s) return 0;
    s->world = world;
    s->local_id = local_transform;
    s->world_id = world_transform;
    s->ctx = ctx;

    ecs_entity_desc_t ed;
    memset(&ed, 0, sizeof ed);
    ed.name = "GpuTransformCullSystem";
    ecs_id_t add[3] = {ecs_pair(EcsDependsOn, EcsPostUpdate), EcsPostUpdate, 0}; /* ecs.zig:1224-1235 */
    ed.add = add;
    ecs_system_desc_t sd;
    memset(&sd, 0, sizeof sd);
    sd.entity = ecs_entity_init(world, &ed);
    sd.run = run;
    sd.ctx = s;
    sd.query.terms[0].id = world_transform;
    sd.query.terms[0].inout = EcsInOut;
    sd.query.terms[1].id = local_transform;
    sd.query.terms[1].inout = EcsInOut;
    sd.query.terms[2].id = world_transform;
    sd.query.terms[2].inout = EcsIn;
    sd.query.terms[2].src.id = EcsUp;
    sd.query.cache_kind = EcsQueryCacheAuto; /* what systems get by default; the walk-free frame needs the cache */
    s->system = ecs_system_init(world, &sd);
    if (!s->system) {
        free(s);
        return 0;
    }
    s->incremental = 1;
    s->skip_walks = 1;
    s->trust_observers = 1;
    s->structure_dirty = 1;
    s->readback = 1;
    s->patching = 1;
    s->idle_skip = getenv("YUME_GPU_SYSTEM_NO_IDLE_SKIP") == NULL; /* (the knob: A/B) */
    s->refresh = getenv("YUME_GPU_SYSTEM_NO_REFRESH") == NULL; /* (the knob: A/B and the tests that compare both ways) */
    {
        long cpus = sysconf(_SC_NPROCESSORS_ONLN);
        s->threads = cpus >= 16 ? 16 : (cpus > 1 ? (int)cpus : 1); /* placing rows is bound by memory latency: helpers overlap the misses */
    }
    for (int k = 0; k < 2; ++k) {
        ecs_observer_desc_t od;
        memset(&od, 0, sizeof od);
        od.query.terms[0].id = k == 0 ? local_transform : world_transform;
        od.events[0] = EcsOnSet;
        od.callback = k == 0 ? on_set_local : on_set_world;
        od.ctx = s;
        s->observers[k] = ecs_observer_init(world, &od);
    }
    for (int k = 0; k < 2; ++k) { /* any component or tag, any pair: added or removed */
        ecs_observer_desc_t od;
        memset(&od, 0, sizeof od);
        od.query.terms[0].id = k == 0 ? EcsWildcard : ecs_pair(EcsWildcard, EcsWildcard);
        /* world.enable(e, false) (ecs.zig:546) adds the Disabled tag: the row moves to a table no query matches by
         * default - not an observer's either, which would then never hear of it leaving (or of it coming back) */
        od.query.flags = EcsQueryMatchDisabled | EcsQueryMatchPrefab;
        od.events[0] = EcsOnAdd;
        od.events[1] = EcsOnRemove;
        od.callback = note_structure;
        od.ctx = s;
        s->observers[2 + k] = ecs_observer_init(world, &od);
    }
    for (int k = 0; k < 2; ++k) { /* tables with one of the components created or deleted */
        ecs_observer_desc_t od;
        memset(&od, 0, sizeof od);
        od.query.terms[0].id = k == 0 ? local_transform : world_transform;
        od.query.flags = EcsQueryMatchDisabled | EcsQueryMatchPrefab;
        od.events[0] = EcsOnTableCreate;
        od.events[1] = EcsOnTableDelete;
        od.callback = note_table;
        od.ctx = s;
        s->observers[4 + k] = ecs_observer_init(world, &od);
    }
    *out_state = s;
    return s->system;
}

int GpuTransformCullSystem_setBounds(GpuTransformCullSystem* s, ecs_entity_t entity, const float min_max[6]) {
    if (!s || !entity || !min_max) return YH_ERR_BAD_ARG;
    BoundsEntry* b = bounds_find(s, entity);
    if (!b) {
        if ((s->bounds_n + 1) * 2 > s->bounds_cap) {
            const int rc = bounds_grow(s);
            if (rc != YH_OK) return rc;
        }
        uint64_t i = hash64(entity) & (s->bounds_cap - 1);
        while (s->bounds[i].key) i = (i + 1) & (s->bounds_cap - 1);
        b = &s->bounds[i];
        b->key = entity;
        s->bounds_n += 1;
    }
    memcpy(b->mm, min_max, sizeof b->mm);
    if (s->bounds_stale || !s->scene) {
        s->bounds_stale = 1;
        return YH_OK;
    }
    /* a scene is up: one entity's bounds do not make every entity's go up again */
    uint32_t slot;
    if (!slot_lookup(s, entity, &slot)) return YH_OK; /* no slot yet: whatever gives it one (patch, rebuild) reads the entry */
    if (s->scene_bounds_kind == YH_BOUNDS_MESH_TABLE) {
        s->bounds_stale = 1; /* (an entry of its own in the mesh table: the full upload hands those out) */
        return YH_OK;
    }
    const float* mn[3] = {&b->mm[0], &b->mm[1], &b->mm[2]};
    const float* mx[3] = {&b->mm[3], &b->mm[4], &b->mm[5]};
    return yh_upload_aabb(s->scene, mn, mx, slot, 1);
}

int GpuTransformCullSystem_setFrusta(GpuTransformCullSystem* s, const float* planes, uint32_t n_frusta) {
    if (!s || n_frusta > YH_MAX_FRUSTA || (n_frusta && !planes)) return YH_ERR_BAD_ARG;
    if (n_frusta) memcpy(s->planes, planes, sizeof(float) * 24u * n_frusta);
    s->n_frusta = n_frusta;
    s->n_explicit = n_frusta; /* 0: back to the Camera entities, if a camera component is known */
    memset(s->frustum_camera, 0, sizeof s->frustum_camera);
    return YH_OK;
}

uint64_t GpuTransformCullSystem_visible(GpuTransformCullSystem* s, uint32_t frustum, ecs_entity_t* out, uint64_t cap) {
    if (!s || !s->scene || frustum >= s->n_frusta) return 0;
    uint32_t counts[YH_MAX_FRUSTA], bnd[YH_MAX_FRUSTA];
    if (yh_get_counts(s->scene, counts, bnd) != YH_OK) return 0;
    const uint64_t total = counts[frustum];
    if (!out || !cap || !total) return total;
    uint32_t* slots = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)total);
    if (!slots) return 0;
    uint64_t written = 0;
    if (yh_get_draw_list(s->scene, frustum, slots, total, &written) != YH_OK) {
        free(slots);
        return 0;
    }
    const uint64_t m = written < cap ? written : cap;
    for (uint64_t i = 0; i < m; ++i) out[i] = s->entity_of_slot[slots[i]];
    free(slots);
    return written;
}

int GpuTransformCullSystem_status(const GpuTransformCullSystem* s) { return s ? s->status : YH_ERR_BAD_ARG; }
uint64_t GpuTransformCullSystem_entities(const GpuTransformCullSystem* s) { return s ? s->n : 0; }
uint32_t GpuTransformCullSystem_levels(const GpuTransformCullSystem* s) { return s ? s->n_levels : 0; }
uint64_t GpuTransformCullSystem_rebuilds(const GpuTransformCullSystem* s) { return s ? s->rebuilds : 0; }

int GpuTransformCullSystem_setIncremental(GpuTransformCullSystem* s, int on) {
    if (!s) return YH_ERR_BAD_ARG;
    s->incremental = on != 0;
    s->cached_n = 0; /* the next frame rebuilds: whole columns travel once, tracking is switched on the scene */
    s->n_dirty[0] = s->n_dirty[1] = 0;
    return YH_OK;
}

int GpuTransformCullSystem_setPatching(GpuTransformCullSystem* s, int on) {
    if (!s) return YH_ERR_BAD_ARG;
    s->patching = on != 0;
    s->cached_n = 0; /* the next frame rebuilds: the slot order gains or loses its spare slots */
    return YH_OK;
}
uint64_t GpuTransformCullSystem_patches(const GpuTransformCullSystem* s) { return s ? s->patches : 0; }

int GpuTransformCullSystem_setSkipWalks(GpuTransformCullSystem* s, int on) {
    if (!s) return YH_ERR_BAD_ARG;
    s->skip_walks = on != 0;
    return YH_OK;
}

int GpuTransformCullSystem_setTrustObservers(GpuTransformCullSystem* s, int on) {
    if (!s) return YH_ERR_BAD_ARG;
    s->trust_observers = on != 0;
    return YH_OK;
}

/* OnSet of Mesh / Material: the keys of those entities go stale.  A few of them are re-read one by one before the next
 * frame (mesh_ids_refresh); while every id is due to be read afresh anyway, or once there are too many, nothing is listed. */
#define MESH_DIRTY_MAX 65536u
static void note_mesh_set(ecs_iter_t* it) {
    GpuTransformCullSystem* s = (GpuTransformCullSystem*)it->ctx;
    if (s->bounds_stale) return;
    if (s->n_mesh_dirty + (uint64_t)it->count > MESH_DIRTY_MAX) {
        s->bounds_stale = 1;
        s->n_mesh_dirty = 0;
        return;
    }
    if (s->n_mesh_dirty + (uint64_t)it->count > s->cap_mesh_dirty) {
        uint64_t cap = s->cap_mesh_dirty ? s->cap_mesh_dirty * 2 : 256;
        while (cap < s->n_mesh_dirty + (uint64_t)it->count) cap *= 2;
        ecs_entity_t* d = (ecs_entity_t*)realloc(s->mesh_dirty, cap * sizeof(ecs_entity_t));
        if (!d) {
            s->bounds_stale = 1;
            s->n_mesh_dirty = 0;
            return;
        }
        s->mesh_dirty = d;
        s->cap_mesh_dirty = cap;
    }
    memcpy(s->mesh_dirty + s->n_mesh_dirty, it->entities, sizeof(ecs_entity_t) * (size_t)it->count);
    s->n_mesh_dirty += (uint64_t)it->count;
}

int GpuTransformCullSystem_setCameraComponent(GpuTransformCullSystem* s, ecs_entity_t camera, uint32_t view_projection_offset) {
    if (!s || !camera) return YH_ERR_BAD_ARG;
    const ecs_type_info_t* ti = ecs_get_type_info(s->world, camera);
    if (!ti || (uint64_t)view_projection_offset + 64 > (uint64_t)ti->size) return YH_ERR_BAD_ARG;
    if (s->camera_query) ecs_query_fini(s->camera_query);
    ecs_query_desc_t qd;
    memset(&qd, 0, sizeof qd);
    qd.terms[0].id = camera;
    qd.terms[0].inout = EcsInOut; /* the camera pipeline writes the three matrices, as updateMatrices does */
    qd.cache_kind = EcsQueryCacheAuto;
    s->camera_query = ecs_query_init(s->world, &qd);
    if (!s->camera_query) return YH_ERR_BAD_ARG;
    s->camera_id = camera;
    s->camera_vp_offset = view_projection_offset;
    return YH_OK;
}

int GpuTransformCullSystem_setCameraPipeline(GpuTransformCullSystem* s, const GpuCameraLayout* layout, float aspect) {
    if (!s) return YH_ERR_BAD_ARG;
    if (!layout) {
        s->camera_pipeline = 0;
        return YH_OK;
    }
    if (!s->camera_id) return YH_ERR_STATE; /* setCameraComponent first */
    const ecs_type_info_t* ti = ecs_get_type_info(s->world, s->camera_id);
    const uint64_t size = ti ? (uint64_t)ti->size : 0;
    if ((uint64_t)layout->kind_offset + 1 > size || (uint64_t)layout->options_offset + 12 > size ||
        (uint64_t)layout->view_offset + 64 > size || (uint64_t)layout->projection_offset + 64 > size ||
        (uint64_t)layout->view_projection_offset + 64 > size || !(aspect > 0.0f))
        return YH_ERR_BAD_ARG;
    s->camera_layout = *layout;
    s->camera_aspect = aspect;
    s->camera_pipeline = 1;
    return YH_OK;
}

int GpuTransformCullSystem_setCameraAspect(GpuTransformCullSystem* s, float aspect) {
    if (!s || !(aspect > 0.0f)) return YH_ERR_BAD_ARG;
    s->camera_aspect = aspect;
    return YH_OK;
}

int GpuTransformCullSystem_setMeshComponent(GpuTransformCullSystem* s, ecs_entity_t mesh, uint32_t bounds_offset,
                                            int32_t vertices_count_offset) {
    if (!s || !mesh) return YH_ERR_BAD_ARG;
    const ecs_type_info_t* ti = ecs_get_type_info(s->world, mesh);
    if (!ti || ti->size != (ecs_size_t)sizeof(void*)) return YH_ERR_BAD_ARG; /* Mesh{ ref: *meshes.Mesh } */
    s->mesh_id = mesh;
    s->mesh_bounds_offset = bounds_offset;
    s->mesh_vertices_offset = vertices_count_offset;
    if (!s->observers[6]) { /* a Mesh set on an entity that already has one moves no row: only the bounds go stale */
        ecs_observer_desc_t od;
        memset(&od, 0, sizeof od);
        od.query.terms[0].id = mesh;
        od.events[0] = EcsOnSet;
        od.callback = note_mesh_set;
        od.ctx = s;
        s->observers[6] = ecs_observer_init(s->world, &od);
    }
    s->cached_n = 0; /* the device scene changes its bounds kind: rebuild */
    s->structure_dirty = 1;
    return YH_OK;
}

int GpuTransformCullSystem_setMaterialComponent(GpuTransformCullSystem* s, ecs_entity_t material) {
    if (!s || !material) return YH_ERR_BAD_ARG;
    if (!s->mesh_id) return YH_ERR_STATE; /* setMeshComponent first: a key is a (material, mesh) pair */
    const ecs_type_info_t* ti = ecs_get_type_info(s->world, material);
    if (!ti || ti->size != (ecs_size_t)sizeof(void*)) return YH_ERR_BAD_ARG; /* Material{ ref: *shading.Material } */
    s->material_id = material;
    if (!s->observers[7]) { /* a Material set on an entity that already has one moves no row: only the keys go stale */
        ecs_observer_desc_t od;
        memset(&od, 0, sizeof od);
        od.query.terms[0].id = material;
        od.events[0] = EcsOnSet;
        od.callback = note_mesh_set;
        od.ctx = s;
        s->observers[7] = ecs_observer_init(s->world, &od);
    }
    s->bounds_stale = 1;
    return YH_OK;
}

static int by_pair(const void* a, const void* b) {
    const GpuDrawBucket* x = (const GpuDrawBucket*)a;
    const GpuDrawBucket* y = (const GpuDrawBucket*)b;
    if (x->material != y->material) return (uintptr_t)x->material < (uintptr_t)y->material ? -1 : 1;
    if (x->mesh != y->mesh) return (uintptr_t)x->mesh < (uintptr_t)y->mesh ? -1 : 1;
    return 0;
}

int GpuTransformCullSystem_drawBuckets(GpuTransformCullSystem* s, uint32_t frustum, uint32_t batch_offset, GpuDrawBucket* buckets,
                                       uint32_t cap_buckets, uint32_t* n_buckets, ecs_entity_t* entities, uint64_t cap,
                                       uint64_t* written) {
    if (!s || !n_buckets || !written) return YH_ERR_BAD_ARG;
    *n_buckets = 0;
    *written = 0;
    if (!s->scene || s->scene_bounds_kind != YH_BOUNDS_MESH_TABLE || frustum >= s->n_frusta) return YH_ERR_STATE;
    int rc = yh_build_mesh_buckets(s->scene, frustum, batch_offset);
    if (rc != YH_OK) return rc;
    const uint32_t nm = s->n_meshes;
    yh_bucket* raw = (yh_bucket*)malloc(sizeof(yh_bucket) * (nm ? nm : 1));
    uint32_t* slots = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(s->n_slots ? s->n_slots : 1));
    if (!raw || !slots) rc = YH_ERR_OOM;
    uint64_t total = 0;
    if (rc == YH_OK) rc = yh_get_mesh_buckets(s->scene, frustum, raw, nm, slots, s->n_slots, &total);
    if (rc == YH_OK) {
        uint32_t k = 0;
        for (uint32_t m = 0; m < nm; ++m) { /* entries with something to draw: a mesh, and visible entities */
            if (!raw[m].count || !s->entry_mesh[m]) continue;
            if (buckets && k < cap_buckets) {
                buckets[k].material = s->entry_material[m];
                buckets[k].mesh = s->entry_mesh[m];
                buckets[k].first = raw[m].first;
                buckets[k].count = raw[m].count;
            }
            ++k;
        }
        *n_buckets = k;
        *written = total;
        if (buckets && k > cap_buckets) rc = YH_ERR_CAPACITY;
        else if (buckets) qsort(buckets, k, sizeof(GpuDrawBucket), by_pair); /* material first: one pipeline bind per material */
        if (entities) {
            if (total > cap) rc = rc == YH_OK ? YH_ERR_CAPACITY : rc;
            else
                for (uint64_t i = 0; i < total; ++i) entities[i] = s->entity_of_slot[slots[i]];
        }
    }
    free(raw);
    free(slots);
    return rc;
}

int GpuTransformCullSystem_invalidateBounds(GpuTransformCullSystem* s) {
    if (!s) return YH_ERR_BAD_ARG;
    s->bounds_stale = 1;
    return YH_OK;
}

ecs_entity_t GpuTransformCullSystem_frustumCamera(const GpuTransformCullSystem* s, uint32_t frustum) {
    return (s && frustum < s->n_frusta) ? s->frustum_camera[frustum] : 0;
}
uint32_t GpuTransformCullSystem_frusta(const GpuTransformCullSystem* s) { return s ? s->n_frusta : 0; }
yh_scene* GpuTransformCullSystem_scene(GpuTransformCullSystem* s) { return s ? s->scene : NULL; }

int GpuTransformCullSystem_setReadback(GpuTransformCullSystem* s, int on) {
    if (!s) return YH_ERR_BAD_ARG;
    s->readback = on != 0;
    return YH_OK;
}

int GpuTransformCullSystem_fetchWorld(GpuTransformCullSystem* s, ecs_entity_t entity, float matrix[16]) {
    if (!s || !matrix || !s->scene) return YH_ERR_BAD_ARG;
    uint32_t slot;
    if (!slot_lookup(s, entity, &slot)) return YH_ERR_BAD_ARG;
    uint8_t row[ROW];
    const int rc = yh_download_world_aos68(s->scene, row, slot, 1);
    if (rc == YH_OK) memcpy(matrix, row, 64);
    return rc;
}

int GpuTransformCullSystem_setThreads(GpuTransformCullSystem* s, int n) {
    if (!s || n < 1) return YH_ERR_BAD_ARG;
    pool_stop(&s->pool); /* started again with the new number by the next job */
    s->threads = n > MAX_THREADS ? MAX_THREADS : n;
    return YH_OK;
}

int GpuTransformCullSystem_structureDirty(const GpuTransformCullSystem* s) { return s ? s->structure_dirty : -1; }
void GpuTransformCullSystem_clearStructureDirty(GpuTransformCullSystem* s) {
    if (s) s->structure_dirty = 0;
}
uint64_t GpuTransformCullSystem_idleFrames(const GpuTransformCullSystem* s) { return s ? s->frames_idle : 0; }
uint64_t GpuTransformCullSystem_cullOnlyFrames(const GpuTransformCullSystem* s) { return s ? s->frames_cull_only : 0; }
uint64_t GpuTransformCullSystem_cleanFrames(const GpuTransformCullSystem* s) { return s ? s->frames_clean : 0; }

void GpuTransformCullSystem_walkStats(const GpuTransformCullSystem* s, uint64_t* walks, uint64_t* skipped) {
    if (walks) *walks = s ? s->walks : 0;
    if (skipped) *skipped = s ? s->walks_skipped : 0;
}

void GpuTransformCullSystem_lastTimes(const GpuTransformCullSystem* s, double ms[6]) {
    for (int i = 0; i < 6; ++i) ms[i] = s ? s->last_ms[i] : 0.0;
}

void GpuTransformCullSystem_lastTraffic(const GpuTransformCullSystem* s, uint64_t* rows_up, uint64_t* rows_down) {
    if (rows_up) *rows_up = s ? s->last_up_rows : 0;
    if (rows_down) *rows_down = s ? s->last_down_rows : 0;
}

void GpuTransformCullSystem_destroy(GpuTransformCullSystem* s) {
    if (!s) return;
    index_join(s);
    pool_stop(&s->pool);
    if (s->scene) yh_scene_destroy(s->scene);
    if (s->camera_query) ecs_query_fini(s->camera_query);
    free(s->batches);
    free(s->walk_entities);
    free(s->mesh_dirty);
    map_free(&s->table_batch);
    free(s->slot_has_kids);
    free(s->cached_entities);
    free(s->cached_src);
    free(s->cached_counts);
    free(s->cached_levels);
    free(s->cached_tables);
    free(s->cached_cols);
    free(s->entity_of_slot);
    free(s->order);
    free_stage(s);
    free(s->batch_of);
    free(s->bounds);
    map_free(&s->slot_of);
    for (int k = 0; k < 2; ++k) {
        free(s->dirty[k]);
        free(s->stamp[k]);
    }
    for (unsigned k = 0; k < sizeof s->observers / sizeof s->observers[0]; ++k)
        if (s->observers[k] && s->world) ecs_delete(s->world, s->observers[k]);
    if (s->slot_buf_pinned) yh_host_free(s->ctx, s->slot_buf);
    else free(s->slot_buf);
    free(s->touched);
    free(s->deleted);
    for (uint32_t l = 0; l < YH_MAX_LEVELS; ++l) free(s->free_runs[l]);
    free(s->mesh_table);
    free(s->mesh_vcount);
    free((void*)s->entry_material);
    free((void*)s->entry_mesh);
    map_free(&s->mesh_index);
    free(s->cached_slot);
    free(s);
}
