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
#define ROWS_PER_THREAD_MIN 2048u /* below this many changed rows per helper a thread costs more than it saves */

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

    float planes[YH_MAX_FRUSTA * 24];
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
    int walk_cache_valid;         /* cached_* describe the batches (false after a patch: the next walk rebuilds) */
    uint32_t* cached_slot;        /* per batch: slot_begin and parent_slot as the last rebuild assigned them */
    uint64_t slot_entries;        /* entries put into slot_of (stale ones stay until the next rebuild) */
    uint64_t patches, last_patch_tables;

    int status;
    uint64_t rebuilds;
};

/* ---- small open-addressing maps (entity ids are never 0) ------------------------------------------------- */
static uint64_t hash64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

/* Large maps are cut into 16 partitions by the top bits of the hash, each a table of its own (probing wraps inside
 * the partition): map_fill_mt hands the partitions to helper threads, every one of which then inserts into 1/16 of the
 * table - a piece that stays in its core's cache - without sharing a line with another thread, so no atomics. */
#define MAP_PART_BITS 4u
#define MAP_PART_MIN_CAP 65536u
#ifdef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE_IF_ANY MADV_POPULATE_WRITE
#else
#define MADV_POPULATE_WRITE_IF_ANY 23 /* Linux 5.14+; older kernels refuse it and the pages come on first touch */
#endif
static void map_free(Map* m) {
    if (m->e && m->mapped) munmap(m->e, m->mapped);
    else free(m->e);
    m->e = NULL;
    m->mapped = 0;
    m->cap = 0;
}
/* A fresh table of a million entities is tens of MB of pages nobody has touched: sixteen threads faulting them in 4 KB
 * at a time contend for the address space's lock (15 + 25 ms of the 1M-entity rebuild were that).  Large tables come
 * from mmap, transparent huge pages asked for and the pages populated by the kernel in one go. */
#define MAP_MMAP_MIN_BYTES ((size_t)4 << 20)
static int map_init(Map* m, uint64_t n) {
    uint64_t cap = 16;
    while (cap < n * 2) cap <<= 1;
    const size_t bytes = (size_t)cap * sizeof(MapEntry);
    m->mapped = 0;
    m->e = NULL;
    if (bytes >= MAP_MMAP_MIN_BYTES) {
        void* p = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (p != MAP_FAILED) {
            (void)madvise(p, bytes, MADV_HUGEPAGE);
            (void)madvise(p, bytes, MADV_POPULATE_WRITE_IF_ANY);
            m->e = (MapEntry*)p;
            m->mapped = bytes;
        }
    }
    if (!m->e) m->e = (MapEntry*)calloc(cap, sizeof(MapEntry));
    m->cap = cap;
    m->pbits = cap >= MAP_PART_MIN_CAP ? MAP_PART_BITS : 0u;
    m->pmask = (cap >> m->pbits) - 1;
    m->overflow = 0;
    return m->e ? 0 : -1;
}
/* the same table emptied for another fill, when it is large enough for n keys and not absurdly larger */
static int map_reuse(Map* m, uint64_t n) {
    if (!m->e || m->cap < n * 2 || m->cap > n * 8 + 64) return 0;
    memset(m->e, 0, (size_t)m->cap * sizeof(MapEntry));
    m->overflow = 0;
    return 1;
}

/* home slot of a hash */
static inline MapEntry* map_home(const Map* m, uint64_t h, uint64_t* i) {
    *i = h & m->pmask;
    return m->e + (m->pbits ? (h >> (64u - m->pbits)) * (m->pmask + 1) : 0u);
}

static int map_put_hashed(Map* m, ecs_entity_t k, uint64_t h, uint32_t v) {
    uint64_t i;
    MapEntry* part = map_home(m, h, &i);
    for (uint64_t probes = 0; probes <= m->pmask; ++probes) {
        MapEntry* e = &part[i];
        if (!e->key || e->key == k) {
            const int fresh = e->key == 0;
            e->key = k;
            e->value = v;
            return fresh;
        }
        i = (i + 1) & m->pmask;
    }
    m->overflow = 1;
    return 0;
}

/* returns 1 if the key is new to the map */
static int map_put(Map* m, ecs_entity_t k, uint32_t v) { return map_put_hashed(m, k, hash64(k), v); }

static inline void map_prefetch(const Map* m, ecs_entity_t k, int for_write) {
    uint64_t i;
    const MapEntry* part = map_home(m, hash64(k), &i);
    if (for_write) __builtin_prefetch(&part[i], 1);
    else __builtin_prefetch(&part[i], 0);
}

/* keys[i] -> i for every non-zero key.  A map of a million entities is far larger than the caches and every insert a
 * miss: the home slot of the key sixteen ahead is prefetched while this one is placed (43 + 36 ms of the 1M-entity
 * rebuild were these two loops). */
static void map_fill(Map* m, const ecs_entity_t* keys, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) {
        if (i + 16 < n) map_prefetch(m, keys[i + 16], 1);
        if (keys[i]) map_put(m, keys[i], (uint32_t)i);
    }
}

typedef struct MapFillJob {
    Map* m;
    const ecs_entity_t* keys;
    uint64_t n;
    uint32_t p0, p1; /* partitions [p0, p1) are this thread's */
} MapFillJob;
static void* map_fill_parts(void* arg) {
    const MapFillJob* j = (const MapFillJob*)arg;
    Map* m = j->m;
    const uint32_t shift = 64u - m->pbits;
    for (uint64_t i = 0; i < j->n; ++i) { /* every thread reads all keys (sequential, shared) and places its own */
        const ecs_entity_t k = j->keys[i];
        if (!k) continue;
        const uint64_t h = hash64(k);
        const uint32_t part = (uint32_t)(h >> shift);
        if (part >= j->p0 && part < j->p1) map_put_hashed(m, k, h, (uint32_t)i);
    }
    return NULL;
}
/* the same by up to 16 threads (30 -> ~4 ms per million keys: each thread's share of the table fits its cache) */
static void map_fill_mt(Map* m, const ecs_entity_t* keys, uint64_t n, int threads) {
    const uint32_t parts = 1u << m->pbits;
    uint32_t T = threads < 1 ? 1u : (uint32_t)threads;
    if (T > parts) T = parts;
    if (T > MAX_THREADS) T = MAX_THREADS;
    if (!m->pbits || T < 2 || n < 65536) {
        map_fill(m, keys, n);
        return;
    }
    pthread_t th[MAX_THREADS];
    MapFillJob job[MAX_THREADS];
    int started[MAX_THREADS];
    for (uint32_t t = 0; t < T; ++t) {
        job[t].m = m;
        job[t].keys = keys;
        job[t].n = n;
        job[t].p0 = parts * t / T;
        job[t].p1 = parts * (t + 1) / T;
        started[t] = t > 0 && pthread_create(&th[t], NULL, map_fill_parts, &job[t]) == 0;
    }
    for (uint32_t t = 0; t < T; ++t)
        if (!started[t]) map_fill_parts(&job[t]);
    for (uint32_t t = 1; t < T; ++t)
        if (started[t]) pthread_join(th[t], NULL);
}

static int map_get(const Map* m, ecs_entity_t k, uint32_t* v) {
    uint64_t i;
    const MapEntry* part = map_home(m, hash64(k), &i);
    for (uint64_t probes = 0; probes <= m->pmask; ++probes) {
        const MapEntry* e = &part[i];
        if (!e->key) return 0;
        if (e->key == k) {
            *v = e->value;
            return 1;
        }
        i = (i + 1) & m->pmask;
    }
    return 0;
}

/* entity -> slot.  Entries of entities that left since the last rebuild stay in the map (patch() does not erase): an
 * entry counts only while the slot still names the entity. */
static int slot_lookup(const GpuTransformCullSystem* s, ecs_entity_t e, uint32_t* slot) {
    return s->slot_of.e && map_get(&s->slot_of, e, slot) && *slot < s->n_slots && s->entity_of_slot[*slot] == e;
}

static BoundsEntry* bounds_find(const GpuTransformCullSystem* s, ecs_entity_t k) {
    if (!s->bounds_cap) return NULL;
    uint64_t i = hash64(k) & (s->bounds_cap - 1);
    while (s->bounds[i].key) {
        if (s->bounds[i].key == k) return &s->bounds[i];
        i = (i + 1) & (s->bounds_cap - 1);
    }
    return NULL;
}

static int bounds_grow(GpuTransformCullSystem* s) {
    const uint64_t old_cap = s->bounds_cap;
    BoundsEntry* old = s->bounds;
    const uint64_t cap = old_cap ? old_cap * 2 : 1024;
    BoundsEntry* fresh = (BoundsEntry*)calloc(cap, sizeof(BoundsEntry));
    if (!fresh) return YH_ERR_OOM;
    s->bounds = fresh;
    s->bounds_cap = cap;
    for (uint64_t j = 0; j < old_cap; ++j) {
        if (!old[j].key) continue;
        uint64_t i = hash64(old[j].key) & (cap - 1);
        while (fresh[i].key) i = (i + 1) & (cap - 1);
        fresh[i] = old[j];
    }
    free(old);
    return YH_OK;
}

/* ---- the walk ---------------------------------------------------------------------------------------------- */
static int reserve_rows(GpuTransformCullSystem* s, uint64_t need) {
    if (need <= s->cap_n) return YH_OK;
    uint64_t cap = s->cap_n ? s->cap_n : 1024;
    while (cap < need) cap *= 2;
    ecs_entity_t* a = (ecs_entity_t*)realloc(s->walk_entities, cap * sizeof(ecs_entity_t));
    if (!a) return YH_ERR_OOM;
    s->walk_entities = a;
    s->cap_n = cap;
    return YH_OK;
}

/* second half of the walk, shared by the helper threads: column indices and entity ids of the whole tables (read-only
 * calls into flecs: nothing mutates the world while a system runs) */
#define COLS_PENDING (-2)
typedef struct WalkJob {
    GpuTransformCullSystem* s;
    uint32_t b0, b1;
} WalkJob;
static void* walk_finish_range(void* arg) {
    const WalkJob* j = (const WalkJob*)arg;
    GpuTransformCullSystem* s = j->s;
    for (uint32_t i = j->b0; i < j->b1; ++i) {
        Batch* b = &s->batches[i];
        if (b->col_w != COLS_PENDING) continue;
        b->col_w = ecs_table_get_column_index(s->world, b->table, s->world_id);
        b->col_l = ecs_table_get_column_index(s->world, b->table, s->local_id);
        memcpy(s->walk_entities + b->walk_begin, ecs_table_entities(b->table), sizeof(ecs_entity_t) * (size_t)b->count);
    }
    return NULL;
}
static void walk_finish(GpuTransformCullSystem* s) {
    uint32_t parts = s->n_batches / 4096u > (uint32_t)s->threads ? (uint32_t)s->threads : s->n_batches / 4096u;
    if (parts > MAX_THREADS) parts = MAX_THREADS;
    if (parts < 2) {
        WalkJob all = {s, 0, s->n_batches};
        walk_finish_range(&all);
        return;
    }
    pthread_t th[MAX_THREADS];
    WalkJob job[MAX_THREADS];
    int started[MAX_THREADS];
    for (uint32_t p = 0; p < parts; ++p) {
        job[p].s = s;
        job[p].b0 = (uint32_t)((uint64_t)s->n_batches * p / parts);
        job[p].b1 = (uint32_t)((uint64_t)s->n_batches * (p + 1) / parts);
        started[p] = p > 0 && pthread_create(&th[p], NULL, walk_finish_range, &job[p]) == 0;
    }
    for (uint32_t p = 0; p < parts; ++p)
        if (!started[p]) walk_finish_range(&job[p]);
    for (uint32_t p = 1; p < parts; ++p)
        if (started[p]) pthread_join(th[p], NULL);
}

static int walk(GpuTransformCullSystem* s, ecs_iter_t* it) {
    int rc = YH_OK;
    s->n_batches = 0;
    s->n = 0;
    s->walk_is_whole_tables = 1;
    s->walks += 1;
    while (ecs_iter_next(it)) { /* must run to the end even after an error: the iterator owns resources */
        if (rc != YH_OK || it->count == 0) continue;
        if (s->n_batches == s->cap_batches) {
            const uint32_t cap = s->cap_batches ? s->cap_batches * 2 : 256;
            Batch* b = (Batch*)realloc(s->batches, cap * sizeof(Batch));
            if (!b) {
                rc = YH_ERR_OOM;
                continue;
            }
            s->batches = b;
            s->cap_batches = cap;
        }
        if (s->n + (uint64_t)it->count > 0x40000000ull) {
            rc = YH_ERR_UNSUPPORTED;
            continue;
        }
        if ((rc = reserve_rows(s, s->n + (uint64_t)it->count)) != YH_OK) continue;
        Batch* b = &s->batches[s->n_batches++];
        b->table = it->table;
        if (!it->table || it->offset != 0 || it->count != ecs_table_count(it->table)) {
            s->walk_is_whole_tables = 0;
            b->col_w = it->table ? ecs_table_get_column_index(it->real_world, it->table, s->world_id) : -1;
            b->col_l = it->table ? ecs_table_get_column_index(it->real_world, it->table, s->local_id) : -1;
            b->world_col = ecs_field_w_size(it, ROW, 0);
            b->local_col = ecs_field_w_size(it, ROW, 1);
            memcpy(s->walk_entities + s->n, it->entities, sizeof(ecs_entity_t) * (size_t)it->count);
        } else {
            /* A whole table (the only kind a reference scene has).  Its two column indices and its entity ids are not
             * fetched here: that is two map lookups and a copy from one more cold allocation per table, a third of the
             * walk's time at 300k tables - walk_finish hands both to the helper threads.  The columns themselves are
             * fetched when a row of them is first needed (batch_cols). */
            b->col_w = b->col_l = COLS_PENDING;
            b->world_col = b->local_col = NULL;
        }
        b->up_row = NULL;
        b->src = ecs_field_src(it, 2);
        b->count = (uint32_t)it->count;
        b->walk_begin = (uint32_t)s->n;
        b->slot_begin = 0;
        b->level = -1;
        s->n += (uint64_t)it->count;
    }
    if (rc == YH_OK) walk_finish(s);
    return rc;
}

static int same_as_last_walk(const GpuTransformCullSystem* s) {
    if (!s->scene || !s->walk_cache_valid || s->n != s->cached_n || s->n_batches != s->cached_batches) return 0;
    for (uint32_t i = 0; i < s->n_batches; ++i)
        if (s->batches[i].src != s->cached_src[i] || s->batches[i].count != s->cached_counts[i]) return 0;
    return memcmp(s->walk_entities, s->cached_entities, s->n * sizeof(ecs_entity_t)) == 0;
}

static void fill_batch_of(GpuTransformCullSystem* s);

static void free_stage(GpuTransformCullSystem* s) {
    if (s->pinned) {
        yh_host_free(s->ctx, s->stage_w);
        yh_host_free(s->ctx, s->stage_l);
    } else {
        free(s->stage_w);
        free(s->stage_l);
    }
    s->stage_w = s->stage_l = NULL;
    s->pinned = 0;
}

/* ---- rebuild of the slot order ----------------------------------------------------------------------------- */
typedef struct KeyedBatch {
    uint32_t key;   /* slot of the batch's parent */
    uint32_t batch; /* index into batches[] = walk order */
} KeyedBatch;
static int by_key(const void* a, const void* b) {
    const KeyedBatch* x = (const KeyedBatch*)a;
    const KeyedBatch* y = (const KeyedBatch*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->batch < y->batch ? -1 : (x->batch > y->batch); /* stable: walk order among equals */
}

/* stable sort by key: three passes of an 11-bit radix sort (keys are slots, < 2^30) instead of qsort's ~18 rounds of
 * comparator calls per element (300k tables: 15 -> 2 ms); small inputs go through qsort */
static void sort_keyed(KeyedBatch* a, KeyedBatch* tmp, uint32_t n) {
    if (n < 1024) {
        qsort(a, n, sizeof(KeyedBatch), by_key);
        return;
    }
    KeyedBatch *src = a, *dst = tmp;
    for (uint32_t pass = 0; pass < 3; ++pass) {
        uint32_t count[2049];
        const uint32_t shift = 11u * pass;
        memset(count, 0, sizeof count);
        for (uint32_t i = 0; i < n; ++i) count[((src[i].key >> shift) & 2047u) + 1]++;
        for (uint32_t d = 0; d < 2048; ++d) count[d + 1] += count[d];
        for (uint32_t i = 0; i < n; ++i) dst[count[(src[i].key >> shift) & 2047u]++] = src[i];
        KeyedBatch* t = src;
        src = dst;
        dst = t;
    }
    if (src != a) memcpy(a, src, sizeof(KeyedBatch) * n); /* (three passes: the result is in tmp) */
}

typedef struct RowsToSlotsJob {
    GpuTransformCullSystem* s;
    const uint32_t* row_batch;
    uint64_t i0, i1; /* range of the map's table */
} RowsToSlotsJob;
static void* rows_to_slots_range(void* arg) {
    const RowsToSlotsJob* j = (const RowsToSlotsJob*)arg;
    const GpuTransformCullSystem* s = j->s;
    MapEntry* e = s->slot_of.e;
    for (uint64_t i = j->i0; i < j->i1; ++i) {
        if (!e[i].key) continue;
        const uint32_t row = e[i].value;
        const Batch* b = &s->batches[j->row_batch[row]];
        e[i].value = b->slot_begin + (row - b->walk_begin);
    }
    return NULL;
}
static void map_rows_to_slots(GpuTransformCullSystem* s, const uint32_t* row_batch) {
    const uint64_t cap = s->slot_of.cap;
    uint32_t parts = cap / 65536u > (uint64_t)s->threads ? (uint32_t)s->threads : (uint32_t)(cap / 65536u);
    if (parts > MAX_THREADS) parts = MAX_THREADS;
    if (parts < 1) parts = 1;
    pthread_t th[MAX_THREADS];
    RowsToSlotsJob job[MAX_THREADS];
    int started[MAX_THREADS];
    for (uint32_t p = 0; p < parts; ++p) {
        job[p].s = s;
        job[p].row_batch = row_batch;
        job[p].i0 = cap * p / parts;
        job[p].i1 = cap * (p + 1) / parts;
        started[p] = p > 0 && pthread_create(&th[p], NULL, rows_to_slots_range, &job[p]) == 0;
    }
    for (uint32_t p = 0; p < parts; ++p)
        if (!started[p]) rows_to_slots_range(&job[p]);
    for (uint32_t p = 1; p < parts; ++p)
        if (started[p]) pthread_join(th[p], NULL);
}

static double now_ms(void);
/* YUME_GPU_SYSTEM_TRACE=1: the rebuild's steps with their wall time on stderr */
#define TRACE_STEP(what)                                                                  \
    do {                                                                                  \
        if (trace) {                                                                      \
            const double t_now = now_ms();                                                \
            fprintf(stderr, "GpuTransformCullSystem rebuild: %-28s %8.3f ms\n", what, t_now - t_step); \
            t_step = t_now;                                                               \
        }                                                                                 \
    } while (0)

static int rebuild(GpuTransformCullSystem* s) {
    const int trace = getenv("YUME_GPU_SYSTEM_TRACE") != NULL;
    double t_step = trace ? now_ms() : 0.0;
    const uint64_t n = s->n;
    const uint32_t nb = s->n_batches;
    int rc = YH_OK;
    Map* const rows = &s->slot_of; /* entity -> walk row while the order is worked out, entity -> slot from then on */
    uint32_t* parent_batch = (uint32_t*)malloc(sizeof(uint32_t) * (nb ? nb : 1));
    uint32_t* parent_row = (uint32_t*)malloc(sizeof(uint32_t) * (nb ? nb : 1)); /* walk row of the parent */
    uint32_t* key = (uint32_t*)malloc(sizeof(uint32_t) * (nb ? nb : 1));
    /* (slots = rows + the spare slots of every level: at most an eighth more, and SPARE_MIN per level) */
    uint32_t* stack = (uint32_t*)malloc(sizeof(uint32_t) * (nb ? nb : 1));
    uint32_t* row_batch = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n ? n : 1)); /* walk row -> batch */
    uint32_t* by_level = (uint32_t*)malloc(sizeof(uint32_t) * (nb ? nb : 1));        /* batches grouped by level, walk order inside */
    KeyedBatch* keyed = (KeyedBatch*)malloc(sizeof(KeyedBatch) * (nb ? nb : 1) * 2); /* (second half: the sort's other buffer) */
    uint32_t level_offsets[YH_MAX_LEVELS + 1];
    free(s->order);
    s->order = (uint32_t*)malloc(sizeof(uint32_t) * (nb ? nb : 1));
    uint32_t* parent_index = NULL;
    if (!parent_batch || !parent_row || !key || !stack || !row_batch || !by_level || !keyed || !s->order) {
        rc = YH_ERR_OOM;
        goto done;
    }
    /* entity -> walk row (filled by the helper threads, a partition of the table each), walk row -> batch; then the
     * parent of every table in two prefetched passes: its row through the map, its batch through the row table */
    {   /* The list of slots that sparse uploads and changed rows cross PCIe in (page-locked when possible) doubles as the
         * parent index of this rebuild: yh_upload_hierarchy then copies from page-locked memory (9 -> 1 ms at 1M). */
        const uint64_t want = n + n / 8 + (uint64_t)SPARE_MIN * (YH_MAX_LEVELS + 1);
        if (s->slot_buf_cap < want) {
            if (s->slot_buf_pinned) yh_host_free(s->ctx, s->slot_buf);
            else free(s->slot_buf);
            s->slot_buf = (uint32_t*)yh_host_alloc(s->ctx, sizeof(uint32_t) * want);
            s->slot_buf_pinned = s->slot_buf != NULL;
            if (!s->slot_buf) s->slot_buf = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)want);
            s->slot_buf_cap = s->slot_buf ? want : 0;
            if (!s->slot_buf) {
                rc = YH_ERR_OOM;
                goto done;
            }
        }
        parent_index = s->slot_buf;
    }
    if (!map_reuse(rows, n + n / 8)) { /* (an eighth of headroom for the entities later patches bring) */
        map_free(rows);
        if (map_init(rows, n + n / 8) != 0) {
            rc = YH_ERR_OOM;
            goto done;
        }
    }
    TRACE_STEP("allocations, the map's table");
    map_fill_mt(rows, s->walk_entities, n, s->threads);
    if (rows->overflow) {
        rc = YH_ERR_OOM;
        goto done;
    }
    TRACE_STEP("entity -> walk row map");
    for (uint32_t b = 0; b < nb; ++b)
        for (uint32_t r = 0; r < s->batches[b].count; ++r) row_batch[s->batches[b].walk_begin + r] = b;
    TRACE_STEP("walk row -> table");
    for (uint32_t b = 0; b < nb; ++b) {
        uint32_t row;
        if (b + 16 < nb) map_prefetch(rows, s->batches[b + 16].src, 0);
        parent_row[b] = map_get(rows, s->batches[b].src, &row) ? row : YH_NO_PARENT;
    }
    for (uint32_t b = 0; b < nb; ++b) {
        if (b + 16 < nb && parent_row[b + 16] != YH_NO_PARENT) __builtin_prefetch(&row_batch[parent_row[b + 16]]);
        /* no row: the scene root, or any holder of a WorldTransform without a local one */
        parent_batch[b] = parent_row[b] != YH_NO_PARENT ? row_batch[parent_row[b]] : YH_NO_PARENT;
    }
    TRACE_STEP("parents of tables");
    /* levels, memoised along parent chains (ChildOf is acyclic) */
    uint32_t n_levels = 0;
    for (uint32_t b = 0; b < nb; ++b) {
        uint32_t top = 0, cur = b;
        while (s->batches[cur].level < 0 && parent_batch[cur] != YH_NO_PARENT) {
            stack[top++] = cur;
            cur = parent_batch[cur];
            if (top >= nb) break;
        }
        if (s->batches[cur].level < 0) s->batches[cur].level = 0;
        int32_t lvl = s->batches[cur].level;
        while (top) s->batches[stack[--top]].level = ++lvl;
        if ((uint32_t)s->batches[b].level + 1 > n_levels) n_levels = (uint32_t)s->batches[b].level + 1;
    }
    if (n_levels > YH_MAX_LEVELS) {
        rc = YH_ERR_UNSUPPORTED;
        goto done;
    }
    TRACE_STEP("levels");
    /* level by level: order the level's batches by the slot of their parent, hand out slots; with patching on every
     * level's range ends in spare slots for the tables that change later (patch()) */
    const int spare_slots = s->patching && s->incremental;
    if (spare_slots) /* (levels nobody lives on yet: spare slots only) */
        n_levels = n_levels + SPARE_LEVELS <= YH_MAX_LEVELS ? n_levels + SPARE_LEVELS : YH_MAX_LEVELS;
    uint32_t placed = 0, slot = 0;
    uint32_t level_first[YH_MAX_LEVELS + 2];
    { /* the batches of every level, walk order inside a level (a counting sort by level) */
        memset(level_first, 0, sizeof level_first);
        for (uint32_t b = 0; b < nb; ++b) level_first[(uint32_t)s->batches[b].level + 1]++;
        for (uint32_t l = 0; l <= YH_MAX_LEVELS; ++l) level_first[l + 1] += level_first[l];
        uint32_t cursor[YH_MAX_LEVELS + 1];
        memcpy(cursor, level_first, sizeof cursor);
        for (uint32_t b = 0; b < nb; ++b) by_level[cursor[(uint32_t)s->batches[b].level]++] = b;
    }
    for (uint32_t lvl = 0; lvl < n_levels; ++lvl) {
        level_offsets[lvl] = slot;
        const uint32_t first = placed;
        for (uint32_t q = level_first[lvl]; q < level_first[lvl + 1]; ++q) {
            const uint32_t b = by_level[q];
            if (lvl == 0) {
                key[b] = 0;
            } else {
                const Batch* pb = &s->batches[parent_batch[b]];
                key[b] = pb->slot_begin + (parent_row[b] - pb->walk_begin); /* parent's slot: assigned one level up */
            }
            keyed[placed].key = key[b];
            keyed[placed].batch = b;
            placed++;
        }
        sort_keyed(keyed + first, keyed + nb + first, placed - first); /* stable: walk order among equals */
        for (uint32_t i = first; i < placed; ++i) s->order[i] = keyed[i].batch;
        for (uint32_t i = first; i < placed; ++i) {
            Batch* b = &s->batches[s->order[i]];
            b->slot_begin = slot;
            const uint32_t p = lvl == 0 ? YH_NO_PARENT : key[s->order[i]];
            b->parent_slot = p;
            for (uint32_t r = 0; r < b->count; ++r) parent_index[slot + r] = p;
            slot += b->count;
        }
        s->level_fill[lvl] = slot;
        if (spare_slots) {
            const uint32_t rows = slot - level_offsets[lvl], spare = rows / 8u > SPARE_MIN ? rows / 8u : SPARE_MIN;
            /* a dead slot is computed like any other: it needs a parent inside the level before its own */
            for (uint32_t r = 0; r < spare; ++r) parent_index[slot + r] = lvl == 0 ? YH_NO_PARENT : level_offsets[lvl - 1];
            slot += spare;
        }
    }
    level_offsets[n_levels] = slot;
    const uint64_t N = slot; /* slots of the device scene */
    memcpy(s->level_begin, level_offsets, sizeof(uint32_t) * (n_levels + 1));
    memset(s->n_free, 0, sizeof s->n_free); /* (a rebuild compacts: no dead runs) */
    TRACE_STEP("sort by parent slot, slots");

    const uint32_t want_bounds = s->mesh_id ? YH_BOUNDS_MESH_TABLE : YH_BOUNDS_PER_ENTITY;
    if (!s->scene || s->scene_entities != N || s->scene_levels != n_levels || s->scene_bounds_kind != want_bounds) {
        if (s->scene) yh_scene_destroy(s->scene);
        s->scene = NULL;
        yh_scene_desc d;
        memset(&d, 0, sizeof d);
        d.n_entities = N;
        d.n_levels = n_levels;
        d.n_frusta = YH_MAX_FRUSTA;
        d.local_kind = YH_LOCAL_MAT4;
        d.bounds_kind = want_bounds;
        d.max_meshes = want_bounds == YH_BOUNDS_MESH_TABLE ? MESH_TABLE_MAX : 0;
        s->scene_bounds_kind = want_bounds;
        s->n_meshes = 0;
        if ((rc = yh_scene_create(s->ctx, &d, &s->scene)) != YH_OK) goto done;
        s->scene_entities = N;
        s->scene_levels = n_levels;
        /* the staging columns every row crosses PCIe through: page-locked, so the copies run at link speed */
        free_stage(s);
        s->stage_w = (uint8_t*)yh_host_alloc(s->ctx, N * ROW);
        s->stage_l = (uint8_t*)yh_host_alloc(s->ctx, N * ROW);
        s->pinned = s->stage_w && s->stage_l;
        if (!s->pinned) { /* no page-locked memory to be had: ordinary memory works, slower */
            free_stage(s);
            s->stage_w = (uint8_t*)malloc((size_t)N * ROW);
            s->stage_l = (uint8_t*)malloc((size_t)N * ROW);
        }
        ecs_entity_t* eo = (ecs_entity_t*)realloc(s->entity_of_slot, (size_t)N * sizeof(ecs_entity_t));
        if (eo) s->entity_of_slot = eo;
        uint32_t* bo = (uint32_t*)realloc(s->batch_of, (size_t)N * sizeof(uint32_t));
        if (bo) s->batch_of = bo;
        if (!s->stage_w || !s->stage_l || !eo || !bo) {
            rc = YH_ERR_OOM;
            goto done;
        }
    }
    if (spare_slots) /* the rows of spare slots travel with the whole columns: defined bytes */
        for (uint32_t lvl = 0; lvl < n_levels; ++lvl) {
            const size_t first = (size_t)s->level_fill[lvl] * ROW, bytes = (size_t)(level_offsets[lvl + 1] - s->level_fill[lvl]) * ROW;
            memset(s->stage_w + first, 0, bytes);
            memset(s->stage_l + first, 0, bytes);
        }
    s->n_slots = N;
    TRACE_STEP("device scene, staging columns");
    if ((rc = yh_upload_hierarchy(s->scene, parent_index, level_offsets)) != YH_OK) goto done;
    TRACE_STEP("hierarchy up");
    if (spare_slots) { /* (a scene that is reused may carry the dead slots of earlier patches) */
        if ((rc = yh_scene_set_alive_range(s->scene, 0, N, 1)) != YH_OK) goto done;
        for (uint32_t lvl = 0; lvl < n_levels; ++lvl)
            if ((rc = yh_scene_set_alive_range(s->scene, s->level_fill[lvl], level_offsets[lvl + 1] - s->level_fill[lvl], 0)) != YH_OK)
                goto done;
    }
    TRACE_STEP("alive slots");
    memset(s->entity_of_slot, 0, (size_t)N * sizeof(ecs_entity_t)); /* 0: the slot holds no entity */
    memset(s->batch_of, 0xFF, (size_t)N * sizeof(uint32_t));
    TRACE_STEP("slot tables cleared");
    for (uint32_t b = 0; b < nb; ++b)
        memcpy(s->entity_of_slot + s->batches[b].slot_begin, s->walk_entities + s->batches[b].walk_begin,
               sizeof(ecs_entity_t) * s->batches[b].count);
    s->n_levels = n_levels;
    s->bounds_stale = 1;
    fill_batch_of(s);
    TRACE_STEP("entities and tables of slots");
    /* entity -> slot: the map already holds every entity with its walk row; a row's slot is its batch's first slot + its
     * offset in the batch.  The values are rewritten in place, a share of the table per helper thread. */
    map_rows_to_slots(s, row_batch);
    s->slot_entries = n;
    TRACE_STEP("entity -> slot map");
    for (int k = 0; k < 2; ++k) {
        free(s->stamp[k]);
        s->stamp[k] = (uint32_t*)calloc((size_t)(N ? N : 1), sizeof(uint32_t));
    }
    s->epoch = 0;
    if (!s->stamp[0] || !s->stamp[1] || !s->slot_buf) {
        rc = YH_ERR_OOM;
        goto done;
    }
    if ((rc = yh_scene_track_changes(s->scene, s->incremental)) != YH_OK) goto done;
    TRACE_STEP("stamps, slot list, tracking");

    /* remember this walk */
    {
        /* (the walk's entity ids become the remembered ones by a swap of the two buffers: 8 MB not copied) */
        ecs_entity_t* ce = s->walk_entities;
        s->walk_entities = s->cached_entities;
        s->cached_entities = ce;
        {
            const uint64_t c = s->cap_n;
            s->cap_n = s->cap_cached;
            s->cap_cached = c;
        }
        ecs_entity_t* cs = (ecs_entity_t*)realloc(s->cached_src, (size_t)(nb ? nb : 1) * sizeof(ecs_entity_t));
        uint32_t* cc = (uint32_t*)realloc(s->cached_counts, (size_t)(nb ? nb : 1) * sizeof(uint32_t));
        int32_t* cl = (int32_t*)realloc(s->cached_levels, (size_t)(nb ? nb : 1) * sizeof(int32_t));
        ecs_table_t** ct = (ecs_table_t**)realloc(s->cached_tables, (size_t)(nb ? nb : 1) * sizeof(ecs_table_t*));
        int32_t* ccol = (int32_t*)realloc(s->cached_cols, (size_t)(nb ? nb : 1) * 2 * sizeof(int32_t));
        uint32_t* cslot = (uint32_t*)realloc(s->cached_slot, (size_t)(nb ? nb : 1) * 2 * sizeof(uint32_t));
        if (cslot) s->cached_slot = cslot;
        if (ct) s->cached_tables = ct;
        if (ccol) s->cached_cols = ccol;
        if (cl) s->cached_levels = cl;
        if (cs) s->cached_src = cs;
        if (cc) s->cached_counts = cc;
        if (!ce || !cs || !cc || !cl || !ct || !ccol || !cslot) {
            s->cached_n = 0;
            s->cached_batches = 0;
            rc = YH_ERR_OOM;
            goto done;
        }
        for (uint32_t b = 0; b < nb; ++b) {
            cs[b] = s->batches[b].src;
            cc[b] = s->batches[b].count;
            cl[b] = s->batches[b].level;
            ct[b] = s->batches[b].table;
            ccol[2 * b] = s->batches[b].col_w;
            ccol[2 * b + 1] = s->batches[b].col_l;
            cslot[2 * b] = s->batches[b].slot_begin;
            cslot[2 * b + 1] = s->batches[b].parent_slot;
        }
        s->cached_n = n;
        s->cached_batches = nb;
        s->walk_cache_valid = 1;
    }
    s->rebuilds += 1;
    TRACE_STEP("walk remembered");
done:
    free(parent_batch);
    free(parent_row);
    free(key);
    free(stack);
    free(row_batch);
    free(by_level);
    free(keyed);
    if (rc != YH_OK) {
        s->cached_n = 0, s->cached_batches = 0;
        map_free(&s->slot_of); /* (it may hold walk rows, not slots) */
    }
    return rc;
}

/* (the mesh table is kept on the host between uploads: patch() appends to it) */
static int mesh_table_reset(GpuTransformCullSystem* s) {
    if (!s->mesh_table) s->mesh_table = (float*)malloc(sizeof(float) * 6 * (size_t)MESH_TABLE_MAX);
    if (!s->mesh_vcount) s->mesh_vcount = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)MESH_TABLE_MAX);
    if (!s->entry_material) s->entry_material = (const void**)malloc(sizeof(void*) * (size_t)MESH_TABLE_MAX);
    if (!s->entry_mesh) s->entry_mesh = (const void**)malloc(sizeof(void*) * (size_t)MESH_TABLE_MAX);
    map_free(&s->mesh_index);
    if (!s->mesh_table || !s->mesh_vcount || !s->entry_material || !s->entry_mesh || map_init(&s->mesh_index, 4096) != 0) return YH_ERR_OOM;
    memset(s->mesh_table, 0, sizeof(float) * 6); /* entry 0: no mesh = an empty box at the origin */
    s->mesh_vcount[0] = 0;
    s->entry_material[0] = s->entry_mesh[0] = NULL;
    s->mesh_count = 1;
    return YH_OK;
}
static int mesh_table_append(GpuTransformCullSystem* s, const float mm[6], uint32_t vertices, uint32_t* id) {
    if (s->mesh_count == MESH_TABLE_MAX) return YH_ERR_CAPACITY;
    *id = s->mesh_count++;
    memcpy(s->mesh_table + 6 * (size_t)*id, mm, 24);
    s->mesh_vcount[*id] = vertices;
    s->entry_material[*id] = s->entry_mesh[*id] = NULL;
    return YH_OK;
}
/* The draw loop rebinds the pipeline when the material changes and the vertex buffer when the mesh does
 * (RenderApi.zig:1411, 1434): a table entry - the key the device buckets a draw list by - stands for one (material,
 * mesh) pair.  The map is keyed by a 64-bit mix of the two pointers; a hit is checked against the entry's own pair and
 * a (never yet seen) collision of two pairs moves on to the next key of the chain. */
static uint64_t pair_key(const void* material, const void* mesh, uint32_t attempt) {
    uint64_t k = (uint64_t)(uintptr_t)mesh ^ (hash64((uint64_t)(uintptr_t)material + 0x9E3779B97F4A7C15ull * (attempt + 1u)) << 1);
    if (!material && !attempt) k = (uint64_t)(uintptr_t)mesh; /* meshes only: the pointer itself, as before */
    return k ? k : 1;
}
/* table entry of a (material, mesh) pair: looked up, or appended from the mesh's BoundingBox{mins, maxs} */
static int mesh_table_entry(GpuTransformCullSystem* s, const uint8_t* mesh, const void* material, uint32_t* id) {
    uint64_t key = 0;
    for (uint32_t attempt = 0;; ++attempt) {
        key = pair_key(material, mesh, attempt);
        if (!map_get(&s->mesh_index, (ecs_entity_t)key, id)) break;
        if (s->entry_mesh[*id] == (const void*)mesh && s->entry_material[*id] == material) return YH_OK;
        if (attempt == 64) return YH_ERR_STATE;
    }
    if (s->mesh_index.cap < (uint64_t)s->mesh_count * 2 + 2) {
        Map bigger = {0};
        if (map_init(&bigger, s->mesh_index.cap) != 0) return YH_ERR_OOM; /* (map_init doubles what it is asked for) */
        for (uint64_t q = 0; q < s->mesh_index.cap; ++q)
            if (s->mesh_index.e[q].key) map_put(&bigger, s->mesh_index.e[q].key, s->mesh_index.e[q].value);
        map_free(&s->mesh_index);
        s->mesh_index = bigger;
    }
    float mm[6];
    uint32_t vertices = 0;
    memcpy(mm, mesh + s->mesh_bounds_offset, 24);
    if (s->mesh_vertices_offset >= 0) {
        uint64_t vc;
        memcpy(&vc, mesh + s->mesh_vertices_offset, 8); /* vertices_count: usize */
        vertices = vc > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)vc;
    }
    const int rc = mesh_table_append(s, mm, vertices, id);
    if (rc == YH_OK) {
        s->entry_material[*id] = material;
        s->entry_mesh[*id] = mesh;
        map_put(&s->mesh_index, (ecs_entity_t)key, *id);
    }
    return rc;
}
/* mesh ids of one batch's rows into ids[0 .. count): the Mesh column where the table has one, 0 elsewhere, an entry of
 * its own for an entity with setBounds */
static int mesh_ids_of_batch(GpuTransformCullSystem* s, const Batch* b, uint32_t* ids) {
    memset(ids, 0, sizeof(uint32_t) * b->count);
    const int32_t col = b->table ? ecs_table_get_column_index(s->world, b->table, s->mesh_id) : -1;
    if (col >= 0) {
        const uint8_t* const* refs = (const uint8_t* const*)ecs_table_get_column(b->table, col, 0);
        const int32_t mcol = s->material_id ? ecs_table_get_column_index(s->world, b->table, s->material_id) : -1;
        const void* const* mats = mcol >= 0 ? (const void* const*)ecs_table_get_column(b->table, mcol, 0) : NULL;
        const uint8_t* prev = NULL;
        const void* prev_mat = NULL;
        uint32_t prev_id = 0;
        for (uint32_t r = 0; r < b->count; ++r) { /* one hash lookup per run of rows that share a mesh and a material */
            const uint8_t* mesh = refs[r];
            const void* mat = mats ? mats[r] : NULL;
            if (!mesh) continue;
            if (mesh != prev || mat != prev_mat) {
                const int rc = mesh_table_entry(s, mesh, mat, &prev_id);
                if (rc != YH_OK) return rc;
                prev = mesh;
                prev_mat = mat;
            }
            ids[r] = prev_id;
        }
    }
    for (uint32_t r = 0; r < b->count && s->bounds_n; ++r) { /* per-entity overrides */
        const BoundsEntry* o = bounds_find(s, s->entity_of_slot[b->slot_begin + r]);
        if (!o) continue;
        const int rc = mesh_table_append(s, o->mm, 0, &ids[r]);
        if (rc != YH_OK) return rc;
    }
    return YH_OK;
}
static int mesh_table_upload(GpuTransformCullSystem* s) {
    int rc = yh_upload_mesh_aabbs(s->scene, s->mesh_table, s->mesh_count);
    if (rc == YH_OK && s->mesh_vertices_offset >= 0) rc = yh_upload_vertex_counts(s->scene, s->mesh_vcount, s->mesh_count);
    if (rc == YH_OK) s->n_meshes = s->mesh_count;
    return rc;
}

/* Bounds from the Mesh component (engine/components/mesh.zig: `Mesh{ ref: *meshes.Mesh }`; the AABB is
 * meshes.Mesh.bounds, engine/meshes.zig:72-95, 104-110): one table entry per distinct mesh, entry 0 the empty box of
 * entities that have none, a u32 id per entity.  setBounds overrides get entries of their own. */
typedef struct MeshRefsJob {
    GpuTransformCullSystem* s;
    const uint8_t** refs; /* per slot: the Mesh component's pointer (NULL: no mesh) */
    const void** mats;    /* per slot: the Material component's pointer, or NULL (array absent without a Material id) */
    uint32_t b0, b1;
} MeshRefsJob;
/* the Mesh / Material columns of a range of batches into per-slot arrays: two column lookups per table and one cold
 * column each, shared by the helper threads (read-only calls into flecs) */
static void* mesh_refs_range(void* arg) {
    const MeshRefsJob* j = (const MeshRefsJob*)arg;
    GpuTransformCullSystem* s = j->s;
    for (uint32_t bi = j->b0; bi < j->b1; ++bi) {
        const Batch* b = &s->batches[bi];
        if (!b->table || !b->count) continue;
        const int32_t col = ecs_table_get_column_index(s->world, b->table, s->mesh_id);
        if (col < 0) continue;
        memcpy(j->refs + b->slot_begin, ecs_table_get_column(b->table, col, 0), sizeof(void*) * b->count);
        const int32_t mcol = j->mats ? ecs_table_get_column_index(s->world, b->table, s->material_id) : -1;
        if (mcol >= 0) memcpy(j->mats + b->slot_begin, ecs_table_get_column(b->table, mcol, 0), sizeof(void*) * b->count);
    }
    return NULL;
}

static int upload_bounds_from_meshes(GpuTransformCullSystem* s) {
    const uint64_t n = s->n_slots;
    int rc = mesh_table_reset(s);
    uint32_t* ids = (uint32_t*)calloc((size_t)(n ? n : 1), sizeof(uint32_t));
    const uint8_t** refs = (const uint8_t**)calloc((size_t)(n ? n : 1), sizeof(void*));
    const void** mats = s->material_id ? (const void**)calloc((size_t)(n ? n : 1), sizeof(void*)) : NULL;
    if (!ids || !refs || (s->material_id && !mats)) rc = YH_ERR_OOM;
    if (rc != YH_OK) goto done;
    {   /* (1) the components' pointers, per slot, by the helper threads */
        uint32_t parts = s->n_batches / 4096u > (uint32_t)s->threads ? (uint32_t)s->threads : s->n_batches / 4096u;
        if (parts > MAX_THREADS) parts = MAX_THREADS;
        if (parts < 1) parts = 1;
        pthread_t th[MAX_THREADS];
        MeshRefsJob job[MAX_THREADS];
        int started[MAX_THREADS];
        for (uint32_t p = 0; p < parts; ++p) {
            job[p].s = s;
            job[p].refs = refs;
            job[p].mats = mats;
            job[p].b0 = (uint32_t)((uint64_t)s->n_batches * p / parts);
            job[p].b1 = (uint32_t)((uint64_t)s->n_batches * (p + 1) / parts);
            started[p] = p > 0 && pthread_create(&th[p], NULL, mesh_refs_range, &job[p]) == 0;
        }
        for (uint32_t p = 0; p < parts; ++p)
            if (!started[p]) mesh_refs_range(&job[p]);
        for (uint32_t p = 1; p < parts; ++p)
            if (started[p]) pthread_join(th[p], NULL);
    }
    {   /* (2) table entries: one hash lookup per run of slots that share a mesh and a material (this thread: the table grows) */
        const uint8_t* prev = NULL;
        const void* prev_mat = NULL;
        uint32_t prev_id = 0;
        for (uint64_t e = 0; e < n && rc == YH_OK; ++e) {
            const uint8_t* mesh = refs[e];
            const void* mat = mats ? mats[e] : NULL;
            if (!mesh) continue;
            if (mesh != prev || mat != prev_mat) {
                rc = mesh_table_entry(s, mesh, mat, &prev_id);
                prev = mesh;
                prev_mat = mat;
            }
            ids[e] = prev_id;
        }
        for (uint64_t e = 0; e < n && rc == YH_OK && s->bounds_n; ++e) { /* per-entity overrides: entries of their own */
            const BoundsEntry* o = s->entity_of_slot[e] ? bounds_find(s, s->entity_of_slot[e]) : NULL;
            if (o) rc = mesh_table_append(s, o->mm, 0, &ids[e]);
        }
    }
    if (rc != YH_OK) goto done;
    /* ids first while they are all 0 .. old table, then the table, then the ids (the table may not shrink below an
     * id in use): zero ids are valid for every table */
    if (s->n_meshes) {
        memset(refs, 0, (size_t)n * sizeof(uint32_t)); /* (done with: n zero ids) */
        rc = yh_upload_mesh_ids(s->scene, (const uint32_t*)refs, 0, n);
    }
    if (rc == YH_OK) rc = mesh_table_upload(s);
    if (rc == YH_OK) rc = yh_upload_mesh_ids(s->scene, ids, 0, n);
    if (rc == YH_OK) s->bounds_stale = 0;
done:
    free(ids);
    free(refs);
    free(mats);
    return rc;
}

typedef struct BoundsJob {
    const GpuTransformCullSystem* s;
    float* planes;
    uint64_t e0, e1;
} BoundsJob;
static void* bounds_thread(void* arg) {
    const BoundsJob* j = (const BoundsJob*)arg;
    const GpuTransformCullSystem* s = j->s;
    const uint64_t n = s->n_slots;
    for (uint64_t e = j->e0; e < j->e1; ++e) {
        const BoundsEntry* b = s->entity_of_slot[e] ? bounds_find(s, s->entity_of_slot[e]) : NULL;
        if (b)
            for (int k = 0; k < 6; ++k) j->planes[(size_t)k * n + e] = b->mm[k];
    }
    return NULL;
}

static int upload_bounds(GpuTransformCullSystem* s) {
    if (s->scene_bounds_kind == YH_BOUNDS_MESH_TABLE) return upload_bounds_from_meshes(s);
    const uint64_t n = s->n_slots;
    float* planes = (float*)calloc((size_t)n * 6, sizeof(float)); /* no bounds = empty box at the origin */
    if (!planes) return YH_ERR_OOM;
    if (s->bounds_n) { /* one hash lookup per slot, every one a cache miss: helpers share them (read-only) */
        uint64_t parts = n / ROWS_PER_THREAD_MIN > (uint64_t)s->threads ? (uint64_t)s->threads : n / ROWS_PER_THREAD_MIN;
        if (parts < 1) parts = 1;
        pthread_t th[MAX_THREADS];
        BoundsJob job[MAX_THREADS];
        int started[MAX_THREADS];
        for (uint64_t p = 0; p < parts; ++p) {
            job[p].s = s;
            job[p].planes = planes;
            job[p].e0 = n * p / parts;
            job[p].e1 = n * (p + 1) / parts;
            started[p] = p > 0 && pthread_create(&th[p], NULL, bounds_thread, &job[p]) == 0;
        }
        for (uint64_t p = 0; p < parts; ++p)
            if (!started[p]) bounds_thread(&job[p]);
        for (uint64_t p = 1; p < parts; ++p)
            if (started[p]) pthread_join(th[p], NULL);
    }
    const float* mn[3] = {planes, planes + n, planes + 2 * n};
    const float* mx[3] = {planes + 3 * n, planes + 4 * n, planes + 5 * n};
    const int rc = yh_upload_aabb(s->scene, mn, mx, 0, n);
    free(planes);
    if (rc == YH_OK) s->bounds_stale = 0;
    return rc;
}

/* when slots are restored from the cached order only slot_begin / level need refreshing: same walk, same order */
static void restore_slots(GpuTransformCullSystem* s) {
    for (uint32_t i = 0; i < s->n_batches; ++i) {
        Batch* b = &s->batches[i];
        b->level = s->cached_levels[i];
        b->slot_begin = s->cached_slot[2 * i]; /* (levels end in spare slots: not a running sum of the counts) */
        b->parent_slot = s->cached_slot[2 * i + 1];
    }
}

/* column pointers of a batch restored without a walk */
static Batch* batch_cols(Batch* b) {
    if (!b->world_col) {
        b->world_col = ecs_table_get_column(b->table, b->col_w, 0);
        b->local_col = ecs_table_get_column(b->table, b->col_l, 0);
    }
    return b;
}

/* batch holding a slot: binary search over the batches in slot order */
static const Batch* batch_of_slot(GpuTransformCullSystem* s, uint32_t slot) {
    if (s->batch_of) return batch_cols(&s->batches[s->batch_of[slot]]);
    uint32_t lo = 0, hi = s->n_batches - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) / 2;
        if (s->batches[s->order[mid]].slot_begin <= slot) lo = mid;
        else hi = mid - 1;
    }
    return batch_cols(&s->batches[s->order[lo]]);
}

/* slot -> batch table; rebuilt whenever slots are (re)assigned */
static void fill_batch_of(GpuTransformCullSystem* s) {
    if (!s->batch_of) return;
    for (uint32_t b = 0; b < s->n_batches; ++b)
        for (uint32_t r = 0; r < s->batches[b].count; ++r) s->batch_of[s->batches[b].slot_begin + r] = b;
}

/* Structural events: an id came to or left a table that holds one of the two components (entities created, deleted,
 * re-parented, or moved between archetypes by an unrelated component: in each case rows enter, leave or move inside
 * tables this system has cached), or such a table was created or deleted. */
#define MAX_TOUCHED 65536u
static void touch_table(GpuTransformCullSystem* s, const ecs_world_t* w, ecs_table_t* t) {
    if (!t) return;
    if (ecs_table_get_column_index(w, t, s->local_id) < 0 && ecs_table_get_column_index(w, t, s->world_id) < 0) return;
    s->structure_dirty = 1;
    if (s->need_full || (s->n_touched && s->touched[s->n_touched - 1] == t)) return; /* (runs of events on one table) */
    if (s->n_touched == s->cap_touched) {
        const uint32_t cap = s->cap_touched ? s->cap_touched * 2 : 256;
        ecs_table_t** g = cap > MAX_TOUCHED ? NULL : (ecs_table_t**)realloc(s->touched, sizeof(ecs_table_t*) * cap);
        if (!g) { /* too many tables for a patch (or no memory): the next frame walks the query */
            s->need_full = 1;
            return;
        }
        s->touched = g;
        s->cap_touched = cap;
    }
    s->touched[s->n_touched++] = t;
}
static void note_structure(ecs_iter_t* it) {
    GpuTransformCullSystem* s = (GpuTransformCullSystem*)it->ctx;
    touch_table(s, it->real_world, it->table);
    touch_table(s, it->real_world, it->other_table);
}
static void note_table(ecs_iter_t* it) {
    GpuTransformCullSystem* s = (GpuTransformCullSystem*)it->ctx;
    s->structure_dirty = 1;
    if (it->event != EcsOnTableDelete || s->need_full) return;
    /* The pointer dangles from here on (and may be handed out again for another table): it leaves the list of touched
     * tables, which patch() dereferences, and joins the list of deleted ones, which it only compares with. */
    ecs_table_t* t = it->table;
    uint32_t kept = 0;
    for (uint32_t i = 0; i < s->n_touched; ++i)
        if (s->touched[i] != t) s->touched[kept++] = s->touched[i];
    s->n_touched = kept;
    if (s->n_deleted && s->deleted[s->n_deleted - 1] == t) return; /* (one event per component the table holds) */
    if (s->n_deleted == s->cap_deleted) {
        const uint32_t cap = s->cap_deleted ? s->cap_deleted * 2 : 256;
        ecs_table_t** g = cap > MAX_TOUCHED ? NULL : (ecs_table_t**)realloc(s->deleted, sizeof(ecs_table_t*) * cap);
        if (!g) {
            s->need_full = 1;
            return;
        }
        s->deleted = g;
        s->cap_deleted = cap;
    }
    s->deleted[s->n_deleted++] = t;
}

/* OnSet observers: the same events that run the components' onSet hooks */
static void note_set(ecs_iter_t* it, int k) {
    GpuTransformCullSystem* s = (GpuTransformCullSystem*)it->ctx;
    if (!s->incremental) return;
    if (s->n_dirty[k] + (uint64_t)it->count > s->cap_dirty[k]) {
        uint64_t cap = s->cap_dirty[k] ? s->cap_dirty[k] * 2 : 1024;
        while (cap < s->n_dirty[k] + (uint64_t)it->count) cap *= 2;
        ecs_entity_t* d = (ecs_entity_t*)realloc(s->dirty[k], cap * sizeof(ecs_entity_t));
        if (!d) {
            s->cached_n = 0; /* out of memory: force a rebuild, i.e. a full upload, next frame */
            return;
        }
        s->dirty[k] = d;
        s->cap_dirty[k] = cap;
    }
    memcpy(s->dirty[k] + s->n_dirty[k], it->entities, sizeof(ecs_entity_t) * (size_t)it->count);
    s->n_dirty[k] += (uint64_t)it->count;
}
static void on_set_local(ecs_iter_t* it) { note_set(it, 0); }
static void on_set_world(ecs_iter_t* it) { note_set(it, 1); }

/* The walk-free frame.  Iterating ~10^5 small tables costs flecs ~0.3 us each (a 1M-entity random tree: 130 ms per
 * frame, far more than the GPU frame), so when nothing structural happened the last walk is reused: the query's cache
 * has not matched or unmatched a table since (ecs_query_match_count: also covers tables that emptied, filled, were
 * deleted, or whose `Up` source changed), and every cached table still holds exactly the entities it held (count +
 * ids compared: catches rows that moved between tables because an unrelated component came or went).  Column
 * pointers are fetched afresh.  Returns 1 if the batches were restored, 0 if the iterator has to be walked. */
static int restore_walk(GpuTransformCullSystem* s, const ecs_iter_t* it) {
    if (!s->scene || !s->cached_n || !s->walk_cache_valid || !s->walk_is_whole_tables || !it->query) return 0;
    if (it->query->cache_kind != EcsQueryCacheAll) return 0; /* no cache, no match count */
    if (ecs_query_match_count(it->query) != s->cached_match_count) return 0;
    uint64_t row = 0;
    for (uint32_t b = 0; b < s->cached_batches; ++b) {
        const ecs_table_t* t = s->cached_tables[b];
        const uint32_t count = s->cached_counts[b];
        if ((uint32_t)ecs_table_count(t) != count) return 0;
        if (memcmp(ecs_table_entities(t), s->cached_entities + row, sizeof(ecs_entity_t) * count) != 0) return 0;
        row += count;
    }
    row = 0;
    for (uint32_t b = 0; b < s->cached_batches; ++b) {
        Batch* bt = &s->batches[b];
        bt->table = s->cached_tables[b];
        bt->col_w = s->cached_cols[2 * b];
        bt->col_l = s->cached_cols[2 * b + 1];
        bt->world_col = NULL; /* fetched on first use (batch_cols): a frame in which nothing moved touches no column */
        bt->local_col = NULL;
        bt->up_row = NULL;
        bt->src = s->cached_src[b];
        bt->count = s->cached_counts[b];
        bt->walk_begin = (uint32_t)row;
        bt->slot_begin = 0;
        bt->level = -1;
        row += bt->count;
    }
    s->n_batches = s->cached_batches;
    s->n = s->cached_n;
    s->walks_skipped += 1;
    return 1;
}

/* Changed rows [i0, i1) of one staging column (k = 0 LocalTransform, 1 WorldTransform; ascending slots) back into their
 * tables: runs of consecutive slots inside one batch go in one copy.  Touches no shared state but the rows themselves
 * (column pointers that are not cached yet are looked up, not cached: several helpers may meet in one batch), so
 * ranges can run side by side. */
static void scatter_range(GpuTransformCullSystem* s, int k, const uint32_t* slots, const uint8_t* stage, uint64_t i0,
                          uint64_t i1) {
    const Batch* b = NULL;
    uint8_t* col = NULL;
    for (uint64_t i = i0; i < i1;) {
        const uint32_t slot = slots[i];
        if (!b || slot < b->slot_begin || slot >= b->slot_begin + b->count) {
            b = &s->batches[s->batch_of[slot]];
            if (k) col = (uint8_t*)(b->world_col ? b->world_col : ecs_table_get_column(b->table, b->col_w, 0));
            else col = (uint8_t*)(b->local_col ? b->local_col : ecs_table_get_column(b->table, b->col_l, 0));
        }
        const uint32_t batch_end = b->slot_begin + b->count;
        uint64_t j = i + 1;
        while (j < i1 && slots[j] == slot + (uint32_t)(j - i) && slots[j] < batch_end) ++j;
        memcpy(col + (size_t)(slot - b->slot_begin) * ROW, stage + i * ROW, (size_t)(j - i) * ROW);
        i = j;
    }
}

/* The copy back is host-memory work on up to hundreds of thousands of small tables, and it is as long as the PCIe
 * transfer itself: helpers share it and start while rows are still arriving.  The library delivers the rows in pieces
 * (yh_download_changed_rows_aos68) and says how many are in place; every helper claims the next SCATTER_GRAIN rows,
 * waits until they have landed and copies them into their tables.  Threads live for this one call: the system owns no
 * thread between frames. */
#define SCATTER_GRAIN 1024u
typedef struct ScatterStream {
    GpuTransformCullSystem* s;
    int k;
    const uint32_t* slots;
    const uint8_t* stage;
    uint64_t total;
    uint32_t helpers;        /* threads at work, the calling one included */
    _Atomic uint64_t ready;  /* rows delivered so far */
    _Atomic uint64_t cursor; /* next row nobody has claimed */
    _Atomic int stop;        /* the download failed: rows that have not landed never will */
} ScatterStream;

static void scatter_work(ScatterStream* t) {
    for (;;) {
        const uint64_t i0 = atomic_fetch_add_explicit(&t->cursor, SCATTER_GRAIN, memory_order_relaxed);
        if (i0 >= t->total) return;
        const uint64_t i1 = i0 + SCATTER_GRAIN < t->total ? i0 + SCATTER_GRAIN : t->total;
        for (uint32_t spins = 0; atomic_load_explicit(&t->ready, memory_order_acquire) < i1; ++spins) {
            if (atomic_load_explicit(&t->stop, memory_order_relaxed)) return;
            if (spins < 32) {
                sched_yield();
            } else { /* a piece takes ~100 us to cross PCIe: sixteen spinning helpers only burn the process' CPU share */
                const struct timespec nap = {0, 5000};
                nanosleep(&nap, NULL);
            }
        }
        scatter_range(t->s, t->k, t->slots, t->stage, i0, i1);
    }
}
static void* scatter_thread(void* arg) {
    scatter_work((ScatterStream*)arg);
    return NULL;
}
/* The first helper starts the others (and joins them), so the calling thread spends one pthread_create, not fifteen,
 * before the copies start. */
static void* scatter_first_helper(void* arg) {
    ScatterStream* t = (ScatterStream*)arg;
    pthread_t th[MAX_THREADS];
    int started[MAX_THREADS];
    for (uint32_t p = 2; p < t->helpers; ++p) started[p] = pthread_create(&th[p], NULL, scatter_thread, t) == 0;
    scatter_work(t);
    for (uint32_t p = 2; p < t->helpers; ++p)
        if (started[p]) pthread_join(th[p], NULL);
    return NULL;
}
static void scatter_rows_ready(void* user, uint64_t rows_ready) {
    atomic_store_explicit(&((ScatterStream*)user)->ready, rows_ready, memory_order_release);
}

/* One component's changed rows: device -> staging column -> tables.  k = 0 LocalTransform, 1 WorldTransform. */
static int fetch_changed(GpuTransformCullSystem* s, int k, uint64_t m) {
    uint8_t* stage = k ? s->stage_w : s->stage_l;
    const int which = k ? 0 : 1; /* the library's numbering: 0 WorldTransform, 1 LocalTransform */
    uint64_t written = 0;
    int rc;
    uint64_t helpers = m / ROWS_PER_THREAD_MIN;
    if (helpers > (uint64_t)s->threads) helpers = (uint64_t)s->threads;
    if (helpers < 2) {
        if ((rc = yh_download_changed_rows_aos68(s->scene, which, s->slot_buf, stage, s->n_slots, &written, NULL, NULL)) != YH_OK) return rc;
        if (written != m) return YH_ERR_STATE; /* (the counts were read a moment ago: they agree) */
        scatter_range(s, k, s->slot_buf, stage, 0, written);
        return YH_OK;
    }
    ScatterStream t;
    t.s = s;
    t.k = k;
    t.slots = s->slot_buf;
    t.stage = stage;
    t.total = m;
    atomic_init(&t.ready, 0);
    atomic_init(&t.cursor, 0);
    atomic_init(&t.stop, 0);
    t.helpers = (uint32_t)helpers;
    pthread_t first;
    const int started = pthread_create(&first, NULL, scatter_first_helper, &t) == 0;
    rc = yh_download_changed_rows_aos68(s->scene, which, s->slot_buf, stage, s->n_slots, &written, scatter_rows_ready, &t);
    if (rc != YH_OK || written != m) atomic_store(&t.stop, 1);
    else scatter_work(&t); /* every row is in place: this thread joins in, and finishes alone if no helper started */
    if (started) pthread_join(first, NULL);
    return rc != YH_OK ? rc : (written == m ? YH_OK : YH_ERR_STATE);
}

/* Frames ran without readback and now the slot order is about to be rebuilt from the flecs columns (a structural
 * change): the device holds the newer matrices, and they are addressed by the OLD slot order, whose tables may no
 * longer exist.  Bring every row home first, entity by entity (one component lookup each: a structural change is
 * already a slow frame) — except rows whose onSet fired since the last frame: there flecs holds the newer value. */
static int catch_up_by_entity(GpuTransformCullSystem* s) {
    const uint64_t n = s->scene_entities;
    int rc;
    if ((rc = yh_download_world_aos68(s->scene, s->stage_w, 0, n)) != YH_OK ||
        (rc = yh_download_local_aos68(s->scene, s->stage_l, 0, n)) != YH_OK)
        return rc;
    uint8_t* keep = (uint8_t*)calloc((size_t)(n ? n : 1), 1); /* bit 0: local edited on the host, bit 1: world */
    if (!keep) return YH_ERR_OOM;
    for (int k = 0; k < 2; ++k)
        for (uint64_t i = 0; i < s->n_dirty[k]; ++i) {
            uint32_t slot;
            if (slot_lookup(s, s->dirty[k][i], &slot) && slot < n) keep[slot] |= (uint8_t)(1 << k);
        }
    for (uint64_t slot = 0; slot < n; ++slot) {
        const ecs_entity_t e = s->entity_of_slot[slot];
        if (!e || !ecs_is_alive(s->world, e)) continue;
        if (!(keep[slot] & 2)) {
            void* w = (void*)ecs_get_id(s->world, e, s->world_id);
            if (w) memcpy(w, s->stage_w + slot * ROW, ROW);
        }
        if (!(keep[slot] & 1)) {
            void* l = (void*)ecs_get_id(s->world, e, s->local_id);
            if (l) memcpy(l, s->stage_l + slot * ROW, ROW);
        }
    }
    free(keep);
    s->host_stale = 0;
    return YH_OK;
}

/* The rows whose onSet fired, from their tables into the staging column (duplicates dropped, slots into slot_buf).
 * Every row costs a chain of cache misses (entity -> slot map, stamp, slot -> batch, the batch, the row itself, all in
 * tables far larger than the caches), so the chain is walked for GATHER_BLOCK rows at a time, each link prefetched for
 * the whole block before the next link is followed: the misses of a block overlap instead of queueing up. */
#define GATHER_BLOCK 64u
typedef struct GatherJob {
    GpuTransformCullSystem* s;
    int k, shared;    /* shared: other ranges run beside this one (no caching of column pointers, atomic stamps) */
    uint8_t* stage;
    uint64_t i0, i1;  /* range of the dirty list; its rows land at stage row i0 onwards, their slots at slot_buf[i0 ..] */
    uint64_t m;       /* rows found */
} GatherJob;
static void* gather_range(void* arg) {
    GatherJob* g = (GatherJob*)arg;
    GpuTransformCullSystem* s = g->s;
    const int k = g->k;
    const ecs_entity_t* ents = s->dirty[k];
    uint32_t* stamp = s->stamp[k];
    uint8_t* stage = g->stage + g->i0 * ROW;
    uint32_t* out_slots = s->slot_buf + g->i0;
    uint64_t m = 0;
    uint32_t slot[GATHER_BLOCK];
    const uint8_t* src[GATHER_BLOCK];
    for (uint64_t i0 = g->i0; i0 < g->i1; i0 += GATHER_BLOCK) {
        const uint32_t nb = (uint32_t)(g->i1 - i0 < GATHER_BLOCK ? g->i1 - i0 : GATHER_BLOCK);
        for (uint32_t j = 0; j < nb; ++j) map_prefetch(&s->slot_of, ents[i0 + j], 0);
        for (uint32_t j = 0; j < nb; ++j) {
            if (!slot_lookup(s, ents[i0 + j], &slot[j])) {
                slot[j] = UINT32_MAX; /* not a row of this system (the root, or an entity deleted since) */
                continue;
            }
            __builtin_prefetch(&stamp[slot[j]]);
            if (s->batch_of) __builtin_prefetch(&s->batch_of[slot[j]]);
        }
        for (uint32_t j = 0; j < nb; ++j) {
            if (slot[j] == UINT32_MAX) continue;
            /* set more than once since the last frame (another range may hold the other set: one of the two wins) */
            if (__atomic_exchange_n(&stamp[slot[j]], s->epoch, __ATOMIC_RELAXED) == s->epoch) {
                slot[j] = UINT32_MAX;
                continue;
            }
            if (s->batch_of) __builtin_prefetch(&s->batches[s->batch_of[slot[j]]]);
        }
        for (uint32_t j = 0; j < nb; ++j) {
            if (slot[j] == UINT32_MAX) continue;
            const uint8_t* col;
            const Batch* b;
            if (g->shared && s->batch_of) { /* (column pointers that are not cached yet are looked up, not cached) */
                b = &s->batches[s->batch_of[slot[j]]];
                if (k == 0) col = (const uint8_t*)(b->local_col ? b->local_col : ecs_table_get_column(b->table, b->col_l, 0));
                else col = (const uint8_t*)(b->world_col ? b->world_col : ecs_table_get_column(b->table, b->col_w, 0));
            } else {
                b = batch_of_slot(s, slot[j]);
                col = (const uint8_t*)(k == 0 ? b->local_col : b->world_col);
            }
            src[j] = col + (size_t)(slot[j] - b->slot_begin) * ROW;
            __builtin_prefetch(src[j]);
            __builtin_prefetch(src[j] + 64);
        }
        for (uint32_t j = 0; j < nb; ++j) {
            if (slot[j] == UINT32_MAX) continue;
            memcpy(stage + m * ROW, src[j], ROW);
            out_slots[m++] = slot[j];
        }
    }
    g->m = m;
    return NULL;
}
/* Long lists are shared by helper threads, each with a range of its own (its rows land at the range's own offset and
 * are moved together afterwards): 10k rows take 1.5 ms on one thread, all of it cache misses, 0.6 ms on four (twice as
 * many threads gain nothing more: starting one costs what 400 rows do). */
static uint64_t gather_dirty(GpuTransformCullSystem* s, int k, uint8_t* stage) {
    const uint64_t nd = s->n_dirty[k];
    uint32_t parts = nd / ROWS_PER_THREAD_MIN > (uint64_t)s->threads ? (uint32_t)s->threads : (uint32_t)(nd / ROWS_PER_THREAD_MIN);
    if (parts < 2 || !s->batch_of || nd > s->n_slots) { /* (a list longer than the staging column holds duplicates: one range compacts as it goes) */
        GatherJob all = {s, k, 0, stage, 0, nd, 0};
        gather_range(&all);
        return all.m;
    }
    pthread_t th[MAX_THREADS];
    GatherJob job[MAX_THREADS];
    int started[MAX_THREADS];
    for (uint32_t p = 0; p < parts; ++p) {
        const GatherJob j = {s, k, 1, stage, nd * p / parts, nd * (p + 1) / parts, 0};
        job[p] = j;
        started[p] = p > 0 && pthread_create(&th[p], NULL, gather_range, &job[p]) == 0;
    }
    for (uint32_t p = 0; p < parts; ++p)
        if (!started[p]) gather_range(&job[p]); /* part 0, and any part whose thread did not start */
    for (uint32_t p = 1; p < parts; ++p)
        if (started[p]) pthread_join(th[p], NULL);
    uint64_t m = job[0].m;
    for (uint32_t p = 1; p < parts; ++p) {
        if (job[p].m && m != job[p].i0) {
            memmove(stage + m * ROW, stage + job[p].i0 * ROW, job[p].m * ROW);
            memmove(s->slot_buf + m, s->slot_buf + job[p].i0, job[p].m * sizeof(uint32_t));
        }
        m += job[p].m;
    }
    return m;
}

/* ---- whole columns between the tables and the staging columns, shared by helper threads ------------------------ */
typedef struct ColumnsJob {
    GpuTransformCullSystem* s;
    uint32_t b0, b1; /* batches */
    int to_stage;    /* 1: tables -> staging columns, 0: back */
} ColumnsJob;
static void* columns_thread(void* arg) {
    const ColumnsJob* j = (const ColumnsJob*)arg;
    GpuTransformCullSystem* s = j->s;
    for (uint32_t i = j->b0; i < j->b1; ++i) {
        if (!s->batches[i].count) continue; /* (a batch whose rows moved away: patch()) */
        const Batch* b = batch_cols(&s->batches[i]); /* each batch is one thread's: its column pointers are filled in here */
        uint8_t *sw = s->stage_w + (size_t)b->slot_begin * ROW, *sl = s->stage_l + (size_t)b->slot_begin * ROW;
        const size_t bytes = (size_t)b->count * ROW;
        if (j->to_stage) {
            memcpy(sw, b->world_col, bytes);
            memcpy(sl, b->local_col, bytes);
        } else {
            memcpy(b->world_col, sw, bytes);
            memcpy(b->local_col, sl, bytes);
        }
    }
    return NULL;
}
/* 136 MB in 300k pieces at 1M entities: 57-78 ms on one thread, most of it cache misses */
static void copy_columns(GpuTransformCullSystem* s, int to_stage) {
    uint32_t parts = s->n / ROWS_PER_THREAD_MIN > (uint64_t)s->threads ? (uint32_t)s->threads : (uint32_t)(s->n / ROWS_PER_THREAD_MIN);
    if (parts > s->n_batches) parts = s->n_batches;
    if (parts < 2) {
        ColumnsJob all = {s, 0, s->n_batches, to_stage};
        columns_thread(&all);
        return;
    }
    pthread_t th[MAX_THREADS];
    ColumnsJob job[MAX_THREADS];
    int started[MAX_THREADS];
    for (uint32_t p = 0; p < parts; ++p) {
        job[p].s = s;
        job[p].b0 = (uint32_t)((uint64_t)s->n_batches * p / parts);
        job[p].b1 = (uint32_t)((uint64_t)s->n_batches * (p + 1) / parts);
        job[p].to_stage = to_stage;
        started[p] = p > 0 && pthread_create(&th[p], NULL, columns_thread, &job[p]) == 0;
    }
    for (uint32_t p = 0; p < parts; ++p)
        if (!started[p]) columns_thread(&job[p]); /* part 0, and any part whose thread did not start */
    for (uint32_t p = 1; p < parts; ++p)
        if (started[p]) pthread_join(th[p], NULL);
}

/* ---- small structural changes without a rebuild ----------------------------------------------------------- */
typedef struct PatchItem {
    ecs_table_t* table;
    uint32_t old_batch; /* UINT32_MAX: the table was not cached */
    uint32_t count;     /* rows it holds now (0: it left the query) */
    int32_t col_w, col_l, level;
    ecs_entity_t src;
    uint32_t new_batch;
    uint32_t first; /* the slots its rows take */
} PatchItem;
static int by_ptr(const void* a, const void* b) {
    const uintptr_t x = (uintptr_t) * (ecs_table_t* const*)a, y = (uintptr_t) * (ecs_table_t* const*)b;
    return x < y ? -1 : (x > y);
}
static int by_level(const void* a, const void* b) {
    const PatchItem* x = (const PatchItem*)a;
    const PatchItem* y = (const PatchItem*)b;
    return x->level < y->level ? -1 : (x->level > y->level);
}
static int32_t level_of_slot(const GpuTransformCullSystem* s, uint32_t slot) {
    int32_t l = 0;
    while ((uint32_t)l + 1 < s->n_levels && s->level_begin[l + 1] <= slot) ++l;
    return l;
}
static int upload_parent_run(GpuTransformCullSystem* s, uint32_t first, uint32_t count, uint32_t parent_slot) {
    for (uint32_t r = 0; r < count; ++r) s->slot_buf[r] = parent_slot; /* (slot_buf: n_slots words, free between its uses) */
    return yh_upload_parent_range(s->scene, s->slot_buf, first, count);
}

/* The tables the observers listed are compared with their batches; the rows of one that changed (entities created in
 * it, deleted from it, moved in or out by a re-parenting or by an unrelated component) move to spare slots of its
 * level, its old slots die, the children of every row that moved learn their parent's new slot.  Nothing else is
 * touched: no query walk, no re-sort, no column upload beyond the moved rows; the cost is two passes over the batch
 * table (which table is whose batch, which batches are children of a moved row) plus the moved rows.
 * Returns YH_OK (the batches follow the world again), 1 (not a case for a patch: walk and rebuild), or an error. */
static int patch(GpuTransformCullSystem* s) {
    int rc = YH_OK;
    uint32_t nt = 0;
    if (s->n_touched) {
        qsort(s->touched, s->n_touched, sizeof(ecs_table_t*), by_ptr);
        for (uint32_t i = 0; i < s->n_touched; ++i)
            if (!nt || s->touched[nt - 1] != s->touched[i]) s->touched[nt++] = s->touched[i];
    }
    uint32_t nd = 0; /* deleted tables, sorted, unique */
    if (s->n_deleted) {
        qsort(s->deleted, s->n_deleted, sizeof(ecs_table_t*), by_ptr);
        for (uint32_t i = 0; i < s->n_deleted; ++i)
            if (!nd || s->deleted[nd - 1] != s->deleted[i]) s->deleted[nd++] = s->deleted[i];
    }
    s->last_patch_tables = nt + nd;
    if (!nt && !nd) return YH_OK;
    PatchItem* items = (PatchItem*)calloc((size_t)nt + 1, sizeof(PatchItem));
    uint32_t cap_items = nt + 1;
    uint32_t* old_of = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)nt + 1)); /* per listed table: its batch, or UINT32_MAX */
    uint32_t* kids = NULL;  /* batches whose parent row moves inside its level */
    uint8_t* reloc = NULL;
    int32_t* new_level = NULL;
    SlotRun* plan[YH_MAX_LEVELS]; /* the levels' free runs as they will be after this patch (NULL: level not involved) */
    uint32_t plan_n[YH_MAX_LEVELS], plan_fill[YH_MAX_LEVELS], plan_extra[YH_MAX_LEVELS];
    memset(plan, 0, sizeof plan);
    memset(plan_extra, 0, sizeof plan_extra);
    uint32_t n_kids = 0, cap_kids = 0;
    Map moved = {0}; /* entity -> 1 + index of the item whose new content holds it, 0: it leaves */
    uint64_t rows_in = 0, rows_out = 0;
    uint32_t ni = 0;
    /* Frames ran without readback: the device holds the newer rows.  A row that was on the device changes slots THERE
     * (yh_move_rows); only rows of entities the scene has never held come from the flecs columns. */
    const int dev_newer = s->host_stale;
    uint32_t *mv_src = NULL, *mv_dst = NULL, *fresh_slots = NULL;
    uint8_t *fresh_w = NULL, *fresh_l = NULL;
    uint64_t n_mv = 0, n_fresh = 0;
    if (!items || !old_of) {
        rc = YH_ERR_OOM;
        goto done;
    }
    /* which listed table is which batch: one pass over the batches, a binary search each */
    memset(old_of, 0xFF, sizeof(uint32_t) * nt);
    for (uint32_t b = 0; b < s->n_batches; ++b) {
        const Batch* bt = &s->batches[b];
        if (!bt->count || !bt->table) continue;
        if (nd) { /* a batch of a deleted table dies (compared, never dereferenced); the pointer may be listed as well,
                   * then as ANOTHER table that was given the same address: a new table to this system */
            uint32_t dl = 0, dh = nd;
            while (dl < dh) {
                const uint32_t mid = (dl + dh) / 2;
                if ((uintptr_t)s->deleted[mid] < (uintptr_t)bt->table) dl = mid + 1;
                else dh = mid;
            }
            if (dl < nd && s->deleted[dl] == bt->table) {
                if (ni == cap_items) {
                    PatchItem* g = (PatchItem*)realloc(items, sizeof(PatchItem) * (size_t)cap_items * 2);
                    if (!g) {
                        rc = YH_ERR_OOM;
                        goto done;
                    }
                    items = g;
                    cap_items *= 2;
                }
                PatchItem gone;
                memset(&gone, 0, sizeof gone);
                gone.old_batch = b;
                gone.new_batch = UINT32_MAX;
                gone.src = bt->src;
                gone.level = bt->level;
                items[ni++] = gone;
                continue;
            }
        }
        uint32_t lo = 0, hi = nt;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) / 2;
            if ((uintptr_t)s->touched[mid] < (uintptr_t)bt->table) lo = mid + 1;
            else hi = mid;
        }
        if (lo < nt && s->touched[lo] == bt->table) old_of[lo] = b;
    }
    /* what every listed table holds now, and where its rows belong */
    for (uint32_t i = 0; i < nt; ++i) {
        ecs_table_t* t = s->touched[i];
        PatchItem it;
        memset(&it, 0, sizeof it);
        it.table = t;
        it.old_batch = old_of[i];
        it.new_batch = UINT32_MAX;
        it.col_w = ecs_table_get_column_index(s->world, t, s->world_id);
        it.col_l = ecs_table_get_column_index(s->world, t, s->local_id);
        it.count = it.col_w >= 0 && it.col_l >= 0 ? (uint32_t)ecs_table_count(t) : 0u;
        /* Disabled entities and prefabs are not matched by the system's query (flecs leaves both out unless a term asks
         * for them; the engine exposes world.enable(e, false): ecs.zig:546), so the walk never sees their tables: a
         * patch must not either - the rows of such a table leave, as they do on the rebuild path. */
        if (it.count && (ecs_table_has_id(s->world, t, EcsDisabled) || ecs_table_has_id(s->world, t, EcsPrefab))) it.count = 0;
        if (it.old_batch == UINT32_MAX && !it.count) continue;
        const ecs_entity_t* ents = it.count ? ecs_table_entities(t) : NULL;
        if (it.old_batch != UINT32_MAX) {
            const Batch* b = &s->batches[it.old_batch];
            it.src = b->src;
            it.level = b->level;
            if (it.count == b->count && memcmp(ents, s->entity_of_slot + b->slot_begin, sizeof(ecs_entity_t) * it.count) == 0) {
                /* Same rows in the same order (a component came and went, or rows were added and removed again): the
                 * batch stays, but its columns may have been reallocated meanwhile - the cached pointers are dropped
                 * and fetched afresh on first use (batch_cols). */
                s->batches[it.old_batch].world_col = s->batches[it.old_batch].local_col = NULL;
                s->batches[it.old_batch].up_row = NULL;
                continue;
            }
        } else {
            it.src = ecs_get_parent(s->world, ents[0]); /* ChildOf is part of the archetype: one parent per table */
            if (!it.src) continue;                      /* no parent: the Up term does not match this table */
            if (!ecs_has_id(s->world, it.src, s->world_id)) goto not_a_patch; /* Up resolves further up, or not at all */
            it.level = -1; /* one below its parent's, found below (the parent may be a row of this very patch) */
        }
        if (ni == cap_items) {
            PatchItem* g = (PatchItem*)realloc(items, sizeof(PatchItem) * (size_t)cap_items * 2);
            if (!g) {
                rc = YH_ERR_OOM;
                goto done;
            }
            items = g;
            cap_items *= 2;
        }
        items[ni++] = it;
    }
    if (!ni) goto done;
    /* the rows that move or leave */
    for (uint32_t i = 0; i < ni; ++i) {
        rows_in += items[i].count;
        if (items[i].old_batch != UINT32_MAX) rows_out += s->batches[items[i].old_batch].count;
    }
    if (map_init(&moved, rows_in + rows_out + 8) != 0) {
        rc = YH_ERR_OOM;
        goto done;
    }
    for (uint32_t i = 0; i < ni; ++i) {
        if (items[i].old_batch == UINT32_MAX) continue;
        const Batch* b = &s->batches[items[i].old_batch];
        for (uint32_t r = 0; r < b->count; ++r) map_put(&moved, s->entity_of_slot[b->slot_begin + r], 0);
    }
    for (uint32_t i = 0; i < ni; ++i) {
        const ecs_entity_t* ents = items[i].count ? ecs_table_entities(items[i].table) : NULL;
        for (uint32_t r = 0; r < items[i].count; ++r) map_put(&moved, ents[r], 1 + i);
    }
    /* Levels.  A batch lives one level below its parent row.  A row that changed its level (a subtree re-parented to
     * another depth) takes every batch below it along: those batches were not listed - their tables did not change -
     * but their rows move to the spare slots of another level all the same.  reloc: 1 = listed batch (dies, its table's
     * rows are placed anew), 2 = follows its parent to new_level.  Repeated until nothing changes (a subtree's height). */
    reloc = (uint8_t*)calloc(s->n_batches ? s->n_batches : 1, 1);
    new_level = (int32_t*)malloc(sizeof(int32_t) * (s->n_batches ? s->n_batches : 1));
    if (!reloc || !new_level) {
        rc = YH_ERR_OOM;
        goto done;
    }
    for (uint32_t i = 0; i < ni; ++i)
        if (items[i].old_batch != UINT32_MAX) reloc[items[i].old_batch] = 1;
    for (uint32_t iter = 0;; ++iter) {
        if (iter > YH_MAX_LEVELS + 2u) goto not_a_patch;
        int changed = 0;
        for (uint32_t i = 0; i < ni; ++i) { /* the listed tables: one level below their parent, wherever that is now */
            PatchItem* it = &items[i];
            int32_t lvl;
            uint32_t where, ps;
            if (!it->count) continue; /* (a table that left or was deleted: its batch dies, nothing is placed) */
            if (it->src == s->root) lvl = 0;
            else if (map_get(&moved, it->src, &where)) {
                if (!where) goto not_a_patch; /* the parent left */
                lvl = items[where - 1].level < 0 ? -1 : items[where - 1].level + 1;
            } else if (slot_lookup(s, it->src, &ps)) {
                const uint32_t pb = s->batch_of[ps];
                lvl = (pb != UINT32_MAX && reloc[pb] == 2 ? new_level[pb] : level_of_slot(s, ps)) + 1;
            } else goto not_a_patch; /* a parent this system does not hold (a second root) */
            if (lvl != it->level) {
                it->level = lvl;
                changed = 1;
            }
        }
        n_kids = 0;
        for (uint32_t c = 0; c < s->n_batches; ++c) { /* every other batch: does its parent row move? */
            const Batch* bt = &s->batches[c];
            if (!bt->count || reloc[c] == 1 || bt->level == 0) continue;
            const uint32_t pb = bt->parent_slot < s->n_slots ? s->batch_of[bt->parent_slot] : UINT32_MAX;
            if (pb == UINT32_MAX) goto not_a_patch; /* (a live batch under a dead slot: not a state patch() leaves) */
            if (!reloc[pb]) continue;
            int32_t want;
            if (reloc[pb] == 1) {
                uint32_t where;
                if (!map_get(&moved, bt->src, &where) || !where) goto not_a_patch; /* its parent leaves while it stays: not a leaf operation */
                if (items[where - 1].level < 0) { /* (not known yet: next round) */
                    changed = 1;
                    continue;
                }
                want = items[where - 1].level + 1;
            } else {
                want = new_level[pb] + 1;
            }
            if (want == bt->level && reloc[c] == 0) { /* the parent row moves inside its level: a new parent slot is all */
                if (n_kids == cap_kids) {
                    const uint32_t cap = cap_kids ? cap_kids * 2 : 64;
                    uint32_t* g = (uint32_t*)realloc(kids, sizeof(uint32_t) * cap);
                    if (!g) {
                        rc = YH_ERR_OOM;
                        goto done;
                    }
                    kids = g;
                    cap_kids = cap;
                }
                kids[n_kids++] = c;
            } else if (reloc[c] != 2 || new_level[c] != want) {
                reloc[c] = 2;
                new_level[c] = want;
                changed = 1;
            }
        }
        if (!changed) break;
    }
    {   /* the batches that follow their parents become items of their own: same table, same rows, another level */
        uint32_t n_follow = 0;
        for (uint32_t c = 0; c < s->n_batches; ++c) n_follow += reloc[c] == 2;
        if (n_follow) {
            PatchItem* g = (PatchItem*)realloc(items, sizeof(PatchItem) * ((size_t)ni + n_follow));
            if (!g) {
                rc = YH_ERR_OOM;
                goto done;
            }
            items = g;
            cap_items = ni + n_follow;
            for (uint32_t c = 0; c < s->n_batches; ++c) {
                if (reloc[c] != 2) continue;
                const Batch* bt = &s->batches[c];
                PatchItem it;
                memset(&it, 0, sizeof it);
                it.table = bt->table;
                it.old_batch = c;
                it.new_batch = UINT32_MAX;
                it.col_w = bt->col_w;
                it.col_l = bt->col_l;
                it.count = bt->count;
                it.src = bt->src;
                it.level = new_level[c];
                if ((uint32_t)ecs_table_count(bt->table) != bt->count) goto not_a_patch; /* (its table changed unseen?) */
                items[ni++] = it;
                rows_in += bt->count;
                rows_out += bt->count;
            }
        }
    }
    for (uint32_t i = 0; i < ni; ++i) {
        if (!items[i].count) continue;
        if (items[i].level < 0 || (uint32_t)items[i].level >= s->n_levels) goto not_a_patch; /* deeper than the scene's levels */
    }
    /* Where the rows go: into a run of slots that died earlier (or dies in this very patch) if one is long enough -
     * best fit -, else into the untouched tail of the level's range.  Planned on copies of the levels' lists: nothing
     * has changed yet if a level turns out to be full. */
    qsort(items, ni, sizeof(PatchItem), by_level); /* (parents before children; equal levels in any order) */
    for (uint32_t i = 0; i < ni; ++i) {
        if (items[i].old_batch != UINT32_MAX) plan_extra[s->batches[items[i].old_batch].level] += 1;
        if (items[i].count) plan_extra[items[i].level] += 1; /* (a run that is split stays one run: no growth; counted for the level's sake) */
    }
    for (uint32_t l = 0; l < s->n_levels; ++l) {
        if (!plan_extra[l]) continue;
        plan[l] = (SlotRun*)malloc(sizeof(SlotRun) * ((size_t)s->n_free[l] + plan_extra[l] + 1));
        if (!plan[l]) {
            rc = YH_ERR_OOM;
            goto done;
        }
        if (s->n_free[l]) memcpy(plan[l], s->free_runs[l], sizeof(SlotRun) * s->n_free[l]);
        plan_n[l] = s->n_free[l];
        plan_fill[l] = s->level_fill[l];
    }
    for (uint32_t i = 0; i < ni; ++i) {
        if (items[i].old_batch == UINT32_MAX) continue;
        const Batch* b = &s->batches[items[i].old_batch];
        plan[b->level][plan_n[b->level]].first = b->slot_begin;
        plan[b->level][plan_n[b->level]++].count = b->count;
    }
    for (uint32_t i = 0; i < ni; ++i) {
        PatchItem* it = &items[i];
        if (!it->count) continue;
        const uint32_t l = (uint32_t)it->level;
        uint32_t r = plan_n[l]; /* best fit: the shortest run that is long enough (first fit splinters the long runs) */
        for (uint32_t q = 0; q < plan_n[l]; ++q)
            if (plan[l][q].count >= it->count && (r == plan_n[l] || plan[l][q].count < plan[l][r].count)) r = q;
        if (r < plan_n[l]) {
            it->first = plan[l][r].first;
            plan[l][r].first += it->count;
            plan[l][r].count -= it->count;
            if (!plan[l][r].count) plan[l][r] = plan[l][--plan_n[l]];
        } else if ((uint64_t)plan_fill[l] + it->count <= s->level_begin[l + 1]) {
            it->first = plan_fill[l];
            plan_fill[l] += it->count;
        } else {
            goto not_a_patch; /* the level's spare slots are used up: the rebuild compacts */
        }
    }
    if ((s->slot_entries + rows_in) * 10 > s->slot_of.cap * 7) goto not_a_patch;          /* the entity -> slot map is filling up */
    if (rows_in > s->n_slots / 4) goto not_a_patch;                                       /* most of a level moves: a rebuild is cheaper */
    /* room for the new batches */
    if (s->n_batches + ni > s->cap_batches) {
        uint32_t cap = s->cap_batches ? s->cap_batches : 256;
        while (cap < s->n_batches + ni) cap *= 2;
        Batch* nb = (Batch*)realloc(s->batches, cap * sizeof(Batch));
        if (!nb) {
            rc = YH_ERR_OOM;
            goto done;
        }
        s->batches = nb;
        s->cap_batches = cap;
    }

    if (dev_newer) { /* old slot -> new slot of every row the device holds; the others are copied out of their tables */
        mv_src = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(rows_in + 1));
        mv_dst = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(rows_in + 1));
        fresh_slots = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(rows_in + 1));
        fresh_w = (uint8_t*)malloc((size_t)(rows_in + 1) * ROW);
        fresh_l = (uint8_t*)malloc((size_t)(rows_in + 1) * ROW);
        if (!mv_src || !mv_dst || !fresh_slots || !fresh_w || !fresh_l) {
            rc = YH_ERR_OOM;
            goto done;
        }
        for (uint32_t i = 0; i < ni; ++i) {
            const PatchItem* it = &items[i];
            if (!it->count) continue;
            const ecs_entity_t* ents = ecs_table_entities(it->table);
            const uint8_t* wc = (const uint8_t*)ecs_table_get_column(it->table, it->col_w, 0);
            const uint8_t* lc = (const uint8_t*)ecs_table_get_column(it->table, it->col_l, 0);
            for (uint32_t r = 0; r < it->count; ++r) {
                uint32_t os;
                if (slot_lookup(s, ents[r], &os)) {
                    mv_src[n_mv] = os;
                    mv_dst[n_mv++] = it->first + r;
                } else {
                    fresh_slots[n_fresh] = it->first + r;
                    memcpy(fresh_w + n_fresh * ROW, wc + (size_t)r * ROW, ROW);
                    memcpy(fresh_l + n_fresh * ROW, lc + (size_t)r * ROW, ROW);
                    ++n_fresh;
                }
            }
        }
    }
    /* from here on the state changes: an error leaves it to the next frame's rebuild (the caller sees to that) */
    for (uint32_t i = 0; i < ni; ++i) { /* the old slots die */
        if (items[i].old_batch == UINT32_MAX) continue;
        Batch* b = &s->batches[items[i].old_batch];
        if ((rc = yh_scene_set_alive_range(s->scene, b->slot_begin, b->count, 0)) != YH_OK) goto done;
        memset(s->entity_of_slot + b->slot_begin, 0, sizeof(ecs_entity_t) * b->count);
        memset(s->batch_of + b->slot_begin, 0xFF, sizeof(uint32_t) * b->count);
        s->n -= b->count;
        b->count = 0;
        b->table = NULL;
        b->world_col = b->local_col = NULL;
    }
    for (uint32_t l = 0; l < s->n_levels; ++l) { /* the levels' lists of dead runs and tails as planned */
        if (!plan[l]) continue;
        SlotRun* old_runs = s->free_runs[l];
        s->free_runs[l] = plan[l];
        s->n_free[l] = plan_n[l];
        s->level_fill[l] = plan_fill[l];
        plan[l] = old_runs; /* (freed with the plan) */
    }
    if (n_mv && (rc = yh_move_rows(s->scene, mv_src, mv_dst, n_mv)) != YH_OK) goto done; /* (sources read first: a new slot may be another row's old one) */
    for (uint32_t i = 0; i < ni; ++i) { /* the rows take their slots */
        PatchItem* it = &items[i];
        if (!it->count) continue;
        const uint32_t bi = s->n_batches++, first = it->first;
        it->new_batch = bi;
        Batch* b = &s->batches[bi];
        memset(b, 0, sizeof *b);
        b->table = it->table;
        b->col_w = it->col_w;
        b->col_l = it->col_l;
        b->src = it->src;
        b->count = it->count;
        b->slot_begin = first;
        b->level = it->level;
        b->world_col = ecs_table_get_column(it->table, it->col_w, 0);
        b->local_col = ecs_table_get_column(it->table, it->col_l, 0);
        const ecs_entity_t* ents = ecs_table_entities(it->table);
        memcpy(s->entity_of_slot + first, ents, sizeof(ecs_entity_t) * it->count);
        for (uint32_t r = 0; r < it->count; ++r) {
            s->slot_entries += (uint64_t)map_put(&s->slot_of, ents[r], first + r); /* (a row that only moved keeps its entry) */
            s->batch_of[first + r] = bi;
        }
        s->n += it->count;
        /* rows as flecs holds them, flags included (a new entity's LocalTransform is dirty: transform.zig:46-51; a
         * re-parented one keeps its flags: ecs.zig:470-477 sets none) */
        if ((rc = yh_scene_set_alive_range(s->scene, first, it->count, 1)) != YH_OK) goto done;
        if (!dev_newer) {
            if ((rc = yh_upload_world_aos68(s->scene, b->world_col, first, it->count)) != YH_OK ||
                (rc = yh_upload_local_aos68(s->scene, b->local_col, first, it->count)) != YH_OK)
                goto done;
            s->last_up_rows += 2u * it->count;
        }
        if (s->scene_bounds_kind == YH_BOUNDS_MESH_TABLE) {
            if (s->bounds_stale || !s->mesh_index.e) continue; /* every id is about to be read afresh anyway */
            /* mesh ids of the moved rows; a mesh seen for the first time extends the table (which goes up first) */
            const uint32_t had = s->mesh_count;
            if ((rc = mesh_ids_of_batch(s, b, s->slot_buf)) != YH_OK) {
                if (rc != YH_ERR_CAPACITY) goto done;
                s->bounds_stale = 1; /* the table is full of entries nobody uses any more: the full upload compacts it */
                rc = YH_OK;
                continue;
            }
            if (s->mesh_count != had && (rc = mesh_table_upload(s)) != YH_OK) goto done;
            if ((rc = yh_upload_mesh_ids(s->scene, s->slot_buf, first, it->count)) != YH_OK) goto done;
        } else {
            float* planes = (float*)calloc((size_t)it->count * 6, sizeof(float));
            if (!planes) {
                rc = YH_ERR_OOM;
                goto done;
            }
            for (uint32_t r = 0; r < it->count && s->bounds_n; ++r) {
                const BoundsEntry* be = bounds_find(s, ents[r]);
                if (be)
                    for (int k = 0; k < 6; ++k) planes[(size_t)k * it->count + r] = be->mm[k];
            }
            const float* mn[3] = {planes, planes + it->count, planes + 2 * (size_t)it->count};
            const float* mx[3] = {planes + 3 * (size_t)it->count, planes + 4 * (size_t)it->count, planes + 5 * (size_t)it->count};
            rc = yh_upload_aabb(s->scene, mn, mx, first, it->count);
            free(planes);
            if (rc != YH_OK) goto done;
        }
    }
    if (n_fresh) { /* entities the scene sees for the first time: their rows as flecs holds them */
        if ((rc = yh_upload_world_rows_aos68(s->scene, fresh_slots, fresh_w, n_fresh)) != YH_OK ||
            (rc = yh_upload_local_rows_aos68(s->scene, fresh_slots, fresh_l, n_fresh)) != YH_OK)
            goto done;
        s->last_up_rows += 2u * n_fresh;
    }
    for (uint32_t i = 0; i < ni; ++i) { /* parent slots of the moved rows (their parent may have moved in this patch too) */
        const PatchItem* it = &items[i];
        if (!it->count) continue;
        Batch* b = &s->batches[it->new_batch];
        uint32_t ps = YH_NO_PARENT;
        if (b->level > 0 && !slot_lookup(s, b->src, &ps)) {
            rc = YH_ERR_STATE;
            goto done;
        }
        b->parent_slot = ps;
        if ((rc = upload_parent_run(s, b->slot_begin, b->count, ps)) != YH_OK) goto done;
    }
    for (uint32_t k = 0; k < n_kids; ++k) { /* and of the batches whose parent row moved inside its level */
        Batch* c = &s->batches[kids[k]];
        uint32_t ps;
        if (!c->count) continue;
        if (!slot_lookup(s, c->src, &ps)) {
            rc = YH_ERR_STATE;
            goto done;
        }
        c->parent_slot = ps;
        if ((rc = upload_parent_run(s, c->slot_begin, c->count, ps)) != YH_OK) goto done;
    }
    s->patches += 1;
    s->walk_cache_valid = 0; /* cached_* no longer describe the batches: whatever walks the query next rebuilds */
done:
    free(items);
    free(old_of);
    free(kids);
    free(reloc);
    free(new_level);
    map_free(&moved);
    free(mv_src);
    free(mv_dst);
    free(fresh_slots);
    free(fresh_w);
    free(fresh_l);
    for (uint32_t l = 0; l < YH_MAX_LEVELS; ++l) free(plan[l]);
    return rc;
not_a_patch:
    free(items);
    free(old_of);
    free(kids);
    free(reloc);
    free(new_level);
    map_free(&moved);
    free(mv_src);
    free(mv_dst);
    free(fresh_slots);
    free(fresh_w);
    free(fresh_l);
    for (uint32_t l = 0; l < YH_MAX_LEVELS; ++l) free(plan[l]);
    return 1;
}

/* A rebuild without a walk of the query.  When the structure changed in a way patch() does not follow (a level's spare
 * slots used up, too many rows moving, patching switched off) the batches still describe every table as of the last
 * frame and the observers have listed every table that rows entered or left since - so the new batch list is the old
 * one minus the batches of deleted or emptied tables, with the listed tables read afresh and the new ones appended;
 * flecs' own iteration of 300k cached tables (~100 ms per million entities) is not needed to find that out.  rebuild()
 * then sorts them into slots as after a walk (tables of one level that share a parent slot keep the order of this
 * list instead of the query's: the order of siblings' tables is not defined by the reference either).
 * Returns YH_OK (batches, walk_entities, n describe the world: rebuild), or 1: walk the query instead - a listed table
 * holds a WorldTransform without a LocalTransform (a holder of a parent matrix came or went: the Up term may resolve differently),
 * a new table's parent has no WorldTransform, a kept table's parent left. */
static int refresh_batches(GpuTransformCullSystem* s) {
    if (!s->walk_is_whole_tables || !s->entity_of_slot || !s->n_batches) return 1;
    uint32_t nt = 0, nd = 0;
    if (s->n_touched) {
        qsort(s->touched, s->n_touched, sizeof(ecs_table_t*), by_ptr);
        for (uint32_t i = 0; i < s->n_touched; ++i)
            if (!nt || s->touched[nt - 1] != s->touched[i]) s->touched[nt++] = s->touched[i];
    }
    if (s->n_deleted) {
        qsort(s->deleted, s->n_deleted, sizeof(ecs_table_t*), by_ptr);
        for (uint32_t i = 0; i < s->n_deleted; ++i)
            if (!nd || s->deleted[nd - 1] != s->deleted[i]) s->deleted[nd++] = s->deleted[i];
    }
    int rc = 1;
    const uint32_t nb_old = s->n_batches;
    uint8_t* seen = (uint8_t*)calloc((size_t)nt + 1, 1);       /* listed table i had a batch */
    Batch* fresh = (Batch*)malloc(sizeof(Batch) * ((size_t)nb_old + nt + 1));
    Map left = {0}; /* entities of batches that die or are read afresh -> 0, those found again in a listed table -> 1 */
    uint64_t rows_old = 0;
    uint32_t nf = 0;
    uint64_t n = 0;
    if (!seen || !fresh) {
        rc = YH_ERR_OOM;
        goto done;
    }
    /* A listed table with a WorldTransform but no LocalTransform holds (or held) entities that are parents only - the
     * scene root's kind: one of them came or went, so the Up term of tables that are not listed may resolve differently
     * now.  (A table with the LocalTransform alone is what an entity passes through while it is being created: never
     * matched, never a source.) */
    for (uint32_t i = 0; i < nt; ++i) {
        const int w = ecs_table_get_column_index(s->world, s->touched[i], s->world_id) >= 0;
        const int l = ecs_table_get_column_index(s->world, s->touched[i], s->local_id) >= 0;
        if (w && !l) goto done; /* rc = 1 */
    }
    /* old batches: dead ones and those of deleted tables go, those of listed tables are read afresh below */
    for (uint32_t b = 0; b < nb_old; ++b) {
        const Batch* bt = &s->batches[b];
        if (!bt->count || !bt->table) continue;
        uint32_t lo = 0, hi = nd;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) / 2;
            if ((uintptr_t)s->deleted[mid] < (uintptr_t)bt->table) lo = mid + 1;
            else hi = mid;
        }
        const int deleted = lo < nd && s->deleted[lo] == bt->table;
        lo = 0, hi = nt;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) / 2;
            if ((uintptr_t)s->touched[mid] < (uintptr_t)bt->table) lo = mid + 1;
            else hi = mid;
        }
        const int listed = !deleted && lo < nt && s->touched[lo] == bt->table;
        if (deleted || listed) {
            rows_old += bt->count;
            if (listed) seen[lo] = 1;
            fresh[nf] = *bt; /* (kept for the moment, marked: its old rows are needed for the map below) */
            fresh[nf].level = deleted ? -2 : -3;
            fresh[nf].walk_begin = lo; /* index into touched[] */
            nf++;
            continue;
        }
        fresh[nf] = *bt;
        fresh[nf].level = -1;
        nf++;
    }
    {
        uint64_t rows_new = 0;
        for (uint32_t i = 0; i < nt; ++i) rows_new += (uint64_t)ecs_table_count(s->touched[i]);
        if (map_init(&left, rows_old + rows_new + 8) != 0) {
            rc = YH_ERR_OOM;
            goto done;
        }
    }
    for (uint32_t b = 0; b < nf; ++b)
        if (fresh[b].level <= -2)
            for (uint32_t r = 0; r < fresh[b].count; ++r) map_put(&left, s->entity_of_slot[fresh[b].slot_begin + r], 0);
    /* the listed tables as they are now */
    uint32_t kept = 0;
    for (uint32_t b = 0; b < nf; ++b) { /* compact: batches of deleted tables drop out, listed ones are re-read in place */
        Batch bt = fresh[b];
        if (bt.level == -2) continue;
        if (bt.level == -3) {
            ecs_table_t* t = s->touched[bt.walk_begin];
            bt.col_w = ecs_table_get_column_index(s->world, t, s->world_id);
            bt.col_l = ecs_table_get_column_index(s->world, t, s->local_id);
            bt.count = bt.col_w >= 0 && bt.col_l >= 0 ? (uint32_t)ecs_table_count(t) : 0u;
            if (bt.count && (ecs_table_has_id(s->world, t, EcsDisabled) || ecs_table_has_id(s->world, t, EcsPrefab))) bt.count = 0;
            if (!bt.count) continue;
            bt.level = -3; /* (entities from the table, not from the slots) */
        }
        fresh[kept++] = bt;
    }
    nf = kept;
    for (uint32_t i = 0; i < nt; ++i) { /* tables this system has not held */
        if (seen[i]) continue;
        ecs_table_t* t = s->touched[i];
        Batch bt;
        memset(&bt, 0, sizeof bt);
        bt.table = t;
        bt.col_w = ecs_table_get_column_index(s->world, t, s->world_id);
        bt.col_l = ecs_table_get_column_index(s->world, t, s->local_id);
        bt.count = bt.col_w >= 0 && bt.col_l >= 0 ? (uint32_t)ecs_table_count(t) : 0u;
        if (bt.count && (ecs_table_has_id(s->world, t, EcsDisabled) || ecs_table_has_id(s->world, t, EcsPrefab))) bt.count = 0;
        if (!bt.count) continue;
        bt.src = ecs_get_parent(s->world, ecs_table_entities(t)[0]); /* ChildOf is part of the archetype: one parent per table */
        if (!bt.src) continue;                                        /* no parent: the Up term does not match this table */
        if (!ecs_has_id(s->world, bt.src, s->world_id)) goto done;    /* Up resolves further up, or not at all: rc = 1 */
        bt.level = -3;
        fresh[nf++] = bt;
    }
    for (uint32_t b = 0; b < nf; ++b)
        if (fresh[b].level == -3) {
            const ecs_entity_t* ents = ecs_table_entities(fresh[b].table);
            for (uint32_t r = 0; r < fresh[b].count; ++r) map_put(&left, ents[r], 1);
        }
    /* a table that stays while its parent row left is not something this list can describe: the query decides */
    for (uint32_t b = 0; b < nf; ++b) {
        uint32_t there;
        if (fresh[b].src != s->root && map_get(&left, fresh[b].src, &there) && !there) goto done; /* rc = 1 */
        n += fresh[b].count;
    }
    if (n > 0x40000000ull || !nf) goto done;
    if (reserve_rows(s, n) != YH_OK) {
        rc = YH_ERR_OOM;
        goto done;
    }
    {   /* walk order = this list's order; entities of untouched batches from their slots */
        uint64_t row = 0;
        for (uint32_t b = 0; b < nf; ++b) {
            Batch* bt = &fresh[b];
            /* (walk_entities is not entity_of_slot: no overlap) */
            if (bt->level == -3) memcpy(s->walk_entities + row, ecs_table_entities(bt->table), sizeof(ecs_entity_t) * bt->count);
            else memcpy(s->walk_entities + row, s->entity_of_slot + bt->slot_begin, sizeof(ecs_entity_t) * bt->count);
            row += bt->count;
        }
        row = 0;
        for (uint32_t b = 0; b < nf; ++b) {
            Batch* bt = &fresh[b];
            bt->world_col = bt->local_col = NULL; /* fetched on first use (batch_cols) */
            bt->up_row = NULL;
            bt->walk_begin = (uint32_t)row;
            bt->slot_begin = 0;
            bt->level = -1;
            row += bt->count;
        }
    }
    free(s->batches);
    s->batches = fresh;
    s->cap_batches = nb_old + nt + 1;
    s->n_batches = nf;
    s->n = n;
    fresh = NULL;
    rc = YH_OK;
done:
    free(seen);
    free(fresh);
    map_free(&left);
    return rc;
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
        const int ld = lrow[64] != 0, wd = wrow[64] != 0;
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
            if (s->host_stale && s->scene && (rc = catch_up_by_entity(s)) != YH_OK) {
                s->status = rc;
                return;
            }
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
    if (s->bounds_stale && (rc = upload_bounds(s)) != YH_OK) {
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
            const uint64_t m = gather_dirty(s, k, stage);
            rc = k == 0 ? yh_upload_local_rows_aos68(s->scene, s->slot_buf, stage, m)
                        : yh_upload_world_rows_aos68(s->scene, s->slot_buf, stage, m);
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
    rc = yh_frame(s->scene, s->planes, s->n_frusta, cull_flags); /* returns when the device is done */
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
        if ((rc = yh_get_changed_counts(s->scene, &mw, &ml)) != YH_OK || (mw && (rc = fetch_changed(s, 1, mw)) != YH_OK) ||
            (ml && (rc = fetch_changed(s, 0, ml)) != YH_OK)) {
            s->status = rc;
            return;
        }
        s->last_down_rows = mw + ml;
    }
    s->n_dirty[0] = s->n_dirty[1] = 0;
    if (root_row) root_row[64] = 0; /* PostWorldTransform reaches the root too (single-term query) */
    s->last_ms[0] = t_inputs - t_begin;
    s->last_ms[1] = t_up - t_inputs;
    s->last_ms[2] = t_frame - t_up;
    s->last_ms[3] = now_ms() - t_frame;
    s->status = YH_OK;
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

static void note_mesh_set(ecs_iter_t* it) { ((GpuTransformCullSystem*)it->ctx)->bounds_stale = 1; }

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
    s->threads = n > MAX_THREADS ? MAX_THREADS : n;
    return YH_OK;
}

int GpuTransformCullSystem_structureDirty(const GpuTransformCullSystem* s) { return s ? s->structure_dirty : -1; }
void GpuTransformCullSystem_clearStructureDirty(GpuTransformCullSystem* s) {
    if (s) s->structure_dirty = 0;
}
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
    if (s->scene) yh_scene_destroy(s->scene);
    if (s->camera_query) ecs_query_fini(s->camera_query);
    free(s->batches);
    free(s->walk_entities);
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
