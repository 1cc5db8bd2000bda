/*
 * flecs_harness.c — CPU ORACLE validation against the REAL flecs (test infrastructure, this container only).
 *
 * Links the reference's vendored flecs 4.0.4 (compiled in place from /root/reference/vendor/flecs/flecs.c by
 * oracle/Makefile, never copied) and drives the transform path exactly the way the reference registers it:
 *   - components LocalTransform / WorldTransform: 68 bytes, align 4, hooks onAdd / onSet
 *     (engine/components/transform.zig:8-12, 38-51, 62-66, 92-97; registration engine/ecs.zig:110-182)
 *   - the scene root owns a WorldTransform{IDENTITY} (engine/GameApp.zig:77,95), every entity is created
 *     parent-first under it with a (ChildOf, parent) pair (GameApp.zig:204-215)
 *   - TransformSyncSystem's three query terms: WorldTransform inout, LocalTransform inout, WorldTransform in
 *     with src = Up (engine/systems/TransformSyncSystem.zig:25-37), then PostWorldTransform and
 *     PostLocalTransform (engine/systems/PostTransformSystem.zig:12-37).
 * The per-batch arithmetic is the oracle's restatement (yume_oracle.c) — Zig cannot be compiled here — so what
 * this binary pins is the ECS side: which rows flecs batches together, that the `Up` field is ONE shared value
 * per batch, parent-first table order (F5), and dirty-flag propagation across frames.
 *
 * Built twice by oracle/Makefile: as is (the reference's three CPU systems), and with -DUSE_GPU_SYSTEM, where the
 * SAME world, components, hooks, scene and edits are driven through yume_amd/host/GpuTransformCullSystem.c instead
 * (binary flecs_gpu_harness, needs a GPU).  tests/test_gpu_flecs_system.py compares the two outputs byte for byte.
 *
 * usage: flecs_harness <scene.bin> <out.bin>
 * scene.bin: "YFH1"|"YFH2" u32 n, n_levels, n_frames | u32 level_offsets[n_levels+1] | u32 parent[n]
 *            | f32 local[n][16] | f32 root[16]
 *            | YFH2 only: u32 n_frusta, f32 planes[n_frusta][24], f32 bounds[n][6] (min.xyz, max.xyz)
 *            | per frame: u32 n_edits, then {u32 entity (0xFFFFFFFF = root), u32 kind, f32 m[16]} applied before the
 *              frame.  kind 0 / 1: set + modified of the local / world matrix (editor/editors/transform.zig:76-79);
 *              kind 2: delete the entity (children go with it, flecs' ChildOf cleanup);
 *              kind 3: re-parent under entity index bits(m[0]) (ecs.zig:470-477 setParent);
 *              kinds 4 / 5: change an entity's Mesh / a camera's view_projection (GPU build with YFH_ECS_INPUTS=1, see below)
 *              kind 6: create entity `entity` - one whose parent[] entry is 0xFFFFFFFE, i.e. not created at start - as
 *              a child of entity index bits(m[0]), with its local matrix (and bounds) from the scene's arrays
 *              (GameApp.zig:204-215: new(), setParent, set LocalTransform)
 *              kind 7: world.enable(entity, bits(m[0]) != 0) (ecs.zig:546)
 *              kind 8: the entity's Material := material bits(m[0]) (GPU build with YFH_MATERIALS=1)
 * out.bin:   per frame: world rows n x 68 B, local rows n x 68 B (all zero for deleted entities), u64 batches,
 *            u64 rows_seen, u32 up_fields_shared; GPU build + YFH2: then per frustum u32 count, u32 entity_index[count]
 *            (GPU build: batches = slot-order rebuilds so far, rows_seen = entities in the device scene), then
 *            u64 rows_up, u64 rows_down = 68-byte rows that crossed PCIe in the frame, u64 walks, u64 walks_skipped, u64 patches
 *            (query iterations done / frames that reused the last one).  GPU build, env YFH_PLAIN=1: whole columns both
 *            ways every frame instead of the incremental mode; YFH_ALWAYS_WALK=1: iterate the query every frame;
 *            YFH_NO_READBACK=1: rows stay on the device (setReadback(0)) until the last frame, which brings whole columns back;
 *            YFH_VALIDATE=1: do not trust the structural-change observers, compare every cached table every frame;
 *            YFH_NO_PATCH=1: setPatching(0); YFH_READBACK_OFF_FROM=k: setReadback(0) from frame k to the last but one.
 *            YFH_CAMERA_PIPELINE="i,j" (implies YFH_ECS_INPUTS): the two Camera components sit on scene entities i and j
 *            (camera 0 perspective 75 deg / 0.1 / 200, camera 1 orthographic) with all three matrices zero, and the system
 *            derives this frame's matrices itself (setCameraPipeline, aspect 2); after `patches` every frame then writes
 *            per camera f32 view[16], projection[16], view_projection[16] as the component holds them.
 *            YFH_MATERIALS=1 (implies YFH_ECS_INPUTS): every entity with a mesh also gets a Material{ref} component,
 *            material (e / 4) % 3 of three; after the above every frame writes frustum 0's draw buckets
 *            (GpuTransformCullSystem_drawBuckets): u32 n_buckets, per bucket u32 material, mesh (indices), first, count,
 *            then u64 n_records and u32 entity_index[n_records] in record order.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "flecs.h"
#include "yume_oracle.h"
#ifdef USE_GPU_SYSTEM
#include "GpuTransformCullSystem.h"
#endif

static ecs_entity_t g_local_id, g_world_id;
static uint64_t g_batches, g_rows;
static uint32_t g_up_shared = 1;

/* LocalTransform.onAdd (transform.zig:38-44): give the entity a WorldTransform{IDENTITY, dirty}. */
static void local_on_add(ecs_iter_t* it) {
    ecs_world_t* w = ecs_get_world(it->world);
    yo_transform68 wt;
    memset(&wt, 0, sizeof wt);
    yo_identity(&wt.matrix);
    wt.dirty = 1;
    for (int i = 0; i < it->count; ++i) ecs_set_id(w, it->entities[i], g_world_id, sizeof wt, &wt);
}

/* onSet of either component (transform.zig:46-51, 92-97): dirty = true. */
static void any_on_set(ecs_iter_t* it) {
    yo_transform68* rows = (yo_transform68*)ecs_field_w_size(it, sizeof(yo_transform68), 0);
    if (!rows) return;
    for (int i = 0; i < it->count; ++i) rows[i].dirty = 1;
}

/* TransformSyncSystem.zig:41-49 -> :53-74 */
static void transform_sync(ecs_iter_t* it) {
    yo_transform68* wts = (yo_transform68*)ecs_field_w_size(it, sizeof(yo_transform68), 0);
    yo_transform68* lts = (yo_transform68*)ecs_field_w_size(it, sizeof(yo_transform68), 1);
    const yo_transform68* parent = (const yo_transform68*)ecs_field_w_size(it, sizeof(yo_transform68), 2);
    if (ecs_field_is_self(it, 2)) g_up_shared = 0; /* the Up term must NOT be a per-row column */
    yo_transform_sync_batch(wts, lts, parent, (size_t)it->count);
    g_batches += 1;
    g_rows += (uint64_t)it->count;
}

/* PostTransformSystem.zig:31-35 */
static void post_clear(ecs_iter_t* it) {
    yo_post_transform((yo_transform68*)ecs_field_w_size(it, sizeof(yo_transform68), 0), (size_t)it->count);
}

static ecs_entity_t make_component(ecs_world_t* w, const char* name, ecs_iter_action_t on_add) {
    ecs_entity_desc_t ed = {0};
    ed.use_low_id = true;
    ed.name = name;
    ed.symbol = name;
    ecs_component_desc_t cd = {0};
    cd.entity = ecs_entity_init(w, &ed);
    cd.type.size = 68; /* @sizeOf */
    cd.type.alignment = 4; /* @alignOf */
    cd.type.hooks.on_add = on_add;
    cd.type.hooks.on_set = any_on_set;
    return ecs_component_init(w, &cd);
}

static ecs_entity_t make_system(ecs_world_t* w, const char* name, ecs_iter_action_t cb, int n_terms) {
    ecs_entity_desc_t ed = {0};
    ed.name = name;
    ecs_system_desc_t sd = {0};
    sd.entity = ecs_entity_init(w, &ed);
    sd.callback = cb;
    if (n_terms == 3) {
        sd.query.terms[0].id = g_world_id;
        sd.query.terms[0].inout = EcsInOut;
        sd.query.terms[1].id = g_local_id;
        sd.query.terms[1].inout = EcsInOut;
        sd.query.terms[2].id = g_world_id;
        sd.query.terms[2].inout = EcsIn;
        sd.query.terms[2].src.id = EcsUp;
    } else {
        sd.query.terms[0].id = (n_terms == 1) ? g_world_id : g_local_id;
        sd.query.terms[0].inout = EcsInOut;
    }
    return ecs_system_init(w, &sd);
}

#define RD(ptr, sz)                                            \
    if (fread((ptr), 1, (sz), fin) != (size_t)(sz)) {          \
        fprintf(stderr, "flecs_harness: short read\n");        \
        return 2;                                              \
    }

int main(int argc, char** argv) {
    if (argc != 3) {
        fprintf(stderr, "usage: %s scene.bin out.bin\n", argv[0]);
        return 2;
    }
    FILE* fin = fopen(argv[1], "rb");
    FILE* fout = fopen(argv[2], "wb");
    if (!fin || !fout) return 2;
    char magic[4];
    uint32_t n, n_levels, n_frames;
    RD(magic, 4);
    const int v2 = memcmp(magic, "YFH2", 4) == 0;
    if (!v2 && memcmp(magic, "YFH1", 4) != 0) return 2;
    RD(&n, 4);
    RD(&n_levels, 4);
    RD(&n_frames, 4);
    uint32_t* lo = malloc(sizeof(uint32_t) * (n_levels + 1));
    uint32_t* parent = malloc(sizeof(uint32_t) * n);
    float* local = malloc(sizeof(float) * 16 * (size_t)n);
    float root_m[16];
    RD(lo, 4 * (n_levels + 1));
    RD(parent, 4 * (size_t)n);
    RD(local, 64 * (size_t)n);
    RD(root_m, 64);
    uint32_t n_frusta = 0;
    float planes[8 * 24];
    float* bounds = NULL;
    if (v2) {
        RD(&n_frusta, 4);
        if (n_frusta > 8) return 2;
        RD(planes, 96 * n_frusta);
        bounds = malloc(sizeof(float) * 6 * (size_t)n);
        RD(bounds, 24 * (size_t)n);
    }

    ecs_world_t* w = ecs_init();
    g_world_id = make_component(w, "WorldTransform", NULL);
    g_local_id = make_component(w, "LocalTransform", local_on_add);

    /* scene root: world.set(scene_root, WorldTransform{IDENTITY}) (GameApp.zig:95) */
    ecs_entity_t root = ecs_new(w);
    yo_transform68 t;
    memset(&t, 0, sizeof t);
    memcpy(t.matrix.v, root_m, 64);
    ecs_set_id(w, root, g_world_id, sizeof t, &t);

    ecs_entity_t* ent = calloc(n, sizeof(ecs_entity_t));
    for (uint32_t e = 0; e < n; ++e) { /* level-major order = parent first (GameApp.zig:204-215) */
        if (parent[e] == 0xFFFFFFFEu) continue; /* created later, by an edit of kind 6 */
        const ecs_entity_t p = parent[e] == 0xFFFFFFFFu ? root : ent[parent[e]];
        ent[e] = ecs_new_w_id(w, ecs_pair(EcsChildOf, p));
        memset(&t, 0, sizeof t);
        memcpy(t.matrix.v, local + 16 * (size_t)e, 64);
        ecs_set_id(w, ent[e], g_local_id, sizeof t, &t);
    }

#ifdef USE_GPU_SYSTEM
    /* YFH_ECS_INPUTS=1: the system reads its cull inputs from the ECS, as in the engine — frusta from Camera entities
     * (240-byte component, view_projection at offset 176: engine/components/camera.zig:54-66), bounds from the mesh a
     * Mesh{ref} component points to (engine/components/mesh.zig, engine/meshes.zig:104-110) — instead of setFrusta /
     * setBounds.  Entity e refers to mesh e % 4 - 1 (none for e % 4 == 0); camera 0 is the benchmark's main camera,
     * camera 1 its second shadow cascade.  Edit kinds 4 (entity's Mesh := mesh bits(m[0]) - 1) and 5 (camera e's
     * view_projection := m) exist in this mode only. */
    typedef struct FakeMesh { /* the layout of engine/meshes.zig Mesh up to its bounds */
        uint8_t handle[16];
        uint64_t vertices_count;
        void* vertices;
        float bounds[6];
        uint8_t vertex_buffer[32];
    } FakeMesh;
    static FakeMesh fake_meshes[3];
    const char* cam_pipeline = getenv("YFH_CAMERA_PIPELINE");
    const int with_materials = getenv("YFH_MATERIALS") != NULL;
    const int ecs_inputs = getenv("YFH_ECS_INPUTS") != NULL || cam_pipeline != NULL || with_materials;
    static uint8_t fake_materials[3][64]; /* stand-ins for shading.Material: only their addresses matter */
    ecs_entity_t material_id = 0;
    ecs_entity_t camera_id = 0, mesh_id = 0, cams[2] = {0, 0};
    if (ecs_inputs) {
        ecs_entity_desc_t ed = {0};
        ed.use_low_id = true;
        ed.name = "Camera";
        ecs_component_desc_t cd = {0};
        cd.entity = ecs_entity_init(w, &ed);
        cd.type.size = 240;
        cd.type.alignment = 4;
        camera_id = ecs_component_init(w, &cd);
        ecs_entity_desc_t ed2 = {0};
        ed2.use_low_id = true;
        ed2.name = "Mesh";
        ecs_component_desc_t cd2 = {0};
        cd2.entity = ecs_entity_init(w, &ed2);
        cd2.type.size = 8;
        cd2.type.alignment = 8;
        mesh_id = ecs_component_init(w, &cd2);
        for (int k = 1; k <= 3; ++k) {
            FakeMesh* m = &fake_meshes[k - 1];
            memset(m, 0xAB, sizeof *m);
            m->vertices_count = 36u * (uint64_t)k;
            const float b[6] = {-0.5f * k, -0.25f * k - 0.5f, -1.0f, 0.5f * k, 0.25f * k + 0.5f, 1.5f};
            memcpy(m->bounds, b, sizeof b);
        }
        if (with_materials) {
            ecs_entity_desc_t ed3 = {0};
            ed3.use_low_id = true;
            ed3.name = "Material";
            ecs_component_desc_t cd3 = {0};
            cd3.entity = ecs_entity_init(w, &ed3);
            cd3.type.size = 8;
            cd3.type.alignment = 8;
            material_id = ecs_component_init(w, &cd3);
        }
        for (uint32_t e = 0; e < n; ++e)
            if (e % 4 && ent[e]) {
                const FakeMesh* ref = &fake_meshes[e % 4 - 1];
                ecs_set_id(w, ent[e], mesh_id, sizeof ref, &ref);
                if (with_materials) {
                    const void* mat = fake_materials[(e / 4) % 3];
                    ecs_set_id(w, ent[e], material_id, sizeof mat, &mat);
                }
            }
        uint8_t cam[240];
        if (cam_pipeline) { /* cameras ON scene entities, matrices left for the system to derive */
            unsigned ci = 0, cj = 0;
            if (sscanf(cam_pipeline, "%u,%u", &ci, &cj) != 2 || ci >= n || cj >= n || !ent[ci] || !ent[cj]) return 2;
            const float persp[3] = {(float)(75.0 * 3.14159265358979323846 / 180.0), 0.1f, 200.0f};
            memset(cam, 0, sizeof cam);
            memcpy(cam + 4, persp, sizeof persp); /* opts.data.perspective (camera.zig:28-32) */
            cams[0] = ent[ci];
            ecs_set_id(w, cams[0], camera_id, sizeof cam, cam);
            memset(cam, 0, sizeof cam);
            cam[0] = 1; /* CameraKind.orthographic */
            cams[1] = ent[cj];
            ecs_set_id(w, cams[1], camera_id, sizeof cam, cam);
        } else {
        float proj[16], view[16], vp[16];
        const float eye0[3] = {0.0f, 40.0f, 150.0f}, at[3] = {0.0f, 0.0f, 0.0f}, up[3] = {0.0f, 1.0f, 0.0f};
        yh_perspective((float)(75.0 * 3.14159265358979323846 / 180.0), 2.0f, 0.1f, 200.0f, proj);
        proj[5] = proj[5] * -1.0f;
        yh_look_at(eye0, at, up, view);
        yh_mat4_mul(proj, view, vp);
        memset(cam, 0x5A, sizeof cam);
        memcpy(cam + 176, vp, 64);
        cams[0] = ecs_new(w);
        ecs_set_id(w, cams[0], camera_id, sizeof cam, cam);
        const float eye1[3] = {0.4f * 200.0f, 1.0f * 200.0f, 0.3f * 200.0f};
        yh_orthographic(-25.0f, 25.0f, -25.0f, 25.0f, 0.0f, 400.0f, proj);
        yh_look_at(eye1, at, up, view);
        yh_mat4_mul(proj, view, vp);
        memcpy(cam + 176, vp, 64);
        cams[1] = ecs_new(w);
        ecs_set_id(w, cams[1], camera_id, sizeof cam, cam);
        }
    }
    yh_ctx* ctx = NULL;
    if (yh_init(0, &ctx) != YH_OK) {
        fprintf(stderr, "flecs_gpu_harness: yh_init: %s\n", yh_last_error(NULL));
        return 3;
    }
    GpuTransformCullSystem* gpu = NULL;
    const ecs_entity_t gsys = GpuTransformCullSystem_registerTo(w, g_local_id, g_world_id, ctx, &gpu);
    if (!gsys) return 3;
    if (ecs_inputs) {
        if (GpuTransformCullSystem_setCameraComponent(gpu, camera_id, 176) != YH_OK ||
            GpuTransformCullSystem_setMeshComponent(gpu, mesh_id, 32, 16) != YH_OK)
            return 3;
        if (with_materials && GpuTransformCullSystem_setMaterialComponent(gpu, material_id) != YH_OK) return 3;
        if (cam_pipeline) { /* @offsetOf of opts.kind, opts.data, view, projection, view_projection (camera.zig:42-66) */
            const GpuCameraLayout layout = {0, 4, 48, 112, 176};
            if (GpuTransformCullSystem_setCameraPipeline(gpu, &layout, 2.0f) != YH_OK) return 3;
        }
        n_frusta = 2;
    } else if (v2) {
        for (uint32_t e = 0; e < n; ++e)
            if (ent[e]) GpuTransformCullSystem_setBounds(gpu, ent[e], bounds + 6 * (size_t)e);
        GpuTransformCullSystem_setFrusta(gpu, planes, n_frusta);
    }
    if (getenv("YFH_PLAIN")) GpuTransformCullSystem_setIncremental(gpu, 0);
    if (getenv("YFH_ALWAYS_WALK")) GpuTransformCullSystem_setSkipWalks(gpu, 0);
    const int no_readback = getenv("YFH_NO_READBACK") != NULL; /* results stay on the device until the last frame */
    if (no_readback) GpuTransformCullSystem_setReadback(gpu, 0);
    if (getenv("YFH_THREADS")) GpuTransformCullSystem_setThreads(gpu, atoi(getenv("YFH_THREADS")));
    if (getenv("YFH_VALIDATE")) GpuTransformCullSystem_setTrustObservers(gpu, 0); /* compare every table every frame */
    if (getenv("YFH_NO_PATCH")) GpuTransformCullSystem_setPatching(gpu, 0);       /* every structural change: walk + rebuild */
    ecs_entity_t* vis = malloc(sizeof(ecs_entity_t) * n);
    uint32_t* vis_idx = malloc(sizeof(uint32_t) * n);
    uint32_t* idx_of_id = NULL;
    uint32_t idx_cap = 0;
    int idx_fresh = 0;
#else
    (void)planes;
    const ecs_entity_t tss = make_system(w, "TransformSyncSystem", transform_sync, 3);
    const ecs_entity_t post_w = make_system(w, "PostWorldTransform", post_clear, 1);
    const ecs_entity_t post_l = make_system(w, "PostLocalTransform", post_clear, 2);
#endif

    yo_transform68* rows = malloc(sizeof(yo_transform68) * n);
    for (uint32_t f = 0; f < n_frames; ++f) {
        uint32_t n_edits;
        RD(&n_edits, 4);
        for (uint32_t i = 0; i < n_edits; ++i) {
            uint32_t e, kind;
            float m[16];
            RD(&e, 4);
            RD(&kind, 4);
            RD(m, 64);
            if (kind == 6) {
                uint32_t np;
                memcpy(&np, m, 4);
                ent[e] = ecs_new_w_id(w, ecs_pair(EcsChildOf, np == 0xFFFFFFFFu ? root : ent[np]));
                memset(&t, 0, sizeof t);
                memcpy(t.matrix.v, local + 16 * (size_t)e, 64);
                ecs_set_id(w, ent[e], g_local_id, sizeof t, &t);
#ifdef USE_GPU_SYSTEM
                if (ecs_inputs && e % 4) {
                    const FakeMesh* ref = &fake_meshes[e % 4 - 1];
                    ecs_set_id(w, ent[e], mesh_id, sizeof ref, &ref);
                    if (with_materials) {
                        const void* mat = fake_materials[(e / 4) % 3];
                        ecs_set_id(w, ent[e], material_id, sizeof mat, &mat);
                    }
                } else if (!ecs_inputs && v2) {
                    GpuTransformCullSystem_setBounds(gpu, ent[e], bounds + 6 * (size_t)e);
                }
#endif
                continue;
            }
            const ecs_entity_t target = e == 0xFFFFFFFFu ? root : ent[e];
            if (kind == 2) {
                ecs_delete(w, target);
                continue;
            }
            if (kind == 3) {
                uint32_t np;
                memcpy(&np, m, 4);
                ecs_add_id(w, target, ecs_pair(EcsChildOf, np == 0xFFFFFFFFu ? root : ent[np]));
                continue;
            }
            if (kind == 7) { /* world.enable(entity, bits(m[0]) != 0) (ecs.zig:546): the Disabled tag comes or goes */
                uint32_t on;
                memcpy(&on, m, 4);
                ecs_enable(w, target, on != 0);
                continue;
            }
            if (kind == 8) {
#ifdef USE_GPU_SYSTEM
                if (with_materials) {
                    uint32_t k;
                    memcpy(&k, m, 4);
                    const void* mat = fake_materials[k % 3];
                    ecs_set_id(w, target, material_id, sizeof mat, &mat);
                }
#endif
                continue;
            }
            if (kind == 4 || kind == 5) { /* cull inputs held by the ECS: GPU build with YFH_ECS_INPUTS only */
#ifdef USE_GPU_SYSTEM
                if (ecs_inputs && kind == 4) {
                    uint32_t k;
                    memcpy(&k, m, 4);
                    const FakeMesh* ref = &fake_meshes[k - 1];
                    ecs_set_id(w, target, mesh_id, sizeof ref, &ref);
                } else if (ecs_inputs) {
                    uint8_t* cam = (uint8_t*)ecs_ensure_id(w, cams[e], camera_id);
                    memcpy(cam + 176, m, 64);
                    ecs_modified_id(w, cams[e], camera_id);
                }
#endif
                continue;
            }
            const ecs_entity_t comp = kind == 0 ? g_local_id : g_world_id;
            yo_transform68* p = (yo_transform68*)ecs_ensure_id(w, target, comp);
            memcpy(p->matrix.v, m, 64);
            ecs_modified_id(w, target, comp); /* -> onSet -> dirty */
        }
        g_batches = g_rows = 0;
#ifdef USE_GPU_SYSTEM
        if (no_readback && f + 1 == n_frames) GpuTransformCullSystem_setReadback(gpu, 1); /* the last frame catches flecs up */
        if (getenv("YFH_READBACK_OFF_FROM")) { /* readback on, then off from that frame, on again for the last one */
            const uint32_t from = (uint32_t)atoi(getenv("YFH_READBACK_OFF_FROM"));
            if (f == from) GpuTransformCullSystem_setReadback(gpu, 0);
            if (f + 1 == n_frames) GpuTransformCullSystem_setReadback(gpu, 1);
        }
#endif
        if (f > 0 && getenv("YFH_FRAME_GAP_MS")) { /* an engine's frames are apart: idle time between them (measurement aid, untimed) */
            struct timespec gap = {atol(getenv("YFH_FRAME_GAP_MS")) / 1000, 1000000L * (atol(getenv("YFH_FRAME_GAP_MS")) % 1000)};
            nanosleep(&gap, NULL);
        }
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        /* PostUpdate: TransformSyncSystem, then its two dependents (ecs.zig:277-293) */
#ifdef USE_GPU_SYSTEM
        ecs_run(w, gsys, 0.016f, NULL);
        if (GpuTransformCullSystem_status(gpu) != YH_OK) {
            fprintf(stderr, "flecs_gpu_harness: frame %u: status %d: %s\n", f, GpuTransformCullSystem_status(gpu),
                    yh_last_error(ctx));
            return 4;
        }
        g_batches = GpuTransformCullSystem_rebuilds(gpu);
        g_rows = GpuTransformCullSystem_entities(gpu);
#else
        ecs_run(w, tss, 0.016f, NULL);
        ecs_run(w, post_w, 0.016f, NULL);
        ecs_run(w, post_l, 0.016f, NULL);
#endif
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (getenv("YFH_TIMES")) /* wall time of the frame's systems, edits excluded */
        {
            fprintf(stderr, "frame %u: %.3f ms, %u edits\n", f, (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6, n_edits);
#ifdef USE_GPU_SYSTEM
            double st[6];
            GpuTransformCullSystem_lastTimes(gpu, st);
            fprintf(stderr, "stages %u: inputs %.3f (walk %.3f, rebuild %.3f) up %.3f frame %.3f down %.3f; cull-only frames so far %llu; idle frames so far %llu\n", f, st[0],
                    st[4], st[5], st[1], st[2], st[3], (unsigned long long)GpuTransformCullSystem_cullOnlyFrames(gpu),
                    (unsigned long long)GpuTransformCullSystem_idleFrames(gpu));
#endif
        }
        for (int comp = 0; comp < 2; ++comp) {
            for (uint32_t e = 0; e < n; ++e) {
                memset(&rows[e], 0, sizeof rows[e]);
                if (!ent[e] || !ecs_is_alive(w, ent[e])) continue;
                const yo_transform68* p = (const yo_transform68*)ecs_get_id(w, ent[e], comp == 0 ? g_world_id : g_local_id);
                rows[e] = *p;
                rows[e].pad[0] = rows[e].pad[1] = rows[e].pad[2] = 0;
            }
            fwrite(rows, sizeof(yo_transform68), n, fout);
        }
        fwrite(&g_batches, 8, 1, fout);
        fwrite(&g_rows, 8, 1, fout);
        fwrite(&g_up_shared, 4, 1, fout);
#ifdef USE_GPU_SYSTEM
        for (uint32_t fr = 0; fr < n_frusta; ++fr) {
            const uint64_t cnt = GpuTransformCullSystem_visible(gpu, fr, vis, n);
            {   /* entity id -> scene index: the low 32 bits of an id are its index in flecs' entity table */
                uint32_t max_id = 0;
                for (uint32_t e = 0; e < n; ++e)
                    if ((uint32_t)ent[e] > max_id) max_id = (uint32_t)ent[e];
                if (max_id >= idx_cap) {
                    free(idx_of_id);
                    idx_cap = max_id + 1024;
                    idx_of_id = malloc(sizeof(uint32_t) * idx_cap);
                    idx_fresh = 0;
                }
                if (!idx_fresh || n_edits) {
                    for (uint32_t e = 0; e < n; ++e)
                        if (ent[e]) idx_of_id[(uint32_t)ent[e]] = e;
                    idx_fresh = 1;
                }
            }
            for (uint64_t i = 0; i < cnt; ++i) vis_idx[i] = idx_of_id[(uint32_t)vis[i]];
            const uint32_t c32 = (uint32_t)cnt;
            fwrite(&c32, 4, 1, fout);
            fwrite(vis_idx, 4, c32, fout);
        }
        uint64_t traffic[2];
        GpuTransformCullSystem_lastTraffic(gpu, &traffic[0], &traffic[1]);
        fwrite(traffic, 8, 2, fout);
        uint64_t walk_stats[2];
        GpuTransformCullSystem_walkStats(gpu, &walk_stats[0], &walk_stats[1]);
        fwrite(walk_stats, 8, 2, fout);
        const uint64_t patches = GpuTransformCullSystem_patches(gpu);
        fwrite(&patches, 8, 1, fout);
        if (cam_pipeline)
            for (int k = 0; k < 2; ++k) {
                const uint8_t* cam = (const uint8_t*)ecs_get_id(w, cams[k], camera_id);
                if (!cam) return 5;
                fwrite(cam + 48, 64, 3, fout);
                if (GpuTransformCullSystem_frustumCamera(gpu, (uint32_t)k) != cams[k]) return 6; /* query order = creation order here */
            }
        if (with_materials) {
            static GpuDrawBucket bk[64];
            uint32_t nb = 0;
            uint64_t nrec = 0;
            if (GpuTransformCullSystem_drawBuckets(gpu, 0, 0, bk, 64, &nb, vis, n, &nrec) != YH_OK) {
                fprintf(stderr, "flecs_gpu_harness: drawBuckets: %s\n", yh_last_error(ctx));
                return 7;
            }
            fwrite(&nb, 4, 1, fout);
            for (uint32_t k = 0; k < nb; ++k) {
                const uint32_t rec[4] = {bk[k].material ? (uint32_t)(((const uint8_t*)bk[k].material - &fake_materials[0][0]) / 64) : 0xFFFFFFFFu,
                                         (uint32_t)((const FakeMesh*)bk[k].mesh - fake_meshes), bk[k].first, bk[k].count};
                fwrite(rec, 4, 4, fout);
            }
            fwrite(&nrec, 8, 1, fout);
            for (uint32_t e = 0; e < n; ++e)
                if (ent[e]) idx_of_id[(uint32_t)ent[e]] = e;
            for (uint64_t i = 0; i < nrec; ++i) vis_idx[i] = idx_of_id[(uint32_t)vis[i]];
            fwrite(vis_idx, 4, nrec, fout);
        }
#endif
    }
#ifdef USE_GPU_SYSTEM
    GpuTransformCullSystem_destroy(gpu);
    yh_shutdown(ctx);
#endif
    fclose(fout);
    fclose(fin);
    ecs_fini(w);
    return 0;
}
