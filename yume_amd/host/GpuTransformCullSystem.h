/*
 * GpuTransformCullSystem — engine-side system that replaces TransformSyncSystem + PostTransformSystem with the
 * MI355X path, written against the flecs C API the engine's ecs.zig wraps.
 *
 * Mirrors the reference's registration interface:
 *     engine/systems/TransformSyncSystem.zig:12   pub fn registerTo(world: ecs.World) ecs.Entity
 *     engine/systems/PostTransformSystem.zig:12   pub fn registerTo(world, scope, transform_sync_sys) ecs.Entity
 *     editor/Editor.zig:153-162                   the call site (register, then tag the returned entity)
 * Same contract: creates a system entity named "GpuTransformCullSystem" in phase PostUpdate over the same three
 * query terms (WorldTransform inout, LocalTransform inout, WorldTransform in src=Up) and returns it; the dirty reset
 * happens inside the same system, so the two PostTransform systems are not registered when this one is.
 * It uses SystemDesc.run (ecs.zig:1209), i.e. it walks all matched tables once per frame instead of being called
 * per table: columns (ecs.field, ecs.zig:695-705) are copied into one depth-sorted staging column, sent through
 * yh_upload_*_aos68 -> yh_frame -> yh_download_*_aos68, and copied back, so every CPU reader of
 * WorldTransform.matrix (RenderApi.zig:1398-1402, SceneWindow.zig:334-351) keeps working.
 *
 * The Zig twin of this file is zig/engine/systems/GpuTransformCullSystem.zig.
 */
#ifndef YUME_GPU_TRANSFORM_CULL_SYSTEM_H
#define YUME_GPU_TRANSFORM_CULL_SYSTEM_H

#include "flecs.h"
#include "yume_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct GpuTransformCullSystem GpuTransformCullSystem;

/* registerTo(world): local_transform / world_transform are the component ids (ecs.typeId(T), ecs.zig:691-693).
 * The context is created by the caller (yh_init) and outlives the system.  Returns the system entity, 0 on error
 * (yh_last_error(ctx) tells why).  *out_state receives the system's state for the calls below. */
ecs_entity_t GpuTransformCullSystem_registerTo(ecs_world_t* world, ecs_entity_t local_transform,
                                               ecs_entity_t world_transform, yh_ctx* ctx,
                                               GpuTransformCullSystem** out_state);

/* Local-space bounds of an entity's mesh (engine/meshes.zig:72-95); entities without one get an empty box at the
 * origin.  min_max: 6 floats.  Takes effect at the next frame. */
int GpuTransformCullSystem_setBounds(GpuTransformCullSystem* s, ecs_entity_t entity, const float min_max[6]);

/* Frusta of the next frames: n x 24 floats (yh_extract_planes of Camera.view_projection, camera.zig:300-302). */
int GpuTransformCullSystem_setFrusta(GpuTransformCullSystem* s, const float* planes, uint32_t n_frusta);

/* Inputs read from the ECS itself (optional).
 * setCameraComponent: the id of the engine's Camera component and @offsetOf(Camera, "view_projection")
 * (engine/components/camera.zig:54-66: 176 in the reference's layout).  Every frame culls against one frustum per
 * Camera entity (query order, at most YH_MAX_FRUSTA), planes taken from its view_projection as the engine last updated
 * it (camera.zig:300-302; the reference does that at draw time, GameWindow.zig:154-178, so a camera that moved is seen
 * one frame late - unless setCameraPipeline, below, is on).  frustumCamera(f): the entity behind
 * frustum f.  Frusta handed over with setFrusta take precedence; setFrusta(NULL, 0) returns to the cameras.
 * setMeshComponent: the id of the Mesh component (`Mesh{ ref: *meshes.Mesh }`, engine/components/mesh.zig), and
 * @offsetOf(meshes.Mesh, "bounds") / @offsetOf(meshes.Mesh, "vertices_count") inside what `ref` points to
 * (engine/meshes.zig:104-110; vertices_count_offset < 0: none).  The device scene then holds one AABB per distinct
 * mesh and a mesh id per entity (entities without a Mesh: an empty box at the origin), so per-mesh draw buckets
 * (yh_build_mesh_buckets on GpuTransformCullSystem_scene) work on the system's scene; bounds follow Mesh
 * sets and structural changes by themselves, invalidateBounds re-reads them when a mesh asset itself was reloaded.
 * setBounds stays as a per-entity override.  At most 65535 distinct meshes + overrides (YH_ERR_CAPACITY beyond). */
int GpuTransformCullSystem_setCameraComponent(GpuTransformCullSystem* s, ecs_entity_t camera, uint32_t view_projection_offset);
/* The camera matrix pipeline: THIS frame's frusta from the cameras' own transforms.  The reference
 * updates a camera's matrices at draw time from the WorldTransform the frame just wrote
 * (editor/windows/GameWindow.zig:154-178: transform.decompose() -> rotation.toEuler() -> camera.updateMatrices(pos,
 * euler, aspect)); culling has to know them before the frame runs.  With a layout set, every frame first works out each
 * Camera entity's world matrix of this frame on the host - the three-branch rule (TransformSyncSystem.zig:60-73)
 * replayed down its chain of `Up` ancestors over the rows the frame is about to read: at most `depth` products per
 * camera, left to right like the kernels, so the value equals the row the frame then writes - runs
 * yh_camera_update_from_world on it, writes view / projection / view_projection into the component as updateMatrices
 * does (components/camera.zig:161-166, 253-302) and culls against those planes.  Without a layout (the default) the
 * frusta come from view_projection as the engine last left it: a camera that moved is seen one frame late.
 * Offsets inside the Camera component (components/camera.zig:42-66): kind = @offsetOf(Camera, "opts") (+ 0: the u8
 * CameraKind, 0 perspective / 1 orthographic), options = offset of `opts.data` (PerspectiveOptions{fovy_rad, near,
 * far}; an orthographic camera's options are not read, camera.zig:291-298 ignores them too), then the three matrices:
 * 0 / 4 / 48 / 112 / 176 in the reference's layout.  aspect: GameWindow.zig:155, game_view_size.x / .y; the host
 * calls setCameraAspect when the view is resized.  Cameras the system does not hold a slot for (no LocalTransform, or
 * no WorldTransform at all) use their WorldTransform as flecs holds it, or keep their stored view_projection.   */
typedef struct GpuCameraLayout {
    uint32_t kind_offset, options_offset, view_offset, projection_offset, view_projection_offset;
} GpuCameraLayout;
int GpuTransformCullSystem_setCameraPipeline(GpuTransformCullSystem* s, const GpuCameraLayout* layout /* NULL: off */, float aspect);
int GpuTransformCullSystem_setCameraAspect(GpuTransformCullSystem* s, float aspect);
int GpuTransformCullSystem_setMeshComponent(GpuTransformCullSystem* s, ecs_entity_t mesh, uint32_t bounds_offset,
                                            int32_t vertices_count_offset);
/* Material x mesh draw buckets (engine/GAL/vulkan/RenderApi.zig:1410-1441: the draw loop binds the
 * pipeline and its descriptor sets whenever `material.ref` differs from the previous object's, the vertex buffer
 * whenever the mesh does).  setMaterialComponent: the id of the Material component (`Material{ ref: *shading.Material }`,
 * engine/components/material.zig; after setMeshComponent).  Every distinct (material, mesh) pair among the entities
 * then gets its own entry of the device scene's mesh table - same AABB and vertex count for entries that share a mesh -
 * so yh_build_mesh_buckets groups a frustum's visible entities by pair; Material sets and structural changes are
 * followed like Mesh sets.  Up to 65535 pairs (+ setBounds overrides).
 * drawBuckets: builds the buckets of `frustum` on the device (object data and draw records in bucket order, record i =
 * {vertices_count, 1, 0, batch_offset + i}: yh_object_data_device) and returns the non-empty ones that have a mesh,
 * SORTED BY MATERIAL, THEN MESH (pointer order): walking them in that order binds every pipeline once and every vertex
 * buffer once per material; records [first, first + count) of a bucket are one vkCmdDrawIndirect.  entities (optional):
 * the entity behind every record, record order (all `written` of them: entities without a mesh are listed too, in the
 * records before the first bucket's).  YH_ERR_CAPACITY with *n_buckets / *written = what is needed.               */
typedef struct GpuDrawBucket {
    const void* material; /* shading.Material*, NULL for entities without a Material */
    const void* mesh;     /* meshes.Mesh* */
    uint32_t first, count;
} GpuDrawBucket;
int GpuTransformCullSystem_setMaterialComponent(GpuTransformCullSystem* s, ecs_entity_t material);
int GpuTransformCullSystem_drawBuckets(GpuTransformCullSystem* s, uint32_t frustum, uint32_t batch_offset, GpuDrawBucket* buckets,
                                       uint32_t cap_buckets, uint32_t* n_buckets, ecs_entity_t* entities, uint64_t cap,
                                       uint64_t* written);
int GpuTransformCullSystem_invalidateBounds(GpuTransformCullSystem* s);
ecs_entity_t GpuTransformCullSystem_frustumCamera(const GpuTransformCullSystem* s, uint32_t frustum);
uint32_t GpuTransformCullSystem_frusta(const GpuTransformCullSystem* s);
yh_scene* GpuTransformCullSystem_scene(GpuTransformCullSystem* s);

/* Visible entities of the last frame for one frustum, ascending slot order.  Returns the count; writes at most cap. */
uint64_t GpuTransformCullSystem_visible(GpuTransformCullSystem* s, uint32_t frustum, ecs_entity_t* out, uint64_t cap);

/* Last error code of the per-frame run (0 = ok): system callbacks cannot return one.  A negative status means the frame
 * wrote nothing back; the next frame walks the query and rebuilds from flecs' columns.  If results were being left on the
 * device (setReadback(s, 0)) and the failure struck while a structural change was being patched in, the rows computed on the
 * device since readback was switched off are given up (the half-rewritten slot table is not read). */
int GpuTransformCullSystem_status(const GpuTransformCullSystem* s);
uint64_t GpuTransformCullSystem_entities(const GpuTransformCullSystem* s);
uint32_t GpuTransformCullSystem_levels(const GpuTransformCullSystem* s); /* levels of the device scene (the spare ones included) */
uint64_t GpuTransformCullSystem_rebuilds(const GpuTransformCullSystem* s);

/* Incremental mode (on by default): only rows whose onSet fired are uploaded, only rows the frame touched are
 * downloaded.  Off: whole columns both ways every frame.  Switching costs one full upload. */
int GpuTransformCullSystem_setIncremental(GpuTransformCullSystem* s, int on);
/* Small structural changes without a rebuild (default on; incremental mode with readback only).  Entities created,
 * deleted or re-parented between frames (editor/Editor.zig:590-636, GameApp.zig:204-215) change a few tables; the
 * observers list them, and the next frame moves just their rows: every level's range of slots ends in spare slots
 * (an eighth of the level, at least 1024; dead on the device: yh_scene_set_alive_range), the rows of a table that changed
 * take spare slots of its level, its old slots die, children of moved rows learn the new parent slot
 * (yh_upload_parent_range); a subtree re-parented to another depth takes the batches below it along to the spare slots
 * of their new levels; the batches of tables flecs deleted (the children of a deleted parent) die without their
 * pointers being touched.  No walk of the query, no re-sort, no column upload.  What a patch cannot follow - a subtree
 * that would reach below the scene's deepest level, a level's spare slots used up, a quarter of the slots moving -
 * falls back to the walk and the rebuild, which also compacts the slots again.
 * `patches` counts the frames that took this path.                                                             */
int GpuTransformCullSystem_setPatching(GpuTransformCullSystem* s, int on);
uint64_t GpuTransformCullSystem_patches(const GpuTransformCullSystem* s);
/* Walk-free frames (on by default): while the query's cache has not (un)matched a table and every matched table
 * still holds the entities it held, the per-frame iteration of the query (~0.3 us per table in flecs) is skipped and
 * the last walk's tables are reused.  walkStats: iterations done / frames that skipped it. */
int GpuTransformCullSystem_setSkipWalks(GpuTransformCullSystem* s, int on);
void GpuTransformCullSystem_walkStats(const GpuTransformCullSystem* s, uint64_t* walks, uint64_t* skipped);
/* Structural-change tracking (on by default).  Observers on every id added to or removed from a table that holds one
 * of the two components (entities created, deleted, re-parented, moved between archetypes by an unrelated component)
 * and on such tables being created or deleted raise a flag; while it is clear a frame neither iterates the query nor
 * looks at any table (cleanFrames counts those): its cost is the rows that changed, not the tables that exist
 * (editor/Editor.zig:590-636 re-parenting, engine/GameApp.zig:204-215 scene load are what raises it in the engine).
 * Off: every frame compares every cached table's entities with the world's, as round 1 did (20 ms per frame for 300k
 * tables).  structureDirty / clearStructureDirty: the flag itself (tests). */
int GpuTransformCullSystem_setTrustObservers(GpuTransformCullSystem* s, int on);
int GpuTransformCullSystem_structureDirty(const GpuTransformCullSystem* s);
void GpuTransformCullSystem_clearStructureDirty(GpuTransformCullSystem* s);
uint64_t GpuTransformCullSystem_cleanFrames(const GpuTransformCullSystem* s);
/* Frames that had nothing to do at all - no structural change, no row set, the same root, bounds and frusta as the frame
 * before - and so launched nothing: the last frame's rows and lists stand (incremental mode). */
uint64_t GpuTransformCullSystem_idleFrames(const GpuTransformCullSystem* s);
/* Frames in which nothing moved but the frusta or the bounds changed (a camera moved): one launch culled the world
 * matrices the device holds (YH_FRAME_CULL_ONLY). */
uint64_t GpuTransformCullSystem_cullOnlyFrames(const GpuTransformCullSystem* s);
/* Readback (on by default): the rows a frame recomputed come back into the flecs columns, so every CPU reader of
 * WorldTransform.matrix keeps working (RenderApi.zig:1398-1402, SceneWindow.zig:334-351).  Off, for a renderer that
 * takes matrices and draw records on the device (yh_build_object_data / yh_object_data_device): nothing comes back, a
 * frame that moved every entity costs what the GPU frame costs; flecs' copies of recomputed rows go stale until
 * readback is switched on again (the next frame then brings whole columns back); structural changes are still patched:
 * rows the device holds change slots there (yh_move_rows), only new entities' rows are read from flecs.  fetchWorld: one entity's current
 * world matrix from the device, whatever the mode (the gizmo reads the selected entity's: SceneWindow.zig:334-351). */
int GpuTransformCullSystem_setReadback(GpuTransformCullSystem* s, int on);
int GpuTransformCullSystem_fetchWorld(GpuTransformCullSystem* s, ecs_entity_t entity, float matrix[16]);
/* Threads that share the host side of a frame (default: the CPUs online, at most 16; 1 = the calling thread only).
 * The per-frame jobs - edited rows gathered from their tables, changed rows placed back into them - go to n - 1 workers
 * that are started by the first such job and live until _destroy (or the next _setThreads): asleep on a condition variable
 * between frames, so they cost nothing while nothing moves.  The long jobs of a rebuilt frame (the second half of a
 * walk, whole columns, bounds) go to the same workers; only the fills of the hash maps make threads of their own for the
 * one call (one of them runs on the thread that builds patch()'s look-up tables in the background). */
int GpuTransformCullSystem_setThreads(GpuTransformCullSystem* s, int n);
/* 68-byte rows that crossed PCIe in the last frame, each direction (both components counted). */
/* Wall time of the last frame by stage, ms: [0] tables, root, bounds, frusta; [1] changed rows gathered and sent up;
 * [2] the device frame (yh_frame + wait); [3] changed rows brought down and placed in their tables; and, both inside
 * [0] and zero in a frame without structural change: [4] the walk of the query, [5] the rebuild of the slot order, or
 * the patch that replaced both. */
void GpuTransformCullSystem_lastTimes(const GpuTransformCullSystem* s, double ms[6]);
void GpuTransformCullSystem_lastTraffic(const GpuTransformCullSystem* s, uint64_t* rows_up, uint64_t* rows_down);

void GpuTransformCullSystem_destroy(GpuTransformCullSystem* s);

#ifdef __cplusplus
}
#endif
#endif
