//! Drop-in for TransformSyncSystem.zig + PostTransformSystem.zig on an MI355X: the same
//! `pub fn registerTo(world: ecs.World) ecs.Entity` (engine/systems/TransformSyncSystem.zig:12), so the call site
//! (editor/Editor.zig:153-162, see zig/yume.patch) keeps tagging the returned system entity.
//! The system body lives in C (yume_amd/host/GpuTransformCullSystem.c, compiled into the engine by build.zig next
//! to the other C shims); this file hands it the component ids and layouts and keeps the state pointer.
//! NOT compiled in this repository's CI (no Zig toolchain in the build image).

const c = @import("clibs");
const std = @import("std");

const ecs = @import("../ecs.zig");
const hip = @import("../hip/yume_hip.zig");
const meshes = @import("../meshes.zig");
const Mat4 = @import("../math3d.zig").Mat4;

const LocalTransform = ecs.components.LocalTransform;
const WorldTransform = ecs.components.WorldTransform;
const Camera = ecs.components.Camera;
const Mesh = ecs.components.Mesh;
const Material = ecs.components.Material;

pub const State = opaque {};

extern fn GpuTransformCullSystem_registerTo(
    world: ?*c.ecs_world_t,
    local_transform: c.ecs_entity_t,
    world_transform: c.ecs_entity_t,
    ctx: *hip.Ctx,
    out_state: *?*State,
) c.ecs_entity_t;
extern fn GpuTransformCullSystem_setCameraComponent(s: *State, camera: c.ecs_entity_t, view_projection_offset: u32) c_int;
pub const CameraLayout = extern struct { kind_offset: u32, options_offset: u32, view_offset: u32, projection_offset: u32, view_projection_offset: u32 };
extern fn GpuTransformCullSystem_setCameraPipeline(s: *State, layout: ?*const CameraLayout, aspect: f32) c_int;
extern fn GpuTransformCullSystem_setCameraAspect(s: *State, aspect: f32) c_int;
extern fn GpuTransformCullSystem_setMeshComponent(s: *State, mesh: c.ecs_entity_t, bounds_offset: u32, vertices_count_offset: i32) c_int;
pub const DrawBucket = extern struct { material: ?*const anyopaque, mesh: ?*const anyopaque, first: u32, count: u32 };
extern fn GpuTransformCullSystem_setMaterialComponent(s: *State, material: c.ecs_entity_t) c_int;
extern fn GpuTransformCullSystem_drawBuckets(s: *State, frustum: u32, batch_offset: u32, buckets: ?[*]DrawBucket, cap_buckets: u32, n_buckets: *u32, entities: ?[*]c.ecs_entity_t, cap: u64, written: *u64) c_int;
extern fn GpuTransformCullSystem_invalidateBounds(s: *State) c_int;
extern fn GpuTransformCullSystem_frustumCamera(s: *const State, frustum: u32) c.ecs_entity_t;
extern fn GpuTransformCullSystem_frusta(s: *const State) u32;
extern fn GpuTransformCullSystem_scene(s: *State) ?*hip.Scene;
extern fn GpuTransformCullSystem_setBounds(s: *State, entity: c.ecs_entity_t, min_max: *const [6]f32) c_int;
extern fn GpuTransformCullSystem_setFrusta(s: *State, planes: ?[*]const f32, n_frusta: u32) c_int;
extern fn GpuTransformCullSystem_visible(s: *State, frustum: u32, out: ?[*]c.ecs_entity_t, cap: u64) u64;
extern fn GpuTransformCullSystem_status(s: *const State) c_int;
extern fn GpuTransformCullSystem_entities(s: *const State) u64;
extern fn GpuTransformCullSystem_levels(s: *const State) u32;
extern fn GpuTransformCullSystem_rebuilds(s: *const State) u64;
extern fn GpuTransformCullSystem_setIncremental(s: *State, on: c_int) c_int;
extern fn GpuTransformCullSystem_setPatching(s: *State, on: c_int) c_int;
extern fn GpuTransformCullSystem_patches(s: *const State) u64;
extern fn GpuTransformCullSystem_setSkipWalks(s: *State, on: c_int) c_int;
extern fn GpuTransformCullSystem_setTrustObservers(s: *State, on: c_int) c_int;
extern fn GpuTransformCullSystem_setReadback(s: *State, on: c_int) c_int;
extern fn GpuTransformCullSystem_fetchWorld(s: *State, entity: c.ecs_entity_t, matrix: *[16]f32) c_int;
extern fn GpuTransformCullSystem_setThreads(s: *State, n: c_int) c_int;
extern fn GpuTransformCullSystem_cleanFrames(s: *const State) u64;
extern fn GpuTransformCullSystem_idleFrames(s: *const State) u64;
extern fn GpuTransformCullSystem_cullOnlyFrames(s: *const State) u64;
extern fn GpuTransformCullSystem_walkStats(s: *const State, walks: ?*u64, skipped: ?*u64) void;
extern fn GpuTransformCullSystem_lastTimes(s: *const State, ms: *[6]f64) void;
extern fn GpuTransformCullSystem_lastTraffic(s: *const State, rows_up: ?*u64, rows_down: ?*u64) void;
extern fn GpuTransformCullSystem_destroy(s: ?*State) void;

var gpu: ?*hip.Ctx = null;
var state: ?*State = null;
/// Why the last registerTo / frame failed (0 = ok, otherwise a hip.Status value); `registerTo` cannot return an error
/// without changing the reference's signature, system callbacks cannot return one at all.
var last_status: c_int = 0;

comptime {
    // the C side moves rows as 68 opaque bytes: {Mat4, bool dirty}, align 4 (components/transform.zig:8-12, 62-66)
    std.debug.assert(@sizeOf(WorldTransform) == 68 and @sizeOf(LocalTransform) == 68);
    std.debug.assert(@offsetOf(WorldTransform, "dirty") == 64 and @offsetOf(LocalTransform, "dirty") == 64);
    // Mesh{ ref: *meshes.Mesh } (components/mesh.zig) is one pointer; BoundingBox is {mins, maxs: Vec3} (meshes.zig:72-75)
    std.debug.assert(@sizeOf(Mesh) == @sizeOf(*meshes.Mesh) and @sizeOf(meshes.BoundingBox) == 24);
}

/// Same signature and contract as TransformSyncSystem.registerTo (TransformSyncSystem.zig:12): registers the system
/// in phase PostUpdate over the same three query terms and returns its entity.  Do NOT also register
/// PostTransformSystem: the dirty reset is part of the GPU frame.  If no MI355X context can be created the returned
/// entity is 0 and `status()` says why (the reference's own start-up failures @panic, GameApp.zig:57-72; a caller that
/// wants that writes `if (tss == 0) @panic(...)`).
pub fn registerTo(world: ecs.World) ecs.Entity {
    if (gpu == null) {
        last_status = hip.yh_init(0, &gpu);
        if (last_status != 0) return 0;
    }
    const sys = GpuTransformCullSystem_registerTo(
        world.inner,
        ecs.typeId(LocalTransform),
        ecs.typeId(WorldTransform),
        gpu.?,
        &state,
    );
    if (sys == 0) {
        last_status = @intFromEnum(hip.Status.bad_arg);
        return 0;
    }
    // the system takes its cull inputs from the ECS itself: one frustum per Camera entity (camera.zig:300-302), one
    // AABB per distinct mesh a Mesh component refers to (meshes.zig:72-95, 104-110)
    last_status = GpuTransformCullSystem_setCameraComponent(state.?, ecs.typeId(Camera), @offsetOf(Camera, "view_projection"));
    if (last_status == 0)
        last_status = GpuTransformCullSystem_setMeshComponent(
            state.?,
            ecs.typeId(Mesh),
            @offsetOf(meshes.Mesh, "bounds"),
            @offsetOf(meshes.Mesh, "vertices_count"),
        );
    // ... and one draw bucket per (material, mesh) pair (the draw loop's rebinding rule, RenderApi.zig:1411, 1434)
    if (last_status == 0) last_status = GpuTransformCullSystem_setMaterialComponent(state.?, ecs.typeId(Material));
    return sys;
}

/// The camera matrix pipeline: every frame derives each Camera entity's view / projection / view_projection from the
/// WorldTransform that frame is about to write - what GameWindow.zig:160-161 does at draw time
/// (`transform.decompose()` -> `rotation.toEuler()` -> `camera.updateMatrices(pos, euler, aspect)`) - writes them into
/// the component and culls against THIS frame's frusta.  `aspect` is GameWindow.zig:155's game_view_size.x / .y; call
/// again (or `setCameraAspect`) when the game view is resized.  Off (the default): frusta come from view_projection as
/// the last draw left it, one frame late for a camera that moved.
pub fn setCameraPipeline(on: bool, aspect: f32) void {
    const CameraOptions = @TypeOf(@as(Camera, undefined).opts);
    const layout = CameraLayout{
        .kind_offset = @offsetOf(Camera, "opts") + @offsetOf(CameraOptions, "kind"),
        .options_offset = @offsetOf(Camera, "opts") + @offsetOf(CameraOptions, "data"),
        .view_offset = @offsetOf(Camera, "view"),
        .projection_offset = @offsetOf(Camera, "projection"),
        .view_projection_offset = @offsetOf(Camera, "view_projection"),
    };
    last_status = GpuTransformCullSystem_setCameraPipeline(state.?, if (on) &layout else null, aspect);
}
pub fn setCameraAspect(aspect: f32) void {
    last_status = GpuTransformCullSystem_setCameraAspect(state.?, aspect);
}

/// Override: bounds of one entity that are not its mesh's (engine/meshes.zig:72-95 `Mesh.bounds` is the default).
pub fn setBounds(entity: ecs.Entity, min: [3]f32, max: [3]f32) void {
    const mm = [6]f32{ min[0], min[1], min[2], max[0], max[1], max[2] };
    last_status = GpuTransformCullSystem_setBounds(state.?, entity, &mm);
}

/// Override: frusta of the coming frames instead of the Camera entities' (e.g. the editor camera, which is not an
/// entity: editor/windows/SceneWindow.zig).  An empty slice returns to the cameras.
pub fn setFrusta(view_projections: []const Mat4) void {
    var planes: [hip.max_frusta][24]f32 = undefined;
    const n = @min(view_projections.len, hip.max_frusta);
    for (view_projections[0..n], 0..) |vp, i| hip.yh_extract_planes(&vp.values, &planes[i]);
    last_status = GpuTransformCullSystem_setFrusta(state.?, if (n == 0) null else @ptrCast(&planes), @intCast(n));
}

/// A mesh asset was reloaded in place (assets.zig reload hooks): read every AABB again at the next frame.
pub fn invalidateBounds() void {
    _ = GpuTransformCullSystem_invalidateBounds(state.?);
}

/// Frusta the last frame culled against, and the Camera entity behind frustum `f` (0 for an override frustum):
/// GameWindow.zig:154-178 draws once per camera and looks its list up by this.
pub fn frusta() u32 {
    return if (state) |s| GpuTransformCullSystem_frusta(s) else 0;
}
pub fn frustumCamera(f: u32) ecs.Entity {
    return GpuTransformCullSystem_frustumCamera(state.?, f);
}

/// Entities inside frustum `frustum` after the last `world.progress`; the renderer's draw loop
/// (RenderApi.zig:1380-1434) iterates this list instead of every MeshRenderer.
pub fn visible(frustum: u32, out: []ecs.Entity) []ecs.Entity {
    const n = GpuTransformCullSystem_visible(state.?, frustum, out.ptr, out.len);
    return out[0..@min(n, out.len)];
}

/// Frustum `frustum`'s visible entities as draw buckets, one per (material, mesh) pair, sorted by material and then
/// mesh: the draw loop (RenderApi.zig:1410-1441) binds `material.ref.pipeline` once per run of equal materials, the
/// vertex buffer once per bucket, and issues one vkCmdDrawIndirect over records [first, first + count) of the
/// device-side draw records (hip.yh_object_data_device) - instead of one vkCmdDraw per object.
pub fn drawBuckets(frustum: u32, batch_offset: u32, out: []DrawBucket) []DrawBucket {
    var n: u32 = 0;
    var records: u64 = 0;
    last_status = GpuTransformCullSystem_drawBuckets(state.?, frustum, batch_offset, out.ptr, @intCast(out.len), &n, null, 0, &records);
    return out[0..@min(n, out.len)];
}

/// The device scene (draw-list consumer: hip.yh_build_object_data / yh_build_mesh_buckets / yh_object_data_device
/// fill GPUObjectData and draw records on the device, RenderApi.zig:1397-1441).
pub fn scene() ?*hip.Scene {
    return if (state) |s| GpuTransformCullSystem_scene(s) else null;
}

/// Results stay on the device (a renderer that imports the device buffers); `fetchWorld` reads one entity's matrix
/// (the gizmo's selected entity, SceneWindow.zig:334-351).
pub fn setReadback(on: bool) void {
    _ = GpuTransformCullSystem_setReadback(state.?, @intFromBool(on));
}
pub fn fetchWorld(entity: ecs.Entity) ?Mat4 {
    var m: Mat4 = undefined;
    if (GpuTransformCullSystem_fetchWorld(state.?, entity, &m.values) != 0) return null;
    return m;
}

/// Wall time of the last frame by stage, ms: inputs, rows up, device frame, rows down and placed; query walk and
/// slot-order rebuild (parts of the first, zero without a structural change).
pub fn lastTimes() [6]f64 {
    var ms = [6]f64{ 0, 0, 0, 0, 0, 0 };
    if (state) |s| GpuTransformCullSystem_lastTimes(s, &ms);
    return ms;
}

/// Frames whose structural change (entities created, deleted, re-parented) was followed by moving the rows of the
/// tables that changed to spare slots, without a walk of the query or a rebuild; `setPatching(false)` rebuilds always.
pub fn patches() u64 {
    return if (state) |s| GpuTransformCullSystem_patches(s) else 0;
}
pub fn setPatching(on: bool) void {
    _ = GpuTransformCullSystem_setPatching(state.?, @intFromBool(on));
}
/// Frames that had nothing to do (no structural change, no row set, the same root, bounds and frusta as the frame before)
/// and launched nothing: the last frame's rows and lists stand.
pub fn idleFrames() u64 {
    return if (state) |s| GpuTransformCullSystem_idleFrames(s) else 0;
}
/// Frames in which only the frusta or the bounds changed: one launch culled the matrices the device holds.
pub fn cullOnlyFrames() u64 {
    return if (state) |s| GpuTransformCullSystem_cullOnlyFrames(s) else 0;
}

/// Status of the last frame (system callbacks cannot return one), or of the last call above that failed.
pub fn status() c_int {
    if (state) |s| {
        const frame = GpuTransformCullSystem_status(s);
        if (frame != 0) return frame;
    }
    return last_status;
}

pub fn deinit() void {
    GpuTransformCullSystem_destroy(state);
    state = null;
    hip.yh_shutdown(gpu);
    gpu = null;
}
