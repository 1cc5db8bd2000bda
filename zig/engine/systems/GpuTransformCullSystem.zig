//! Drop-in for TransformSyncSystem.zig + PostTransformSystem.zig on an MI355X: same `registerTo(world) ecs.Entity`
//! contract, so the call site (editor/Editor.zig:153-162) keeps tagging the returned system entity.
//! The system body lives in C (yume_amd/host/GpuTransformCullSystem.c, compiled into the engine by build.zig next
//! to clibs.c); this file only hands it the component ids and keeps the state pointer.
//! NOT compiled in this repository's CI (no Zig toolchain in the build image).

const c = @import("clibs");
const std = @import("std");

const ecs = @import("../ecs.zig");
const hip = @import("../hip/yume_hip.zig");
const Mat4 = @import("../math3d.zig").Mat4;

const LocalTransform = ecs.components.LocalTransform;
const WorldTransform = ecs.components.WorldTransform;

pub const State = opaque {};

extern fn GpuTransformCullSystem_registerTo(
    world: ?*c.ecs_world_t,
    local_transform: c.ecs_entity_t,
    world_transform: c.ecs_entity_t,
    ctx: *hip.Ctx,
    out_state: *?*State,
) c.ecs_entity_t;
extern fn GpuTransformCullSystem_setBounds(s: *State, entity: c.ecs_entity_t, min_max: *const [6]f32) c_int;
extern fn GpuTransformCullSystem_setFrusta(s: *State, planes: [*]const f32, n_frusta: u32) c_int;
extern fn GpuTransformCullSystem_visible(s: *State, frustum: u32, out: ?[*]c.ecs_entity_t, cap: u64) u64;
extern fn GpuTransformCullSystem_setIncremental(s: *State, on: c_int) c_int;
extern fn GpuTransformCullSystem_lastTraffic(s: *const State, rows_up: ?*u64, rows_down: ?*u64) void;
extern fn GpuTransformCullSystem_status(s: *const State) c_int;
extern fn GpuTransformCullSystem_destroy(s: ?*State) void;

var gpu: ?*hip.Ctx = null;
var state: ?*State = null;

comptime {
    // the C side moves rows as 68 opaque bytes: {Mat4, bool dirty}, align 4 (components/transform.zig:8-12, 62-66)
    std.debug.assert(@sizeOf(WorldTransform) == 68 and @sizeOf(LocalTransform) == 68);
    std.debug.assert(@offsetOf(WorldTransform, "dirty") == 64 and @offsetOf(LocalTransform, "dirty") == 64);
}

/// Same shape as TransformSyncSystem.registerTo.  Do NOT also register PostTransformSystem: the dirty reset is part
/// of the GPU frame.  Returns the system entity; error.YumeHip if no MI355X context could be created.
pub fn registerTo(world: ecs.World) !ecs.Entity {
    if (gpu == null) try hip.check(hip.yh_init(0, &gpu));
    const sys = GpuTransformCullSystem_registerTo(
        world.inner,
        ecs.typeId(LocalTransform),
        ecs.typeId(WorldTransform),
        gpu.?,
        &state,
    );
    if (sys == 0) return error.YumeHip;
    return sys;
}

/// Mesh bounds of an entity (engine/meshes.zig:72-95 `Mesh.bounds`): call when a MeshRenderer is set.
pub fn setBounds(entity: ecs.Entity, min: [3]f32, max: [3]f32) !void {
    const mm = [6]f32{ min[0], min[1], min[2], max[0], max[1], max[2] };
    try hip.check(GpuTransformCullSystem_setBounds(state.?, entity, &mm));
}

/// Frusta of the coming frames, from each camera's view_projection (components/camera.zig:300-302).
pub fn setFrusta(view_projections: []const Mat4) !void {
    var planes: [hip.max_frusta][24]f32 = undefined;
    const n = @min(view_projections.len, hip.max_frusta);
    for (view_projections[0..n], 0..) |vp, i| hip.yh_extract_planes(&vp.values, &planes[i]);
    try hip.check(GpuTransformCullSystem_setFrusta(state.?, @ptrCast(&planes), @intCast(n)));
}

/// Entities inside frustum `frustum` after the last `world.progress`; the renderer's draw loop
/// (RenderApi.zig:1380-1434) iterates this list instead of every MeshRenderer.
pub fn visible(frustum: u32, out: []ecs.Entity) []ecs.Entity {
    const n = GpuTransformCullSystem_visible(state.?, frustum, out.ptr, out.len);
    return out[0..@min(n, out.len)];
}

pub fn status() c_int {
    return if (state) |s| GpuTransformCullSystem_status(s) else 0;
}

pub fn deinit() void {
    GpuTransformCullSystem_destroy(state);
    state = null;
    hip.yh_shutdown(gpu);
    gpu = null;
}
