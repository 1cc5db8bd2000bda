//! Zig binding of include/yume_hip.h: one `extern fn` per C symbol the engine-side system and the renderer need.
//! NOT compiled in this repository's CI (no Zig toolchain in the build image): it is the stub a maintainer of the
//! reference adds as engine/hip/yume_hip.zig; INTEGRATION.md lists the build.zig lines that go with it.
//! Struct layouts are checked on the C side by tests/test_abi.py (yh_scene_desc = 40 bytes, yh_timing = 24 bytes).

pub const Ctx = opaque {};
pub const Scene = opaque {};

pub const Status = enum(c_int) {
    ok = 0,
    bad_arg = -1,
    oom = -2,
    hip = -3,
    rccl = -4,
    capacity = -5,
    state = -6,
    unsupported = -7,
    _,
};

pub const max_frusta = 8;
pub const max_levels = 64;
pub const no_parent: u32 = 0xFFFF_FFFF;

pub const LocalKind = enum(u32) { trs = 0, mat4 = 1 };
pub const BoundsKind = enum(u32) { none = 0, per_entity = 1, mesh_table = 2 };

pub const SceneDesc = extern struct {
    n_entities: u64,
    n_levels: u32 = 1,
    n_frusta: u32 = 1,
    local_kind: u32 = @intFromEnum(LocalKind.mat4),
    bounds_kind: u32 = @intFromEnum(BoundsKind.per_entity),
    max_meshes: u32 = 0,
    id_base: u32 = 0,
    draw_capacity: u64 = 0,
};

pub const frame_all_dirty: u32 = 0x1;
pub const frame_no_cull: u32 = 0x2;
pub const frame_async: u32 = 0x4;

pub extern fn yh_abi_version() u32;
pub extern fn yh_init(device_id: c_int, out: *?*Ctx) c_int;
pub extern fn yh_shutdown(ctx: ?*Ctx) void;
pub extern fn yh_last_error(ctx: ?*const Ctx) [*:0]const u8;
pub extern fn yh_wait(ctx: *Ctx) c_int;

pub extern fn yh_scene_create(ctx: *Ctx, desc: *const SceneDesc, out: *?*Scene) c_int;
pub extern fn yh_scene_destroy(scene: ?*Scene) void;
pub extern fn yh_upload_hierarchy(s: *Scene, parent_index: [*]const u32, level_offsets: [*]const u32) c_int;
pub extern fn yh_set_root(s: *Scene, root_world: *const [16]f32) c_int;
pub extern fn yh_upload_root_aos68(s: *Scene, row: *const anyopaque) c_int;
pub extern fn yh_upload_local_aos68(s: *Scene, column: *const anyopaque, first: u64, count: u64) c_int;
pub extern fn yh_upload_world_aos68(s: *Scene, column: *const anyopaque, first: u64, count: u64) c_int;
pub extern fn yh_upload_local_rows_aos68(s: *Scene, slots: [*]const u32, rows: *const anyopaque, count: u64) c_int;
pub extern fn yh_upload_world_rows_aos68(s: *Scene, slots: [*]const u32, rows: *const anyopaque, count: u64) c_int;
pub extern fn yh_scene_track_changes(s: *Scene, enable: c_int) c_int;
pub extern fn yh_get_changed_count(s: *Scene, n_changed: *u64) c_int;
pub extern fn yh_download_changed_aos68(s: *Scene, slots: ?[*]u32, world_rows: ?*anyopaque, local_rows: ?*anyopaque, cap: u64, written: *u64) c_int;
pub extern fn yh_mark_dirty(s: *Scene, local: c_int, world: c_int, first: u64, count: u64) c_int;
pub extern fn yh_upload_aabb(s: *Scene, min_planes: *const [3][*]const f32, max_planes: *const [3][*]const f32, first: u64, count: u64) c_int;
pub extern fn yh_frame(s: *Scene, planes: ?[*]const f32, n_frusta: u32, flags: u32) c_int;
/// yh_frame for hosts that may not allocate around the frame (no counterpart in the reference): same results,
/// stage-3 scratch in the library's static device array (16 MiB per GPU), `capacity` beyond ~119M entity-frusta.
pub const static_scratch_bytes: u32 = 16 << 20;
pub extern fn yh_frame_static(s: *Scene, planes: ?[*]const f32, n_frusta: u32, flags: u32) c_int;
pub extern fn yh_get_counts(s: *Scene, counts: [*]u32, boundary_counts: ?[*]u32) c_int;
pub extern fn yh_get_draw_list(s: *Scene, frustum: u32, out: [*]u32, cap: u64, written: *u64) c_int;
pub extern fn yh_download_world_aos68(s: *Scene, column: *anyopaque, first: u64, count: u64) c_int;
pub extern fn yh_download_local_aos68(s: *Scene, column: *anyopaque, first: u64, count: u64) c_int;
pub extern fn yh_camera_view_projection(orthographic: c_int, pos: *const [3]f32, rot: *const [3]f32, aspect: f32, fovy_rad: f32, near_z: f32, far_z: f32, out: *[16]f32) void;
pub extern fn yh_upload_vertex_counts(s: *Scene, counts: [*]const u32, n: u64) c_int;
pub extern fn yh_build_object_data(s: *Scene, frustum: u32, batch_offset: u32) c_int;
pub extern fn yh_object_data_device(s: *Scene, frustum: u32, cmds_device: ?*?*anyopaque) ?*anyopaque;
pub const Bucket = extern struct { mesh: u32, first: u32, count: u32 };
pub extern fn yh_build_mesh_buckets(s: *Scene, frustum: u32, batch_offset: u32) c_int;
pub extern fn yh_get_mesh_buckets(s: *Scene, frustum: u32, buckets: ?[*]Bucket, cap_buckets: u32, entities: ?[*]u32, cap: u64, written: *u64) c_int;
pub extern fn yh_decompose(s: *Scene, which: c_int, planes: *const [10][*]f32, valid: ?[*]u8, first: u64, count: u64) c_int;
pub extern fn yh_extract_planes(view_projection: *const [16]f32, planes: *[24]f32) void;

pub fn check(rc: c_int) !void {
    if (rc != 0) return error.YumeHip;
}
