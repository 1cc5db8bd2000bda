"""ctypes binding of the C ABI in include/yume_hip.h (libyume_hip.so, built in-tree for gfx950).

The product path has no CPU fallback: if the shared library is missing this module raises, and any
compute call without a GPU fails with the library's own error.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# YUME_HIP_LIB: development knob for A/B runs of two builds of the same library; never a different implementation
LIB_PATH = os.environ.get("YUME_HIP_LIB") or os.path.join(_HERE, "libyume_hip.so")

YH_MAX_FRUSTA = 8
YH_MAX_LEVELS = 64
YH_NO_PARENT = 0xFFFFFFFF
YH_UNIQUE_ID_BYTES = 128
YH_STATIC_SCRATCH_BYTES = 16 << 20

YH_OK = 0
YH_ERR_BAD_ARG = -1
YH_ERR_OOM = -2
YH_ERR_HIP = -3
YH_ERR_RCCL = -4
YH_ERR_CAPACITY = -5
YH_ERR_STATE = -6
YH_ERR_UNSUPPORTED = -7

YH_LOCAL_TRS, YH_LOCAL_MAT4 = 0, 1
YH_BOUNDS_NONE, YH_BOUNDS_PER_ENTITY, YH_BOUNDS_MESH_TABLE = 0, 1, 2

YH_FRAME_ALL_DIRTY = 0x1
YH_FRAME_NO_CULL = 0x2
YH_FRAME_ASYNC = 0x4
YH_FRAME_ORDERED = 0x8
YH_FRAME_LISTS_ON_DEMAND = 0x10
YH_FRAME_CULL_ONLY = 0x20
YH_TRACK_ROWS = 1
YH_TRACK_REWRITTEN_LOCALS = 2


class SceneDesc(C.Structure):
    _fields_ = [
        ("n_entities", C.c_uint64),
        ("n_levels", C.c_uint32),
        ("n_frusta", C.c_uint32),
        ("local_kind", C.c_uint32),
        ("bounds_kind", C.c_uint32),
        ("max_meshes", C.c_uint32),
        ("id_base", C.c_uint32),
        ("draw_capacity", C.c_uint64),
    ]


class Timing(C.Structure):
    _fields_ = [
        ("last_frame_ms", C.c_float),
        ("last_kernel_ms", C.c_float),
        ("n_launches", C.c_uint32),
        ("n_groups", C.c_uint32),
        ("frames", C.c_uint64),
    ]


class DeviceInfo(C.Structure):
    _fields_ = [
        ("device_id", C.c_int),
        ("n_cus", C.c_int),
        ("wavefront_size", C.c_int),
        ("hbm_bytes", C.c_uint64),
        ("arch", C.c_char * 32),
        ("name", C.c_char * 128),
    ]


class SynthDesc(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("index_base", C.c_uint64)]


_vp = C.c_void_p
_u64 = C.c_uint64
_u32 = C.c_uint32
_int = C.c_int

# name -> (restype, argtypes); every function include/yume_hip.h declares
ROWS_READY_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64)  # yh_rows_ready_fn

SIGNATURES = {
    "yh_abi_version": (_u32, []),
    "yh_init": (_int, [_int, C.POINTER(_vp)]),
    "yh_shutdown": (None, [_vp]),
    "yh_last_error": (C.c_char_p, [_vp]),
    "yh_device_info_get": (_int, [_vp, C.POINTER(DeviceInfo)]),
    "yh_wait": (_int, [_vp]),
    "yh_stream": (_vp, [_vp]),
    "yh_host_alloc": (_vp, [_vp, _u64]),
    "yh_host_free": (None, [_vp, _vp]),
    "yh_scene_create": (_int, [_vp, C.POINTER(SceneDesc), C.POINTER(_vp)]),
    "yh_scene_destroy": (None, [_vp]),
    "yh_upload_hierarchy": (_int, [_vp, _vp, _vp]),
    "yh_set_root": (_int, [_vp, _vp]),
    "yh_upload_root_aos68": (_int, [_vp, _vp]),
    "yh_upload_trs": (_int, [_vp, _vp, _u64, _u64]),
    "yh_upload_local_aos68": (_int, [_vp, _vp, _u64, _u64]),
    "yh_upload_world_aos68": (_int, [_vp, _vp, _u64, _u64]),
    "yh_mark_dirty": (_int, [_vp, _int, _int, _u64, _u64]),
    "yh_upload_aabb": (_int, [_vp, _vp, _vp, _u64, _u64]),
    "yh_upload_mesh_aabbs": (_int, [_vp, _vp, _u32]),
    "yh_upload_mesh_ids": (_int, [_vp, _vp, _u64, _u64]),
    "yh_frame": (_int, [_vp, _vp, _u32, _u32]),
    "yh_frame_static": (_int, [_vp, _vp, _u32, _u32]),
    "yh_get_counts": (_int, [_vp, _vp, _vp]),
    "yh_get_draw_list": (_int, [_vp, _u32, _vp, _u64, C.POINTER(_u64)]),
    "yh_download_world_soa": (_int, [_vp, _vp, _u64, _u64]),
    "yh_download_world_aos68": (_int, [_vp, _vp, _u64, _u64]),
    "yh_download_local_aos68": (_int, [_vp, _vp, _u64, _u64]),
    "yh_compose": (_int, [_vp, _vp, _u64, _u64]),
    "yh_timings": (_int, [_vp, C.POINTER(Timing)]),
    "yh_scene_set_kernel_timing": (_int, [_vp, _int]),
    "yh_scene_set_kernel_timing_period": (_int, [_vp, _u32]),
    "yh_upload_local_rows_aos68": (_int, [_vp, _vp, _vp, _u64]),
    "yh_upload_world_rows_aos68": (_int, [_vp, _vp, _vp, _u64]),
    "yh_move_rows": (_int, [_vp, _vp, _vp, _u64]),
    "yh_scene_track_changes": (_int, [_vp, _int]),
    "yh_get_changed_count": (_int, [_vp, C.POINTER(_u64)]),
    "yh_download_changed_aos68": (_int, [_vp, _vp, _vp, _vp, _u64, C.POINTER(_u64)]),
    "yh_upload_parent_range": (_int, [_vp, _vp, _u64, _u64]),
    "yh_scene_set_alive_range": (_int, [_vp, _u64, _u64, _int]),
    "yh_scene_set_alive_runs": (_int, [_vp, _vp, _vp, _vp, _u32]),
    "yh_upload_parent_rows": (_int, [_vp, _vp, _vp, _u64]),
    "yh_upload_mesh_id_rows": (_int, [_vp, _vp, _vp, _u64]),
    "yh_upload_aabb_rows": (_int, [_vp, _vp, _vp, _u64]),
    "yh_get_changed_counts": (_int, [_vp, C.POINTER(_u64), C.POINTER(_u64)]),
    "yh_download_changed_rows_aos68": (_int, [_vp, _int, _vp, _vp, _u64, C.POINTER(_u64), _vp, _vp]),
    "yh_upload_vertex_counts": (_int, [_vp, _vp, _u64]),
    "yh_build_object_data": (_int, [_vp, _u32, _u32]),
    "yh_get_object_data": (_int, [_vp, _u32, _vp, _vp, _u64, C.POINTER(_u64)]),
    "yh_object_data_device": (_vp, [_vp, _u32, C.POINTER(_vp)]),
    "yh_build_mesh_buckets": (_int, [_vp, _u32, _u32]),
    "yh_get_mesh_buckets": (_int, [_vp, _u32, _vp, _u32, _vp, _u64, C.POINTER(_u64)]),
    "yh_allgather_draw_lists_async": (_int, [_vp]),
    "yh_gather_selftest": (_int, [_vp, _u32]),
    "yh_gather_selftest_shards": (_int, [_vp, _u32]),
    "yh_get_gathered_counts": (_int, [_vp, _vp]),
    "yh_decompose": (_int, [_vp, _int, _vp, _vp, _u64, _u64]),
    "yh_scene_set_plan": (_int, [_vp, _u32, _vp]),
    "yh_scene_set_lds_parents": (_int, [_vp, _int]),
    "yh_scene_get_groups": (_int, [_vp, _vp, _u32, C.POINTER(_u32)]),
    "yh_scene_get_group_timings": (_int, [_vp, _vp, _vp, _u32]),
    "yh_scene_get_group_kernel_names": (_int, [_vp, _vp, _u64, C.POINTER(_u32)]),
    "yh_scene_set_grouping": (_int, [_vp, _u32]),
    "yh_stream_probe": (_int, [_vp, _u64, _u64, _u32, C.POINTER(C.c_float)]),
    "yh_extract_planes": (None, [_vp, _vp]),
    "yh_mat4_mul": (None, [_vp, _vp, _vp]),
    "yh_perspective": (None, [C.c_float, C.c_float, C.c_float, C.c_float, _vp]),
    "yh_orthographic": (None, [C.c_float] * 6 + [_vp]),
    "yh_look_at": (None, [_vp, _vp, _vp, _vp]),
    "yh_camera_view_projection": (None, [_int, _vp, _vp, C.c_float, C.c_float, C.c_float, C.c_float, _vp]),
    "yh_camera_update_matrices": (None, [_int, _vp, _vp, C.c_float, C.c_float, C.c_float, C.c_float, _vp, _vp, _vp]),
    "yh_mat4_decompose": (_int, [_vp, _vp, _vp, _vp]),
    "yh_quat_to_euler": (None, [_vp, _vp]),
    "yh_camera_update_from_world": (_int, [_int, _vp, C.c_float, C.c_float, C.c_float, C.c_float, _vp, _vp, _vp]),
    "yh_scene_fill_synthetic": (_int, [_vp, C.POINTER(SynthDesc)]),
    "yh_scene_fill_synthetic_shard": (_int, [_vp, C.POINTER(SynthDesc), _vp]),
    "yh_download_trs": (_int, [_vp, _vp, _u64, _u64]),
    "yh_download_aabb": (_int, [_vp, _vp, _vp, _u64, _u64]),
    "yh_partition_subtrees": (_int, [_vp, _vp, _u32, _u32, _vp, _vp]),
    "yh_shard_extract": (_int, [_vp, _vp, _u32, _vp, _u32, _vp, _vp, _vp]),
    "yh_comm_unique_id": (_int, [_vp]),
    "yh_comm_init": (_int, [_vp, _int, _int, _vp]),
    "yh_comm_destroy": (None, [_vp]),
    "yh_comm_describe": (_int, [C.c_char_p, _u64]),
    "yh_allgather_draw_lists": (_int, [_vp, _vp]),
    "yh_get_gathered_draw_list": (_int, [_vp, _u32, _vp, _u64, C.POINTER(_u64)]),
}

_LIB = None


class YumeHipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"[yh_status {code}] {message}")
        self.code = code


def lib():
    """Load libyume_hip.so (once).  Raises if it has not been built: there is no fallback path."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing. Build it with `make -C yume_amd/csrc` or "
                "`python -c 'import __graft_entry__ as g; g.build()'`; yume_amd has no CPU fallback."
            )
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(rc, ctx_handle=None):
    if rc != YH_OK:
        msg = lib().yh_last_error(ctx_handle)
        raise YumeHipError(rc, msg.decode("utf-8", "replace") if msg else "")
    return rc
