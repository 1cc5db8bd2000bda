"""ctypes front end of the CPU ORACLE (test infrastructure, NOT the product).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import this module.
It loads ``oracle/libyume_oracle.so`` (built by ``make -C oracle`` / ``__graft_entry__.build()``), the
scalar C restatement of engine/math3d.zig + engine/systems/TransformSyncSystem.zig.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

T68 = np.dtype({"names": ["m", "dirty", "pad"], "formats": [("<f4", 16), "u1", ("u1", 3)], "itemsize": 68})
SEED = 0x59554D45


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libyume_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        L = C.CDLL(path)
        L.yo_synth_raw.restype = C.c_uint64
        L.yo_synth_raw.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64]
        L.yo_frame_aos68.restype = C.c_size_t
        L.yo_inverse.restype = C.c_int
        L.yo_decompose.restype = C.c_int
        L.yo_camera_update_from_world.restype = C.c_int
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---- math3d -------------------------------------------------------------------------------------------
def mul(a, b):
    a, b = _f32(a).reshape(16), _f32(b).reshape(16)
    out = np.empty(16, np.float32)
    lib().yo_mul(_p(a), _p(b), _p(out))
    return out


def compose(t, q, s):
    t, q, s = _f32(t), _f32(q), _f32(s)
    out = np.empty(16, np.float32)
    lib().yo_compose(_p(t), _p(q), _p(s), _p(out))
    return out


def from_quat(q):
    q = _f32(q)
    out = np.empty(16, np.float32)
    lib().yo_from_quat(_p(q), _p(out))
    return out


def inverse(m):
    m = _f32(m).reshape(16)
    out = np.empty(16, np.float32)
    ok = lib().yo_inverse(_p(m), _p(out))
    return out if ok else None


def perspective(fovy, aspect, near, far):
    out = np.empty(16, np.float32)
    lib().yo_perspective(C.c_float(fovy), C.c_float(aspect), C.c_float(near), C.c_float(far), _p(out))
    return out


def orthographic(l, r, b, t, n, f):
    out = np.empty(16, np.float32)
    lib().yo_orthographic(*(C.c_float(x) for x in (l, r, b, t, n, f)), _p(out))
    return out


def look_at(eye, at, up):
    eye, at, up = _f32(eye), _f32(at), _f32(up)
    out = np.empty(16, np.float32)
    lib().yo_look_at(_p(eye), _p(at), _p(up), _p(out))
    return out


def rotation(axis, angle_rad):
    axis = _f32(axis)
    out = np.empty(16, np.float32)
    lib().yo_rotation(_p(axis), C.c_float(angle_rad), _p(out))
    return out


def camera_view_projection(orthographic, pos, rot, aspect=2.0, fovy_rad=1.3089969, near=0.1, far=200.0):
    pos, rot = _f32(pos), _f32(rot)
    out = np.empty(16, np.float32)
    lib().yo_camera_view_projection(int(orthographic), _p(pos), _p(rot), C.c_float(aspect), C.c_float(fovy_rad),
                                    C.c_float(near), C.c_float(far), _p(out))
    return out


def quat_to_euler(q):
    q = _f32(q).reshape(4)
    out = np.empty(3, np.float32)
    lib().yo_quat_to_euler(_p(q), _p(out))
    return out


def decompose(m):
    """Mat4.decompose of one matrix: (t, q xyzw, s, ok); the outputs are undefined where ok is False."""
    m = _f32(m).reshape(16)
    t, q, s = np.zeros(3, np.float32), np.zeros(4, np.float32), np.zeros(3, np.float32)
    ok = lib().yo_decompose(_p(m), _p(t), _p(q), _p(s))
    return t, q, s, bool(ok)


def camera_update_from_world(orthographic, world, aspect=2.0, fovy_rad=1.3089969, near=0.1, far=200.0):
    """GameWindow.zig:160-161: (view, projection, view_projection, decompose succeeded)."""
    world = _f32(world).reshape(16)
    view, proj, vp = (np.empty(16, np.float32) for _ in range(3))
    ok = lib().yo_camera_update_from_world(int(orthographic), _p(world), C.c_float(aspect), C.c_float(fovy_rad),
                                           C.c_float(near), C.c_float(far), _p(view), _p(proj), _p(vp))
    return view, proj, vp, bool(ok)


def extract_planes(vp):
    vp = _f32(vp).reshape(16)
    out = np.empty(24, np.float32)
    lib().yo_extract_planes(_p(vp), _p(out))
    return out


# ---- systems ------------------------------------------------------------------------------------------
def identity68(dirty=1):
    r = np.zeros(1, T68)
    r["m"][0, [0, 5, 10, 15]] = 1.0
    r["dirty"] = dirty
    return r


def compose_batch(trs):
    """trs: (10, n) f32 -> LocalTransform column (n,) T68, dirty = 1."""
    trs = _f32(trs)
    n = trs.shape[1]
    out = np.zeros(n, T68)
    lib().yo_compose_batch(_p(trs), C.c_size_t(n), C.c_size_t(0), C.c_size_t(n), _p(out))
    return out


def decompose_batch(column):
    """(n,) T68 -> ((10, n) f32 planes t.xyz q.xyzw s.xyz, (n,) u8 valid): decompose() catch IDENTITY per row."""
    assert column.dtype == T68
    n = column.shape[0]
    trs = np.zeros((10, n), np.float32)
    valid = np.zeros(n, np.uint8)
    lib().yo_decompose_batch(_p(np.ascontiguousarray(column)), C.c_size_t(n), _p(trs), C.c_size_t(n), _p(valid))
    return trs, valid


def transform_sync_batch(world, local, parent):
    """TransformSyncSystem.zig:53-74 on one table batch (in place); parent: (1,) T68."""
    assert world.dtype == T68 and local.dtype == T68 and parent.dtype == T68
    lib().yo_transform_sync_batch(_p(world), _p(local), _p(parent), C.c_size_t(world.shape[0]))


def frame_aos68(world, local, parent, level_offsets, root, n_threads=1):
    """One world.progress of TransformSync + PostTransform, in place.  Returns the batch count."""
    assert world.dtype == T68 and local.dtype == T68 and root.dtype == T68
    parent = np.ascontiguousarray(parent, np.uint32)
    lo = np.ascontiguousarray(level_offsets, np.uint32)
    return lib().yo_frame_aos68(_p(world), _p(local), _p(parent), _p(lo), C.c_uint32(len(lo) - 1), _p(root),
                                C.c_int(n_threads))


def cull(world, bounds, planes, mesh_id=None, n_threads=1):
    """world: (n,) T68; bounds: (n,6) or (n_meshes,6) with mesh_id; planes: (F,24).  -> vis, boundary (F,n) u8."""
    n = world.shape[0]
    planes = _f32(planes).reshape(-1, 24)
    F = planes.shape[0]
    bounds = _f32(bounds)
    vis = np.zeros((F, n), np.uint8)
    bnd = np.zeros((F, n), np.uint8)
    mid = None if mesh_id is None else np.ascontiguousarray(mesh_id, np.uint32)
    lib().yo_cull(_p(world), C.c_size_t(n), _p(bounds), _p(mid) if mid is not None else None, _p(planes),
                  C.c_uint32(F), _p(vis), _p(bnd), C.c_int(n_threads))
    return vis, bnd


def soa_to_aos68(planes16, dirty=1):
    planes16 = _f32(planes16)
    n = planes16.shape[1]
    out = np.zeros(n, T68)
    lib().yo_soa_to_aos68(_p(planes16), C.c_size_t(n), C.c_size_t(0), C.c_size_t(n), C.c_uint8(dirty), _p(out))
    return out


def aos68_to_soa(col):
    n = col.shape[0]
    out = np.empty((16, n), np.float32)
    lib().yo_aos68_to_soa(_p(col), C.c_size_t(n), _p(out), C.c_size_t(n), C.c_size_t(0))
    return out


# ---- synthetic scenes ---------------------------------------------------------------------------------
def synth_level_offsets(n, n_levels):
    out = np.zeros(n_levels + 1, np.uint32)
    lib().yo_synth_level_offsets(C.c_uint64(n), C.c_uint32(n_levels), _p(out))
    return out


def synth_parents(level_offsets, seed=SEED, index_base=0):
    lo = np.ascontiguousarray(level_offsets, np.uint32)
    out = np.empty(int(lo[-1]), np.uint32)
    lib().yo_synth_parents(C.c_uint64(seed), C.c_uint64(index_base), _p(lo), C.c_uint32(len(lo) - 1), _p(out))
    return out


def synth_trs(n, n_level0, seed=SEED, index_base=0):
    out = np.empty((10, n), np.float32)
    lib().yo_synth_trs(C.c_uint64(seed), C.c_uint64(index_base), C.c_uint64(n_level0), C.c_size_t(0),
                       C.c_size_t(n), _p(out), C.c_size_t(n))
    return out


def synth_aabb(n, seed=SEED, index_base=0):
    out = np.empty((n, 6), np.float32)
    lib().yo_synth_aabb(C.c_uint64(seed), C.c_uint64(index_base), C.c_size_t(0), C.c_size_t(n), _p(out))
    return out


def synth_mesh_table(n_meshes, seed=SEED):
    out = np.empty((n_meshes, 6), np.float32)
    lib().yo_synth_mesh_table(C.c_uint64(seed), C.c_uint32(n_meshes), _p(out))
    return out


def synth_mesh_ids(n, n_meshes, seed=SEED, index_base=0):
    out = np.empty(n, np.uint32)
    lib().yo_synth_mesh_ids(C.c_uint64(seed), C.c_uint64(index_base), C.c_uint32(n_meshes), C.c_size_t(0),
                            C.c_size_t(n), _p(out))
    return out


# ---- cameras of the benchmark configs -------------------------------------------------
def main_camera_planes():
    """perspective(75 deg, aspect 2, 0.1, 200) with the Vulkan Y flip (camera.zig:269-273) x lookAt((0,40,150)->0)."""
    proj = perspective(np.float32(np.deg2rad(75.0)), 2.0, 0.1, 200.0)
    proj[5] = np.float32(proj[5] * np.float32(-1.0))
    view = look_at([0, 40, 150], [0, 0, 0], [0, 1, 0])
    return extract_planes(mul(proj, view))


def cascade_planes():
    """4 orthographic shadow cascades, extents 12.5/25/50/100, depth 0..400, along light dir (-0.4,-1,-0.3)."""
    d = np.array([-0.4, -1.0, -0.3], np.float32)
    eye = (-d * np.float32(200.0)).astype(np.float32)
    view = look_at(eye, [0, 0, 0], [0, 1, 0])
    out = []
    for e in (12.5, 25.0, 50.0, 100.0):
        proj = orthographic(-e, e, -e, e, 0.0, 400.0)
        out.append(extract_planes(mul(proj, view)))
    return np.stack(out)


def frame_trs(trs, parent, level_offsets, root=None, n_threads=1):
    """Convenience: compose TRS, run one all-dirty frame; returns world column (T68, dirty cleared)."""
    local = compose_batch(trs)
    n = local.shape[0]
    world = np.zeros(n, T68)
    world["m"][:, [0, 5, 10, 15]] = 1.0  # LocalTransform.onAdd -> WorldTransform{IDENTITY, dirty} (transform.zig:38-44)
    world["dirty"] = 1
    r = identity68() if root is None else root
    frame_aos68(world, local, parent, level_offsets, r, n_threads)
    return world
