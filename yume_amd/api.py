"""Thin numpy-facing wrappers over the C ABI (include/yume_hip.h): Context (one GPU) and Scene.

All compute happens in libyume_hip.so's HIP kernels; these classes only marshal pointers.
"""
import ctypes as C

import numpy as np

from . import _lib as L

T68 = np.dtype({"names": ["m", "dirty", "pad"], "formats": [("<f4", 16), "u1", ("u1", 3)], "itemsize": 68})
"""flecs column row of LocalTransform / WorldTransform: extern struct { matrix: Mat4, dirty: bool }
(engine/components/transform.zig:8-12, 62-66): 68 bytes, align 4."""


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _plane_ptrs(arr2d):
    """(np, n) C-contiguous f32 -> array of np row pointers."""
    assert arr2d.dtype == np.float32 and arr2d.flags.c_contiguous and arr2d.ndim == 2
    n = arr2d.shape[0]
    ptrs = (C.c_void_p * n)()
    for p in range(n):
        ptrs[p] = arr2d[p].ctypes.data
    return ptrs


class Context:
    """One GPU (yh_ctx): device, stream, optional RCCL communicator."""

    def __init__(self, device_id=0):
        self._h = C.c_void_p()
        L.check(L.lib().yh_init(int(device_id), C.byref(self._h)))
        self.device_id = device_id
        self.rank, self.n_ranks = 0, 1

    @property
    def handle(self):
        return self._h

    def device_info(self):
        info = L.DeviceInfo()
        L.check(L.lib().yh_device_info_get(self._h, C.byref(info)), self._h)
        return {
            "device_id": info.device_id,
            "n_cus": info.n_cus,
            "wavefront_size": info.wavefront_size,
            "hbm_bytes": info.hbm_bytes,
            "arch": info.arch.decode(),
            "name": info.name.decode(),
        }

    def wait(self):
        L.check(L.lib().yh_wait(self._h), self._h)

    def host_array(self, shape, dtype):
        """A numpy array in page-locked, device-visible host memory (yh_host_alloc): copies to and from it run
        asynchronously, and a few thousand sparse rows go through the import / export kernels alone.  Freed with
        `host_free(array)`; the array must not be used afterwards."""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape))
        p = L.lib().yh_host_alloc(self._h, max(1, n * dtype.itemsize))
        if not p:
            raise MemoryError("yh_host_alloc failed")
        buf = (C.c_uint8 * (n * dtype.itemsize)).from_address(p)
        a = np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)
        self._host = getattr(self, "_host", {})
        self._host[a.__array_interface__["data"][0]] = p
        return a

    def host_free(self, a):
        p = getattr(self, "_host", {}).pop(a.__array_interface__["data"][0], None)
        if p:
            L.lib().yh_host_free(self._h, C.c_void_p(p))

    def stream(self):
        return L.lib().yh_stream(self._h)

    def stream_probe(self, read_bytes, write_bytes, launches=50):
        """Mean ms per launch of a plain read + write stream over a working set of this size (measurement harness)."""
        ms = C.c_float()
        L.check(L.lib().yh_stream_probe(self._h, int(read_bytes), int(write_bytes), int(launches), C.byref(ms)), self._h)
        return ms.value

    # ---- RCCL ----
    @staticmethod
    def comm_unique_id():
        buf = (C.c_ubyte * L.YH_UNIQUE_ID_BYTES)()
        L.check(L.lib().yh_comm_unique_id(buf))
        return bytes(buf)

    @staticmethod
    def comm_describe():
        """Which RCCL the library resolved, e.g. 'rccl 2.22.3 from /.../librccl.so'."""
        buf = C.create_string_buffer(640)
        L.check(L.lib().yh_comm_describe(buf, len(buf)))
        return buf.value.decode("utf-8", "replace")

    def comm_init(self, n_ranks, rank, unique_id):
        buf = (C.c_ubyte * L.YH_UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        L.check(L.lib().yh_comm_init(self._h, int(n_ranks), int(rank), buf), self._h)
        self.rank, self.n_ranks = rank, n_ranks

    def close(self):
        if self._h:
            L.lib().yh_shutdown(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class Scene:
    """Device-resident entity arrays of one (shard of a) scene (yh_scene)."""

    def __init__(self, ctx, n_entities, n_levels=1, n_frusta=1, local_kind=L.YH_LOCAL_TRS,
                 bounds_kind=L.YH_BOUNDS_PER_ENTITY, max_meshes=0, id_base=0, draw_capacity=0):
        self.ctx = ctx
        self.n = int(n_entities)
        self.n_levels = int(n_levels)
        self.n_frusta = int(n_frusta)
        self.local_kind = local_kind
        self.bounds_kind = bounds_kind
        self.max_meshes = self.n_meshes = int(max_meshes)
        d = L.SceneDesc(self.n, self.n_levels, self.n_frusta, local_kind, bounds_kind, max_meshes, id_base,
                        draw_capacity)
        self._h = C.c_void_p()
        L.check(L.lib().yh_scene_create(ctx.handle, C.byref(d), C.byref(self._h)), ctx.handle)

    def _ck(self, rc):
        return L.check(rc, self.ctx.handle)

    def close(self):
        if self._h:
            L.lib().yh_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- uploads ----
    def upload_hierarchy(self, parent_index, level_offsets):
        p = np.ascontiguousarray(parent_index, np.uint32)
        lo = np.ascontiguousarray(level_offsets, np.uint32)
        if p.shape != (self.n,) or lo.shape != (self.n_levels + 1,):
            raise ValueError("parent_index must have n_entities entries and level_offsets n_levels+1")
        self._ck(L.lib().yh_upload_hierarchy(self._h, _ptr(p), _ptr(lo)))
        self.level_offsets = lo.copy()

    def upload_flat(self):
        """Flat scene: one level, every entity a child of the scene root."""
        self.upload_hierarchy(np.full(self.n, L.YH_NO_PARENT, np.uint32), np.array([0, self.n], np.uint32))

    def set_root(self, m):
        m = np.ascontiguousarray(m, np.float32).reshape(16)
        self._ck(L.lib().yh_set_root(self._h, _ptr(m)))

    def upload_root_aos68(self, row):
        """Root WorldTransform as a 68-byte row: matrix and the row's own dirty byte."""
        row = np.ascontiguousarray(row).view(np.uint8).reshape(68)
        self._ck(L.lib().yh_upload_root_aos68(self._h, _ptr(row)))

    def upload_trs(self, trs, first=0):
        trs = np.ascontiguousarray(trs, np.float32)
        assert trs.shape[0] == 10
        self._ck(L.lib().yh_upload_trs(self._h, _plane_ptrs(trs), first, trs.shape[1]))

    def upload_aabb(self, mins, maxs, first=0):
        mins = np.ascontiguousarray(mins, np.float32)
        maxs = np.ascontiguousarray(maxs, np.float32)
        assert mins.shape[0] == 3 and maxs.shape == mins.shape
        self._ck(L.lib().yh_upload_aabb(self._h, _plane_ptrs(mins), _plane_ptrs(maxs), first, mins.shape[1]))

    def upload_mesh_aabbs(self, table):
        table = np.ascontiguousarray(table, np.float32).reshape(-1, 6)
        self._ck(L.lib().yh_upload_mesh_aabbs(self._h, _ptr(table), table.shape[0]))
        self.n_meshes = table.shape[0]

    def upload_mesh_ids(self, ids, first=0):
        ids = np.ascontiguousarray(ids, np.uint32)
        self._ck(L.lib().yh_upload_mesh_ids(self._h, _ptr(ids), first, ids.shape[0]))

    def upload_local_aos68(self, col, first=0):
        assert col.dtype == T68 and col.flags.c_contiguous
        self._ck(L.lib().yh_upload_local_aos68(self._h, _ptr(col), first, col.shape[0]))

    def upload_world_aos68(self, col, first=0):
        assert col.dtype == T68 and col.flags.c_contiguous
        self._ck(L.lib().yh_upload_world_aos68(self._h, _ptr(col), first, col.shape[0]))

    def mark_dirty(self, local=False, world=False, first=0, count=None):
        count = self.n - first if count is None else count
        self._ck(L.lib().yh_mark_dirty(self._h, int(local), int(world), first, count))

    def fill_synthetic(self, seed=0x59554D45, index_base=0, global_of_local=None):
        """Synthetic TRS / bounds generated on the device.  global_of_local (a shard's map to global slots): entity e
        draws the values of global entity index_base + global_of_local[e]."""
        d = L.SynthDesc(seed, index_base)
        if global_of_local is None:
            self._ck(L.lib().yh_scene_fill_synthetic(self._h, C.byref(d)))
        else:
            g = np.ascontiguousarray(global_of_local, np.uint32)
            assert g.shape == (self.n,)
            self._ck(L.lib().yh_scene_fill_synthetic_shard(self._h, C.byref(d), _ptr(g)))
        self.n_meshes = self.max_meshes

    def set_grouping(self, max_group_levels):
        self._ck(L.lib().yh_scene_set_grouping(self._h, int(max_group_levels)))

    def set_lds_parents(self, enable=True):
        """One-level launches stage their parents in LDS (same results; an A/B variant)."""
        self._ck(L.lib().yh_scene_set_lds_parents(self._h, int(bool(enable))))

    def set_plan(self, last_levels=()):
        """Explicit launch plan: group i ends at level last_levels[i]; empty = back to the planner."""
        ll = np.ascontiguousarray(last_levels, np.uint32)
        self._ck(L.lib().yh_scene_set_plan(self._h, ll.size, _ptr(ll) if ll.size else None))

    # ---- per frame ----
    def frame(self, planes=None, flags=L.YH_FRAME_ALL_DIRTY):
        if planes is None:
            self._ck(L.lib().yh_frame(self._h, None, 0, flags))
            self._last_frusta = 0
            return
        planes = np.ascontiguousarray(planes, np.float32).reshape(-1, 24)
        self._ck(L.lib().yh_frame(self._h, _ptr(planes), planes.shape[0], flags))
        self._last_frusta = 0 if flags & L.YH_FRAME_NO_CULL else planes.shape[0]

    def frame_static(self, planes=None, flags=L.YH_FRAME_ALL_DIRTY):
        """yh_frame_static: the same frame with its stage-3 scratch in the library's static device array (allocates
        nothing).  Results are read exactly as after frame()."""
        if planes is None:
            self._ck(L.lib().yh_frame_static(self._h, None, 0, flags))
            self._last_frusta = 0
            return
        planes = np.ascontiguousarray(planes, np.float32).reshape(-1, 24)
        self._ck(L.lib().yh_frame_static(self._h, _ptr(planes), planes.shape[0], flags))
        self._last_frusta = 0 if flags & L.YH_FRAME_NO_CULL else planes.shape[0]

    def counts(self):
        c = np.zeros(L.YH_MAX_FRUSTA, np.uint32)
        b = np.zeros(L.YH_MAX_FRUSTA, np.uint32)
        self._ck(L.lib().yh_get_counts(self._h, _ptr(c), _ptr(b)))
        return c, b

    def draw_list(self, frustum):
        c, _ = self.counts()
        out = np.empty(int(c[frustum]), np.uint32)
        written = C.c_uint64()
        self._ck(L.lib().yh_get_draw_list(self._h, frustum, _ptr(out), out.shape[0], C.byref(written)))
        return out[: written.value]

    # ---- downloads ----
    def world_soa(self, first=0, count=None):
        count = self.n - first if count is None else count
        out = np.empty((16, count), np.float32)
        self._ck(L.lib().yh_download_world_soa(self._h, _plane_ptrs(out), first, count))
        return out

    def world_aos68(self, first=0, count=None):
        count = self.n - first if count is None else count
        out = np.zeros(count, T68)
        self._ck(L.lib().yh_download_world_aos68(self._h, _ptr(out), first, count))
        return out

    def local_aos68(self, first=0, count=None):
        count = self.n - first if count is None else count
        out = np.zeros(count, T68)
        self._ck(L.lib().yh_download_local_aos68(self._h, _ptr(out), first, count))
        return out

    def decompose(self, which="world", first=0, count=None):
        """Batched Mat4.decompose: ((10, count) planes t.xyz q.xyzw s.xyz, (count,) u8 valid)."""
        count = self.n - first if count is None else count
        out = np.zeros((10, count), np.float32)
        valid = np.zeros(count, np.uint8)
        self._ck(L.lib().yh_decompose(self._h, 0 if which == "world" else 1, _plane_ptrs(out), _ptr(valid), first, count))
        return out, valid

    # ---- draw-list consumer ----
    def upload_vertex_counts(self, counts):
        counts = np.ascontiguousarray(counts, np.uint32)
        self._ck(L.lib().yh_upload_vertex_counts(self._h, _ptr(counts), counts.size))

    def object_data(self, frustum=0, batch_offset=0):
        """(model matrices [k,16], draw records [k,4]) of the frustum's visible entities, list order."""
        self._ck(L.lib().yh_build_object_data(self._h, frustum, batch_offset))
        counts, _ = self.counts()
        k = int(counts[frustum])
        m = np.zeros((k, 16), np.float32)
        c = np.zeros((k, 4), np.uint32)
        wr = C.c_uint64()
        self._ck(L.lib().yh_get_object_data(self._h, frustum, _ptr(m), _ptr(c), k, C.byref(wr)))
        assert wr.value == k
        return m, c

    def mesh_buckets(self, frustum=0, batch_offset=0):
        """Visible entities grouped by mesh id: (buckets [n_meshes,3] = mesh/first/count, bucket-ordered global ids
        [k], model matrices [k,16], draw records [k,4])."""
        self._ck(L.lib().yh_build_mesh_buckets(self._h, frustum, batch_offset))
        counts, _ = self.counts()
        k = int(counts[frustum])
        nm = int(self.n_meshes)
        b = np.zeros((nm, 3), np.uint32)
        ids = np.zeros(k, np.uint32)
        wr = C.c_uint64()
        self._ck(L.lib().yh_get_mesh_buckets(self._h, frustum, _ptr(b), nm, _ptr(ids), k, C.byref(wr)))
        assert wr.value == k
        m = np.zeros((k, 16), np.float32)
        c = np.zeros((k, 4), np.uint32)
        self._ck(L.lib().yh_get_object_data(self._h, frustum, _ptr(m), _ptr(c), k, C.byref(wr)))
        return b, ids, m, c

    # ---- sparse rows / change tracking (Mat4 scenes) ----
    def upload_local_rows(self, slots, rows):
        slots = np.ascontiguousarray(slots, np.uint32)
        rows = np.ascontiguousarray(rows, T68)
        assert slots.shape == rows.shape
        self._ck(L.lib().yh_upload_local_rows_aos68(self._h, _ptr(slots), _ptr(rows), slots.size))

    def upload_world_rows(self, slots, rows):
        slots = np.ascontiguousarray(slots, np.uint32)
        rows = np.ascontiguousarray(rows, T68)
        assert slots.shape == rows.shape
        self._ck(L.lib().yh_upload_world_rows_aos68(self._h, _ptr(slots), _ptr(rows), slots.size))

    def move_rows(self, src_slots, dst_slots):
        """Both rows of src_slots[i] become the rows of dst_slots[i], on the device (sources read first)."""
        src = np.ascontiguousarray(src_slots, np.uint32)
        dst = np.ascontiguousarray(dst_slots, np.uint32)
        assert src.shape == dst.shape
        self._ck(L.lib().yh_move_rows(self._h, _ptr(src), _ptr(dst), src.size))

    def track_changes(self, enable=True):
        self._ck(L.lib().yh_scene_track_changes(self._h, int(enable)))  # True / YH_TRACK_ROWS, YH_TRACK_REWRITTEN_LOCALS

    def changed(self):
        """(slots, world rows, local rows) the last frame recomputed or found edited; ascending slots."""
        n = C.c_uint64()
        self._ck(L.lib().yh_get_changed_count(self._h, C.byref(n)))
        slots = np.zeros(n.value, np.uint32)
        w = np.zeros(n.value, T68)
        l = np.zeros(n.value, T68)
        wr = C.c_uint64()
        self._ck(L.lib().yh_download_changed_aos68(self._h, _ptr(slots), _ptr(w), _ptr(l), n.value, C.byref(wr)))
        return slots, w, l

    def upload_parent_range(self, parents, first):
        """Parent indices of slots [first, first + len(parents)), all of one level."""
        parents = np.ascontiguousarray(parents, np.uint32)
        self._ck(L.lib().yh_upload_parent_range(self._h, _ptr(parents), first, parents.size))

    def set_alive_range(self, first, count, alive):
        """Slots that hold no entity are never visible and never reported as changed (Mat4 scenes)."""
        self._ck(L.lib().yh_scene_set_alive_range(self._h, first, count, int(bool(alive))))

    def set_alive_runs(self, first, count, alive):
        """The same for many runs in one call, applied in the order given."""
        first = np.ascontiguousarray(first, np.uint32)
        count = np.ascontiguousarray(count, np.uint32)
        alive = np.ascontiguousarray(alive, np.uint8)
        assert first.size == count.size == alive.size
        self._ck(L.lib().yh_scene_set_alive_runs(self._h, _ptr(first), _ptr(count), _ptr(alive), first.size))

    def upload_parent_rows(self, slots, parents):
        """Parent indices of single slots (any levels)."""
        slots = np.ascontiguousarray(slots, np.uint32)
        parents = np.ascontiguousarray(parents, np.uint32)
        assert slots.size == parents.size
        self._ck(L.lib().yh_upload_parent_rows(self._h, _ptr(slots), _ptr(parents), slots.size))

    def upload_mesh_id_rows(self, slots, ids):
        slots = np.ascontiguousarray(slots, np.uint32)
        ids = np.ascontiguousarray(ids, np.uint32)
        assert slots.size == ids.size
        self._ck(L.lib().yh_upload_mesh_id_rows(self._h, _ptr(slots), _ptr(ids), slots.size))

    def upload_aabb_rows(self, slots, min_max):
        """Boxes of single slots: min_max is (len(slots), 6) = min.xyz, max.xyz."""
        slots = np.ascontiguousarray(slots, np.uint32)
        min_max = np.ascontiguousarray(min_max, np.float32).reshape(-1, 6)
        assert slots.size == min_max.shape[0]
        self._ck(L.lib().yh_upload_aabb_rows(self._h, _ptr(slots), _ptr(min_max), slots.size))

    def changed_counts(self):
        """(rows whose WorldTransform changed, rows whose LocalTransform changed) in the last tracked frame."""
        nw, nl = C.c_uint64(), C.c_uint64()
        self._ck(L.lib().yh_get_changed_counts(self._h, C.byref(nw), C.byref(nl)))
        return nw.value, nl.value

    def changed_rows(self, which, on_ready=None, out=None):
        """(slots, rows) of one component's changed rows (which: 0 WorldTransform, 1 LocalTransform), streamed:
        on_ready(rows_ready) is called after each piece has landed.  out = (slots, rows): the caller's own arrays
        (e.g. Context.host_array ones) with room for the rows; views of their first n entries are returned."""
        n = self.changed_counts()[which]
        if out is not None:
            assert out[0].size >= n and out[1].size >= n and out[0].dtype == np.uint32 and out[1].dtype == T68
            slots, rows = out[0][:n], out[1][:n]
        else:
            slots = np.zeros(n, np.uint32)
            rows = np.zeros(n, T68)
        wr = C.c_uint64()
        cb = L.ROWS_READY_FN((lambda user, r: on_ready(int(r))) if on_ready else 0)
        self._ck(L.lib().yh_download_changed_rows_aos68(self._h, which, _ptr(slots), _ptr(rows), n, C.byref(wr),
                                                        C.cast(cb, C.c_void_p), None))
        assert wr.value == n
        return slots, rows

    def compose(self, first=0, count=None):
        count = self.n - first if count is None else count
        out = np.empty((16, count), np.float32)
        self._ck(L.lib().yh_compose(self._h, _plane_ptrs(out), first, count))
        return out

    def download_trs(self, first=0, count=None):
        count = self.n - first if count is None else count
        out = np.empty((10, count), np.float32)
        self._ck(L.lib().yh_download_trs(self._h, _plane_ptrs(out), first, count))
        return out

    def download_aabb(self, first=0, count=None):
        count = self.n - first if count is None else count
        mins = np.empty((3, count), np.float32)
        maxs = np.empty((3, count), np.float32)
        self._ck(L.lib().yh_download_aabb(self._h, _plane_ptrs(mins), _plane_ptrs(maxs), first, count))
        return mins, maxs

    # ---- measurement ----
    def timings(self):
        t = L.Timing()
        self._ck(L.lib().yh_timings(self._h, C.byref(t)))
        return {"last_frame_ms": t.last_frame_ms, "last_kernel_ms": t.last_kernel_ms, "n_launches": t.n_launches,
                "n_groups": t.n_groups, "frames": t.frames}

    def set_kernel_timing(self, enable, period=1):
        """0 off, 1 every launch + frame, 2+i only launch i; events on every period-th frame."""
        self._ck(L.lib().yh_scene_set_kernel_timing_period(self._h, int(period)))
        self._ck(L.lib().yh_scene_set_kernel_timing(self._h, int(enable)))

    def groups(self):
        n = C.c_uint32()
        q = np.zeros((L.YH_MAX_LEVELS, 4), np.uint32)
        self._ck(L.lib().yh_scene_get_groups(self._h, _ptr(q), L.YH_MAX_LEVELS, C.byref(n)))
        return q[: n.value].copy()

    def group_kernel_names(self):
        """Kernel (rocprofv3 spelling) each level group's launch of the last frame ran."""
        buf = C.create_string_buffer(64 * 200)
        n = C.c_uint32()
        self._ck(L.lib().yh_scene_get_group_kernel_names(self._h, buf, len(buf), C.byref(n)))
        return buf.value.decode().split("\n")[: n.value]

    def group_timings(self):
        ms = np.zeros(L.YH_MAX_LEVELS, np.float32)
        cnt = np.zeros(L.YH_MAX_LEVELS, np.uint32)
        self._ck(L.lib().yh_scene_get_group_timings(self._h, _ptr(ms), _ptr(cnt), L.YH_MAX_LEVELS))
        return ms, cnt

    # ---- multi-GPU ----
    def allgather_draw_lists(self):
        """(n_ranks, frusta of the last frame) visible counts; lists via gathered_draw_list()."""
        out = np.zeros((self.ctx.n_ranks, max(1, getattr(self, "_last_frusta", self.n_frusta))), np.uint32)
        self._ck(L.lib().yh_allgather_draw_lists(self._h, _ptr(out)))
        return out

    def allgather_draw_lists_async(self):
        """Enqueue the exchange (no host sync); read with gathered_counts() / gathered_draw_list()."""
        self._ck(L.lib().yh_allgather_draw_lists_async(self._h))

    def gather_selftest(self, fake_ranks):
        """One-GPU rehearsal of the gather's expansion: fake_ranks copies of this scene's own visibility words."""
        self._ck(L.lib().yh_gather_selftest(self._h, int(fake_ranks)))

    @staticmethod
    def gather_selftest_shards(shards):
        """One-GPU rehearsal of the gather with R different shard scenes (rank order); results land in shards[0]."""
        arr = (C.c_void_p * len(shards))(*[s._h for s in shards])
        L.check(L.lib().yh_gather_selftest_shards(arr, len(shards)), shards[0].ctx.handle)

    def gathered_counts(self, n_frusta, n_ranks=None):
        out = np.zeros((n_ranks or self.ctx.n_ranks, max(1, n_frusta)), np.uint32)
        self._ck(L.lib().yh_get_gathered_counts(self._h, _ptr(out)))
        return out

    def gathered_draw_list(self, frustum, total):
        out = np.empty(int(total), np.uint32)
        written = C.c_uint64()
        self._ck(L.lib().yh_get_gathered_draw_list(self._h, frustum, _ptr(out), out.shape[0], C.byref(written)))
        return out[: written.value]


# ---- frustum helpers (host scalar code in the library; same arithmetic as engine/math3d.zig) ----
def mat4_mul(a, b):
    a = np.ascontiguousarray(a, np.float32).reshape(16)
    b = np.ascontiguousarray(b, np.float32).reshape(16)
    out = np.empty(16, np.float32)
    L.lib().yh_mat4_mul(_ptr(a), _ptr(b), _ptr(out))
    return out


def perspective(fovy_rad, aspect, near, far):
    out = np.empty(16, np.float32)
    L.lib().yh_perspective(fovy_rad, aspect, near, far, _ptr(out))
    return out


def orthographic(l, r, b, t, n, f):
    out = np.empty(16, np.float32)
    L.lib().yh_orthographic(l, r, b, t, n, f, _ptr(out))
    return out


def look_at(eye, at, up):
    eye, at, up = (np.ascontiguousarray(v, np.float32) for v in (eye, at, up))
    out = np.empty(16, np.float32)
    L.lib().yh_look_at(_ptr(eye), _ptr(at), _ptr(up), _ptr(out))
    return out


def camera_view_projection(orthographic, pos, rot, aspect=2.0, fovy_rad=1.3089969, near=0.1, far=200.0):
    """Camera.view_projection as engine/components/camera.zig:253-302 updates it."""
    pos, rot = (np.ascontiguousarray(v, np.float32) for v in (pos, rot))
    out = np.empty(16, np.float32)
    L.lib().yh_camera_view_projection(int(orthographic), _ptr(pos), _ptr(rot), aspect, fovy_rad, near, far, _ptr(out))
    return out


def mat4_decompose(m):
    """`matrix.decompose() catch Decomposed.IDENTITY` (components/transform.zig:68-70): (t, q xyzw, s, valid)."""
    m = np.ascontiguousarray(m, np.float32).reshape(16)
    t, q, s = np.empty(3, np.float32), np.empty(4, np.float32), np.empty(3, np.float32)
    ok = L.lib().yh_mat4_decompose(_ptr(m), _ptr(t), _ptr(q), _ptr(s))
    return t, q, s, bool(ok)


def quat_to_euler(q):
    """Quat.toEuler (math3d.zig:418-442): degrees."""
    q = np.ascontiguousarray(q, np.float32).reshape(4)
    out = np.empty(3, np.float32)
    L.lib().yh_quat_to_euler(_ptr(q), _ptr(out))
    return out


def camera_update_from_world(orthographic, world, aspect=2.0, fovy_rad=1.3089969, near=0.1, far=200.0):
    """GameWindow.zig:160-161 for one Camera entity: (view, projection, view_projection, decompose succeeded)."""
    world = np.ascontiguousarray(world, np.float32).reshape(16)
    view, proj, vp = (np.empty(16, np.float32) for _ in range(3))
    ok = L.lib().yh_camera_update_from_world(int(orthographic), _ptr(world), aspect, fovy_rad, near, far, _ptr(view), _ptr(proj), _ptr(vp))
    return view, proj, vp, bool(ok)


def extract_planes(view_projection):
    vp = np.ascontiguousarray(view_projection, np.float32).reshape(16)
    out = np.empty(24, np.float32)
    L.lib().yh_extract_planes(_ptr(vp), _ptr(out))
    return out
