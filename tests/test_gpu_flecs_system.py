"""The engine-side drop-in, end to end: the REAL flecs 4.0.4 world (the reference's vendored copy, compiled in place
into oracle/_ref/) with yume_amd/host/GpuTransformCullSystem.c registered instead of TransformSyncSystem +
PostTransformSystem must leave every WorldTransform / LocalTransform row — matrices and dirty bytes — exactly as the
reference's three CPU systems leave them, frame after frame, through local edits, world-only edits (inverse branch),
a singular matrix, a re-rooting, deletions and a re-parenting (slot-order rebuilds).  Both binaries are built from
oracle/flecs_harness.c by `make -C oracle ref`; the visible lists of the GPU run are checked against the oracle's
culling of the CPU run's world matrices."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPU = os.path.join(ROOT, "oracle", "_ref", "flecs_harness")
# YFH_GPU_BIN: another build of the same harness, e.g. oracle/_ref/flecs_gpu_harness_asan (host-side sanitizers)
GPU = os.environ.get("YFH_GPU_BIN") or os.path.join(ROOT, "oracle", "_ref", "flecs_gpu_harness")

pytestmark = [
    pytest.mark.gpu,
    pytest.mark.skipif(not (os.path.exists(CPU) and os.path.exists(GPU)), reason="oracle/_ref harnesses not built"),
]


def write_scene(path, lo, parent, local_m, root_m, planes, bounds, frames):
    n = parent.shape[0]
    with open(path, "wb") as f:
        f.write(b"YFH2")
        f.write(struct.pack("<III", n, len(lo) - 1, len(frames)))
        f.write(np.ascontiguousarray(lo, np.uint32).tobytes())
        f.write(np.ascontiguousarray(parent, np.uint32).tobytes())
        f.write(np.ascontiguousarray(local_m, np.float32).tobytes())
        f.write(np.ascontiguousarray(root_m, np.float32).tobytes())
        f.write(struct.pack("<I", planes.shape[0]))
        f.write(np.ascontiguousarray(planes, np.float32).tobytes())
        f.write(np.ascontiguousarray(bounds, np.float32).tobytes())
        for edits in frames:
            f.write(struct.pack("<I", len(edits)))
            for e, kind, m in edits:
                f.write(struct.pack("<II", 0xFFFFFFFF if e < 0 else e, kind))
                f.write(np.ascontiguousarray(m, np.float32).tobytes())


def read_out(path, n, n_frames, n_frusta, T68, gpu=False, n_cams=0, buckets=False):
    """n_cams (YFH_CAMERA_PIPELINE runs): the frame's tuple ends in an (n_cams, 3, 16) array, the view / projection /
    view_projection matrices the Camera components hold after the frame."""
    raw = open(path, "rb").read()
    res, pos = [], 0
    for _ in range(n_frames):
        w = np.frombuffer(raw, T68, n, pos).copy()
        pos += 68 * n
        l = np.frombuffer(raw, T68, n, pos).copy()
        pos += 68 * n
        a, b, shared = struct.unpack_from("<QQI", raw, pos)
        pos += 20
        vis = []
        for _f in range(n_frusta):
            (c,) = struct.unpack_from("<I", raw, pos)
            vis.append(np.frombuffer(raw, np.uint32, c, pos + 4).copy())
            pos += 4 + 4 * c
        if gpu:
            vis.append(struct.unpack_from("<QQQQQ", raw, pos))  # last element: (rows up, rows down, walks, skipped, patches)
            pos += 40
        if n_cams:
            cams = np.frombuffer(raw, np.float32, n_cams * 48, pos).reshape(n_cams, 3, 16).copy()
            pos += n_cams * 192
            res.append((w, l, a, b, vis, cams))
            continue
        if buckets:  # YFH_MATERIALS runs: ((material, mesh, first, count) rows, entity indices in record order)
            (nb,) = struct.unpack_from("<I", raw, pos)
            bk = np.frombuffer(raw, np.uint32, nb * 4, pos + 4).reshape(nb, 4).copy()
            pos += 4 + 16 * nb
            (nrec,) = struct.unpack_from("<Q", raw, pos)
            rec = np.frombuffer(raw, np.uint32, nrec, pos + 8).copy()
            pos += 8 + 4 * nrec
            res.append((w, l, a, b, vis, (bk, rec)))
            continue
        res.append((w, l, a, b, vis))
    assert pos == len(raw)
    return res


def reparent_arg(index):
    m = np.zeros(16, np.float32)
    m.view(np.uint32)[0] = index
    return m


@pytest.mark.parametrize("plain,always_walk,validate,no_patch",
                         [(False, False, False, False), (True, False, False, False), (False, True, False, False), (False, False, True, False),
                          (False, False, False, True)],
                         ids=["incremental", "whole-columns", "incremental-always-walk", "incremental-validate-every-frame",
                              "incremental-no-patching"])
@pytest.mark.parametrize("n,levels", [(10_000, 1), (4096, 8), (50_000, 6)])
def test_gpu_system_in_real_flecs_equals_the_cpu_systems(oracle, tmp_path, n, levels, plain, always_walk, validate, no_patch):
    rng = np.random.default_rng(n + levels)
    lo = oracle.synth_level_offsets(n, levels)
    par = oracle.synth_parents(lo)
    local = oracle.compose_batch(oracle.synth_trs(n, int(lo[1])))
    bounds = oracle.synth_aabb(n)
    planes = np.concatenate([oracle.main_camera_planes().reshape(1, 24), oracle.cascade_planes()[:2]])

    def rnd_m(scale=5.0):
        return oracle.compose(rng.uniform(-scale, scale, 3), rng.uniform(-1, 1, 4), rng.uniform(0.5, 2, 3))

    frames = [[], []]
    frames.append([(int(e), 0, rnd_m()) for e in np.unique(rng.integers(0, n, 300))])
    frames.append([(int(e), 1, rnd_m(50.0)) for e in np.unique(rng.integers(0, n, 200))])
    frames.append([(int(rng.integers(0, int(lo[1]))), 0, np.zeros(16, np.float32))])
    frames.append([(-1, 1, oracle.compose([1, 2, 3], [0, 0.7071068, 0, 0.7071068], [2, 2, 2]))])
    # structural changes: delete a few last-level entities (no children), re-parent a few others one level up or to
    # another parent of the same level (leaves only, so flecs' table order stays parent-first: SURVEY F5)
    last = np.arange(int(lo[-2]), n)
    doomed = rng.choice(last, 20, replace=False)
    frames.append([(int(e), 2, np.zeros(16, np.float32)) for e in doomed])
    movers = np.setdiff1d(last, doomed)[:10]
    if levels > 1:
        new_par = rng.integers(int(lo[-3]), int(lo[-2]), movers.size)
        frames.append([(int(e), 3, reparent_arg(int(p))) for e, p in zip(movers, new_par)] + [(int(movers[0]), 0, rnd_m())])
    else:
        frames.append([(int(movers[0]), 0, rnd_m())])
    frames.append([])

    scene, out_c, out_g = tmp_path / "scene.bin", tmp_path / "cpu.bin", tmp_path / "gpu.bin"
    write_scene(scene, lo, par, local["m"], oracle.identity68()["m"][0], planes, bounds, frames)
    subprocess.check_call([CPU, str(scene), str(out_c)], timeout=300)
    env = dict(os.environ)
    env.pop("YFH_PLAIN", None)
    env.pop("YFH_ALWAYS_WALK", None)
    if plain:
        env["YFH_PLAIN"] = "1"
    if always_walk:
        env["YFH_ALWAYS_WALK"] = "1"
    env.pop("YFH_VALIDATE", None)
    if validate:
        env["YFH_VALIDATE"] = "1"
    env.pop("YFH_NO_PATCH", None)
    if no_patch:
        env["YFH_NO_PATCH"] = "1"
    subprocess.check_call([GPU, str(scene), str(out_g)], timeout=300, env=env)
    cpu = read_out(out_c, n, len(frames), 0, oracle.T68)
    gpu = read_out(out_g, n, len(frames), planes.shape[0], oracle.T68, gpu=True)

    alive = np.ones(n, bool)
    rebuilds_before = patches_before = 0
    # leaf deletions and re-parentings at the same depth are patched (no walk, no rebuild) in the default mode
    patching = not (plain or always_walk or validate or no_patch)
    for fi, ((cw, cl, _, rows, _), (gw, gl, rebuilds, g_rows, vis)) in enumerate(zip(cpu, gpu)):
        for e, kind, _m in frames[fi]:
            if kind == 2:
                alive[e] = False
        with np.errstate(all="ignore"):
            # IEEE value equality; the only representation freedom is the sign of a zero (compose/mul)
            traffic = f"(rows up, rows down, walks, walks skipped, patches) = {vis[-1]}"
            bad = np.flatnonzero(~np.all((gw["m"] == cw["m"]) | (np.isnan(gw["m"]) & np.isnan(cw["m"])), axis=1))
            assert bad.size == 0, f"frame {fi}: {bad.size} world matrices differ, first rows {bad[:8]} {traffic}"
            assert np.array_equal(gl["m"], cl["m"], equal_nan=True), f"frame {fi}: local matrices {traffic}"
        assert np.array_equal(gw["dirty"], cw["dirty"]) and np.array_equal(gl["dirty"], cl["dirty"]), f"frame {fi}: dirty bytes"
        assert g_rows == rows == alive.sum()
        structural = any(kind in (2, 3) for _e, kind, _m in frames[fi])
        up, down, walks, skipped, patches = vis[-1]
        # (one level = ONE table of n rows: moving it whole is no patch, the system rebuilds)
        patched = structural and patching and levels > 1
        assert rebuilds == rebuilds_before + (1 if (fi == 0 or (structural and not patched)) else 0), f"frame {fi}: rebuild count {traffic}"
        assert patches == patches_before + (1 if patched else 0), f"frame {fi}: patch count {traffic}"
        rebuilds_before, patches_before = rebuilds, patches
        assert walks + skipped == fi + 1
        if always_walk:
            assert skipped == 0
        elif plain or validate:  # the query is iterated on the first frame and on the frames after a structural change, only
            assert walks == rebuilds, f"frame {fi}: {walks} walks for {rebuilds} rebuilds"
        else:  # incremental mode with trusted observers: later rebuilds take the tables from the observers' lists
            assert walks == 1, f"frame {fi}: {walks} walks for {rebuilds} rebuilds"
        if plain:
            assert up == down == 2 * g_rows, f"frame {fi}: whole columns travel in plain mode"
        elif not (fi == 0 or structural):
            n_edits = len(frames[fi])
            assert up <= n_edits, f"frame {fi}: {up} rows uploaded for {n_edits} edits"
            if n_edits == 0:
                assert up == 0 and down == 0, f"frame {fi}: nothing moved, yet rows crossed PCIe"
            assert down < 2 * g_rows or fi == 5  # only the re-rooting frame touches every row
        idx = np.flatnonzero(alive)
        exp_vis, _ = oracle.cull(np.ascontiguousarray(cw[idx]), np.ascontiguousarray(bounds[idx]), planes)
        for f in range(planes.shape[0]):
            assert np.array_equal(np.sort(vis[f]), idx[np.flatnonzero(exp_vis[f])]), f"frame {fi} frustum {f}: visible set"


def test_long_edit_lists_are_gathered_by_helper_threads(oracle, tmp_path):
    """Frames with tens of thousands of onSet calls: the rows are collected from their tables by several threads, each
    with a range of the list (2048 rows per thread at least); entities set twice in a frame, in the same range or in
    two, travel once with their last value.  Against the CPU systems, row for row."""
    n, levels = 60_000, 6
    rng = np.random.default_rng(77)
    lo = oracle.synth_level_offsets(n, levels)
    par = oracle.synth_parents(lo)
    local = oracle.compose_batch(oracle.synth_trs(n, int(lo[1])))
    bounds = oracle.synth_aabb(n)
    planes = oracle.main_camera_planes().reshape(1, 24)

    def rnd_m(scale=5.0):
        return oracle.compose(rng.uniform(-scale, scale, 3), rng.uniform(-1, 1, 4), rng.uniform(0.5, 2, 3))

    many = rng.permutation(n)[:14_000]
    again = np.concatenate([many[:1500], many[-1500:]])  # set twice: near the first set, and a whole list away from it
    frames = [[], [],
              [(int(e), 0, rnd_m()) for e in many] + [(int(e), 0, rnd_m()) for e in again],
              [(int(e), 1, rnd_m(50.0)) for e in rng.permutation(n)[:9_000]],
              [(int(e), int(k), rnd_m()) for e, k in zip(rng.permutation(n)[:20_000], rng.integers(0, 2, 20_000))],
              []]
    scene, out_c, out_g = tmp_path / "scene.bin", tmp_path / "cpu.bin", tmp_path / "gpu.bin"
    write_scene(scene, lo, par, local["m"], oracle.identity68()["m"][0], planes, bounds, frames)
    subprocess.check_call([CPU, str(scene), str(out_c)], timeout=300)
    env = {k: v for k, v in os.environ.items() if not k.startswith("YFH_")}
    env["YFH_THREADS"] = "6"
    subprocess.check_call([GPU, str(scene), str(out_g)], timeout=300, env=env)
    cpu = read_out(out_c, n, len(frames), 0, oracle.T68)
    gpu = read_out(out_g, n, len(frames), 1, oracle.T68, gpu=True)
    for fi, ((cw, cl, _, rows, _), (gw, gl, _reb, g_rows, vis)) in enumerate(zip(cpu, gpu)):
        with np.errstate(all="ignore"):
            assert np.array_equal(gw["m"], cw["m"], equal_nan=True), f"frame {fi}: world matrices {vis[-1]}"
            assert np.array_equal(gl["m"], cl["m"], equal_nan=True), f"frame {fi}: local matrices {vis[-1]}"
        assert np.array_equal(gw["dirty"], cw["dirty"]) and np.array_equal(gl["dirty"], cl["dirty"]), f"frame {fi}: dirty bytes"
        assert g_rows == rows == n
        if fi >= 1:
            assert vis[-1][0] == len({e for e, _k, _m in frames[fi]}), f"frame {fi}: every edited entity travels once: {vis[-1]}"
        exp_vis, _ = oracle.cull(np.ascontiguousarray(cw), bounds, planes)
        assert np.array_equal(np.sort(vis[0]), np.flatnonzero(exp_vis[0])), f"frame {fi}: visible set"


def test_results_can_stay_on_the_device_until_readback_is_switched_on(oracle, tmp_path):
    """setReadback(0): nothing comes back (a device consumer reads matrices and lists where they are), the device stays
    right frame after frame — visible sets of every frame, structural changes included — and the frame after readback is
    switched on again leaves flecs' columns exactly as the CPU systems have them."""
    n, levels = 30_000, 6
    rng = np.random.default_rng(77)
    lo = oracle.synth_level_offsets(n, levels)
    par = oracle.synth_parents(lo)
    local = oracle.compose_batch(oracle.synth_trs(n, int(lo[1])))
    bounds = oracle.synth_aabb(n)
    planes = oracle.main_camera_planes().reshape(1, 24)

    def rnd_m(scale=5.0):
        return oracle.compose(rng.uniform(-scale, scale, 3), rng.uniform(-1, 1, 4), rng.uniform(0.5, 2, 3))

    last = np.arange(int(lo[-2]), n)
    doomed = rng.choice(last, 15, replace=False)
    movers = np.setdiff1d(last, doomed)[:8]
    frames = [[], [(int(e), 0, rnd_m()) for e in np.unique(rng.integers(0, n, 200))],
              [(int(e), 1, rnd_m(50.0)) for e in np.unique(rng.integers(0, n, 100))],
              [(-1, 1, oracle.compose([1, 2, 3], [0, 0.7071068, 0, 0.7071068], [2, 2, 2]))],
              # a structural change while the host copies are stale, together with fresh edits of other rows
              [(int(e), 2, np.zeros(16, np.float32)) for e in doomed] + [(int(e), 0, rnd_m()) for e in movers[:3]],
              [(int(e), 3, reparent_arg(int(p))) for e, p in zip(movers, rng.integers(int(lo[-3]), int(lo[-2]), movers.size))],
              [(int(movers[1]), 0, rnd_m())], []]
    scene, out_c, out_g = tmp_path / "scene.bin", tmp_path / "cpu.bin", tmp_path / "gpu.bin"
    write_scene(scene, lo, par, local["m"], oracle.identity68()["m"][0], planes, bounds, frames)
    subprocess.check_call([CPU, str(scene), str(out_c)], timeout=300)
    env = {k: v for k, v in os.environ.items() if not k.startswith("YFH_")}
    env["YFH_NO_READBACK"] = "1"
    subprocess.check_call([GPU, str(scene), str(out_g)], timeout=300, env=env)
    cpu = read_out(out_c, n, len(frames), 0, oracle.T68)
    gpu = read_out(out_g, n, len(frames), 1, oracle.T68, gpu=True)
    alive = np.ones(n, bool)
    for fi, ((cw, cl, _, rows, _), (gw, gl, _rebuilds, g_rows, vis)) in enumerate(zip(cpu, gpu)):
        for e, kind, _m in frames[fi]:
            if kind == 2:
                alive[e] = False
        assert g_rows == rows == alive.sum()
        idx = np.flatnonzero(alive)
        exp_vis, _ = oracle.cull(np.ascontiguousarray(cw[idx]), np.ascontiguousarray(bounds[idx]), planes)
        assert np.array_equal(np.sort(vis[0]), idx[np.flatnonzero(exp_vis[0])]), f"frame {fi}: visible set (device state)"
        up, down, _walks, _skipped, _patches = vis[-1]
        structural = any(kind in (2, 3) for _e, kind, _m in frames[fi])
        if fi + 1 < len(frames):
            assert down == 0 or structural or fi == 0, f"frame {fi}: rows came back although readback is off"
        else:  # the last frame ran with readback on: whole columns came home
            # (both columns whole: the rows and the spare slots every level ends in)
            assert 2 * g_rows <= down <= 2 * (g_rows + g_rows // 8 + 1024 * levels)
            with np.errstate(all="ignore"):
                assert np.array_equal(gw["m"], cw["m"], equal_nan=True) and np.array_equal(gl["m"], cl["m"], equal_nan=True)
            assert np.array_equal(gw["dirty"], cw["dirty"]) and np.array_equal(gl["dirty"], cl["dirty"])


def test_the_system_reads_cameras_and_mesh_bounds_from_the_ecs(oracle, tmp_path):
    """SURVEY 8f N4 / the drop-in's own inputs: no setFrusta, no setBounds.  Two Camera entities (240-byte component,
    view_projection at offset 176 as in engine/components/camera.zig) give the frusta, a Mesh{ref} component pointing at
    a meshes.Mesh-shaped struct gives each entity its AABB (three meshes; every fourth entity has none: an empty box);
    mid-run a camera's matrix and some entities' meshes change.  Visible sets must be the oracle's for those inputs."""
    from yume_amd import api, synth

    n, levels = 40_000, 6
    rng = np.random.default_rng(9)
    lo = oracle.synth_level_offsets(n, levels)
    par = oracle.synth_parents(lo)
    local = oracle.compose_batch(oracle.synth_trs(n, int(lo[1])))
    mesh_box = {0: [0, 0, 0, 0, 0, 0]}
    for k in (1, 2, 3):
        mesh_box[k] = [-0.5 * k, -0.25 * k - 0.5, -1.0, 0.5 * k, 0.25 * k + 0.5, 1.5]
    mesh_of = (np.arange(n) % 4).astype(np.int64)
    cam = [synth.main_camera_planes(), synth.cascade_planes()[1]]

    def arg_u32(v):
        m = np.zeros(16, np.float32)
        m.view(np.uint32)[0] = v
        return m

    new_vp = api.mat4_mul(api.orthographic(-60, 60, -60, 60, 0.0, 400.0), api.look_at([10, 90, 20], [0, 0, 0], [0, 1, 0]))
    remesh = np.unique(rng.integers(0, n, 500))
    remesh = remesh[mesh_of[remesh] != 0][:300]  # entities that already have a Mesh: a set, not an add (no row moves)
    gain = np.unique(rng.integers(0, n, 200))
    gain = gain[mesh_of[gain] == 0][:40]          # entities that get their first Mesh: an archetype move (structural)
    frames = [[], [(int(e), 0, oracle.compose(rng.uniform(-5, 5, 3), rng.uniform(-1, 1, 4), [1, 1, 1])) for e in np.unique(rng.integers(0, n, 300))],
              [(1, 5, new_vp)],
              [(int(e), 4, arg_u32(int(mesh_of[e]) % 3 + 1)) for e in remesh],
              [(int(e), 4, arg_u32(2)) for e in gain], []]
    scene, out_c, out_g = tmp_path / "scene.bin", tmp_path / "cpu.bin", tmp_path / "gpu.bin"
    write_scene(scene, lo, par, local["m"], oracle.identity68()["m"][0], np.zeros((1, 24), np.float32), np.zeros((n, 6), np.float32), frames)
    subprocess.check_call([CPU, str(scene), str(out_c)], timeout=300)
    env = {k: v for k, v in os.environ.items() if not k.startswith("YFH_")}
    env["YFH_ECS_INPUTS"] = "1"
    subprocess.check_call([GPU, str(scene), str(out_g)], timeout=300, env=env)
    cpu = read_out(out_c, n, len(frames), 0, oracle.T68)
    gpu = read_out(out_g, n, len(frames), 2, oracle.T68, gpu=True)
    rebuilds_before = patches_before = 0
    for fi, ((cw, cl, _, _rows, _), (gw, gl, rebuilds, _g_rows, vis)) in enumerate(zip(cpu, gpu)):
        for e, kind, m in frames[fi]:
            if kind == 4:
                mesh_of[e] = int(m.view(np.uint32)[0])
            if kind == 5:
                cam[e] = api.extract_planes(m)
        with np.errstate(all="ignore"):
            assert np.array_equal(gw["m"], cw["m"], equal_nan=True) and np.array_equal(gl["m"], cl["m"], equal_nan=True), f"frame {fi}"
        bounds = np.array([mesh_box[int(k)] for k in mesh_of], np.float32)
        planes = np.stack(cam).astype(np.float32).reshape(2, 24)
        exp_vis, _ = oracle.cull(np.ascontiguousarray(cw), bounds, planes)
        for f in range(2):
            assert np.array_equal(np.sort(vis[f]), np.flatnonzero(exp_vis[f])), f"frame {fi} frustum {f}: visible set"
            assert vis[f].size > 0
        # a Mesh set on an entity that has one moves no row (nothing structural); a first Mesh is an archetype move: the
        # rows of forty tables move to spare slots (a patch, mesh ids included) - or, spare slots lacking, a rebuild
        patches = vis[-1][4]
        assert rebuilds - rebuilds_before + patches - patches_before == (1 if fi in (0, 4) else 0), f"frame {fi}: {rebuilds} rebuilds, {patches} patches"
        rebuilds_before, patches_before = rebuilds, patches


def test_material_and_mesh_draw_buckets_from_the_ecs(oracle, tmp_path):
    """SURVEY 8f N2, the material half (RenderApi.zig:1410-1441: the pipeline is rebound when `material.ref` changes,
    the vertex buffer when the mesh does).  Three materials x three meshes held by Material{ref} / Mesh{ref} components;
    the system gives every pair its own key and the device buckets frustum 0's visible entities by it.  Per frame: the
    buckets are sorted by material, then mesh, each pair once; a bucket's records hold exactly the visible entities of
    that pair, in draw-list order; entities without a mesh come before the first bucket; replaying the reference's
    rebinding rule over the buckets binds every material once.  Mid-run: materials set on entities that have one (no row
    moves), entities that change mesh, entities that lose their place in the frustum through an edit."""
    n, levels = 30_000, 5
    rng = np.random.default_rng(21)
    lo = oracle.synth_level_offsets(n, levels)
    par = oracle.synth_parents(lo)
    local = oracle.compose_batch(oracle.synth_trs(n, int(lo[1])))
    from yume_amd import synth

    mesh_box = {0: [0, 0, 0, 0, 0, 0]}
    for k in (1, 2, 3):
        mesh_box[k] = [-0.5 * k, -0.25 * k - 0.5, -1.0, 0.5 * k, 0.25 * k + 0.5, 1.5]
    mesh_of = (np.arange(n) % 4).astype(np.int64)          # 0: none
    mat_of = ((np.arange(n) // 4) % 3).astype(np.int64)    # meaningful where mesh_of != 0
    planes = np.stack([synth.main_camera_planes(), synth.cascade_planes()[1]]).astype(np.float32).reshape(2, 24)

    def arg_u32(v):
        m = np.zeros(16, np.float32)
        m.view(np.uint32)[0] = v
        return m

    has_mesh = np.flatnonzero(mesh_of != 0)
    remat = rng.choice(has_mesh, 400, replace=False)
    remesh = rng.choice(has_mesh, 300, replace=False)
    frames = [[],
              [(int(e), 8, arg_u32(int(mat_of[e] + 1) % 3)) for e in remat],
              [(int(e), 4, arg_u32(int(mesh_of[e]) % 3 + 1)) for e in remesh],
              [(int(e), 0, oracle.compose(rng.uniform(-5, 5, 3), rng.uniform(-1, 1, 4), [1, 1, 1])) for e in np.unique(rng.integers(0, n, 300))],
              []]
    scene, out_c, out_g = tmp_path / "scene.bin", tmp_path / "cpu.bin", tmp_path / "gpu.bin"
    write_scene(scene, lo, par, local["m"], oracle.identity68()["m"][0], np.zeros((1, 24), np.float32), np.zeros((n, 6), np.float32), frames)
    subprocess.check_call([CPU, str(scene), str(out_c)], timeout=300)
    env = {k: v for k, v in os.environ.items() if not k.startswith("YFH_")}
    env["YFH_MATERIALS"] = "1"
    subprocess.check_call([GPU, str(scene), str(out_g)], timeout=300, env=env)
    cpu = read_out(out_c, n, len(frames), 0, oracle.T68)
    gpu = read_out(out_g, n, len(frames), 2, oracle.T68, gpu=True, buckets=True)
    for fi, ((cw, cl, _, _rows, _), (gw, gl, _reb, _g_rows, vis, (bk, rec))) in enumerate(zip(cpu, gpu)):
        for e, kind, m in frames[fi]:
            if kind == 8:
                mat_of[e] = int(m.view(np.uint32)[0]) % 3
            if kind == 4:
                mesh_of[e] = int(m.view(np.uint32)[0])
        with np.errstate(all="ignore"):
            assert np.array_equal(gw["m"], cw["m"], equal_nan=True) and np.array_equal(gl["m"], cl["m"], equal_nan=True), f"frame {fi}"
        bounds = np.array([mesh_box[int(k)] for k in mesh_of], np.float32)
        exp_vis, _ = oracle.cull(np.ascontiguousarray(cw), bounds, planes)
        visible = np.flatnonzero(exp_vis[0])
        assert np.array_equal(np.sort(vis[0]), visible), f"frame {fi}: visible set"
        assert np.array_equal(np.sort(rec), visible), f"frame {fi}: the records are the visible entities, each once"
        # buckets: sorted by (material, mesh), every pair at most once, together all visible entities that have a mesh
        keys = [(int(a), int(b)) for a, b, _f, _c in bk]
        assert keys == sorted(set(keys)), f"frame {fi}: bucket order {keys}"
        list_pos = {int(e): i for i, e in enumerate(vis[0])}  # the frustum's draw list (slot order)
        covered = 0
        for mat, mesh, first, count in bk:
            ents = rec[first:first + count]
            assert count > 0 and np.all(mesh_of[ents] == mesh + 1) and np.all(mat_of[ents] == mat), f"frame {fi}: bucket ({mat}, {mesh})"
            want = visible[(mesh_of[visible] == mesh + 1) & (mat_of[visible] == mat)]
            assert np.array_equal(np.sort(ents), want), f"frame {fi}: bucket ({mat}, {mesh}) misses or adds entities"
            order = [list_pos[int(e)] for e in ents]
            assert order == sorted(order), f"frame {fi}: bucket ({mat}, {mesh}) is not in draw-list order"
            covered += count
        assert covered == np.count_nonzero(mesh_of[visible] != 0)
        first_bucket = int(min(bk[:, 2])) if len(bk) else rec.size
        assert np.all(mesh_of[rec[:first_bucket]] == 0), f"frame {fi}: entities without a mesh come before the buckets"
        # the reference's rebinding rule over the buckets in the order handed out: one pipeline bind per material present,
        # one vertex-buffer bind per (material, mesh) pair
        mats = [k[0] for k in keys]
        assert 1 + sum(a != b for a, b in zip(mats[1:], mats[:-1])) == len(set(mats))
        assert len(keys) == len({(int(mat_of[e]), int(mesh_of[e]) - 1) for e in visible if mesh_of[e] != 0})
        assert len(keys) >= 6, "the scene was meant to show most of the nine pairs"


@pytest.mark.parametrize("mode", ["incremental", "whole-columns", "no-readback"])
def test_cameras_follow_their_transforms_in_the_same_frame(oracle, tmp_path, mode):
    """SURVEY 8f N4, the camera matrix pipeline (editor/windows/GameWindow.zig:154-178: per Camera entity
    transform.decompose() -> rotation.toEuler() -> camera.updateMatrices).  Two cameras sit ON scene entities deep in
    the hierarchy - a perspective one under a node that moves every frame, an orthographic one - and the system derives
    THIS frame's matrices from the world matrix the frame is about to write.  Per frame: the matrices the components
    hold equal the oracle's pipeline run on the CPU systems' world row of the same frame, and the visible sets equal the
    oracle's cull with those planes.  A system that read the stored view_projection (last frame's) fails both."""
    n, levels = 40_000, 6
    rng = np.random.default_rng(31)
    lo = oracle.synth_level_offsets(n, levels)
    par = oracle.synth_parents(lo)
    local = oracle.compose_batch(oracle.synth_trs(n, int(lo[1])))
    mesh_box = {0: [0, 0, 0, 0, 0, 0]}
    for k in (1, 2, 3):
        mesh_box[k] = [-0.5 * k, -0.25 * k - 0.5, -1.0, 0.5 * k, 0.25 * k + 0.5, 1.5]
    bounds = np.array([mesh_box[int(k)] for k in np.arange(n) % 4], np.float32)
    cam_e = [int(lo[4]) + 17, int(lo[3]) + 5]  # level 4 and level 3
    chain = [cam_e[0]]
    while par[chain[-1]] != 0xFFFFFFFF:
        chain.append(int(par[chain[-1]]))  # the perspective camera's ancestors, parent first ... level 0 last

    def rnd_m(scale=5.0):
        return oracle.compose(rng.uniform(-scale, scale, 3), rng.uniform(-1, 1, 4), rng.uniform(0.8, 1.25, 3))

    frames = [
        [],
        [(chain[1], 0, rnd_m())],                                  # the camera's parent moves
        [(chain[2], 0, rnd_m()), (cam_e[1], 0, rnd_m())],          # its grandparent; the orthographic camera itself
        [],                                                        # nothing moves: matrices must stay what they were
        [(chain[-1], 0, rnd_m(60.0)), (chain[0], 0, rnd_m())],     # its level-0 ancestor AND its own local matrix
        [(chain[0], 1, rnd_m(40.0))],                              # the camera's WORLD matrix is edited (reverse branch)
        [(chain[1], 1, rnd_m(40.0)), (chain[0], 1, rnd_m(40.0))],  # parent world-edited (dirty parent wins: forward branch)
        [(-1, 1, oracle.compose([1, 2, 3], [0, 0.7071068, 0, 0.7071068], [1, 1, 1]))],  # the scene root
        [(chain[1], 0, np.zeros(16, np.float32))],                 # a singular ancestor: decompose fails -> IDENTITY
        [(chain[1], 0, rnd_m())] + [(int(e), 0, rnd_m()) for e in np.unique(rng.integers(0, n, 200))],
    ]
    scene, out_c, out_g = tmp_path / "scene.bin", tmp_path / "cpu.bin", tmp_path / "gpu.bin"
    write_scene(scene, lo, par, local["m"], oracle.identity68()["m"][0], np.zeros((1, 24), np.float32), np.zeros((n, 6), np.float32), frames)
    subprocess.check_call([CPU, str(scene), str(out_c)], timeout=300)
    env = {k: v for k, v in os.environ.items() if not k.startswith("YFH_")}
    env["YFH_CAMERA_PIPELINE"] = f"{cam_e[0]},{cam_e[1]}"
    if mode == "whole-columns":
        env["YFH_PLAIN"] = "1"
    if mode == "no-readback":
        env["YFH_READBACK_OFF_FROM"] = "2"  # frames 2 .. 8 leave their rows on the device: the replay reads them there
    subprocess.check_call([GPU, str(scene), str(out_g)], timeout=300, env=env)
    cpu = read_out(out_c, n, len(frames), 0, oracle.T68)
    gpu = read_out(out_g, n, len(frames), 2, oracle.T68, gpu=True, n_cams=2)
    fovy = np.float32(75.0 * np.pi / 180.0)
    late = 0
    prev_planes = None
    for fi, ((cw, cl, _, _rows, _), (gw, gl, _reb, _g_rows, vis, cams)) in enumerate(zip(cpu, gpu)):
        if mode != "no-readback" or fi < 2 or fi + 1 == len(frames):
            with np.errstate(all="ignore"):
                assert np.array_equal(gw["m"], cw["m"], equal_nan=True) and np.array_equal(gl["m"], cl["m"], equal_nan=True), f"frame {fi}"
        planes = []
        for k, e in enumerate(cam_e):
            view, proj, vp, ok = oracle.camera_update_from_world(k == 1, cw["m"][e], 2.0, fovy, 0.1, 200.0)
            for name, got, exp in zip(("view", "projection", "view_projection"), cams[k], (view, proj, vp)):
                assert np.array_equal(got, exp, equal_nan=True), f"frame {fi} camera {k}: {name} is not this frame's"
            planes.append(oracle.extract_planes(vp))
        if fi == 8:  # world = parent * 0 -> all zero: InvalidMatrix -> Decomposed.IDENTITY -> a camera at the origin
            assert not oracle.decompose(cw["m"][cam_e[0]])[3]
        planes = np.stack(planes).astype(np.float32).reshape(2, 24)
        exp_vis, _ = oracle.cull(np.ascontiguousarray(cw), bounds, planes)
        for f in range(2):
            assert np.array_equal(np.sort(vis[f]), np.flatnonzero(exp_vis[f])), f"frame {fi} frustum {f}: visible set"
        if prev_planes is not None:  # what a system one frame late would have listed
            old_vis, _ = oracle.cull(np.ascontiguousarray(cw), bounds, prev_planes)
            late += int(not np.array_equal(old_vis[0], exp_vis[0]))
        prev_planes = planes
        assert np.flatnonzero(exp_vis[0]).size > 0 or fi == 8
    assert late >= 5, "the script never moved the camera: the test would not tell this frame's frusta from last frame's"


@pytest.mark.parametrize("seed,ecs_inputs,stale_from,rebuilds_by",  # more seeds: YH_STRUCT_SEEDS=40
                         [(k, k % 4 == 3, 5 if k % 4 == 1 else 0, "") for k in range(int(os.environ.get("YH_STRUCT_SEEDS", "4")))]
                         + [(0, False, 0, "lists"), (1, False, 5, "lists"), (3, True, 0, "lists"), (2, False, 0, "walk")])
def test_random_structural_edits_are_patched_or_rebuilt_and_stay_equal_to_the_cpu_systems(oracle, tmp_path, seed, ecs_inputs,
                                                                                          stale_from, rebuilds_by):
    """Editor traffic (Editor.zig:590-636, GameApp.zig:204-215): entities created (under old parents, under parents
    created a frame earlier), leaves deleted, leaves and whole subtrees re-parented (same depth and other depths),
    parents with children deleted, mixed with matrix edits, frame after frame.  The system patches what it can (no walk,
    no rebuild) and rebuilds otherwise; either way every row and every visible set equals the CPU systems' - and a
    burst of creations larger than a level's spare slots must fall back to the rebuild.  stale_from: results stay on the
    device from that frame to the last but one (setReadback(0)): the patches then move rows on the device, flecs' columns
    are compared again after the last frame has brought everything home, the visible sets every frame.
    rebuilds_by: patching off, so that EVERY structural change rebuilds - "lists": from the batches and the observers'
    lists of touched tables (refresh_batches: the query is walked once, in the first frame), "walk": by walking the query
    as rounds 1-2 did (YUME_GPU_SYSTEM_NO_REFRESH)."""
    rng = np.random.default_rng(1000 + seed)
    n_frames = int(os.environ.get("YH_STRUCT_FRAMES", "26"))  # a soak: YH_STRUCT_FRAMES=200
    n0, levels, extra = int(os.environ.get("YH_STRUCT_N", "20000")), 6, 6000 * max(1, n_frames // 26)  # a larger world: YH_STRUCT_N=200000
    n = n0 + extra
    lo0 = oracle.synth_level_offsets(n0, levels)
    par0 = oracle.synth_parents(lo0)
    lo = np.concatenate([lo0, [n]]).astype(np.uint32)  # the harness only sizes arrays with it
    parent = np.concatenate([par0, np.full(extra, 0xFFFFFFFE, np.uint32)])  # the extra entities are created by edits
    local = oracle.compose_batch(oracle.synth_trs(n, int(lo0[1])))
    bounds = oracle.synth_aabb(n)
    planes = np.concatenate([oracle.main_camera_planes().reshape(1, 24), oracle.cascade_planes()[:1]])

    def rnd_m(scale=5.0):
        return oracle.compose(rng.uniform(-scale, scale, 3), rng.uniform(-1, 1, 4), rng.uniform(0.5, 2, 3))

    # a host-side model of the tree, to pick legal edits
    cur_parent = {e: (int(par0[e]) if par0[e] != 0xFFFFFFFF else -1) for e in range(n0)}
    children = {}
    for e, p in cur_parent.items():
        children.setdefault(p, set()).add(e)
    depth = {e: int(np.searchsorted(lo0, e, side="right") - 1) for e in range(n0)}
    alive = set(range(n0))
    next_new = n0
    had_children = np.zeros(n0, bool)
    had_children[par0[par0 != 0xFFFFFFFF]] = True

    # the CPU systems visit tables in creation order and are parent-first only while every entity's own table
    # is older than the tables of its children.  The model keeps the birth of every table an entity's children live in
    # (one per parent; with ecs_inputs two: children with and without a Mesh) and never moves an entity that has, or ever
    # had, children into a table younger than theirs - where the reference itself would read last frame's parent
    # matrix (tools/f5_check.py shows such a case: seed 16, frame 133 of a 150-frame script before this rule).
    def kind_of(e):
        return int(e % 4 != 0) if ecs_inputs else 0

    birth = {}
    for e in range(n0):
        if cur_parent[e] >= 0:
            birth.setdefault((cur_parent[e], kind_of(e)), e)  # born with the first child of that kind (creation order)
    seq = [n0]

    def can_place(x, p):
        tb = birth.get((p, kind_of(x)))
        return all((x, k) not in birth or (tb is not None and tb < birth[(x, k)]) for k in (0, 1))

    def placed(x, p):
        if (p, kind_of(x)) not in birth:
            birth[(p, kind_of(x))] = seq[0]
        seq[0] += 1

    def subtree(e):
        out, stack = [], [e]
        while stack:
            x = stack.pop()
            out.append(x)
            stack.extend(children.get(x, ()))
        return out

    def height(e):
        return max(depth[x] for x in subtree(e)) - depth[e]

    frames = [[], []]
    for fi in range(2, n_frames):
        edits = []
        kind_of_frame = rng.integers(0, 8)
        pool = np.array(sorted(alive))
        if kind_of_frame == 0:  # matrix edits only
            pass
        elif kind_of_frame in (1, 2) and next_new + 40 <= n:  # create a few leaves; some under parents created last frame
            for _ in range(int(rng.integers(1, 30))):
                p = int(rng.choice(pool))
                if depth[p] + 1 >= levels:
                    continue
                e = next_new
                next_new += 1
                edits.append((e, 6, reparent_arg(p)))
                placed(e, p)
                cur_parent[e], depth[e] = p, depth[p] + 1
                children.setdefault(p, set()).add(e)
                alive.add(e)
        elif kind_of_frame == 3:  # delete leaves
            leaves = [int(e) for e in rng.choice(pool, 60) if not children.get(int(e))]
            for e in dict.fromkeys(leaves[:25]):
                edits.append((e, 2, np.zeros(16, np.float32)))
                children[cur_parent[e]].discard(e)
                alive.discard(e)
        elif kind_of_frame == 4:  # re-parent leaves to a parent of the same depth, and to other depths
            for e in dict.fromkeys(int(x) for x in rng.choice(pool, 40)):
                if children.get(e) or e not in alive:
                    continue
                p = int(rng.choice(pool))
                if p == e or depth[p] + 1 >= levels or p not in alive or not can_place(e, p):
                    continue
                edits.append((e, 3, reparent_arg(p)))
                placed(e, p)
                children[cur_parent[e]].discard(e)
                cur_parent[e], depth[e] = p, depth[p] + 1
                children.setdefault(p, set()).add(e)
        elif kind_of_frame == 5:  # re-parent a subtree (not a leaf operation: the system must fall back)
            # (the CPU systems visit tables in creation order and are parent-first only while every table of
            # parents was created before the tables of their children; the subtree's root therefore moves into a table
            # that exists since the scene was built - a parent with children of its own, no deeper than the old parent)
            for _try in range(40):
                e, p = int(rng.choice(pool)), int(rng.choice(pool))
                if not children.get(e) or p in subtree(e) or depth[p] + 1 + height(e) >= levels:
                    continue
                if p >= n0 or not had_children[p] or depth[p] > depth[e] - 1 or depth[e] == 0 or not can_place(e, p):
                    continue
                edits.append((e, 3, reparent_arg(p)))
                placed(e, p)
                children[cur_parent[e]].discard(e)
                shift = depth[p] + 1 - depth[e]
                for x in subtree(e):
                    depth[x] += shift
                cur_parent[e] = p
                children.setdefault(p, set()).add(e)
                break
        elif kind_of_frame == 6:  # delete a parent: its subtree goes with it (flecs' ChildOf cleanup)
            for _try in range(20):
                e = int(rng.choice(pool))
                if not children.get(e) or depth[e] == 0:
                    continue
                edits.append((e, 2, np.zeros(16, np.float32)))
                children[cur_parent[e]].discard(e)
                for x in subtree(e):
                    alive.discard(x)
                break
        elif next_new + 1500 <= n:  # a burst of creations under ONE parent: more than the 1024 spare slots of a small level hold
            p = int(rng.choice([x for x in pool if depth[int(x)] + 1 < levels]))
            for _ in range(int(rng.integers(1100, 1500))):
                e = next_new
                next_new += 1
                edits.append((e, 6, reparent_arg(p)))
                placed(e, p)
                cur_parent[e], depth[e] = p, depth[p] + 1
                children.setdefault(p, set()).add(e)
                alive.add(e)
        pool = np.array(sorted(alive))
        for e in np.unique(rng.choice(pool, int(rng.integers(0, 60)))):
            edits.append((int(e), int(rng.integers(0, 2)), rnd_m()))
        frames.append(edits)
    frames.append([])

    scene, out_c, out_g = tmp_path / "scene.bin", tmp_path / "cpu.bin", tmp_path / "gpu.bin"
    write_scene(scene, lo, parent, local["m"], oracle.identity68()["m"][0], planes, bounds, frames)
    subprocess.check_call([CPU, str(scene), str(out_c)], timeout=300)
    env = {k: v for k, v in os.environ.items() if not k.startswith("YFH_")}
    if ecs_inputs:
        env["YFH_ECS_INPUTS"] = "1"
    if stale_from:
        env["YFH_READBACK_OFF_FROM"] = str(stale_from)
    env.pop("YUME_GPU_SYSTEM_NO_REFRESH", None)
    if rebuilds_by:
        env["YFH_NO_PATCH"] = "1"
        if rebuilds_by == "walk":
            env["YUME_GPU_SYSTEM_NO_REFRESH"] = "1"
    subprocess.check_call([GPU, str(scene), str(out_g)], timeout=300, env=env)
    cpu = read_out(out_c, n, len(frames), 0, oracle.T68)
    gpu = read_out(out_g, n, len(frames), 2 if ecs_inputs else planes.shape[0], oracle.T68, gpu=True)
    patches = rebuilds = stale_patches = 0
    cull_bounds, cull_planes = bounds, planes
    if ecs_inputs:  # the harness' Mesh components (entity e: mesh e % 4, none for 0) and its two Camera entities
        from yume_amd import synth

        box = {0: [0, 0, 0, 0, 0, 0]}
        for k in (1, 2, 3):
            box[k] = [-0.5 * k, -0.25 * k - 0.5, -1.0, 0.5 * k, 0.25 * k + 0.5, 1.5]
        cull_bounds = np.array([box[e % 4] for e in range(n)], np.float32)
        cull_planes = np.stack([synth.main_camera_planes(), synth.cascade_planes()[1]]).astype(np.float32).reshape(2, 24)
    for fi, ((cw, cl, _, rows, _), (gw, gl, g_rebuilds, g_rows, vis)) in enumerate(zip(cpu, gpu)):
        stats = f"(rows up, rows down, walks, skipped, patches) = {vis[-1]}, rebuilds {g_rebuilds}, edits {[(e, k) for e, k, _ in frames[fi]][:6]}"
        host_current = not stale_from or fi < stale_from or fi + 1 == len(frames)  # flecs' columns are the CPU systems'
        with np.errstate(all="ignore"):
            bad = np.flatnonzero(~np.all((gw["m"] == cw["m"]) | (np.isnan(gw["m"]) & np.isnan(cw["m"])), axis=1))
            assert bad.size == 0 or not host_current, f"frame {fi}: {bad.size} world matrices differ, first rows {bad[:8]} {stats}"
            assert np.array_equal(gl["m"], cl["m"], equal_nan=True) or not host_current, f"frame {fi}: local matrices {stats}"
        assert (np.array_equal(gw["dirty"], cw["dirty"]) and np.array_equal(gl["dirty"], cl["dirty"])) or not host_current, f"frame {fi}: dirty bytes {stats}"
        assert g_rows == rows, f"frame {fi}: {g_rows} rows on the device, {rows} in the CPU systems {stats}"
        structural = any(kind in (2, 3, 6) for _e, kind, _m in frames[fi])
        d_patch, d_rebuild = vis[-1][4] - patches, g_rebuilds - rebuilds
        patches, rebuilds = vis[-1][4], g_rebuilds
        if stale_from and stale_from < fi < len(frames) - 1:
            stale_patches += d_patch
        if fi and structural:
            assert d_patch + d_rebuild == 1, f"frame {fi}: a structural change is either patched or rebuilt {stats}"
        elif fi:
            assert d_patch == d_rebuild == 0, f"frame {fi}: nothing structural happened {stats}"
        here = np.flatnonzero(cw["m"].any(axis=1) | cl["m"].any(axis=1))  # rows of entities that exist
        exp_vis, _ = oracle.cull(np.ascontiguousarray(cw[here]), np.ascontiguousarray(cull_bounds[here]), cull_planes)
        for f in range(cull_planes.shape[0]):
            assert np.array_equal(np.sort(vis[f]), here[np.flatnonzero(exp_vis[f])]), f"frame {fi} frustum {f}: visible set {stats}"
    walks = gpu[-1][4][-1][2]
    if rebuilds_by:
        assert patches == 0 and rebuilds >= 8, f"every structural frame was meant to rebuild: {patches} patches, {rebuilds} rebuilds"
        assert (walks == 1) if rebuilds_by == "lists" else (walks == rebuilds), f"{walks} walks of the query for {rebuilds} rebuilds ({rebuilds_by})"
        return
    assert patches >= 3 and rebuilds >= 1, f"the script was meant to exercise the patch path: {patches} patches, {rebuilds} rebuilds"
    assert not stale_from or stale_patches >= 3, f"patches on device-resident rows were meant to happen: {stale_patches}"
    assert walks == 1, f"rebuilds after the first frame take the tables from the observers' lists, not from a walk: {walks} walks"


@pytest.mark.parametrize("no_patch", [False, True], ids=["patched", "rebuilt"])
def test_disabled_entities_leave_and_return_and_reallocated_columns_are_refetched(oracle, tmp_path, no_patch):
    """world.enable(e, false) (ecs.zig:546) gives an entity the Disabled tag: its row moves to a table no query matches,
    the CPU systems stop seeing it (its rows and dirty bytes freeze), and so must the GPU system on the patched path as
    on the rebuilt one - never listed, never written - until it is enabled again.  Same script: fifty children are
    created in an EXISTING table and deleted again in the same frame (the table's columns may have been reallocated
    while its rows and their order are what they were), then an old row of that table is edited: the system has to
    fetch the column pointers afresh (host ASan run: make -C oracle _ref/flecs_gpu_harness_asan)."""
    n0, levels, extra = 20_000, 5, 200
    n = n0 + extra
    rng = np.random.default_rng(11)
    lo0 = oracle.synth_level_offsets(n0, levels)
    par0 = oracle.synth_parents(lo0)
    lo = np.concatenate([lo0, [n]]).astype(np.uint32)
    parent = np.concatenate([par0, np.full(extra, 0xFFFFFFFE, np.uint32)])
    local = oracle.compose_batch(oracle.synth_trs(n, int(lo0[1])))
    bounds = oracle.synth_aabb(n)
    planes = oracle.main_camera_planes().reshape(1, 24)

    def rnd_m(scale=5.0):
        return oracle.compose(rng.uniform(-scale, scale, 3), rng.uniform(-1, 1, 4), rng.uniform(0.5, 2, 3))

    def flag(on):
        m = np.zeros(16, np.float32)
        m.view(np.uint32)[0] = on
        return m

    last = np.arange(int(lo0[-2]), n0)
    off = [int(e) for e in rng.choice(last, 24, replace=False)]
    busy_parent = int(np.bincount(par0[int(lo0[-2]):n0]).argmax())       # the parent with the most children
    siblings = np.flatnonzero(par0 == busy_parent)
    siblings = [int(e) for e in siblings if int(e) not in off]
    burst = list(range(n0, n0 + 50))
    frames = [
        [],
        [(e, 7, flag(0)) for e in off] + [(off[0], 0, rnd_m()), (off[1], 1, rnd_m(30.0))],   # disabled, two of them edited while disabled
        [(int(par0[e]), 0, rnd_m()) for e in off[:6]],                                       # parents of disabled leaves move
        [(e, 7, flag(1)) for e in off[:12]]
        + [(e, 6, reparent_arg(busy_parent)) for e in burst] + [(e, 2, np.zeros(16, np.float32)) for e in burst]
        + [(siblings[0], 0, rnd_m())],
        [(siblings[1], 1, rnd_m(30.0))],
        [(e, 7, flag(1)) for e in off[12:]] + [(off[2], 0, rnd_m())],
        [],
    ]
    scene, out_c, out_g = tmp_path / "scene.bin", tmp_path / "cpu.bin", tmp_path / "gpu.bin"
    write_scene(scene, lo, parent, local["m"], oracle.identity68()["m"][0], planes, bounds, frames)
    subprocess.check_call([CPU, str(scene), str(out_c)], timeout=300)
    env = {k: v for k, v in os.environ.items() if not k.startswith("YFH_")}
    if no_patch:
        env["YFH_NO_PATCH"] = "1"
    subprocess.check_call([GPU, str(scene), str(out_g)], timeout=300, env=env)
    cpu = read_out(out_c, n, len(frames), 0, oracle.T68)
    gpu = read_out(out_g, n, len(frames), 1, oracle.T68, gpu=True)
    enabled = np.zeros(n, bool)
    enabled[:n0] = True
    for fi, ((cw, cl, _, rows, _), (gw, gl, rebuilds, g_rows, vis)) in enumerate(zip(cpu, gpu)):
        for e, kind, m in frames[fi]:
            if kind == 7:
                enabled[e] = bool(m.view(np.uint32)[0])
        stats = f"(rows up, rows down, walks, skipped, patches) = {vis[-1]}, rebuilds {rebuilds}"
        with np.errstate(all="ignore"):
            assert np.array_equal(gw["m"], cw["m"], equal_nan=True) and np.array_equal(gl["m"], cl["m"], equal_nan=True), f"frame {fi} {stats}"
        assert np.array_equal(gw["dirty"], cw["dirty"]) and np.array_equal(gl["dirty"], cl["dirty"]), f"frame {fi}: dirty bytes {stats}"
        assert g_rows == rows == enabled.sum(), f"frame {fi}: {g_rows} rows on the device, {rows} in the CPU systems, {enabled.sum()} enabled {stats}"
        idx = np.flatnonzero(enabled)
        exp_vis, _ = oracle.cull(np.ascontiguousarray(cw[idx]), np.ascontiguousarray(bounds[idx]), planes)
        assert np.array_equal(np.sort(vis[0]), idx[np.flatnonzero(exp_vis[0])]), f"frame {fi}: visible set {stats}"
    # the rows of the two entities edited while disabled kept their dirty bytes until they came back (no system saw them)
    assert cpu[2][0]["dirty"][off[1]] == 1 and cpu[2][1]["dirty"][off[0]] == 1
    if not no_patch:
        assert gpu[-1][2] == 1 and gpu[-1][4][-1][4] == 3, f"three patched frames, no rebuild: {gpu[-1][4][-1]}, rebuilds {gpu[-1][2]}"


def test_readback_switched_off_and_on_again_after_patches(oracle, tmp_path):
    """Patches leave dead batches and dead slots behind; whole columns must still travel correctly when readback is
    switched off afterwards (a structural change is then patched on the device's own rows) and on again."""
    n, levels = 30_000, 6
    rng = np.random.default_rng(5)
    lo = oracle.synth_level_offsets(n, levels)
    par = oracle.synth_parents(lo)
    local = oracle.compose_batch(oracle.synth_trs(n, int(lo[1])))
    bounds = oracle.synth_aabb(n)
    planes = oracle.main_camera_planes().reshape(1, 24)

    def rnd_m(scale=5.0):
        return oracle.compose(rng.uniform(-scale, scale, 3), rng.uniform(-1, 1, 4), rng.uniform(0.5, 2, 3))

    last = np.arange(int(lo[-2]), n)
    doomed = rng.choice(last, 30, replace=False)
    movers = np.setdiff1d(last, doomed)[:12]
    new_par = rng.integers(int(lo[-3]), int(lo[-2]), movers.size)
    frames = [[], [(int(e), 2, np.zeros(16, np.float32)) for e in doomed[:15]],                      # patched
              [(int(e), 3, reparent_arg(int(p))) for e, p in zip(movers, new_par)],                   # patched
              [(int(e), 0, rnd_m()) for e in np.unique(rng.integers(0, int(lo[-2]), 200))],
              [(int(e), 1, rnd_m(50.0)) for e in np.unique(rng.integers(0, int(lo[-2]), 100))],       # readback off from here
              [(int(e), 2, np.zeros(16, np.float32)) for e in doomed[15:]],                           # structural, stale host: patched on the device
              [(-1, 1, oracle.compose([1, 2, 3], [0, 0.7071068, 0, 0.7071068], [2, 2, 2]))],
              []]                                                                                    # readback on: columns come home
    scene, out_c, out_g = tmp_path / "scene.bin", tmp_path / "cpu.bin", tmp_path / "gpu.bin"
    write_scene(scene, lo, par, local["m"], oracle.identity68()["m"][0], planes, bounds, frames)
    subprocess.check_call([CPU, str(scene), str(out_c)], timeout=300)
    env = {k: v for k, v in os.environ.items() if not k.startswith("YFH_")}
    env["YFH_READBACK_OFF_FROM"] = "4"
    subprocess.check_call([GPU, str(scene), str(out_g)], timeout=300, env=env)
    cpu = read_out(out_c, n, len(frames), 0, oracle.T68)
    gpu = read_out(out_g, n, len(frames), 1, oracle.T68, gpu=True)
    assert gpu[2][4][-1][4] == 2, "frames 1 and 2 were meant to be patched"
    for fi, ((cw, cl, _, rows, _), (gw, gl, _reb, g_rows, vis)) in enumerate(zip(cpu, gpu)):
        assert g_rows == rows
        here = np.flatnonzero(cw["m"].any(axis=1) | cl["m"].any(axis=1))
        exp_vis, _ = oracle.cull(np.ascontiguousarray(cw[here]), np.ascontiguousarray(bounds[here]), planes)
        assert np.array_equal(np.sort(vis[0]), here[np.flatnonzero(exp_vis[0])]), f"frame {fi}: visible set (device state)"
        if fi < 4 or fi + 1 == len(frames):  # while readback is on, flecs' columns are the CPU systems'
            with np.errstate(all="ignore"):
                assert np.array_equal(gw["m"], cw["m"], equal_nan=True), f"frame {fi}: world matrices"
                assert np.array_equal(gl["m"], cl["m"], equal_nan=True), f"frame {fi}: local matrices"
            assert np.array_equal(gw["dirty"], cw["dirty"]) and np.array_equal(gl["dirty"], cl["dirty"]), f"frame {fi}: dirty bytes"


@pytest.mark.parametrize("seed", range(int(os.environ.get("YH_STRUCT_SEEDS", "3"))))
def test_subtrees_moved_to_other_depths_patched_equals_rebuilt(oracle, tmp_path, seed):
    """Whole subtrees dragged under parents of any depth, children created under parents created in the same frame,
    leaves deleted, matrices edited: the patched system (rows of every batch below a moved row follow it to another
    level's spare slots) against the same system rebuilding on every structural change (setPatching(0)), row for row and
    visible set for visible set.  (The CPU systems are no yardstick here: moved subtrees leave flecs' table order
    parent-last, where the reference lags a frame - SURVEY F5; both GPU runs are parent-first.)"""
    rng = np.random.default_rng(5000 + seed)
    n0, levels, extra = 12_000, 7, 800
    n = n0 + extra
    lo0 = oracle.synth_level_offsets(n0, levels - 2)  # two levels of headroom for subtrees that move down
    par0 = oracle.synth_parents(lo0)
    lo = np.concatenate([lo0, [n0, n0, n]]).astype(np.uint32)
    parent = np.concatenate([par0, np.full(extra, 0xFFFFFFFE, np.uint32)])
    local = oracle.compose_batch(oracle.synth_trs(n, int(lo0[1])))
    bounds = oracle.synth_aabb(n)
    planes = np.concatenate([oracle.main_camera_planes().reshape(1, 24), oracle.cascade_planes()[:1]])

    def rnd_m(scale=5.0):
        return oracle.compose(rng.uniform(-scale, scale, 3), rng.uniform(-1, 1, 4), rng.uniform(0.5, 2, 3))

    cur_parent = {e: (int(par0[e]) if par0[e] != 0xFFFFFFFF else -1) for e in range(n0)}
    children = {}
    for e, p in cur_parent.items():
        children.setdefault(p, set()).add(e)
    depth = {e: int(np.searchsorted(lo0, e, side="right") - 1) for e in range(n0)}
    alive = set(range(n0))
    next_new = n0

    def subtree(e):
        out, stack = [], [e]
        while stack:
            x = stack.pop()
            out.append(x)
            stack.extend(children.get(x, ()))
        return out

    # the scene's deepest level sets how many levels the device scene has until a rebuild: the first frames push one
    # chain down to the last level so that later moves have room on every level
    frames = [[]]
    deepest = max(depth, key=depth.get)
    chain = []
    for _ in range(levels - 1 - depth[deepest]):
        e = next_new
        next_new += 1
        chain.append((e, 6, reparent_arg(deepest)))
        cur_parent[e], depth[e] = deepest, depth[deepest] + 1
        children.setdefault(deepest, set()).add(e)
        alive.add(e)
        deepest = e
        frames.append([chain[-1]])
    for fi in range(22):
        edits = []
        pool = np.array(sorted(alive))
        what = rng.integers(0, 5)
        if what == 0:  # a subtree under a parent of ANY depth (that keeps it inside the scene's levels)
            for _try in range(40):
                e, p = int(rng.choice(pool)), int(rng.choice(pool))
                sub = subtree(e)
                if depth[e] == 0 or not children.get(e) or p in sub or len(sub) > 400:
                    continue
                if depth[p] + 1 + max(depth[x] for x in sub) - depth[e] >= levels:
                    continue
                edits.append((e, 3, reparent_arg(p)))
                children[cur_parent[e]].discard(e)
                shift = depth[p] + 1 - depth[e]
                for x in sub:
                    depth[x] += shift
                cur_parent[e] = p
                children.setdefault(p, set()).add(e)
                break
        elif what == 1 and next_new + 30 <= n:  # a parent and its children created in ONE frame
            p = int(rng.choice(pool))
            if depth[p] + 2 < levels:
                e = next_new
                next_new += 1
                edits.append((e, 6, reparent_arg(p)))
                cur_parent[e], depth[e] = p, depth[p] + 1
                children.setdefault(p, set()).add(e)
                alive.add(e)
                for _ in range(int(rng.integers(1, 6))):
                    c = next_new
                    next_new += 1
                    edits.append((c, 6, reparent_arg(e)))
                    cur_parent[c], depth[c] = e, depth[e] + 1
                    children.setdefault(e, set()).add(c)
                    alive.add(c)
        elif what == 2:  # leaves deleted
            for e in dict.fromkeys(int(x) for x in rng.choice(pool, 40)):
                if children.get(e) or e not in alive:
                    continue
                edits.append((e, 2, np.zeros(16, np.float32)))
                children[cur_parent[e]].discard(e)
                alive.discard(e)
        elif what == 3:  # leaves under other parents of any depth
            for e in dict.fromkeys(int(x) for x in rng.choice(pool, 30)):
                p = int(rng.choice(pool))
                if children.get(e) or p == e or depth[p] + 1 >= levels or depth[e] == 0:
                    continue
                edits.append((e, 3, reparent_arg(p)))
                children[cur_parent[e]].discard(e)
                cur_parent[e], depth[e] = p, depth[p] + 1
                children.setdefault(p, set()).add(e)
        pool = np.array(sorted(alive))
        for e in np.unique(rng.choice(pool, int(rng.integers(0, 80)))):
            edits.append((int(e), int(rng.integers(0, 2)), rnd_m()))
        frames.append(edits)
    frames.append([])

    scene = tmp_path / "scene.bin"
    write_scene(scene, lo, parent, local["m"], oracle.identity68()["m"][0], planes, bounds, frames)
    outs = {}
    for mode in ("patched", "rebuilt", "patched, results left on the device from frame 3"):
        env = {k: v for k, v in os.environ.items() if not k.startswith("YFH_")}
        if mode == "rebuilt":
            env["YFH_NO_PATCH"] = "1"
        elif mode != "patched":
            env["YFH_READBACK_OFF_FROM"] = "3"
        out = tmp_path / f"{mode}.bin"
        subprocess.check_call([GPU, str(scene), str(out)], timeout=300, env=env)
        outs[mode] = read_out(out, n, len(frames), planes.shape[0], oracle.T68, gpu=True)
    for fi, ((pw, pl, _pr, p_rows, pvis), (rw, rl, _rr, r_rows, rvis)) in enumerate(zip(outs["patched"], outs["rebuilt"])):
        stats = f"patched {pvis[-1]}, rebuilt {rvis[-1]}, edits {[(e, k) for e, k, _ in frames[fi]][:6]}"
        with np.errstate(all="ignore"):
            assert np.array_equal(pw["m"], rw["m"], equal_nan=True), f"frame {fi}: world matrices {stats}"
            assert np.array_equal(pl["m"], rl["m"], equal_nan=True), f"frame {fi}: local matrices {stats}"
        assert np.array_equal(pw["dirty"], rw["dirty"]) and np.array_equal(pl["dirty"], rl["dirty"]), f"frame {fi}: dirty bytes {stats}"
        assert p_rows == r_rows == len(alive_at(frames, fi, n0, parent)), f"frame {fi}: rows {stats}"
        for f in range(planes.shape[0]):
            assert np.array_equal(np.sort(pvis[f]), np.sort(rvis[f])), f"frame {fi} frustum {f}: visible set {stats}"
    assert outs["patched"][-1][4][-1][4] >= 8, f"most structural frames were meant to be patched: {outs['patched'][-1][4][-1]}"
    # the scene is built with two empty levels below its deepest one: subtrees that end up deeper than anything did
    # before are patched as well - nothing here rebuilds after the first frame
    assert outs["patched"][-1][2] == 1, f"{outs['patched'][-1][2]} rebuilds"
    # subtrees that change depth while flecs' columns are stale: every row moves on the device (yh_move_rows); the
    # visible sets are the device's every frame, the columns are compared once the last frame has brought them home
    stale = outs["patched, results left on the device from frame 3"]
    for fi, ((sw, sl, _sr, s_rows, svis), (rw, rl, _rr, r_rows, rvis)) in enumerate(zip(stale, outs["rebuilt"])):
        assert s_rows == r_rows, f"frame {fi}: rows"
        for f in range(planes.shape[0]):
            assert np.array_equal(np.sort(svis[f]), np.sort(rvis[f])), f"frame {fi} frustum {f}: visible set, stale host {svis[-1]}"
        if fi < 3 or fi + 1 == len(frames):
            with np.errstate(all="ignore"):
                assert np.array_equal(sw["m"], rw["m"], equal_nan=True) and np.array_equal(sl["m"], rl["m"], equal_nan=True), f"frame {fi}: matrices, stale host"
            assert np.array_equal(sw["dirty"], rw["dirty"]) and np.array_equal(sl["dirty"], rl["dirty"]), f"frame {fi}: dirty bytes, stale host"
    assert stale[-1][4][-1][4] >= 8, f"most structural frames were meant to be patched on the device: {stale[-1][4][-1]}"


def alive_at(frames, upto, n0, parent):
    """Entities that exist after frame `upto` of a script (creations, deletions with their subtrees)."""
    par = {e: int(parent[e]) for e in range(n0)}
    for fi in range(upto + 1):
        for e, kind, m in frames[fi]:
            if kind == 6 or kind == 3:
                par[e] = int(np.asarray(m, np.float32).view(np.uint32)[0])
            elif kind == 2:
                doomed, grew = {e}, True
                while grew:
                    more = {x for x, p in par.items() if p in doomed and x not in doomed}
                    grew = bool(more)
                    doomed |= more
                for x in doomed:
                    par.pop(x, None)
    return par


def test_a_parent_deleted_with_its_subtree_is_patched(oracle, tmp_path):
    """flecs deletes the tables of a deleted parent's children with them (their ChildOf pair names a dead entity): the
    system hears of the deleted tables, lets their batches die without touching the pointers, and does not rebuild."""
    n, levels = 20_000, 6
    lo = oracle.synth_level_offsets(n, levels)
    par = oracle.synth_parents(lo)
    local = oracle.compose_batch(oracle.synth_trs(n, int(lo[1])))
    bounds = oracle.synth_aabb(n)
    planes = oracle.main_camera_planes().reshape(1, 24)
    victims = [int(lo[2]) + 3, int(lo[3]) + 7, int(lo[1]) + 1]  # parents on three levels, each with a subtree below
    frames = [[], [(victims[0], 2, np.zeros(16, np.float32))], [],
              [(victims[1], 2, np.zeros(16, np.float32)), (victims[2], 2, np.zeros(16, np.float32)),
               (0, 0, oracle.compose([1, 2, 3], [0, 0, 0, 1], [1, 1, 1]))], []]
    scene, out_c, out_g = tmp_path / "scene.bin", tmp_path / "cpu.bin", tmp_path / "gpu.bin"
    write_scene(scene, lo, par, local["m"], oracle.identity68()["m"][0], planes, bounds, frames)
    subprocess.check_call([CPU, str(scene), str(out_c)], timeout=300)
    env = {k: v for k, v in os.environ.items() if not k.startswith("YFH_")}
    subprocess.check_call([GPU, str(scene), str(out_g)], timeout=300, env=env)
    cpu = read_out(out_c, n, len(frames), 0, oracle.T68)
    gpu = read_out(out_g, n, len(frames), 1, oracle.T68, gpu=True)
    for fi, ((cw, cl, _, rows, _), (gw, gl, rebuilds, g_rows, vis)) in enumerate(zip(cpu, gpu)):
        with np.errstate(all="ignore"):
            assert np.array_equal(gw["m"], cw["m"], equal_nan=True) and np.array_equal(gl["m"], cl["m"], equal_nan=True), f"frame {fi}"
        assert np.array_equal(gw["dirty"], cw["dirty"]) and np.array_equal(gl["dirty"], cl["dirty"]), f"frame {fi}: dirty bytes"
        assert g_rows == rows, f"frame {fi}: {g_rows} rows on the device, {rows} in the CPU systems"
        assert rebuilds == 1, f"frame {fi}: {rebuilds} rebuilds {vis[-1]}"
        here = np.flatnonzero(cw["m"].any(axis=1) | cl["m"].any(axis=1))
        exp_vis, _ = oracle.cull(np.ascontiguousarray(cw[here]), np.ascontiguousarray(bounds[here]), planes)
        assert np.array_equal(np.sort(vis[0]), here[np.flatnonzero(exp_vis[0])]), f"frame {fi}: visible set"
    assert gpu[-1][4][-1][4] == 2 and gpu[-1][3] < n - 3, f"two patched frames, subtrees gone: {gpu[-1][4][-1]}, rows {gpu[-1][3]}"


def test_dead_slots_are_reused_by_later_patches(oracle, tmp_path):
    """A leaf that hops between two parents with some hundred children each moves ~600 rows per frame (both tables'
    rows are placed anew); the runs of slots that die are handed out again, so forty such frames need no rebuild
    although the level's spare slots would be gone after three."""
    n, levels = 6000, 3
    lo = np.array([0, 10, 30, n], np.uint32)
    par = np.full(n, 0xFFFFFFFF, np.uint32)
    par[10:30] = np.arange(20) // 2
    par[30:] = 10 + (np.arange(n - 30) * 20 // (n - 30))  # ~300 children under each of the 20 parents of level 1
    local = oracle.compose_batch(oracle.synth_trs(n, 10))
    bounds = oracle.synth_aabb(n)
    planes = oracle.main_camera_planes().reshape(1, 24)
    hopper, a, b = n - 1, 10, 11
    frames = [[]] + [[(hopper, 3, reparent_arg(a if k % 2 == 0 else b))] for k in range(40)] + [[]]
    scene, out_c, out_g = tmp_path / "scene.bin", tmp_path / "cpu.bin", tmp_path / "gpu.bin"
    write_scene(scene, lo, par, local["m"], oracle.identity68()["m"][0], planes, bounds, frames)
    subprocess.check_call([CPU, str(scene), str(out_c)], timeout=300)
    env = {k: v for k, v in os.environ.items() if not k.startswith("YFH_")}
    subprocess.check_call([GPU, str(scene), str(out_g)], timeout=300, env=env)
    cpu = read_out(out_c, n, len(frames), 0, oracle.T68)
    gpu = read_out(out_g, n, len(frames), 1, oracle.T68, gpu=True)
    for fi, ((cw, cl, _, rows, _), (gw, gl, rebuilds, g_rows, vis)) in enumerate(zip(cpu, gpu)):
        with np.errstate(all="ignore"):
            assert np.array_equal(gw["m"], cw["m"], equal_nan=True) and np.array_equal(gl["m"], cl["m"], equal_nan=True), f"frame {fi}"
        assert g_rows == rows == n
        exp_vis, _ = oracle.cull(np.ascontiguousarray(cw), bounds, planes)
        assert np.array_equal(np.sort(vis[0]), np.flatnonzero(exp_vis[0])), f"frame {fi}: visible set"
    assert gpu[-1][2] == 1 and gpu[-1][4][-1][4] == 40, f"rebuilds {gpu[-1][2]}, (up, down, walks, skipped, patches) {gpu[-1][4][-1]}"


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["readback", "no-readback", "ecs-inputs"])
def test_a_patch_of_hundreds_of_tables_goes_up_in_one_call_per_array(oracle, tmp_path, mode):
    """Editor traffic in bulk: 700 leaves re-parented in one frame (1400 tables touched: above the 256 from which a patch
    takes its level's untouched tail first), then 500 of them deleted and matrices edited, then 400 entities created under
    as many parents.  Each such frame is ONE patch whose alive runs, rows, boxes or mesh ids and parent indices travel in one
    library call per array; rows and visible sets equal the CPU systems' every frame.  "no-readback": the rows change
    slots on the device (yh_move_rows), flecs' columns are compared once the last frame has brought them home."""
    n0, levels, extra = 60_000, 6, 400
    n = n0 + extra
    rng = np.random.default_rng(77)
    lo0 = oracle.synth_level_offsets(n0, levels)
    par0 = oracle.synth_parents(lo0)
    lo = np.concatenate([lo0, [n]]).astype(np.uint32)  # the harness only sizes arrays with it
    parent = np.concatenate([par0, np.full(extra, 0xFFFFFFFE, np.uint32)])  # the extra entities are created by edits (kind 6)
    local = oracle.compose_batch(oracle.synth_trs(n, int(lo0[1])))
    bounds = oracle.synth_aabb(n)
    planes = oracle.main_camera_planes().reshape(1, 24)

    def rnd_m(scale=5.0):
        return oracle.compose(rng.uniform(-scale, scale, 3), rng.uniform(-1, 1, 4), rng.uniform(0.5, 2, 3))

    last = np.arange(int(lo0[-2]), n0)
    movers = rng.choice(last, 700, replace=False)
    new_par = rng.integers(int(lo0[-3]), int(lo0[-2]), movers.size)
    doomed = movers[:500]
    born = np.arange(n0, n)
    born_par = rng.choice(np.arange(int(lo0[-3]), int(lo0[-2])), extra, replace=False)
    frames = [
        [],
        [(int(e), 3, reparent_arg(int(p))) for e, p in zip(movers, new_par)],
        [(int(e), 2, np.zeros(16, np.float32)) for e in doomed] + [(int(e), 0, rnd_m()) for e in rng.choice(int(lo0[-2]), 300, replace=False)],
        [(int(e), 6, reparent_arg(int(p))) for e, p in zip(born, born_par)],
        [(int(e), 1, rnd_m(30.0)) for e in born[:50]],
        [],
    ]
    scene, out_c, out_g = tmp_path / "scene.bin", tmp_path / "cpu.bin", tmp_path / "gpu.bin"
    write_scene(scene, lo, parent, local["m"], oracle.identity68()["m"][0], planes, bounds, frames)
    subprocess.check_call([CPU, str(scene), str(out_c)], timeout=300)
    env = {k: v for k, v in os.environ.items() if not k.startswith("YFH_")}
    if mode == "no-readback":
        env["YFH_READBACK_OFF_FROM"] = "1"
    if mode == "ecs-inputs":
        env["YFH_ECS_INPUTS"] = "1"
    subprocess.check_call([GPU, str(scene), str(out_g)], timeout=300, env=env)
    nf = 2 if mode == "ecs-inputs" else 1
    cpu = read_out(out_c, n, len(frames), 0, oracle.T68)
    gpu = read_out(out_g, n, len(frames), nf, oracle.T68, gpu=True)
    cull_bounds, cull_planes = bounds, planes
    if mode == "ecs-inputs":
        from yume_amd import synth

        box = {0: [0, 0, 0, 0, 0, 0]}
        for k in (1, 2, 3):
            box[k] = [-0.5 * k, -0.25 * k - 0.5, -1.0, 0.5 * k, 0.25 * k + 0.5, 1.5]
        cull_bounds = np.array([box[e % 4] for e in range(n)], np.float32)
        cull_planes = np.stack([synth.main_camera_planes(), synth.cascade_planes()[1]]).astype(np.float32).reshape(2, 24)
    for fi, ((cw, cl, _, rows, _), (gw, gl, rebuilds, g_rows, vis)) in enumerate(zip(cpu, gpu)):
        stats = f"(rows up, rows down, walks, skipped, patches) = {vis[-1]}, rebuilds {rebuilds}"
        host_current = mode != "no-readback" or fi == 0 or fi + 1 == len(frames)
        if host_current:
            with np.errstate(all="ignore"):
                assert np.array_equal(gw["m"], cw["m"], equal_nan=True) and np.array_equal(gl["m"], cl["m"], equal_nan=True), f"frame {fi} {stats}"
            assert np.array_equal(gw["dirty"], cw["dirty"]) and np.array_equal(gl["dirty"], cl["dirty"]), f"frame {fi}: dirty bytes {stats}"
        assert g_rows == rows, f"frame {fi}: {g_rows} rows on the device, {rows} in the CPU systems {stats}"
        here = np.flatnonzero(cw["m"].any(axis=1) | cl["m"].any(axis=1))
        exp_vis, _ = oracle.cull(np.ascontiguousarray(cw[here]), np.ascontiguousarray(cull_bounds[here]), cull_planes)
        for f in range(cull_planes.shape[0]):
            assert np.array_equal(np.sort(vis[f]), here[np.flatnonzero(exp_vis[f])]), f"frame {fi} frustum {f}: visible set {stats}"
    assert gpu[-1][2] == 1 and gpu[-1][4][-1][4] == 3, f"three patched frames, no rebuild: {gpu[-1][4][-1]}, rebuilds {gpu[-1][2]}"


@pytest.mark.gpu
def test_frames_with_nothing_to_do_launch_nothing_and_change_nothing(oracle, tmp_path):
    """Idle frames: no structural change, no row set, the same root, bounds and frusta as the frame before - the system
    launches nothing and the last frame's rows and lists stand.  The run's output must be byte for byte what the same run
    produces with the shortcut switched off (YUME_GPU_SYSTEM_NO_IDLE_SKIP), and exactly the frames without edits that
    follow a frame of the same kind count as idle: not the first frame, not a frame with an edit, a deletion or a moved
    root, not the frame after a structural change's patch either way."""
    n, levels = 20_000, 5
    rng = np.random.default_rng(9)
    lo = oracle.synth_level_offsets(n, levels)
    par = oracle.synth_parents(lo)
    local = oracle.compose_batch(oracle.synth_trs(n, int(lo[1])))
    bounds = oracle.synth_aabb(n)
    planes = oracle.main_camera_planes().reshape(1, 24)
    m = oracle.compose([1, 2, 3], [0, 0, 0, 1], [1, 1, 1])
    leaf = n - 1
    frames = [[], [], [(17, 0, m)], [], [], [(-1, 1, oracle.compose([1, 2, 3], [0, 0.7071068, 0, 0.7071068], [2, 2, 2]))], [],
              [(leaf, 2, np.zeros(16, np.float32))], [], [(int(lo[2]) + 3, 1, m)], []]
    idle_expected = [0, 1, 1, 2, 3, 3, 4, 4, 5, 5, 6]  # running count after every frame
    scene = tmp_path / "scene.bin"
    write_scene(scene, lo, par, local["m"], oracle.identity68()["m"][0], planes, bounds, frames)
    outs = {}
    for skip in (True, False):
        env = {k: v for k, v in os.environ.items() if not k.startswith("YFH_")}
        env["YFH_TIMES"] = "1"
        env.pop("YUME_GPU_SYSTEM_NO_IDLE_SKIP", None)
        if not skip:
            env["YUME_GPU_SYSTEM_NO_IDLE_SKIP"] = "1"
        out = tmp_path / f"gpu_{int(skip)}.bin"
        r = subprocess.run([GPU, str(scene), str(out)], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        idle = [int(line.rsplit(" ", 1)[1]) for line in r.stderr.splitlines() if line.startswith("stages ")]
        assert idle == (idle_expected if skip else [0] * len(frames)), f"idle frames {idle}"
        outs[skip] = open(out, "rb").read()
    # (the traffic counters at the end of every frame's record are equal too: an idle frame moves no rows either way)
    assert outs[True] == outs[False], "the shortcut changed what the frames produce"
    out_c = tmp_path / "cpu.bin"
    subprocess.check_call([CPU, str(scene), str(out_c)], timeout=300)
    cpu = read_out(out_c, n, len(frames), 0, oracle.T68)
    gpu = read_out(tmp_path / "gpu_1.bin", n, len(frames), 1, oracle.T68, gpu=True)
    for fi, ((cw, cl, _, _rows, _), (gw, gl, _r, _g, vis)) in enumerate(zip(cpu, gpu)):
        with np.errstate(all="ignore"):
            assert np.array_equal(gw["m"], cw["m"], equal_nan=True) and np.array_equal(gl["m"], cl["m"], equal_nan=True), f"frame {fi}"
        here = np.flatnonzero(cw["m"].any(axis=1) | cl["m"].any(axis=1))
        exp_vis, _ = oracle.cull(np.ascontiguousarray(cw[here]), np.ascontiguousarray(bounds[here]), planes)
        assert np.array_equal(np.sort(vis[0]), here[np.flatnonzero(exp_vis[0])]), f"frame {fi}: visible set"


@pytest.mark.gpu
def test_frames_in_which_only_a_camera_or_a_mesh_changed_cull_without_running_the_rule(oracle, tmp_path):
    """The common frame: nothing moved, a camera did (or a MeshComponent was set on an entity that has one).  The system
    passes YH_FRAME_CULL_ONLY: one launch culls the world matrices the device holds.  Exactly those frames count, the run's
    output is byte for byte what it is with the shortcuts switched off (every frame runs the rule), and the visible sets
    are the oracle's for each frame's cameras and meshes."""
    from yume_amd import api, synth

    n, levels = 30_000, 6
    rng = np.random.default_rng(21)
    lo = oracle.synth_level_offsets(n, levels)
    par = oracle.synth_parents(lo)
    local = oracle.compose_batch(oracle.synth_trs(n, int(lo[1])))
    mesh_box = {0: [0, 0, 0, 0, 0, 0]}
    for k in (1, 2, 3):
        mesh_box[k] = [-0.5 * k, -0.25 * k - 0.5, -1.0, 0.5 * k, 0.25 * k + 0.5, 1.5]
    mesh_of = (np.arange(n) % 4).astype(np.int64)
    cam = [synth.main_camera_planes(), synth.cascade_planes()[1]]

    def arg_u32(v):
        m = np.zeros(16, np.float32)
        m.view(np.uint32)[0] = v
        return m

    def vp(k):
        return api.mat4_mul(api.perspective(np.radians(70.0), 1.6, 0.1, 250.0), api.look_at([3.0 * k, 30.0, 120.0 - 5 * k], [0, 0, 0], [0, 1, 0]))

    remesh = np.unique(rng.integers(0, n, 400))
    remesh = remesh[mesh_of[remesh] != 0][:200]
    m = oracle.compose([1, 2, 3], [0, 0, 0, 1], [1, 1, 1])
    frames = [[], [(0, 5, vp(1))], [(0, 5, vp(2)), (1, 5, vp(7))], [], [(int(lo[3]) + 5, 0, m)], [(0, 5, vp(3))],
              [(int(e), 4, arg_u32(int(mesh_of[e]) % 3 + 1)) for e in remesh], [(0, 5, vp(4)), (77, 1, m)], [(1, 5, vp(5))], []]
    cull_only_expected = [0, 1, 2, 2, 2, 3, 4, 4, 5, 5]  # running count after every frame
    scene, out_c = tmp_path / "scene.bin", tmp_path / "cpu.bin"
    write_scene(scene, lo, par, local["m"], oracle.identity68()["m"][0], np.zeros((1, 24), np.float32), np.zeros((n, 6), np.float32), frames)
    outs = {}
    for shortcut in (True, False):
        env = {k: v for k, v in os.environ.items() if not k.startswith("YFH_")}
        env.update(YFH_TIMES="1", YFH_ECS_INPUTS="1")
        env.pop("YUME_GPU_SYSTEM_NO_IDLE_SKIP", None)
        if not shortcut:
            env["YUME_GPU_SYSTEM_NO_IDLE_SKIP"] = "1"
        out = tmp_path / f"gpu_{int(shortcut)}.bin"
        r = subprocess.run([GPU, str(scene), str(out)], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        got = [int(line.split("cull-only frames so far ")[1].split(";")[0]) for line in r.stderr.splitlines() if line.startswith("stages ")]
        assert got == (cull_only_expected if shortcut else [0] * len(frames)), f"cull-only frames {got}"
        outs[shortcut] = open(out, "rb").read()
    assert outs[True] == outs[False], "cull-only frames changed what the frames produce"
    subprocess.check_call([CPU, str(scene), str(out_c)], timeout=300)
    cpu = read_out(out_c, n, len(frames), 0, oracle.T68)
    gpu = read_out(tmp_path / "gpu_1.bin", n, len(frames), 2, oracle.T68, gpu=True)
    for fi, ((cw, cl, _, _rows, _), (gw, gl, _rb, _g_rows, vis)) in enumerate(zip(cpu, gpu)):
        for e, kind, mm in frames[fi]:
            if kind == 4:
                mesh_of[e] = int(mm.view(np.uint32)[0])
            if kind == 5:
                cam[e] = api.extract_planes(mm)
        with np.errstate(all="ignore"):
            assert np.array_equal(gw["m"], cw["m"], equal_nan=True) and np.array_equal(gl["m"], cl["m"], equal_nan=True), f"frame {fi}"
        bounds = np.array([mesh_box[int(k)] for k in mesh_of], np.float32)
        exp_vis, _ = oracle.cull(np.ascontiguousarray(cw), bounds, np.stack(cam).astype(np.float32).reshape(2, 24))
        for f in range(2):
            assert np.array_equal(np.sort(vis[f]), np.flatnonzero(exp_vis[f])), f"frame {fi} frustum {f}: visible set"
