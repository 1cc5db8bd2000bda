"""GPU parity of the drop-in column path: flecs 68-byte {Mat4, dirty} columns in, the literal three-branch
rule of TransformSyncSystem.zig:60-73 + the PostTransform dirty reset on the device, columns out — checked
frame after frame against the oracle's reference-faithful AoS system."""
import os

import numpy as np
import pytest

from .test_gpu_parity import assert_same_values

pytestmark = pytest.mark.gpu


def build(oracle, n, n_levels, seed=7):
    lo = oracle.synth_level_offsets(n, n_levels)
    par = oracle.synth_parents(lo, seed=seed)
    trs = oracle.synth_trs(n, int(lo[1]), seed=seed)
    bounds = oracle.synth_aabb(n, seed=seed)
    local = oracle.compose_batch(trs)
    world = np.zeros(n, oracle.T68)
    world["m"][:, [0, 5, 10, 15]] = 1
    world["dirty"] = 1
    return lo, par, bounds, local, world


def test_aos68_roundtrip(ctx, oracle):
    from yume_amd import Scene, _lib

    for n in (1, 255, 256, 257, 100_003):
        rng = np.random.default_rng(n)
        col = np.zeros(n, oracle.T68)
        col["m"] = rng.standard_normal((n, 16)).astype(np.float32)
        col["dirty"] = rng.integers(0, 2, n)
        col["pad"] = rng.integers(0, 255, (n, 3))  # padding bytes are whatever flecs left there
        with Scene(ctx, n, n_frusta=0, bounds_kind=0, local_kind=_lib.YH_LOCAL_MAT4) as s:
            s.upload_flat()
            s.upload_local_aos68(col)
            s.upload_world_aos68(col[::-1].copy())
            l = s.local_aos68()
            w = s.world_aos68()
            assert np.array_equal(l["m"].view(np.uint32), col["m"].view(np.uint32))
            assert np.array_equal(l["dirty"], col["dirty"]) and not l["pad"].any()
            assert np.array_equal(w["m"].view(np.uint32), col["m"][::-1].view(np.uint32))
            assert np.array_equal(w["dirty"], col["dirty"][::-1])
            # partial ranges
            if n > 300:
                s.upload_local_aos68(col[:100], first=200)
                assert np.array_equal(s.local_aos68(200, 100)["m"], col["m"][:100])
                assert np.array_equal(s.local_aos68(0, 200)["m"], col["m"][:200])


def test_dirty_frames_match_reference_system(ctx, oracle):
    """Six frames of edits: nothing dirty, local edits, world-only edits (inverse branch), a singular parent,
    re-rooting — device columns must equal the oracle's after every frame, dirty bytes included."""
    from yume_amd import Scene, _lib

    n = 60_000
    lo, par, bounds, local, world = build(oracle, n, 6)
    planes = oracle.main_camera_planes().reshape(1, 24)
    root = oracle.identity68()
    rng = np.random.default_rng(11)
    mins = np.ascontiguousarray(bounds[:, 0:3].T)
    maxs = np.ascontiguousarray(bounds[:, 3:6].T)

    with Scene(ctx, n, n_levels=6, n_frusta=1, local_kind=_lib.YH_LOCAL_MAT4) as s:
        s.upload_hierarchy(par, lo)
        s.upload_aabb(mins, maxs)
        s.upload_local_aos68(local)
        s.upload_world_aos68(world)

        def step(tag):
            oracle.frame_aos68(world, local, par, lo, root)
            s.frame(planes, flags=0)
            gw, gl = s.world_aos68(), s.local_aos68()
            with np.errstate(all="ignore"):
                assert_same_values(gw["m"], world["m"], f"{tag}: world")
                assert_same_values(gl["m"], local["m"], f"{tag}: local")
            assert not gw["dirty"].any() and not gl["dirty"].any(), tag
            vis, bnd = oracle.cull(world, bounds, planes)
            assert np.array_equal(s.draw_list(0), np.flatnonzero(vis[0]).astype(np.uint32)), tag
            assert s.counts()[1][0] == np.count_nonzero(bnd[0])

        step("frame 0 (everything dirty)")
        step("frame 1 (nothing dirty)")

        # local edits on random entities (inspector drag: Mat4.recompose + modified(LocalTransform))
        idx = np.unique(rng.integers(0, n, 500))
        for e in idx:
            local["m"][e] = oracle.compose(rng.uniform(-5, 5, 3), rng.uniform(-1, 1, 4), rng.uniform(0.5, 2, 3))
        local["dirty"][idx] = 1
        s.upload_local_aos68(local)  # whole column, dirty bytes travel with it
        step("frame 2 (local edits)")

        # world-only edits (gizmo: SceneWindow.zig:334-351) -> inverse branch re-derives the local
        idx = np.unique(rng.integers(int(lo[1]), n, 300))
        for e in idx:
            world["m"][e] = oracle.compose(rng.uniform(-50, 50, 3), rng.uniform(-1, 1, 4), rng.uniform(0.5, 2, 3))
        world["dirty"][idx] = 1
        s.upload_world_aos68(world)
        step("frame 3 (world edits)")

        # a singular parent: its children edited in world space take IDENTITY as the inverse
        p = int(lo[1]) + 5
        local["m"][p] = 0
        local["dirty"][p] = 1
        s.upload_local_aos68(local[p : p + 1], first=p)
        step("frame 4 (singular parent, forward)")
        kids = np.flatnonzero(par == p)
        if kids.size:
            world["dirty"][kids] = 1
            s.mark_dirty(world=True, first=int(kids[0]), count=int(kids.size))  # siblings are contiguous
            step("frame 5 (children of a singular parent, inverse -> identity)")

        # re-rooting: root WorldTransform set -> dirty -> everything recomputes
        root["m"][0] = oracle.compose([1, 2, 3], [0, 0.7071068, 0, 0.7071068], [2, 2, 2])
        root["dirty"] = 1
        s.set_root(root["m"][0])
        step("frame 6 (root moved)")


@pytest.mark.parametrize("seed", range(int(os.environ.get("YH_FUZZ_SEEDS", "8"))))
def test_random_edit_scripts_on_flecs_columns(ctx, oracle, seed):
    """Seeded random differential test of the literal three-branch rule: random level shapes (1..10 uneven levels, so
    single-level and chain launches, XCD-ordered and plain ones all occur), and per frame a random mix of local edits,
    world-only edits (inverse branch), singular matrices and root moves, uploaded as sparse rows; change tracking on in
    half the runs.  After every frame both device columns, dirty bytes, the draw list and (when tracked) the changed
    rows must be what the oracle's CPU systems leave."""
    from yume_amd import Scene, _lib

    rng = np.random.default_rng(5000 + seed)
    n_levels = int(rng.integers(1, 11))
    sizes = [int(rng.choice([rng.integers(1, 70), rng.integers(70, 3000), rng.integers(3000, 60_000)], p=[0.2, 0.4, 0.4]))
             for _ in range(n_levels)]
    lo = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    n = int(lo[-1])
    par = oracle.synth_parents(lo, seed=seed)
    bounds = oracle.synth_aabb(n, seed=seed)
    local = oracle.compose_batch(oracle.synth_trs(n, int(lo[1]), seed=seed))
    world = np.zeros(n, oracle.T68)
    world["m"][:, [0, 5, 10, 15]] = 1
    world["dirty"] = 1
    root = oracle.identity68()
    planes = oracle.main_camera_planes().reshape(1, 24)
    track = seed % 3  # off, both lists, YH_TRACK_REWRITTEN_LOCALS (the list of yh_download_changed_aos68 is the same in both)
    other_cams = oracle.cascade_planes()

    def rnd_m(scale):
        return oracle.compose(rng.uniform(-scale, scale, 3), rng.uniform(-1, 1, 4), rng.uniform(0.5, 2, 3))

    with Scene(ctx, n, n_levels=n_levels, n_frusta=1, local_kind=_lib.YH_LOCAL_MAT4) as s:
        s.upload_hierarchy(par, lo)
        s.upload_aabb(np.ascontiguousarray(bounds[:, 0:3].T), np.ascontiguousarray(bounds[:, 3:6].T))
        s.upload_local_aos68(local)
        s.upload_world_aos68(world)
        s.track_changes(track)
        for frame in range(7):
            if frame >= 1:
                kind = rng.random()
                if kind < 0.15:
                    pass  # nothing moves
                else:
                    k = int(rng.integers(1, max(2, n // 20)))
                    idx = np.unique(rng.integers(0, n, k)).astype(np.uint32)
                    for e in idx:
                        local["m"][e] = 0 if rng.random() < 0.02 else rnd_m(5)
                    local["dirty"][idx] = 1
                    s.upload_local_rows(idx, local[idx])
                    if rng.random() < 0.6:  # world-only edits elsewhere (gizmo drags)
                        widx = np.setdiff1d(np.unique(rng.integers(0, n, max(1, k // 3))).astype(np.uint32), idx)
                        for e in widx:
                            world["m"][e] = rnd_m(50)
                        world["dirty"][widx] = 1
                        if widx.size:
                            s.upload_world_rows(widx, world[widx])
                    if rng.random() < 0.2:
                        root["m"][0] = rnd_m(3)
                        root["dirty"] = 1
                        s.set_root(root["m"][0])
            before_dirty = (local["dirty"] | world["dirty"]).astype(bool)
            w0, l0, ld0 = world["m"].copy(), local["m"].copy(), local["dirty"].astype(bool)
            oracle.frame_aos68(world, local, par, lo, root)
            s.frame(planes, flags=0)
            gw, gl = s.world_aos68(), s.local_aos68()
            with np.errstate(all="ignore"):
                assert_same_values(gw["m"], world["m"], f"seed {seed} frame {frame}: world")
                assert_same_values(gl["m"], local["m"], f"seed {seed} frame {frame}: local")
            assert not gw["dirty"].any() and not gl["dirty"].any()
            with np.errstate(all="ignore"):
                vis, bnd = oracle.cull(world, bounds, planes)
            assert np.array_equal(s.draw_list(0), np.flatnonzero(vis[0]).astype(np.uint32)), (seed, frame)
            if track:
                slots, wr, lr = s.changed()
                assert np.all(np.diff(slots.astype(np.int64)) > 0)
                touched = np.zeros(n, bool)
                touched[slots] = True
                # every row that was dirty before the sweep or whose world matrix changed must be reported
                with np.errstate(all="ignore"):
                    moved = ~np.all((w0 == world["m"]) | (np.isnan(w0) & np.isnan(world["m"])), axis=1)
                assert np.all(touched[before_dirty]) and np.all(touched[moved]), (seed, frame)
                with np.errstate(all="ignore"):
                    assert_same_values(wr["m"], world["m"][slots], "changed world rows")
                    assert_same_values(lr["m"], local["m"][slots], "changed local rows")
                if track == 2:  # the LocalTransform list: exactly the rows whose matrix the frame rewrote
                    sl, _rows = s.changed_rows(1)
                    with np.errstate(all="ignore"):
                        rewritten = ~np.all((l0 == local["m"]) | (np.isnan(l0) & np.isnan(local["m"])), axis=1)
                    assert np.all(np.isin(np.flatnonzero(rewritten), sl)), (seed, frame)
                    assert not np.any(ld0[sl]), f"seed {seed} frame {frame}: a row that was only set came back"
            if rng.random() < 0.4:  # a camera moved, nothing else did: the same rows, another list, nothing reported
                cam = other_cams[int(rng.integers(0, other_cams.shape[0]))].reshape(1, 24)
                s.frame(cam, _lib.YH_FRAME_CULL_ONLY | (_lib.YH_FRAME_ASYNC if rng.random() < 0.5 else 0))
                with np.errstate(all="ignore"):
                    vis, bnd = oracle.cull(world, bounds, cam)
                assert np.array_equal(s.draw_list(0), np.flatnonzero(vis[0]).astype(np.uint32)), (seed, frame, "cull only")
                assert s.counts()[1][0] == np.count_nonzero(bnd[0])
                if track:
                    assert s.changed_counts() == (0, 0)


@pytest.mark.parametrize("mode", [1, 2])
def test_sparse_rows_and_change_tracking(ctx, oracle, mode):
    """only dirty rows go up, only rows the frame touched come back.  A host mirror patched with the
    returned rows must equal the oracle's full columns after every frame, and untouched frames return nothing.
    mode 2 (YH_TRACK_REWRITTEN_LOCALS): the LocalTransform list holds the reverse-branch rows alone - the host clears
    the flags of the local rows it sent itself - and the mirror still equals the oracle's columns."""
    from yume_amd import Scene, _lib

    n = 80_000
    lo, par, bounds, local, world = build(oracle, n, 7)
    planes = oracle.main_camera_planes().reshape(1, 24)
    root = oracle.identity68()
    rng = np.random.default_rng(5)
    with Scene(ctx, n, n_levels=7, n_frusta=1, local_kind=_lib.YH_LOCAL_MAT4) as s:
        with pytest.raises(_lib.YumeHipError) as e:
            s.changed()
        assert e.value.code == _lib.YH_ERR_STATE  # tracking is off
        s.upload_hierarchy(par, lo)
        s.upload_aabb(np.ascontiguousarray(bounds[:, 0:3].T), np.ascontiguousarray(bounds[:, 3:6].T))
        s.upload_local_aos68(local)
        s.upload_world_aos68(world)
        with pytest.raises(_lib.YumeHipError) as e:
            s.track_changes(3)
        assert e.value.code == _lib.YH_ERR_BAD_ARG
        s.track_changes(mode)
        mirror_w, mirror_l = world.copy(), local.copy()  # what the host (flecs) holds

        def step(tag, expect_changed=None):
            before_w, before_l = world.copy(), local.copy()
            oracle.frame_aos68(world, local, par, lo, root)
            s.frame(planes, flags=0)
            slots, rw, rl = s.changed()
            touched = np.flatnonzero(before_w["dirty"] | before_l["dirty"] | (world["dirty"] != before_w["dirty"]) |
                                     (world["m"] != before_w["m"]).any(axis=1) | (local["m"] != before_l["m"]).any(axis=1))
            oracle_changed = np.flatnonzero(np.any(world["m"].view(np.uint32) != before_w["m"].view(np.uint32), axis=1) |
                                            np.any(local["m"].view(np.uint32) != before_l["m"].view(np.uint32), axis=1))
            assert np.all(np.diff(slots.astype(np.int64)) > 0), tag
            assert np.isin(oracle_changed, slots).all(), f"{tag}: a changed row is missing from the list"
            assert np.isin(touched, slots).all(), tag
            if expect_changed is not None:
                assert slots.size == expect_changed, f"{tag}: {slots.size} rows returned"
            # the same rows as two streamed lists (yh_download_changed_rows_aos68): WorldTransform rows = the list above;
            # LocalTransform rows = exactly the rows whose local flag was set or that took the reverse branch
            pieces = []
            sw, rw2 = s.changed_rows(0, on_ready=pieces.append)
            sl, rl2 = s.changed_rows(1)
            assert (len(sw), len(sl)) == s.changed_counts(), tag
            assert np.array_equal(sw, slots) and np.array_equal(rw2, rw), tag
            assert pieces == sorted(pieces) and (not len(sw) or pieces[-1] == len(sw)), f"{tag}: pieces {pieces}"
            wd_after = np.zeros(n, bool)
            reverse = np.zeros(n, bool)
            for lv in range(len(lo) - 1):
                a, b = int(lo[lv]), int(lo[lv + 1])
                pd = wd_after[par[a:b]] if lv else np.zeros(b - a, bool)
                ld, wd = before_l["dirty"][a:b] != 0, before_w["dirty"][a:b] != 0
                wd_after[a:b] = pd | ld | wd
                reverse[a:b] = wd & ~ld & ~pd
            assert np.array_equal(sw, np.flatnonzero(wd_after)), tag
            assert np.array_equal(sl, np.flatnonzero(reverse | ((before_l["dirty"] != 0) & (mode == 1)))), f"{tag}: LocalTransform list"
            assert np.array_equal(rl2, rl[np.searchsorted(slots, sl)]), tag
            assert np.isin(np.flatnonzero(np.any(local["m"].view(np.uint32) != before_l["m"].view(np.uint32), axis=1)), sl).all(), tag
            mirror_w[slots] = rw
            mirror_l[sl] = rl2  # the split lists alone keep the host's columns equal to the oracle's
            mirror_w["dirty"] = 0  # PostTransformSystem on the host side of a row that was dirty is in the returned rows;
            mirror_l["dirty"] = 0  # rows not returned were clean already
            oracle_w, oracle_l = world.copy(), local.copy()
            with np.errstate(all="ignore"):
                assert_same_values(mirror_w["m"], oracle_w["m"], f"{tag}: world mirror")
                assert_same_values(mirror_l["m"], oracle_l["m"], f"{tag}: local mirror")
            assert not rw["dirty"].any() and not rl["dirty"].any()
            # and the device columns themselves agree
            assert_same_values(s.world_aos68()["m"], oracle_w["m"], f"{tag}: device world")

        step("frame 0 (everything dirty)", expect_changed=n)
        step("frame 1 (nothing dirty)", expect_changed=0)

        idx = np.unique(rng.integers(0, n, 400)).astype(np.uint32)
        for e_ in idx:
            local["m"][e_] = oracle.compose(rng.uniform(-5, 5, 3), rng.uniform(-1, 1, 4), rng.uniform(0.5, 2, 3))
        local["dirty"][idx] = 1
        s.upload_local_rows(idx, local[idx])
        if mode == 2:
            mirror_l[idx] = local[idx]  # the host's own edit: such a row does not come back in this mode
        step("frame 2 (sparse local edits)")

        idx = np.unique(rng.integers(int(lo[1]), n, 250)).astype(np.uint32)
        for e_ in idx:
            world["m"][e_] = oracle.compose(rng.uniform(-50, 50, 3), rng.uniform(-1, 1, 4), rng.uniform(0.5, 2, 3))
        world["dirty"][idx] = 1
        s.upload_world_rows(idx[::-1], world[idx[::-1]])  # any order
        step("frame 3 (sparse world edits)")
        step("frame 4 (nothing dirty)", expect_changed=0)

        leaf = np.array([n - 1], np.uint32)  # one leaf: exactly one row comes back
        local["m"][leaf] = oracle.compose([1, 1, 1], [0, 0, 0, 1], [1, 1, 1])
        local["dirty"][leaf] = 1
        s.upload_local_rows(leaf, local[leaf])
        if mode == 2:
            mirror_l[leaf] = local[leaf]
        step("frame 5 (one leaf)", expect_changed=1)

        with pytest.raises(_lib.YumeHipError) as e:
            s.upload_local_rows(np.array([n], np.uint32), local[:1])
        assert e.value.code == _lib.YH_ERR_BAD_ARG


def test_mat4_all_dirty_equals_trs_path(ctx, oracle):
    """The same scene through both device representations gives the same world matrices and draw lists."""
    from yume_amd import Scene, _lib

    n = 300_000
    lo, par, bounds, local, world = build(oracle, n, 8, seed=oracle.SEED)
    trs = oracle.synth_trs(n, int(lo[1]))
    planes = oracle.main_camera_planes().reshape(1, 24)
    mins = np.ascontiguousarray(bounds[:, 0:3].T)
    maxs = np.ascontiguousarray(bounds[:, 3:6].T)
    with Scene(ctx, n, n_levels=8, n_frusta=1, local_kind=_lib.YH_LOCAL_MAT4) as a, \
         Scene(ctx, n, n_levels=8, n_frusta=1) as b:
        for s in (a, b):
            s.upload_hierarchy(par, lo)
            s.upload_aabb(mins, maxs)
        a.upload_local_aos68(local)
        b.upload_trs(trs)
        a.frame(planes, flags=_lib.YH_FRAME_ALL_DIRTY)
        b.frame(planes)
        assert_same_values(a.world_soa(), b.world_soa(), "mat4 vs trs")
        assert np.array_equal(np.sort(a.draw_list(0)), np.sort(b.draw_list(0)))


def test_batched_decompose_equals_the_oracle_bit_for_bit(ctx, oracle):
    """Mat4.decompose over device columns; every operation is a correctly rounded IEEE op in the
    reference's order, so the planes must be bit-identical (including which rows are InvalidMatrix)."""
    from yume_amd import Scene, _lib

    n = 200_000
    lo, par, bounds, local, world = build(oracle, n, 6)
    root = oracle.identity68()
    oracle.frame_aos68(world, local, par, lo, root)
    rng = np.random.default_rng(9)
    # edge rows: singular, w = 0, all zero, mirrored, perspective partition, scaled w, non-finite
    world["m"][0, 4:7] = 0
    world["m"][1, 15] = 0
    world["m"][2, :] = 0
    world["m"][3, 0:3] *= -1
    world["m"][4, 3] = 0.5
    world["m"][5] *= 3.0
    world["m"][6, 12] = np.inf
    world["m"][7, 5] = np.nan
    idx = rng.integers(8, n, 500)
    world["m"][idx, 0:3] *= -1  # mirrored subtrees
    with Scene(ctx, n, n_levels=6, n_frusta=0, bounds_kind=0, local_kind=_lib.YH_LOCAL_MAT4) as s:
        s.upload_hierarchy(par, lo)
        s.upload_world_aos68(world)
        s.upload_local_aos68(local)
        with np.errstate(all="ignore"):
            for which, col in (("world", world), ("local", local)):
                got, valid = s.decompose(which)
                exp, exp_valid = oracle.decompose_batch(col)
                assert np.array_equal(valid, exp_valid), which
                same = (got.view(np.uint32) == exp.view(np.uint32)) | (np.isnan(got) & np.isnan(exp))
                assert same.all(), f"{which}: {np.count_nonzero(~same)} elements differ, first at {np.argwhere(~same)[0]}"
            assert not valid.all() or True
            part, pv = s.decompose("world", first=1000, count=777)
            assert np.array_equal(part.view(np.uint32), oracle.decompose_batch(world[1000:1777])[0].view(np.uint32))
    with Scene(ctx, 100, n_frusta=0, bounds_kind=0) as s:  # TRS scene: world only
        s.upload_flat()
        with pytest.raises(_lib.YumeHipError) as e:
            s.decompose("local")
        assert e.value.code == _lib.YH_ERR_STATE


@pytest.mark.parametrize("seed", range(6))
def test_chain_replay_equals_one_level_per_launch_on_tiny_mixed_levels(ctx, oracle, seed):
    """k_chain_mat4 replays its ancestors' rule in registers while other lanes OF THE SAME LAUNCH rewrite those
    ancestors' rows (wdirty 0 -> 1, world in the forward branch, lmat in the reverse branch).  The invariant beside the
    kernel says the replay never reads what a lane rewrites.  Forced shape: 8 levels of 2..60 entities (the whole tree
    is a few waves of one launch), every mix of local-dirty / world-dirty / clean rows and a clean or dirty root —
    the grouped plan must leave exactly what one launch per level (set_grouping(1): parents finished before children
    start) and the oracle leave, rows, flags and lists."""
    from yume_amd import Scene, _lib

    rng = np.random.default_rng(900 + seed)
    sizes = [int(rng.integers(2, 60)) for _ in range(8)]
    lo = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    n = int(lo[-1])
    par = oracle.synth_parents(lo, seed=seed)
    bounds = oracle.synth_aabb(n, seed=seed)
    planes = oracle.main_camera_planes().reshape(1, 24)
    local0 = oracle.compose_batch(oracle.synth_trs(n, int(lo[1]), seed=seed))
    world0 = np.zeros(n, oracle.T68)
    world0["m"][:, [0, 5, 10, 15]] = 1

    def script(s, local, world, root, plan_one):
        if plan_one:
            s.set_grouping(1)
        out = []
        r = np.random.default_rng(77 + seed)  # the same edits for both plans
        for frame in range(6):
            if frame == 0:
                local["dirty"] = 1
                world["dirty"] = 1
            else:
                ld = r.random(n) < 0.25
                wd = (r.random(n) < 0.25) & ~ld
                for e in np.flatnonzero(ld):
                    local["m"][e] = 0 if r.random() < 0.05 else oracle.compose(r.uniform(-5, 5, 3), r.uniform(-1, 1, 4), r.uniform(0.5, 2, 3))
                for e in np.flatnonzero(wd):
                    world["m"][e] = oracle.compose(r.uniform(-50, 50, 3), r.uniform(-1, 1, 4), r.uniform(0.5, 2, 3))
                local["dirty"][ld] = 1
                world["dirty"][wd] = 1
                if frame % 2 == 0:
                    root["m"][0] = oracle.compose(r.uniform(-3, 3, 3), r.uniform(-1, 1, 4), [1, 1, 1])
                    root["dirty"] = 1
                    s.set_root(root["m"][0])
            s.upload_local_aos68(local)
            s.upload_world_aos68(world)
            oracle.frame_aos68(world, local, par, lo, root)
            s.frame(planes, flags=0)
            out.append((s.world_aos68(), s.local_aos68(), s.draw_list(0).copy(), world.copy(), local.copy()))
        return out

    runs = []
    for plan_one in (False, True):
        with Scene(ctx, n, n_levels=8, n_frusta=1, local_kind=_lib.YH_LOCAL_MAT4) as s:
            s.upload_hierarchy(par, lo)
            s.upload_aabb(np.ascontiguousarray(bounds[:, 0:3].T), np.ascontiguousarray(bounds[:, 3:6].T))
            groups = s.groups()
            if not plan_one:
                assert len(groups) < 8, "the planner is expected to group these tiny levels into chain launches"
            runs.append(script(s, local0.copy(), world0.copy(), oracle.identity68(), plan_one))
    for frame, (a, b) in enumerate(zip(*runs)):
        with np.errstate(all="ignore"):
            assert_same_values(a[0]["m"], b[0]["m"], f"frame {frame}: world, chain vs one level per launch")
            assert_same_values(a[1]["m"], b[1]["m"], f"frame {frame}: local, chain vs one level per launch")
            assert_same_values(a[0]["m"], a[3]["m"], f"frame {frame}: world vs oracle")
            assert_same_values(a[1]["m"], a[4]["m"], f"frame {frame}: local vs oracle")
        assert np.array_equal(a[2], b[2]) and not a[0]["dirty"].any() and not a[1]["dirty"].any()


def test_spare_slots_dead_slots_and_a_moved_row(ctx, oracle):
    """A hierarchy that changes a little without a new yh_upload_hierarchy (yh_scene_set_alive_range,
    yh_upload_parent_range): every level has spare slots at its end, marked dead; a dead slot is never listed and never
    reported as changed whatever its rows hold; an entity that moves under another parent takes a spare slot of its
    level.  Lists, change lists and matrices against the oracle's compact scene, slot by slot."""
    from yume_amd import Scene, _lib

    n, levels, spare = 40_000, 6, 37
    lo, par, bounds, local, world = build(oracle, n, levels)
    planes = np.concatenate([oracle.main_camera_planes().reshape(1, 24), oracle.cascade_planes()[2:3]])
    root = oracle.identity68()
    # slot of every entity: level k's entities, then `spare` dead slots
    slot_lo = np.array([int(lo[k]) + k * spare for k in range(levels + 1)], np.uint32)
    level_of = np.searchsorted(lo, np.arange(n), side="right") - 1
    slot_of = (np.arange(n) + level_of * spare).astype(np.uint32)
    N = int(slot_lo[-1])
    spar = np.zeros(N, np.uint32)
    spar[slot_of] = np.where(par == 0xFFFFFFFF, 0xFFFFFFFF, slot_of[np.minimum(par, n - 1)])
    alive = np.zeros(N, bool)
    alive[slot_of] = True
    for k in range(levels):  # dead slots still need a parent inside the level before theirs
        dead = np.arange(int(lo[k + 1]) + k * spare, int(slot_lo[k + 1]))
        spar[dead] = 0xFFFFFFFF if k == 0 else slot_lo[k - 1]
    rng = np.random.default_rng(11)

    def padded(col):  # a compact column in slot order; dead rows hold junk, dirty flags included
        out = np.zeros(N, oracle.T68)
        out["m"] = rng.standard_normal((N, 16)).astype(np.float32)
        out["dirty"] = 1
        out[slot_of] = col
        return out

    with Scene(ctx, N, n_levels=levels, n_frusta=2, local_kind=_lib.YH_LOCAL_MAT4) as s:
        s.upload_hierarchy(spar, slot_lo)
        b = np.zeros((N, 6), np.float32)
        b[slot_of] = bounds
        b[~alive] = [-1e3, -1e3, -1e3, 1e3, 1e3, 1e3]  # huge boxes: visible if anything is
        s.upload_aabb(np.ascontiguousarray(b[:, 0:3].T), np.ascontiguousarray(b[:, 3:6].T))
        s.upload_local_aos68(padded(local))
        s.upload_world_aos68(padded(world))
        s.track_changes(True)
        for k in range(levels):
            s.set_alive_range(int(lo[k + 1]) + k * spare, spare, False)
        with pytest.raises(_lib.YumeHipError) as e:
            s.set_alive_range(N - 1, 2, False)
        assert e.value.code == _lib.YH_ERR_BAD_ARG

        def check(tag):
            oracle.frame_aos68(world, local, par, lo, root)
            s.frame(planes, flags=0)
            vis, bnd = oracle.cull(world, bounds, planes, n_threads=4)
            counts, bcounts = s.counts()
            for f in range(2):
                assert np.array_equal(s.draw_list(f), np.sort(slot_of[np.flatnonzero(vis[f])])), f"{tag}: frustum {f}"
                assert bcounts[f] == np.count_nonzero(bnd[f]), tag
            sw, rw = s.changed_rows(0)
            sl, rl = s.changed_rows(1)
            assert alive[sw].all() and alive[sl].all(), f"{tag}: a dead slot was reported as changed"
            dev = s.world_aos68()
            assert_same_values(dev["m"][slot_of], world["m"], f"{tag}: world")
            return sw

        assert check("frame 0").size == n  # everything dirty: every live row, no dead one
        assert check("frame 1").size == 0
        # move the last entity (a leaf of the last level) under another parent: a spare slot of its level, the old one dead
        e_, new_par = n - 1, int(lo[levels - 2]) + 5
        old_slot, new_slot = int(slot_of[e_]), int(lo[levels]) + (levels - 1) * spare + 3
        with pytest.raises(_lib.YumeHipError) as e:  # a parent from the wrong level
            s.upload_parent_range(np.array([slot_of[0]], np.uint32), new_slot)
        assert e.value.code == _lib.YH_ERR_BAD_ARG
        with pytest.raises(_lib.YumeHipError) as e:  # a range across two levels
            s.upload_parent_range(np.array([slot_of[new_par]] * 2, np.uint32), int(slot_lo[levels - 1]) - 1)
        assert e.value.code == _lib.YH_ERR_BAD_ARG
        s.set_alive_range(old_slot, 1, False)
        s.set_alive_range(new_slot, 1, True)
        s.upload_parent_range(np.array([slot_of[new_par]], np.uint32), new_slot)
        par[e_] = new_par
        slot_of[e_] = new_slot
        alive[old_slot], alive[new_slot] = False, True
        local["dirty"][e_] = 1  # (the reference's re-parenting leaves flags alone; here the move is made visible)
        s.upload_local_aos68(local[e_:e_ + 1], first=new_slot)
        s.upload_world_aos68(world[e_:e_ + 1], first=new_slot)
        mn, mx = bounds[e_:e_ + 1, 0:3].T, bounds[e_:e_ + 1, 3:6].T
        s.upload_aabb(np.ascontiguousarray(mn), np.ascontiguousarray(mx), first=new_slot)
        moved = check("frame 2 (one entity moved)")
        assert np.array_equal(moved, [new_slot])
        assert check("frame 3").size == 0


def test_many_runs_and_single_slots_in_one_call_equal_the_range_calls(ctx, oracle):
    """yh_scene_set_alive_runs / yh_upload_parent_rows / yh_upload_aabb_rows / yh_upload_mesh_id_rows (what a structural
    patch of the drop-in uses when it touches thousands of tables) against the one-range-per-call entry points: two
    scenes with spare slots get the same 700 moves - a leaf takes a spare slot of its level under another parent, its
    old slot dies - one through the batched calls, one run by run; frames, draw lists, change lists, boxes and matrices
    must be identical, and both equal the oracle.  Bad arguments are refused before anything is written."""
    from yume_amd import Scene, _lib

    n, levels, spare = 60_000, 6, 1000
    lo, par, bounds, local, world = build(oracle, n, levels)
    planes = oracle.main_camera_planes().reshape(1, 24)
    root = oracle.identity68()
    slot_lo = np.array([int(lo[k]) + k * spare for k in range(levels + 1)], np.uint32)
    level_of = np.searchsorted(lo, np.arange(n), side="right") - 1
    slot_of = (np.arange(n) + level_of * spare).astype(np.uint32)
    N = int(slot_lo[-1])
    spar = np.zeros(N, np.uint32)
    spar[slot_of] = np.where(par == 0xFFFFFFFF, 0xFFFFFFFF, slot_of[np.minimum(par, n - 1)])
    for k in range(levels):
        dead = np.arange(int(lo[k + 1]) + k * spare, int(slot_lo[k + 1]))
        spar[dead] = 0xFFFFFFFF if k == 0 else slot_lo[k - 1]
    rng = np.random.default_rng(23)

    def padded(col):
        out = np.zeros(N, oracle.T68)
        out[slot_of] = col
        return out

    # the moves: 700 leaves of the last two levels, each to a spare slot of its own level under a random parent
    movers = np.concatenate([rng.choice(np.arange(int(lo[levels - 1]), n), 400, replace=False),
                             rng.choice(np.arange(int(lo[levels - 2]), int(lo[levels - 1])), 300, replace=False)])
    # (level levels-2 entities that move must have no children here: pick ones nobody names as parent)
    movers = np.array([e for e in movers if not np.any(par == e)], np.int64)
    new_par = np.array([rng.integers(int(lo[level_of[e] - 1]), int(lo[level_of[e]])) for e in movers])
    next_spare = {k: int(lo[k + 1]) + k * spare for k in range(levels)}
    new_slot = np.zeros(movers.size, np.uint32)
    for i, e in enumerate(movers):
        new_slot[i] = next_spare[level_of[e]]
        next_spare[level_of[e]] += 1
    old_slot = slot_of[movers].copy()

    scenes = [Scene(ctx, N, n_levels=levels, n_frusta=1, local_kind=_lib.YH_LOCAL_MAT4) for _ in range(2)]
    try:
        b = np.zeros((N, 6), np.float32)
        b[slot_of] = bounds
        for s in scenes:
            s.upload_hierarchy(spar, slot_lo)
            s.upload_aabb(np.ascontiguousarray(b[:, 0:3].T), np.ascontiguousarray(b[:, 3:6].T))
            s.upload_local_aos68(padded(local))
            s.upload_world_aos68(padded(world))
            s.track_changes(True)
        batched, ranged = scenes
        # dead spare slots: one call against one per level
        firsts = np.array([int(lo[k + 1]) + k * spare for k in range(levels)], np.uint32)
        batched.set_alive_runs(firsts, np.full(levels, spare, np.uint32), np.zeros(levels, np.uint8))
        for f in firsts:
            ranged.set_alive_range(int(f), spare, False)
        with pytest.raises(_lib.YumeHipError) as e:
            batched.set_alive_runs([N - 1], [2], [0])
        assert e.value.code == _lib.YH_ERR_BAD_ARG
        with pytest.raises(_lib.YumeHipError) as e:  # a parent from the wrong level: nothing is written
            batched.upload_parent_rows([int(new_slot[0]), int(new_slot[1])], [int(slot_of[new_par[0]]), int(slot_of[0])])
        assert e.value.code == _lib.YH_ERR_BAD_ARG
        with pytest.raises(_lib.YumeHipError) as e:
            batched.upload_aabb_rows([N], np.zeros((1, 6), np.float32))
        assert e.value.code == _lib.YH_ERR_BAD_ARG
        for s in scenes:
            s.frame(planes, flags=0)
        oracle.frame_aos68(world, local, par, lo, root)  # frame 0: everything dirty
        # the moves
        par[movers] = new_par
        slot_of[movers] = new_slot
        local["dirty"][movers] = 1
        rows_l, rows_w = local[movers], world[movers]
        batched.set_alive_runs(np.concatenate([old_slot, new_slot]), np.ones(2 * movers.size, np.uint32),
                               np.concatenate([np.zeros(movers.size, np.uint8), np.ones(movers.size, np.uint8)]))
        batched.upload_parent_rows(new_slot, slot_of[new_par])
        batched.upload_aabb_rows(new_slot, bounds[movers])
        batched.upload_local_rows(new_slot, rows_l)
        batched.upload_world_rows(new_slot, rows_w)
        for i, e_ in enumerate(movers):
            ranged.set_alive_range(int(old_slot[i]), 1, False)
            ranged.set_alive_range(int(new_slot[i]), 1, True)
            ranged.upload_parent_range(np.array([slot_of[new_par[i]]], np.uint32), int(new_slot[i]))
            ranged.upload_aabb(np.ascontiguousarray(bounds[e_:e_ + 1, 0:3].T), np.ascontiguousarray(bounds[e_:e_ + 1, 3:6].T), first=int(new_slot[i]))
            ranged.upload_local_aos68(local[e_:e_ + 1], first=int(new_slot[i]))
            ranged.upload_world_aos68(world[e_:e_ + 1], first=int(new_slot[i]))
        for fi in range(2):
            oracle.frame_aos68(world, local, par, lo, root)
            for s in scenes:
                s.frame(planes, flags=0)
            vis, _ = oracle.cull(world, bounds, planes, n_threads=4)
            lists = [s.draw_list(0) for s in scenes]
            assert np.array_equal(lists[0], lists[1]), f"frame {fi + 1}: draw lists of the two scenes"
            assert np.array_equal(lists[0], np.sort(slot_of[np.flatnonzero(vis[0])])), f"frame {fi + 1}: against the oracle"
            cw = [s.changed_rows(0)[0] for s in scenes]
            assert np.array_equal(cw[0], cw[1]), f"frame {fi + 1}: change lists"
            if fi == 0:
                assert np.array_equal(np.sort(cw[0]), np.sort(new_slot)), "exactly the moved rows were recomputed"
            dev = [s.world_aos68() for s in scenes]
            assert_same_values(dev[0]["m"][slot_of], world["m"], f"frame {fi + 1}: world, batched")
            assert_same_values(dev[1]["m"][slot_of], world["m"], f"frame {fi + 1}: world, ranged")
        mn0, mx0 = batched.download_aabb()
        mn1, mx1 = ranged.download_aabb()
        assert np.array_equal(mn0[:, slot_of], mn1[:, slot_of]) and np.array_equal(mx0[:, slot_of], mx1[:, slot_of])
    finally:
        for s in scenes:
            s.close()

    # mesh-table scenes: ids of single slots
    nm = 16  # (fill_synthetic hands out ids 0 .. 15)
    with Scene(ctx, 5000, n_frusta=1, bounds_kind=2, max_meshes=16) as a, Scene(ctx, 5000, n_frusta=1, bounds_kind=2, max_meshes=16) as c:
        table = rng.uniform(-2, 2, (nm, 6)).astype(np.float32)
        table[:, 3:] = np.abs(table[:, 3:]) + 0.1
        table[:, :3] = -np.abs(table[:, :3]) - 0.1
        ids = rng.integers(0, nm, 5000).astype(np.uint32)
        sl = rng.choice(5000, 1200, replace=False).astype(np.uint32)
        for s in (a, c):
            s.upload_flat()
            s.fill_synthetic()
            s.upload_mesh_aabbs(table)
        a.upload_mesh_id_rows(sl, ids[sl])
        for e_ in sl:
            c.upload_mesh_ids(ids[e_:e_ + 1], first=int(e_))
        with pytest.raises(_lib.YumeHipError) as e:
            a.upload_mesh_id_rows([3], [nm])
        assert e.value.code == _lib.YH_ERR_BAD_ARG
        for s in (a, c):
            s.frame(planes)
        assert np.array_equal(a.draw_list(0), c.draw_list(0))


def test_rows_change_slots_on_the_device(ctx, oracle):
    """yh_move_rows: both rows of src[i], flags included, become the rows of dst[i]; sources are read before any
    destination is written (a rotation: every destination is another pair's source), rows outside dst keep theirs;
    bad slots and TRS scenes are refused."""
    from yume_amd import Scene, _lib

    n = 70_001
    rng = np.random.default_rng(21)
    world, local = np.zeros(n, oracle.T68), np.zeros(n, oracle.T68)
    for col in (world, local):
        col["m"] = rng.standard_normal((n, 16)).astype(np.float32)
        col["dirty"] = rng.integers(0, 2, n)
    with Scene(ctx, n, n_levels=1, n_frusta=1, local_kind=_lib.YH_LOCAL_MAT4) as s:
        s.upload_flat()
        s.upload_world_aos68(world)
        s.upload_local_aos68(local)
        ring = rng.permutation(n)[:20_000].astype(np.uint32)          # src[i] -> src[i + 1]: all of them overlap
        rest = np.setdiff1d(np.arange(n, dtype=np.uint32), ring)
        src = np.concatenate([ring, rest[:300]])                      # plus plain moves onto slots nobody reads
        dst = np.concatenate([np.roll(ring, -1), rest[300:600]])
        s.move_rows(src, dst)
        exp_w, exp_l = world.copy(), local.copy()
        exp_w[dst], exp_l[dst] = world[src], local[src]
        gw, gl = s.world_aos68(), s.local_aos68()
        for got, exp, what in ((gw, exp_w, "world"), (gl, exp_l, "local")):
            assert np.array_equal(got["m"].view(np.uint32), exp["m"].view(np.uint32)), what
            assert np.array_equal(got["dirty"], exp["dirty"]), what + " flags"
        s.move_rows(np.zeros(0, np.uint32), np.zeros(0, np.uint32))
        for bad_src, bad_dst in (([n], [0]), ([0], [n])):
            with pytest.raises(_lib.YumeHipError) as e:
                s.move_rows(np.array(bad_src, np.uint32), np.array(bad_dst, np.uint32))
            assert e.value.code == _lib.YH_ERR_BAD_ARG
    with Scene(ctx, 100, n_levels=1, n_frusta=1) as t:
        with pytest.raises(_lib.YumeHipError) as e:
            t.move_rows(np.array([1], np.uint32), np.array([2], np.uint32))
        assert e.value.code == _lib.YH_ERR_STATE


@pytest.mark.parametrize("kind", ["mat4", "mat4-mesh-table", "trs"])
def test_cull_only_frames_equal_a_full_frame_with_the_same_frusta(ctx, oracle, kind):
    """YH_FRAME_CULL_ONLY (only a camera or the bounds changed): one launch culls the world matrices the device holds.
    Its lists, counts and boundary counts equal the oracle's cull of the same rows and a full frame's with the same
    frusta; the world rows and flags stay as they were; the changed lists are empty; dead slots stay out of the lists;
    and it is refused before the first frame, with a dirty root's row and together with YH_FRAME_ALL_DIRTY."""
    from yume_amd import Scene, _lib

    n = 300_001  # a last chunk of one entity
    lo, par, bounds, local, world = build(oracle, n, 7, seed=oracle.SEED + 3)
    root = oracle.identity68()
    cams = np.concatenate([oracle.main_camera_planes().reshape(1, 24), oracle.cascade_planes()[:3]])
    mesh = kind == "mat4-mesh-table"
    if mesh:
        rng = np.random.default_rng(5)
        table = np.concatenate([rng.uniform(-3, 0, (32, 3)), rng.uniform(0, 3, (32, 3))], axis=1).astype(np.float32)
        ids = rng.integers(0, 32, n).astype(np.uint32)
        bounds = table[ids]
    kw = dict(n_levels=7, n_frusta=4)
    if kind != "trs":
        kw["local_kind"] = _lib.YH_LOCAL_MAT4
    if mesh:
        kw.update(bounds_kind=_lib.YH_BOUNDS_MESH_TABLE, max_meshes=32)
    with Scene(ctx, n, **kw) as s:
        s.upload_hierarchy(par, lo)
        if mesh:
            s.upload_mesh_aabbs(table)
            s.upload_mesh_ids(ids)
        else:
            s.upload_aabb(np.ascontiguousarray(bounds[:, 0:3].T), np.ascontiguousarray(bounds[:, 3:6].T))
        if kind == "trs":
            trs = oracle.synth_trs(n, int(lo[1]))
            s.upload_trs(trs)
            exp_w = oracle.frame_trs(trs, par, lo, n_threads=8)
            full = _lib.YH_FRAME_ALL_DIRTY
        else:
            s.upload_local_aos68(local)
            s.upload_world_aos68(world)
            s.upload_root_aos68(root)
            s.track_changes(True)
            oracle.frame_aos68(world, local, par, lo, root)
            exp_w = world
            full = 0
        with pytest.raises(_lib.YumeHipError) as e:  # no world matrices yet
            s.frame(cams[:1], _lib.YH_FRAME_CULL_ONLY)
        assert e.value.code == _lib.YH_ERR_STATE
        s.frame(cams[:1], full)
        before = s.world_soa().copy()

        def check(nf, planes, tag, dead=None):
            vis, bnd = oracle.cull(exp_w, bounds, planes, n_threads=8)
            if dead is not None:
                vis[:, dead] = 0
                bnd[:, dead] = 0
            counts, bcounts = s.counts()
            for f in range(nf):
                assert counts[f] == np.count_nonzero(vis[f]) and bcounts[f] == np.count_nonzero(bnd[f]), (tag, f)
                assert np.array_equal(s.draw_list(f), np.flatnonzero(vis[f]).astype(np.uint32)), (tag, f)

        for rnd, sel in enumerate([[1, 2, 3], [0], [3, 1], [0, 1, 2, 3], [2]]):
            planes = cams[sel]
            fl = _lib.YH_FRAME_CULL_ONLY | (_lib.YH_FRAME_ASYNC if rnd % 2 else 0)
            s.frame(planes, fl)
            assert "k_cull_only" in s.group_kernel_names()[0]
            check(len(sel), planes, f"cull-only {rnd}")
            if kind != "trs":
                assert s.changed_counts() == (0, 0)
        assert np.array_equal(s.world_soa().view(np.uint32), before.view(np.uint32))
        # a full frame with the same frusta gives the same lists (nothing is dirty: the Mat4 rule leaves every row alone)
        planes = cams[[2, 0]]
        s.frame(planes, _lib.YH_FRAME_CULL_ONLY)
        a = [s.draw_list(f).copy() for f in range(2)]
        s.frame(planes, full)
        assert all(np.array_equal(a[f], s.draw_list(f)) for f in range(2))
        if kind != "trs":
            # bounds edited (MeshComponent set / asset loaded), nothing moved: cull-only sees the new boxes
            sl = np.arange(10, n, 977, dtype=np.uint32)
            if mesh:
                ids[sl] = (ids[sl] + 7) % 32
                s.upload_mesh_id_rows(sl, ids[sl])
                bounds = table[ids]
            else:
                bounds = bounds.copy()
                bounds[sl, 3:6] += 40.0
                s.upload_aabb_rows(sl, bounds[sl])
            s.frame(cams[:2], _lib.YH_FRAME_CULL_ONLY)
            check(2, cams[:2], "after a bounds edit")
            # dead slots (entities deleted since) are in no list
            s.set_alive_range(1000, 5000, False)
            s.frame(cams[:2], _lib.YH_FRAME_CULL_ONLY)
            check(2, cams[:2], "dead slots", dead=slice(1000, 6000))
            # what is pending stays pending: a moved root and an edited row wait for the next full frame, which then
            # gives what the oracle gives
            s.set_alive_range(1000, 5000, True)
            root["m"][0] = oracle.compose([3, -2, 5], [0, 0.38268343, 0, 0.92387953], [1, 1, 1])
            root["dirty"] = 1
            s.upload_root_aos68(root)
            edit = np.array([int(lo[2]) + 11], np.uint32)
            local["m"][edit] = oracle.compose([9, 9, 9], [0, 0, 0, 1], [2, 2, 2])
            local["dirty"][edit] = 1
            s.upload_local_rows(edit, local[edit])
            s.frame(cams[:2], _lib.YH_FRAME_CULL_ONLY)
            check(2, cams[:2], "pending edits do not show yet")
            assert s.changed_counts() == (0, 0)
            assert np.array_equal(s.world_soa().view(np.uint32), before.view(np.uint32))
            oracle.frame_aos68(world, local, par, lo, root)
            s.frame(cams[:2], 0)
            assert_same_values(s.world_aos68()["m"], world["m"], "the full frame after")
            check(2, cams[:2], "the full frame after")
            assert s.changed_counts()[0] == n
        with pytest.raises(_lib.YumeHipError) as e:
            s.frame(cams[:2], _lib.YH_FRAME_CULL_ONLY | _lib.YH_FRAME_ALL_DIRTY)
        assert e.value.code == _lib.YH_ERR_BAD_ARG


def test_sparse_rows_through_page_locked_buffers_equal_the_staged_path(ctx, oracle):
    """A few thousand sparse rows between page-locked host memory (yh_host_alloc) and the scene go through the import /
    export kernels alone, no copy engine, no device staging: what lands on the device and what comes back must be what
    the staged path (ordinary numpy memory) moves - rows, slots, dirty bytes - and the frame after equals the oracle's."""
    from yume_amd import Scene, _lib

    n = 120_000
    lo, par, bounds, local, world = build(oracle, n, 6, seed=oracle.SEED + 9)
    root = oracle.identity68()
    planes = oracle.main_camera_planes().reshape(1, 24)
    rng = np.random.default_rng(77)
    with Scene(ctx, n, n_levels=6, n_frusta=1, local_kind=_lib.YH_LOCAL_MAT4) as a, \
         Scene(ctx, n, n_levels=6, n_frusta=1, local_kind=_lib.YH_LOCAL_MAT4) as b:
        for s in (a, b):
            s.upload_hierarchy(par, lo)
            s.upload_aabb(np.ascontiguousarray(bounds[:, 0:3].T), np.ascontiguousarray(bounds[:, 3:6].T))
            s.upload_local_aos68(local)
            s.upload_world_aos68(world)
            s.track_changes(_lib.YH_TRACK_REWRITTEN_LOCALS)
            s.frame(planes, 0)
        oracle.frame_aos68(world, local, par, lo, root)
        for k in (1, 300, 5000, 32768, 40000):  # the last one is over the limit: staged in both scenes
            idx = np.unique(rng.integers(0, n, k)).astype(np.uint32)
            for e_ in idx[:200]:
                local["m"][e_] = oracle.compose(rng.uniform(-5, 5, 3), rng.uniform(-1, 1, 4), rng.uniform(0.5, 2, 3))
            local["dirty"][idx] = 1
            widx = np.setdiff1d(np.unique(rng.integers(0, n, max(1, k // 4))).astype(np.uint32), idx)
            for e_ in widx[:100]:
                world["m"][e_] = oracle.compose(rng.uniform(-50, 50, 3), rng.uniform(-1, 1, 4), rng.uniform(0.5, 2, 3))
            world["dirty"][widx] = 1
            p_slots, p_rows = ctx.host_array(idx.size, np.uint32), ctx.host_array(idx.size, oracle.T68)
            p_slots[:], p_rows[:] = idx, local[idx]
            a.upload_local_rows(p_slots, p_rows)
            b.upload_local_rows(idx, local[idx])
            if widx.size:
                w_slots, w_rows = ctx.host_array(widx.size, np.uint32), ctx.host_array(widx.size, oracle.T68)
                w_slots[:], w_rows[:] = widx, world[widx]
                a.upload_world_rows(w_slots, w_rows)
                b.upload_world_rows(widx, world[widx])
                ctx.host_free(w_slots)
                ctx.host_free(w_rows)
            ctx.host_free(p_slots)
            ctx.host_free(p_rows)
            oracle.frame_aos68(world, local, par, lo, root)
            a.frame(planes, 0)
            b.frame(planes, 0)
            assert_same_values(a.world_aos68()["m"], world["m"], f"{k} rows: world through page-locked buffers")
            assert_same_values(a.local_aos68()["m"], local["m"], f"{k} rows: local through page-locked buffers")
            for which in (0, 1):  # the changed rows down again: page-locked destination against ordinary memory
                nw = a.changed_counts()[which]
                assert nw == b.changed_counts()[which]
                d_slots, d_rows = ctx.host_array(max(1, nw), np.uint32), ctx.host_array(max(1, nw), oracle.T68)
                sa, ra = a.changed_rows(which, out=(d_slots, d_rows))
                sb, rb = b.changed_rows(which)
                assert sa.size == nw == sb.size
                assert np.array_equal(sa, sb) and np.array_equal(ra.view(np.uint8), rb.view(np.uint8)), (k, which)
                ctx.host_free(d_slots)
                ctx.host_free(d_rows)

