"""GPU parity tests: the HIP path, called through the C ABI (ctypes -> libyume_hip.so), against the CPU
oracle on the same seeded inputs.

Bar: world matrices equal as IEEE values element for element (zero tolerance; the only freedom is the sign
of a zero, see yh_math.hpp m4_compose) — far inside north_star's 1e-5 relative; visible-index SETS exact;
boundary counts exact.  NaN positions must coincide.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL_TOL = 0.0  # north_star allows 1e-5 relative per element; we require equality


def assert_same_values(got, exp, what=""):
    got = np.asarray(got)
    exp = np.asarray(exp)
    assert got.shape == exp.shape, what
    nan_g, nan_e = np.isnan(got), np.isnan(exp)
    assert np.array_equal(nan_g, nan_e), f"{what}: NaN pattern differs"
    ok = (got == exp) | nan_g
    if not ok.all():
        bad = np.argwhere(~ok)[0]
        raise AssertionError(f"{what}: {np.count_nonzero(~ok)} elements differ, first at {tuple(bad)}: "
                             f"{got[tuple(bad)]!r} vs {exp[tuple(bad)]!r}")


def make_scene(oracle, n, n_levels, seed=None, index_base=0):
    seed = oracle.SEED if seed is None else seed
    lo = oracle.synth_level_offsets(n, n_levels)
    par = oracle.synth_parents(lo, seed=seed, index_base=index_base)
    trs = oracle.synth_trs(n, int(lo[1]), seed=seed, index_base=index_base)
    bounds = oracle.synth_aabb(n, seed=seed, index_base=index_base)
    return lo, par, trs, bounds


def gpu_scene(ctx, lo, par, trs, bounds, n_frusta, **kw):
    from yume_amd import Scene

    n = trs.shape[1]
    s = Scene(ctx, n, n_levels=len(lo) - 1, n_frusta=n_frusta, **kw)
    s.upload_hierarchy(par, lo)
    s.upload_trs(trs)
    s.upload_aabb(np.ascontiguousarray(bounds[:, 0:3].T), np.ascontiguousarray(bounds[:, 3:6].T))
    return s


def check_frame(oracle, s, lo, par, trs, bounds, planes, threads=4, id_base=0):
    planes = np.asarray(planes, np.float32).reshape(-1, 24)
    exp_world = oracle.frame_trs(trs, par, lo, n_threads=threads)
    vis, bnd = oracle.cull(exp_world, bounds, planes, n_threads=threads)
    s.frame(planes)
    got = s.world_soa()
    assert_same_values(got, np.ascontiguousarray(exp_world["m"].T), "world")
    counts, bcounts = s.counts()
    n_bnd_total = 0
    for f in range(planes.shape[0]):
        exp_set = np.flatnonzero(vis[f]).astype(np.uint32) + np.uint32(id_base)
        lst = s.draw_list(f)
        assert counts[f] == exp_set.size == lst.size, f"frustum {f}: count {counts[f]} vs {exp_set.size}"
        assert np.array_equal(lst, exp_set), f"frustum {f}: draw list differs (must be the visible set in ascending order)"
        assert bcounts[f] == np.count_nonzero(bnd[f]), f"frustum {f}: boundary count"
        n_bnd_total += int(bcounts[f])
    return n_bnd_total


# ---------------------------------------------------------------------------------------------- generator
def test_device_generator_matches_oracle(ctx, oracle):
    from yume_amd import Scene

    n = 200_000
    lo = oracle.synth_level_offsets(n, 8)
    par = oracle.synth_parents(lo)
    for base in (0, 5_000_000_000):
        with Scene(ctx, n, n_levels=8, n_frusta=1) as s:
            s.upload_hierarchy(par, lo)
            s.fill_synthetic(index_base=base)
            trs = s.download_trs()
            mins, maxs = s.download_aabb()
        exp = oracle.synth_trs(n, int(lo[1]), index_base=base)
        diff = (trs.view(np.uint32) != exp.view(np.uint32)).sum(axis=1)
        assert not diff.any(), f"planes differing (count per plane): {diff.tolist()}"
        b = oracle.synth_aabb(n, index_base=base)
        assert np.array_equal(mins.T, b[:, 0:3]) and np.array_equal(maxs.T, b[:, 3:6])


# ------------------------------------------------------------------------------------------------ compose
def test_compose_random(ctx, oracle):
    from yume_amd import Scene

    n = 100_000
    trs = oracle.synth_trs(n, n // 2)
    with Scene(ctx, n, n_frusta=0, bounds_kind=0) as s:
        s.upload_flat()
        s.upload_trs(trs)
        got = s.compose()
    exp = oracle.compose_batch(trs)["m"].T
    assert_same_values(got, exp, "compose")


def test_compose_edge_cases(ctx, oracle):
    """Identity, zero scale, unnormalised and zero quaternions, signed zeros, overflow and non-finite inputs:
    the closed form must fall back to the literal two-product path wherever the reference yields NaN."""
    from yume_amd import Scene

    inf, nan, big = np.inf, np.nan, 3e38
    rows = [
        [0, 0, 0, 0, 0, 0, 1, 1, 1, 1],
        [1, 2, 3, 0, 0, 0, 0, 1, 1, 1],
        [1, 2, 3, 0.5, 0.5, 0.5, 0.5, 0, 0, 0],
        [-0.0, 0.0, -0.0, -0.0, 0.0, -0.0, 1, -1, 2, -3],
        [1, 2, 3, 2, -3, 4, 5, 1, 1, 1],
        [inf, 0, 0, 0, 0, 0, 1, 1, 1, 1],
        [0, -inf, 0, 0, 0, 0, 1, 1, 1, 1],
        [nan, 0, 0, 0, 0, 0, 1, 1, 1, 1],
        [0, 0, 0, inf, 0, 0, 1, 1, 1, 1],
        [0, 0, 0, 0, 0, 0, inf, 1, 1, 1],
        [0, 0, 0, 0, nan, 0, 1, 1, 1, 1],
        [0, 0, 0, 0, 0, 0, 1, inf, 1, 1],
        [0, 0, 0, 0, 0.5, 0.5, 0, inf, 1, 1],
        [0, 0, 0, 0, 0, 0, 1, 1, nan, 1],
        [0, 0, 0, 1e20, 1e20, 0, 1, 1, 1, 1],
        [0, 0, 0, 1e19, 1, 1, 1e19, 1, 1, 1],
        [big, big, big, 0.1, 0.2, 0.3, 0.9, big, big, big],
        [1, 1, 1, 0.1, 0.2, 0.3, 0.9, 1e-40, 1e-45, 1],
        [1e-45, -1e-45, 0, 1e-30, 1e-30, 1e-30, 1e-30, 1, 1, 1],
    ]
    trs = np.ascontiguousarray(np.array(rows, np.float32).T)
    n = trs.shape[1]
    with Scene(ctx, n, n_frusta=0, bounds_kind=0) as s:
        s.upload_flat()
        s.upload_trs(trs)
        got = s.compose()
        s.frame(None)
        world = s.world_soa()
    exp = oracle.compose_batch(trs)
    with np.errstate(all="ignore"):
        assert_same_values(got, exp["m"].T, "compose edge")
        expw = oracle.frame_trs(trs, np.full(n, 0xFFFFFFFF, np.uint32), np.array([0, n], np.uint32))
        assert_same_values(world, expw["m"].T, "world edge")


# ------------------------------------------------------------------------------------------ whole frames
def test_cfg1_flat_10k(ctx, oracle):
    """BASELINE configs[0]: 10k flat entities, TRS->world + frustum cull."""
    lo, par, trs, bounds = make_scene(oracle, 10_000, 1)
    with gpu_scene(ctx, lo, par, trs, bounds, 1) as s:
        check_frame(oracle, s, lo, par, trs, bounds, oracle.main_camera_planes())


@pytest.mark.parametrize("n", [1, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 5000])
def test_ragged_sizes_flat(ctx, oracle, n):
    lo, par, trs, bounds = make_scene(oracle, n, 1)
    with gpu_scene(ctx, lo, par, trs, bounds, 1) as s:
        check_frame(oracle, s, lo, par, trs, bounds, oracle.main_camera_planes())


@pytest.mark.parametrize("group", [0, 1, 2, 3, 4, 8])
def test_tree_4096_all_groupings(ctx, oracle, group):
    lo, par, trs, bounds = make_scene(oracle, 4096, 8)
    with gpu_scene(ctx, lo, par, trs, bounds, 1) as s:
        s.set_grouping(group)
        check_frame(oracle, s, lo, par, trs, bounds, oracle.main_camera_planes())


@pytest.mark.parametrize("group", [0, 1, 2, 4, 8])
def test_cfg2_tree_1m(ctx, oracle, group):
    """BASELINE configs[1]: 1M entities, random 8-level hierarchy, single camera frustum."""
    lo, par, trs, bounds = make_scene(oracle, 1_000_000, 8)
    with gpu_scene(ctx, lo, par, trs, bounds, 1) as s:
        s.set_grouping(group)
        check_frame(oracle, s, lo, par, trs, bounds, oracle.main_camera_planes())
        # a second frame must give the same answer (counters reset, lists rewritten)
        check_frame(oracle, s, lo, par, trs, bounds, oracle.main_camera_planes())


@pytest.mark.parametrize("plan", [[7], [3, 7], [5, 7], [0, 1, 2, 3, 4, 5, 6, 7], [1, 4, 7], [0, 6, 7]])
def test_explicit_launch_plans_give_identical_results(ctx, oracle, plan):
    from yume_amd import _lib

    lo, par, trs, bounds = make_scene(oracle, 50_000, 8)
    with gpu_scene(ctx, lo, par, trs, bounds, 1) as s:
        s.set_plan(plan)
        g = s.groups()
        assert [int(q[1]) for q in g] == plan and int(g[0][0]) == 0
        check_frame(oracle, s, lo, par, trs, bounds, oracle.main_camera_planes())
        s.set_plan()  # back to the planner
        check_frame(oracle, s, lo, par, trs, bounds, oracle.main_camera_planes())
        for bad in ([3, 6], [3, 3, 7], [8], [7, 7]):  # not covering, not increasing, out of range
            with pytest.raises(_lib.YumeHipError) as e:
                s.set_plan(bad)
            assert e.value.code == _lib.YH_ERR_BAD_ARG


def test_planner_groups_small_levels_and_triples_large_ones(ctx, oracle):
    from yume_amd import Scene

    # (large trees are cut from the last level backwards: a level of 5M entities alone, launches of at most ~6.1M)
    for n, expect in ((1_000_000, [5, 7]), (12_500_000, [3, 6, 7]), (4096, [7]), (4_000_000, [4, 7]), (8_000_000, [2, 5, 7]),
                      (25_000_000, [2, 5, 6, 7])):
        lo = oracle.synth_level_offsets(n, 8)
        with Scene(ctx, n, n_levels=8, n_frusta=0, bounds_kind=0) as s:
            s.upload_hierarchy(oracle.synth_parents(lo), lo)
            assert [int(q[1]) for q in s.groups()] == expect
    # culling scenes small enough for the expansion to ride on the next frame's first launch: a lighter first launch
    for n, expect in ((1_000_000, [4, 7]), (500_000, [5, 7]), (250_000, [7]), (12_500_000, [3, 6, 7])):
        lo = oracle.synth_level_offsets(n, 8)
        with Scene(ctx, n, n_levels=8, n_frusta=1) as s:
            s.upload_hierarchy(oracle.synth_parents(lo), lo)
            assert [int(q[1]) for q in s.groups()] == expect


def test_deep_and_uneven_hierarchy(ctx, oracle):
    """12 levels with hand-made uneven level sizes (some levels smaller than their parent level), root != identity."""
    sizes = [3, 50, 7, 900, 901, 20, 4000, 1, 300, 2000, 5, 1024]
    lo = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    n = int(lo[-1])
    par = oracle.synth_parents(lo)
    trs = oracle.synth_trs(n, int(lo[1]))
    bounds = oracle.synth_aabb(n)
    root = oracle.identity68()
    root["m"][0] = oracle.compose([3, -2, 1], [0.1, 0.2, 0.3, 0.9], [1.5, 0.5, 2])
    planes = oracle.main_camera_planes().reshape(1, 24)
    for group in (0, 1, 3, 8):
        with gpu_scene(ctx, lo, par, trs, bounds, 1) as s:
            s.set_grouping(group)
            s.set_root(root["m"][0])
            exp = oracle.frame_trs(trs, par, lo, root=root.copy())
            vis, _ = oracle.cull(exp, bounds, planes)
            s.frame(planes)
            assert_same_values(s.world_soa(), exp["m"].T, f"world g={group}")
            assert np.array_equal(s.draw_list(0), np.flatnonzero(vis[0]).astype(np.uint32))


@pytest.mark.parametrize("seed", range(int(os.environ.get("YH_FUZZ_SEEDS", "12"))))  # more seeds: YH_FUZZ_SEEDS=300
def test_random_scene_shapes_plans_and_frame_sequences(ctx, oracle, seed):
    """Seeded random differential test: level counts 1..12, uneven level sizes from a few entities to ~200k (so that
    launches below and above the XCD-ordering threshold, ragged tiles and chunk-straddling level boundaries all occur),
    random launch plans, id offsets, 1..4 frusta, and a random mix of synchronous and YH_FRAME_ASYNC frames (deferred
    expansion riding on the next frame; lists on demand).  World matrices, lists and boundary counts must equal the oracle's."""
    from yume_amd import _lib

    rng = np.random.default_rng(1000 + seed)
    rng_flag = np.random.default_rng(9000 + seed)  # (its own stream: the scenes of the seeds stay what they were)

    def on_demand():  # a third of the asynchronous frames leave their lists unexpanded unless read
        return _lib.YH_FRAME_LISTS_ON_DEMAND if rng_flag.random() < 0.33 else 0

    n_levels = int(rng.integers(1, 13))
    sizes = [int(rng.choice([rng.integers(1, 70), rng.integers(70, 5000), rng.integers(5000, 200_000)], p=[0.2, 0.4, 0.4]))
             for _ in range(n_levels)]
    lo = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    n = int(lo[-1])
    par = oracle.synth_parents(lo, seed=seed)
    trs = oracle.synth_trs(n, int(lo[1]), seed=seed)
    bounds = oracle.synth_aabb(n, seed=seed)
    base = int(rng.integers(0, 1 << 20))
    cams = np.concatenate([oracle.main_camera_planes().reshape(1, 24), oracle.cascade_planes()[:3]])
    exp_w = oracle.frame_trs(trs, par, lo, n_threads=8)
    vis_all, bnd_all = oracle.cull(exp_w, bounds, cams, n_threads=8)
    with gpu_scene(ctx, lo, par, trs, bounds, 4, id_base=base) as s:
        for rnd in range(6):
            if rng.random() < 0.5:  # a random plan: cut points between levels, at most 8 levels per group
                cuts = sorted(set(rng.choice(n_levels, size=int(rng.integers(1, n_levels + 1)), replace=False).tolist()) | {n_levels - 1})
                ok, prev = True, -1
                for c in cuts:
                    ok &= c - prev <= 8
                    prev = c
                s.set_plan(cuts if ok else [])
            else:
                s.set_plan([])
            nf = int(rng.integers(1, 5))
            for _ in range(int(rng.integers(0, 4))):  # frames that are overwritten before anyone looks
                s.frame(cams[: int(rng.integers(1, 5))], _lib.YH_FRAME_ALL_DIRTY | _lib.YH_FRAME_ASYNC | on_demand())
            s.frame(cams[:nf], _lib.YH_FRAME_ALL_DIRTY | ((_lib.YH_FRAME_ASYNC | on_demand()) if rng.random() < 0.7 else 0))
            counts, bcounts = s.counts()
            for f in range(nf):
                exp_set = np.flatnonzero(vis_all[f]).astype(np.uint32) + np.uint32(base)
                assert counts[f] == exp_set.size and bcounts[f] == np.count_nonzero(bnd_all[f]), (seed, rnd, f, s.groups().tolist())
                assert np.array_equal(s.draw_list(f), exp_set), (seed, rnd, f, s.groups().tolist())
            if rnd % 3 == 0:
                assert_same_values(s.world_soa(), exp_w["m"].T, f"world seed {seed} round {rnd}")
            if rng_flag.random() < 0.5:
                # a camera moved, nothing else did (YH_FRAME_CULL_ONLY), behind whatever the frames above left pending -
                # a deferred expansion, lists on demand nobody read: another choice of frusta over the same matrices
                sel = rng_flag.permutation(4)[: int(rng_flag.integers(1, 5))]
                if rng_flag.random() < 0.5:
                    s.frame(cams[:nf], _lib.YH_FRAME_ALL_DIRTY | _lib.YH_FRAME_ASYNC | on_demand())
                s.frame(cams[sel], _lib.YH_FRAME_CULL_ONLY | (_lib.YH_FRAME_ASYNC if rng_flag.random() < 0.5 else 0))
                counts, bcounts = s.counts()
                for k, f in enumerate(sel):
                    exp_set = np.flatnonzero(vis_all[f]).astype(np.uint32) + np.uint32(base)
                    assert counts[k] == exp_set.size and bcounts[k] == np.count_nonzero(bnd_all[f]), (seed, rnd, "cull only", k)
                    assert np.array_equal(s.draw_list(k), exp_set), (seed, rnd, "cull only", k)


def test_cfg3_five_frusta_1m(ctx, oracle):
    """Small instance of configs[2]: main camera + 4 shadow cascades."""
    lo, par, trs, bounds = make_scene(oracle, 1_000_000, 1)
    planes = np.concatenate([oracle.main_camera_planes().reshape(1, 24), oracle.cascade_planes()])
    with gpu_scene(ctx, lo, par, trs, bounds, 5) as s:
        check_frame(oracle, s, lo, par, trs, bounds, planes)


@pytest.mark.parametrize("tile_words,n", [(32, 1_000_000), (64, 1_000_000), (256, 1_000_003), (256, 700_001), (64, 5_000_000)])
def test_expansion_knobs_give_the_same_lists(ctx, oracle, monkeypatch, tile_words, n):
    """The expansion's tuning knob (words per tile: read when a scene is created, YH_TILE_WORDS, the A/B of
    profiles/r02) changes the schedule only: lists, counts and boundary counts stay the oracle's.  Sizes with a ragged
    last tile, both the self-counting path (few tiles) and k_count's."""
    monkeypatch.setenv("YH_TILE_WORDS", str(tile_words))
    lo, par, trs, bounds = make_scene(oracle, n, 1)
    planes = np.concatenate([oracle.main_camera_planes().reshape(1, 24), oracle.cascade_planes()[1:3]])
    with gpu_scene(ctx, lo, par, trs, bounds, 3) as s:
        check_frame(oracle, s, lo, par, trs, bounds, planes, threads=8)


def test_cfg3_full_16m(ctx, oracle):
    """BASELINE configs[2] at full size: 16M flat entities, 5 frusta."""
    n = 16 * 1024 * 1024
    lo, par, trs, bounds = make_scene(oracle, n, 1)
    planes = np.concatenate([oracle.main_camera_planes().reshape(1, 24), oracle.cascade_planes()])
    with gpu_scene(ctx, lo, par, trs, bounds, 5) as s:
        nb = check_frame(oracle, s, lo, par, trs, bounds, planes, threads=16)
    print("boundary cases at 16M x 5 frusta:", nb)


def test_mesh_table_bounds(ctx, oracle):
    """configs[4] flavour: instanced AABBs (mesh table + id), 4M flat instances, id_base offset."""
    from yume_amd import Scene

    n, nm, base = 4_000_000, 16, 1000
    trs = oracle.synth_trs(n, n)
    table = oracle.synth_mesh_table(nm)
    ids = oracle.synth_mesh_ids(n, nm)
    planes = oracle.main_camera_planes().reshape(1, 24)
    with Scene(ctx, n, n_frusta=1, bounds_kind=2, max_meshes=nm, id_base=base) as s:
        s.upload_flat()
        s.upload_trs(trs)
        s.upload_mesh_aabbs(table)
        s.upload_mesh_ids(ids)
        s.frame(planes)
        exp = oracle.frame_trs(trs, np.full(n, 0xFFFFFFFF, np.uint32), np.array([0, n], np.uint32), n_threads=8)
        vis, bnd = oracle.cull(exp, table, planes, mesh_id=ids, n_threads=8)
        assert_same_values(s.world_soa(), exp["m"].T, "world")
        assert np.array_equal(s.draw_list(0), np.flatnonzero(vis[0]).astype(np.uint32) + np.uint32(base))
        assert s.counts()[1][0] == np.count_nonzero(bnd[0])
        # device-side fill of the same scene gives the same inputs
        s.fill_synthetic()
        s.frame(planes)
        assert np.array_equal(s.draw_list(0), np.flatnonzero(vis[0]).astype(np.uint32) + np.uint32(base))


def test_mesh_ids_default_to_zero_and_the_table_cannot_shrink_below_an_id_in_use(ctx, oracle):
    """A frame (and the consumers) before yh_upload_mesh_ids covered every entity must index the mesh table in bounds:
    ids start out as 0.  A later table with fewer meshes than the largest id in use is refused."""
    from yume_amd import Scene, _lib

    n, nm = 300_001, 200
    trs = oracle.synth_trs(n, n)
    table = oracle.synth_mesh_table(nm)
    ids = oracle.synth_mesh_ids(n, nm)
    planes = oracle.main_camera_planes().reshape(1, 24)
    exp = oracle.frame_trs(trs, np.full(n, 0xFFFFFFFF, np.uint32), np.array([0, n], np.uint32), n_threads=8)
    with Scene(ctx, n, n_frusta=1, bounds_kind=2, max_meshes=nm) as s:
        s.upload_flat()
        s.upload_trs(trs)
        s.upload_mesh_aabbs(table)
        s.upload_vertex_counts(np.arange(nm, dtype=np.uint32) + 3)
        # no ids at all: every entity is mesh 0
        s.frame(planes)
        vis0, _ = oracle.cull(exp, table, planes, mesh_id=np.zeros(n, np.uint32), n_threads=8)
        assert np.array_equal(s.draw_list(0), np.flatnonzero(vis0[0]).astype(np.uint32))
        b, bids, _, cmds = s.mesh_buckets(0)
        assert b[0, 2] == bids.size and not b[1:, 2].any() and (cmds[:, 0] == 3).all()
        # ids for the first half only: the second half stays mesh 0
        half = n // 2
        s.upload_mesh_ids(ids[:half])
        mixed = np.concatenate([ids[:half], np.zeros(n - half, np.uint32)])
        s.frame(planes)
        vis1, _ = oracle.cull(exp, table, planes, mesh_id=mixed, n_threads=8)
        assert np.array_equal(s.draw_list(0), np.flatnonzero(vis1[0]).astype(np.uint32))
        # shrinking the table below an id in use is refused; after a full re-upload of small ids it is accepted
        top = int(ids[:half].max())
        with pytest.raises(_lib.YumeHipError) as e:
            s.upload_mesh_aabbs(table[:top])
        assert e.value.code == _lib.YH_ERR_STATE
        s.upload_mesh_aabbs(table[: top + 1])
        s.upload_mesh_ids(ids % 7)
        s.upload_mesh_aabbs(table[:7])
        s.frame(planes)
        vis2, _ = oracle.cull(exp, table[:7], planes, mesh_id=ids % 7, n_threads=8)
        assert np.array_equal(s.draw_list(0), np.flatnonzero(vis2[0]).astype(np.uint32))
        b, bids, _, _ = s.mesh_buckets(0)
        assert b.shape[0] == 7 and b[:, 2].sum() == bids.size


def test_draw_list_consumer_object_data_and_draw_records(ctx, oracle):
    """GPUObjectData{model_matrix} and vkCmdDraw arguments for the visible entities, in list order —
    the reference fills them on the CPU (GAL/vulkan/RenderApi.zig:1397-1404, 1441)."""
    from yume_amd import Scene, _lib

    # hierarchy, per-entity bounds, per-entity vertex counts, two frusta
    lo, par, trs, bounds = make_scene(oracle, 300_000, 6)
    n = trs.shape[1]
    planes = np.concatenate([oracle.main_camera_planes().reshape(1, 24), oracle.cascade_planes()[:1]])
    vcount = (np.arange(n, dtype=np.uint32) * 7 + 3) % 1000
    with gpu_scene(ctx, lo, par, trs, bounds, 2) as s:
        with pytest.raises(_lib.YumeHipError) as e:
            s.object_data(0)
        assert e.value.code == _lib.YH_ERR_STATE  # no frame yet
        s.upload_vertex_counts(vcount)
        s.frame(planes)
        exp = oracle.frame_trs(trs, par, lo)
        for f, off in ((0, 0), (1, 5000)):
            lst = s.draw_list(f)
            m, c = s.object_data(f, batch_offset=off)
            assert m.shape[0] == lst.size > 0
            assert_same_values(m, exp["m"][lst], f"object data {f}")
            assert np.array_equal(c[:, 0], vcount[lst]) and np.all(c[:, 1] == 1) and np.all(c[:, 2] == 0)
            assert np.array_equal(c[:, 3], off + np.arange(lst.size, dtype=np.uint32))
        with pytest.raises(_lib.YumeHipError):
            s.upload_vertex_counts(vcount[:10])  # per-entity scenes want n counts

    # instanced: mesh table, vertex count per mesh, global ids offset by id_base
    n, nm, base = 200_000, 16, 4096
    trs = oracle.synth_trs(n, n)
    ids = oracle.synth_mesh_ids(n, nm)
    per_mesh = np.arange(nm, dtype=np.uint32) * 36 + 12
    with Scene(ctx, n, n_frusta=1, bounds_kind=2, max_meshes=nm, id_base=base) as s:
        s.upload_flat()
        s.upload_trs(trs)
        s.upload_mesh_aabbs(oracle.synth_mesh_table(nm))
        s.upload_mesh_ids(ids)
        s.upload_vertex_counts(per_mesh)
        s.frame(planes[:1])
        lst = s.draw_list(0) - np.uint32(base)
        m, c = s.object_data(0)
        exp = oracle.frame_trs(trs, np.full(n, 0xFFFFFFFF, np.uint32), np.array([0, n], np.uint32))
        assert_same_values(m, exp["m"][lst], "instanced object data")
        assert np.array_equal(c[:, 0], per_mesh[ids[lst]])


@pytest.mark.parametrize("n,nm,base", [(200_000, 16, 4096), (3_000_000, 16, 0), (50_000, 256, 7), (1000, 1, 0), (100_000, 3, 0),
                                       # more than 256 keys - one per (material, mesh) pair of a large scene: slices per wave, counters in memory
                                       (300_000, 257, 3), (2_000_000, 4096, 0), (70_001, 1000, 5), (400_000, 65536, 11)])
def test_mesh_buckets_are_a_stable_sort_of_the_draw_list(ctx, oracle, n, nm, base):
    """visible entities grouped by mesh id, ascending entity inside a mesh (= numpy's stable argsort
    of the list's mesh ids); records, matrices and the bucket table follow that order.  Replaying the reference's draw
    loop (RenderApi.zig:1410-1441) over the result binds every mesh exactly once."""
    from yume_amd import Scene, _lib

    trs = oracle.synth_trs(n, n)
    ids = oracle.synth_mesh_ids(n, nm)
    per_mesh = np.arange(nm, dtype=np.uint32) * 36 + 12
    planes = oracle.main_camera_planes().reshape(1, 24)
    with Scene(ctx, n, n_frusta=1, bounds_kind=2, max_meshes=nm, id_base=base) as s:
        s.upload_flat()
        s.upload_trs(trs)
        s.upload_mesh_aabbs(oracle.synth_mesh_table(nm))
        s.upload_mesh_ids(ids)
        s.upload_vertex_counts(per_mesh)
        with pytest.raises(_lib.YumeHipError):
            s.mesh_buckets(0)  # no frame yet
        exp = oracle.frame_trs(trs, np.full(n, 0xFFFFFFFF, np.uint32), np.array([0, n], np.uint32), n_threads=8)
        for rep in range(2):  # the second pass reuses the scratch of the first
            s.frame(planes)
            lst = s.draw_list(0)
            loc = lst - np.uint32(base)
            order = np.argsort(ids[loc], kind="stable")
            b, got_ids, m, c = s.mesh_buckets(0, batch_offset=100 * rep)
            assert np.array_equal(got_ids, lst[order])
            assert_same_values(m, exp["m"][loc[order]], "bucketed object data")
            assert np.array_equal(c[:, 0], per_mesh[ids[loc[order]]]) and np.all(c[:, 1] == 1) and np.all(c[:, 2] == 0)
            assert np.array_equal(c[:, 3], 100 * rep + np.arange(lst.size, dtype=np.uint32))
            cnt = np.bincount(ids[loc], minlength=nm).astype(np.uint32)
            assert np.array_equal(b[:, 0], np.arange(nm)) and np.array_equal(b[:, 2], cnt)
            assert np.array_equal(b[:, 1], np.cumsum(cnt) - cnt)
            # the reference's rebinding rule over the bucketed order: one vertex-buffer bind per mesh present
            meshes = ids[got_ids - np.uint32(base)]
            binds = 1 + np.count_nonzero(meshes[1:] != meshes[:-1]) if meshes.size else 0
            assert binds == np.count_nonzero(cnt)
        # the list-order consumer still works on the same buffers afterwards
        m2, c2 = s.object_data(0)
        assert_same_values(m2, exp["m"][loc], "list-order object data after bucketing")
        with pytest.raises(_lib.YumeHipError) as e:
            _lib.check(_lib.lib().yh_get_mesh_buckets(s._h, 0, None, 0, None, 0, None))
        assert e.value.code == _lib.YH_ERR_BAD_ARG


def test_mesh_buckets_follow_a_mesh_table_that_grows_from_few_keys_to_many(ctx, oracle):
    """The same scene bucketed with 8 keys (tiles sorted in LDS), then with 3000 (the many-keys path: its scratch is
    allocated on the way), then with 8 again: each time numpy's stable argsort of the list's keys."""
    from yume_amd import Scene

    n, cap = 500_000, 4096
    trs = oracle.synth_trs(n, n)
    planes = oracle.main_camera_planes().reshape(1, 24)
    with Scene(ctx, n, n_frusta=1, bounds_kind=2, max_meshes=cap) as s:
        s.upload_flat()
        s.upload_trs(trs)
        before = 0
        for nm in (8, 3000, 8):
            ids = oracle.synth_mesh_ids(n, nm, seed=77 + nm)
            table = np.tile(np.float32([-1, -1, -1, 1, 1, 1]), (nm, 1))
            before, shrinking = nm, nm < before
            if shrinking:  # a table may not shrink below the largest id in use: the ids first
                s.upload_mesh_ids(ids)
                s.upload_mesh_aabbs(table)
            else:
                s.upload_mesh_aabbs(table)
                s.upload_mesh_ids(ids)
            s.upload_vertex_counts(np.arange(nm, dtype=np.uint32) + 3)
            s.frame(planes)
            lst = s.draw_list(0)
            order = np.argsort(ids[lst], kind="stable")
            b, got_ids, m, c = s.mesh_buckets(0)
            assert b.shape[0] == nm and np.array_equal(got_ids, lst[order]), nm
            assert np.array_equal(c[:, 0], ids[lst[order]] + 3)
            cnt = np.bincount(ids[lst], minlength=nm).astype(np.uint32)
            assert np.array_equal(b[:, 2], cnt) and np.array_equal(b[:, 1], np.cumsum(cnt) - cnt)


def test_mesh_buckets_need_a_mesh_table(ctx, oracle):
    from yume_amd import _lib

    lo, par, trs, bounds = make_scene(oracle, 5000, 2)
    with gpu_scene(ctx, lo, par, trs, bounds, 1) as s:
        s.frame(oracle.main_camera_planes().reshape(1, 24))
        with pytest.raises(_lib.YumeHipError) as e:
            s.mesh_buckets(0)
        assert e.value.code == _lib.YH_ERR_STATE


@pytest.mark.parametrize("n,levels", [(300_000, 6), (5000, 1), (1_000_000, 8), (4_200_000, 8)])
def test_async_frames_defer_their_expansion_and_give_the_same_lists(ctx, oracle, n, levels):
    """YH_FRAME_ASYNC frames of TRS scenes whose first launch is small (up to 1M entities: the 4.2M tree's levels 0..4,
    with k_count's tile bases in between) leave their list expansion to the next frame's first launch (or to the
    first reader).  Whatever reads next — counts, a list, the object-data consumer, a plain yh_wait — has to see
    exactly the frame's own lists, boundary counts included, with the frusta changing from frame to frame."""
    from yume_amd import _lib

    lo, par, trs, bounds = make_scene(oracle, n, levels)
    n = trs.shape[1]
    cams = [oracle.main_camera_planes().reshape(1, 24), oracle.cascade_planes()[:1], oracle.cascade_planes()[1:2],
            oracle.cascade_planes()[2:3]]
    exp_w = oracle.frame_trs(trs, par, lo, n_threads=8)
    expect = {}

    def expected(k):
        if k not in expect:
            expect[k] = oracle.cull(exp_w, bounds, np.concatenate(cams[: 1 + k]), n_threads=8)
        return expect[k]

    flags = _lib.YH_FRAME_ALL_DIRTY | _lib.YH_FRAME_ASYNC
    with gpu_scene(ctx, lo, par, trs, bounds, 4) as s:
        readers = ["counts", "list", "object", "wait", "sync-frame"]
        for rnd in range(10):
            nf = 1 + rnd % 4
            for j in range(1 + rnd % 3):  # one to three frames back to back, then the one that is checked
                s.frame(np.concatenate(cams[: 1 + (rnd + j) % 4]), flags)
            s.frame(np.concatenate(cams[:nf]), flags)
            assert "_x<" in s.group_kernel_names()[0]  # the first launch carried the frame before's expansion
            vis, bnd = expected(nf - 1)
            how = readers[rnd % len(readers)]
            if how == "object":
                m, c = s.object_data(nf - 1)
                lst = np.flatnonzero(vis[nf - 1]).astype(np.uint32)
                assert_same_values(m, exp_w["m"][lst], "object data behind a deferred expansion")
            elif how == "wait":
                ctx.wait()
            elif how == "sync-frame":
                s.frame(np.concatenate(cams[:nf]), _lib.YH_FRAME_ALL_DIRTY)
            counts, bcounts = s.counts()
            for f in range(nf):
                assert counts[f] == np.count_nonzero(vis[f]) and bcounts[f] == np.count_nonzero(bnd[f]), (rnd, how, f)
                assert np.array_equal(s.draw_list(f), np.flatnonzero(vis[f]).astype(np.uint32)), (rnd, how, f)


@pytest.mark.parametrize("n,levels", [(200_000, 5), (4_300_000, 1), (4_200_000, 8)])
def test_lists_on_demand_frames(ctx, oracle, n, levels):
    """YH_FRAME_LISTS_ON_DEMAND (frames whose words go to the gather): a frame nobody reads runs neither k_count nor
    k_expand and is dropped by the next frame; one that is read gives exactly its own lists, counts and boundary counts
    (the boundary counters rotate through three sets that k_expand normally zeroes: some entities sit on a plane here, so
    a set that was not zeroed would show); ordinary deferred frames after a dropped one carry no stale expansion."""
    from yume_amd import _lib

    lo, par, trs, bounds = make_scene(oracle, n, levels)
    n = trs.shape[1]
    exp_w = oracle.frame_trs(trs, par, lo, n_threads=8)
    # a plane x >= -d that the box of some entity touches (t = s + r within 1e-5 of 0): boundary cases in every frame that uses it
    touching = None
    for e in range(0, n, max(1, n // 200)):
        m, b = exp_w["m"][e].astype(np.float64), bounds[e].astype(np.float64)
        c, h = 0.5 * (b[:3] + b[3:]), 0.5 * (b[3:] - b[:3])
        d = -((c[0] * m[0] + c[1] * m[4] + c[2] * m[8] + m[12]) + (abs(m[0]) * h[0] + abs(m[4]) * h[1] + abs(m[8]) * h[2]))
        cand = np.array([[1, 0, 0, d], [-1, 0, 0, 1e6], [0, 1, 0, 1e6], [0, -1, 0, 1e6], [0, 0, 1, 1e6], [0, 0, -1, 1e6]],
                        np.float32).reshape(1, 24)
        if oracle.cull(exp_w[e:e + 1], bounds[e:e + 1], cand)[1][0][0]:
            touching = cand
            break
    assert touching is not None, "no entity could be put on a plane"
    cams = [oracle.main_camera_planes().reshape(1, 24), touching, oracle.cascade_planes()[1:2]]
    expect = {}

    def expected(key):
        if key not in expect:
            expect[key] = oracle.cull(exp_w, bounds, np.concatenate([cams[k] for k in key]), n_threads=8)
        return expect[key]

    def check(s, key):
        vis, bnd = expected(key)
        counts, bcounts = s.counts()
        for f in range(len(key)):
            assert counts[f] == np.count_nonzero(vis[f]) and bcounts[f] == np.count_nonzero(bnd[f]), (key, f, counts, bcounts)
            assert np.array_equal(s.draw_list(f), np.flatnonzero(vis[f]).astype(np.uint32)), (key, f)

    demand = _lib.YH_FRAME_ALL_DIRTY | _lib.YH_FRAME_ASYNC | _lib.YH_FRAME_LISTS_ON_DEMAND
    plain = _lib.YH_FRAME_ALL_DIRTY | _lib.YH_FRAME_ASYNC
    with gpu_scene(ctx, lo, par, trs, bounds, 2) as s:
        n_groups = len(s.groups())
        for k in range(75):  # unread frames, the touching plane among them; the ring of 64 counter sets wraps
            s.frame(np.concatenate([cams[k % 3], cams[(k + 1) % 3]]), demand)
            # the transform launches alone; every 32nd frame one small launch zeroes the next 32 sets of boundary counters
            assert s.timings()["n_launches"] == n_groups + (1 if k % 32 == 0 else 0), k
        s.frame(np.concatenate([cams[1], cams[0]]), demand)
        check(s, (1, 0))
        check(s, (1, 0))  # a second read finds the same lists
        assert np.count_nonzero(expected((1, 0))[1][0]) >= 1, "the scene was meant to hold boundary cases"
        for rnd in range(4):  # dropped frames between ordinary deferred ones, read every other round
            s.frame(cams[2], demand)
            s.frame(np.concatenate([cams[rnd % 3], cams[1]]), plain)
            s.frame(np.concatenate([cams[1], cams[(rnd + 1) % 3]]), plain if rnd % 2 else demand)
            check(s, (1, (rnd + 1) % 3))
        s.frame(cams[1], demand)
        ctx.wait()  # a wait alone leaves the lists obtainable
        check(s, (1,))
        assert_same_values(s.world_soa(), exp_w["m"].T, "world")


def test_boundary_cases_are_counted(ctx, oracle):
    """Entities placed exactly on / within 1e-5 of a plane are reported in boundary_counts."""
    from yume_amd import Scene

    n = 1000
    trs = np.zeros((10, n), np.float32)
    trs[6] = 1
    trs[7:10] = 1
    trs[0] = np.linspace(-3, 3, n, dtype=np.float32)
    trs[0, 10] = -1.0
    trs[0, 11] = np.float32(-0.999995)
    pl = np.array([[1, 0, 0, 0], [-1, 0, 0, 100], [0, 1, 0, 100], [0, -1, 0, 100], [0, 0, 1, 100], [0, 0, -1, 100]],
                  np.float32).reshape(1, 24)
    bounds = np.tile(np.array([-1, -1, -1, 1, 1, 1], np.float32), (n, 1))
    lo = np.array([0, n], np.uint32)
    par = np.full(n, 0xFFFFFFFF, np.uint32)
    with gpu_scene(ctx, lo, par, trs, bounds, 1) as s:
        nb = check_frame(oracle, s, lo, par, trs, bounds, pl)
    assert nb >= 2


def test_no_cull_and_empty_visibility(ctx, oracle):
    lo, par, trs, bounds = make_scene(oracle, 30_000, 4)
    far_away = np.array([[1, 0, 0, -1e6]] * 6, np.float32).reshape(1, 24)  # nothing satisfies x >= 1e6
    from yume_amd import _lib

    with gpu_scene(ctx, lo, par, trs, bounds, 1) as s:
        s.frame(far_away)
        assert s.counts()[0][0] == 0 and s.draw_list(0).size == 0
        s.frame(None, flags=_lib.YH_FRAME_ALL_DIRTY | _lib.YH_FRAME_NO_CULL)
        assert_same_values(s.world_soa(), oracle.frame_trs(trs, par, lo)["m"].T, "world")


# ---------------------------------------------------------------------------------------- error behaviour
def test_error_codes(ctx, oracle):
    from yume_amd import Scene, _lib

    E = _lib.YumeHipError
    with pytest.raises(E) as e:
        Scene(ctx, 0)
    assert e.value.code == _lib.YH_ERR_BAD_ARG
    with pytest.raises(E) as e:
        Scene(ctx, 10, n_frusta=9)
    assert e.value.code == _lib.YH_ERR_BAD_ARG
    lo, par, trs, bounds = make_scene(oracle, 1000, 3)
    with Scene(ctx, 1000, n_levels=3, n_frusta=1) as s:
        with pytest.raises(E) as e:
            s.frame(oracle.main_camera_planes())
        assert e.value.code == _lib.YH_ERR_STATE  # hierarchy first
        bad = par.copy()
        bad[-1] = 0  # a level-2 entity pointing into level 0
        with pytest.raises(E) as e:
            s.upload_hierarchy(bad, lo)
        assert e.value.code == _lib.YH_ERR_BAD_ARG
        s.upload_hierarchy(par, lo)
        with pytest.raises(E) as e:
            s.upload_trs(trs, first=1)
        assert e.value.code == _lib.YH_ERR_BAD_ARG  # range out of bounds
        with pytest.raises(E) as e:
            s.upload_local_aos68(np.zeros(4, oracle.T68))
        assert e.value.code == _lib.YH_ERR_STATE
        with pytest.raises(E) as e:
            s.frame(np.zeros((2, 24), np.float32))
        assert e.value.code == _lib.YH_ERR_BAD_ARG  # more frusta than draw lists
        with pytest.raises(E) as e:
            s.frame(oracle.main_camera_planes(), flags=0)
        assert e.value.code == _lib.YH_ERR_UNSUPPORTED


def test_draw_list_overflow_reports_required_size(ctx, oracle):
    import ctypes as C

    from yume_amd import _lib

    lo, par, trs, bounds = make_scene(oracle, 50_000, 1)
    with gpu_scene(ctx, lo, par, trs, bounds, 1, draw_capacity=100) as s:
        s.frame(oracle.main_camera_planes())
        counts, _ = s.counts()
        exp = oracle.cull(oracle.frame_trs(trs, par, lo), bounds, oracle.main_camera_planes())[0][0].sum()
        assert counts[0] == exp > 100  # the count keeps the required size
        out = np.empty(100, np.uint32)
        written = C.c_uint64()
        rc = _lib.lib().yh_get_draw_list(s._h, 0, out.ctypes.data_as(C.c_void_p), 100, C.byref(written))
        assert rc == _lib.YH_ERR_CAPACITY and written.value == exp
    with gpu_scene(ctx, lo, par, trs, bounds, 1) as s:
        s.frame(oracle.main_camera_planes())
        out = np.empty(10, np.uint32)
        written = C.c_uint64()
        rc = _lib.lib().yh_get_draw_list(s._h, 0, out.ctypes.data_as(C.c_void_p), 10, C.byref(written))
        assert rc == _lib.YH_ERR_CAPACITY and written.value == exp


def test_reported_kernel_names_are_kernels_of_the_library(ctx, oracle):
    """bench.py looks a launch's PMC traffic up by the name the library reports for it
    (yh_scene_get_group_kernel_names): every name must be a kernel the library really contains, spelled as the
    symbol table (and rocprofv3's kernel trace) spells it."""
    import subprocess

    from yume_amd import Scene, _lib

    so = _lib.LIB_PATH
    nm = "nm" if subprocess.run(["which", "nm"], capture_output=True).returncode == 0 else "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm, "-C", so], capture_output=True, text=True, check=True).stdout
    have = {parts[2] for parts in (line.split(None, 2) for line in out.splitlines()) if len(parts) == 3}
    planes = oracle.main_camera_planes().reshape(1, 24)
    seen = set()
    cases = [(300_000, 8, 1, 0, 0), (300_000, 8, 1, 0, 1), (5_000_000, 8, 1, 0, 0), (6_000_000, 1, 1, 0, 0), (200_000, 1, 2, 0, 0),
             (50_000, 3, 0, 0, 0), (80_000, 6, 1, 1, 0), (80_000, 1, 2, 1, 0), (4_000_000, 2, 1, 1, 0)]
    for n, levels, bounds, mat4, one_per_level in cases:
        lo = oracle.synth_level_offsets(n, levels)
        with Scene(ctx, n, n_levels=levels, n_frusta=1 if bounds else 0, bounds_kind=bounds, max_meshes=4 if bounds == 2 else 0,
                   local_kind=_lib.YH_LOCAL_MAT4 if mat4 else _lib.YH_LOCAL_TRS) as s:
            s.upload_hierarchy(oracle.synth_parents(lo), lo)
            if not mat4:
                s.fill_synthetic()
            elif bounds == 2:
                s.upload_mesh_aabbs(oracle.synth_mesh_table(4))
            if one_per_level:
                s.set_grouping(1)
            for flags in ((0,) if mat4 else (_lib.YH_FRAME_ALL_DIRTY, _lib.YH_FRAME_ALL_DIRTY | _lib.YH_FRAME_ASYNC, _lib.YH_FRAME_ALL_DIRTY | _lib.YH_FRAME_ASYNC)):
                s.frame(planes if bounds else None, flags)
                names = s.group_kernel_names()
                assert len(names) == len(s.groups())
                for k in names:
                    assert k in have, k
                    seen.add(k.split("<")[0])
            ctx.wait()
    assert {"void k_frame_trs", "void k_frame_trs_big", "void k_frame_chain", "void k_frame_chain_big", "void k_frame_chain_x",
            "void k_frame_trs_x", "void k_level_mat4", "void k_chain_mat4"} <= seen, seen


def test_planes_that_share_normals_are_culled_exactly_like_planes_tested_one_by_one(ctx, oracle):
    """Frames whose planes share normals (cfg3's cascades: one light direction) are tested per distinct normal on the
    device (cull_planned); the oracle tests every plane on its own.  Adversarial plane sets: exact duplicates, exactly
    opposite normals, zero components of either sign, a NaN plane, eight identical frusta, planes that touch bounds
    exactly (boundary flags) — lists, counts and boundary counts must be the oracle's."""
    from yume_amd import Scene

    n = 200_003
    trs = oracle.synth_trs(n, n)
    trs[0:3, :50] = 0.0  # a few unit boxes at the origin, unrotated: d = +-1 makes exact boundary / signed-zero cases
    trs[3:6, :50] = 0.0
    trs[6:10, :50] = 1.0
    bounds = oracle.synth_aabb(n)
    bounds[:50] = [-1, -1, -1, 1, 1, 1]
    main = oracle.main_camera_planes().reshape(6, 4)
    casc = oracle.cascade_planes().reshape(4, 6, 4)
    rng = np.random.default_rng(5)

    def slab(nrm, lo, hi):  # two opposite planes: lo <= n.x <= hi
        nrm = np.asarray(nrm, np.float32)
        return [np.concatenate([nrm, [np.float32(-lo)]]), np.concatenate([-nrm, [np.float32(hi)]])]

    axis = slab([1, 0, 0], -30, 30) + slab([0, 1, 0], -20, 60) + slab([0, 0, 1], -90, 90)
    axis_negzero = [p.copy() for p in axis]
    for p in axis_negzero:
        p[:3][p[:3] == 0] = -0.0  # the same normals with the other zero sign: must still group, must still agree
    touching = slab([1, 0, 0], 1.0, 30) + slab([0, 1, 0], -1.0, -1.0) + slab([0, 0, 1], -90, 90)
    nanp = [p.copy() for p in axis]
    nanp[2][1] = np.nan
    sets = {
        "cascades + main": np.concatenate([main[None], casc]),
        "eight identical frusta": np.repeat(main[None], 8, axis=0),
        "axis slabs, zero signs mixed": np.stack([np.stack(axis), np.stack(axis_negzero), casc[1]]),
        "touching planes": np.stack([np.stack(touching), np.stack(axis)]),
        "a NaN normal component": np.stack([np.stack(nanp), np.stack(axis), np.stack(axis)]),
        "one orthographic frustum": casc[2][None],
        "shuffled plane order": np.stack([casc[0][rng.permutation(6)], casc[3][rng.permutation(6)], main[rng.permutation(6)]]),
    }
    flat_par, flat_lo = np.full(n, 0xFFFFFFFF, np.uint32), np.array([0, n], np.uint32)
    world = oracle.frame_trs(trs, flat_par, flat_lo, n_threads=8)
    with Scene(ctx, n, n_frusta=8) as s:
        s.upload_flat()
        s.upload_trs(trs)
        s.upload_aabb(np.ascontiguousarray(bounds[:, :3].T), np.ascontiguousarray(bounds[:, 3:].T))
        for name, planes in sets.items():
            planes = np.ascontiguousarray(planes, np.float32).reshape(-1, 24)
            F = planes.shape[0]
            s.frame(planes)
            with np.errstate(all="ignore"):
                vis, bnd = oracle.cull(world, bounds, planes, n_threads=8)
            counts, bcounts = s.counts()
            for f in range(F):
                assert np.array_equal(s.draw_list(f), np.flatnonzero(vis[f]).astype(np.uint32)), (name, f)
                assert bcounts[f] == np.count_nonzero(bnd[f]), (name, f, "boundary count")
            if name == "touching planes":
                assert bcounts[0] > 0, "the touching set is meant to produce boundary cases"


@pytest.mark.parametrize("n,levels", [(4096, 8), (300_000, 6), (1_000_000, 8)])
def test_lds_staged_parents_give_identical_results(ctx, oracle, n, levels):
    """The one-level kernel with its parents staged in LDS (yh_scene_set_lds_parents: the variant north_star names)
    against the oracle and against the default L1/L2 gather, one level per launch; also with UNSORTED siblings and with
    parents whose children span more rows than the staging area holds (the kernel's fallback)."""
    from yume_amd import Scene

    lo = oracle.synth_level_offsets(n, levels)
    par = oracle.synth_parents(lo)
    rng = np.random.default_rng(3)
    shuffled = par.copy()
    for k in range(1, levels):  # the same multiset of parents per level, in random order
        seg = slice(int(lo[k]), int(lo[k + 1]))
        shuffled[seg] = rng.permutation(shuffled[seg])
    trs = oracle.synth_trs(n, int(lo[1]))
    bounds = oracle.synth_aabb(n)
    planes = oracle.main_camera_planes().reshape(1, 24)
    for parents in (par, shuffled):
        exp = oracle.frame_trs(trs, parents, lo, n_threads=8)
        vis, _ = oracle.cull(exp, bounds, planes, n_threads=8)
        with Scene(ctx, n, n_levels=levels, n_frusta=1) as s:
            s.upload_hierarchy(parents, lo)
            s.upload_trs(trs)
            s.upload_aabb(np.ascontiguousarray(bounds[:, :3].T), np.ascontiguousarray(bounds[:, 3:].T))
            s.set_grouping(1)
            s.set_lds_parents(True)
            s.frame(planes)
            assert any("k_frame_trs_lds" in k for k in s.group_kernel_names())
            assert_same_values(s.world_soa(), exp["m"].T, "world, parents staged in LDS")
            assert np.array_equal(s.draw_list(0), np.flatnonzero(vis[0]).astype(np.uint32))
            s.set_lds_parents(False)
            s.frame(planes)
            assert not any("k_frame_trs_lds" in k for k in s.group_kernel_names())
            assert_same_values(s.world_soa(), exp["m"].T, "world, parents gathered")
