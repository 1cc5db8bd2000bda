"""The multi-GPU path end to end on ONE GPU, with real shards (no 8-GPU node is available to the builder):

    one hierarchy -> yh_partition_subtrees into R parts -> yh_shard_extract -> R scenes on the one GPU (id_base =
    shard_base, as §8e numbers global ids) -> yh_frame on each -> the R DIFFERENT sets of visibility words, rows of
    different length, through the gather's expansion (yh_gather_selftest_shards: exactly the buffer a rank holds after
    the ncclAllGather, shorter rows zero-padded as send_pad pads them) -> the gathered lists, mapped back through
    global_of_local, against the oracle's visible set of the UNSHARDED scene; per-rank counts; world matrices.

What one GPU cannot show is the transport itself (RCCL between ranks): that is the one-rank communicator tests and the
driver's multi-GPU bench."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scene(oracle, n, levels, seed=3):
    lo = oracle.synth_level_offsets(n, levels)
    par = oracle.synth_parents(lo, seed=seed)
    trs = oracle.synth_trs(n, int(lo[1]), seed=seed)
    bounds = oracle.synth_aabb(n, seed=seed)
    return lo, par, trs, bounds


@pytest.mark.parametrize("n,levels,R,uneven", [(1_200_000, 8, 2, False), (1_200_000, 8, 3, True), (1_200_000, 8, 8, True),
                                               (70_000, 5, 8, False), (300_000, 1, 3, True)])
def test_partition_shards_frames_and_gather_reproduce_the_unsharded_scene(ctx, oracle, n, levels, R, uneven):
    from yume_amd import Scene, shard

    lo, par, trs, bounds = _scene(oracle, n, levels)
    planes = np.concatenate([oracle.main_camera_planes().reshape(1, 24), oracle.cascade_planes()[2:3]])
    F = planes.shape[0]
    world = oracle.frame_trs(trs, par, lo, n_threads=8)
    vis, bnd = oracle.cull(world, bounds, planes, n_threads=8)

    if uneven:  # shards of clearly different length: cut into R + 1 parts and give the last one to part 0 as well
        part_of, sizes = shard.partition_subtrees(par, lo, R + 1)
        part_of[part_of == R] = 0
        sizes = np.bincount(part_of, minlength=R).astype(np.uint64)
    else:
        part_of, sizes = shard.partition_subtrees(par, lo, R)
    assert sizes.sum() == n and sizes.min() > 0
    bases = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)  # shard_base[g] (§8e)

    scenes, maps = [], []
    try:
        for r in range(R):
            lp, llo, g = shard.shard_extract(par, lo, part_of, r, sizes[r])
            s = Scene(ctx, int(sizes[r]), n_levels=levels, n_frusta=F, id_base=int(bases[r]))
            scenes.append(s)
            maps.append(g)
            s.upload_hierarchy(lp, llo)
            s.upload_trs(np.ascontiguousarray(trs[:, g]))
            b = bounds[g]
            s.upload_aabb(np.ascontiguousarray(b[:, :3].T), np.ascontiguousarray(b[:, 3:].T))
        # the device-side generator fills a shard from its entities' GLOBAL indices: the same inputs as the upload
        with Scene(ctx, int(sizes[0]), n_levels=levels, n_frusta=F) as filled:
            filled.upload_hierarchy(*shard.shard_extract(par, lo, part_of, 0, sizes[0])[:2])
            filled.fill_synthetic(seed=3, global_of_local=maps[0])
            assert np.array_equal(filled.download_trs().view(np.uint32), np.ascontiguousarray(trs[:, maps[0]]).view(np.uint32))
            mins, maxs = filled.download_aabb()
            assert np.array_equal(np.concatenate([mins, maxs]).T, bounds[maps[0]])
        for s in scenes:
            s.frame(planes)
        if uneven:
            assert len({(int(x) + 63) // 64 for x in sizes}) > 1, "the word rows are meant to differ in length"
        # every shard's world matrices are the unsharded scene's rows (value for value)
        for s, g in zip(scenes, maps):
            assert np.array_equal(s.world_soa().T.view(np.uint32), world["m"][g].view(np.uint32))
        Scene.gather_selftest_shards(scenes)
        gc = scenes[0].gathered_counts(F, n_ranks=R)
        for f in range(F):
            want_global = np.flatnonzero(vis[f]).astype(np.uint32)
            for r in range(R):  # per-rank counts = the unsharded visible set restricted to the shard
                assert gc[r, f] == np.count_nonzero(vis[f][maps[r]]) == scenes[r].counts()[0][f]
            got = scenes[0].gathered_draw_list(f, gc[:, f].sum()).astype(np.int64)
            assert got.size == want_global.size
            rank = np.searchsorted(bases, got, side="right") - 1
            assert np.all(np.diff(rank) >= 0), "rank-major order"
            glob = np.empty(got.size, np.uint32)
            for r in range(R):
                m = rank == r
                local = got[m] - bases[r]
                assert np.all(np.diff(local) > 0), "ascending inside a rank"
                glob[m] = maps[r][local]
                assert np.array_equal(local.astype(np.uint32) + np.uint32(bases[r]), scenes[r].draw_list(f))
            assert np.array_equal(np.sort(glob), want_global), f"frustum {f}: gathered set != the oracle's unsharded visible set"
        # boundary cases: counted per shard, they add up to the unsharded scene's
        for f in range(F):
            assert sum(int(s.counts()[1][f]) for s in scenes) == np.count_nonzero(bnd[f])
    finally:
        for s in scenes:
            s.close()


def test_gather_selftest_shards_rejects_mismatched_shards(ctx, oracle):
    from yume_amd import Scene, _lib

    planes = np.concatenate([oracle.main_camera_planes().reshape(1, 24), oracle.cascade_planes()[:1]])
    with Scene(ctx, 5000, n_frusta=2) as a, Scene(ctx, 3000, n_frusta=2) as b:
        for s in (a, b):
            s.upload_flat()
            s.fill_synthetic()
        a.frame(planes)
        with pytest.raises(_lib.YumeHipError) as e:  # b has run no frame
            Scene.gather_selftest_shards([a, b])
        assert e.value.code == _lib.YH_ERR_STATE
        b.frame(planes[:1])
        with pytest.raises(_lib.YumeHipError) as e:  # different numbers of frusta
            Scene.gather_selftest_shards([a, b])
        assert e.value.code == _lib.YH_ERR_BAD_ARG
