"""GPU tests against the committed golden fixtures (tests/golden/*.npz) and of the RCCL entry points with a
one-rank communicator (the only RCCL configuration a one-GPU box can run; the N > 1 logic is covered on CPU
by tests/test_sharding.py and at round end by the driver's multi-GPU bench)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", ["cfg1_flat10k", "tree4096_l8", "frusta5_flat4096"])
def test_gpu_matches_golden_fixture(ctx, oracle, name):
    from tests.golden.make_golden import scene_for
    from yume_amd import Scene

    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    sc = scene_for(oracle, name)  # inputs are regenerated from the seed; expected outputs come from the fixture
    F = sc["planes"].shape[0]
    with Scene(ctx, sc["n"], n_levels=len(sc["level_offsets"]) - 1, n_frusta=F) as s:
        s.upload_hierarchy(sc["parent"], sc["level_offsets"])
        s.upload_trs(sc["trs"])
        s.upload_aabb(np.ascontiguousarray(sc["bounds"][:, :3].T), np.ascontiguousarray(sc["bounds"][:, 3:].T))
        s.frame(sc["planes"])
        assert np.array_equal(s.world_soa().T, g["world"])  # values equal (the fixture holds no NaN)
        counts, bcounts = s.counts()
        for f in range(F):
            assert np.array_equal(s.draw_list(f), g[f"vis{f}"]), f"frustum {f}"
            assert counts[f] == g[f"vis{f}"].size and bcounts[f] == g[f"bnd{f}"].size


def test_rccl_one_rank_allgather(oracle):
    from yume_amd import Context, Scene

    n = 200_000
    lo = oracle.synth_level_offsets(n, 5)
    par = oracle.synth_parents(lo)
    planes = np.concatenate([oracle.main_camera_planes().reshape(1, 24), oracle.cascade_planes()[:1]])
    with Context(0) as c:
        c.comm_init(1, 0, Context.comm_unique_id())
        with Scene(c, n, n_levels=5, n_frusta=2, id_base=7_000_000) as s:
            s.upload_hierarchy(par, lo)
            s.fill_synthetic()
            s.frame(planes)
            counts, _ = s.counts()
            all_counts = s.allgather_draw_lists()
            assert all_counts.shape == (1, 2) and np.array_equal(all_counts[0], counts[:2])
            for f in range(2):
                local = s.draw_list(f)
                gathered = s.gathered_draw_list(f, all_counts[:, f].sum())
                assert np.array_equal(gathered, local) and (local.size == 0 or local.min() >= 7_000_000)
            # a second frame + gather reuses the buffers
            s.frame(planes)
            assert np.array_equal(s.allgather_draw_lists()[0], counts[:2])


@pytest.mark.parametrize("on_demand", [False, True])
def test_rccl_one_rank_allgather_without_host_sync(oracle, on_demand):
    """The pipelined exchange (visibility words all-gathered on the collective stream, expanded on a third stream):
    same lists as the scene's own, frame after frame, with and without host syncs in between.  on_demand: the frames
    leave their own lists unexpanded (YH_FRAME_LISTS_ON_DEMAND, what bench.py's multi-GPU step passes): the gather needs
    only the words, and the scene's own lists, read after it, are still the last frame's."""
    from yume_amd import Context, Scene, _lib

    n = 300_000
    lo = oracle.synth_level_offsets(n, 6)
    par = oracle.synth_parents(lo)
    cams = [oracle.main_camera_planes().reshape(1, 24), oracle.cascade_planes()[:1], oracle.cascade_planes()[1:2]]
    with Context(0) as c:
        c.comm_init(1, 0, Context.comm_unique_id())
        with Scene(c, n, n_levels=6, n_frusta=2, id_base=11) as s:
            s.upload_hierarchy(par, lo)
            s.fill_synthetic()
            flags = _lib.YH_FRAME_ALL_DIRTY | _lib.YH_FRAME_ASYNC | (_lib.YH_FRAME_LISTS_ON_DEMAND if on_demand else 0)
            for k in range(21):  # back-to-back frames with different second frusta: each gather must see its own frame
                # (21 rounds of 1-2 frames: the ring of word sets wraps several times)
                planes = np.concatenate([cams[0], cams[1 + k % 2]])
                s.frame(planes, flags)
                s.allgather_draw_lists_async()
                if k % 2 == 1:
                    s.frame(planes, flags)  # a frame enqueued behind a pending gather: its expansion has to wait
                    s.allgather_draw_lists_async()
                counts, _ = s.counts()
                gc = s.gathered_counts(2)
                assert gc.shape == (1, 2) and np.array_equal(gc[0], counts[:2])
                for f in range(2):
                    assert np.array_equal(s.gathered_draw_list(f, gc[0, f]), s.draw_list(f))
            # 50 frames and gathers enqueued without any host sync (the ring of word sets wraps six times)
            for k in range(50):
                planes = np.concatenate([cams[0], cams[1 + (k * 7 // 3) % 2]])
                s.frame(planes, flags)
                s.allgather_draw_lists_async()
            counts, _ = s.counts()
            gc = s.gathered_counts(2)
            assert np.array_equal(gc[0], counts[:2])
            for f in range(2):
                assert np.array_equal(s.gathered_draw_list(f, gc[0, f]), s.draw_list(f))
            # the synchronous exchange still works afterwards
            assert np.array_equal(s.allgather_draw_lists()[0], counts[:2])
            assert np.array_equal(s.gathered_draw_list(1, counts[1]), s.draw_list(1))


@pytest.mark.parametrize("n,levels,ranks,base", [(100_000, 3, 8, 0), (1_000_000, 1, 4, 5000), (777, 1, 3, 0), (20_000_000, 1, 2, 0)])
def test_gather_expansion_with_several_ranks_on_one_gpu(ctx, oracle, n, levels, ranks, base):
    """The no-sync gather sends visibility words and every rank expands all ranks' words itself.  One GPU cannot run
    a real R-rank exchange, so yh_gather_selftest feeds that expansion R copies of this scene's own words (rows
    zero-padded as for a shorter shard), rank r owning ids id_base + r*n: the gathered list has to be the scene's own
    list R times, shifted, and the per-rank counts the scene's counts."""
    from yume_amd import Scene

    planes = np.concatenate([oracle.main_camera_planes().reshape(1, 24), oracle.cascade_planes()[:2]])
    lo = oracle.synth_level_offsets(n, levels)
    with Scene(ctx, n, n_levels=levels, n_frusta=3, id_base=base) as s:
        s.upload_hierarchy(oracle.synth_parents(lo), lo)
        s.fill_synthetic()
        for nf in (3, 2):  # the second pass with fewer frusta than the scene was created for
            s.frame(planes[:nf])
            counts, _ = s.counts()
            s.gather_selftest(ranks)
            gc = s.gathered_counts(nf, n_ranks=ranks)
            assert gc.shape == (ranks, nf) and np.all(gc == counts[:nf][None, :])
            for f in range(nf):
                own = s.draw_list(f).astype(np.int64)
                exp = np.concatenate([own + r * n for r in range(ranks)]).astype(np.uint32)
                assert np.array_equal(s.gathered_draw_list(f, gc[:, f].sum()), exp), f"frustum {f}"


def test_allgather_without_communicator_is_a_state_error(ctx, oracle):
    from yume_amd import Scene, _lib

    with Scene(ctx, 1000, n_frusta=1) as s:
        s.upload_flat()
        s.fill_synthetic()
        s.frame(oracle.main_camera_planes())
        with pytest.raises(_lib.YumeHipError) as e:
            s.allgather_draw_lists()
        assert e.value.code == _lib.YH_ERR_STATE
