"""CPU tests of the N > 1 host logic: subtree-closed partitioning, shard extraction, and the end-of-frame exchange
between two gloo ranks in both of the library's forms - index lists (yh_allgather_draw_lists) and visibility words
expanded by every rank (yh_allgather_draw_lists_async, the one the benchmark times) - with tests/gloo_transport.py as
the transport.  The compute inside a rank is the oracle here: what is under test is the sharding, the id bases, the
padding of uneven shards and the gathered order, not the kernels (those: tests/test_gpu_shards.py on one GPU)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(oracle, n=60_000, levels=8):
    lo = oracle.synth_level_offsets(n, levels)
    par = oracle.synth_parents(lo)
    trs = oracle.synth_trs(n, int(lo[1]))
    bounds = oracle.synth_aabb(n)
    return lo, par, trs, bounds


@pytest.mark.parametrize("parts", [1, 2, 3, 8])
def test_partition_is_subtree_closed_and_balanced(oracle, parts):
    from yume_amd import shard

    lo, par, _, _ = build(oracle)
    n = par.shape[0]
    part_of, sizes = shard.partition_subtrees(par, lo, parts)
    assert sizes.sum() == n and np.array_equal(np.bincount(part_of, minlength=parts), sizes.astype(np.int64))
    child = np.arange(int(lo[1]), n)
    assert np.array_equal(part_of[child], part_of[par[child]])  # no parent is ever on another GPU
    # greedy largest-first: no part exceeds the mean by more than the largest subtree
    root_of = np.arange(n, dtype=np.int64)
    for k in range(1, len(lo) - 1):
        s = slice(int(lo[k]), int(lo[k + 1]))
        root_of[s] = root_of[par[s]]
    biggest = np.bincount(root_of).max()
    assert sizes.max() - n / parts <= biggest


@pytest.mark.parametrize("parts", [2, 4])
def test_shards_reproduce_the_global_result(oracle, parts):
    from yume_amd import shard

    lo, par, trs, bounds = build(oracle)
    planes = oracle.main_camera_planes().reshape(1, 24)
    world = oracle.frame_trs(trs, par, lo)
    vis, _ = oracle.cull(world, bounds, planes)
    part_of, sizes = shard.partition_subtrees(par, lo, parts)
    seen = []
    for p in range(parts):
        lp, llo, g = shard.shard_extract(par, lo, part_of, p, sizes[p])
        assert llo[-1] == sizes[p] and np.all(np.diff(g.astype(np.int64)) > 0)  # global order kept
        for k in range(1, len(llo) - 1):  # still depth-sorted with siblings sorted by parent
            q = lp[llo[k] : llo[k + 1]]
            assert q.size == 0 or (q.min() >= llo[k - 1] and q.max() < llo[k] and np.all(np.diff(q.astype(np.int64)) >= 0))
        w = oracle.frame_trs(np.ascontiguousarray(trs[:, g]), lp, llo)
        assert np.array_equal(w["m"].view(np.uint32), world["m"][g].view(np.uint32))
        v, _ = oracle.cull(w, bounds[g], planes)
        seen.append(g[np.flatnonzero(v[0])])
    assert np.array_equal(np.sort(np.concatenate(seen)), np.flatnonzero(vis[0]).astype(np.uint32))


def test_partition_rejects_bad_hierarchy(oracle):
    from yume_amd import _lib, shard

    lo, par, _, _ = build(oracle, 5000, 4)
    bad = par.copy()
    bad[-1] = 0
    with pytest.raises(_lib.YumeHipError) as e:
        shard.partition_subtrees(bad, lo, 2)
    assert e.value.code == _lib.YH_ERR_BAD_ARG


def _rank_main(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gloo_transport

    from oracle import pyoracle as oracle
    from yume_amd import shard

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, par, trs, bounds = build(oracle, 40_000, 6)
        planes = np.concatenate([oracle.main_camera_planes().reshape(1, 24), oracle.cascade_planes()[:2]])
        part_of, sizes = shard.partition_subtrees(par, lo, world)
        lp, llo, g = shard.shard_extract(par, lo, part_of, rank, sizes[rank])
        w = oracle.frame_trs(np.ascontiguousarray(trs[:, g]), lp, llo)
        v, _ = oracle.cull(w, bounds[g], planes)
        local = [g[np.flatnonzero(v[f])].astype(np.uint32) for f in range(planes.shape[0])]
        counts, lists = gloo_transport.allgather_lists(local)
        # every rank must end up with every frustum's complete list
        world_m = oracle.frame_trs(trs, par, lo)
        vis, _ = oracle.cull(world_m, bounds, planes)
        ok = counts.shape == (world, planes.shape[0]) and int(counts[rank, 0]) == len(local[0])
        for f in range(planes.shape[0]):
            ok = ok and np.array_equal(np.sort(lists[f]), np.flatnonzero(vis[f]).astype(np.uint32))
            ok = ok and int(counts[:, f].sum()) == int(vis[f].sum())
        # The words protocol, as the shards really are on N GPUs: LOCAL slots, id_base = the sizes of the ranks before,
        # rows of the shorter shard zero-padded; the gathered list is rank-major and ascending inside a rank, and maps
        # back through every rank's global_of_local to the unsharded visible set.
        id_base = int(sizes[:rank].sum())
        wcounts, wlists = gloo_transport.allgather_words(gloo_transport.pack_words(v), int(sizes[rank]), id_base)
        g_all = np.concatenate([shard.shard_extract(par, lo, part_of, r, sizes[r])[2] for r in range(world)])  # shard-major -> global
        ok = ok and np.array_equal(wcounts, counts)
        for f in range(planes.shape[0]):
            ids = wlists[f].astype(np.int64)
            ok = ok and np.array_equal(np.sort(g_all[ids]), np.flatnonzero(vis[f]).astype(np.uint32))
            bounds_r = np.concatenate([[0], np.cumsum(wcounts[:, f])])
            for r in range(world):  # rank r's part: ids inside [base_r, base_r + size_r), ascending
                part = ids[bounds_r[r]:bounds_r[r + 1]]
                base = int(sizes[:r].sum())
                ok = ok and bool(np.all(np.diff(part) > 0)) and (part.size == 0 or (part[0] >= base and part[-1] < base + int(sizes[r])))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_two_rank_allgather_of_draw_lists_gloo():
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(results) == [(0, True), (1, True)]
