"""yh_frame_static — the entry point that allocates nothing (its stage-3 scratch lies in a file-scope device array
of the library) — against yh_frame on the same scenes, through the C ABI, tolerance 0.

Two calls in a row on one GPU, the second on a smaller scene: the array keeps the first call's contents, so the
second call checks that everything that must start at zero is cleared and that nothing stale is read.  Collected
last (test_zz_*): the round's end runs `pytest -x`."""
import ctypes

import numpy as np
import pytest

from .test_gpu_parity import assert_same_values

pytestmark = pytest.mark.gpu


def _results(s, n_frusta):
    counts, bcounts = s.counts()
    return (s.world_soa().view(np.uint32).copy(), [s.draw_list(f).copy() for f in range(n_frusta)],
            counts[:n_frusta].copy(), bcounts[:n_frusta].copy())


def _same(a, b):
    assert np.array_equal(a[0], b[0]), "world matrices differ (bit patterns)"
    for f, (x, y) in enumerate(zip(a[1], b[1])):
        assert np.array_equal(x, y), f"draw list of frustum {f} differs"
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])


def test_frame_static_equals_frame_on_two_calls_in_a_row(ctx, oracle):
    from yume_amd import Scene, _lib

    cams = np.concatenate([oracle.main_camera_planes().reshape(1, 24), oracle.cascade_planes()[1:3]])
    # call 1: 1.3M entities in a 7-level tree, 3 frusta (several launches, 64-word expand tiles, the k_count path off)
    n1 = 1_300_003
    lo1 = oracle.synth_level_offsets(n1, 7)
    # call 2, smaller: 70k flat instances with a mesh table, 1 frustum, an id offset
    n2 = 70_001
    with Scene(ctx, n1, n_levels=7, n_frusta=3) as big, \
            Scene(ctx, n2, n_levels=1, n_frusta=1, bounds_kind=_lib.YH_BOUNDS_MESH_TABLE, max_meshes=16, id_base=900) as small:
        big.upload_hierarchy(oracle.synth_parents(lo1), lo1)
        big.fill_synthetic()
        small.upload_flat()
        small.fill_synthetic(index_base=5)
        big.frame(cams)
        want_big = _results(big, 3)
        small.frame(cams[:1])
        want_small = _results(small, 1)
        assert want_big[2].min() > 0 and want_small[2][0] > 0  # the comparison is not vacuous
        # the two static calls, back to back (the second finds the first one's words and counters in the array)
        big.frame_static(cams)
        got_big = _results(big, 3)
        small.frame_static(cams[:1])
        got_small = _results(small, 1)
        _same(got_big, want_big)
        _same(got_small, want_small)
        # and once more the other way round, enqueued without a host sync in between
        flags = _lib.YH_FRAME_ALL_DIRTY | _lib.YH_FRAME_ASYNC
        small.frame_static(cams[:1], flags)
        big.frame_static(cams[:2], flags)
        big.frame(cams[:2])
        want2 = _results(big, 2)
        big.frame_static(cams[:2], flags)
        _same(_results(big, 2), want2)
        _same(_results(small, 1), want_small)
        # a deferred expansion left by an asynchronous yh_frame is flushed, not lost, by the static call behind it
        big.frame(cams, flags)
        big.frame_static(cams)
        _same(_results(big, 3), want_big)
        # a camera moved, nothing else did: YH_FRAME_CULL_ONLY through the static call equals yh_frame's
        big.frame(cams[[2, 0]], _lib.YH_FRAME_CULL_ONLY)
        want3 = _results(big, 2)
        big.frame_static(cams[[2, 0]], _lib.YH_FRAME_CULL_ONLY)
        _same(_results(big, 2), want3)
        big.frame_static(cams, _lib.YH_FRAME_CULL_ONLY | _lib.YH_FRAME_ASYNC)
        _same(_results(big, 3), want_big)
        # the words of a static frame are not the scene's: the pipelined gather must refuse them
        with pytest.raises(_lib.YumeHipError) as e:
            big.allgather_draw_lists_async()
        assert e.value.code == _lib.YH_ERR_STATE


def test_frame_static_matches_the_oracle_and_mat4_scenes(ctx, oracle):
    """The static call against the CPU oracle directly, on imported flecs columns (literal three-branch rule)."""
    from yume_amd import Scene, _lib

    n, L = 50_000, 5
    lo = oracle.synth_level_offsets(n, L)
    par = oracle.synth_parents(lo)
    local = oracle.compose_batch(oracle.synth_trs(n, int(lo[1])))
    bounds = oracle.synth_aabb(n)
    planes = oracle.main_camera_planes().reshape(1, 24)
    world = np.zeros(n, oracle.T68)
    world["m"][:, [0, 5, 10, 15]] = 1
    world["dirty"] = 1
    local["dirty"] = 1
    with Scene(ctx, n, n_levels=L, n_frusta=1, local_kind=_lib.YH_LOCAL_MAT4) as s:
        s.upload_hierarchy(par, lo)
        s.upload_local_aos68(local)
        s.upload_world_aos68(world)
        s.upload_aabb(np.ascontiguousarray(bounds[:, :3].T), np.ascontiguousarray(bounds[:, 3:].T))
        s.frame_static(planes, 0)
        oracle.frame_aos68(world, local, par, lo, oracle.identity68(), 1)
        vis, bnd = oracle.cull(world, bounds, planes)
        got = s.world_aos68()
        assert np.array_equal(got["m"].view(np.uint32), world["m"].view(np.uint32)) and not got["dirty"].any()
        assert np.array_equal(s.draw_list(0), np.flatnonzero(vis[0]).astype(np.uint32))
        assert s.counts()[1][0] == np.count_nonzero(bnd[0])


def test_frame_static_rejects_what_does_not_fit(ctx, oracle):
    from yume_amd import Scene, _lib

    n = 17_000_000  # x 8 frusta: 17.0 MB of visibility words alone, the array holds 16 MiB
    planes = np.repeat(oracle.main_camera_planes().reshape(1, 24), 8, axis=0)
    with Scene(ctx, n, n_levels=1, n_frusta=8, bounds_kind=_lib.YH_BOUNDS_MESH_TABLE, max_meshes=4, draw_capacity=1024) as s:
        s.upload_flat()
        s.fill_synthetic()
        with pytest.raises(_lib.YumeHipError) as e:
            s.frame_static(planes)
        assert e.value.code == _lib.YH_ERR_CAPACITY and "static" in str(e.value)
        s.frame_static(planes[:7])  # 7 frusta fit
        c7 = s.counts()[0][:7].copy()
        s.frame(planes[:7])
        assert np.array_equal(s.counts()[0][:7], c7) and c7[0] > 0


def test_frame_static_frames_replayed_from_a_hip_graph(ctx, oracle):
    """What the entry point that allocates nothing is for: yh_frame_static calls (YH_FRAME_ASYNC) captured into a
    hipGraph on the context's stream (yh_stream) and replayed.  The replay must give the lists of the captured frames'
    LAST frame - also after the scene's inputs were changed between two replays (the graph reads the device arrays)."""
    from yume_amd import Scene, _lib

    hip = ctypes.CDLL("libamdhip64.so")
    for fn in ("hipStreamBeginCapture", "hipStreamEndCapture", "hipGraphInstantiate", "hipGraphLaunch", "hipGraphExecDestroy",
               "hipGraphDestroy"):
        getattr(hip, fn).restype = ctypes.c_int
    n, levels = 300_000, 6
    lo = oracle.synth_level_offsets(n, levels)
    par = oracle.synth_parents(lo)
    trs = oracle.synth_trs(n, int(lo[1]))
    bounds = oracle.synth_aabb(n)
    cams = [oracle.main_camera_planes().reshape(1, 24), oracle.cascade_planes()[1:2]]
    flags = _lib.YH_FRAME_ALL_DIRTY | _lib.YH_FRAME_ASYNC
    with Scene(ctx, n, n_levels=levels, n_frusta=2) as s:
        s.upload_hierarchy(par, lo)
        s.upload_trs(trs)
        s.upload_aabb(np.ascontiguousarray(bounds[:, :3].T), np.ascontiguousarray(bounds[:, 3:].T))
        s.frame_static(np.concatenate(cams))  # (everything the call sets up lazily exists before the capture)
        stream = ctypes.c_void_p(ctx.stream())
        graph, gexec = ctypes.c_void_p(), ctypes.c_void_p()
        assert hip.hipStreamBeginCapture(stream, 1) == 0  # hipStreamCaptureModeThreadLocal
        try:
            s.frame_static(cams[1], flags)
            s.frame_static(np.concatenate(cams), flags)
        finally:
            rc = hip.hipStreamEndCapture(stream, ctypes.byref(graph))
        assert rc == 0 and graph.value
        assert hip.hipGraphInstantiate(ctypes.byref(gexec), graph, None, None, ctypes.c_ulonglong(0)) == 0
        try:
            for rnd in range(3):
                if rnd:  # other inputs under the same graph
                    trs = trs.copy()
                    trs[0] += np.float32(3.0 * rnd)
                    s.upload_trs(trs)
                assert hip.hipGraphLaunch(gexec, stream) == 0
                ctx.wait()
                exp_w = oracle.frame_trs(trs, par, lo, n_threads=8)
                vis, bnd = oracle.cull(exp_w, bounds, np.concatenate(cams), n_threads=8)
                counts, bcounts = s.counts()
                for f in range(2):
                    assert counts[f] == np.count_nonzero(vis[f]) and bcounts[f] == np.count_nonzero(bnd[f]), (rnd, f)
                    assert np.array_equal(s.draw_list(f), np.flatnonzero(vis[f]).astype(np.uint32)), (rnd, f)
                assert_same_values(s.world_soa(), exp_w["m"].T, f"world, replay {rnd}")
        finally:
            hip.hipGraphExecDestroy(gexec)
            hip.hipGraphDestroy(graph)


def test_a_second_context_is_turned_away_while_a_static_frame_is_in_flight(ctx, oracle):
    """All contexts of one GPU share that GPU's static array.  A second context that asks for a static frame while the
    first one's has not been waited for gets YH_ERR_STATE (and leaves the first frame's results intact) instead of
    racing with it; after the first context has been waited for, the second one's call goes through, and both give what
    yh_frame gives."""
    from yume_amd import Context, Scene, _lib

    planes = oracle.main_camera_planes().reshape(1, 24)
    n = 300_000
    lo = oracle.synth_level_offsets(n, 4)
    par = oracle.synth_parents(lo)
    other = Context(0)
    try:
        with Scene(ctx, n, n_levels=4, n_frusta=1) as a, Scene(other, n, n_levels=4, n_frusta=1) as b:
            for s in (a, b):
                s.upload_hierarchy(par, lo)
                s.fill_synthetic()
            a.frame(planes)
            want = _results(a, 1)
            ctx.wait()
            other.wait()
            a.frame_static(planes, _lib.YH_FRAME_ALL_DIRTY | _lib.YH_FRAME_ASYNC)  # in flight: nobody has waited for it
            with pytest.raises(_lib.YumeHipError) as e:
                b.frame_static(planes, _lib.YH_FRAME_ALL_DIRTY | _lib.YH_FRAME_ASYNC)
            assert e.value.code == _lib.YH_ERR_STATE and "yh_frame_static" in str(e.value) and "another context" in str(e.value)
            a.frame_static(planes, _lib.YH_FRAME_ALL_DIRTY | _lib.YH_FRAME_ASYNC)  # the owner itself may go on: one stream orders its calls
            b.frame(planes)  # yh_frame is always welcome: its scratch is the scene's own
            _same(_results(a, 1), want)  # (reading results waits for context `ctx`: the array is free again)
            b.frame_static(planes)       # synchronous: claimed and released inside the call
            _same(_results(b, 1), want)
            a.frame_static(planes)
            _same(_results(a, 1), want)
            # errors of the static entry point name it (they used to say "yh_frame:")
            with pytest.raises(_lib.YumeHipError) as e:
                a.frame_static(planes, 0)
            assert "yh_frame_static:" in str(e.value)
    finally:
        other.close()
