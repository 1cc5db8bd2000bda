"""PCIe side of the flecs drop-in's all-rows frame, without flecs: 1M-entity Mat4 scene, root moved, then the changed
WorldTransform rows through yh_download_changed_rows_aos68 into page-locked / ordinary memory, and through round 1's
yh_download_changed_aos68 (both components, one chunk at a time).  usage: python tools/download_probe.py [n]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as oracle  # noqa: E402  (measurement script: the oracle generates the scene)
from yume_amd import Context, Scene, _lib as L  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ctx = Context(0)
lo = oracle.synth_level_offsets(n, 8)
par = oracle.synth_parents(lo)
local = oracle.compose_batch(oracle.synth_trs(n, int(lo[1])))
world = np.zeros(n, oracle.T68)
root = oracle.identity68()
lib = L.lib()
with Scene(ctx, n, n_levels=8, n_frusta=1, local_kind=L.YH_LOCAL_MAT4) as s:
    s.upload_hierarchy(par, lo)
    s.upload_local_aos68(local)
    s.upload_world_aos68(world)
    s.track_changes(True)
    planes = oracle.main_camera_planes().reshape(1, 24)
    s.frame(planes, flags=0)
    pinned = lib.yh_host_alloc(ctx._h, n * 68)
    pinned2 = lib.yh_host_alloc(ctx._h, n * 68)
    slots = np.zeros(n, np.uint32)
    plain = np.zeros(n, oracle.T68)
    plain2 = np.zeros(n, oracle.T68)
    wr = C.c_uint64()
    for label, fn in (
        ("streamed, WorldTransform rows only, page-locked", lambda: lib.yh_download_changed_rows_aos68(s._h, 0, slots.ctypes.data, pinned, n, C.byref(wr), None, None)),
        ("streamed, WorldTransform rows only, ordinary memory", lambda: lib.yh_download_changed_rows_aos68(s._h, 0, slots.ctypes.data, plain.ctypes.data, n, C.byref(wr), None, None)),
        ("round 1: both components, page-locked", lambda: lib.yh_download_changed_aos68(s._h, slots.ctypes.data, pinned, pinned2, n, C.byref(wr))),
        ("round 1: both components, ordinary memory", lambda: lib.yh_download_changed_aos68(s._h, slots.ctypes.data, plain.ctypes.data, plain2.ctypes.data, n, C.byref(wr))),
    ):
        ts = []
        for rep in range(5):
            root["m"][0] = oracle.compose([1, 2, 3 + rep], [0, 0.7071068, 0, 0.7071068], [2, 2, 2])
            root["dirty"] = 1
            s.upload_root_aos68(root)
            s.frame(planes, flags=0)
            t0 = time.perf_counter()
            rc = fn()
            ts.append((time.perf_counter() - t0) * 1e3)
            assert rc == 0 and wr.value == n, (rc, wr.value)
        print(f"{label:55s} n={n}: ms {' '.join(f'{t:.3f}' for t in ts)}  ({n * 68 / min(ts) / 1e6:.1f} GB/s per component)")
    lib.yh_host_free(ctx._h, pinned)
    lib.yh_host_free(ctx._h, pinned2)
