"""Frame time of the literal-rule path (Mat4 columns, k_level_mat4 + dirty sweep): python tools/mat4_bench.py [n] [levels] [all|dirty|clean]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from yume_amd import Context, Scene, _lib, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = Context(0)
lo = synth.level_offsets(n, levels)
par = synth.parents(lo)
planes = synth.main_camera_planes().reshape(1, 24)
# locals: composed on a TRS scene of the same synthetic inputs, then handed to the Mat4 scene as columns
with Scene(ctx, n, n_levels=levels, n_frusta=1) as t:
    t.upload_hierarchy(par, lo)
    t.fill_synthetic()
    local_m = t.compose()
    mins, maxs = t.download_aabb()
T68 = np.dtype([("m", np.float32, 16), ("dirty", np.uint8), ("pad", np.uint8, 3)])
col = np.zeros(n, T68)
col["m"] = local_m.T
col["dirty"] = 1
s = Scene(ctx, n, n_levels=levels, n_frusta=1, local_kind=_lib.YH_LOCAL_MAT4)
s.upload_hierarchy(par, lo)
s.upload_aabb(mins, maxs)
s.upload_local_aos68(col)


def timed(flags, frames=200):
    for _ in range(10):
        s.frame(planes, flags | _lib.YH_FRAME_ASYNC)
    ctx.wait()
    t0 = time.perf_counter()
    for _ in range(frames):
        s.frame(planes, flags | _lib.YH_FRAME_ASYNC)
    ctx.wait()
    return (time.perf_counter() - t0) / frames * 1e6


mode = sys.argv[3] if len(sys.argv) > 3 else "all"  # "dirty" / "clean" / "cull": one kind of frame only, change tracking on (profiling runs)
if mode != "all":
    s.track_changes(True)
    if mode == "cull":
        s.frame(planes, 0)
    us = timed({"dirty": _lib.YH_FRAME_ALL_DIRTY, "cull": _lib.YH_FRAME_CULL_ONLY}.get(mode, 0), 100)
    what = {"dirty": "every row dirty", "cull": "cull only"}.get(mode, "nothing dirty")
    print(f"n={n} L={levels} change tracking on, {what}: {us:8.1f} us/frame, kernels {s.group_kernel_names()}")
    sys.exit(0)
for track in (False, True):
    s.track_changes(track)
    a = timed(_lib.YH_FRAME_ALL_DIRTY)
    b = timed(0)
    print(f"n={n} L={levels} change tracking {'on ' if track else 'off'}: all dirty {a:8.1f} us/frame ({n / a / 1e3:.1f} Gent/s), "
          f"nothing dirty {b:8.1f} us/frame ({n / b / 1e3:.1f} Gent/s), "
          f"cull only {(c := timed(_lib.YH_FRAME_CULL_ONLY)):8.1f} us/frame ({n / c / 1e3:.1f} Gent/s; 88.1 B/entity read: {n * 88.14 / c / 1e6:.2f} TB/s)")
