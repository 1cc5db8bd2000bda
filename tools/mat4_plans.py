"""Launch plans of large Mat4 (flecs-column) scenes against the planner's: python tools/mat4_plans.py <entities>"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from yume_amd import Context, Scene, _lib, synth
n, levels = int(sys.argv[1]), 8
ctx = Context(0)
lo = synth.level_offsets(n, levels); par = synth.parents(lo)
planes = synth.main_camera_planes().reshape(1, 24)
with Scene(ctx, n, n_levels=levels, n_frusta=1) as t:
    t.upload_hierarchy(par, lo); t.fill_synthetic(); local_m = t.compose(); mins, maxs = t.download_aabb()
T68 = np.dtype([("m", np.float32, 16), ("dirty", np.uint8), ("pad", np.uint8, 3)])
col = np.zeros(n, T68); col["m"] = local_m.T; col["dirty"] = 1
s = Scene(ctx, n, n_levels=levels, n_frusta=1, local_kind=_lib.YH_LOCAL_MAT4)
s.upload_hierarchy(par, lo); s.upload_aabb(mins, maxs); s.upload_local_aos68(col); s.track_changes(True)
def timed(flags, frames=150):
    for _ in range(10): s.frame(planes, flags | _lib.YH_FRAME_ASYNC)
    ctx.wait(); t0 = time.perf_counter()
    for _ in range(frames): s.frame(planes, flags | _lib.YH_FRAME_ASYNC)
    ctx.wait(); return (time.perf_counter() - t0) / frames * 1e6
for rep in range(2):
    for plan in ([], [1, 4, 5, 6, 7], [4, 5, 6, 7], [2, 4, 5, 6, 7], [3, 5, 6, 7], [1, 3, 4, 5, 6, 7]):
        s.set_plan(plan)
        print(n, plan or "planner", [int(q[1]) for q in s.groups()], "dirty %.1f us  clean %.1f us" % (timed(_lib.YH_FRAME_ALL_DIRTY), timed(0)), flush=True)
