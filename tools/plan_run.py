"""Run K frames of one launch plan (for rocprofv3): python tools/plan_run.py <n> <levels> <last,last,..> <kind 0|1|2> [frames] [frusta]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from yume_amd import Context, Scene, _lib, synth  # noqa: E402

n, levels = int(sys.argv[1]), int(sys.argv[2])
last = [int(x) for x in sys.argv[3].split(",")] if sys.argv[3] != "auto" else []
kind = int(sys.argv[4])
frames = int(sys.argv[5]) if len(sys.argv) > 5 else 200
F = int(sys.argv[6]) if len(sys.argv) > 6 else 1
ctx = Context(0)
lo = synth.level_offsets(n, levels)
par = synth.parents(lo)
planes = np.concatenate([synth.main_camera_planes().reshape(1, 24), synth.cascade_planes()])[:F]
s = Scene(ctx, n, n_levels=levels, n_frusta=F)
s.upload_hierarchy(par, lo)
s.fill_synthetic()
if last:
    s.set_plan(last)
flags = _lib.YH_FRAME_ALL_DIRTY | _lib.YH_FRAME_ASYNC
for _ in range(20):
    s.frame(planes, flags)
ctx.wait()
t = time.perf_counter()
for _ in range(frames):
    s.frame(planes, flags)
t_submit = time.perf_counter() - t
ctx.wait()
t_all = time.perf_counter() - t
print(f"plan {s.groups().tolist()} kind {kind}: {t_all / frames * 1e6:.2f} us/frame, host submit {t_submit / frames * 1e6:.2f} us/frame")
