"""Per-100-frame wall time of back-to-back cfg2 frames (looks for slow phases): python tools/phase_probe.py [chunks] [sync_every]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from yume_amd import Context, Scene, _lib, synth  # noqa: E402

chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n, levels = 1_000_000, 8
ctx = Context(0)
lo = synth.level_offsets(n, levels)
s = Scene(ctx, n, n_levels=levels, n_frusta=1)
s.upload_hierarchy(synth.parents(lo), lo)
s.fill_synthetic()
planes = synth.main_camera_planes().reshape(1, 24)
flags = _lib.YH_FRAME_ALL_DIRTY | _lib.YH_FRAME_ASYNC
out = []
t_start = time.perf_counter()
for c in range(chunks):
    t0 = time.perf_counter()
    for _ in range(100):
        s.frame(planes, flags)
    t_sub = time.perf_counter() - t0
    ctx.wait()
    out.append((time.perf_counter() - t_start, (time.perf_counter() - t0) / 100 * 1e6, t_sub / 100 * 1e6))
for t, us, sub in out:
    print(f"t={t*1e3:7.1f} ms  {us:6.1f} us/frame  (submit {sub:5.1f})")
