"""Sweep launch plans (which levels share a launch, chain or tree kernel) on a synthetic tree and time whole frames.
usage: python tools/plan_sweep.py <n_entities> [levels] [frames]"""
import itertools
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from yume_amd import Context, Scene, _lib, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 8
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 100
ctx = Context(0)
lo = synth.level_offsets(n, levels)
par = synth.parents(lo)
planes = synth.main_camera_planes().reshape(1, 24)
s = Scene(ctx, n, n_levels=levels, n_frusta=1)
s.upload_hierarchy(par, lo)
s.fill_synthetic()
flags = _lib.YH_FRAME_ALL_DIRTY | _lib.YH_FRAME_ASYNC


def run(k):
    for _ in range(k):
        s.frame(planes, flags)
    ctx.wait()


def timed():
    run(10)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        run(frames)
        best = min(best, (time.perf_counter() - t) / frames)
    return best * 1e6


results = []
ref_counts = None
for cuts in itertools.product([0, 1], repeat=levels - 1):
    last = [k for k in range(levels - 1) if cuts[k]] + [levels - 1]
    sizes = np.diff([-1] + last)
    for kinds in [None]:
        s.set_plan(last)
        us = timed()
        c, _ = s.counts()
        if ref_counts is None:
            ref_counts = int(c[0])
        assert int(c[0]) == ref_counts, (last, kinds, c[0], ref_counts)
        results.append((us, last, "single" if sizes.max() == 1 else "chain"))
results.sort(key=lambda r: r[0])
print(f"n={n} levels={levels} level sizes={np.diff(lo).tolist()}")
for us, last, kind in results[:15]:
    print(f"{us:8.2f} us  {n / us / 1e3:7.2f} Gent/s  groups end at {last}  {kind}")
print("...")
for us, last, kind in results:
    if len(last) in (1, levels) or last == [levels - 2, levels - 1]:
        print(f"{us:8.2f} us  {n / us / 1e3:7.2f} Gent/s  groups end at {last}  {kind}")
s.set_plan()
print("planner:", s.groups().tolist(), f"{timed():.2f} us")
