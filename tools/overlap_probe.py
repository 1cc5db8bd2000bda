"""A/B for "hide the top-of-tree launch": how much of a small, latency-bound launch sequence (levels 0..4 of the cfg2
tree: 121k entities) disappears when it runs on a second stream beside the bandwidth-bound frames of the full scene?
Two contexts on one GPU (one stream each), frames enqueued alternately without host syncs; compared with each scene
alone.  Upper bound of what double-buffered top levels on a second stream could gain, before the cross-stream
event record / wait packets (2.4 + 3.6 us each way, measured in round 1) are paid.
    python tools/overlap_probe.py [frames]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yume_amd import Context, Scene, _lib, synth  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
planes = synth.main_camera_planes().reshape(1, 24)
flags = _lib.YH_FRAME_ALL_DIRTY | _lib.YH_FRAME_ASYNC


def make(ctx, n, levels):
    lo = synth.level_offsets(n, levels)
    s = Scene(ctx, n, n_levels=levels, n_frusta=1)
    s.upload_hierarchy(synth.parents(lo), lo)
    s.fill_synthetic()
    return s


a, b = Context(0), Context(0)
lo_full = synth.level_offsets(1_000_000, 8)
n_top = int(lo_full[5])
full = make(a, 1_000_000, 8)
top = make(b, n_top, 5)  # the same sizes as levels 0..4 of the full tree


def run(scenes, k):
    for s in scenes:
        s.ctx.wait()
    t0 = time.perf_counter()
    for i in range(k):
        for s in scenes:
            s.frame(planes, flags)
        if (i + 1) % 256 == 0:
            for s in scenes:
                s.ctx.wait()
    for s in scenes:
        s.ctx.wait()
    return (time.perf_counter() - t0) / k * 1e6


for s in (full, top):
    run([s], 200)
t_full = min(run([full], K) for _ in range(3))
t_top = min(run([top], K) for _ in range(3))
t_both = min(run([full, top], K) for _ in range(3))
print(f"full cfg2 scene alone (levels 0..4 | 5..7):        {t_full:7.2f} us per frame")
print(f"its top levels alone ({n_top} entities, 5 levels):  {t_top:7.2f} us per frame")
print(f"both, alternately enqueued on two streams:         {t_both:7.2f} us per pair")
print(f"sum of the two alone: {t_full + t_top:.2f} us; hidden by running side by side: {t_full + t_top - t_both:.2f} us "
      f"({(t_full + t_top - t_both) / t_top * 100:.0f} % of the small sequence)")
print("a double-buffered top launch on a second stream would add one event record (2.4 us) and one satisfied wait packet "
      "(3.6 us) to the frames' stream per frame (DESIGN.md 6, measured in round 1): net = hidden - 6.0 us")
