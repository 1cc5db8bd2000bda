"""A/B of the in-launch hand-off: cfg2 (1M entities, 8 levels, 1 frustum), frames enqueued with
YH_FRAME_ASYNC, as bench.py does.  A = the product: levels 0..4 (+ the previous frame's expansion) and levels 5..7 as two
launches.  B = the same work in ONE grid, bottom waves polling a counter the top workgroups release (k_frame_handoff,
-DYH_AB_HANDOFF build, YUME_HIP_LIB=tools/ab/libyume_hip_handoff.so).  Parity first (world rows, draw list, counts of B
against A, bit for bit), then wall time per frame over 3 x 2000 frames each, interleaved."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yume_amd import Context, Scene, _lib, synth  # noqa: E402

n, levels = int(os.environ.get("AB_N", 1_000_000)), 8
ctx = Context(0)
lo = synth.level_offsets(n, levels)
parent = synth.parents(lo)
planes = synth.main_camera_planes().reshape(1, 24)
scene = Scene(ctx, n, n_levels=levels, n_frusta=1, bounds_kind=1)
scene.upload_hierarchy(parent, lo)
scene.fill_synthetic()
scene.set_grouping(0)
flags = _lib.YH_FRAME_ALL_DIRTY | _lib.YH_FRAME_ASYNC
print("groups", scene.groups())


def run(k):
    for i in range(k):
        scene.frame(planes, flags)
        if (i + 1) % 256 == 0:
            ctx.wait()
    ctx.wait()


def result():
    run(3)
    w = scene.world_soa().copy()
    lst = scene.draw_list(0).copy()
    c = scene.counts()
    return w, lst, c


os.environ["YH_HANDOFF"] = "0"
wa, la, ca = result()
os.environ["YH_HANDOFF"] = "1"
wb, lb, cb = result()
names = scene.group_kernel_names()
print("B kernels:", names)
assert "handoff" in names[0], "the experiment kernel did not run"
gave_up = ctypes.c_uint32(0)
_lib.lib().yh_ab_handoff_gave_up.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
_lib.lib().yh_ab_handoff_gave_up(ctx.handle, ctypes.byref(gave_up))
print("waves that gave up waiting:", gave_up.value)
same = wa.view(np.uint32).tobytes() == wb.view(np.uint32).tobytes() and np.array_equal(la, lb) and np.array_equal(np.asarray(ca[0]), np.asarray(cb[0]))
print("parity B == A:", same, "visible", len(la))
if not same or gave_up.value:
    if not os.environ.get("AB_TIMING_ONLY"):
        sys.exit(1)
# one probe frame per setting: every workgroup's first thread stores four readings of the 100 MHz clock in slots of its own
_lib.lib().yh_ab_handoff_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
for naps in (1, 65535):
    os.environ["YH_HANDOFF"] = "1"
    os.environ["YH_HO_NAPS"] = str(naps)
    run(50)
    for _ in range(3):
        os.environ["YH_HO_PROBE"] = "1"
        run(2)  # (the first frame after a wait has no pending expansion and runs as two launches)
        os.environ["YH_HO_PROBE"] = "0"
        st = np.zeros((8192, 4), np.uint64)
        shape = np.zeros(3, np.uint32)
        rc = _lib.lib().yh_ab_handoff_stamps(ctx.handle, st.ctypes.data, shape.ctypes.data)
        assert rc == 0
        nx, nt, nb = (int(v) for v in shape)
        st = st[: nx + nt + nb].astype(np.int64)
        t0 = st[:, 0].min()
        us = lambda v: (int(v) - int(t0)) / 100.0
        ex, top, bot = st[:nx], st[nx:nx + nt], st[nx + nt:]
        bot = bot[bot[:, 1] > 0]  # (workgroups past the launch's last tile store nothing after their start)
        print(f"probe frame ({'no wait: timing only' if naps == 65535 else 'waiting'}), us after the first workgroup's start: "
              f"{nx} expansion workgroups start {us(ex[:, 0].min()):.1f}..{us(ex[:, 0].max()):.1f}, done by {us(ex[:, 3].max()):.1f}; "
              f"{nt} top workgroups start {us(top[:, 0].min()):.1f}..{us(top[:, 0].max()):.1f}, done by {us(top[:, 3].max()):.1f}; "
              f"{len(bot)} bottom workgroups start {us(bot[:, 0].min()):.1f}..{us(bot[:, 0].max()):.1f} "
              f"({int((bot[:, 0] < top[:, 3].max()).sum())} of them before the last top workgroup is done), "
              f"past their wait {us(bot[:, 2].min()):.1f}..{us(bot[:, 2].max()):.1f}, median time from past-the-wait to done "
              f"{float(np.median(bot[:, 3] - bot[:, 2])) / 100:.1f}, median time from start to the wait {float(np.median(bot[:, 1] - bot[:, 0])) / 100:.1f}, "
              f"all done by {us(bot[:, 3].max()):.1f}; top workgroups done at (percentiles 10/50/90/99) "
              + "/".join(f"{(np.percentile(top[:, 3], q) - t0) / 100:.1f}" for q in (10, 50, 90, 99)))
        run(20)
for rep in range(3):
    for mode, naps in (("0", 1), ("1", 0), ("1", 1), ("1", 4), ("1", 65535)):
        os.environ["YH_HANDOFF"] = mode
        os.environ["YH_HO_NAPS"] = str(naps)
        run(200)
        t0 = time.perf_counter()
        run(2000)
        dt = time.perf_counter() - t0
        print(f"rep {rep} {f'B hand-off, one grid, {naps:2d} naps of s_sleep 8 per poll' if mode == '1' else 'A two launches'}: {dt / 2000 * 1e6:.2f} us per frame")
_lib.lib().yh_ab_handoff_gave_up(ctx.handle, ctypes.byref(gave_up))
print("waves that gave up waiting (after timing):", gave_up.value)
