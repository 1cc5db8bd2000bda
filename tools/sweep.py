"""Per-launch timing sweep over level groupings / workloads (GPU box only)."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yume_amd import Context, Scene, _lib, synth

def run(ctx, name, n, levels, planes, bounds=1, groupings=(0,), steps=100):
    lo = synth.level_offsets(n, levels)
    par = synth.parents(lo)
    F = planes.shape[0]
    with Scene(ctx, n, n_levels=levels, n_frusta=F, bounds_kind=bounds, max_meshes=16 if bounds == 2 else 0) as s:
        s.upload_hierarchy(par, lo)
        s.fill_synthetic()
        for g in groupings:
            s.set_grouping(g)
            fl = _lib.YH_FRAME_ALL_DIRTY | _lib.YH_FRAME_ASYNC
            for _ in range(10): s.frame(planes, fl)
            ctx.wait()
            t0 = time.perf_counter()
            for _ in range(steps): s.frame(planes, fl)
            ctx.wait()
            dt = (time.perf_counter() - t0) / steps
            s.set_kernel_timing(True)
            for _ in range(steps): s.frame(planes, fl)
            ms, cnt = s.group_timings()
            s.set_kernel_timing(False)
            gr = s.groups()
            print(f"{name} g={g}: frame {dt*1e6:8.1f} us  {n/dt/1e9:6.2f} Gent/s | launches:",
                  " ".join(f"[L{q[0]}-{q[1]} n={q[3]-q[2]} {ms[i]*1e3:.1f}us {(q[3]-q[2])/ms[i]/1e6:.1f}G/s]" for i, q in enumerate(gr)), f"[expand {ms[63]*1e3:.1f}us]", flush=True)

if __name__ == "__main__":
    main = synth.main_camera_planes().reshape(1, 24)
    five = np.concatenate([main, synth.cascade_planes()])
    which = sys.argv[1:] or ["cfg2", "cfg3", "cfg4", "flat1"]
    with Context(0) as ctx:
        if "cfg2s" in which: run(ctx, "cfg2 1M L8", 1_000_000, 8, main, groupings=(1, 8))
        if "cfg4s" in which: run(ctx, "cfg4 12.5M L8", 12_500_000, 8, main, groupings=(1,))
        if "cfg2" in which: run(ctx, "cfg2 1M L8", 1_000_000, 8, main, groupings=(1, 2, 4, 8, 0))
        if "flat1" in which: run(ctx, "flat 16M F1", 16 * 1024 * 1024, 1, main)
        if "cfg3" in which: run(ctx, "cfg3 16M F5", 16 * 1024 * 1024, 1, five)
        if "cfg4" in which: run(ctx, "cfg4 12.5M L8", 12_500_000, 8, main, groupings=(1, 2, 4, 0))
        if "cfg5" in which: run(ctx, "cfg5 64M mesh", 64_000_000, 1, main, bounds=2, steps=20)
