"""Times the draw-list consumers (yh_build_object_data, yh_build_mesh_buckets) on a flat instanced scene."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from yume_amd import Context, Scene, _lib as L  # noqa: E402
from yume_amd import synth  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 64_000_000
nm = int(sys.argv[2]) if len(sys.argv) > 2 else 16
with Context(0) as ctx, Scene(ctx, n, n_frusta=1, bounds_kind=2, max_meshes=nm) as s:
    s.upload_flat()
    s.fill_synthetic()
    s.upload_vertex_counts(np.arange(nm, dtype=np.uint32) * 36 + 12)
    planes = synth.main_camera_planes().reshape(1, 24)
    s.frame(planes)
    k = int(s.counts()[0][0])
    for name, fn in (("object_data", L.lib().yh_build_object_data), ("mesh_buckets", L.lib().yh_build_mesh_buckets)):
        for _ in range(3):
            L.check(fn(s._h, 0, 0))
        ctx.wait()
        t0 = time.perf_counter()
        for _ in range(20):
            L.check(fn(s._h, 0, 0))
        ctx.wait()
        dt = (time.perf_counter() - t0) / 20
        print(f"{name}: n={n} visible={k} meshes={nm}: {dt*1e3:.3f} ms, {k*(64+16+4+64)/dt/1e9:.0f} GB/s (148 B per visible entity)")
