"""Writes the 1M-entity scene script of tools/flecs_bench.py's first six frames to a file, for runs of the harness under
a profiler: python tools/flecs_scene_1m.py <path>   (frames: build, idle, 1 % of the rows edited, idle, root moved, idle)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as oracle  # noqa: E402  (measurement script)
from tests.test_gpu_flecs_system import write_scene  # noqa: E402

n = 1_000_000
rng = np.random.default_rng(1)
lo = oracle.synth_level_offsets(n, 8)
par = oracle.synth_parents(lo)
local = oracle.compose_batch(oracle.synth_trs(n, int(lo[1])))
m = oracle.compose([1, 2, 3], [0, 0.7071068, 0, 0.7071068], [2, 2, 2])
frames = [[], [], [(int(e), 0, m) for e in np.unique(rng.integers(0, n, 10_000))], [], [(-1, 1, m)], []]
write_scene(sys.argv[1], lo, par, local["m"], oracle.identity68()["m"][0], oracle.main_camera_planes().reshape(1, 24), oracle.synth_aabb(n), frames)
