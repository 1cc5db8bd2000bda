"""What a frame costs the drop-in when only a camera moved (Camera.view_projection changed, no row set, no structural
change: the frame culls again, recomputes nothing): 1M entities inside the reference's flecs, cameras and mesh bounds read
from the ECS.  usage: python tools/camera_move_cost.py"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as oracle  # noqa: E402
from tests.test_gpu_flecs_system import GPU, write_scene  # noqa: E402

n, levels = 1_000_000, 8
lo = oracle.synth_level_offsets(n, levels)
par = oracle.synth_parents(lo)
local = oracle.compose_batch(oracle.synth_trs(n, int(lo[1])))
bounds = oracle.synth_aabb(n)
planes = oracle.main_camera_planes().reshape(1, 24)
frames = [[], []]
for k in range(6):  # camera 0 looks from a slightly different place every frame
    view = oracle.look_at([0.0 + k, 40.0, 150.0 - 2 * k], [0, 0, 0], [0, 1, 0])
    proj = oracle.perspective(np.radians(75.0), 2.0, 0.1, 200.0)
    vp = oracle.mul(proj, view)
    frames.append([(0, 5, np.asarray(vp, np.float32).reshape(16))])
frames.append([])
with tempfile.TemporaryDirectory() as d:
    scene = os.path.join(d, "scene.bin")
    write_scene(scene, lo, par, local["m"], oracle.identity68()["m"][0], planes, bounds, frames)
    e = dict({k: v for k, v in os.environ.items() if not k.startswith("YFH_")}, YFH_TIMES="1", YFH_ECS_INPUTS="1")
    r = subprocess.run([GPU, scene, os.path.join(d, "out.bin")], env=e, capture_output=True, text=True, timeout=1200)
    ms = [float(line.split(": ")[1].split(" ms")[0]) for line in r.stderr.splitlines() if line.startswith("frame ")]
    print(f"n={n}: frames in which only camera 0 moved, ms: " + ", ".join(f"{x:.3f}" for x in ms[2:8])
          + f"; a frame with nothing to do: {ms[1]:.3f}, {ms[8]:.3f}", flush=True)
    print("\n".join(line for line in r.stderr.splitlines() if line.startswith("stages ") and line.split()[1] in ("3:", "4:")))
