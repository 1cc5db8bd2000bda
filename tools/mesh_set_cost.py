"""What a frame costs the drop-in when k entities get another Mesh (an OnSet of the Mesh component, no structural change):
1M entities / ~500k tables inside the reference's flecs, cameras and mesh bounds read from the ECS.  usage: python tools/mesh_set_cost.py"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as oracle  # noqa: E402
from tests.test_gpu_flecs_system import GPU, reparent_arg, write_scene  # noqa: E402

n, levels = 1_000_000, 8
rng = np.random.default_rng(5)
lo = oracle.synth_level_offsets(n, levels)
par = oracle.synth_parents(lo)
local = oracle.compose_batch(oracle.synth_trs(n, int(lo[1])))
bounds = oracle.synth_aabb(n)
planes = oracle.main_camera_planes().reshape(1, 24)
ks = [1, 100, 10000]
frames = [[], []]
for k in ks:
    who = rng.choice(np.arange(n)[np.arange(n) % 4 != 0], k, replace=False)  # entities that have a Mesh (e % 4 != 0)
    frames.append([(int(e), 4, reparent_arg(int(e) % 3 + 1)) for e in who])
    frames.append([])
with tempfile.TemporaryDirectory() as d:
    scene = os.path.join(d, "scene.bin")
    write_scene(scene, lo, par, local["m"], oracle.identity68()["m"][0], planes, bounds, frames)
    e = dict({k: v for k, v in os.environ.items() if not k.startswith("YFH_")}, YFH_TIMES="1", YFH_ECS_INPUTS="1")
    r = subprocess.run([GPU, scene, os.path.join(d, "out.bin")], env=e, capture_output=True, text=True, timeout=1200)
    ms = [float(line.split(": ")[1].split(" ms")[0]) for line in r.stderr.splitlines() if line.startswith("frame ")]
    print(f"n={n}: entities whose Mesh was set in the frame -> ms: " + ", ".join(f"{k}: {ms[2 + 2 * i]:.2f}" for i, k in enumerate(ks))
          + f"; a frame in which nothing moved: {ms[1]:.2f}", flush=True)
