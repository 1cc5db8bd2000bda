"""The drop-in's all-rows frame, repeated (how steady is it?): 1M entities, the root moved every other frame.
usage: python tools/flecs_allrows_repeat.py [n] [repeats]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as oracle  # noqa: E402  (measurement script: the oracle generates the scene)
from tests.test_gpu_flecs_system import GPU, write_scene  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
lo = oracle.synth_level_offsets(n, 8)
par = oracle.synth_parents(lo)
local = oracle.compose_batch(oracle.synth_trs(n, int(lo[1])))
bounds = oracle.synth_aabb(n)
planes = oracle.main_camera_planes().reshape(1, 24)
frames = [[]]
for k in range(reps):
    frames += [[(-1, 1, oracle.compose([1, 2, 3 + k], [0, 0.7071068, 0, 0.7071068], [2, 2, 2]))], []]
with tempfile.TemporaryDirectory() as d:
    scene = os.path.join(d, "scene.bin")
    write_scene(scene, lo, par, local["m"], oracle.identity68()["m"][0], planes, bounds, frames)
    for label, env in (("16 helpers (default)", {}), ("8 helpers", {"YFH_THREADS": "8"}), ("4 helpers", {"YFH_THREADS": "4"}),
                       ("16 helpers, ECS inputs", {"YFH_ECS_INPUTS": "1"}), ("8 helpers, ECS inputs", {"YFH_ECS_INPUTS": "1", "YFH_THREADS": "8"})):
        e = dict({k: v for k, v in os.environ.items() if not k.startswith("YFH_")}, YFH_TIMES="1", **env)
        r = subprocess.run([GPU, scene, os.path.join(d, "out.bin")], env=e, capture_output=True, text=True, timeout=1200)
        t = [float(line.split(": ")[1].split(" ms")[0]) for line in r.stderr.splitlines() if line.startswith("frame ")]
        moved = t[1::2]
        print(f"{label:22s} n={n}: all-rows frames ms " + " ".join(f"{x:.2f}" for x in moved) + f"   median {np.median(moved):.2f}  max {max(moved):.2f}", flush=True)
