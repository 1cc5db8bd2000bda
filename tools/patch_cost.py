"""What a structural frame costs the drop-in as a function of how many tables it touches: k last-level leaves moved under k
other parents (2k tables touched) in one frame, k = 1 .. 4096, at 1M entities / 300k tables inside the reference's flecs.
usage: python tools/patch_cost.py [n]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as oracle  # noqa: E402
from tests.test_gpu_flecs_system import GPU, reparent_arg, write_scene  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
levels = 8
rng = np.random.default_rng(3)
lo = oracle.synth_level_offsets(n, levels)
par = oracle.synth_parents(lo)
local = oracle.compose_batch(oracle.synth_trs(n, int(lo[1])))
bounds = oracle.synth_aabb(n)
planes = oracle.main_camera_planes().reshape(1, 24)
ks = [1, 16, 128, 512, 2048, 4096]
leaves = rng.permutation(np.arange(int(lo[-2]), n))
frames, at = [[], []], 0
for k in ks:
    moved = leaves[at:at + k]
    at += k
    new_par = rng.integers(int(lo[-3]), int(lo[-2]), k)
    frames.append([(int(e), 3, reparent_arg(int(p))) for e, p in zip(moved, new_par)])
    frames.append([])
with tempfile.TemporaryDirectory() as d:
    scene = os.path.join(d, "scene.bin")
    write_scene(scene, lo, par, local["m"], oracle.identity68()["m"][0], planes, bounds, frames)
    for label, env in (("patched (default)", {}), ("rebuilt from the observers' lists (setPatching(0))", {"YFH_NO_PATCH": "1"})):
        e = dict({k: v for k, v in os.environ.items() if not k.startswith("YFH_")}, YFH_TIMES="1", **env)
        r = subprocess.run([GPU, scene, os.path.join(d, "out.bin")], env=e, capture_output=True, text=True, timeout=1200)
        ms = [float(line.split(": ")[1].split(" ms")[0]) for line in r.stderr.splitlines() if line.startswith("frame ")]
        print(f"{label}: n={n}; leaves moved per frame -> ms: " + ", ".join(f"{k}: {ms[2 + 2 * i]:.2f}" for i, k in enumerate(ks)), flush=True)
