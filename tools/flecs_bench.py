"""End-to-end frame time inside the REAL flecs (oracle/_ref harnesses): the reference's CPU systems against the GPU
system, whole-column and incremental modes.  usage: python tools/flecs_bench.py [n] [levels]
Frames: 0 everything dirty (first frame), 1 nothing dirty, 2 1 % local edits, 3 nothing dirty, 4 root moved, 5 nothing dirty,
6 one leaf re-parented (structural change, same entity count), 7 nothing dirty, 8 that leaf deleted (entity count changes), 9 nothing dirty."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as oracle  # noqa: E402  (measurement script: the oracle generates the scene)
from tests.test_gpu_flecs_system import CPU, GPU, write_scene  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rng = np.random.default_rng(1)
lo = oracle.synth_level_offsets(n, levels)
par = oracle.synth_parents(lo)
local = oracle.compose_batch(oracle.synth_trs(n, int(lo[1])))
bounds = oracle.synth_aabb(n)
planes = oracle.main_camera_planes().reshape(1, 24)
m = oracle.compose([1, 2, 3], [0, 0, 0, 1], [1, 1, 1])
edits = [(int(e), 0, m) for e in np.unique(rng.integers(0, n, n // 100))]
leaf = n - 1  # a last-level entity: frame 6 moves it under another parent (same entity count), frame 8 deletes it
new_parent = int(lo[-3]) if levels > 1 else 0
frames = [[], [], edits, [], [(-1, 1, oracle.compose([1, 2, 3], [0, 0.7071068, 0, 0.7071068], [2, 2, 2]))], []]
if levels > 2:
    from tests.test_gpu_flecs_system import reparent_arg
    frames += [[(leaf, 3, reparent_arg(new_parent))], [], [(leaf, 2, np.zeros(16, np.float32))], []]
with tempfile.TemporaryDirectory() as d:
    scene = os.path.join(d, "scene.bin")
    write_scene(scene, lo, par, local["m"], oracle.identity68()["m"][0], planes, bounds, frames)
    only = os.environ.get("FLECS_BENCH_ONLY")  # a substring of the labels to run
    for label, binary, env in (("CPU systems (reference rule, real flecs)", CPU, {}),
                               ("GPU system, whole columns", GPU, {"YFH_PLAIN": "1"}),
                               ("GPU system, incremental, every table validated every frame (round 1)", GPU, {"YFH_VALIDATE": "1"}),
                               ("GPU system, incremental, 1 thread copies rows back", GPU, {"YFH_THREADS": "1"}),
                               ("GPU system, incremental", GPU, {}),
                               ("GPU system, incremental, every structural change rebuilt (setPatching(0))", GPU, {"YFH_NO_PATCH": "1"}),
                               ("GPU system, incremental, cameras and mesh bounds read from the ECS", GPU, {"YFH_ECS_INPUTS": "1"}),
                               ("GPU system, incremental, results stay on the device (last frame reads all back)", GPU, {"YFH_NO_READBACK": "1"})):
        if only and only not in label:
            continue
        e = dict(os.environ, YFH_TIMES="1", **env)
        for k in ("YFH_PLAIN", "YFH_VALIDATE", "YFH_ALWAYS_WALK", "YFH_NO_READBACK", "YFH_THREADS", "YFH_NO_PATCH", "YFH_ECS_INPUTS"):
            if k not in env:
                e.pop(k, None)
        r = subprocess.run([binary, scene, os.path.join(d, "out.bin")], env=e, capture_output=True, text=True, timeout=1200)
        times = [line.split(": ")[1].split(" ms")[0] for line in r.stderr.splitlines() if line.startswith("frame ")]
        print(f"{label:90s} n={n} L={levels}  ms per frame: " + "  ".join(times), flush=True)
        for line in r.stderr.splitlines():
            if line.startswith("stages ") and (env in ({}, {"YFH_ECS_INPUTS": "1"}) or os.environ.get("FLECS_BENCH_ALL_STAGES")) and line.split()[1] in ("0:", "2:", "4:", "6:", "8:"):
                print("    " + line, flush=True)
            elif os.environ.get("FLECS_BENCH_ALL_STAGES") and (line.startswith("  ") or "walk_tables" in line):
                print("    " + line[:400], flush=True)
