"""Edited-row frames of the drop-in inside the reference's flecs, 1M entities: what a frame costs with k rows set since the
last one (k = 1k, 10k, 100k) and with the root moved (every WorldTransform row comes back).
usage: python tools/ab_workers.py [harness binary ...]   (default: oracle/_ref/flecs_gpu_harness)"""
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
rng = np.random.default_rng(3)
lo = oracle.synth_level_offsets(n, levels)
par = oracle.synth_parents(lo)
local = oracle.compose_batch(oracle.synth_trs(n, int(lo[1])))
bounds = oracle.synth_aabb(n)
planes = oracle.main_camera_planes().reshape(1, 24)
m = oracle.compose([1, 2, 3], [0, 0, 0, 1], [1, 1, 1])
kinds = [("1k rows set", 1000)] * 4 + [("10k rows set", 10_000)] * 4 + [("100k rows set", 100_000)] * 3
frames = [[], []]
labels = ["first", "idle"]
for name, k in kinds:
    frames.append([(int(e), 0, m) for e in rng.choice(n, k, replace=False)])
    labels.append(name)
for k in range(4):
    frames.append([(-1, 1, oracle.compose([1 + k, 2, 3], [0, 0, 0, 1], [1, 1, 1]))])
    labels.append("root moved")
with tempfile.TemporaryDirectory() as d:
    scene = os.path.join(d, "scene.bin")
    write_scene(scene, lo, par, local["m"], oracle.identity68()["m"][0], planes, bounds, frames)
    for binary in (sys.argv[1:] or [GPU]):
        for rep in range(2):
            e = dict({k: v for k, v in os.environ.items() if not k.startswith("YFH_")}, YFH_TIMES="1")
            if os.environ.get("AB_GAP_MS"):  # frames that far apart, as an engine's are (idle workers, cooler caches)
                e["YFH_FRAME_GAP_MS"] = os.environ["AB_GAP_MS"]
            if os.environ.get("AB_ECS_INPUTS"):  # cameras and mesh bounds read from the ECS
                e["YFH_ECS_INPUTS"] = "1"
            r = subprocess.run([binary, scene, os.path.join(d, "out.bin")], env=e, capture_output=True, text=True, timeout=1200)
            ms = [float(line.split(": ")[1].split(" ms")[0]) for line in r.stderr.splitlines() if line.startswith("frame ")]
            by = {}
            for lab, x in zip(labels, ms):
                by.setdefault(lab, []).append(x)
            if os.environ.get("AB_STAGES") and rep == 1:
                print("\n".join(f"  {labels[int(line.split()[1][:-1])]}: {line}" if line.startswith("stages ") else line
                                for line in r.stderr.splitlines() if line.startswith("stages ") or line.startswith("  ")))
            print(os.path.basename(binary), "run", rep, "; ".join(f"{lab}: " + " ".join(f"{x:.2f}" for x in v) for lab, v in by.items() if lab not in ("first", "idle")), flush=True)
