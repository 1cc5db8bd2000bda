"""Steps of the drop-in's first frame (walk + rebuild) at 1M entities / 300k tables, from YUME_GPU_SYSTEM_TRACE=1: python tools/trace_rebuild.py"""
import os, subprocess, sys, tempfile
import numpy as np
sys.path.insert(0, "/root/repo"); os.chdir("/root/repo")
from oracle import pyoracle as oracle
from tests.test_gpu_flecs_system import GPU, write_scene
n = 1_000_000
lo = oracle.synth_level_offsets(n, 8); par = oracle.synth_parents(lo)
local = oracle.compose_batch(oracle.synth_trs(n, int(lo[1]))); bounds = oracle.synth_aabb(n)
planes = oracle.main_camera_planes().reshape(1, 24)
frames = [[], []]
with tempfile.TemporaryDirectory() as d:
    scene = os.path.join(d, "scene.bin")
    write_scene(scene, lo, par, local["m"], oracle.identity68()["m"][0], planes, bounds, frames)
    e = dict({k: v for k, v in os.environ.items() if not k.startswith("YFH_")}, YFH_TIMES="1", YUME_GPU_SYSTEM_TRACE="1")
    r = subprocess.run([GPU, scene, os.path.join(d, "out.bin")], env=e, capture_output=True, text=True, timeout=600)
    print(r.stderr[-3000:])
