"""Pins the oracle's ECS semantics against the REAL flecs 4.0.4 vendored by the reference.

oracle/_ref/flecs_harness (built by `make -C oracle ref` from /root/reference/vendor/flecs/flecs.c where it
lies) registers the 68-byte components with the reference's hooks and TransformSyncSystem's three query terms
(src = Up), creates the hierarchy parent-first under a scene root, and runs the systems frame by frame.  The
flat-array oracle (yo_frame_aos68) must reproduce its world AND local columns bit for bit, dirty bytes included,
through local edits, world-only edits (inverse branch) and a re-rooting.  Skips where the reference (and hence
the binary) is absent, e.g. on the GPU box."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "oracle", "_ref", "flecs_harness")

pytestmark = pytest.mark.skipif(not os.path.exists(HARNESS), reason="oracle/_ref/flecs_harness not built (no /root/reference)")


def run_harness(tmp_path, lo, parent, local_m, root_m, frames):
    """frames: list of edit lists [(entity or -1 for root, kind 0 local / 1 world, m16)]."""
    n = parent.shape[0]
    scene = tmp_path / "scene.bin"
    out = tmp_path / "out.bin"
    with open(scene, "wb") as f:
        f.write(b"YFH1")
        f.write(struct.pack("<III", n, len(lo) - 1, len(frames)))
        f.write(np.ascontiguousarray(lo, np.uint32).tobytes())
        f.write(np.ascontiguousarray(parent, np.uint32).tobytes())
        f.write(np.ascontiguousarray(local_m, np.float32).tobytes())
        f.write(np.ascontiguousarray(root_m, np.float32).tobytes())
        for edits in frames:
            f.write(struct.pack("<I", len(edits)))
            for e, kind, m in edits:
                f.write(struct.pack("<II", 0xFFFFFFFF if e < 0 else e, kind))
                f.write(np.ascontiguousarray(m, np.float32).tobytes())
    subprocess.check_call([HARNESS, str(scene), str(out)], timeout=600)
    raw = open(out, "rb").read()
    from oracle import pyoracle as o

    res, pos = [], 0
    for _ in frames:
        w = np.frombuffer(raw, o.T68, n, pos).copy()
        pos += 68 * n
        l = np.frombuffer(raw, o.T68, n, pos).copy()
        pos += 68 * n
        batches, rows, shared = struct.unpack_from("<QQI", raw, pos)
        pos += 20
        res.append((w, l, batches, rows, shared))
    return res


def bits(a):
    return np.ascontiguousarray(a).view(np.uint8)


@pytest.mark.parametrize("n,levels", [(10_000, 1), (4096, 8), (30_000, 5)])
def test_oracle_equals_flecs_driven_systems(oracle, tmp_path, n, levels):
    rng = np.random.default_rng(n)
    lo = oracle.synth_level_offsets(n, levels)
    par = oracle.synth_parents(lo)
    trs = oracle.synth_trs(n, int(lo[1]))
    local = oracle.compose_batch(trs)
    world = np.zeros(n, oracle.T68)
    world["m"][:, [0, 5, 10, 15]] = 1
    world["dirty"] = 1
    root = oracle.identity68()

    def rnd_m(scale=5.0):
        return oracle.compose(rng.uniform(-scale, scale, 3), rng.uniform(-1, 1, 4), rng.uniform(0.5, 2, 3))

    frames = [[], []]  # frame 0: everything dirty; frame 1: nothing dirty
    frames.append([(int(e), 0, rnd_m()) for e in np.unique(rng.integers(0, n, 200))])           # local edits
    frames.append([(int(e), 1, rnd_m(50.0)) for e in np.unique(rng.integers(0, n, 150))])       # world-only edits
    frames.append([(int(rng.integers(0, int(lo[1]))), 0, np.zeros(16, np.float32))])            # a singular matrix
    frames.append([(-1, 1, oracle.compose([1, 2, 3], [0, 0.7071068, 0, 0.7071068], [2, 2, 2]))])  # root moved
    got = run_harness(tmp_path, lo, par, local["m"], root["m"][0], frames)

    for fi, (edits, (gw, gl, batches, rows, shared)) in enumerate(zip(frames, got)):
        for e, kind, m in edits:  # the same edits on the oracle's columns: set + modified -> dirty
            col = root if e < 0 else (local if kind == 0 else world)
            i = 0 if e < 0 else e
            col["m"][i] = m
            col["dirty"][i] = 1
        exp_batches = oracle.frame_aos68(world, local, par, lo, root)
        assert shared == 1, "flecs handed the Up term out as a per-row column"
        with np.errstate(all="ignore"):
            assert np.array_equal(bits(gw["m"]), bits(world["m"])), f"frame {fi}: world matrices differ from flecs"
            assert np.array_equal(bits(gl["m"]), bits(local["m"])), f"frame {fi}: local matrices differ from flecs"
        assert np.array_equal(gw["dirty"], world["dirty"]) and np.array_equal(gl["dirty"], local["dirty"])
        assert rows == n  # every entity is matched (all live under the root, which owns a WorldTransform)
        # one flecs table per (ChildOf, parent): the oracle's batch grouping is the same partition of the rows
        assert batches == exp_batches, f"frame {fi}: {batches} flecs batches vs {exp_batches} oracle batches"


def test_structural_change_observers_of_the_gpu_system_inside_real_flecs(tmp_path):
    """GpuTransformCullSystem's clean frames trust its observers: every operation that lets rows enter, leave or move
    inside a table with one of the transform components must raise its structure flag, nothing else may (CPU only: the
    system is registered in the reference's flecs, never run)."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    flecs_o = os.path.join(root, "oracle", "_ref", "flecs.o")
    flecs_h = "/root/reference/vendor/flecs"
    lib = os.path.join(root, "yume_amd", "libyume_hip.so")
    if not (os.path.exists(flecs_o) and os.path.exists(os.path.join(flecs_h, "flecs.h")) and os.path.exists(lib)):
        pytest.skip("needs oracle/_ref/flecs.o (make -C oracle ref) and the reference's flecs.h")
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-std=gnu11", "-O1", "-Wall", "-DFLECS_CONFIG_HEADER=1", "-I", flecs_h, "-I", os.path.join(root, "yume_amd", "host"),
                           "-I", os.path.join(root, "include"), "-o", str(exe), os.path.join(root, "tests", "flecs_structure_probe.c"),
                           os.path.join(root, "yume_amd", "host", "GpuTransformCullSystem.c"), flecs_o, "-L", os.path.join(root, "yume_amd"),
                           "-lyume_hip", f"-Wl,-rpath,{os.path.join(root, 'yume_amd')}", "-lm", "-lpthread"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    rows = [line.split() for line in r.stdout.splitlines()]
    assert len(rows) == 16, r.stdout + r.stderr
    for name, want, got in rows:
        assert want == got, f"{name}: structure flag {got}, expected {want}"
    assert r.returncode == 0


def test_host_data_structures_of_the_gpu_system(tmp_path):
    """The partitioned map filled by helper threads (1, 4, 16), its reuse, small maps, the radix sort of tables by
    parent slot against qsort, and the workers that outlive a frame (every job of 9000 batches exactly once; once more
    under ThreadSanitizer): tests/host_units_probe.c includes GpuTransformCullSystem.c and calls the static functions
    as they are (CPU only)."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    flecs_o = os.path.join(root, "oracle", "_ref", "flecs.o")
    flecs_h = "/root/reference/vendor/flecs"
    lib = os.path.join(root, "yume_amd", "libyume_hip.so")
    if not (os.path.exists(flecs_o) and os.path.exists(os.path.join(flecs_h, "flecs.h")) and os.path.exists(lib)):
        pytest.skip("needs oracle/_ref/flecs.o (make -C oracle ref) and the reference's flecs.h")
    exe = tmp_path / "units"
    subprocess.check_call(["gcc", "-std=gnu11", "-O1", "-Wall", "-Wno-unused-function", "-DFLECS_CONFIG_HEADER=1", "-I", flecs_h,
                           "-I", os.path.join(root, "yume_amd", "host"), "-I", os.path.join(root, "include"), "-o", str(exe),
                           os.path.join(root, "tests", "host_units_probe.c"), flecs_o, "-L", os.path.join(root, "yume_amd"),
                           "-lyume_hip", f"-Wl,-rpath,{os.path.join(root, 'yume_amd')}", "-lm", "-lpthread"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    rows = [line.split() for line in r.stdout.splitlines()]
    assert len(rows) == 7, r.stdout + r.stderr
    for name, verdict in rows:
        assert verdict == "ok", f"{name}: {verdict}"
    assert r.returncode == 0
    # the worker pool once more under ThreadSanitizer (host code only: nothing here touches a device)
    tsan = tmp_path / "units_tsan"
    built = subprocess.run(["gcc", "-std=gnu11", "-O1", "-g", "-fsanitize=thread", "-Wno-unused-function", "-DFLECS_CONFIG_HEADER=1",
                            "-DPROBE_POOL_ONLY=1", "-I", flecs_h, "-I", os.path.join(root, "yume_amd", "host"),
                            "-I", os.path.join(root, "include"), "-o", str(tsan), os.path.join(root, "tests", "host_units_probe.c"),
                            flecs_o, "-L", os.path.join(root, "yume_amd"), "-lyume_hip",
                            f"-Wl,-rpath,{os.path.join(root, 'yume_amd')}", "-lm", "-lpthread"], capture_output=True, text=True)
    if built.returncode != 0:
        pytest.skip("no ThreadSanitizer runtime for gcc here: " + built.stderr[-300:])
    r = subprocess.run([str(tsan)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "worker_pool ok" in r.stdout and "ThreadSanitizer" not in r.stderr, r.stdout + r.stderr[-3000:]
