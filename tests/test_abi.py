"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol include/yume_hip.h
declares, and its scalar host helpers agree with the oracle.  No compute call is made without a GPU."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "yume_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(yh_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ["yh_init", "yh_shutdown", "yh_scene_create", "yh_upload_trs", "yh_upload_local_aos68",
                 "yh_upload_hierarchy", "yh_upload_aabb", "yh_frame", "yh_get_counts", "yh_get_draw_list",
                 "yh_download_world_aos68", "yh_last_error", "yh_timings", "yh_comm_init", "yh_allgather_draw_lists"]:
        assert must in names


def test_library_exports_every_declared_symbol():
    from yume_amd import _lib

    lib = C.CDLL(_lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in include/yume_hip.h but not exported"


def test_binding_table_matches_header():
    from yume_amd import _lib

    assert sorted(_lib.SIGNATURES) == declared_functions()
    _lib.lib()  # sets argtypes for all of them
    assert _lib.lib().yh_abi_version() == 1


def test_header_compiles_as_plain_c(tmp_path):
    import subprocess

    src = tmp_path / "t.c"
    src.write_text('#include "yume_hip.h"\nint main(void){ yh_scene_desc d; (void)d; return sizeof(yh_timing) == 24 ? 0 : 1; }\n')
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           str(src), "-o", str(exe)])
    subprocess.check_call([str(exe)])


def test_struct_layouts_match_ctypes():
    from yume_amd import _lib

    assert C.sizeof(_lib.SceneDesc) == 40
    assert C.sizeof(_lib.Timing) == 24
    assert C.sizeof(_lib.SynthDesc) == 16
    assert C.sizeof(_lib.DeviceInfo) == 4 * 4 + 8 + 32 + 128


def test_host_helpers_match_oracle(oracle):
    from yume_amd import api, synth

    rng = np.random.default_rng(0)
    for _ in range(20):
        a = rng.standard_normal(16).astype(np.float32)
        b = rng.standard_normal(16).astype(np.float32)
        assert np.array_equal(api.mat4_mul(a, b).view(np.uint32), oracle.mul(a, b).view(np.uint32))
        eye, at = rng.uniform(-50, 50, 3), rng.uniform(-5, 5, 3)
        assert np.array_equal(api.look_at(eye, at, [0, 1, 0]), oracle.look_at(eye, at, [0, 1, 0]))
        vp = api.mat4_mul(api.perspective(1.0, 1.5, 0.1, 300.0), api.look_at(eye, at, [0, 1, 0]))
        assert np.array_equal(api.extract_planes(vp), oracle.extract_planes(vp))
    assert np.array_equal(api.perspective(1.3, 2.0, 0.1, 200.0), oracle.perspective(1.3, 2.0, 0.1, 200.0))
    assert np.array_equal(api.orthographic(-10, 10, -10, 10, 0, 200), oracle.orthographic(-10, 10, -10, 10, 0, 200))
    assert np.array_equal(synth.main_camera_planes(), oracle.main_camera_planes())
    assert np.array_equal(synth.cascade_planes(), oracle.cascade_planes())
    # Camera.view_projection (camera.zig:253-302): library helper == oracle restatement, both camera kinds
    for ortho in (0, 1):
        for _ in range(10):
            pos, rot = rng.uniform(-100, 100, 3), rng.uniform(-180, 180, 3) if not ortho else rng.uniform(-3, 3, 3)
            a = api.camera_view_projection(ortho, pos, rot, 1.7, 1.2, 0.1, 500.0)
            assert np.array_equal(a, oracle.camera_view_projection(ortho, pos, rot, 1.7, 1.2, 0.1, 500.0))


def test_camera_pipeline_helpers_match_oracle_and_known_answers(oracle):
    """GameWindow.zig:160-161 per camera: decompose -> Quat.toEuler (math3d.zig:418-442) -> updateMatrices
    (camera.zig:161-166).  The library's host helpers against the oracle's twins, bit for bit, plus the answers the
    reference's formulas fix without any trigonometry table: identity, the |sinp| >= 1 branch, the degrees conversion."""
    from yume_amd import api

    rng = np.random.default_rng(5)
    for i in range(400):
        m = oracle.compose(rng.uniform(-50, 50, 3), rng.uniform(-1, 1, 4), rng.uniform(0.3, 3, 3))
        if i % 9 == 0:
            m[rng.integers(0, 16)] = 0.0
        if i % 31 == 0:
            m[:] = 0.0  # InvalidMatrix -> Decomposed.IDENTITY (components/transform.zig:68-70)
        t, q, s, ok = api.mat4_decompose(m)
        ot, oq, os_, ook = oracle.decompose(m)
        assert ok == ook
        if ok:
            assert np.array_equal(t, ot) and np.array_equal(q, oq, equal_nan=True) and np.array_equal(s, os_, equal_nan=True)
        else:
            assert np.array_equal(np.concatenate([t, q, s]), np.float32([0, 0, 0, 0, 0, 0, 1, 1, 1, 1]))
        assert np.array_equal(api.quat_to_euler(q), oracle.quat_to_euler(q), equal_nan=True)
        for ortho in (0, 1):
            a = api.camera_update_from_world(ortho, m, 1.6, 1.1, 0.2, 300.0)
            b = oracle.camera_update_from_world(ortho, m, 1.6, 1.1, 0.2, 300.0)
            assert a[3] == b[3] == ok
            for x, y in zip(a[:3], b[:3]):
                assert np.array_equal(x, y, equal_nan=True)
    assert np.array_equal(oracle.quat_to_euler([0, 0, 0, 1]), np.zeros(3, np.float32))
    # sinp = 2 (w y - z x) = 1 exactly: the copysign(half_pi) branch, converted with mulf(180).divf(pi)
    h = np.float32(np.sqrt(np.float32(0.5)))
    e = oracle.quat_to_euler([0, h, 0, h])
    if np.float32(2) * (h * h) >= 1:
        assert e[1] == np.float32(np.float32(np.pi / 2) * np.float32(180)) / np.float32(np.pi)
    assert oracle.quat_to_euler([0, -1, 0, 1])[1] < 0 and oracle.quat_to_euler([0, 1, 0, 1])[1] > 0  # |sinp| = 2: clamped, signed
    assert np.allclose(oracle.quat_to_euler([np.sin(0.25), 0, 0, np.cos(0.25)]), [np.degrees(0.5), 0, 0], atol=1e-4)
    assert np.allclose(oracle.quat_to_euler([0, 0, np.sin(-0.4), np.cos(-0.4)]), [0, 0, np.degrees(-0.8)], atol=1e-4)
    # a camera whose world matrix is a pure translation: view = T(-pos), projection Y-flipped
    w = oracle.compose([3, -4, 5], [0, 0, 0, 1], [1, 1, 1])
    view, proj, vp, ok = oracle.camera_update_from_world(0, w, 2.0, 1.3, 0.1, 200.0)
    assert ok and np.array_equal(view, oracle.compose([-3, 4, -5], [0, 0, 0, 1], [1, 1, 1]))
    p = oracle.perspective(1.3, 2.0, 0.1, 200.0).copy()
    p[5] *= -1
    assert np.array_equal(proj, p) and np.array_equal(vp, oracle.mul(p, view))


def test_camera_view_projection_properties(oracle):
    """Identities the reference's camera update implies (the trig values themselves are libm's: parity unpinned)."""
    I = oracle.identity68()["m"][0]
    assert np.array_equal(oracle.rotation([0, 0, 0], 1.0), I)            # zero axis -> IDENTITY (math3d.zig:693-694)
    assert np.array_equal(oracle.rotation([0, 2, 0], 0.7), oracle.rotation([0, 1, 0], 0.7))  # axis is normalised
    r = oracle.rotation([0, 0, 1], np.float32(np.pi / 2)).reshape(4, 4)  # columns
    assert np.allclose(r[0][:3], [0, 1, 0], atol=1e-6) and np.allclose(r[1][:3], [-1, 0, 0], atol=1e-6)
    # no rotation, camera at the origin: view = IDENTITY, so view_projection = projection with the Y flip
    vp = oracle.camera_view_projection(0, [0, 0, 0], [0, 0, 0], 2.0, 1.3, 0.1, 200.0)
    p = oracle.perspective(1.3, 2.0, 0.1, 200.0).copy()
    p[5] *= -1
    assert np.array_equal(vp, oracle.mul(p, I))
    # a point straight ahead (-Z) of a camera moved to +10 z is inside all six planes; one behind it is not
    planes = oracle.extract_planes(oracle.camera_view_projection(0, [0, 0, 10], [0, 0, 0])).reshape(6, 4)
    inside = lambda q: bool(np.all(planes[:, :3] @ np.asarray(q, np.float32) + planes[:, 3] >= 0))
    assert inside([0, 0, 0]) and not inside([0, 0, 20]) and not inside([0, 0, -500])
    # orthographic: the engine's fixed +-10 box, depth 0..200, no Y flip
    vo = oracle.camera_view_projection(1, [0, 0, 0], [0, 0, 0])
    assert np.array_equal(vo, oracle.mul(oracle.orthographic(-10, 10, -10, 10, 0, 200), I))


def test_harness_hierarchy_matches_oracle(oracle):
    from yume_amd import synth

    for n, L in [(255, 8), (4096, 8), (100_000, 5), (1_000_000, 8)]:
        lo = synth.level_offsets(n, L)
        assert np.array_equal(lo, oracle.synth_level_offsets(n, L))
        for base in (0, 7 * n):
            assert np.array_equal(synth.parents(lo, index_base=base), oracle.synth_parents(lo, index_base=base))


def test_product_fails_loudly_without_gpu_or_library():
    """No CPU fallback: without a GPU yh_init reports a HIP error instead of computing anything."""
    import torch

    from yume_amd import Context, _lib

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.YumeHipError) as e:
        Context(0)
    assert e.value.code == _lib.YH_ERR_HIP


def test_one_rccl_per_process_whichever_loads_first():
    """RCCL is dlopen'ed by SONAME (yh_comm_describe says which file that was).  After `import torch` the library must
    get torch's bundled copy - ONE RCCL in the process, the case of bench.py at N > 1 - and in the other order (the
    library first, torch afterwards: two copies) the process must still exit cleanly."""
    import sys

    code_a = "import torch; from yume_amd import Context; print(Context.comm_describe())"
    a = subprocess.run([sys.executable, "-c", code_a], capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert a.returncode == 0, a.stderr[-500:]
    import torch

    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    if os.path.exists(os.path.join(torch_lib, "librccl.so")):
        assert torch_lib in a.stdout, f"a second RCCL beside torch's: {a.stdout}"
        v = torch.cuda.nccl.version()
        assert f"rccl {v[0]}.{v[1]}.{v[2]} " in a.stdout, (a.stdout, v)
    code_b = "from yume_amd import Context; print(Context.comm_describe()); import torch; print(torch.cuda.nccl.version())"
    b = subprocess.run([sys.executable, "-c", code_b], capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert b.returncode == 0 and b.stdout.startswith("rccl "), (b.returncode, b.stdout, b.stderr[-500:])


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "yume_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".c")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in txt and "yume_oracle" not in txt and "import oracle" not in txt, f


def test_bench_parses_and_names_the_largest_single_gpu_configuration_by_default():
    """The driver-timed line must be the largest single-GPU configuration (cfg5, 256M instances: far outside the
    Infinity Cache), with the other GPU configurations as sub-records."""
    import importlib.util
    import sys

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        args = bench.parse()
    finally:
        sys.argv = argv
    assert args.workload == "cfg5" and args.gpus == 1 and not args.no_sub
    assert bench.WORKLOADS["cfg5"]["n"] == max(w["n"] for w in bench.WORKLOADS.values())
    assert [w for w in bench.SUB_ORDER if w != args.workload] == ["cfg2", "cfg3", "cfg4"]
    # the transform launch is charged its own bytes only: the 4 B per emitted index belong to the expansion
    lo = np.array([0, 1000], np.uint32)
    own, lists = bench.algorithmic_bytes(bench.WORKLOADS["cfg3"], lo, None, 0, 1000, 5, 1234.0)
    assert own == 1000 * (40 + 64 + 24) + 5 * (1000 / 8 + 1000 / 64) and lists == 4 * 1234.0


def test_zig_binding_is_generated_from_the_header_and_complete():
    """zig/engine/hip/yume_hip.zig binds EVERY symbol of include/yume_hip.h (one `pub extern fn` each, same number of
    parameters) and is exactly what tools/gen_zig_ffi.py makes of the header today."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("gen_zig_ffi", os.path.join(ROOT, "tools", "gen_zig_ffi.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    zig = open(os.path.join(ROOT, "zig", "engine", "hip", "yume_hip.zig")).read()
    assert zig == gen.generate(), "run `python tools/gen_zig_ffi.py`: the committed Zig binding is not what the header generates"
    bound = dict(re.findall(r"pub extern fn (yh_[a-z0-9_]+)\((.*?)\) [^;]+;", zig))
    assert sorted(bound) == declared_functions()
    for ret, name, args in gen.prototypes():
        n_zig = 0 if not bound[name].strip() else len(re.findall(r"[A-Za-z_][A-Za-z0-9_]*: ", bound[name]))
        assert n_zig == len(args), f"{name}: {len(args)} C parameters, {n_zig} in Zig"


def test_zig_system_binds_every_function_of_the_c_system_header():
    """zig/engine/systems/GpuTransformCullSystem.zig declares every GpuTransformCullSystem_* function of the C header
    it wraps (test hooks aside), and keeps the reference's registerTo signature (TransformSyncSystem.zig:12)."""
    h = open(os.path.join(ROOT, "yume_amd", "host", "GpuTransformCullSystem.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    declared = set(re.findall(r"\b(GpuTransformCullSystem_[A-Za-z]+)\s*\(", h))
    z = open(os.path.join(ROOT, "zig", "engine", "systems", "GpuTransformCullSystem.zig")).read()
    bound = set(re.findall(r"extern fn (GpuTransformCullSystem_[A-Za-z]+)\(", z))
    hooks = {"GpuTransformCullSystem_structureDirty", "GpuTransformCullSystem_clearStructureDirty"}
    assert declared - hooks == bound, (sorted(declared - hooks - bound), sorted(bound - declared))
    assert "pub fn registerTo(world: ecs.World) ecs.Entity {" in z


def test_integration_patch_applies_to_the_reference_tree(tmp_path):
    """zig/yume.patch (build.zig, engine/ecs.zig `systems` export, the editor's call site) applies cleanly to the
    reference as it is here.  Skips where the reference is absent (GPU box)."""
    import shutil

    ref = "/root/reference"
    if not os.path.exists(os.path.join(ref, "build.zig")):
        pytest.skip("no reference tree")
    for rel in ("build.zig", "engine/ecs.zig", "editor/Editor.zig"):
        os.makedirs(os.path.dirname(tmp_path / rel), exist_ok=True)
        shutil.copy(os.path.join(ref, rel), tmp_path / rel)
    subprocess.check_call(["patch", "-p1", "-s", "-i", os.path.join(ROOT, "zig", "yume.patch")], cwd=tmp_path)
    assert "GpuTransformCullSystem.c" in open(tmp_path / "build.zig").read()
    assert "GpuTransformCullSystem = @import" in open(tmp_path / "engine/ecs.zig").read()
    ed = open(tmp_path / "editor/Editor.zig").read()
    assert "GpuTransformCullSystem.registerTo(ctx.world)" in ed and "PostTransformSystem.registerTo" not in ed


def test_python_constants_equal_the_headers_defines():
    """Flags, error codes and enums that yume_amd/_lib.py spells out again must be the header's values."""
    import re

    from yume_amd import _lib

    text = open(os.path.join(ROOT, "include", "yume_hip.h")).read()
    defines = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"^#define\s+(YH_[A-Z0-9_]+)\s+\(?(-?(?:0x[0-9a-fA-F]+|\d+))u?\)?\s*(?:/\*.*)?$", text, re.M)}
    enums = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"\b(YH_[A-Z0-9_]+)\s*=\s*(-?(?:0x[0-9a-fA-F]+|\d+))", text)}
    known = dict(defines, **enums)
    checked = 0
    for name in dir(_lib):
        if name.startswith("YH_") and isinstance(getattr(_lib, name), int) and name in known:
            assert getattr(_lib, name) == known[name], name
            checked += 1
    assert checked >= 12 and _lib.YH_FRAME_LISTS_ON_DEMAND == defines["YH_FRAME_LISTS_ON_DEMAND"]
