"""CPU tests of the oracle itself (no GPU): algebraic known answers recoverable from the reference
(the reference holds no tests or vectors), internal consistency, and the committed golden
fixtures."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
I16 = np.eye(4, dtype=np.float32).reshape(16)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_layout_is_68_bytes(oracle):
    # engine/components/transform.zig:8-12 — extern struct { matrix: Mat4, dirty: bool }: 68 B, align 4
    assert oracle.T68.itemsize == 68
    assert oracle.T68.fields["dirty"][1] == 64


def test_compose_identity(oracle):
    # compose(0, IDENTITY, 1) == IDENTITY (math3d.zig:816-822 with Quat.IDENTITY)
    assert np.array_equal(oracle.compose([0, 0, 0], [0, 0, 0, 1], [1, 1, 1]), I16)


def test_mul_identity(oracle):
    rng = np.random.default_rng(1)
    for _ in range(50):
        m = rng.standard_normal(16).astype(np.float32)
        assert np.array_equal(oracle.mul(m, I16), m)
        assert np.array_equal(oracle.mul(I16, m), m)


def test_mul_matches_float64_rounding_order(oracle):
    # element (c,r) = ((a.i[r]*b.c.x + a.j[r]*b.c.y) + a.k[r]*b.c.z) + a.t[r]*b.c.w in f32, unfused
    rng = np.random.default_rng(2)
    a = rng.standard_normal(16).astype(np.float32)
    b = rng.standard_normal(16).astype(np.float32)
    got = oracle.mul(a, b)
    exp = np.empty(16, np.float32)
    f = np.float32
    for c in range(4):
        for r in range(4):
            acc = f(a[0 + r] * b[c * 4 + 0])
            acc = f(acc + f(a[4 + r] * b[c * 4 + 1]))
            acc = f(acc + f(a[8 + r] * b[c * 4 + 2]))
            acc = f(acc + f(a[12 + r] * b[c * 4 + 3]))
            exp[c * 4 + r] = acc
    assert np.array_equal(bits(got), bits(exp))


def test_from_quat_formula(oracle):
    # math3d.zig:727-749: column i = (1-2(yy+zz), 2(xy+wz), 2(xz-wy), 0) ...
    f = np.float32
    x, y, z, w = (f(0.3), f(-0.5), f(0.7), f(0.2))
    m = oracle.from_quat([x, y, z, w])
    assert m[0] == f(f(1.0) - f(f(2.0) * f(f(y * y) + f(z * z))))
    assert m[1] == f(f(2.0) * f(f(x * y) + f(w * z)))
    assert m[2] == f(f(2.0) * f(f(x * z) - f(w * y)))
    assert m[6] == f(f(2.0) * f(f(y * z) + f(w * x)))
    assert m[3] == 0 and m[7] == 0 and m[11] == 0 and m[15] == 1
    assert np.array_equal(m[12:15], np.zeros(3, np.float32))


def test_compose_closed_form_equals_two_products(oracle):
    # compose == [R.i*s.x | R.j*s.y | R.k*s.z | (t,1)] value-for-value for finite operands
    rng = np.random.default_rng(3)
    for _ in range(200):
        t = rng.uniform(-100, 100, 3).astype(np.float32)
        q = rng.uniform(-1, 1, 4).astype(np.float32)
        s = rng.uniform(0.1, 3, 3).astype(np.float32)
        m = oracle.compose(t, q, s)
        r = oracle.from_quat(q)
        exp = r.copy()
        exp[0:3] = r[0:3] * s[0]
        exp[4:7] = r[4:7] * s[1]
        exp[8:11] = r[8:11] * s[2]
        exp[12:15] = t
        assert np.array_equal(m, exp)


def test_inverse(oracle):
    rng = np.random.default_rng(4)
    for _ in range(50):
        m = oracle.compose(rng.uniform(-10, 10, 3), rng.uniform(-1, 1, 4), rng.uniform(0.5, 2, 3))
        inv = oracle.inverse(m)
        assert inv is not None
        assert np.abs(oracle.mul(inv, m) - I16).max() < 1e-4
    assert oracle.inverse(np.zeros(16, np.float32)) is None  # error.DeterminantZero (math3d.zig:789)


def test_perspective_constants(oracle):
    # math3d.zig:649-659, defaults camera.zig:28-32
    f = np.float32
    fovy, aspect, near, far = f(np.deg2rad(75.0)), f(2.0), f(0.1), f(200.0)
    m = oracle.perspective(fovy, aspect, near, far)
    ff = f(1.0) / np.tan(f(fovy / f(2.0)), dtype=np.float32)
    assert abs(m[5] - ff) <= 2e-7 * abs(ff)
    assert m[0] == f(m[5] / aspect)
    assert m[10] == f(far / f(near - far)) and m[11] == -1.0
    assert m[14] == f(-f(far * near) / f(far - near)) and m[15] == 0.0


def test_orthographic_and_lookat(oracle):
    m = oracle.orthographic(-10, 10, -10, 10, 0, 200)  # camera.zig:296
    assert m[0] == np.float32(0.1) and m[5] == np.float32(-0.1) and m[10] == np.float32(1 / -200) and m[15] == 1
    v = oracle.look_at([0, 0, 5], [0, 0, 0], [0, 1, 0])
    exp = I16.copy()
    exp[14] = -5
    assert np.array_equal(v, exp)


def test_system_three_branches(oracle):
    """TransformSyncSystem.zig:60-73 on hand-built rows."""
    T = oracle.T68
    parent = oracle.identity68(dirty=0)
    parent["m"][0] = oracle.compose([1, 2, 3], [0, 0, 0, 1], [2, 2, 2])
    local = np.zeros(3, T)
    world = np.zeros(3, T)
    for i in range(3):
        local["m"][i] = oracle.compose([i, 0, 0], [0, 0, 0, 1], [1, 1, 1])
        world["m"][i] = I16
    # row 0: nothing dirty -> skipped; row 1: local dirty -> forward; row 2: world dirty only -> inverse
    local["dirty"] = [0, 1, 0]
    world["dirty"] = [0, 0, 1]
    world["m"][2] = oracle.compose([5, 5, 5], [0, 0, 0, 1], [1, 1, 1])
    w0 = world.copy()
    l0 = local.copy()
    oracle.transform_sync_batch(world, local, parent)
    assert np.array_equal(world["m"][0], w0["m"][0]) and world["dirty"][0] == 0
    assert np.array_equal(world["m"][1], oracle.mul(parent["m"][0], l0["m"][1])) and world["dirty"][1] == 1
    assert np.array_equal(world["m"][2], w0["m"][2]) and world["dirty"][2] == 1
    assert np.array_equal(local["m"][2], oracle.mul(oracle.inverse(parent["m"][0]), w0["m"][2]))
    # singular parent -> IDENTITY substitutes the inverse (TransformSyncSystem.zig:70)
    parent["m"][0] = 0
    local["dirty"] = 0
    world["dirty"] = 1
    oracle.transform_sync_batch(world, local, parent)
    assert np.array_equal(local["m"][2], world["m"][2])


def test_dirty_propagates_down_and_clears(oracle):
    lo = oracle.synth_level_offsets(2000, 4)
    par = oracle.synth_parents(lo)
    trs = oracle.synth_trs(2000, int(lo[1]))
    local = oracle.compose_batch(trs)
    world = np.zeros(2000, oracle.T68)
    world["m"][:, [0, 5, 10, 15]] = 1
    world["dirty"] = 1
    root = oracle.identity68()
    oracle.frame_aos68(world, local, par, lo, root)
    assert not world["dirty"].any() and not local["dirty"].any() and root["dirty"][0] == 0
    w1 = world.copy()
    # nothing dirty -> nothing changes
    oracle.frame_aos68(world, local, par, lo, root)
    assert np.array_equal(world["m"], w1["m"])
    # dirty one level-1 entity: exactly its subtree changes
    e = int(lo[1]) + 3
    local["m"][e] = oracle.compose([9, 9, 9], [0, 0, 0, 1], [1, 1, 1])
    local["dirty"][e] = 1
    oracle.frame_aos68(world, local, par, lo, root)
    changed = np.flatnonzero((world["m"] != w1["m"]).any(axis=1))
    sub = {e}
    for k in range(2, 4):
        for c in range(int(lo[k]), int(lo[k + 1])):
            if int(par[c]) in sub:
                sub.add(c)
    assert set(changed.tolist()) <= sub and e in changed


def test_threads_do_not_change_results(oracle):
    lo = oracle.synth_level_offsets(50_000, 8)
    par = oracle.synth_parents(lo)
    trs = oracle.synth_trs(50_000, int(lo[1]))
    w1 = oracle.frame_trs(trs, par, lo, n_threads=1)
    w4 = oracle.frame_trs(trs, par, lo, n_threads=4)
    assert np.array_equal(bits(w1["m"]), bits(w4["m"]))


def test_planes_are_normalised_and_inside_point_is_visible(oracle):
    pl = oracle.main_camera_planes().reshape(6, 4)
    assert np.allclose(np.linalg.norm(pl[:, :3], axis=1), 1.0, atol=1e-6)
    # the look-at target (origin) is in front of the camera: inside all six planes
    assert (pl[:, 3] > 0).all()
    world = np.zeros(2, oracle.T68)
    world["m"][:, [0, 5, 10, 15]] = 1
    world["m"][1, 12:15] = [0, 0, 1000]  # far behind the camera
    b = np.tile(np.array([-1, -1, -1, 1, 1, 1], np.float32), (2, 1))
    vis, bnd = oracle.cull(world, b, pl.reshape(1, 24))
    assert vis[0, 0] == 1 and vis[0, 1] == 0 and not bnd.any()


def test_boundary_flag(oracle):
    # a unit box whose face lies exactly on a plane x >= 0: s + r == 0 -> visible and boundary
    planes = np.zeros((1, 24), np.float32)
    planes[0, 0::4] = 0  # placeholder
    pl = np.array([[1, 0, 0, 0], [-1, 0, 0, 100], [0, 1, 0, 100], [0, -1, 0, 100], [0, 0, 1, 100], [0, 0, -1, 100]], np.float32)
    world = np.zeros(3, oracle.T68)
    world["m"][:, [0, 5, 10, 15]] = 1
    world["m"][0, 12] = -1.0  # box [-2,0]: touches the plane
    world["m"][1, 12] = -1.1  # just outside
    world["m"][2, 12] = -0.999995  # within 1e-5
    b = np.tile(np.array([-1, -1, -1, 1, 1, 1], np.float32), (3, 1))
    vis, bnd = oracle.cull(world, b, pl.reshape(1, 24))
    assert vis[0].tolist() == [1, 0, 1] and bnd[0].tolist() == [1, 0, 1]


def test_synth_generator_properties(oracle):
    lo = oracle.synth_level_offsets(1_000_000, 8)
    assert lo[0] == 0 and lo[-1] == 1_000_000 and (np.diff(lo.astype(np.int64)) > 0).all()
    par = oracle.synth_parents(lo)
    for k in range(1, 8):
        p = par[lo[k] : lo[k + 1]]
        assert p.min() >= lo[k - 1] and p.max() < lo[k] and (np.diff(p.astype(np.int64)) >= 0).all()
    trs = oracle.synth_trs(10_000, 100)
    q = trs[3:7]
    assert np.allclose((q * q).sum(0), 1.0, atol=1e-5)
    assert np.abs(trs[0:3, :100]).max() > 5 and np.abs(trs[0:3, 100:]).max() <= 5
    # counter-based: any sub-range reproduces
    lib = oracle.lib()
    sub = np.empty((10, 10_000), np.float32)
    import ctypes as C

    lib.yo_synth_trs(C.c_uint64(oracle.SEED), C.c_uint64(0), C.c_uint64(100), C.c_size_t(5000), C.c_size_t(1000),
                     sub.ctypes.data_as(C.c_void_p), C.c_size_t(10_000))
    assert np.array_equal(sub[:, 5000:6000], trs[:, 5000:6000])


@pytest.mark.parametrize("name", ["cfg1_flat10k", "tree4096_l8", "frusta5_flat4096"])
def test_golden_fixtures(oracle, name):
    """Committed vectors (tests/golden/, made by tests/golden/make_golden.py from the oracle after it was
    checked against the real-flecs harness): the oracle must keep reproducing them bit for bit."""
    path = os.path.join(GOLDEN, name + ".npz")
    if not os.path.exists(path):
        pytest.skip("fixture not generated yet")
    from tests.golden.make_golden import run_case

    g = np.load(path)
    got = run_case(oracle, name)
    assert np.array_equal(bits(got["world"]), g["world"].view(np.uint32))
    for k in g.files:
        if k.startswith("vis") or k.startswith("bnd"):
            assert np.array_equal(got[k], g[k]), k


def test_decompose_inverts_compose_and_flags_invalid_matrices(oracle):
    """Mat4.decompose (math3d.zig:828-981): recovers T, R (up to sign), S of Mat4.compose for positive scales; negative
    determinants come back with all three scales negated; singular matrices and w = 0 are InvalidMatrix -> IDENTITY."""
    rng = np.random.default_rng(3)
    n = 2000
    t = rng.uniform(-100, 100, (3, n)).astype(np.float32)
    q = rng.uniform(-1, 1, (4, n)).astype(np.float32)
    q /= np.linalg.norm(q, axis=0, keepdims=True).astype(np.float32)
    s = rng.uniform(0.2, 3, (3, n)).astype(np.float32)
    col = oracle.compose_batch(np.concatenate([t, q, s]))
    d, valid = oracle.decompose_batch(col)
    assert valid.all()
    assert np.allclose(d[0:3], t, rtol=0, atol=0)  # the translation column is copied (divided by w = 1)
    assert np.allclose(d[7:10], s, rtol=2e-5)
    sign = np.sign((d[3:7] * q).sum(axis=0))
    assert np.allclose(d[3:7] * sign, q, atol=2e-5)
    # recompose(decompose(M)) == M up to rounding (math3d.zig:824-826)
    back = oracle.compose_batch(d)
    assert np.allclose(back["m"], col["m"], rtol=1e-4, atol=1e-4)
    # mirrored: one negative scale -> determinant < 0 -> every scale negated, rotation still proper
    m = col[:100].copy()
    m["m"][:, 0:3] *= -1
    dm, vm = oracle.decompose_batch(m)
    assert vm.all() and (dm[7:10] < 0).all()
    # invalid: zero scale on an axis (singular 3x3), w = 0
    bad = col[:4].copy()
    bad["m"][0, 4:7] = 0
    bad["m"][1, 15] = 0
    bad["m"][2, :] = 0
    db, vb = oracle.decompose_batch(bad)
    assert vb.tolist() == [0, 0, 0, 1]
    for i in range(3):
        assert db[:, i].tolist() == [0, 0, 0, 0, 0, 0, 1, 1, 1, 1]
    # a perspective partition is cleared, w != 1 divides everything first
    p = col[:1].copy()
    p["m"][0] *= 2.0
    p["m"][0, 3] = 0.25
    dp, vp = oracle.decompose_batch(p)
    assert vp[0] == 1 and np.allclose(dp[0:3, 0], t[:, 0], rtol=1e-6) and np.allclose(dp[7:10, 0], s[:, 0], rtol=2e-5)
