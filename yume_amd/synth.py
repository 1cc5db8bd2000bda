"""Synthetic scenes of the benchmark configs — product-side measurement harness.

Hierarchies are built on the host (numpy): level sizes proportional to 2^k, every entity picks a uniform
random parent in the previous level, siblings sorted by parent (level-major, depth-sorted).  TRS and
bounds are filled on the device by ``Scene.fill_synthetic`` from the same counter-based hash
(splitmix64 keyed by (seed, stream, global index)), so nothing of size N ever crosses PCIe.
Cameras come from the library's scalar helpers (same arithmetic as engine/math3d.zig).
"""
import numpy as np

from . import api

SEED = 0x59554D45
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def synth_raw(seed, stream, index):
    """64 random bits for (seed, stream, index); index may be a uint64 array."""
    with np.errstate(over="ignore"):
        key = _mix64(np.uint64(seed) ^ (np.uint64(0x9E3779B97F4A7C15) * np.uint64(stream + 1)))
        return _mix64(key + np.asarray(index, np.uint64))


def level_offsets(n, n_levels):
    """Level sizes proportional to 2^k (k = 0..n_levels-1), summing to n."""
    if n < (1 << n_levels) - 1:
        raise ValueError("n too small for this depth")
    denom = (1 << n_levels) - 1
    out = np.zeros(n_levels + 1, np.uint32)
    acc = 0
    for k in range(n_levels):
        nk = (n << k) // denom if k < n_levels - 1 else n - acc
        acc += nk
        out[k + 1] = acc
    return out


def parents(offsets, seed=SEED, index_base=0):
    """Uniform random parent in the previous level, siblings sorted by parent (stream 14)."""
    offsets = np.asarray(offsets, np.uint32)
    n = int(offsets[-1])
    out = np.empty(n, np.uint32)
    out[: offsets[1]] = 0xFFFFFFFF
    for k in range(1, len(offsets) - 1):
        lo, hi = int(offsets[k]), int(offsets[k + 1])
        plo, n_prev = int(offsets[k - 1]), int(offsets[k]) - int(offsets[k - 1])
        idx = np.arange(index_base + lo, index_base + hi, dtype=np.uint64)
        pick = (synth_raw(seed, 14, idx) >> np.uint64(32)) % np.uint64(n_prev)
        out[lo:hi] = np.sort(pick.astype(np.uint32)) + np.uint32(plo)
    return out


def main_camera_planes():
    """Default camera (camera.zig:28-32: fovy 75 deg, near 0.1, far 200), window 1600x800 -> aspect 2, with the
    Vulkan Y flip (camera.zig:269-273); view = lookAt(eye (0,40,150) -> origin)."""
    proj = api.perspective(float(np.float32(np.deg2rad(75.0))), 2.0, 0.1, 200.0)
    proj[5] = np.float32(proj[5] * np.float32(-1.0))
    view = api.look_at([0, 40, 150], [0, 0, 0], [0, 1, 0])
    return api.extract_planes(api.mat4_mul(proj, view))


def cascade_planes():
    """4 orthographic shadow cascades (extents 12.5/25/50/100, depth 0..400) along light dir (-0.4,-1,-0.3)."""
    d = np.array([-0.4, -1.0, -0.3], np.float32)
    eye = (-d * np.float32(200.0)).astype(np.float32)
    view = api.look_at(eye, [0, 0, 0], [0, 1, 0])
    out = []
    for e in (12.5, 25.0, 50.0, 100.0):
        out.append(api.extract_planes(api.mat4_mul(api.orthographic(-e, e, -e, e, 0.0, 400.0), view)))
    return np.stack(out)


def distinct_parents(parent, offsets):
    """Number of distinct parents referenced by each level > 0 (for the algorithmic-bytes figure)."""
    total = 0
    for k in range(1, len(offsets) - 1):
        p = parent[int(offsets[k]) : int(offsets[k + 1])]
        if p.size:
            total += int(np.count_nonzero(np.diff(p)) + 1)
    return total
