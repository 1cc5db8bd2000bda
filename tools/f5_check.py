"""Is a mismatch between the GPU system and the CPU systems in the structural fuzz the reference's own table-order effect
(a child's table visited before its parent's reads last frame's parent matrix)?  For every mismatching row of
one frame: which parent row reproduces the GPU's value, and whether the CPU's value is that parent's matrix of THIS frame or
of the frame BEFORE times the local matrix.  usage (after a pytest run with --basetemp=<dir> that kept scene and outputs):
python tools/f5_check.py <dir> [frame]     (measurement aid: reads the harness outputs only)"""
import glob, os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
from oracle import pyoracle as oracle
from tests.test_gpu_flecs_system import read_out
base = sys.argv[1]
d = glob.glob(base + "/test_random_structural*")[0]
n, n_frames = 50000, 151
cpu = read_out(os.path.join(d, "cpu.bin"), n, n_frames, 0, oracle.T68)
gpu = read_out(os.path.join(d, "gpu.bin"), n, n_frames, 2, oracle.T68, gpu=True)
f = int(sys.argv[2]) if len(sys.argv) > 2 else 133
cw, cl = cpu[f][0], cpu[f][1]
gw = gpu[f][0]
bad = np.flatnonzero(~np.all(gw["m"] == cw["m"], axis=1))
print("mismatching rows at frame", f, bad)
def mul(a, b):  # column-major 4x4 as stored: world = parent * local
    A = a.reshape(4, 4).T.astype(np.float64); B = b.reshape(4, 4).T.astype(np.float64)
    return (A @ B).T.reshape(16)
for c in bad:
    lc = cl["m"][c]
    # which parent gives the GPU's row?
    Wg = gw["m"].reshape(-1, 4, 4).transpose(0, 2, 1).astype(np.float64)
    L = lc.reshape(4, 4).T.astype(np.float64)
    prod = (Wg @ L).transpose(0, 2, 1).reshape(-1, 16)
    err = np.abs(prod - gw["m"][c].astype(np.float64)).max(axis=1)
    p = int(np.argmin(err))
    print(f"row {c}: GPU row = world[{p}] (this frame) x local[{c}] within {err[p]:.2e}")
    for name, src in (("this frame", cpu[f][0]), ("the frame before", cpu[f - 1][0])):
        e = np.abs(mul(src["m"][p], lc) - cw["m"][c].astype(np.float64)).max()
        print(f"   CPU row vs CPU world[{p}] of {name} x local: max diff {e:.2e}")
    print(f"   parent {p}: CPU world changed in this frame: {not np.array_equal(cpu[f][0]['m'][p], cpu[f-1][0]['m'][p])}; GPU parent row == CPU parent row: {np.array_equal(gw['m'][p], cw['m'][p])}")
    later = [k for k in range(f + 1, n_frames) if np.array_equal(cpu[k][0]["m"][c], gpu[f][0]["m"][c])]
    print(f"   CPU catches up with the GPU's row in frames: {later[:5]}")
