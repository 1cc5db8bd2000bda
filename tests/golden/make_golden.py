"""Generates the committed golden fixtures from the CPU oracle.

The reference holds no vectors for this path, so these are produced here: inputs come from
the seeded counter-based generator (reproducible from (seed, n, levels) alone, so they are not stored),
expected outputs from the oracle AFTER its ECS semantics were checked bit-for-bit against the real
vendored flecs (oracle/flecs_harness.c, tests/test_flecs_harness.py).  Run:  python -m tests.golden.make_golden
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (n, n_levels, frusta)
    "cfg1_flat10k": (10_000, 1, "main"),        # BASELINE configs[0]
    "tree4096_l8": (4096, 8, "main"),           # small instance of configs[1]
    "frusta5_flat4096": (4096, 1, "main+casc"),  # small instance of configs[2]
}


def planes_for(oracle, kind):
    main = oracle.main_camera_planes().reshape(1, 24)
    if kind == "main":
        return main
    return np.concatenate([main, oracle.cascade_planes()], axis=0)


def scene_for(oracle, name):
    n, n_levels, kind = CASES[name]
    lo = oracle.synth_level_offsets(n, n_levels)
    par = oracle.synth_parents(lo)
    trs = oracle.synth_trs(n, int(lo[1]))
    bounds = oracle.synth_aabb(n)
    return dict(n=n, level_offsets=lo, parent=par, trs=trs, bounds=bounds, planes=planes_for(oracle, kind))


def run_case(oracle, name):
    s = scene_for(oracle, name)
    world = oracle.frame_trs(s["trs"], s["parent"], s["level_offsets"])
    vis, bnd = oracle.cull(world, s["bounds"], s["planes"])
    out = {"world": np.ascontiguousarray(world["m"])}
    for f in range(vis.shape[0]):
        out[f"vis{f}"] = np.flatnonzero(vis[f]).astype(np.uint32)
        out[f"bnd{f}"] = np.flatnonzero(bnd[f]).astype(np.uint32)
    return out


def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import pyoracle as oracle

    for name in CASES:
        out = run_case(oracle, name)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
