"""Multi-GPU host logic: subtree-closed sharding (C ABI: yh_partition_subtrees / yh_shard_extract) and the
all-gather of per-GPU visible counts and compacted draw lists.

One process per GPU.  Entities are independent except for the parent->child chain, which is closed inside a
level-0 subtree, so every subtree goes to exactly one GPU and the data path needs no collective; the only
exchange is the all-gather at the end of a frame, and its one transport is RCCL called by the library
on its collective stream (Scene.allgather_draw_lists / allgather_draw_lists_async).  There is no stand-by transport in
this package: a communicator that cannot be brought up is an error (the CPU tests of the N > 1 logic carry their own
gloo transport, tests/gloo_transport.py).
"""
import ctypes as C

import numpy as np

from . import _lib as L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def partition_subtrees(parent, level_offsets, n_parts):
    """-> (part_of_entity uint32[n], part_sizes uint64[n_parts]); greedy largest-first over level-0 subtrees."""
    parent = np.ascontiguousarray(parent, np.uint32)
    lo = np.ascontiguousarray(level_offsets, np.uint32)
    part_of = np.empty(parent.shape[0], np.uint32)
    sizes = np.zeros(n_parts, np.uint64)
    L.check(L.lib().yh_partition_subtrees(_p(parent), _p(lo), len(lo) - 1, int(n_parts), _p(part_of), _p(sizes)))
    return part_of, sizes


def shard_extract(parent, level_offsets, part_of, part, part_size):
    """-> (local_parent, local_level_offsets, global_of_local) of one shard: its own depth-sorted arrays."""
    parent = np.ascontiguousarray(parent, np.uint32)
    lo = np.ascontiguousarray(level_offsets, np.uint32)
    part_of = np.ascontiguousarray(part_of, np.uint32)
    lp = np.empty(int(part_size), np.uint32)
    llo = np.zeros(len(lo), np.uint32)
    g = np.empty(int(part_size), np.uint32)
    L.check(L.lib().yh_shard_extract(_p(parent), _p(lo), len(lo) - 1, _p(part_of), int(part), _p(lp), _p(llo), _p(g)))
    return lp, llo, g
