"""Multi-GPU host logic: subtree-closed sharding (C ABI: yh_partition_subtrees / yh_shard_extract) and the
all-gather of per-GPU visible counts and compacted draw lists.

One process per GPU.  Entities are independent except for the parent->child chain, which is closed inside a
level-0 subtree, so every subtree goes to exactly one GPU and the data path needs no collective; the only
exchange is the all-gather at the end of a frame.  Two transports for that all-gather:
  * RCCL called by the library on its own stream (Scene.allgather_draw_lists) — the product path, and
  * torch.distributed (any backend: "nccl" = RCCL on ROCm, "gloo" on CPU) over host/torch buffers — used by
    the CPU tests of the N > 1 logic and as bench.py's stand-by if the library's communicator cannot be built.
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


def allgather_draw_lists_torch(local_lists, device="cpu"):
    """All-gather of variable-length draw lists with torch.distributed (counts first, then lists padded to the
    longest).  local_lists: one uint32 array per frustum (global ids).  Returns (all_counts [ranks, F], lists:
    per frustum the concatenation over ranks in rank order)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    F = len(local_lists)
    counts = torch.tensor([len(x) for x in local_lists], dtype=torch.int64, device=device)
    all_counts = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts)
    all_counts = torch.stack(all_counts).cpu().numpy()
    out = []
    for f in range(F):
        m = int(all_counts[:, f].max())
        if m == 0:
            out.append(np.empty(0, np.uint32))
            continue
        pad = torch.zeros(m, dtype=torch.int64, device=device)
        pad[: len(local_lists[f])] = torch.from_numpy(local_lists[f].astype(np.int64)).to(device)
        parts = [torch.zeros_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad)
        out.append(np.concatenate([parts[r][: int(all_counts[r, f])].cpu().numpy().astype(np.uint32) for r in range(world)]))
    return all_counts, out
