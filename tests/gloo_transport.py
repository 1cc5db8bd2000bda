"""Test infrastructure: the N > 1 exchange between gloo ranks on the CPU (tests/test_sharding.py).  NOT product code:
the product's only transport is RCCL inside libyume_hip.so.  Two forms, mirroring the library's two entry points:

  allgather_lists  - yh_allgather_draw_lists: counts first, then the variable-length index lists;
  allgather_words  - yh_allgather_draw_lists_async (the one bench.py times at N > 1): what travels is the frame's
                     VISIBILITY WORDS, one bit per entity and frustum in 64-bit words, rows zero-padded to the longest
                     shard's word count (a fixed size: ONE all-gather, no counts); every rank then expands all ranks'
                     words itself, rank-major, ascending inside a rank, rank r's id_base added - the host-side
                     statement of gather_setup / k_gather_count / k_gather_expand (yume_amd/csrc).
"""
import numpy as np


def allgather_lists(local_lists, device="cpu"):
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


def pack_words(vis):
    """(F, n) 0/1 bytes -> (F, ceil(n / 64)) uint64 visibility words: bit (e & 63) of word (e >> 6), as the frame
    kernels publish them (one 64-bit ballot per wave-sized chunk)."""
    vis = np.asarray(vis, np.uint8)
    F, n = vis.shape
    words = (n + 63) // 64
    padded = np.zeros((F, words * 64), np.uint8)
    padded[:, :n] = vis
    return np.packbits(padded.reshape(F, words, 64), axis=2, bitorder="little").view("<u8").reshape(F, words)


def expand_words(words, id_base):
    """One rank's (F, W) words -> per frustum the ascending global ids id_base + local."""
    out = []
    for row in words:
        bits = np.unpackbits(row.view(np.uint8).reshape(-1, 8), axis=1, bitorder="little").reshape(-1)
        out.append((np.flatnonzero(bits) + id_base).astype(np.uint32))
    return out


def allgather_words(local_words, n_local, id_base, device="cpu"):
    """local_words: (F, ceil(n_local / 64)) uint64.  Returns (all_counts [ranks, F], lists per frustum: rank-major)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    F = local_words.shape[0]
    # first call on a scene: the ranks exchange word counts and id bases once (gather_setup)
    mine = torch.tensor([local_words.shape[1], id_base, F, n_local], dtype=torch.int64, device=device)
    peers = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(peers, mine)
    peers = torch.stack(peers).cpu().numpy()
    assert (peers[:, 2] == F).all(), "every rank must have created its scene with the same n_frusta"
    g_words = int(peers[:, 0].max())
    send = np.zeros((F, g_words), np.uint64)  # send_pad: rows of shorter shards are zero-padded
    send[:, : local_words.shape[1]] = local_words
    t = torch.from_numpy(send.view(np.int64)).to(device)
    parts = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(parts, t)  # the ONE collective of the frame
    counts = np.zeros((world, F), np.int64)
    lists = [[] for _ in range(F)]
    for r in range(world):
        per = expand_words(parts[r].cpu().numpy().view(np.uint64), int(peers[r, 1]))
        for f in range(F):
            counts[r, f] = per[f].size
            lists[f].append(per[f])
    return counts, [np.concatenate(x) if x else np.empty(0, np.uint32) for x in lists]
