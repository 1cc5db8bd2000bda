"""Host cost of the multi-GPU set-up at full size: yh_partition_subtrees + yh_shard_extract (single host thread, C ABI)
on the 100M-entity 8-level tree of BASELINE configs[3], cut into 8 subtree-closed shards.  No GPU needed.
    python tools/partition_bench.py [entities] [parts]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yume_amd import shard, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 8
t = time.perf_counter()
lo = synth.level_offsets(n, 8)
par = synth.parents(lo)
print(f"hierarchy of {n} entities generated in {time.perf_counter() - t:.2f} s (numpy, not part of the path)", flush=True)
t = time.perf_counter()
part_of, sizes = shard.partition_subtrees(par, lo, parts)
t_part = time.perf_counter() - t
print(f"yh_partition_subtrees: {t_part:.3f} s for {n} entities -> {parts} parts ({n / t_part / 1e6:.1f} M entities/s), "
      f"sizes {sizes.min()}..{sizes.max()} (imbalance {(sizes.max() - n / parts) / (n / parts) * 100:.4f} %)", flush=True)
t = time.perf_counter()
lp, llo, g = shard.shard_extract(par, lo, part_of, 0, sizes[0])
t_ext = time.perf_counter() - t
print(f"yh_shard_extract (one shard of {sizes[0]}): {t_ext:.3f} s", flush=True)
child = np.arange(int(llo[1]), int(llo[-1]))
assert np.array_equal(part_of[g[child]], part_of[par[g[child]]])
print(f"level-0 subtrees: {int(lo[1])}, cores used: 1, host: {os.cpu_count()} CPUs")
