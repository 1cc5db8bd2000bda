"""What the gather's expansion costs a rank at N = 8: the words of eight shards (copies of this scene's) expanded on one
GPU, as every rank does per frame (yh_gather_selftest).  Run under rocprofv3 --kernel-trace --stats for the kernel times:
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_gather -- python3 tools/gather_probe.py [entities] [levels] [ranks]"""
import sys

sys.path.insert(0, ".")
from yume_amd import Context, Scene, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ranks = int(sys.argv[3]) if len(sys.argv) > 3 else 8
ctx = Context(0)
lo = synth.level_offsets(n, levels)
par = synth.parents(lo)
planes = synth.main_camera_planes().reshape(1, 24)
with Scene(ctx, n, n_levels=levels, n_frusta=1) as s:
    s.upload_hierarchy(par, lo)
    s.fill_synthetic()
    s.frame(planes)
    own = int(s.counts()[0][0])
    for _ in range(5):
        s.gather_selftest(ranks)
    g = s.gathered_counts(1, ranks)
    print(f"n={n} levels={levels} ranks={ranks}: own list {own} entries, gathered {int(g[:, 0].sum())} entries from {ranks} ranks")
