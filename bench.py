#!/usr/bin/env python3
"""bench.py — Mentities/s transformed + culled per frame, % of the HBM roofline, CPU systems beside it.

A "step" is one frame of the hot path (compose -> level-ordered propagate -> cull + compact, count read-back)
over one synthetic scene that is already resident in HBM.  One process per GPU:

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

The headline (`value`, `ms_per_step`, `roofline`) is the largest single-GPU configuration of BASELINE.json, cfg5:
256M flat instances with instanced AABBs — 29 GB of arrays, far outside the 256 MB Infinity Cache, ~4.7 ms per frame,
so K = 20 steps are ~95 ms of GPU time.  With N > 1 its instances are split over the GPUs (strong scaling) and each
frame ends with the all-gather of the draw lists.  The other GPU configurations run right after it as sub-records
of the same JSON line (`workloads`: cfg2 = 1M entities / 8 levels, cfg3 = 16M flat x 5 frusta, cfg4 = the 12.5M-entity
shard one GPU holds of the 100M-entity scene), each timed over at least 50 ms of frames, each with its own roofline
object.  `--workload cfgN` makes another configuration the headline; `--no-sub` drops the sub-records.
"""
import argparse
import json
import math
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
INFINITY_CACHE_BYTES = 256 << 20

WORKLOADS = {
    # name: (entities per GPU or total, levels, frusta, bounds_kind, scaling)
    "cfg2": dict(n=1_000_000, levels=8, frusta="main", bounds=1, scaling="weak",
                 desc="1M entities, random 8-level hierarchy, single camera frustum (BASELINE configs[1])"),
    "cfg3": dict(n=16 * 1024 * 1024, levels=1, frusta="main+casc", bounds=1, scaling="weak",
                 desc="16M flat entities, 4 shadow cascades + main camera (BASELINE configs[2])"),
    "cfg4": dict(n=12_500_000, levels=8, frusta="main", bounds=1, scaling="weak",
                 desc="12.5M-entity 8-level subtree-closed shard per GPU = 100M on 8 GPUs (BASELINE configs[3])"),
    "cfg5": dict(n=256 * 1000 * 1000, levels=1, frusta="main", bounds=2, scaling="strong",
                 desc="256M flat instances, 16 instanced AABBs, split over the GPUs (BASELINE configs[4])"),
}
SUB_ORDER = ("cfg2", "cfg3", "cfg4", "cfg5")
SUB_MIN_TIMED_S = 0.060   # every sub-record times at least this much GPU work
SUB_BUDGET_S = 240.0      # no further sub-record is started after this many seconds of sub-records
WATCHDOG_S = 420.0        # hard limit on the sub-records: the headline line is printed without the missing ones

# Frames enqueued before the host waits for the device once (inside the timed region: it is part of the cost).  Without
# a bound the HIP runtime's command and kernel-argument pools run dry after ~1000 un-synchronised launches and the
# stream stalls in long phases (measured: 800 frames of cfg2 with no sync take 70 us per frame instead of 34).
FRAMES_IN_FLIGHT = 256


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg5", choices=sorted(WORKLOADS), help="the headline configuration")
    ap.add_argument("--no-sub", action="store_true", help="headline only: no sub-records of the other configurations")
    ap.add_argument("--sub", default="", help="comma-separated sub-records to run (default: every other configuration)")
    ap.add_argument("--entities", type=int, default=0, help="override the headline's entities per GPU")
    ap.add_argument("--grouping", type=int, default=0, help="levels per launch (0 = automatic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-comm", action="store_true",
                    help="rehearsal on one GPU: bring up a one-rank RCCL communicator and run the multi-GPU step "
                         "(frame + pipelined all-gather) - measures what the exchange adds to a frame")
    ap.add_argument("--emulate-ranks", type=int, default=0,
                    help="rehearsal on one GPU: partition each hierarchy as for this many GPUs and run shard 0")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not hand HIP events to the launches")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def algorithmic_bytes(wl, lo, parent, first, end, n_frusta, list_entries):
    """Compulsory HBM bytes of a fully fused pass over entities [first, end).
    Returns (own, lists):
      own   = what the TRANSFORM launch itself has to move: reads TRS 40 B + parent index 4 B (levels > 0) + AABB 24 B
              (or 4 B mesh id) + 64 B per DISTINCT parent per level; writes world 64 B + one visibility bit per entity
              and frustum and one count byte per 64 (the launch's stage-3 output);
      lists = 4 B per emitted draw index — written by k_expand (or by the next frame's fused first launch), so they
              are charged to the frame and to the expansion, never to the transform launch."""
    n = end - first
    b = n * (40 + 64) + n * (24 if wl["bounds"] == 1 else 4)
    n_child = end - max(first, int(lo[1]))
    if n_child > 0:
        b += 4 * n_child
        for k in range(1, len(lo) - 1):  # distinct parents of the levels that intersect [first, end)
            a, z = max(first, int(lo[k])), min(end, int(lo[k + 1]))
            if z > a:
                p = parent[a:z]
                b += 64 * int(np.count_nonzero(np.diff(p)) + 1)
    b += n_frusta * (n / 8.0 + n / 64.0)
    return float(b), 4.0 * float(list_entries)


CPU_SAMPLE = {  # entities of the CPU leg's sample per workload: the workload's own shape, bounded
    "cfg2": 1_000_000,          # the whole workload: 1M entities, 8 levels
    "cfg3": 1_000_000,          # 1M of the 16M flat entities, all 5 frusta
    "cfg4": 12_500_000 // 16,   # a 1/16 sample of one GPU's 12.5M-entity 8-level shard: EXTRAPOLATED
    "cfg5": 1_000_000,          # 1M of the 256M instances, 16 instanced AABBs
}


def cpu_baseline(name, wl, n, lo, parent, planes, seconds):
    """The reference CPU systems (oracle port: AoS 68-byte columns, one batch per table = per run of equal parent,
    forward branch on every row + both dirty sweeps, then the cull), single thread as the reference runs them and on
    all host cores, on a bounded sample of THIS workload's shape.  Only this leg touches oracle/."""
    from oracle import pyoracle as oracle

    n_s = min(n, CPU_SAMPLE.get(name, 1_000_000))
    if n_s != n:
        lo_s = oracle.synth_level_offsets(n_s, len(lo) - 1)
        par_s = oracle.synth_parents(lo_s)
    else:
        lo_s, par_s = lo, parent
    trs = oracle.synth_trs(n_s, int(lo_s[1]))
    if wl["bounds"] == 1:
        bounds, mesh = oracle.synth_aabb(n_s), None
    else:
        bounds, mesh = oracle.synth_mesh_table(16), oracle.synth_mesh_ids(n_s, 16)
    local = oracle.compose_batch(trs)
    world = np.zeros(n_s, oracle.T68)
    world["m"][:, [0, 5, 10, 15]] = 1
    root = oracle.identity68()

    def frame(threads):
        local["dirty"] = 1  # the benchmark marks every LocalTransform dirty (untimed)
        t0 = time.perf_counter()
        oracle.frame_aos68(world, local, par_s, lo_s, root, threads)
        oracle.cull(world, bounds, planes, mesh_id=mesh, n_threads=threads)
        return time.perf_counter() - t0

    out = {}
    for key, threads, budget in (("cpu_baseline", 1, seconds), ("cpu_baseline_mt", min(16, os.cpu_count() or 1), seconds / 3)):
        for _ in range(2):
            frame(threads)
        times = []
        t_end = time.perf_counter() + budget
        while time.perf_counter() < t_end or len(times) < 3:
            times.append(frame(threads))
        med = float(np.median(times))
        out[key] = {
            "value": round(n_s / med / 1e6, 3),
            "unit": "Mentities/s",
            "cores": threads,
            "kind": "port",
            "ns_per_entity": round(med / n_s * 1e9, 2),
            "sample": f"{len(times)} frames of a {n_s}-entity {len(lo_s) - 1}-level scene, {planes.shape[0]} frusta "
                      f"(median; gcc -O2 -ffp-contract=off; AoS-68 propagate + 2 dirty sweeps + cull)"
                      + ("" if n_s == n else f"; a {n_s}/{n} sample of the workload's shape: the rate is EXTRAPOLATED to the full size"),
        }
    return out


def flecs_legs(lo_levels):
    """The same rule driven by the REAL flecs 4.0.4 the reference vendors, twice on one bounded sample scene: the reference's
    three CPU systems (oracle/_ref/flecs_harness: compiled from the reference's flecs sources in place; the per-row
    arithmetic is the oracle's restatement, Zig cannot be built here; transform systems only, the reference has no cull)
    and the drop-in (oracle/_ref/flecs_gpu_harness: yume_amd/host/GpuTransformCullSystem.c registered in the same flecs,
    one frustum culled as well).  Frame 0: everything dirty (creation), frame 1: nothing moved, frames 2-4: the root is
    re-set, so every WorldTransform row is recomputed (and, in the drop-in, crosses PCIe back into flecs' columns),
    frame 5: nothing moved.  The binaries are build artefacts of the container that holds the reference: where they did
    not travel, the records say so instead of silently vanishing.  Returns (cpu_baseline_in_flecs, dropin_in_flecs)."""
    import struct
    import subprocess
    import tempfile

    from oracle import pyoracle as oracle

    cpu_exe = os.path.join(ROOT, "oracle", "_ref", "flecs_harness")
    gpu_exe = os.path.join(ROOT, "oracle", "_ref", "flecs_gpu_harness")
    missing = ("is not on this box (it is built from /root/reference/vendor/flecs by `make -C oracle ref` where the "
               "reference exists)")
    n_s = 200_000
    lo = oracle.synth_level_offsets(n_s, lo_levels)
    par = oracle.synth_parents(lo)
    local = oracle.compose_batch(oracle.synth_trs(n_s, int(lo[1])))
    bounds = oracle.synth_aabb(n_s)
    planes = oracle.main_camera_planes().reshape(1, 24)
    root = oracle.identity68()["m"][0]
    edit = np.asarray(oracle.compose([1, 2, 3], [0, 0, 0, 1], [1, 1, 1]), np.float32)
    frames = [0, 0, 1, 1, 1, 0]  # edits per frame: the root's WorldTransform is set in frames 2-4

    def run(exe):
        if not os.path.exists(exe):
            return None, f"{os.path.relpath(exe, ROOT)} {missing}"
        try:
            r = subprocess.run([exe, scene, os.path.join(d, "out.bin")], capture_output=True, text=True, timeout=300,
                               env=dict({k: v for k, v in os.environ.items() if not k.startswith("YFH_")}, YFH_TIMES="1"))
        except Exception as ex:  # noqa: BLE001
            return None, f"{os.path.basename(exe)} failed: {ex}"
        ms = [float(line.split(": ")[1].split(" ms")[0]) for line in r.stderr.splitlines() if line.startswith("frame ")]
        if len(ms) != len(frames):
            return None, f"{os.path.basename(exe)} printed {len(ms)} frame times (rc {r.returncode})"
        return ms, ""

    with tempfile.TemporaryDirectory() as d:
        scene = os.path.join(d, "scene.bin")
        with open(scene, "wb") as f:
            f.write(b"YFH2")
            f.write(struct.pack("<III", n_s, lo_levels, len(frames)))
            f.write(lo.astype(np.uint32).tobytes())
            f.write(par.astype(np.uint32).tobytes())
            f.write(np.ascontiguousarray(local["m"], np.float32).tobytes())
            f.write(np.ascontiguousarray(root, np.float32).tobytes())
            f.write(struct.pack("<I", planes.shape[0]))
            f.write(np.ascontiguousarray(planes, np.float32).tobytes())
            f.write(np.ascontiguousarray(bounds, np.float32).tobytes())
            for k in frames:
                f.write(struct.pack("<I", k))
                if k:
                    f.write(struct.pack("<II", 0xFFFFFFFF, 1))
                    f.write(edit.tobytes())
        cpu_ms, cpu_err = run(cpu_exe)
        gpu_ms, gpu_err = run(gpu_exe)
    what = f"a {n_s}-entity {lo_levels}-level scene inside the reference's vendored flecs 4.0.4"
    if cpu_ms is None:
        cpu = {"value": None, "kind": "reference-flecs", "sample": cpu_err}
    else:
        med = float(np.median(cpu_ms[2:5]))
        cpu = {"value": round(n_s / med / 1e3, 3), "unit": "Mentities/s", "cores": 1, "kind": "port",
               "ns_per_entity": round(med * 1e6 / n_s, 2), "ms_per_frame": [round(x, 3) for x in cpu_ms],
               "sample": f"frames 2-4 (the root moved: every row recomputed) of {what} (TransformSyncSystem rule + both "
                         f"PostTransform sweeps, no cull; median)"}
    if gpu_ms is None:
        dropin = {"value": None, "sample": gpu_err}
    else:
        med = float(np.median(gpu_ms[2:5]))
        dropin = {"value": round(n_s / med / 1e3, 3), "unit": "Mentities/s", "ms_per_frame": [round(x, 3) for x in gpu_ms],
                  "first_frame_ms": round(gpu_ms[0], 3), "nothing_moved_ms": round(min(gpu_ms[1], gpu_ms[5]), 3),
                  "every_row_recomputed_ms": round(med, 3),
                  "speedup_over_cpu_systems": {"every_row_recomputed": round(float(np.median(cpu_ms[2:5])) / med, 1),
                                               "nothing_moved": round(min(cpu_ms[1], cpu_ms[5]) / min(gpu_ms[1], gpu_ms[5]), 1),
                                               "first_frame": round(cpu_ms[0] / gpu_ms[0], 2)} if cpu_ms else None,
                  "sample": f"GpuTransformCullSystem registered in the same flecs on the same scene, one frustum culled as well, "
                            f"host wall time of the frame's systems (PCIe and the rows written back into flecs' columns "
                            f"included): {what}"}
    return cpu, dropin


def profiled_traffic(workload, symbol):
    """HBM bytes per launch of `symbol` from the committed rocprofv3 PMC passes of this bench command (profiles/r*/,
    newest round first): 2 x FETCH_SIZE + WRITE_SIZE, KiB units, gfx950 FETCH_SIZE correction as calibrated in
    profiles/r01/SUMMARY.md.  PMC passes are separate rocprofv3 runs (tools/profile.sh): the number cannot be taken
    inside a timing run."""
    import csv
    import glob

    for d in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*")), reverse=True):
        vals = {}
        for kind in ("fetch", "write"):
            f = os.path.join(d, f"{workload}_pmc_{kind}_mean_kib_per_dispatch.csv")
            if os.path.exists(f):
                for r in csv.DictReader(open(f)):
                    if r["kernel"] == symbol:
                        vals[kind] = float(r["mean_KiB_per_dispatch"])
        if len(vals) == 2:
            return (2 * vals["fetch"] + vals["write"]) * 1024.0, os.path.relpath(d, ROOT)
    return None, None


class stdout_to_stderr:
    """RCCL prints a version banner on stdout when a communicator comes up; the bench's stdout carries ONE JSON line,
    so communicator set-up runs with file descriptor 1 pointed at stderr."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


class CommFailure(RuntimeError):
    """The library's RCCL communicator could not be brought up (or failed its end-to-end check) on some rank."""


class Job:
    """What every workload of this process shares: the context, the ranks, the transport."""

    def __init__(self, args):
        self.args = args
        self.rank = int(os.environ.get("RANK", "0"))
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.torch = None
        self.ctx = None
        self.info = None
        self.comm_up = False
        self.rccl = ""  # which RCCL the library resolved (yh_comm_describe), next to torch's

    def barrier(self):
        self.ctx.wait()
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if self.dist is None:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device="cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_ok(self, ok):
        if self.dist is None:
            return bool(ok)
        flag = self.torch.tensor([1 if ok else 0], dtype=self.torch.int32, device="cuda")
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN)
        return bool(flag.item())


def run_workload(job, name, steps, warmup, headline, entities=0):
    """One configuration: build the scene in HBM, find its dominant launch, time `steps` frames (or, steps = 0, as
    many as SUB_MIN_TIMED_S needs).  Returns the record; for the headline also what the CPU leg needs."""
    from yume_amd import Context, Scene, _lib, synth

    args, rank, world_size = job.args, job.rank, job.world_size
    wl = WORKLOADS[name]
    n = entities or wl["n"]
    if wl["scaling"] == "strong":
        n = n // world_size
    total = n * world_size
    planes = synth.main_camera_planes().reshape(1, 24)
    if wl["frusta"] == "main+casc":
        planes = np.concatenate([planes, synth.cascade_planes()])
    F = planes.shape[0]

    ctx = job.ctx
    ranks = max(world_size, args.emulate_ranks)
    sharding = None
    if ranks > 1 and wl["levels"] > 1:
        # A hierarchy shards by subtree-closed ranges: ONE tree of n x ranks entities, cut by the
        # library's partitioner into one part per GPU, this rank's part extracted into its own depth-sorted arrays;
        # its entities draw their synthetic values from their GLOBAL indices, emitted ids are shard_base + local.
        from yume_amd import shard

        t0 = time.perf_counter()
        glo = synth.level_offsets(n * ranks, wl["levels"])
        gpar = synth.parents(glo)
        t1 = time.perf_counter()
        part_of, sizes = shard.partition_subtrees(gpar, glo, ranks)
        t2 = time.perf_counter()
        parent, lo, g_of_l = shard.shard_extract(gpar, glo, part_of, rank, sizes[rank])
        t3 = time.perf_counter()
        del gpar, part_of
        n = int(sizes[rank])
        total = int(sizes.sum()) if world_size > 1 else n  # (a one-GPU rehearsal processes shard 0 only)
        id_base = int(sizes[:rank].sum())
        sharding = {"ranks": ranks, "global_entities": int(sizes.sum()), "part_sizes": [int(x) for x in sizes],
                    "host_s": {"build_global_tree_numpy": round(t1 - t0, 3), "yh_partition_subtrees": round(t2 - t1, 3),
                               "yh_shard_extract": round(t3 - t2, 3)},
                    "note": "one-time set-up on the host, single thread, outside the timed region"
                            + ("" if world_size > 1 else f"; rehearsal on one GPU: shard 0 of {ranks}")}
    else:
        # flat scenes shard by contiguous equal ranges: rank r owns global entities [r n, (r + 1) n)
        lo = synth.level_offsets(n, wl["levels"])
        parent = synth.parents(lo, index_base=rank * n)
        g_of_l, id_base = None, rank * n
    scene = Scene(ctx, n, n_levels=wl["levels"], n_frusta=F, bounds_kind=wl["bounds"],
                  max_meshes=16 if wl["bounds"] == 2 else 0, id_base=id_base)
    scene.upload_hierarchy(parent, lo)
    if g_of_l is not None:
        scene.fill_synthetic(global_of_local=g_of_l)  # inputs generated straight into HBM, from global indices
    else:
        scene.fill_synthetic(index_base=rank * n)
    scene.set_grouping(args.grouping)
    flags = _lib.YH_FRAME_ALL_DIRTY | _lib.YH_FRAME_ASYNC

    collective = "none"
    gather = world_size > 1 or args.force_comm
    if gather and not os.environ.get("YUME_BENCH_OWN_LISTS"):  # the gathered lists hold this GPU's part: its own lists are
        flags |= _lib.YH_FRAME_LISTS_ON_DEMAND                   # expanded only if somebody reads them (A/B: tools/ab_lists_on_demand.sh)
    if world_size == 1 and args.force_comm:
        if not job.comm_up:
            with stdout_to_stderr():
                ctx.comm_init(1, 0, Context.comm_unique_id())
            job.comm_up = True
        collective = "rccl (library, collective stream), one-rank rehearsal"
    if world_size > 1:
        # The product's only transport: the library's own RCCL communicator.  If it cannot be brought up and checked end
        # to end on EVERY rank the job has no result: CommFailure ends it with "value": null (main), never a stand-by.
        dist = job.dist
        err = ""
        try:
            with stdout_to_stderr():
                if not job.comm_up:
                    ids = [Context.comm_unique_id() if rank == 0 else None]
                    dist.broadcast_object_list(ids, src=0)
                    ctx.comm_init(world_size, rank, ids[0])
                    job.comm_up = True
                scene.frame(planes, _lib.YH_FRAME_ALL_DIRTY)
                scene.allgather_draw_lists()        # exact-size exchange (host sync): checks the communicator end to end
                scene.allgather_draw_lists_async()  # the pipelined exchange the timed frames use
                scene.gathered_counts(F)
            ok = 1
        except Exception as ex:  # noqa: BLE001
            ok, err = 0, f"rank {rank}: {ex}"
        if not job.all_ok(ok):
            raise CommFailure(err or "the library's RCCL communicator failed on another rank")
        collective = ("rccl (library, collective stream): one ncclAllGather of the frame's VISIBILITY WORDS (n/8 bytes per "
                      "frustum and rank, not index lists: 20-32x fewer bytes over xGMI), every rank expands all ranks' words "
                      "into the gathered draw lists itself (its own part included: the frames pass YH_FRAME_LISTS_ON_DEMAND); "
                      f"runs beside the next frames; {job.rccl}")

    def step():
        scene.frame(planes, flags)
        if gather:
            scene.allgather_draw_lists_async()

    def timed(k):
        job.barrier()
        t0 = time.perf_counter()
        for i in range(k):
            step()
            if (i + 1) % FRAMES_IN_FLIGHT == 0:
                ctx.wait()
        job.barrier()
        return job.max_over_ranks(time.perf_counter() - t0)

    # calibration (untimed): HIP events around every launch to find the dominant launch of a frame
    groups = scene.groups()
    dominant = 0
    ms_all = np.zeros(64)
    if not args.no_kernel_events:
        scene.set_kernel_timing(1)
        for _ in range(8 if n > 64_000_000 else 20):
            step()
        ms_all, _ = scene.group_timings()
        ms_all = ms_all.astype(np.float64)
        dominant = int(np.argmax(ms_all[: len(groups)]))
        scene.set_kernel_timing(0)
    for _ in range(warmup):
        step()
    if steps <= 0:  # sub-record: as many frames as SUB_MIN_TIMED_S of GPU work needs (the same count on every rank)
        probe = timed(10) / 10
        steps = int(min(20000, max(20, math.ceil(SUB_MIN_TIMED_S / max(probe, 1e-7)))))
    # timed region: K frames.  On every period-th frame the transform launches carry a pair of HIP events each
    # (hipExtLaunchKernelGGL on the library's stream: the dispatch's own start/end timestamps, the ones rocprofv3's
    # kernel trace reports; nothing is added to the stream) — all transform launches of a sampled frame, not just the
    # dominant one: a launch whose predecessor is not instrumented reads ~3 us long.  An instrumented frame costs the
    # stream ~20 us: a dozen samples or more, every 64th frame at most for short frames, every 4th for frames of milliseconds.
    period = max(4, min(64, steps // 12))
    if not args.no_kernel_events:
        scene.set_kernel_timing(-1, period)
    elapsed = timed(steps)

    counts, bcounts = scene.counts()
    names = scene.group_kernel_names()
    list_entries = float(counts[:F].sum())
    own_frame, lists_frame = algorithmic_bytes(wl, lo, parent, 0, n, F, list_entries)
    ms_frame = elapsed / steps * 1e3
    working_set = own_frame + lists_frame
    cache_assisted = working_set < INFINITY_CACHE_BYTES
    roofline = None
    if not args.no_kernel_events:
        ms, samples = scene.group_timings()
        ms = ms.astype(np.float64)
        g = dominant
        first, end = int(groups[g][2]), int(groups[g][3])
        own, _ = algorithmic_bytes(wl, lo, parent, first, end, F, 0.0)
        achieved = own / (ms[g] * 1e-3) / 1e9 if ms[g] > 0 else 0.0
        sym = names[g] if g < len(names) else ""
        traffic, traffic_src = (profiled_traffic(name, sym) if n == wl["n"] // (world_size if wl["scaling"] == "strong" else 1)
                                else (None, None))
        # the PMC passes average per kernel NAME: if another launch of the frame runs the same kernel (cfg4: levels 2..4
        # and 5..7 are both k_frame_chain_big<3>), that average is not this launch's traffic
        same = [q for q in names if q == sym]
        if len(same) > 1 and traffic is not None:
            traffic, traffic_src = None, f"{traffic_src}: per-name mean {traffic:.0f} B covers {len(same)} launches of the frame"
        frac = achieved / HBM_PEAK_GBPS
        # What bounds the launch.  A scene whose frame-to-frame working set exceeds the 256 MB Infinity Cache streams
        # from HBM: its yardstick is the HBM peak.  A smaller one is served by the cache: the HBM peak says nothing
        # there, so a plain read + write stream over a working set of the same size and read/write mix is measured on
        # this GPU, in this run (yh_stream_probe), and the launch is quoted against THAT (`cache_ceiling`).
        bound = "infinity_cache" if cache_assisted else "hbm"
        note = None
        cache = None
        if cache_assisted:
            reads = own_frame - n * 64.0 - F * (n / 8.0 + n / 64.0)
            writes = working_set - reads
            probe_ms = ctx.stream_probe(reads, writes, 200)
            ceiling = working_set / (probe_ms * 1e-3) / 1e9
            cache = {"cache_ceiling": round(ceiling, 1), "cache_ceiling_unit": "GB/s",
                     "cache_ceiling_probe": f"yh_stream_probe: {reads / 1e6:.1f} MB read + {writes / 1e6:.1f} MB written per launch, "
                                            f"the same buffers 200 times, {probe_ms * 1e3:.2f} us per launch (plain 16-byte loads and stores)",
                     "frac_of_cache_ceiling": round(achieved / ceiling, 4),
                     "frame_frac_of_cache_ceiling": round(working_set / (ms_frame * 1e-3) / 1e9 / ceiling, 4)}
            note = (f"cache-resident: the frame's {working_set / 1e6:.0f} MB working set stays in the 256 MB Infinity Cache from one "
                    "frame to the next, so `achieved` is not an HBM rate and `frac` (of the HBM peak) is kept for the record only; "
                    "the yardstick is `cache_ceiling`")
        roofline = {
            "bound": bound, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(frac, 4), "traffic": traffic, "traffic_source": traffic_src,
            "kernel_symbol": sym,
            "kernel": f"{sym.split('<')[0].replace('void ', '')} levels {int(groups[g][0])}..{int(groups[g][1])}",
            "kernel_entities": end - first, "kernel_ms": round(float(ms[g]), 5), "kernel_samples": int(samples[g]),
            "kernel_sampling_period": period,
            "algorithmic_bytes_per_launch": own, "algorithmic_bytes_per_entity": round(own / (end - first), 2),
            "bytes_charged": "TRS + parent index + bounds + 64 B per distinct parent in, world matrices + visibility words and "
                             "count bytes out; the 4 B per emitted draw index are k_expand's and are charged to the frame only",
            "launch_ms_all_calibration": [round(float(x), 5) for x in ms_all[: len(groups)]],
            "expand_stage_ms_calibration": round(float(ms_all[63]), 5),
            "frac_of_measured_copy_ceiling_6290": round(achieved / 6290.0, 4),
            "frame_frac": round((own_frame + lists_frame) / (ms_frame * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
        }
        if cache:
            roofline.update(cache)
        if note:
            roofline["note"] = note
    scene.set_kernel_timing(False)

    rec = {
        "value": round(total * steps / elapsed / 1e6, 2),
        "unit": "Mentities/s",
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": round(ms_frame, 5),
        "timed_region_ms": round(elapsed * 1e3, 3),
        "scaling": wl["scaling"],
        "config": {"workload": f"{name}: {wl['desc']}", "entities_per_gpu": n, "entities_total": total,
                   "levels": wl["levels"], "frusta": F, "launches_per_frame": int(len(groups)),
                   "level_groups": [[int(x) for x in q] for q in groups], "kernels": names, "collective": collective,
                   "sharding": sharding},
        "roofline": roofline,
        "frame_hbm_frac": round((own_frame + lists_frame) / (ms_frame * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
        "frame_algorithmic_bytes_per_entity": round((own_frame + lists_frame) / n, 2),
        "visible_fraction": [round(float(c) / n, 4) for c in counts[:F]],
        "boundary_cases": [int(x) for x in bcounts[:F]],
    }
    if headline:
        # un-instrumented pass of the same K steps (reported beside, never as `value`)
        elapsed_plain = timed(steps)
        rec["value_without_kernel_events"] = round(total * steps / elapsed_plain / 1e6, 2)
        # per-frame latency with a host sync every frame (what an engine that reads counts each frame sees)
        lat = []
        for _ in range(min(20, steps)):
            a = time.perf_counter()
            scene.frame(planes, _lib.YH_FRAME_ALL_DIRTY)
            if gather:
                scene.allgather_draw_lists()
            lat.append(time.perf_counter() - a)
        rec["frame_latency_ms_synced_median"] = round(float(np.median(lat)) * 1e3, 5)
    # what the CPU leg of this workload needs (the hierarchy itself only where the leg runs the workload whole)
    extra = (name, wl, n, lo, parent if n <= CPU_SAMPLE.get(name, 0) else None, planes)
    scene.close()
    return rec, extra


def run_mat4(job):
    """The drop-in's own path (rule A3 literally, on imported flecs columns): a 12.5M-entity 8-level Mat4 scene - cfg4's
    per-GPU shard held as 68-byte LocalTransform / WorldTransform columns instead of TRS -, every row dirty, change
    tracking on (what GpuTransformCullSystem runs: the frame also files every row it touched for the readback).
    One launch of k_level_mat4 / k_chain_mat4 per level group, k_sweep_dirty, the expansions of the draw list and of
    the two change lists."""
    from yume_amd import Scene, _lib, synth

    ctx = job.ctx
    n, levels = 12_500_000, 8
    lo = synth.level_offsets(n, levels)
    par = synth.parents(lo)
    planes = synth.main_camera_planes().reshape(1, 24)
    with Scene(ctx, n, n_levels=levels, n_frusta=1) as t:  # locals composed on a TRS scene of the same synthetic inputs
        t.upload_hierarchy(par, lo)
        t.fill_synthetic()
        local_m = t.compose()
        mins, maxs = t.download_aabb()
    T68 = np.dtype([("m", np.float32, 16), ("dirty", np.uint8), ("pad", np.uint8, 3)])
    col = np.zeros(n, T68)
    col["m"] = local_m.T
    col["dirty"] = 1
    del local_m
    s = Scene(ctx, n, n_levels=levels, n_frusta=1, local_kind=_lib.YH_LOCAL_MAT4)
    s.upload_hierarchy(par, lo)
    s.upload_aabb(mins, maxs)
    s.upload_local_aos68(col)
    del col, mins, maxs
    s.track_changes(True)
    flags = _lib.YH_FRAME_ALL_DIRTY | _lib.YH_FRAME_ASYNC

    def timed(k):
        job.barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            s.frame(planes, flags)
        job.barrier()
        return time.perf_counter() - t0

    s.set_kernel_timing(1)  # calibration, untimed: events around every launch (they read 3-5 us long on launches of ~65 us)
    for _ in range(10):
        s.frame(planes, flags)
    ms_all, _ = s.group_timings()
    groups = s.groups()
    names = s.group_kernel_names()
    s.set_kernel_timing(0)
    for _ in range(5):
        s.frame(planes, flags)
    probe = timed(10) / 10
    steps = int(max(20, math.ceil(SUB_MIN_TIMED_S / max(probe, 1e-7))))
    elapsed = timed(steps)
    counts, _b = s.counts()
    nw, nl = s.changed_counts()
    # compulsory bytes of the all-dirty frame: local row 64 + both dirty bytes 2 + parent index 4 (levels > 0) + AABB 24 in,
    # 64 (+1 dirty byte) per distinct parent per level in, world row 64 + both dirty bytes cleared 2 out; stage 3: one bit
    # per entity in each of the three masks (visible, world changed, local changed) + their count bytes, 4 B per listed slot
    b = n * (64 + 2 + 24 + 64 + 2) + 4 * (n - int(lo[1]))
    for k in range(1, levels):
        b += 65 * int(np.count_nonzero(np.diff(par[int(lo[k]):int(lo[k + 1])])) + 1)
    b += 3 * (n / 8.0 + n / 64.0) + 4.0 * (float(counts[0]) + nw + nl)
    ms_frame = elapsed / steps * 1e3
    g = int(np.argmax(ms_all[: len(groups)]))
    first, end = int(groups[g][2]), int(groups[g][3])
    own = (end - first) * (64 + 2 + 24 + 4 + 64 + 1 + 1 / 8 + 1 / 64) + 65 * int(np.count_nonzero(np.diff(par[first:end])) + 1)
    rec = {
        "value": round(n * steps / elapsed / 1e6, 2), "unit": "Mentities/s", "steps": steps, "ms_per_step": round(ms_frame, 5),
        "timed_region_ms": round(elapsed * 1e3, 3), "scaling": "weak",
        "config": {"workload": "mat4: the drop-in's literal-rule path - 12.5M-entity 8-level scene held as 68-byte LocalTransform / "
                               "WorldTransform columns (YH_LOCAL_MAT4), every row dirty, change tracking on",
                   "entities_per_gpu": n, "levels": levels, "frusta": 1, "launches_per_frame": int(len(groups)),
                   "level_groups": [[int(x) for x in q] for q in groups], "kernels": names},
        "roofline": {"bound": "hbm", "achieved": round(own / (float(ms_all[g]) * 1e-3) / 1e9, 1) if ms_all[g] > 0 else 0.0,
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(own / (float(ms_all[g]) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if ms_all[g] > 0 else 0.0,
                     "traffic": None, "kernel_symbol": names[g] if g < len(names) else "",
                     "kernel_entities": end - first, "kernel_ms": round(float(ms_all[g]), 5),
                     "kernel_ms_source": "HIP events recorded around the launch, 10 untimed calibration frames (they read 3-5 us long)",
                     "algorithmic_bytes_per_launch": own,
                     "launch_ms_all_calibration": [round(float(x), 5) for x in ms_all[: len(groups)]],
                     "frame_frac": round(b / (ms_frame * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)},
        "frame_hbm_frac": round(b / (ms_frame * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
        "frame_algorithmic_bytes_per_entity": round(b / n, 2),
        "visible_fraction": [round(float(counts[0]) / n, 4)],
        "changed_rows": {"world": int(nw), "local": int(nl)},
    }
    # the other two kinds of frame the drop-in sends on this scene (never `value`): nothing dirty (the rule finds every
    # row clean), and YH_FRAME_CULL_ONLY (only a camera moved: one launch over the world rows the device holds)
    flags = _lib.YH_FRAME_ASYNC
    for _ in range(5):
        s.frame(planes, flags)
    clean_ms = timed(100) / 100 * 1e3
    flags = _lib.YH_FRAME_ASYNC | _lib.YH_FRAME_CULL_ONLY
    for _ in range(5):
        s.frame(planes, flags)
    cull_ms = timed(100) / 100 * 1e3
    cull_bytes = n * (64 + 24 + 1 / 8 + 1 / 64) + 4.0 * float(counts[0])  # world row, AABB in; visibility bit + count byte, list out
    rec["other_frames"] = {
        "nothing_dirty_ms": round(clean_ms, 5),
        "cull_only_ms": round(cull_ms, 5),
        "cull_only_Mentities_per_s": round(n / cull_ms / 1e3, 2),
        "cull_only_frame_hbm_frac": round(cull_bytes / (cull_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
        "cull_only_kernel": s.group_kernel_names()[0],
    }
    s.close()
    return rec


def main():
    args = parse()
    job = Job(args)
    rank, world_size = job.rank, job.world_size
    if world_size != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_size}; launch with torch.distributed.run",
                  file=sys.stderr)
        if world_size == 1 and args.gpus > 1:
            sys.exit(2)

    import torch

    from yume_amd import Context

    job.torch = torch
    if world_size > 1:
        import torch.distributed as dist

        # one GPU per rank; a launcher that narrows each rank's visible devices to one leaves only index 0
        job.local_rank = job.local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(job.local_rank)
        with stdout_to_stderr():
            dist.init_process_group("nccl", device_id=torch.device("cuda", job.local_rank))
            dist.barrier()  # torch's own communicator comes up (and prints) here, not inside the first timed barrier
        job.dist = dist
    job.ctx = Context(job.local_rank)
    job.info = job.ctx.device_info()

    # Python's cyclic collector is kept out of the timed loops: with torch imported a full collection takes ~40 ms
    # (seen as one 41 ms gap around the 1300th enqueued frame), as long as a thousand cfg2 frames.
    import gc

    gc.collect()
    gc.disable()

    job.rccl = ""
    if world_size > 1:
        v = torch.cuda.nccl.version()
        job.rccl = f"library: {Context.comm_describe()}; torch: rccl {'.'.join(str(x) for x in v)}"
        os.environ.setdefault("YUME_HIP_COMM_LOG", "1")  # one line per rank on stderr when the communicator comes up
        print(f"bench.py rank {rank}: {job.rccl}", file=sys.stderr, flush=True)

    def no_result(err):
        """N > 1 and the library's communicator is not there: the job has no value.  Said in the one JSON line, loudly."""
        if rank == 0:
            print(json.dumps({"metric": "Mentities/s transformed+culled per frame", "value": None, "unit": "Mentities/s",
                              "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
                              "higher_is_better": True, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                              "config": {"workload": args.workload, "rccl": job.rccl},
                              "error": f"the library's RCCL communicator could not be used on every rank: {err[:400]}"}), flush=True)
        sys.stdout.flush()
        os._exit(3)  # (not sys.exit: a rank stuck inside a half-built communicator would hang the interpreter's teardown)

    try:
        head, extra = run_workload(job, args.workload, args.steps, args.warmup, True, args.entities)
    except CommFailure as ex:
        no_result(str(ex))
    result = {
        "metric": "Mentities/s transformed+culled per frame",
        "value": head["value"],
        "unit": "Mentities/s",
        "n_gpus": world_size,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"],
        "higher_is_better": True,
        "scaling": head["scaling"],
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": dict(head["config"], device=job.info["name"], arch=job.info["arch"]),
        "roofline": head["roofline"],
    }
    for k in ("timed_region_ms", "frame_hbm_frac", "frame_algorithmic_bytes_per_entity", "value_without_kernel_events",
              "frame_latency_ms_synced_median", "visible_fraction", "boundary_cases"):
        result[k] = head[k]

    def emit():
        if rank == 0:
            print(json.dumps(result, default=float), flush=True)

    # sub-records: the other configurations, each >= SUB_MIN_TIMED_S of timed frames.  A watchdog makes sure the
    # headline line is printed even if one of them hangs (every rank exits at the same deadline).
    subs = [] if args.no_sub else [w for w in (args.sub.split(",") if args.sub else SUB_ORDER) if w in WORKLOADS and w != args.workload]
    result["workloads"] = {}
    done = threading.Event()

    def watchdog():
        if not done.wait(WATCHDOG_S):
            result["workloads"]["_error"] = f"sub-records did not finish within {WATCHDOG_S:.0f} s; the line was printed by the watchdog"
            emit()
            os._exit(0)

    if subs:
        threading.Thread(target=watchdog, daemon=True).start()
    t_sub = time.perf_counter()
    cpu_legs = [extra]
    for w in subs:
        late = job.max_over_ranks(time.perf_counter() - t_sub) > SUB_BUDGET_S
        if late:
            result["workloads"][w] = {"skipped": f"sub-record budget of {SUB_BUDGET_S:.0f} s used up"}
            continue
        try:
            rec, sub_extra = run_workload(job, w, 0, 20 if WORKLOADS[w]["n"] <= 2_000_000 else 5, False)
            cpu_legs.append(sub_extra)
            ok, err = True, ""
        except Exception as ex:  # noqa: BLE001
            rec, ok, err = None, False, str(ex)
        result["workloads"][w] = rec if ok else {"error": err[:300]}
        if not job.all_ok(ok):  # a rank failed: the others would wait for it in the next workload's collectives
            break
    # the drop-in's own path (Mat4 columns, literal rule, change tracking): one GPU only - it is what ONE engine process runs
    want_mat4 = (subs and not args.sub) or (not args.no_sub and "mat4" in args.sub.split(","))
    if want_mat4 and world_size == 1 and "_error" not in result["workloads"]:
        try:
            result["workloads"]["mat4"] = run_mat4(job)
        except Exception as ex:  # noqa: BLE001
            result["workloads"]["mat4"] = {"error": str(ex)[:300]}
    done.set()

    gc.enable()
    if rank == 0 and world_size == 1 and not args.no_cpu_baseline:
        # every workload's CPU leg on a sample of its own shape: the headline's gets --cpu-seconds, each
        # sub-record's a third of that
        for k, (name, wl, n, lo, parent, planes) in enumerate(cpu_legs):
            legs = cpu_baseline(name, wl, n, lo, parent, planes, args.cpu_seconds if k == 0 else args.cpu_seconds / 3)
            if k == 0:
                result.update(legs)
            elif isinstance(result["workloads"].get(name), dict):
                result["workloads"][name].update(legs)
        result["cpu_baseline_in_flecs"], result["dropin_in_flecs"] = flecs_legs(8)
    elif rank == 0:
        result["cpu_baseline"] = None

    job.ctx.close()
    if job.dist is not None:
        job.dist.barrier()
        job.dist.destroy_process_group()
    emit()


if __name__ == "__main__":
    main()
