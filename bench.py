#!/usr/bin/env python3
"""bench.py — Mentities/s transformed + culled per frame, % of the HBM roofline, CPU systems beside it.

A "step" is one frame of the hot path (compose -> level-ordered propagate -> cull + compact, count read-back)
over one synthetic scene that is already resident in HBM.  One process per GPU:

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workloads (BASELINE.json configs): cfg2 (default; 1M entities, random 8-level tree, 1 frustum, per GPU),
cfg3 (16M flat, 5 frusta), cfg4 (12.5M-entity 8-level shard per GPU = 100M on 8), cfg5 (256M flat instanced
AABBs, split over the GPUs).  With N > 1 every rank owns a subtree-closed shard (its own tree, global ids
offset by shard_base) and each frame ends with the RCCL all-gather of counts and draw lists.  Scaling is
weak for cfg2/cfg3/cfg4 (per-GPU shard fixed) and strong for cfg5.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)

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


# Frames enqueued before the host waits for the device once (inside the timed region: it is part of the cost).  Without
# a bound the HIP runtime's command and kernel-argument pools run dry after ~1000 un-synchronised launches and the
# stream stalls in long phases (measured: 800 frames of cfg2 with no sync take 70 us per frame instead of 34).
FRAMES_IN_FLIGHT = 256


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1600)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--entities", type=int, default=0, help="override entities per GPU")
    ap.add_argument("--grouping", type=int, default=0, help="levels per launch (0 = automatic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-comm", action="store_true",
                    help="rehearsal on one GPU: bring up a one-rank RCCL communicator and run the multi-GPU step "
                         "(frame + pipelined all-gather) - measures what the exchange adds to a frame")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket launches with HIP events")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def algorithmic_bytes(wl, lo, parent, first, end, visible_per_entity_sum):
    """compulsory HBM bytes of a fully fused pass over entities [first, end).
    reads: TRS 40 B + parent index 4 B (levels > 0) + AABB 24 B (or 4 B mesh id) + 64 B per DISTINCT parent per
    level; writes: world 64 B + 4 B per emitted draw index."""
    from yume_amd import synth

    n = end - first
    b = n * (40 + 64) + n * (24 if wl["bounds"] == 1 else 4)
    n_child = end - max(first, int(lo[1]))
    if n_child > 0:
        b += 4 * n_child
        # distinct parents of the levels that intersect [first, end)
        for k in range(1, len(lo) - 1):
            a, z = max(first, int(lo[k])), min(end, int(lo[k + 1]))
            if z > a:
                p = parent[a:z]
                b += 64 * int(np.count_nonzero(np.diff(p)) + 1)
    b += 4 * visible_per_entity_sum * n
    return float(b)


def cpu_baseline(args, wl, n, lo, parent, planes, seconds):
    """The reference CPU systems (oracle port: AoS 68-byte columns, one batch per table, forward branch on every
    row + both dirty sweeps, then the cull), single thread as the reference runs them, on the host cores of
    this box.  Only this leg touches oracle/."""
    from oracle import pyoracle as oracle

    n_s = min(n, 1_000_000)  # bounded sample: at most 1M entities per CPU frame
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
                      + ("" if n_s == n else f"; sampled from the {n}-entity workload"),
        }
    return out


def cpu_baseline_flecs(wl, lo_levels):
    """The same rule driven by the REAL flecs 4.0.4 the reference vendors (oracle/_ref/flecs_harness, compiled from the
    reference's flecs sources in place; the per-row arithmetic is the oracle's restatement: Zig cannot be built here).
    Transform systems only (the reference has no cull), every LocalTransform dirty in frame 0, on a bounded sample."""
    import struct
    import subprocess
    import tempfile

    from oracle import pyoracle as oracle

    here = os.path.dirname(os.path.abspath(__file__))
    exe = os.path.join(here, "oracle", "_ref", "flecs_harness")
    if not os.path.exists(exe):
        return None
    n_s = 200_000
    lo = oracle.synth_level_offsets(n_s, lo_levels)
    par = oracle.synth_parents(lo)
    local = oracle.compose_batch(oracle.synth_trs(n_s, int(lo[1])))
    root = oracle.identity68()["m"][0]
    edit = np.asarray(oracle.compose([1, 2, 3], [0, 0, 0, 1], [1, 1, 1]), np.float32)
    n_frames = 6  # frame 0: everything dirty (creation); then the root is re-set each frame: every row recomputes
    with tempfile.TemporaryDirectory() as d:
        scene = os.path.join(d, "scene.bin")
        with open(scene, "wb") as f:
            f.write(b"YFH1")
            f.write(struct.pack("<III", n_s, lo_levels, n_frames))
            f.write(lo.astype(np.uint32).tobytes())
            f.write(par.astype(np.uint32).tobytes())
            f.write(np.ascontiguousarray(local["m"], np.float32).tobytes())
            f.write(np.ascontiguousarray(root, np.float32).tobytes())
            for k in range(n_frames):
                if k == 0:
                    f.write(struct.pack("<I", 0))
                else:
                    f.write(struct.pack("<III", 1, 0xFFFFFFFF, 1))
                    f.write(edit.tobytes())
        try:
            r = subprocess.run([exe, scene, os.path.join(d, "out.bin")], env=dict(os.environ, YFH_TIMES="1"),
                               capture_output=True, text=True, timeout=300)
        except Exception:  # noqa: BLE001
            return None
    ms = [float(line.split(": ")[1].split(" ms")[0]) for line in r.stderr.splitlines() if line.startswith("frame ")]
    if len(ms) < 2:
        return None
    med = float(np.median(ms[1:]))
    return {"value": round(n_s / med / 1e3, 3), "unit": "Mentities/s", "cores": 1, "kind": "port",
            "ns_per_entity": round(med * 1e6 / n_s, 2),
            "sample": f"{len(ms) - 1} all-dirty frames of a {n_s}-entity {lo_levels}-level scene inside the reference's vendored "
                      f"flecs 4.0.4 (TransformSyncSystem rule + both PostTransform sweeps, no cull; median)"}


def kernel_symbol(levels_in_group, bounds, has_base, n_entities=1 << 40, n_levels=8):
    """Name rocprofv3 reports for the launch of a level group (yume_hip.hip launch_trs / launch_chain / launch_first_x)."""
    hb = "true" if has_base else "false"
    if not has_base and n_entities <= 64 * 1024 * 64:  # up to 1024 expand tiles: the first launch carries the previous frame's expansion
        if levels_in_group == 1:
            return f"void k_frame_trs_x<{bounds}>(FrameArgs, ExpandArgs)"
        g = next(x for x in (2, 3, 4, 6, 8) if levels_in_group <= x)
        return f"void k_frame_chain_x<{g}, {bounds}>(FrameArgs, ExpandArgs)"
    # scenes that outgrow the Infinity Cache use the variants with streaming input loads (YH_BIG_FLAT / YH_BIG_TREE)
    big = n_entities >= ((5 << 20) if n_levels == 1 else (7 << 19))
    if levels_in_group == 1:
        return f"void k_frame_trs{'_big' if big else ''}<{bounds}, {hb}>(FrameArgs)"
    g = next(x for x in (2, 3, 4, 6, 8) if levels_in_group <= x)
    return f"void k_frame_chain{'_big' if big and g <= 3 else ''}<{g}, {bounds}, {hb}>(FrameArgs)"


def profiled_traffic(workload, symbol):
    """HBM bytes per launch of `symbol` from the committed rocprofv3 PMC passes (profiles/r*/, newest round first):
    2 x FETCH_SIZE + WRITE_SIZE, KiB units, gfx950 FETCH_SIZE correction as calibrated in profiles/r01/SUMMARY.md."""
    import csv
    import glob

    here = os.path.dirname(os.path.abspath(__file__))
    for d in sorted(glob.glob(os.path.join(here, "profiles", "r*")), reverse=True):
        vals = {}
        for kind in ("fetch", "write"):
            f = os.path.join(d, f"{workload}_pmc_{kind}_mean_kib_per_dispatch.csv")
            if os.path.exists(f):
                for r in csv.DictReader(open(f)):
                    if r["kernel"] == symbol:
                        vals[kind] = float(r["mean_KiB_per_dispatch"])
        if len(vals) == 2:
            return (2 * vals["fetch"] + vals["write"]) * 1024.0, os.path.relpath(d, here)
    return None, None


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_size}; launch with torch.distributed.run",
                  file=sys.stderr)
        if world_size == 1 and args.gpus > 1:
            sys.exit(2)

    import torch

    from yume_amd import Context, Scene, _lib, synth

    dist = None
    if world_size > 1:
        import torch.distributed as dist

        # one GPU per rank; a launcher that narrows each rank's visible devices to one leaves only index 0
        local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    wl = WORKLOADS[args.workload]
    n = args.entities or wl["n"]
    if wl["scaling"] == "strong":
        n = n // world_size
    total = n * world_size
    planes = synth.main_camera_planes().reshape(1, 24)
    if wl["frusta"] == "main+casc":
        planes = np.concatenate([planes, synth.cascade_planes()])
    F = planes.shape[0]

    ctx = Context(local_rank)
    info = ctx.device_info()
    lo = synth.level_offsets(n, wl["levels"])
    parent = synth.parents(lo, index_base=rank * n)
    scene = Scene(ctx, n, n_levels=wl["levels"], n_frusta=F, bounds_kind=wl["bounds"],
                  max_meshes=16 if wl["bounds"] == 2 else 0, id_base=rank * n)
    scene.upload_hierarchy(parent, lo)
    scene.fill_synthetic(index_base=rank * n)  # inputs generated straight into HBM
    scene.set_grouping(args.grouping)
    flags = _lib.YH_FRAME_ALL_DIRTY | _lib.YH_FRAME_ASYNC

    collective = "none"
    use_lib_comm = True
    gather = world_size > 1 or args.force_comm
    if world_size == 1 and args.force_comm:
        ctx.comm_init(1, 0, Context.comm_unique_id())
        collective = "rccl (library, collective stream), one-rank rehearsal"
    if world_size > 1:
        # product path: the library's own RCCL communicator (ncclAllGather of counts + grouped ncclBroadcast of the
        # exact-size lists on the library's stream).  If it cannot be brought up on every rank, stand by with
        # torch.distributed ("nccl" = RCCL as well) over downloaded lists and say so in the report.
        err = ""
        try:
            ids = [Context.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            ctx.comm_init(world_size, rank, ids[0])
            scene.frame(planes, _lib.YH_FRAME_ALL_DIRTY)
            scene.allgather_draw_lists()        # exact-size exchange (host sync): checks the communicator end to end
            scene.allgather_draw_lists_async()  # the pipelined exchange the timed frames use
            scene.gathered_counts(F)
            ok = 1
        except Exception as ex:  # noqa: BLE001
            ok, err = 0, str(ex)
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        use_lib_comm = bool(flag.item())
        collective = ("rccl (library, collective stream): one ncclAllGather of the frame's visibility words (n/8 bytes per "
                      "frustum and rank), every rank expands all ranks' words into the gathered lists; runs beside the next "
                      "frames" if use_lib_comm
                      else f"torch.distributed nccl stand-by (library communicator failed: {err[:120]})")

    def step():
        scene.frame(planes, flags)
        if gather:
            if use_lib_comm:
                scene.allgather_draw_lists_async()
            else:
                from yume_amd import shard

                shard.allgather_draw_lists_torch([scene.draw_list(f) for f in range(F)], device="cuda")

    def barrier():
        ctx.wait()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # Python's cyclic collector is kept out of the timed loops: with torch imported a full collection takes ~40 ms
    # (seen as one 41 ms gap around the 1300th enqueued frame), as long as a thousand frames.
    import gc

    gc.collect()
    gc.disable()
    # calibration (untimed): HIP events around every launch to find the dominant kernel of a frame
    groups = scene.groups()
    dominant = 0
    if not args.no_kernel_events:
        scene.set_kernel_timing(1)
        for _ in range(20):
            step()
        ms_all, _ = scene.group_timings()
        ms_all = ms_all.astype(np.float64)
        dominant = int(np.argmax(ms_all[: len(groups)]))
        scene.set_kernel_timing(0)
    for _ in range(args.warmup):
        step()
    barrier()
    # timed region: K frames; on every (K / 25)-th frame, between every 4th and every 64th (25 samples at the default
    # K = 1600: an instrumented frame costs the stream ~20 us, 1-2 % of `value` at that rate), the
    # transform launches carry a pair of HIP events each (hipExtLaunchKernelGGL on the library's stream: the dispatch's
    # own start/end timestamps, the ones rocprofv3's kernel trace reports; nothing is added to the stream).  All transform launches of a sampled
    # frame, not just the dominant one: a launch whose predecessor is not instrumented reads ~3 us long.
    period = max(4, min(64, args.steps // 25))
    if not args.no_kernel_events:
        scene.set_kernel_timing(-1, period)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step()
        if (i + 1) % FRAMES_IN_FLIGHT == 0:
            ctx.wait()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # dominant kernel: mean launch duration from the HIP events recorded inside the timed region
    roofline = None
    counts, bcounts = scene.counts()
    vis_sum = float(counts[:F].sum()) / n
    if not args.no_kernel_events:
        ms, samples = scene.group_timings()
        ms = ms.astype(np.float64)
        g = dominant
        first, end = int(groups[g][2]), int(groups[g][3])
        b = algorithmic_bytes(wl, lo, parent, first, end, vis_sum)
        achieved = b / (ms[g] * 1e-3) / 1e9
        sym = kernel_symbol(int(groups[g][1]) - int(groups[g][0]) + 1, wl["bounds"], int(groups[g][0]) > 0, n, wl["levels"])
        traffic, traffic_src = profiled_traffic(args.workload, sym) if n == wl["n"] // (world_size if wl["scaling"] == "strong" else 1) else (None, None)
        # the PMC passes average per kernel NAME: if another launch of the frame runs the same kernel (cfg4: levels 2..4
        # and 5..7 are both k_frame_chain_big<3>), that average is not this launch's traffic
        same = [q for q in groups if kernel_symbol(int(q[1]) - int(q[0]) + 1, wl["bounds"], int(q[0]) > 0, n, wl["levels"]) == sym]
        if len(same) > 1 and traffic is not None:
            traffic, traffic_src = None, f"{traffic_src}: per-name mean {traffic:.0f} B covers {len(same)} launches of the frame"
        roofline = {
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_src,
            "kernel_symbol": sym,
            "kernel": f"{sym.split('<')[0].replace('void ', '')} levels {int(groups[g][0])}..{int(groups[g][1])}",
            "kernel_entities": end - first, "kernel_ms": round(float(ms[g]), 5), "kernel_samples": int(samples[g]), "kernel_sampling_period": period,
            "algorithmic_bytes_per_launch": b, "algorithmic_bytes_per_entity": round(b / (end - first), 2),
            "launch_ms_all_calibration": [round(float(x), 5) for x in ms_all[: len(groups)]],
            "expand_stage_ms_calibration": round(float(ms_all[63]), 5),
            "frac_of_measured_copy_ceiling_6290": round(achieved / 6290.0, 4),
        }
    scene.set_kernel_timing(False)

    # un-instrumented pass of the same K steps (reported beside, never as `value`)
    barrier()
    t1 = time.perf_counter()
    for i in range(args.steps):
        step()
        if (i + 1) % FRAMES_IN_FLIGHT == 0:
            ctx.wait()
    barrier()
    elapsed_plain = time.perf_counter() - t1
    if dist is not None:
        t = torch.tensor([elapsed_plain], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed_plain = float(t.item())

    # per-frame latency with a host sync every frame (what an engine that reads counts each frame sees)
    lat = []
    for _ in range(min(50, args.steps)):
        a = time.perf_counter()
        scene.frame(planes, _lib.YH_FRAME_ALL_DIRTY)
        if gather and use_lib_comm:
            scene.allgather_draw_lists()
        lat.append(time.perf_counter() - a)

    gc.enable()
    b_frame = algorithmic_bytes(wl, lo, parent, 0, n, vis_sum)
    result = {
        "metric": "Mentities/s transformed+culled per frame",
        "value": round(total * args.steps / elapsed / 1e6, 2),
        "unit": "Mentities/s",
        "n_gpus": world_size,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 5),
        "higher_is_better": True,
        "scaling": wl["scaling"],
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{args.workload}: {wl['desc']}", "entities_per_gpu": n, "entities_total": total,
                   "levels": wl["levels"], "frusta": F, "launches_per_frame": int(len(groups)),
                   "level_groups": [[int(x) for x in q] for q in groups], "collective": collective,
                   "device": info["name"], "arch": info["arch"]},
        "roofline": roofline,
        "frame_hbm_frac": round(b_frame * world_size / (elapsed / args.steps) / 1e9 / (HBM_PEAK_GBPS * world_size), 4),
        "frame_algorithmic_bytes_per_entity": round(b_frame / n, 2),
        "value_without_kernel_events": round(total * args.steps / elapsed_plain / 1e6, 2),
        "frame_latency_ms_synced_median": round(float(np.median(lat)) * 1e3, 5),
        "visible_fraction": [round(float(c) / n, 4) for c in counts[:F]],
        "boundary_cases": [int(x) for x in bcounts[:F]],
    }
    if rank == 0 and world_size == 1 and not args.no_cpu_baseline:
        result.update(cpu_baseline(args, wl, n, lo, parent, planes, args.cpu_seconds))
        result["cpu_baseline_in_flecs"] = cpu_baseline_flecs(wl, wl["levels"])
    elif rank == 0:
        result["cpu_baseline"] = None

    scene.close()
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result, default=float), flush=True)


if __name__ == "__main__":
    main()
