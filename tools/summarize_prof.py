"""Turns gpurun_out/prof_<workload>/ (tools/profile.sh) into committed evidence under profiles/."""
import collections, csv, glob, json, os, shutil, sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
out_dir = os.path.join("profiles", rnd)
os.makedirs(out_dir, exist_ok=True)
lines = [f"# rocprofv3 evidence, round {rnd[1:]}", "",
         f"Collected on one MI355X (gfx950) with `tools/profile_{rnd}.sh`: `default` is the driver's own command (`python3 bench.py`,",
         "CPU leg off) under `--kernel-trace --stats`; per workload `tools/profile.sh <workload>` = one `--kernel-trace --stats` run of",
         "`bench.py --workload <w> --no-sub --steps 400 --warmup 20`, then separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` runs",
         "(20 steps).  Counter units are KiB per dispatch.  gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE",
         "reports exactly 1/2 of coalesced streaming reads — calibrated here on the flat 16M kernel, whose compulsory",
         "reads are 64 B x 16,777,216 = 1,048,576 KiB against FETCH_SIZE = 524,327 KiB (ratio 0.50003); WRITE_SIZE is",
         "exact (k_synth writes 64 B/entity: 1,048,576 KiB expected and reported).  traffic = 2 x FETCH_SIZE + WRITE_SIZE.", ""]
for d in sorted(glob.glob("gpurun_out/prof_*")):
    w = os.path.basename(d)[5:]
    stats = sorted(glob.glob(f"{d}/trace/vm/*_kernel_stats.csv"), key=os.path.getmtime, reverse=True)  # newest run first
    if not stats:
        continue
    shutil.copy(stats[0], os.path.join(out_dir, f"{w}_kernel_stats.csv"))
    pmc = {}
    for kind in ("fetch", "write"):
        f = sorted(glob.glob(f"{d}/pmc_{kind}/vm/*_counter_collection.csv"), key=os.path.getmtime, reverse=True)
        if not f:
            continue
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f[0])):
            a = agg[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        pmc[kind] = {k: v[1] / v[0] for k, v in agg.items()}
        with open(os.path.join(out_dir, f"{w}_pmc_{kind}_mean_kib_per_dispatch.csv"), "w") as o:
            o.write("kernel,dispatches,mean_KiB_per_dispatch\n")
            for k, v in sorted(agg.items()):
                o.write(f"\"{k}\",{v[0]},{v[1] / v[0]:.1f}\n")
    try:
        bench = json.loads(open(f"{d}/bench_trace.json").read().strip().splitlines()[-1])
        json.dump(bench, open(os.path.join(out_dir, f"{w}_bench_under_rocprof.json"), "w"), indent=1)
    except Exception:
        bench = None
    lines += [f"## {w}", ""]
    if bench:
        rf = bench.get("roofline") or {}
        lines += [f"bench under rocprof: {bench['value']} {bench['unit']}, {bench['ms_per_step']} ms/frame; dominant kernel "
                  f"`{rf.get('kernel')}` {rf.get('kernel_ms')} ms by HIP events, {rf.get('achieved')} GB/s algorithmic "
                  f"(frac {rf.get('frac')}).  With the profiler attached the launch's event pair reads a few µs above "
                  f"the dispatch time below; in an un-profiled run it reads the same (DESIGN.md §5).", ""]
    lines += ["| kernel | calls | avg ns (kernel-trace) | FETCH_SIZE KiB | WRITE_SIZE KiB | traffic MB (2F+W) |", "|---|---|---|---|---|---|"]
    for r in csv.DictReader(open(stats[0])):
        k = r["Name"]
        if not (k.startswith("void k_") or k.startswith("k_")):
            continue
        fz, wz = pmc.get("fetch", {}).get(k), pmc.get("write", {}).get(k)
        tr = f"{(2 * fz + wz) * 1024 / 1e6:.1f}" if fz is not None and wz is not None else ""
        lines.append(f"| `{k[:60]}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {fz if fz is None else round(fz)} | {wz if wz is None else round(wz)} | {tr} |")
    lines.append("")
open(os.path.join(out_dir, "SUMMARY.md"), "w").write("\n".join(lines))
print("\n".join(lines))
