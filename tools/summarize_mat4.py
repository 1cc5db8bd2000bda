"""gpurun_out/prof_mat4_{dirty,clean}/ (tools/profile_mat4.sh) -> profiles/r02/mat4_literal_rule_12m5.txt"""
import collections, csv, glob

out = []
for mode in ("dirty", "clean"):
    d = f"gpurun_out/prof_mat4_{mode}"
    out.append(open(f"{d}/run.txt").read().strip().splitlines()[-1])
    stats = {r["Name"]: r for r in csv.DictReader(open(sorted(glob.glob(f"{d}/trace/vm/*_kernel_stats.csv"))[-1]))}
    pmc = {}
    for kind in ("fetch", "write"):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(sorted(glob.glob(f"{d}/pmc_{kind}/vm/*_counter_collection.csv"))[-1])):
            agg[r["Kernel_Name"]][0] += 1
            agg[r["Kernel_Name"]][1] += float(r["Counter_Value"])
        pmc[kind] = {k: v[1] / v[0] for k, v in agg.items()}
    out.append("| kernel | calls | avg us (kernel trace) | FETCH_SIZE KiB | WRITE_SIZE KiB | traffic MB (2F+W) | TB/s |")
    out.append("|---|---|---|---|---|---|---|")
    for name, r in sorted(stats.items(), key=lambda kv: -float(kv[1]["TotalDurationNs"])):
        if int(r["Calls"]) < 50:
            continue
        f, w = pmc["fetch"].get(name), pmc["write"].get(name)
        mb = (2 * f + w) * 1024 / 1e6 if f is not None and w is not None else None
        us = float(r["AverageNs"]) / 1e3
        out.append(f"| `{name[:60]}` | {r['Calls']} | {us:.1f} | {f:.0f} | {w:.0f} | {mb:.1f} | {mb / us:.2f} |" if mb else f"| `{name[:60]}` | {r['Calls']} | {us:.1f} | | | | |")
    out.append("")
open("profiles/r02/mat4_literal_rule_12m5.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
