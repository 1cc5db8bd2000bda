# rocprofv3 evidence for the camera-only frame (k_cull_only over 12.5M Mat4 columns); run via gpurun, outputs under gpurun_out/prof_cull
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_cull
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/mat4_bench.py 12500000 8 cull > $OUT/trace.txt 2> $OUT/trace.err
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/mat4_bench.py 12500000 8 cull > $OUT/fetch.txt 2> $OUT/fetch.err
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/mat4_bench.py 12500000 8 cull > $OUT/write.txt 2> $OUT/write.err
python3 - <<'PY'
import csv, glob, collections
out = "gpurun_out/prof_cull"
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 0.5:
            print("stats", r["Name"][:60], r["Calls"], r["AverageNs"])
for kind in ("fetch", "write"):
    acc = collections.defaultdict(list)
    for f in glob.glob(out + f"/pmc_{kind}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if len(v) > 20:
            print(kind, k, len(v), "mean KiB/dispatch %.0f" % (sum(v) / len(v)))
PY
