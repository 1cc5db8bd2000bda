#!/bin/bash
# Per-kernel times of the bucketed draw-list consumer (rocprofv3 kernel trace) at several slice counts.
# usage (GPU box, repo root): bash tools/profile_buckets.sh <keys> <slices>...   -> gpurun_out/bk_<keys>_<slices>.txt
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
keys=$1; shift
for sl in "$@"; do
  if [ "$sl" != default ]; then export YH_BUCKET_SLICES=$sl; else unset YH_BUCKET_SLICES; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bk_${keys}_$sl -o bk -- python3 tools/bucket_bench.py 64e6 $keys > gpurun_out/bk_${keys}_$sl.txt 2> gpurun_out/bk_${keys}_$sl.err
  echo "keys=$keys slices=$sl: $(tail -1 gpurun_out/bk_${keys}_$sl.txt)"
  python3 - "$keys" "$sl" <<'PY'
import csv, glob, sys
f = glob.glob(f"gpurun_out/bk_{sys.argv[1]}_{sys.argv[2]}/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "bucket" in r["Name"] or "build_objects" in r["Name"]:
        print("   %-44s calls %4s  avg %10.1f us" % (r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
