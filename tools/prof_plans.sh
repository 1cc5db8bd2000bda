#!/bin/bash
# kernel-trace of a few launch plans: tools/prof_plans.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/plans
i=0
for spec in "5,7 1" "7 2" "6,7 1" "3,7 1"; do
  set -- $spec
  i=$((i+1))
  (cd $R && timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/plans/p$i -o p -- python3 tools/plan_run.py 1000000 8 $1 $2 300) > $R/gpurun_out/plans/p$i.log 2>&1 || exit 1
  grep "^plan" $R/gpurun_out/plans/p$i.log
  python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/plans/p$i/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if int(r["Calls"])>=300: print("   ", r["Name"][:60], r["Calls"], r["AverageNs"])
PY
done
