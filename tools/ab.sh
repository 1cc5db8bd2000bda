#!/bin/bash
# A/B of builds of libyume_hip.so on the GPU box: tools/ab.sh "<lib1.so> <lib2.so> ..." [workloads...]
# Runs bench.py for each workload with the in-tree library and with each given one; summary of one line per run.
set -e
LIBS=$1; shift
WL=${@:-cfg2 cfg3 cfg4}
mkdir -p gpurun_out
: > gpurun_out/ab.log
for w in $WL; do
  for lib in "" $LIBS; do
    echo "== $w lib=${lib:-tree}" >> gpurun_out/ab.log
    YUME_HIP_LIB=$lib timeout -k 10 200 python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline >> gpurun_out/ab.log 2>&1
  done
done
python - <<'PY'
import json
lab=None
for line in open('gpurun_out/ab.log'):
    if line.startswith('=='): lab=line.strip(); continue
    if line.startswith('{'):
        d=json.loads(line); r=d.get('roofline') or {}
        print(lab, 'value', d['value'], 'plain', d.get('value_without_kernel_events'), 'ms', d['ms_per_step'], 'kernel_ms', r.get('kernel_ms'), 'frac', r.get('frac'), r.get('launch_ms_all_calibration'), r.get('expand_stage_ms_calibration'))
PY
