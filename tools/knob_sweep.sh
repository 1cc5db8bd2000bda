#!/bin/bash
# frame time under the library's tuning knobs (flat 16M x 5 frusta, flat 16M x 1, 12.5M tree)
for cfg in "16777216 1 5" "16777216 1 1" "12500000 8 1"; do
  set -- $cfg
  for it in 1 2 4; do
    for pf in 0 1; do
      echo -n "n=$1 L=$2 F=$3 iters=$it prefetch=$pf: "
      YH_TUNE_ITERS=$it YH_TUNE_PREFETCH=$pf timeout -k 10 120 python tools/plan_run.py $1 $2 auto 0 100 $3 2>&1 | grep "^plan" | sed 's/.*kind 0: //'
    done
  done
done
