#!/bin/bash
# frame time for expand-stage variants: tile words x (counting launch | look-back)
for cfg in "1000000 8 1" "16777216 1 5" "12500000 8 1" "4000000 8 2"; do
  set -- $cfg
  for tw in 64 256; do
    for cnt in 0 100000000; do
      echo -n "n=$1 L=$2 F=$3 words=$tw counted_threshold=$cnt: "
      YH_TUNE_TILE_WORDS=$tw YH_TUNE_COUNTED=$cnt timeout -k 10 120 python tools/plan_run.py $1 $2 auto 0 100 $3 2>&1 | grep "^plan" | sed 's/.*kind 0: //'
    done
  done
done
