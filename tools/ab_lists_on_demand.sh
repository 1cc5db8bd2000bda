# The multi-GPU step rehearsed on one GPU (one-rank communicator, --force-comm): frames that also expand their own lists
# against frames that leave them to the gather (YH_FRAME_LISTS_ON_DEMAND, bench.py's default with a gather).
one() { label=$1; w=$2; shift 2
  env "$@" timeout -k 10 300 python bench.py --workload $w --no-sub --no-cpu-baseline --steps ${STEPS:-400} $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-34s %s frame %.5f ms' % ('$label', '$w', d['ms_per_step']))"
}
for rep in 1 2; do
for spec in "cfg2 2000" "cfg4 400" "cfg3 400" "cfg5 60"; do
  set -- $spec; export STEPS=$2
  EXTRA= one "no gather" $1 A=1
  EXTRA=--force-comm one "gather, own lists expanded too" $1 YUME_BENCH_OWN_LISTS=1
  EXTRA=--force-comm one "gather, own lists on demand" $1 A=1
done; done
