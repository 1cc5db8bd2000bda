# A/B: parents staged in LDS (north_star's wording) vs gathered through L1/L2, one level per launch, and the default plan.
one() { label=$1; w=$2; shift 2
  env "$@" timeout -k 10 200 python bench.py --workload $w --no-sub --no-cpu-baseline --steps ${STEPS:-300} $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('%-70s %s  frame %.5f ms  %9.1f M ent/s  launches (ms): %s' % ('$label', '$w', d['ms_per_step'], d['value'], ' '.join('%.5f' % x for x in r['launch_ms_all_calibration'])))"
}
for w in cfg2 cfg4; do
  if [ $w = cfg2 ]; then export STEPS=1600; else export STEPS=300; fi
  EXTRA="--grouping 1" one "one level per launch, parents gathered through L1/L2" $w A=1
  EXTRA="--grouping 1" one "one level per launch, parents staged in LDS" $w YH_LDS_PARENTS=1
  EXTRA="" one "default plan (chain launches; parents gathered)" $w A=1
done
