# Same-box A/B: visibility words of all frusta in ONE store per wave (lane f holds frustum f's word) against one
# single-lane store per frustum.
one() { label=$1; w=$2; shift 2
  env "$@" timeout -k 10 200 python bench.py --workload $w --no-sub --no-cpu-baseline --steps ${STEPS:-100} 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('%-44s %s  frame %.5f ms  dominant %.5f ms  frac %.4f' % ('$label', '$w', d['ms_per_step'], r['kernel_ms'], r['frac']))"
}
for rep in 1 2; do
for w in cfg3 cfg5 cfg4 cfg2; do
  if [ $w = cfg5 ]; then export STEPS=40; elif [ $w = cfg2 ]; then export STEPS=1600; else export STEPS=200; fi
  one "in-tree: a single-lane store per frustum" $w A=1
  one "one store for the words of all frusta" $w YUME_HIP_LIB=yume_amd/csrc/ab/libyume_hip_publish_one_store.so
done; done
