one() { label=$1; w=$2; shift 2
  env "$@" timeout -k 10 200 python bench.py --workload $w --no-sub --no-cpu-baseline --steps ${STEPS:-100} 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('%-44s %s  frame %.5f ms  dominant %.5f ms  frac %.4f' % ('$label', '$w', d['ms_per_step'], r['kernel_ms'], r['frac']))"
}
for rep in 1 2; do
for w in cfg5 cfg3 cfg4 cfg2; do
  if [ $w = cfg5 ]; then export STEPS=40; elif [ $w = cfg2 ]; then export STEPS=1600; else export STEPS=200; fi
  one "in-tree: sc1 nt, immediate offsets" $w A=1
  one "round-1 nt stores (same tree otherwise)" $w YUME_HIP_LIB=yume_amd/csrc/ab/libyume_hip_worldnt.so
  one "experiment: sc1 nt, address per store" $w YUME_HIP_LIB=yume_amd/csrc/ab/libyume_hip_store4.so
done; done
