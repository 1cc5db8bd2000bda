# GPU side of tools/ab_cache_policy.sh: cfg5 / cfg3 / cfg4 with each store-policy build.
one() { label=$1; w=$2; shift 2
  env "$@" timeout -k 10 200 python bench.py --workload $w --no-sub --no-cpu-baseline --steps ${STEPS:-100} 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('%-44s %s  frame %.5f ms  dominant %.5f ms  frac %.4f' % ('$label', '$w', d['ms_per_step'], r['kernel_ms'], r['frac']))"
}
for w in cfg5 cfg3 cfg4; do
  if [ $w = cfg5 ]; then export STEPS=40; else export STEPS=200; fi
  one "nt (default build)" $w A=1
  one "plain stores" $w YUME_HIP_LIB=yume_amd/csrc/ab/libyume_hip_store1.so
  one "sc1" $w YUME_HIP_LIB=yume_amd/csrc/ab/libyume_hip_store2.so
  one "sc0 sc1" $w YUME_HIP_LIB=yume_amd/csrc/ab/libyume_hip_store3.so
  one "sc1 nt" $w YUME_HIP_LIB=yume_amd/csrc/ab/libyume_hip_store4.so
  one "sc0 nt" $w YUME_HIP_LIB=yume_amd/csrc/ab/libyume_hip_store5.so
done
