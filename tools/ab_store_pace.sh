# Same-box A/B: idle issue slots after each of the sixteen world-matrix stores of the flat frame kernels.
one() { label=$1; w=$2; shift 2
  env "$@" timeout -k 10 200 python bench.py --workload $w --no-sub --no-cpu-baseline --steps ${STEPS:-100} 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('%-44s %s  frame %.5f ms  dominant %.5f ms  frac %.4f' % ('$label', '$w', d['ms_per_step'], r['kernel_ms'], r['frac']))"
}
for rep in 1 2; do
for w in cfg5 cfg3; do
  if [ $w = cfg5 ]; then export STEPS=40; else export STEPS=200; fi
  one "no pacing (s_nop 0 build)" $w YUME_HIP_LIB=yume_amd/csrc/ab/libyume_hip_pace0.so
  one "in-tree: s_nop 7" $w A=1
  one "s_nop 11" $w YUME_HIP_LIB=yume_amd/csrc/ab/libyume_hip_pace11.so
  one "s_nop 15" $w YUME_HIP_LIB=yume_amd/csrc/ab/libyume_hip_pace15.so
  one "s_sleep 1" $w YUME_HIP_LIB=yume_amd/csrc/ab/libyume_hip_pace101.so
  one "s_sleep 2" $w YUME_HIP_LIB=yume_amd/csrc/ab/libyume_hip_pace102.so
done; done
