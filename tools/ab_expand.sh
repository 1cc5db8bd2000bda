# Same-box A/B of the expansion: entries of each word straight to the list against LDS-staged aligned rows.
one() { label=$1; w=$2; shift 2
  env "$@" timeout -k 10 200 python bench.py --workload $w --no-sub --no-cpu-baseline --steps ${STEPS:-100} 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('%-56s %s  frame %.5f ms  dominant %.5f ms  expand stage %.5f ms' % ('$label', '$w', d['ms_per_step'], r['kernel_ms'], r['expand_stage_ms_calibration']))"
}
for rep in 1 2; do
for w in cfg5 cfg3 cfg4 cfg2; do
  if [ $w = cfg5 ]; then export STEPS=40; elif [ $w = cfg2 ]; then export STEPS=1600; else export STEPS=200; fi
  one "in-tree: every word's entries straight to the list" $w A=1
  one "rows staged in LDS (-DYH_EXPAND_LDS)" $w YUME_HIP_LIB=yume_amd/csrc/ab/libyume_hip_expand_lds.so
done; done
