# Same-box A/B: consecutive tiles per k_expand workgroup (base carried, next tile's words prefetched).
one() { label=$1; w=$2; shift 2
  env "$@" timeout -k 10 200 python bench.py --workload $w --no-sub --no-cpu-baseline --steps ${STEPS:-100} 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('%-24s %s  frame %.5f ms  dominant %.5f ms  expand stage %.5f ms' % ('$label', '$w', d['ms_per_step'], r['kernel_ms'], r['expand_stage_ms_calibration']))"
}
for w in cfg5 cfg3 cfg4; do
  if [ $w = cfg5 ]; then export STEPS=40; else export STEPS=200; fi
  for T in 1 2 4 8 16; do one "$T tiles per workgroup" $w YH_EXPAND_TILES=$T; done
done
