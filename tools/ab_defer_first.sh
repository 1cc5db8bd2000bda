# Same-box A/B: how large may a tree's FIRST launch be and still carry the previous frame's expansion (k_frame_chain_x)?
# YH_DEFER_FIRST_MAX = entities in the first launch up to which the expansion rides on it (default 320K).
one() { label=$1; w=$2; n=$3; shift 3
  env "$@" timeout -k 10 200 python bench.py --workload $w --entities $n --no-sub --no-cpu-baseline --steps ${STEPS:-400} 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-22s %s n=%-10s frame %.5f ms  plan %s' % ('$label', '$w', '$n', d['ms_per_step'], d['config'].get('launch_plan')))"
}
for rep in 1 2; do
for n in ${SIZES:-4000000 12500000 25000000}; do
  one "first<=320K (default)" cfg4 $n A=1
  one "first<=1M" cfg4 $n YH_DEFER_FIRST_MAX=1048576
  one "first<=4M" cfg4 $n YH_DEFER_FIRST_MAX=4194304
done; done
for rep in 1 2; do
for n in 1500000 2000000 3000000; do
  STEPS=1000 one "first<=320K" cfg4 $n YH_DEFER_FIRST_MAX=327680
  STEPS=1000 one "first<=1M" cfg4 $n YH_DEFER_FIRST_MAX=1048576
done; done
