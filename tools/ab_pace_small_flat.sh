# Same-box A/B: paced world stores (in-tree) against unpaced ones (ab/libyume_hip_pace0.so = s_nop 0) on FLAT scenes of
# every size class: cache-resident (1M, 4M: k_frame_trs / _x), streamed (16M cfg3-like with 1 and 5 frusta, 64M).
one() { label=$1; w=$2; n=$3; shift 3
  env "$@" timeout -k 10 200 python bench.py --workload $w --entities $n --no-sub --no-cpu-baseline --steps ${STEPS:-200} 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('%-28s %s n=%-10s frame %.5f ms  dominant %.5f ms (%s)' % ('$label', '$w', '$n', d['ms_per_step'], r['kernel_ms'], r['kernel_symbol'][:34]))"
}
for rep in 1 2; do
for spec in "cfg3 1000000 2000" "cfg3 4000000 800" "cfg5 1000000 2000" "cfg5 4000000 800" "cfg5 8000000 400" "cfg5 64000000 100"; do
  set -- $spec; export STEPS=$3
  one "in-tree" $1 $2 A=1
  one "unpaced (s_nop 0)" $1 $2 YUME_HIP_LIB=yume_amd/csrc/ab/libyume_hip_pace0.so
done; done
