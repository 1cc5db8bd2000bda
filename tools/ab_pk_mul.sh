# Same-box A/B: m4_mul with packed f32 instructions (ab/libyume_hip_pk.so, -DYH_PK_MUL) against the scalar build.
one() { label=$1; w=$2; shift 2
  env "$@" timeout -k 10 200 python bench.py --workload $w --no-sub --no-cpu-baseline --steps ${STEPS:-400} 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('%-10s %s frame %.5f ms  dominant %.5f ms (%s)' % ('$label', '$w', d['ms_per_step'], r['kernel_ms'], r['kernel_symbol'][:40]))"
}
for rep in 1 2; do
for spec in ${SPECS:-"cfg2 2000" "cfg4 400" "cfg3 400" "cfg5 60"}; do
  set -- $spec; export STEPS=$2
  one scalar $1 A=1
  one packed $1 YUME_HIP_LIB=yume_amd/csrc/ab/libyume_hip_pk.so
done; done
