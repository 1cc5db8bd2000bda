# Round-2 A/B runs on the GPU box (via gpurun): every line is one bench.py run of one workload, headline only.
#   tools/ab_r02.sh > gpurun_out/ab_r02.txt
set -e
one() { # label, workload, env...
  label=$1; w=$2; shift 2
  env "$@" timeout -k 10 200 python bench.py --workload $w --no-sub --no-cpu-baseline --steps ${STEPS:-200} 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('%-78s %s  frame %.5f ms  %9.1f M ent/s  dominant %.5f ms  frac %.4f  expand stage %.5f ms' % ('$label', '$w', d['ms_per_step'], d['value'], r['kernel_ms'], r['frac'], r['expand_stage_ms_calibration']))"
}
echo "# cfg3 (16M flat x 5 frusta): planes grouped by normal (cull_planned) vs every plane tested on its own (round 1)"
one "cull plan (default)" cfg3 A=1
one "YH_NO_CULL_PLAN=1" cfg3 YH_NO_CULL_PLAN=1
echo "# cfg3: packed f32 (clang SLP / loop vectorizers left on: v_pk_mul_f32 / v_pk_add_f32 in Mat4.mul and the plane tests)"
one "vectorizers on, cull plan" cfg3 YUME_HIP_LIB=yume_amd/csrc/ab/libyume_hip_vectorized.so
one "vectorizers on, YH_NO_CULL_PLAN=1" cfg3 YUME_HIP_LIB=yume_amd/csrc/ab/libyume_hip_vectorized.so YH_NO_CULL_PLAN=1
echo "# cfg3: read-once inputs streamed past the caches (nontemporal loads, default from 5M flat entities) vs ordinary loads"
one "streamed inputs (default)" cfg3 A=1
one "ordinary loads (YH_BIG_FLAT=4000000000)" cfg3 YH_BIG_FLAT=4000000000
echo "# cfg5 / cfg4 / cfg2 with the vectorizers on, for the record"
one "default build" cfg5 STEPS=40 A=1
one "vectorizers on" cfg5 YUME_HIP_LIB=yume_amd/csrc/ab/libyume_hip_vectorized.so
one "default build" cfg4 A=1
one "vectorizers on" cfg4 YUME_HIP_LIB=yume_amd/csrc/ab/libyume_hip_vectorized.so
STEPS=1600 one "default build" cfg2 A=1
STEPS=1600 one "vectorizers on" cfg2 YUME_HIP_LIB=yume_amd/csrc/ab/libyume_hip_vectorized.so
