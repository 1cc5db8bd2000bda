# Round-2 rocprofv3 evidence (run on the GPU box via gpurun).  Every profiler run has a time limit of its own; a failed
# step ends the script.  Outputs: gpurun_out/prof_<w>/ (tools/summarize_prof.py r02 turns them into profiles/r02/).
set -e
export TMPDIR=/tmp
ROOT=$PWD
# (1) the default command itself: the line the driver times, under --kernel-trace --stats
OUT=$ROOT/gpurun_out/prof_default; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/trace.err
# (2) per workload: kernel trace of a longer run + the two HBM traffic passes
for W in cfg5 cfg2 cfg3 cfg4; do
  bash tools/profile.sh $W > gpurun_out/profile_$W.log 2>&1
done
# (3) cfg3: vector instructions issued with and without the shared-normal cull plan
for V in plan noplan; do
  OUT=$ROOT/gpurun_out/prof_cfg3_valu_$V; rm -rf $OUT; mkdir -p $OUT
  if [ $V = noplan ]; then export YH_NO_CULL_PLAN=1; else unset YH_NO_CULL_PLAN; fi
  timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc -- python3 bench.py --workload cfg3 --steps 20 --warmup 5 --no-cpu-baseline --no-sub --no-kernel-events > $OUT/bench.json 2> $OUT/err.txt
done
unset YH_NO_CULL_PLAN
echo done
