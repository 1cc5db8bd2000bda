# Round-3 rocprofv3 evidence (run on the GPU box via gpurun).  Every profiler run has a time limit of its own; a failed
# step ends the script.  Outputs: gpurun_out/prof_<w>/ (tools/summarize_prof.py r03 turns them into profiles/r03/).
set -e
export TMPDIR=/tmp
ROOT=$PWD
rm -rf $ROOT/gpurun_out/prof_*
# (1) the default command itself: the line the driver times, under --kernel-trace --stats
OUT=$ROOT/gpurun_out/prof_default; mkdir -p $OUT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/trace.err
# (2) per workload: kernel trace of a longer run + the two HBM traffic passes
for W in cfg5 cfg2 cfg3 cfg4; do
  bash tools/profile.sh $W > gpurun_out/profile_$W.log 2>&1
done
# (3) the draw-list consumers (list order, bucketed by 16 and by 4096 keys) at 64M instances
OUT=$ROOT/gpurun_out/prof_buckets; mkdir -p $OUT
for K in 16 4096; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace$K -- python3 tools/bucket_bench.py 64e6 $K > $OUT/run$K.txt 2> $OUT/trace$K.err
done
echo done
