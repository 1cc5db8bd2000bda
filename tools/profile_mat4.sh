# rocprofv3 evidence for the literal-rule path (Mat4 columns: what the flecs drop-in runs), frames with every row dirty
# and with nothing dirty, change tracking on: kernel trace, then FETCH_SIZE and WRITE_SIZE passes.  Run via gpurun;
# tools/summarize_mat4.py turns gpurun_out/prof_mat4_* into profiles/r02/mat4_literal_rule_12m5.txt.
set -e
export TMPDIR=/tmp
N=${1:-12500000}; L=${2:-8}
for M in dirty clean; do
  OUT=$PWD/gpurun_out/prof_mat4_$M; rm -rf $OUT; mkdir -p $OUT
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/mat4_bench.py $N $L $M > $OUT/run.txt 2> $OUT/trace.err
  timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/mat4_bench.py $N $L $M > $OUT/fetch.txt 2> $OUT/fetch.err
  timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/mat4_bench.py $N $L $M > $OUT/write.txt 2> $OUT/write.err
done
echo done
