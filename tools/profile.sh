# Collect rocprofv3 evidence on the GPU box (run via gpurun); outputs under gpurun_out/prof_*
set -e
export TMPDIR=/tmp
W=${1:-cfg2}
OUT=$PWD/gpurun_out/prof_$W
rm -rf $OUT; mkdir -p $OUT
# per-kernel time
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload $W --steps ${PROF_STEPS:-400} --warmup 20 --no-cpu-baseline --no-sub > $OUT/bench_trace.json 2> $OUT/trace.err
# HBM traffic counters, one pass each (FETCH_SIZE uses 3 TCC slots, WRITE_SIZE 2)
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --no-sub --no-kernel-events > $OUT/bench_fetch.json 2> $OUT/fetch.err
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --no-sub --no-kernel-events > $OUT/bench_write.json 2> $OUT/write.err
find $OUT -name "*.csv" | head -20
