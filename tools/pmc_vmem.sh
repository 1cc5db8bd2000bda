# VMEM issue-side counters of the frame's kernels (one pass: 8 SQ counters).  usage (on the GPU box): bash tools/pmc_vmem.sh cfg5
export TMPDIR=/tmp
W=${1:-cfg5}
OUT=$PWD/gpurun_out/pmc_vmem_$W
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL --output-format csv -d $OUT -- python3 bench.py --workload $W --steps 6 --warmup 2 --no-cpu-baseline --no-sub --no-kernel-events > $OUT/bench.json 2> $OUT/err.txt || exit 1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    if not (k.startswith("k_expand") or k.startswith("k_count") or "k_frame" in k): continue
    print(k, "launches", len(next(iter(c.values()))))
    for name, v in sorted(c.items()):
        print("    %-32s mean %.4g" % (name, sum(v) / len(v)))
PY
