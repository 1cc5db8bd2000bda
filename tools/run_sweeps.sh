set -e
timeout -k 10 300 python -u -m pytest tests -m gpu -q -x --timeout 180 --timeout-method thread 2>&1 | tail -2
timeout -k 10 300 python bench.py --steps 200 --warmup 20 > gpurun_out/bench_cfg2.json 2> gpurun_out/bench_cfg2.err
cat gpurun_out/bench_cfg2.json
