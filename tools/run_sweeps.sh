set -e
timeout -k 10 300 python -u -m pytest tests -m gpu -q -x --timeout 180 --timeout-method thread 2>&1 | tail -2
timeout -k 10 300 python -u tools/sweep.py cfg2s flat1 cfg3 2>&1 | grep -v amdgpu.ids
