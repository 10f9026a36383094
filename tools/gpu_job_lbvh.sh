#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
timeout 900 python3 -m pytest tests/test_gpu_bvh.py -q -x > gpurun_out/pytest_bvh.txt 2>&1; grep -E "passed|failed|error|Error|assert" gpurun_out/pytest_bvh.txt | tail -8
for b in lbvh gpu sah; do
  python3 bench.py --scene hard --bvh $b --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hard $b', 'ms', round(d['ms_per_step'],2), 'kernel', round(d['roofline']['kernel_ms'],2), d['config']['bvh'])"
done
EVPLP_BVH_BUILDER=gpu timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_end_to_end.py -q -x > gpurun_out/pytest_gpubvh.txt 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_gpubvh.txt | tail -3
