#!/bin/bash
# A/B of the BVH builders on the gather (both scenes), then the parity suites under the SBVH.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
for sc in hard easy; do for b in sah sbvh; do
  python3 bench.py --scene $sc --bvh $b --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sc $b', 'ms', round(d['ms_per_step'],2), 'kernel', round(d['roofline']['kernel_ms'],2), d['config']['bvh'])"
done; done
EVPLP_SBVH_DUP=1.0 python3 bench.py --scene hard --bvh sbvh --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hard sbvh dup1.0', 'ms', round(d['ms_per_step'],2), 'kernel', round(d['roofline']['kernel_ms'],2), d['config']['bvh'])"
EVPLP_BVH_BUILDER=sbvh timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -q -x > gpurun_out/pytest_sbvh.txt 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_sbvh.txt | tail -3
