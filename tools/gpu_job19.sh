#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2r; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bench.py tests/test_gpu_group.py -m gpu -x -q > $O/pytest_a.log 2>&1; echo "rc=$?" >> $O/pytest_a.log; tail -4 $O/pytest_a.log
for ct in 0.5 1 2 3 5; do for sc in hard easy; do
EVPLP_SAH_CT=$ct timeout 600 python3 bench.py --steps 5 --warmup 1 --scene $sc --no-cpu-baseline --no-extras > $O/b.jsonl 2> $O/b.err
python3 -c "
import json
d=json.loads(open('$O/b.jsonl').read().strip().splitlines()[-1]); print('ct $ct $sc kernel_ms',round(d['roofline']['kernel_ms'],2), 'nodes', d['config']['bvh']['nodes'], 'leaves', d['config']['bvh']['leaves'], 'depth', d['config']['bvh']['depth'])"
done; done
