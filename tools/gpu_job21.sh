#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2t; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_a.log 2>&1; echo "rc=$?" >> $O/pytest_a.log; tail -2 $O/pytest_a.log
for lib in "" _gw5 _gw7 _gw8; do for sc in hard easy; do
EVPLP_LIB=$PWD/evplp_amd/lib/libevplp_hip$lib.so timeout 600 python3 bench.py --steps 5 --warmup 1 --scene $sc --no-cpu-baseline --no-extras > $O/b.jsonl 2> $O/b.err
python3 -c "
import json
d=json.loads(open('$O/b.jsonl').read().strip().splitlines()[-1]); print('lib=$lib $sc kernel_ms',round(d['roofline']['kernel_ms'],2))"
done; done
