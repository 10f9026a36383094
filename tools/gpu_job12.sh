#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2k; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log
tail -3 $O/pytest_all.log
for sc in hard easy; do
timeout 600 python3 bench.py --steps 10 --warmup 2 --scene $sc --no-cpu-baseline --no-extras > $O/bench_ir_$sc.jsonl 2> $O/bench_ir_$sc.err
python3 -c "
import json,sys
d=json.loads(open('$O/bench_ir_$sc.jsonl').read().strip().splitlines()[-1]); print('$sc value',round(d['value']),'ms',round(d['ms_per_step'],2),'kernel_ms',round(d['roofline']['kernel_ms'],2),'frac',round(d['roofline']['frac'],4), 'nominal', round(d['roofline']['frac_nominal_pairs'],4))"
done
